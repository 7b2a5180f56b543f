// Full-matrix ("naive") causal attention for short sequences: one workgroup per (sequence,
// head) materialises the whole L x L score matrix in LDS, soft-maxes it row by row and
// multiplies by V.  It exists as the comparison baseline of the prefill benchmark, like the
// reference's naive Triton kernel (reference: benchmark_prefilling.py:58-170), which holds the
// score matrix of one sequence in on-chip memory and therefore only handles L <= 64..128.
// Plain VALU dot products on purpose: this is the thing the MFMA flash kernel is compared to.
#include "common.h"

using namespace mvl;

namespace {

constexpr int kMaxLen = 128;

template <int CODE>
__global__ __launch_bounds__(256) void naive_attention_kernel(
        const uint16_t* __restrict__ q, long qs_t, long qs_h, const uint16_t* __restrict__ k,
        const uint16_t* __restrict__ v, long ks_t, long ks_h, uint16_t* __restrict__ out,
        long os_t, long os_h, const int* __restrict__ cu, float scale, int Hq, int Hkv, int D) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* S = reinterpret_cast<float*>(smem);            // [L][L + 1]
    const int b = blockIdx.y, h = blockIdx.x;
    const int start = cu[b], L = cu[b + 1] - start;
    if (L <= 0 || L > kMaxLen) return;
    const int kvh = h / (Hq / Hkv);
    const int ld = L + 1;
    const int nch = D / 8;
    // scores: thread t takes pairs (i, j), j <= i
    for (int p = threadIdx.x; p < L * L; p += blockDim.x) {
        const int i = p / L, j = p - i * L;
        float acc = -INFINITY;
        if (j <= i) {
            const u32x4_t* qr = reinterpret_cast<const u32x4_t*>(q + (long)(start + i) * qs_t + h * qs_h);
            const u32x4_t* kr = reinterpret_cast<const u32x4_t*>(k + (long)(start + j) * ks_t + kvh * ks_h);
            acc = 0.f;
            for (int c = 0; c < nch; ++c) {
                float a[8], bb[8];
                unpack8<CODE>(qr[c], a);
                unpack8<CODE>(kr[c], bb);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += a[e] * bb[e];
            }
            acc *= scale;
        }
        S[i * ld + j] = acc;
    }
    __syncthreads();
    // softmax: one thread per row
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        float m = -INFINITY;
        for (int j = 0; j <= i; ++j) m = fmaxf(m, S[i * ld + j]);
        float sum = 0.f;
        for (int j = 0; j <= i; ++j) {
            const float e = __expf(S[i * ld + j] - m);
            S[i * ld + j] = e;
            sum += e;
        }
        const float inv = 1.f / sum;
        for (int j = 0; j <= i; ++j) S[i * ld + j] *= inv;
    }
    __syncthreads();
    // O = P V: thread takes (row i, 8-wide chunk of d)
    for (int p = threadIdx.x; p < L * nch; p += blockDim.x) {
        const int i = p / nch, c = p - i * nch;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j <= i; ++j) {
            float vv[8];
            unpack8<CODE>(reinterpret_cast<const u32x4_t*>(v + (long)(start + j) * ks_t + kvh * ks_h)[c], vv);
            const float pj = S[i * ld + j];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += pj * vv[e];
        }
        reinterpret_cast<u32x4_t*>(out + (long)(start + i) * os_t + h * os_h)[c] = pack8<CODE>(acc);
    }
}

}  // namespace

extern "C" int mvl_naive_attention(const void* q, long qs_t, long qs_h, const void* k, const void* v,
                                   long ks_t, long ks_h, void* out, long os_t, long os_h,
                                   const int* cu, float scale, int B, int Hq, int Hkv, int D,
                                   int max_len, int dtype, void* stream) {
    if (B == 0) return 0;
    if (max_len > kMaxLen || D % 8 != 0 || Hq % Hkv != 0) return -1;
    const size_t lds = (size_t)max_len * (max_len + 1) * sizeof(float);
    dim3 grid(Hq, B);
    auto st = (hipStream_t)stream;
    if (dtype == 0)
        naive_attention_kernel<0><<<grid, 256, lds, st>>>(
            (const uint16_t*)q, qs_t, qs_h, (const uint16_t*)k, (const uint16_t*)v, ks_t, ks_h,
            (uint16_t*)out, os_t, os_h, cu, scale, Hq, Hkv, D);
    else
        naive_attention_kernel<1><<<grid, 256, lds, st>>>(
            (const uint16_t*)q, qs_t, qs_h, (const uint16_t*)k, (const uint16_t*)v, ks_t, ks_h,
            (uint16_t*)out, os_t, os_h, cu, scale, Hq, Hkv, D);
    return MVL_CHECK_LAUNCH();
}
