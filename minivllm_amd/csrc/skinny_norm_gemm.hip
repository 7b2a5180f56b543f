// Residual-add + RMSNorm fused into the weight-streaming GEMM of a decode step, for weights
// that already carry the norm's scale vector (W' = W * gamma along K, folded once at load time):
//
//     r   = x + residual                      (rounded to the storage dtype, as add_rms_norm does)
//     out = rsqrt(mean(r^2) + eps) * (r . W'^T)            [optionally through SwiGLU]
//
// At decode sizes the norm kernel in front of a projection is pure launch latency (a few KB of
// data, ~5 us with its launch gap, twice per layer).  Because the per-row factor commutes with
// the matrix product, nothing has to be normalised BEFORE the k-loop:
//
//  * the A fragment of every k-step is formed in registers as x + residual (two 16-byte loads);
//  * the sum of squares comes from the matrix core as well: the A fragment of lane (m, q) is
//    bit-for-bit the B fragment of r^T, so one extra MFMA per k-step accumulates the Gram
//    matrix r r^T, whose diagonal is sum_k r[m][k]^2 - no VALU reduction, no extra pass;
//  * the epilogue multiplies the split-K-reduced sums by rsqrt(diag / K + eps);
//  * workgroup 0 also writes r (the new residual stream) to `res_out`, a buffer distinct from
//    the inputs, which the other workgroups are still reading.
#include "common.h"

using namespace mvl;

namespace {

constexpr int kUnroll = 4;       // 4 k-steps in flight keep the kernel under 96 VGPRs (>= 2 workgroups per CU)

template <int CODE>
__device__ __forceinline__ u32x4_t add_round(u32x4_t a, u32x4_t b) {
    float fa[8], fb[8];
    unpack8<CODE>(a, fa);
    unpack8<CODE>(b, fb);
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] += fb[j];
    return pack8<CODE>(fa);
}

template <int CODE, int EPI, int WAVES, bool HAS_RES>
__global__ __launch_bounds__(WAVES * 64) void skinny_norm_gemm_kernel(
        const uint16_t* __restrict__ x, long x_stride, const uint16_t* __restrict__ res_in,
        long ri_stride, uint16_t* __restrict__ res_out, long ro_stride, float eps,
        const uint16_t* __restrict__ w, const uint16_t* __restrict__ bias,
        uint16_t* __restrict__ out, long o_stride, int M, int K, int up_row_offset) {
    using DT = DType<CODE>;
    using vec8 = typename DT::vec8;
    constexpr int NB = EPI == 1 ? 2 : 1;
    __shared__ float red[WAVES][NB][16][17];
    __shared__ float diag[WAVES][16];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c16 = lane & 15, q4 = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int kslice = K / WAVES;
    const long koff = (long)wv * kslice + 8 * q4;
    const int row = min(c16, M - 1);
    const uint16_t* wp0 = w + (long)(n0 + c16) * K + koff;
    const uint16_t* wp1 = w + (long)(up_row_offset + n0 + c16) * K + koff;
    const uint16_t* xp = x + (long)row * x_stride + koff;
    const uint16_t* rp = HAS_RES ? res_in + (long)row * ri_stride + koff : nullptr;
    // workgroup 0 publishes r in a pass of its own, so that the k-loop carries no stores
    if (HAS_RES && res_out != nullptr && blockIdx.x == 0) {
        const int nchunk = K >> 3;
        for (int m = 0; m < M; ++m)
            for (int c = threadIdx.x; c < nchunk; c += WAVES * 64)
                reinterpret_cast<u32x4_t*>(res_out + m * ro_stride)[c] = add_round<CODE>(
                    reinterpret_cast<const u32x4_t*>(x + m * x_stride)[c],
                    reinterpret_cast<const u32x4_t*>(res_in + m * ri_stride)[c]);
    }

    f32x4_t acc[NB], gram = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    int k = 0;
    for (; k + 32 * kUnroll <= kslice; k += 32 * kUnroll) {
        u32x4_t bw[NB][kUnroll], ax[kUnroll], ar[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            bw[0][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wp0 + k + 32 * u));
            if (NB == 2)
                bw[NB - 1][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wp1 + k + 32 * u));
            ax[u] = *reinterpret_cast<const u32x4_t*>(xp + k + 32 * u);
            if (HAS_RES) ar[u] = *reinterpret_cast<const u32x4_t*>(rp + k + 32 * u);
        }
        __builtin_amdgcn_sched_barrier(0);          // all loads issued before the first use
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const u32x4_t r = HAS_RES ? add_round<CODE>(ax[u], ar[u]) : ax[u];
            const vec8 a = __builtin_bit_cast(vec8, r);
            gram = DT::mfma16(a, a, gram);
#pragma unroll
            for (int b = 0; b < NB; ++b)
                acc[b] = DT::mfma16(a, __builtin_bit_cast(vec8, bw[b][u]), acc[b]);
        }
    }
    for (; k < kslice; k += 32) {
        u32x4_t r = *reinterpret_cast<const u32x4_t*>(xp + k);
        if (HAS_RES) r = add_round<CODE>(r, *reinterpret_cast<const u32x4_t*>(rp + k));
        const vec8 a = __builtin_bit_cast(vec8, r);
        gram = DT::mfma16(a, a, gram);
        acc[0] = DT::mfma16(a, __builtin_bit_cast(vec8, *reinterpret_cast<const u32x4_t*>(wp0 + k)), acc[0]);
        if (NB == 2)
            acc[NB - 1] = DT::mfma16(a, __builtin_bit_cast(vec8, *reinterpret_cast<const u32x4_t*>(wp1 + k)),
                                     acc[NB - 1]);
    }

    // split-K reduction; accumulator layout: col n = lane & 15, row m = 4 q + r.  The diagonal of
    // the Gram tile (row == col) sits in the lanes with 4 q <= c16 < 4 q + 4.
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv][b][4 * q4 + r][c16] = acc[b][r];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (c16 == 4 * q4 + r) diag[wv][c16] = gram[r];
    __syncthreads();
    if (threadIdx.x < 256) {
        const int m = threadIdx.x >> 4, n = threadIdx.x & 15;
        if (m < M) {
            float s[NB], ss = 0.f;
#pragma unroll
            for (int ww = 0; ww < WAVES; ++ww) ss += diag[ww][m];
            const float rstd = rsqrtf(ss / (float)K + eps);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                s[b] = 0.f;
#pragma unroll
                for (int ww = 0; ww < WAVES; ++ww) s[b] += red[ww][b][m][n];
                s[b] *= rstd;
            }
            float r;
            if (EPI == 1) {
                if (bias != nullptr) {
                    s[0] += DT::to_float(bias[n0 + n]);
                    s[NB - 1] += DT::to_float(bias[up_row_offset + n0 + n]);
                }
                r = s[0] / (1.f + __expf(-s[0])) * s[NB - 1];
            } else {
                r = s[0] + (bias != nullptr ? DT::to_float(bias[n0 + n]) : 0.f);
            }
            out[(long)m * o_stride + n0 + n] = DT::from_float(r);
        }
    }
}

template <int CODE, int EPI, bool HAS_RES>
int launch(const void* x, long xs, const void* ri, long ris, void* ro, long ros, float eps,
           const void* w, const void* bias, void* out, long os, int M, int tiles, int K, int up_off,
           hipStream_t st) {
    auto X = (const uint16_t*)x; auto RI = (const uint16_t*)ri; auto RO = (uint16_t*)ro;
    auto W = (const uint16_t*)w; auto B = (const uint16_t*)bias; auto O = (uint16_t*)out;
    if (K % 256 == 0)
        skinny_norm_gemm_kernel<CODE, EPI, 8, HAS_RES><<<tiles, 8 * 64, 0, st>>>(
            X, xs, RI, ris, RO, ros, eps, W, B, O, os, M, K, up_off);
    else if (K % 128 == 0)
        skinny_norm_gemm_kernel<CODE, EPI, 4, HAS_RES><<<tiles, 4 * 64, 0, st>>>(
            X, xs, RI, ris, RO, ros, eps, W, B, O, os, M, K, up_off);
    else
        return -1;
    return MVL_CHECK_LAUNCH();
}

template <int CODE, int EPI>
int launch_r(const void* x, long xs, const void* ri, long ris, void* ro, long ros, float eps,
             const void* w, const void* bias, void* out, long os, int M, int tiles, int K, int up_off,
             hipStream_t st) {
    return ri != nullptr
        ? launch<CODE, EPI, true>(x, xs, ri, ris, ro, ros, eps, w, bias, out, os, M, tiles, K, up_off, st)
        : launch<CODE, EPI, false>(x, xs, ri, ris, ro, ros, eps, w, bias, out, os, M, tiles, K, up_off, st);
}

}  // namespace

// w: the projection weights with the norm's scale vector folded in.  res_in may be null (no
// residual yet: r = x); res_out may be null (r is not wanted) and must not alias x or res_in.
// Other arguments as mvl_skinny_gemm; M <= 16.
extern "C" int mvl_skinny_norm_gemm(const void* x, long x_stride, const void* res_in, long ri_stride,
                                    void* res_out, long ro_stride, float eps, const void* w,
                                    const void* bias, void* out, long o_stride, int M, int N, int K,
                                    int epilogue, int dtype, void* stream) {
    if (M == 0) return 0;
    if (M > 16 || K % 128 != 0) return -1;
    if (res_out != nullptr && (res_out == x || res_out == res_in)) return -1;
    auto st = (hipStream_t)stream;
    if (epilogue == 0) {
        if (N % 16 != 0) return -1;
        return dtype == 0
            ? launch_r<0, 0>(x, x_stride, res_in, ri_stride, res_out, ro_stride, eps, w, bias, out, o_stride, M, N / 16, K, 0, st)
            : launch_r<1, 0>(x, x_stride, res_in, ri_stride, res_out, ro_stride, eps, w, bias, out, o_stride, M, N / 16, K, 0, st);
    }
    if (N % 32 != 0) return -1;
    const int inter = N / 2;
    return dtype == 0
        ? launch_r<0, 1>(x, x_stride, res_in, ri_stride, res_out, ro_stride, eps, w, bias, out, o_stride, M, inter / 16, K, inter, st)
        : launch_r<1, 1>(x, x_stride, res_in, ri_stride, res_out, ro_stride, eps, w, bias, out, o_stride, M, inter / 16, K, inter, st);
}
