// Bandwidth-bound kernels: RMSNorm, fused residual-add + RMSNorm, SwiGLU gate, fused
// per-head q/k RMSNorm + RoPE + KV-cache store, standalone KV-cache store.
//
// They replace the reference's five torch.compile-fused ops and its Triton store kernel
// (reference: src/myvllm/layers/layernorm.py:9-22, rotary_embedding.py:74-83,
// activation.py:15-18, attention.py:7-108).  All of them move 16 bytes per lane per access
// and accumulate in fp32.
#include "common.h"

using namespace mvl;

// ============================================================== RMSNorm
// One workgroup per row; the row stays in registers between the reduction and the scale.
// ADD: x <- x + residual is rounded to the storage dtype first, written to `residual`, and
// the normalised result goes to `out` (which may alias x).
template <int CODE, bool ADD, int MAXC>
__global__ void rms_norm_kernel(const uint16_t* __restrict__ x, uint16_t* residual,
                                const uint16_t* __restrict__ weight, uint16_t* out,
                                int hidden, long x_stride, long r_stride, long o_stride,
                                float eps) {
    __shared__ float scratch[16];
    const long row = blockIdx.x;
    const int nchunk = hidden >> 3;
    const u32x4_t* xr = reinterpret_cast<const u32x4_t*>(x + row * x_stride);
    u32x4_t* rr = ADD ? reinterpret_cast<u32x4_t*>(residual + row * r_stride) : nullptr;
    // the weights are fetched together with the row: their latency overlaps the row's and the
    // reduction instead of following the barrier (the kernel is a chain of memory round trips)
    const u32x4_t* wr = reinterpret_cast<const u32x4_t*>(weight);
    u32x4_t wv[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = threadIdx.x + i * blockDim.x;
        wv[i] = c < nchunk ? wr[c] : u32x4_t{0, 0, 0, 0};
    }
    float v[MAXC][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = threadIdx.x + i * blockDim.x;
        if (c < nchunk) {
            unpack8<CODE>(xr[c], v[i]);
            if (ADD) {
                float r[8];
                unpack8<CODE>(rr[c], r);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[i][j] = DType<CODE>::to_float(DType<CODE>::from_float(v[i][j] + r[j]));
                rr[c] = pack8<CODE>(v[i]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += v[i][j] * v[i][j];
        }
    }
    ss = block_sum(ss, scratch);
    const float rstd = rsqrtf(ss / (float)hidden + eps);
    u32x4_t* orow = reinterpret_cast<u32x4_t*>(out + row * o_stride);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = threadIdx.x + i * blockDim.x;
        if (c < nchunk) {
            float w[8], o[8];
            unpack8<CODE>(wv[i], w);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = v[i][j] * rstd * w[j];
            orow[c] = pack8<CODE>(o);
        }
    }
}

template <int CODE, bool ADD>
static int launch_rms(const void* x, void* residual, const void* weight, void* out, long rows,
                      int hidden, long xs, long rs, long os, float eps, hipStream_t st) {
    if (hidden % 8 != 0 || hidden > 8 * 1024 * 4) return -1;
    if (rows == 0) return 0;
    const int nchunk = hidden / 8;
    int threads = 64;
    while (threads < nchunk && threads < 1024) threads <<= 1;
    const int per = (nchunk + threads - 1) / threads;
    auto X = (const uint16_t*)x; auto R = (uint16_t*)residual;
    auto W = (const uint16_t*)weight; auto O = (uint16_t*)out;
    dim3 grid((unsigned)rows);
    if (per <= 1)
        rms_norm_kernel<CODE, ADD, 1><<<grid, threads, 0, st>>>(X, R, W, O, hidden, xs, rs, os, eps);
    else if (per <= 2)
        rms_norm_kernel<CODE, ADD, 2><<<grid, threads, 0, st>>>(X, R, W, O, hidden, xs, rs, os, eps);
    else
        rms_norm_kernel<CODE, ADD, 4><<<grid, threads, 0, st>>>(X, R, W, O, hidden, xs, rs, os, eps);
    return MVL_CHECK_LAUNCH();
}

extern "C" int mvl_rms_norm(const void* x, const void* weight, void* out, long rows, int hidden,
                            long x_stride, long o_stride, float eps, int dtype, void* stream) {
    auto st = (hipStream_t)stream;
    return dtype == 0
        ? launch_rms<0, false>(x, nullptr, weight, out, rows, hidden, x_stride, 0, o_stride, eps, st)
        : launch_rms<1, false>(x, nullptr, weight, out, rows, hidden, x_stride, 0, o_stride, eps, st);
}

extern "C" int mvl_add_rms_norm(const void* x, void* residual, const void* weight, void* out,
                                long rows, int hidden, long x_stride, long r_stride,
                                long o_stride, float eps, int dtype, void* stream) {
    auto st = (hipStream_t)stream;
    return dtype == 0
        ? launch_rms<0, true>(x, residual, weight, out, rows, hidden, x_stride, r_stride, o_stride, eps, st)
        : launch_rms<1, true>(x, residual, weight, out, rows, hidden, x_stride, r_stride, o_stride, eps, st);
}

// ============================================================ SiluAndMul
// out[r, c] = silu(x[r, c]) * x[r, inter + c]
template <int CODE>
__global__ void silu_mul_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out,
                                long rows, int inter, long x_stride, long o_stride) {
    const int cpr = inter >> 3;
    const long total = rows * cpr;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / cpr;
        const int c = (int)(idx - r * cpr);
        const u32x4_t* xr = reinterpret_cast<const u32x4_t*>(x + r * x_stride);
        float a[8], b[8], o[8];
        unpack8<CODE>(xr[c], a);
        unpack8<CODE>(xr[cpr + c], b);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = a[j] / (1.f + __expf(-a[j])) * b[j];
        reinterpret_cast<u32x4_t*>(out + r * o_stride)[c] = pack8<CODE>(o);
    }
}

// fp32 rows (the precision of the reference's own activation micro-benchmark,
// reference: src/myvllm/layers/activation.py:20-37): 16 bytes = 4 elements per lane
__global__ void silu_mul_f32_kernel(const float* __restrict__ x, float* __restrict__ out,
                                    long rows, int inter, long x_stride, long o_stride) {
    const int cpr = inter >> 2;
    const long total = rows * cpr;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / cpr;
        const int c = (int)(idx - r * cpr);
        const f32x4_t* xr = reinterpret_cast<const f32x4_t*>(x + r * x_stride);
        const f32x4_t a = __builtin_nontemporal_load(xr + c);
        const f32x4_t b = __builtin_nontemporal_load(xr + cpr + c);
        f32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = a[j] / (1.f + __expf(-a[j])) * b[j];
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4_t*>(out + r * o_stride) + c);
    }
}

// dtype: 0 = bf16, 1 = fp16, 2 = fp32
extern "C" int mvl_silu_and_mul(const void* x, void* out, long rows, int inter, long x_stride,
                                long o_stride, int dtype, void* stream) {
    if (inter % (dtype == 2 ? 4 : 8) != 0) return -1;
    if (rows == 0) return 0;
    const long total = rows * (inter / (dtype == 2 ? 4 : 8));
    const int threads = 256;
    long blocks = (total + threads - 1) / threads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    auto st = (hipStream_t)stream;
    if (dtype == 2) {
        if ((x_stride | o_stride) % 4 != 0) return -1;
        silu_mul_f32_kernel<<<(unsigned)blocks, threads, 0, st>>>((const float*)x, (float*)out, rows,
                                                                  inter, x_stride, o_stride);
        return MVL_CHECK_LAUNCH();
    }
    if (dtype == 0)
        silu_mul_kernel<0><<<(unsigned)blocks, threads, 0, st>>>((const uint16_t*)x, (uint16_t*)out,
                                                                 rows, inter, x_stride, o_stride);
    else
        silu_mul_kernel<1><<<(unsigned)blocks, threads, 0, st>>>((const uint16_t*)x, (uint16_t*)out,
                                                                 rows, inter, x_stride, o_stride);
    return MVL_CHECK_LAUNCH();
}

// ========================================= fused q/k RMSNorm + RoPE + KV store
// qkv: [T, (Hq + 2 Hkv) * D], updated in place for the q and k heads.  One head-row
// (token, head) is handled by LPR = D/16 lanes; a lane holds elements [8j, 8j+8) of the first
// half and the matching elements of the second half (16-byte accesses), so the NeoX rotation pairs
// (i, i + D/2) sit in the same lane.  k (after norm + rotation) and v rows are also
// scattered to the paged cache at slot_mapping[t] (logical slot = block * block_size +
// offset; the cache's physical layout is given by strides).
// four floats -> four OCP fp8 (e4m3) bytes in one word, saturating at the format's +-448
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
    auto sat = [](float x) { return fminf(fmaxf(x, -448.f), 448.f); };
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(sat(a), sat(b), 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(sat(c), sat(d), w, true);
    return (uint32_t)w;
}

template <int CODE, int D, bool KV8>
__global__ void qk_norm_rope_store_kernel(
        uint16_t* qkv, long qkv_stride, const int* __restrict__ positions,
        const float* __restrict__ cos_sin, const uint16_t* __restrict__ qw,
        const uint16_t* __restrict__ kw, float eps, int Hq, int Hkv,
        uint16_t* k_cache, uint16_t* v_cache, long c_stride_b, long c_stride_t, long c_stride_h,
        int block_size, const int* __restrict__ slot_mapping, long total_rows, int first_head,
        int write_back) {
    // 16-byte accesses: a lane holds elements [8j, 8j+8) of the first half of its row and the
    // matching 8 of the second half
    constexpr int LPR = D / 16;
    constexpr int RPW = 64 / LPR;                // rows per wave
    const int lane = threadIdx.x & 63;
    const int j = lane % LPR;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long row = wave * RPW + lane / LPR;
    const bool active = row < total_rows;
    // first_head = Hq: only the K and V heads are visited (the Q rows are left to the prefill
    // attention kernel, which transforms them as it loads them)
    const int H = Hq + 2 * Hkv - first_head;
    const long t = active ? row / H : 0;
    const int h = first_head + (active ? (int)(row - t * H) : 0);
    uint16_t* p = qkv + t * qkv_stride + (long)h * D;
    u32x4_t a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    if (active) {
        a = *reinterpret_cast<const u32x4_t*>(p + 8 * j);
        b = *reinterpret_cast<const u32x4_t*>(p + D / 2 + 8 * j);
    }
    const bool is_v = h >= Hq + Hkv;
    const bool is_k = h >= Hq && !is_v;
    float x1[8], x2[8];
    unpack8<CODE>(a, x1);
    unpack8<CODE>(b, x2);
    // every lane takes part in the shuffles; v rows just ignore the result
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += x1[i] * x1[i] + x2[i] * x2[i];
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (active && !is_v) {
        const uint16_t* w = is_k ? kw : qw;
        if (w != nullptr) {
            const float rstd = rsqrtf(ss / (float)D + eps);
            float w1[8], w2[8];
            unpack8<CODE>(*reinterpret_cast<const u32x4_t*>(w + 8 * j), w1);
            unpack8<CODE>(*reinterpret_cast<const u32x4_t*>(w + D / 2 + 8 * j), w2);
#pragma unroll
            for (int i = 0; i < 8; ++i) { x1[i] *= rstd * w1[i]; x2[i] *= rstd * w2[i]; }
        }
        const float* cs = cos_sin + (long)positions[t] * D;
        const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(cs + 8 * j);
        const f32x4_t c1 = *reinterpret_cast<const f32x4_t*>(cs + 8 * j + 4);
        const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(cs + D / 2 + 8 * j);
        const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(cs + D / 2 + 8 * j + 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float cc = i < 4 ? c0[i & 3] : c1[i & 3], sn = i < 4 ? s0[i & 3] : s1[i & 3];
            const float o1 = x1[i] * cc - x2[i] * sn;
            const float o2 = x1[i] * sn + x2[i] * cc;
            x1[i] = o1; x2[i] = o2;
        }
        a = pack8<CODE>(x1);
        b = pack8<CODE>(x2);
        // write_back == 0: the transformed K rows are only needed in the cache (the attention
        // kernel of this step reads them from there), so the pass over qkv is read-only
        if (write_back || !is_k) {
            *reinterpret_cast<u32x4_t*>(p + 8 * j) = a;
            *reinterpret_cast<u32x4_t*>(p + D / 2 + 8 * j) = b;
        }
    }
    if (active && (is_k || is_v) && k_cache != nullptr && slot_mapping != nullptr) {
        const int slot = slot_mapping[t];
        if (slot >= 0) {
            const int blk = slot / block_size, off = slot - blk * block_size;
            const int kvh = is_k ? h - Hq : h - Hq - Hkv;
            const long at = blk * c_stride_b + off * c_stride_t + kvh * c_stride_h;
            if constexpr (KV8) {                 // one byte per element; from the 16-bit values
                uint8_t* dst = reinterpret_cast<uint8_t*>(is_k ? k_cache : v_cache) + at;
                unpack8<CODE>(a, x1);
                unpack8<CODE>(b, x2);
                u32x2_t lo, hi;
                lo[0] = pack_fp8x4(x1[0], x1[1], x1[2], x1[3]);
                lo[1] = pack_fp8x4(x1[4], x1[5], x1[6], x1[7]);
                hi[0] = pack_fp8x4(x2[0], x2[1], x2[2], x2[3]);
                hi[1] = pack_fp8x4(x2[4], x2[5], x2[6], x2[7]);
                *reinterpret_cast<u32x2_t*>(dst + 8 * j) = lo;
                *reinterpret_cast<u32x2_t*>(dst + D / 2 + 8 * j) = hi;
            } else {
                uint16_t* dst = (is_k ? k_cache : v_cache) + at;
                *reinterpret_cast<u32x4_t*>(dst + 8 * j) = a;
                *reinterpret_cast<u32x4_t*>(dst + D / 2 + 8 * j) = b;
            }
        }
    }
}

extern "C" int mvl_qk_norm_rope_store(void* qkv, long qkv_stride, const int* positions,
                                      const float* cos_sin, const void* qw, const void* kw,
                                      float eps, long T, int Hq, int Hkv, int D, void* k_cache,
                                      void* v_cache, long c_stride_b, long c_stride_t,
                                      long c_stride_h, int block_size, const int* slot_mapping,
                                      int dtype, int kv_fp8, int skip_q, int write_back,
                                      void* stream) {
    if (D != 64 && D != 128) return -1;
    if (T == 0) return 0;
    const int first_head = skip_q ? Hq : 0;
    const long rows = T * (Hq + 2 * Hkv - first_head);
    const int rpw = 64 / (D / 16);
    const int threads = 256;
    const long waves = (rows + rpw - 1) / rpw;
    const long blocks = (waves + 3) / 4;
    auto st = (hipStream_t)stream;
#define MVL_LAUNCH_ROPE(CODE, DD, K8)                                                          \
    qk_norm_rope_store_kernel<CODE, DD, K8><<<(unsigned)blocks, threads, 0, st>>>(             \
        (uint16_t*)qkv, qkv_stride, positions, cos_sin, (const uint16_t*)qw,                   \
        (const uint16_t*)kw, eps, Hq, Hkv, (uint16_t*)k_cache, (uint16_t*)v_cache, c_stride_b, \
        c_stride_t, c_stride_h, block_size, slot_mapping, rows, first_head, write_back)
    if (kv_fp8) {
        if (dtype == 0) { if (D == 128) MVL_LAUNCH_ROPE(0, 128, true); else MVL_LAUNCH_ROPE(0, 64, true); }
        else            { if (D == 128) MVL_LAUNCH_ROPE(1, 128, true); else MVL_LAUNCH_ROPE(1, 64, true); }
    } else {
        if (dtype == 0) { if (D == 128) MVL_LAUNCH_ROPE(0, 128, false); else MVL_LAUNCH_ROPE(0, 64, false); }
        else            { if (D == 128) MVL_LAUNCH_ROPE(1, 128, false); else MVL_LAUNCH_ROPE(1, 64, false); }
    }
#undef MVL_LAUNCH_ROPE
    return MVL_CHECK_LAUNCH();
}

// ======================================================== standalone KV store
// key/value: [T, Hkv, D] with a token stride; one 16-byte chunk per thread.
__global__ void store_kvcache_kernel(const uint16_t* __restrict__ key,
                                     const uint16_t* __restrict__ value, long k_stride_t,
                                     long v_stride_t, uint16_t* k_cache, uint16_t* v_cache,
                                     long c_stride_b, long c_stride_t, long c_stride_h,
                                     int block_size, const int* __restrict__ slot_mapping,
                                     long T, int Hkv, int D) {
    const int cpt = Hkv * D / 8;                 // chunks per token
    const long total = T * cpt;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const long t = idx / cpt;
        const int c = (int)(idx - t * cpt);
        const int slot = slot_mapping[t];
        if (slot < 0) continue;
        const int e = c * 8, h = e / D, d = e - h * D;
        const int blk = slot / block_size, off = slot - blk * block_size;
        const long dst = blk * c_stride_b + off * c_stride_t + h * c_stride_h + d;
        *reinterpret_cast<u32x4_t*>(k_cache + dst) =
            *reinterpret_cast<const u32x4_t*>(key + t * k_stride_t + e);
        *reinterpret_cast<u32x4_t*>(v_cache + dst) =
            *reinterpret_cast<const u32x4_t*>(value + t * v_stride_t + e);
    }
}

extern "C" int mvl_store_kvcache(const void* key, const void* value, long k_stride_t,
                                 long v_stride_t, void* k_cache, void* v_cache, long c_stride_b,
                                 long c_stride_t, long c_stride_h, int block_size,
                                 const int* slot_mapping, long T, int Hkv, int D, void* stream) {
    if (D % 8 != 0) return -1;
    if (T == 0) return 0;
    const long total = T * (Hkv * D / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    store_kvcache_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(
        (const uint16_t*)key, (const uint16_t*)value, k_stride_t, v_stride_t, (uint16_t*)k_cache,
        (uint16_t*)v_cache, c_stride_b, c_stride_t, c_stride_h, block_size, slot_mapping, T, Hkv, D);
    return MVL_CHECK_LAUNCH();
}
