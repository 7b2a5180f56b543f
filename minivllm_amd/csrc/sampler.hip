// Token sampler: one pass over the logits per row.
//
// The reference computes softmax(logits / T), divides by Exp(1) noise and takes the
// arg-max (reference: src/myvllm/layers/sampler.py:14-19).  argmax_i p_i / E_i is the same
// as argmax_i (logit_i / T - log E_i), so no normalisation pass is needed: every row is
// read exactly once.  Rows with temperature 0 take the plain arg-max.
//
// The noise for vocabulary entry i of row r at step s is a pure function of
// (seed, s, r, GLOBAL index i), so a tensor-parallel rank that owns a vocabulary shard
// draws exactly the numbers a single GPU would; ranks then only exchange their
// (best score, best index) pairs.
#include "common.h"

using namespace mvl;

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Per-element hash: three 32-bit multiply/xorshift rounds over (key, index).  The row key is
// derived with the 64-bit mixer once per workgroup; per element only 32-bit integer multiplies
// are spent (64-bit ones cost four quarter-rate instructions each on this hardware and made the
// sampler ALU-bound).
__device__ __forceinline__ uint32_t mix32(uint32_t x, uint32_t key_lo, uint32_t key_hi) {
    x += key_lo;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x ^= key_hi; x *= 0x846ca68bu;
    x ^= x >> 16; x *= 0x9e3779b1u;
    return x ^ (x >> 15);
}

// -log(E) with E ~ Exp(1)  ==  -log(-log(u)),  u uniform in (0, 1)
__device__ __forceinline__ float gumbel_noise(uint64_t key, uint32_t idx) {
    const uint32_t z = mix32(idx, (uint32_t)key, (uint32_t)(key >> 32));
    const float u = ((float)(z >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return -__logf(-__logf(u));
}

__device__ __forceinline__ void take_better(float& bv, int& bi, float v, int i) {
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}

// DT: 0 bf16, 1 f16, 2 f32
template <int DT>
__global__ __launch_bounds__(1024) void sample_kernel(
        const void* __restrict__ logits, long row_stride, int vocab_local, int vocab_offset,
        int valid_vocab, const float* __restrict__ temperatures, uint64_t seed,
        const int* __restrict__ step_ptr, float* __restrict__ out_val, int* __restrict__ out_idx) {
    __shared__ float s_val[16];
    __shared__ int s_idx[16];
    const int row = blockIdx.x;
    // gridDim.y workgroups share a row (small batches: one workgroup per row would leave most
    // of the chip idle); each takes a run of 8-element groups and writes a partial result
    const int chunks = gridDim.y, chunk = blockIdx.y;
    const int groups = (vocab_local + 7) / 8;
    const int g_per = (groups + chunks - 1) / chunks;
    const int e_lo = min(vocab_local, chunk * g_per * 8);
    const int e_hi = min(vocab_local, (chunk + 1) * g_per * 8);
    const float temp = temperatures[row];
    const bool greedy = !(temp > 0.f);
    const float inv_t = greedy ? 1.f : 1.f / temp;
    const uint64_t step = (uint64_t)(uint32_t)step_ptr[0];
    const uint64_t key = splitmix64(seed ^ (step * 0x9E3779B97F4A7C15ull)) ^
                         splitmix64(((uint64_t)row << 32) | 0x5bd1e995u);
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if (DT == 2) {
        const float* p = (const float*)logits + (long)row * row_stride;
        for (int i = e_lo + threadIdx.x; i < e_hi; i += blockDim.x) {
            const int gi = vocab_offset + i;
            if (gi >= valid_vocab) break;
            float v = p[i] * inv_t;
            if (!greedy) v += gumbel_noise(key, (uint32_t)gi);
            take_better(bv, bi, v, gi);
        }
    } else {
        const uint16_t* p = (const uint16_t*)logits + (long)row * row_stride;
        const bool vec_ok = (row_stride % 8 == 0) && (((uintptr_t)logits & 15) == 0);
        const int v_lo = e_lo / 8, v_hi = vec_ok ? e_hi / 8 : v_lo;     // e_lo is a multiple of 8
        const u32x4_t* pv = reinterpret_cast<const u32x4_t*>(p);
        for (int c = v_lo + threadIdx.x; c < v_hi; c += blockDim.x) {
            float f[8];
            unpack8<(DT == 0 ? 0 : 1)>(pv[c], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int gi = vocab_offset + c * 8 + j;
                if (gi < valid_vocab) {
                    float v = f[j] * inv_t;
                    if (!greedy) v += gumbel_noise(key, (uint32_t)gi);
                    take_better(bv, bi, v, gi);
                }
            }
        }
        for (int i = v_hi * 8 + threadIdx.x; i < e_hi; i += blockDim.x) {
            const int gi = vocab_offset + i;
            if (gi < valid_vocab) {
                float v = DType<(DT == 0 ? 0 : 1)>::to_float(p[i]) * inv_t;
                if (!greedy) v += gumbel_noise(key, (uint32_t)gi);
                take_better(bv, bi, v, gi);
            }
        }
    }
    // wave arg-max, then across waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        take_better(bv, bi, ov, oi);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) { s_val[wid] = bv; s_idx[wid] = bi; }
    __syncthreads();
    if (wid == 0) {
        bv = lane < nw ? s_val[lane] : -INFINITY;
        bi = lane < nw ? s_idx[lane] : 0x7fffffff;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            take_better(bv, bi, ov, oi);
        }
        if (lane == 0) {
            out_val[row * chunks + chunk] = bv;
            out_idx[row * chunks + chunk] = (bi == 0x7fffffff) ? vocab_offset : bi;
        }
    }
}

// Arg-max over the partial results of one row: grid = rows, 64 threads.
__global__ void sample_merge_kernel(const float* __restrict__ part_val, const int* __restrict__ part_idx,
                                    int chunks, float* __restrict__ out_val, int* __restrict__ out_idx) {
    const int row = blockIdx.x, lane = threadIdx.x;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < chunks; c += 64)
        take_better(bv, bi, part_val[row * chunks + c], part_idx[row * chunks + c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        take_better(bv, bi, ov, oi);
    }
    if (lane == 0) { out_val[row] = bv; out_idx[row] = bi; }
}

// chunks > 1: every row is split over that many workgroups; ws_val / ws_idx hold rows * chunks
// partial results and a second, tiny kernel merges them.
extern "C" int mvl_sample(const void* logits, long row_stride, int rows, int vocab_local,
                          int vocab_offset, int valid_vocab, const float* temperatures,
                          uint64_t seed, const int* step_ptr, float* out_val, int* out_idx,
                          int dtype, int chunks, float* ws_val, int* ws_idx, void* stream) {
    if (rows == 0) return 0;
    if (chunks < 1 || (chunks > 1 && (ws_val == nullptr || ws_idx == nullptr))) return -1;
    auto st = (hipStream_t)stream;
    const int threads = 1024;
    const dim3 grid(rows, chunks);
    float* ov = chunks > 1 ? ws_val : out_val;
    int* oi = chunks > 1 ? ws_idx : out_idx;
    if (dtype == 0)
        sample_kernel<0><<<grid, threads, 0, st>>>(logits, row_stride, vocab_local, vocab_offset,
                                                   valid_vocab, temperatures, seed, step_ptr, ov, oi);
    else if (dtype == 1)
        sample_kernel<1><<<grid, threads, 0, st>>>(logits, row_stride, vocab_local, vocab_offset,
                                                   valid_vocab, temperatures, seed, step_ptr, ov, oi);
    else if (dtype == 2)
        sample_kernel<2><<<grid, threads, 0, st>>>(logits, row_stride, vocab_local, vocab_offset,
                                                   valid_vocab, temperatures, seed, step_ptr, ov, oi);
    else
        return -1;
    int rc = MVL_CHECK_LAUNCH();
    if (rc != 0 || chunks == 1) return rc;
    sample_merge_kernel<<<rows, 64, 0, st>>>(ws_val, ws_idx, chunks, out_val, out_idx);
    return MVL_CHECK_LAUNCH();
}
