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

// ---------------------------------------------------------------- top-k / top-p rows
// Sampling from the distribution restricted to the k most likely tokens and / or to the
// smallest set of most likely tokens whose probability reaches p, without sorting the row.
// The kept set K is an upper set of the row (token i is kept iff fewer than k tokens are
// strictly more likely and the tokens strictly more likely than i hold less than p of the
// mass; ties at the cut are kept), so rejection sampling with a rising pivot is exact:
//   draw c from the pool (Gumbel arg-max; the pool starts as the whole row);
//   one pass over the row gives count and mass of the tokens above c -> is c in K?
//   if not, nothing at or below c is in K either: the pool becomes the tokens above c,
//   and the same pass has already drawn the next candidate from it (fresh noise).
// Conditioned on landing in K a draw proportional to the probabilities over any pool that
// contains K is the restricted distribution itself.  The pool shrinks with every rejection
// (on a flat row the candidate's rank roughly halves per round; on the peaked rows of a
// trained model the first or second candidate is accepted).  Every sum is a fixed-shape tree
// over a fixed thread-to-element map: same inputs, same token.
// The first round uses the plain sampler's noise, so its candidate is the token that
// sampler drew for the row.
namespace {

struct Cand { float v; int i; float s; };         // noisy score, index, scaled logit

__device__ __forceinline__ void take_better(Cand& a, const Cand& b) {
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) a = b;
}

constexpr int kFilterThreads = 1024;
constexpr int kFilterRounds = 96;

// f(scaled logit, index) for every element of the row, 16 bytes per load where the row allows;
// the thread-to-element map depends on nothing but the row's address and length
template <int DT, typename F>
__device__ __forceinline__ void for_each_logit(const char* row, int vocab, float inv_t, F&& f) {
    if (DT == 2) {
        const float* p = (const float*)row;
        for (int i = threadIdx.x; i < vocab; i += kFilterThreads) f(p[i] * inv_t, i);
    } else {
        constexpr int C = DT == 0 ? 0 : 1;
        const uint16_t* p = (const uint16_t*)row;
        const int nv = (((uintptr_t)row & 15) == 0) ? vocab / 8 : 0;
        const u32x4_t* pv = reinterpret_cast<const u32x4_t*>(row);
        for (int c = threadIdx.x; c < nv; c += kFilterThreads) {
            float x[8];
            unpack8<C>(pv[c], x);
#pragma unroll
            for (int j = 0; j < 8; ++j) f(x[j] * inv_t, c * 8 + j);
        }
        for (int i = nv * 8 + threadIdx.x; i < vocab; i += kFilterThreads)
            f(DType<C>::to_float(p[i]) * inv_t, i);
    }
}

}  // namespace

template <int DT>
__global__ __launch_bounds__(kFilterThreads) void sample_filtered_kernel(
        const void* __restrict__ logits, long row_stride, int vocab, const int* __restrict__ logit_rows,
        const int* __restrict__ batch_rows, const float* __restrict__ temperatures,
        const int* __restrict__ top_k, const float* __restrict__ top_p, uint64_t seed,
        const int* __restrict__ step_ptr, int* __restrict__ out_idx, int* __restrict__ out_rounds) {
    __shared__ float s_f[4][16];
    __shared__ int s_i[2][16];
    const int r = blockIdx.x;
    const int row = batch_rows[r];
    const char* base = (const char*)logits + (long)logit_rows[r] * row_stride * (DT == 2 ? 4 : 2);
    const float temp = temperatures[row];
    const bool greedy = !(temp > 0.f);
    const float inv_t = greedy ? 1.f : 1.f / temp;
    const int k = top_k[row];
    const float p = top_p[row];
    const uint64_t step = (uint64_t)(uint32_t)step_ptr[0];
    const uint64_t key0 = splitmix64(seed ^ (step * 0x9E3779B97F4A7C15ull)) ^
                          splitmix64(((uint64_t)row << 32) | 0x5bd1e995u);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;

    // workgroup reductions (16 waves); every thread gets the result
    auto reduce_cand = [&](Cand c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            Cand t{__shfl_xor(c.v, o, 64), __shfl_xor(c.i, o, 64), __shfl_xor(c.s, o, 64)};
            take_better(c, t);
        }
        __syncthreads();
        if (lane == 0) { s_f[0][wid] = c.v; s_i[0][wid] = c.i; s_f[1][wid] = c.s; }
        __syncthreads();
        c = Cand{s_f[0][lane & 15], s_i[0][lane & 15], s_f[1][lane & 15]};
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            Cand t{__shfl_xor(c.v, o, 64), __shfl_xor(c.i, o, 64), __shfl_xor(c.s, o, 64)};
            take_better(c, t);
        }
        return c;
    };
    auto reduce_sum = [&](float x, int n, float& xs, int& ns) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { x += __shfl_xor(x, o, 64); n += __shfl_xor(n, o, 64); }
        __syncthreads();
        if (lane == 0) { s_f[2][wid] = x; s_i[1][wid] = n; }
        __syncthreads();
        x = s_f[2][lane & 15]; n = s_i[1][lane & 15];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { x += __shfl_xor(x, o, 64); n += __shfl_xor(n, o, 64); }
        xs = x; ns = n;
    };

    // ---- pass 0: row maximum, first candidate (and the plain arg-max as a last resort)
    Cand c{-INFINITY, 0x7fffffff, -INFINITY}, top{-INFINITY, 0x7fffffff, -INFINITY};
    float zt = 0.f;                              // this thread's mass relative to its own maximum
    for_each_logit<DT>(base, vocab, inv_t, [&](float sc, int i) {
        if (sc > top.s) zt = zt * __expf(top.s - sc) + 1.f;       // (exp(-inf) = 0 the first time)
        else if (top.s > -INFINITY) zt += __expf(sc - top.s);
        take_better(top, Cand{sc, i, sc});
        take_better(c, Cand{greedy ? sc : sc + gumbel_noise(key0, (uint32_t)i), i, sc});
    });
    const float mt = top.s;
    top = reduce_cand(top);
    c = reduce_cand(c);
    const float m = top.s;
    int rounds = 0;
    const bool filtered = !greedy && (k > 0 || p < 1.f) && vocab > 0;
    if (filtered) {
        // total mass relative to the row maximum (fixed summation order), then the rounds
        float z = mt == -INFINITY ? 0.f : zt * __expf(mt - m);
        int unused;
        reduce_sum(z, 0, z, unused);
        const float p_mass = p * z;
        for (;;) {
            ++rounds;
            const uint64_t key = key0 ^ splitmix64(0xA24BAED4963EE407ull * (uint64_t)rounds);
            Cand nxt{-INFINITY, 0x7fffffff, -INFINITY};
            float mass = 0.f;
            int cnt = 0;
            const float pivot = c.s;
            for_each_logit<DT>(base, vocab, inv_t, [&](float sc, int i) {
                if (sc > pivot) {
                    ++cnt;
                    mass += __expf(sc - m);
                    take_better(nxt, Cand{sc + gumbel_noise(key, (uint32_t)i), i, sc});
                }
            });
            reduce_sum(mass, cnt, mass, cnt);
            if (cnt == 0 || ((k <= 0 || cnt < k) && mass < p_mass)) break;      // c is kept
            if (rounds >= kFilterRounds) { c = top; break; }                    // the arg-max always is
            c = reduce_cand(nxt);
        }
    }
    if (threadIdx.x == 0) {
        out_idx[row] = c.i == 0x7fffffff ? 0 : c.i;
        if (out_rounds != nullptr) out_rounds[row] = rounds;
    }
}

// One workgroup per listed row: logit_rows[r] is the row of ``logits`` to read, batch_rows[r]
// the row of the step it belongs to (index into temperatures / top_k / top_p / out_idx and part
// of the noise key).  top_k <= 0 and top_p >= 1 switch the respective filter off.
extern "C" int mvl_sample_filtered(const void* logits, long row_stride, int vocab, int n_rows,
                                   const int* logit_rows, const int* batch_rows,
                                   const float* temperatures, const int* top_k, const float* top_p,
                                   uint64_t seed, const int* step_ptr, int* out_idx,
                                   int* out_rounds, int dtype, void* stream) {
    if (n_rows == 0) return 0;
    if (vocab <= 0 || dtype < 0 || dtype > 2) return -1;
    auto st = (hipStream_t)stream;
#define MVL_LAUNCH_FILTERED(DT)                                                                 \
    sample_filtered_kernel<DT><<<n_rows, kFilterThreads, 0, st>>>(                              \
        logits, row_stride, vocab, logit_rows, batch_rows, temperatures, top_k, top_p, seed,    \
        step_ptr, out_idx, out_rounds)
    if (dtype == 0) MVL_LAUNCH_FILTERED(0);
    else if (dtype == 1) MVL_LAUNCH_FILTERED(1);
    else MVL_LAUNCH_FILTERED(2);
#undef MVL_LAUNCH_FILTERED
    return MVL_CHECK_LAUNCH();
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
