// All-reduce (sum) for decode-sized tensors over peer-mapped memory: every rank reads the other
// ranks' buffers directly over xGMI instead of going through a ring.
//
// The 8 GPUs of a node are fully connected by point-to-point xGMI links, so a rank can pull from
// all 7 peers at once, one link each.  For the messages of a decode step ([B, hidden] bf16, tens
// of KB to a few MB) a ring collective is latency-bound (2 (P - 1) dependent hops); here the
// whole operation is ONE kernel per rank with one or two flag barriers:
//
//   one-shot (small):  copy x -> my staging buffer | barrier | x = sum over ranks of staging
//   two-shot:          copy | barrier | reduce MY 1/P slice from all ranks' staging into my
//                      result buffer | barrier | gather the P reduced slices
//   No barrier closes a call.  What a later call overwrites is protected by the barriers it has
//   to pass first, given that a rank's calls run one after the other (stream order): a staging
//   buffer exists twice and calls alternate between the halves, so when call n + 2 writes the
//   half of call n every peer has signalled the first barrier of call n + 1, which it reaches
//   only after its kernel of call n - all of its reads - has finished; a result buffer is
//   written behind the first barrier of its call, by which time every peer has entered that
//   call and is done with the previous one.
//   one-shot + norm:   the one-shot form for [rows, hidden] inputs with the consumer folded in:
//                      a block owns whole rows, so after the sum it adds the residual stream,
//                      stores the new residual and writes the RMS-normalised row - the
//                      (residual add + RMSNorm) kernel that follows every row-parallel
//                      projection of a decoder layer disappears for decode-sized batches
//   two-shot + norm:   the same fusion for messages above the one-shot size (a 256-row decode
//                      batch is 2 MB): rank r owns rows [r R, (r + 1) R) of the residual stream
//                      (R = rows / P), reduces exactly those rows from all ranks' staging, adds
//                      ITS slice of the residual, stores it, normalises and publishes the
//                      normalised rows; the ranks then gather each other's normalised rows.  The
//                      residual stream stays row-sharded between the two norms of a layer and
//                      between layers, so only normalised rows ever cross the links a second time
//                      (2 (P-1)/P of the message per rank, as a plain two-shot all-reduce)
//   Both fused forms can take the producing GEMM's not yet summed fp32 K-slice partials instead
//   of its output: the slices are summed while the rows are staged.
//
// Two-shot moves 2 (P-1)/P of the message over each rank's links (like reduce-scatter +
// all-gather), one-shot (P-1) times the message but with one barrier less.
//
// Memory model.  Staging / result buffers and the flag words live in hipMalloc'd memory that the
// peers map through IPC handles.  Data written inside the kernel is published with a
// system-scope release (write back L2) after every storing wave has drained its stores; flags
// are system-scope relaxed atomics written to the PEER's flag array and polled locally; the
// consumer runs a system-scope acquire (invalidate) after its poll matched and before the first
// load of peer data.  Block b of every rank copies and reads the same element subset, so a
// block only ever depends on block b of its peers and needs no grid-wide barrier.  Every spin is
// bounded by wall-clock time (the constant 100 MHz counter): on expiry the block sets a sticky
// status word, does NOT touch the peers' data, overwrites its share of the output with NaN and
// leaves; the host polls the status word and raises.
#include "common.h"
#include <string.h>

using namespace mvl;

namespace {

constexpr int kMaxRanks = 8;
constexpr int kMaxBlocks = 64;
constexpr int kThreads = 512;
constexpr long kDefaultTimeoutTicks = 200000000L;      // 2 s of the 100 MHz counter

// One per rank, in peer-visible memory.  counter[phase][block][src rank] is written by rank
// `src`; epoch[block] is this rank's own call count (one more per launch, for every block whether
// the launch had that many blocks or not: the count's parity selects the staging half).
struct Flags {
    uint32_t counter[3][kMaxBlocks][kMaxRanks];
    uint32_t epoch[kMaxBlocks];
    uint32_t status;                           // != 0: a spin expired (sticky)
};

struct Peers {
    Flags* flags[kMaxRanks];
    uint16_t* stage[kMaxRanks];                // inputs, one buffer per rank
    uint16_t* result[kMaxRanks];               // reduced slices (two-shot)
    long stage_half;                           // elements of one half of a staging buffer
    long timeout_ticks;                        // longest wait for a peer, in 10 ns ticks
    // Rehearsal of one rank of a P-way group on a single GPU: every table entry is this rank's
    // own allocation and lane p of a barrier plays peer p (it raises the flag that peer would
    // raise).  Data paths, flag traffic and fences are those of the real thing; the sums are not.
    int loopback;
};

long g_timeout_ticks = kDefaultTimeoutTicks;

template <int CODE>
__device__ __forceinline__ u32x4_t nan_vec() {
    const uint32_t w = CODE == 0 ? 0x7FC07FC0u : 0x7E007E00u;      // quiet NaN, both halves
    return u32x4_t{w, w, w, w};
}

// All data stores of the block are done -> tell every peer's block `b`, then wait for theirs.
// False when a peer did not arrive in time: the caller must not read peer data.
template <int P>
__device__ __forceinline__ bool block_barrier(const Peers& pr, int rank, int phase, uint32_t epoch,
                                              bool publish) {
    if (publish) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains
    __syncthreads();
    Flags* mine = pr.flags[rank];
    int failed = 0;
    if (threadIdx.x < P) {
        if (publish) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");             // system scope: write back
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const int slot = pr.loopback ? (int)threadIdx.x : rank;
        __hip_atomic_store(&pr.flags[threadIdx.x]->counter[phase][blockIdx.x][slot], epoch,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const long t0 = (long)__builtin_amdgcn_s_memrealtime();
        int spins = 0;
        while (__hip_atomic_load(&mine->counter[phase][blockIdx.x][threadIdx.x], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
            if ((++spins & 63) == 0 && (long)__builtin_amdgcn_s_memrealtime() - t0 > pr.timeout_ticks) {
                __hip_atomic_store(&mine->status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                failed = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                 // system scope: invalidate
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    return __syncthreads_or(failed) == 0;
}

// Last statement of a call: the block's call count advances; block 0 also advances the counts of
// the block indices this launch did not have.
__device__ __forceinline__ void finish_call(Flags* mine, uint32_t epoch) {
    if (threadIdx.x == 0) mine->epoch[blockIdx.x] = epoch;
    if (blockIdx.x == 0 && threadIdx.x >= gridDim.x && threadIdx.x < kMaxBlocks)
        mine->epoch[threadIdx.x] = epoch;
}

template <int CODE>
__device__ __forceinline__ void accumulate(float (&acc)[8], u32x4_t v) {
    float f[8];
    unpack8<CODE>(v, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += f[j];
}

// n_vec = number of 16-byte vectors of the message.  TWO_SHOT needs n_vec % P == 0.
template <int CODE, int P, bool TWO_SHOT>
__global__ __launch_bounds__(kThreads) void all_reduce_kernel(Peers pr, int rank,
                                                               uint16_t* __restrict__ x, long n_vec) {
    Flags* mine = pr.flags[rank];
    const uint32_t epoch = mine->epoch[blockIdx.x] + 1;     // same on every rank: one per call
    const long stride = (long)gridDim.x * kThreads;
    const long first = (long)blockIdx.x * kThreads + threadIdx.x;
    u32x4_t* xv = reinterpret_cast<u32x4_t*>(x);
    const long hv = (long)(epoch & 1) * (pr.stage_half / 8);        // this call's staging half, in vectors

    // a peer that never arrived: this block's share of the output becomes NaN (never a sum over
    // buffers that may not have been written), the call count still advances
    auto give_up = [&]() {
        for (long i = first; i < n_vec; i += stride) xv[i] = nan_vec<CODE>();
        finish_call(mine, epoch);
    };
    if (!TWO_SHOT) {
        u32x4_t* st = reinterpret_cast<u32x4_t*>(pr.stage[rank]) + hv;
        for (long i = first; i < n_vec; i += stride) st[i] = xv[i];
        if (!block_barrier<P>(pr, rank, 0, epoch, true)) { give_up(); return; }
        for (long i = first; i < n_vec; i += stride) {
            u32x4_t v[P];
#pragma unroll
            for (int p = 0; p < P; ++p)
                v[p] = reinterpret_cast<const u32x4_t*>(pr.stage[p])[hv + i];
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < P; ++p) accumulate<CODE>(acc, v[p]);  // rank order: same bits everywhere
            xv[i] = pack8<CODE>(acc);
        }
    } else {
        const long part = n_vec / P;                         // vectors per slice
        u32x4_t* st = reinterpret_cast<u32x4_t*>(pr.stage[rank]) + hv;
        // block b moves the same relative subset of EVERY slice, on every rank
        for (int s = 0; s < P; ++s)
            for (long i = first; i < part; i += stride) st[s * part + i] = xv[s * part + i];
        if (!block_barrier<P>(pr, rank, 0, epoch, true)) { give_up(); return; }
        u32x4_t* res = reinterpret_cast<u32x4_t*>(pr.result[rank]);
        for (long i = first; i < part; i += stride) {
            u32x4_t v[P];
#pragma unroll
            for (int p = 0; p < P; ++p)
                v[p] = reinterpret_cast<const u32x4_t*>(pr.stage[p])[hv + rank * part + i];
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < P; ++p) accumulate<CODE>(acc, v[p]);
            const u32x4_t o = pack8<CODE>(acc);
            res[rank * part + i] = o;
            xv[rank * part + i] = o;
        }
        if (!block_barrier<P>(pr, rank, 1, epoch, true)) { give_up(); return; }
        for (long i = first; i < part; i += stride) {
            u32x4_t v[P];
#pragma unroll
            for (int p = 0; p < P; ++p)
                v[p] = reinterpret_cast<const u32x4_t*>(pr.result[p])[p * part + i];
#pragma unroll
            for (int p = 0; p < P; ++p)
                if (p != rank) xv[p * part + i] = v[p];
        }
    }
    finish_call(mine, epoch);
}

// What a rank contributes to a fused reduce + norm: its [rows, hidden] partial sums in the
// storage dtype, or the not yet summed fp32 K-slice partials [S][rows][hidden] of the GEMM that
// produced them (summed in slice order and rounded once, exactly as the stand-alone partial-sum
// kernel of mid_gemm.hip does, so both routes stage the same bits).
struct RowSource {
    const uint16_t* x;
    const float* parts;                        // nullptr: take x
    int nparts;
    long slab;                                 // rows * hidden
};

template <int CODE>
__device__ __forceinline__ u32x4_t load_chunk(const RowSource& s, long at) {
    if (s.parts == nullptr) return reinterpret_cast<const u32x4_t*>(s.x)[at];
    const float* p = s.parts + at * 8;
    f32x4_t lo = *reinterpret_cast<const f32x4_t*>(p);
    f32x4_t hi = *reinterpret_cast<const f32x4_t*>(p + 4);
    for (int k = 1; k < s.nparts; ++k) {
        lo += *reinterpret_cast<const f32x4_t*>(p + k * s.slab);
        hi += *reinterpret_cast<const f32x4_t*>(p + k * s.slab + 4);
    }
    const float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return pack8<CODE>(f);
}

// Sum over the P staging buffers of chunk `at` (rank order: same bits on every rank), rounded as
// the all-reduce rounds; + residual chunk, rounded as the add + norm kernel rounds.  Returns the
// chunk's share of the sum of squares.
template <int CODE, int P>
__device__ __forceinline__ float reduce_add_chunk(const Peers& pr, long hv, long at, u32x4_t* res_chunk,
                                                  float (&v)[8]) {
    u32x4_t pv[P];
#pragma unroll
    for (int p = 0; p < P; ++p) pv[p] = reinterpret_cast<const u32x4_t*>(pr.stage[p])[hv + at];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < P; ++p) accumulate<CODE>(acc, pv[p]);
    unpack8<CODE>(pack8<CODE>(acc), v);                     // the all-reduce's rounding
    float r[8];
    unpack8<CODE>(*res_chunk, r);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = DType<CODE>::to_float(DType<CODE>::from_float(v[j] + r[j]));
        ss += v[j] * v[j];
    }
    *res_chunk = pack8<CODE>(v);
    return ss;
}

// One-shot all-reduce of [rows, hidden] fused with `r = sum + residual; residual = r;
// out = rms_norm(r) * gamma`.  Block b owns rows b, b + gridDim.x, ...  Rounding points and the
// order of the sum of squares are those of the stand-alone kernels (all-reduce result rounded
// to the storage dtype, then elementwise.hip's add + norm with the same thread <-> chunk
// mapping), so both routes give the same bits.
template <int CODE, int P, int MAXC>
__global__ __launch_bounds__(1024) void all_reduce_norm_kernel(
        Peers pr, int rank, RowSource src, uint16_t* __restrict__ out, uint16_t* __restrict__ residual,
        const uint16_t* __restrict__ gamma, int rows, int hidden, float eps) {
    __shared__ float scratch[16];
    Flags* mine = pr.flags[rank];
    const uint32_t epoch = mine->epoch[blockIdx.x] + 1;
    const int nchunk = hidden >> 3;
    u32x4_t* ov = reinterpret_cast<u32x4_t*>(out);
    u32x4_t* rv = reinterpret_cast<u32x4_t*>(residual);
    const long hv = (long)(epoch & 1) * (pr.stage_half / 8);
    u32x4_t* st = reinterpret_cast<u32x4_t*>(pr.stage[rank]) + hv;

    for (int row = blockIdx.x; row < rows; row += gridDim.x)
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = threadIdx.x + i * blockDim.x;
            if (c < nchunk) st[(long)row * nchunk + c] = load_chunk<CODE>(src, (long)row * nchunk + c);
        }
    if (!block_barrier<P>(pr, rank, 0, epoch, true)) {
        for (int row = blockIdx.x; row < rows; row += gridDim.x)
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = threadIdx.x + i * blockDim.x;
                if (c < nchunk) ov[(long)row * nchunk + c] = nan_vec<CODE>();
            }
        finish_call(mine, epoch);
        return;
    }

    const u32x4_t* wr = reinterpret_cast<const u32x4_t*>(gamma);
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        float v[MAXC][8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = threadIdx.x + i * blockDim.x;
            if (c < nchunk) {
                const long at = (long)row * nchunk + c;
                ss += reduce_add_chunk<CODE, P>(pr, hv, at, rv + at, v[i]);
            }
        }
        ss = block_sum(ss, scratch);
        const float rstd = rsqrtf(ss / (float)hidden + eps);
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = threadIdx.x + i * blockDim.x;
            if (c < nchunk) {
                float w[8], o[8];
                unpack8<CODE>(wr[c], w);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = v[i][j] * rstd * w[j];
                ov[(long)row * nchunk + c] = pack8<CODE>(o);
            }
        }
    }
    finish_call(mine, epoch);
}

// Two-shot form with a row-sharded residual stream: `residual` is THIS rank's [rows / P, hidden]
// slice (rows rank R .. rank R + R - 1 of the stream).  Block b handles local rows j = b,
// b + gridDim.x, ... of EVERY slice in the staging and gather phases, so a block still only
// depends on block b of its peers.  Same rounding points as the one-shot form.
template <int CODE, int P, int MAXC>
__global__ __launch_bounds__(1024) void all_reduce_norm2_kernel(
        Peers pr, int rank, RowSource src, uint16_t* __restrict__ out, uint16_t* __restrict__ residual,
        const uint16_t* __restrict__ gamma, int rows, int hidden, float eps) {
    __shared__ float scratch[16];
    Flags* mine = pr.flags[rank];
    const uint32_t epoch = mine->epoch[blockIdx.x] + 1;
    const int nchunk = hidden >> 3;
    const int R = rows / P;
    u32x4_t* ov = reinterpret_cast<u32x4_t*>(out);
    u32x4_t* rv = reinterpret_cast<u32x4_t*>(residual);
    const long hv = (long)(epoch & 1) * (pr.stage_half / 8);
    u32x4_t* st = reinterpret_cast<u32x4_t*>(pr.stage[rank]) + hv;
    u32x4_t* res = reinterpret_cast<u32x4_t*>(pr.result[rank]);

    auto poison = [&]() {
        for (int s = 0; s < P; ++s)
            for (int j = blockIdx.x; j < R; j += gridDim.x)
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    const int c = threadIdx.x + i * blockDim.x;
                    if (c < nchunk) ov[(long)(s * R + j) * nchunk + c] = nan_vec<CODE>();
                }
        finish_call(mine, epoch);
    };

    // the block's row of several slices at a time: all of their loads are in flight before the
    // first store (a decode batch leaves a block ONE row per slice: slice after slice, every
    // store waiting for its own load, was 8 dependent round trips)
    constexpr int SB = MAXC <= 2 ? P : 2;
    for (int j = blockIdx.x; j < R; j += gridDim.x)
        for (int s0 = 0; s0 < P; s0 += SB) {
            u32x4_t tmp[SB][MAXC];
#pragma unroll
            for (int s = 0; s < SB; ++s)
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    const int c = threadIdx.x + i * blockDim.x;
                    if (c < nchunk) tmp[s][i] = load_chunk<CODE>(src, (long)((s0 + s) * R + j) * nchunk + c);
                }
#pragma unroll
            for (int s = 0; s < SB; ++s)
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    const int c = threadIdx.x + i * blockDim.x;
                    if (c < nchunk) st[(long)((s0 + s) * R + j) * nchunk + c] = tmp[s][i];
                }
        }
    if (!block_barrier<P>(pr, rank, 0, epoch, true)) { poison(); return; }

    const u32x4_t* wr = reinterpret_cast<const u32x4_t*>(gamma);
    for (int j = blockIdx.x; j < R; j += gridDim.x) {
        const long base = (long)(rank * R + j) * nchunk;
        float v[MAXC][8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = threadIdx.x + i * blockDim.x;
            if (c < nchunk) ss += reduce_add_chunk<CODE, P>(pr, hv, base + c, rv + (long)j * nchunk + c, v[i]);
        }
        ss = block_sum(ss, scratch);
        const float rstd = rsqrtf(ss / (float)hidden + eps);
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = threadIdx.x + i * blockDim.x;
            if (c < nchunk) {
                float w[8], o[8];
                unpack8<CODE>(wr[c], w);
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = v[i][k] * rstd * w[k];
                const u32x4_t ov8 = pack8<CODE>(o);
                res[base + c] = ov8;
                ov[base + c] = ov8;
            }
        }
    }
    if (!block_barrier<P>(pr, rank, 1, epoch, true)) { poison(); return; }

    for (int j = blockIdx.x; j < R; j += gridDim.x)
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = threadIdx.x + i * blockDim.x;
            if (c < nchunk) {
                u32x4_t v[P];
#pragma unroll
                for (int p = 0; p < P; ++p)
                    v[p] = reinterpret_cast<const u32x4_t*>(pr.result[p])[(long)(p * R + j) * nchunk + c];
#pragma unroll
                for (int p = 0; p < P; ++p)
                    if (p != rank) ov[(long)(p * R + j) * nchunk + c] = v[p];
            }
        }
    finish_call(mine, epoch);
}

template <int CODE, int P>
int launch_norm(const Peers& pr, int rank, const RowSource& src, void* out, void* residual,
                const void* gamma, int rows, int hidden, float eps, bool two_shot, hipStream_t st) {
    const int nchunk = hidden / 8;
    int threads = 64;                                       // as elementwise.hip's launch_rms
    while (threads < nchunk && threads < 1024) threads <<= 1;
    const int per = (nchunk + threads - 1) / threads;
    const int own = two_shot ? rows / P : rows;
    const unsigned blocks = (unsigned)(own < kMaxBlocks ? own : kMaxBlocks);
    auto O = (uint16_t*)out; auto R = (uint16_t*)residual; auto G = (const uint16_t*)gamma;
#define MVL_NORM(MAXC)                                                                              \
    do {                                                                                            \
        if (two_shot) all_reduce_norm2_kernel<CODE, P, MAXC><<<blocks, threads, 0, st>>>(           \
                pr, rank, src, O, R, G, rows, hidden, eps);                                         \
        else all_reduce_norm_kernel<CODE, P, MAXC><<<blocks, threads, 0, st>>>(                     \
                pr, rank, src, O, R, G, rows, hidden, eps);                                         \
    } while (0)
    if (per <= 1) MVL_NORM(1);
    else if (per <= 2) MVL_NORM(2);
    else if (per <= 4) MVL_NORM(4);
    else return -1;
#undef MVL_NORM
    return MVL_CHECK_LAUNCH();
}

template <int CODE>
int launch_norm_p(const Peers& pr, int world, int rank, const RowSource& src, void* out,
                  void* residual, const void* gamma, int rows, int hidden, float eps, bool two_shot,
                  hipStream_t st) {
    switch (world) {
        case 2: return launch_norm<CODE, 2>(pr, rank, src, out, residual, gamma, rows, hidden, eps, two_shot, st);
        case 4: return launch_norm<CODE, 4>(pr, rank, src, out, residual, gamma, rows, hidden, eps, two_shot, st);
        case 8: return launch_norm<CODE, 8>(pr, rank, src, out, residual, gamma, rows, hidden, eps, two_shot, st);
        default: return -1;
    }
}

template <int CODE, int P>
int launch(const Peers& pr, int rank, void* x, long n_vec, bool two_shot, hipStream_t st) {
    const long per_block = two_shot ? n_vec / P : n_vec;
    long blocks = (per_block + kThreads - 1) / kThreads;
    if (blocks < 1) blocks = 1;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    if (two_shot)
        all_reduce_kernel<CODE, P, true><<<(unsigned)blocks, kThreads, 0, st>>>(pr, rank, (uint16_t*)x, n_vec);
    else
        all_reduce_kernel<CODE, P, false><<<(unsigned)blocks, kThreads, 0, st>>>(pr, rank, (uint16_t*)x, n_vec);
    return MVL_CHECK_LAUNCH();
}

template <int CODE>
int launch_p(const Peers& pr, int world, int rank, void* x, long n_vec, bool two_shot, hipStream_t st) {
    switch (world) {
        case 2: return launch<CODE, 2>(pr, rank, x, n_vec, two_shot, st);
        case 4: return launch<CODE, 4>(pr, rank, x, n_vec, two_shot, st);
        case 8: return launch<CODE, 8>(pr, rank, x, n_vec, two_shot, st);
        default: return -1;
    }
}

}  // namespace

// ------------------------------------------------------------------ memory + IPC plumbing
extern "C" size_t mvl_car_flags_bytes() { return sizeof(Flags); }
extern "C" int mvl_car_ipc_handle_bytes() { return (int)sizeof(hipIpcMemHandle_t); }

// Zero-initialised device allocation outside PyTorch's caching allocator (an IPC handle names a
// whole allocation).  uncached != 0: fine-grained memory for the flag words.
extern "C" int mvl_car_alloc(void** out, size_t bytes, int uncached) {
    hipError_t e = uncached ? hipExtMallocWithFlags(out, bytes, hipDeviceMallocUncached)
                            : hipMalloc(out, bytes);
    if (e != hipSuccess) return (int)e;
    return (int)hipMemset(*out, 0, bytes);
}
extern "C" int mvl_car_free(void* p) { return (int)hipFree(p); }
extern "C" int mvl_car_ipc_get(void* p, void* handle_out) {
    return (int)hipIpcGetMemHandle((hipIpcMemHandle_t*)handle_out, p);
}
extern "C" int mvl_car_ipc_open(const void* handle, void** out) {
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    return (int)hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess);
}
extern "C" int mvl_car_ipc_close(void* p) { return (int)hipIpcCloseMemHandle(p); }

static long g_loopback = 0;

// Shard rehearsal (one rank of a `world`-way group alone on its GPU): every later launch treats
// the pointer tables as all-own and lets barrier lane p stand in for peer p.
extern "C" void mvl_car_set_loopback(int on) { g_loopback = on; }

static void fill_peers(Peers& pr, void* const* flags, void* const* stage, void* const* result, int world,
                       long stage_half_bytes) {
    pr.stage_half = stage_half_bytes / 2;
    for (int p = 0; p < world; ++p) {
        pr.flags[p] = (Flags*)flags[p];
        pr.stage[p] = (uint16_t*)stage[p];
        pr.result[p] = result ? (uint16_t*)result[p] : nullptr;
    }
    pr.timeout_ticks = g_timeout_ticks;
    pr.loopback = (int)g_loopback;
}

// flags / stage / result: `world` device pointers each, indexed by rank (own entries are the
// local allocations, the others IPC mappings).  A staging buffer is two halves of
// stage_half_bytes (a multiple of 16) each, a result buffer one.  x: [n_elems] 16-bit elements,
// in place.
extern "C" int mvl_car_all_reduce(void* const* flags, void* const* stage, void* const* result,
                                  int world, int rank, void* x, long n_elems, int two_shot,
                                  int dtype, long stage_half_bytes, void* stream) {
    if (n_elems == 0) return 0;
    if (world > kMaxRanks || n_elems % 8 != 0 || ((uintptr_t)x & 15) != 0) return -1;
    if (stage_half_bytes % 16 != 0 || n_elems * 2 > stage_half_bytes) return -1;
    const long n_vec = n_elems / 8;
    if (two_shot && n_vec % world != 0) return -1;
    Peers pr;
    fill_peers(pr, flags, stage, result, world, stage_half_bytes);
    auto st = (hipStream_t)stream;
    return dtype == 0 ? launch_p<0>(pr, world, rank, x, n_vec, two_shot != 0, st)
                      : launch_p<1>(pr, world, rank, x, n_vec, two_shot != 0, st);
}

// out [rows, hidden] = rms_norm(sum over ranks + residual) * gamma, residual <- sum + residual.
// The rank's contribution is x [rows, hidden] (contiguous; out may alias it) or, when parts is
// given, the fp32 K-slice partials [nparts][rows][hidden] that sum to it.
// two_shot == 0: `residual` is the full [rows, hidden] stream (replicated on every rank).
// two_shot != 0: rows % world == 0 and `residual` is this rank's [rows / world, hidden] slice.
// rows * hidden * 2 bytes must fit half the staging buffer and the result buffer.
extern "C" int mvl_car_all_reduce_norm(void* const* flags, void* const* stage, void* const* result,
                                       int world, int rank, const void* x, const float* parts,
                                       int nparts, void* out, void* residual, const void* gamma,
                                       int rows, int hidden, float eps, int two_shot, int dtype,
                                       long stage_half_bytes, void* stream) {
    if (rows == 0) return 0;
    if (stage_half_bytes % 16 != 0 || (long)rows * hidden * 2 > stage_half_bytes) return -1;
    if (world > kMaxRanks || hidden % 8 != 0 || hidden > 8 * 1024 * 4 ||
        (((uintptr_t)x | (uintptr_t)parts | (uintptr_t)out | (uintptr_t)residual | (uintptr_t)gamma) & 15) != 0)
        return -1;
    if (two_shot && (rows % world != 0 || result == nullptr)) return -1;
    if (parts == nullptr ? x == nullptr : nparts < 1) return -1;
    Peers pr;
    fill_peers(pr, flags, stage, result, world, stage_half_bytes);
    RowSource src;
    src.x = (const uint16_t*)x; src.parts = parts; src.nparts = nparts; src.slab = (long)rows * hidden;
    auto st = (hipStream_t)stream;
    return dtype == 0
        ? launch_norm_p<0>(pr, world, rank, src, out, residual, gamma, rows, hidden, eps, two_shot != 0, st)
        : launch_norm_p<1>(pr, world, rank, src, out, residual, gamma, rows, hidden, eps, two_shot != 0, st);
}

// Longest time a block waits for a peer's flag before it gives up (default 2 s).  Takes effect
// for launches (and graph captures) made afterwards.
extern "C" void mvl_car_set_timeout_ms(long ms) { g_timeout_ticks = ms > 0 ? ms * 100000L : kDefaultTimeoutTicks; }

// Reads the sticky status word (0 = every barrier of every call so far completed).
extern "C" int mvl_car_status(const void* flags_self) {
    uint32_t s = 0;
    if (hipMemcpy(&s, &((const Flags*)flags_self)->status, sizeof(s), hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    return (int)s;
}
