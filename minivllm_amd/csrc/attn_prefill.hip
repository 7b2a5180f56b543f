// Flash-attention prefill for gfx950: causal, variable-length, GQA, with the keys/values
// read either from packed [T, Hkv, D] tensors or from the paged KV cache through a block
// table (q_len <= k_len, so a cached prefix or an earlier chunk of the same prompt is
// attended exactly).
//
// The reference Triton kernel uses 32x32 tiles at D=128, visits every KV tile even above the
// causal diagonal, and can only see the new tokens' K/V (reference:
// src/myvllm/layers/attention.py:111-280).  Design here:
//
//  * workgroup = 4 waves x 32 query rows (128-row Q tile), KV tiles of 64 rows; Q lives in
//    registers for the whole kernel, K/V tiles go global -> LDS by LDS-DMA into a double
//    buffer, requested one tile ahead (they fly under the PV products of the current tile and
//    the barrier that closes it) with the block-table entry read two tiles ahead; only an fp8
//    cache is staged through registers (it has to be widened on the way);
//  * S^T = K . Q^T on MFMA 32x32x16, so a lane owns one query row: the softmax row
//    reduction is in-register plus one exchange between the two lane halves, and the
//    accumulator is already the B operand of O^T = V^T . P^T (no LDS round trip for P);
//  * V^T fragments come from ds_read_b64_tr_b16; the LDS image is XOR-swizzled so that both
//    the K row reads (ds_read_b128) and the transposed V reads are bank-conflict free;
//  * only KV tiles at or below the causal diagonal are visited, a wave skips the MFMA work
//    of tiles that lie entirely above its own 32 rows, and the mask is applied on diagonal
//    tiles only;
//  * the 1-D grid is remapped so that workgroups sharing an XCD (and its L2) walk the query
//    heads of one KV head and adjacent Q tiles of one sequence, heaviest Q tiles first.
#include "common.h"
#include <type_traits>
#include <cstdlib>

using namespace mvl;

typedef __attribute__((ext_vector_type(8))) short s16x8_t;

namespace {

// f(integral_constant<int, I>) for I = FIRST .. LAST - 1: loop indices that stay constant expressions
// (immediate operands of inline asm)
template <int FIRST, int LAST, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (FIRST < LAST) {
        f(std::integral_constant<int, FIRST>{});
        static_for<FIRST + 1, LAST>(f);
    }
}

constexpr int kBM = 128;           // query rows per workgroup
constexpr int kBN = 64;            // kv rows per tile
constexpr int kThreads = 256;
constexpr float kRescaleThr = 6.0f;   // log2 units: P <= 64 between rescales

struct PrefillArgs {
    const uint16_t* q; long q_stride_t; long q_stride_h;
    const uint16_t* k; const uint16_t* v;      // packed mode: [T, Hkv, D]; paged: caches
    long k_stride_t, k_stride_h;               // packed: token / head strides; paged: cs_t / cs_h
    long cs_b;                                 // paged: block stride
    uint16_t* out; long o_stride_t; long o_stride_h;
    const int* cu_q; const int* cu_k;          // [B+1]
    const int* block_tables; int bt_stride; int bs_log2;
    float scale_log2;
    int B, Hq, Hkv, n_qtiles;
    int n_items;                               // Q-tile items per (sequence, head): n_qtiles, or pairs of them
    int pair;                                  // 0: an item is one Q tile; g > 0: g pairs (n_qtiles - 1 - p, p)
    int total_wg;
    // optional work list (n_items slots): entry (sequence << 16 | pair index) or -1 for a slot
    // without work.  The grid is then n_items x Hq workgroups whatever the sequence lengths are,
    // which is what a captured graph needs: the launch shape depends on the step's token and
    // sequence BUDGET only, the host writes the real work of each replay into the list.
    const int* items;
    // optional: q holds the raw projection and the kernel applies the per-head RMSNorm (weight
    // q_w, may be null) and the rotary embedding (fp32 table [position][D]: cos of the D/2
    // pairs, then sin) to its Q rows while it loads them; null cos_sin: q is used as it is
    const int* positions;                      // [total tokens], indexed like q; null: k_len - q_len + i
    const float* cos_sin;
    const uint16_t* q_w;
    float eps;
#ifdef MVL_PREFILL_PROBE
    unsigned long long* probe;                 // [total_wg][8] timestamps (tools/native/prefill_phase_probe.hip)
#endif
};

#ifdef MVL_PREFILL_PROBE
#define MVL_PROBE(slot, value) do { if (threadIdx.x == 0) a.probe[(long)blockIdx.x * 8 + (slot)] = (value); } while (0)
#else
#define MVL_PROBE(slot, value) do { } while (0)
#endif

// KV8 (paged only): the cache holds OCP fp8 (e4m3) bytes.  LDS-DMA cannot widen, so this variant
// stages K/V tiles through registers: 16-byte loads of 16 fp8 values one tile ahead, widened to
// the query's 16-bit type (exact) and written into the same swizzled LDS image after the current
// tile's MFMA work; everything downstream of the LDS image is unchanged.
// UNI: every row of a 64-row K/V tile comes from one uniform base (packed K/V, or a paged cache
// whose blocks hold a multiple of 64 tokens).  The tile is then described by a buffer descriptor
// built from scalars (block id, token offset, valid rows) and each lane's source offset is a
// constant of the kernel: no per-lane 64-bit address arithmetic and no per-lane block-table
// loads in the loop (those were ~70 of the ~240 vector-ALU instructions of a tile, and this
// kernel's time is the sum of its MFMA and vector-ALU issue).  Rows beyond the context read as 0.
template <int CODE, int D, bool PAGED, bool KV8 = false, bool UNI = false>
__global__ __launch_bounds__(kThreads, 2) void flash_prefill_kernel(PrefillArgs a) {
    static_assert(!KV8 || PAGED, "an fp8 K/V source is always the paged cache");
    static_assert(!(KV8 && UNI), "the fp8 arm stages through registers");
    using DT = DType<CODE>;
    using vec8 = typename DT::vec8;
    constexpr int CPR = D / 8;
    constexpr int KSTEPS = D / 16;             // 32x32x16 k-steps over the head dim
    constexpr int NDB = D / 32;                // 32-wide blocks of the output head dim
    constexpr int ITER = kBN * CPR / kThreads; // staging chunks per thread per tile
    constexpr int CPR8 = D / 16;               // 16-byte chunks of an fp8 row
    constexpr int ITER8 = kBN * CPR8 / kThreads;
    constexpr int TILE_BYTES = kBN * D * 2;

    // two buffers of {K tile, V tile}; tile t lives in buffer t & 1
    extern __shared__ __attribute__((aligned(16))) char smem[];
    MVL_PROBE(0, wall_clock64());

    // ---- work decomposition: XCD-aware remap of the linear block id -----------------
    const int G = a.Hq / a.Hkv;
    constexpr int kItemChunk = 4;
    // the q-norm weights (D values, the same for every row and head) go through LDS: a copy in
    // registers is 32 of them per lane at the prologue's peak, which is what decides whether that
    // prologue fits without spills.  Requested here, stored and published (one barrier) right
    // before the first use, when the load has long returned.
    __shared__ u32x4_t s_qw[D / 8];
    const bool has_qw = a.cos_sin != nullptr && a.q_w != nullptr;
    bool qw_pending = has_qw;
    u32x4_t qw_mine = {0u, 0u, 0u, 0u};
    if (has_qw && threadIdx.x < D / 8)
        qw_mine = *reinterpret_cast<const u32x4_t*>(a.q_w + 8 * threadIdx.x);
    int g, qti, kvh, b;
    if (a.items != nullptr) {
        // Work list: the real entries come first, the unused slots of the budget last, so the
        // list is dealt to the XCDs in chunks of kItemChunk consecutive slots, round robin
        // (blocks with equal id % 8 share an XCD): every XCD gets its share of the real work,
        // the G heads of a (slot, KV head) and the neighbouring pairs of a sequence - which read
        // the same K/V - stay on one L2.  n_items is a multiple of 8 kItemChunk (host).
        const int xcd = blockIdx.x & 7;
        int k = blockIdx.x >> 3;
        g = k % G;                         k /= G;
        const int within = k % kItemChunk; k /= kItemChunk;
        const int chunk = k * 8 + xcd;                       // over (n_items / kItemChunk) x Hkv
        kvh = chunk % a.Hkv;
        const int slot = (chunk / a.Hkv) * kItemChunk + within;
        const int it = a.items[slot];
        if (it < 0) return;                                  // a slot of the budget this step does not use
        b = it >> 16;
        qti = it & 0xffff;
    } else {
        int id = blockIdx.x;
        const int n = a.total_wg;
        const int q8 = n >> 3, r8 = n & 7, x = id & 7;
        // blocks with equal (id % 8) share an XCD: give each XCD a contiguous range
        id = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (id >> 3);
        g = id % G;                id /= G;
        qti = id % a.n_items;      id /= a.n_items;
        kvh = id % a.Hkv;
        b = id / a.Hkv;
    }
    const int head = kvh * G + g;

    const int q_start = a.cu_q[b], q_len = a.cu_q[b + 1] - q_start;
    const int k_start = a.cu_k[b], k_len = a.cu_k[b + 1] - k_start;
    const int pos_off = k_len - q_len;                   // query i sits at key position i + off

    // A workgroup handles one Q tile, or (pair = g > 0) g heavy / light pairs of them in turn:
    // under the causal mask tile i attends to i + 1 key blocks, so the pair (n - 1 - p, p) always
    // costs n + 1 of them; every workgroup of a sequence then runs equally long and the fixed
    // cost of a workgroup (launch, ramp-up, drain) is paid once per 2 g tiles.
    // with a work list the pairs are formed per sequence (from its own tile count), without one
    // from the longest sequence's (shorter ones skip the tiles they do not have)
    const int n_qtiles = a.items != nullptr ? (q_len + kBM - 1) / kBM : a.n_qtiles;
    const int n_pairs = (n_qtiles + 1) >> 1;
    const int n_pass = a.pair ? 2 * a.pair : 1;
#pragma unroll 1
    for (int pass = 0; pass < n_pass; ++pass) {
    int qt;
    if (a.pair) {
        const int p = qti * a.pair + (pass >> 1);
        if (p >= n_pairs) break;
        if ((pass & 1) && 2 * p == n_qtiles - 1) continue;       // the middle tile of an odd count: once
        qt = (pass & 1) ? p : n_qtiles - 1 - p;
    } else {
        qt = n_qtiles - 1 - qti;                         // heaviest tiles first
    }
    const int q0 = qt * kBM;
    if (q0 >= q_len) continue;                           // uniform for the workgroup
    if (pass) __syncthreads();                           // every wave is done with the LDS tiles
    // every per-lane constant is derived inside the pass from a value the compiler cannot see
    // through: hoisted out of the loop they would stay live across it and spill
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31;
    const int hh = lane >> 5;

    // ---- Q^T fragments: lane holds Q[row r32][16 s + 8 hh .. +7] ------------------------
    const int my_q = q0 + 32 * w + r32;                  // query index within the sequence
    const int my_q_c = min(my_q, q_len - 1);
    vec8 qfrag[KSTEPS];
    {
        const uint16_t* qp = a.q + (long)(q_start + my_q_c) * a.q_stride_t + (long)head * a.q_stride_h;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
            qfrag[s] = __builtin_bit_cast(vec8, *reinterpret_cast<const u32x4_t*>(qp + 16 * s + 8 * hh));
    }
    const int my_pos = my_q + pos_off;                   // last key this row may see
    // fused Q transform: the row's position, requested with the Q loads (its table row follows
    // the first tile's DMA)
    // (an unconditional load from an address that is always valid: a conditional one would be
    // waited for where its branch ends, ahead of the first tile's requests)
    // (no positions given: a query's position is its place in the context - what a prefill step
    // feeds - and the table row can be requested without waiting for anything)
    const int my_rope_pos = a.positions == nullptr ? my_q_c + pos_off
        : *(a.cos_sin != nullptr ? a.positions + (q_start + my_q_c) : a.cu_q);
    const int wave_min_pos = q0 + 32 * w + pos_off;
    const int wave_max_pos = min(q0 + 32 * w + 31, q_len - 1) + pos_off;
    const bool wave_live = q0 + 32 * w < q_len;

    const int last_pos = min(q0 + kBM - 1, q_len - 1) + pos_off;
    const int n_tiles = last_pos / kBN + 1;

    // ---- staging: global -> LDS by LDS-DMA (no staging registers, no ds_write) ----------
    // One wave-instruction moves 64 x 16 B to 1 KiB of consecutive LDS (lane l lands at
    // base + 16 l), so the XOR swizzle of the LDS image is applied to the SOURCE address:
    // the slot (row, pos) of the image receives global chunk pos ^ x(row), which is exactly
    // where tile_off(row, ch) looks for chunk ch.
    int st_row[ITER], st_ch[ITER];
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int c = i * kThreads + tid;               // linear 16-byte slot of the tile image
        st_row[i] = c / CPR;
        const int pos = c % CPR;
        st_ch[i] = pos ^ ((((st_row[i] & 3) << 2) | ((st_row[i] >> 2) & 3)) & (CPR - 1));
    }
    const int bs_mask = (1 << a.bs_log2) - 1;
    const int* bt = PAGED ? a.block_tables + (long)b * a.bt_stride : nullptr;
    const long head_off = (long)kvh * a.k_stride_h;

    auto load_block_ids = [&](int t, int (&ids)[ITER]) {
        if (PAGED) {
#pragma unroll
            for (int i = 0; i < ITER; ++i) {
                const int tok = min(t * kBN + st_row[i], k_len - 1);
                ids[i] = bt[tok >> a.bs_log2];
            }
        }
    };
    auto stage_tile = [&](int t, const int (&ids)[ITER]) {
        char* kb = smem + (t & 1) * (2 * TILE_BYTES);
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int tok = min(t * kBN + st_row[i], k_len - 1);
            long off;
            if (PAGED)
                off = (long)ids[i] * a.cs_b + (long)(tok & bs_mask) * a.k_stride_t + head_off + st_ch[i] * 8;
            else
                off = (long)(k_start + tok) * a.k_stride_t + head_off + st_ch[i] * 8;
            char* dst = kb + (i * kThreads + w * 64) * 16;          // wave-uniform
            glds16(a.k + off, dst);
            glds16(a.v + off, dst + TILE_BYTES);
        }
    };

    // UNI: constant per-lane byte offsets inside a tile, the rest of the address is uniform
    uint32_t voff[ITER];
#pragma unroll
    for (int i = 0; i < ITER; ++i)
        voff[i] = (uint32_t)st_row[i] * (uint32_t)(a.k_stride_t * 2) + (uint32_t)st_ch[i] * 16u;
    // the tile's block id through the scalar unit (the table is read-only for the kernel): a
    // vector load would be waited for with vmcnt(0), i.e. together with the K/V requests in flight
    auto tile_block = [&](int t) -> int {
        if (!PAGED) return 0;
        typedef const __attribute__((address_space(4))) int* const_int_p;
        return *(const_int_p)(uintptr_t)(bt + ((t * kBN) >> a.bs_log2));
    };
    auto stage_tile_uni = [&](int t, int blk) {
        char* kb = smem + (t & 1) * (2 * TILE_BYTES);
        const int tok0 = t * kBN;
        const long off = PAGED ? (long)blk * a.cs_b + (long)(tok0 & bs_mask) * a.k_stride_t + head_off
                               : (long)(k_start + tok0) * a.k_stride_t + head_off;
        const int bytes = min(kBN, k_len - tok0) * (int)(a.k_stride_t * 2);
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t*>(a.k + off), 0, bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t*>(a.v + off), 0, bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            char* dst = kb + (i * kThreads + w * 64) * 16;          // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rk, (__attribute__((address_space(3))) void*)dst, 16, (int)voff[i], 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rv, (__attribute__((address_space(3))) void*)(dst + TILE_BYTES), 16, (int)voff[i], 0, 0, 0);
        }
    };
    int blk_next = 0;

    // fp8 cache: thread owns chunks c = i * 256 + tid of the 64 x CPR8 tile, i.e. row c / CPR8,
    // 16 fp8 values = LDS chunks 2 (c % CPR8) and 2 (c % CPR8) + 1 of that row
    auto load_block_ids8 = [&](int t, int (&ids)[ITER8 > 0 ? ITER8 : 1]) {
#pragma unroll
        for (int i = 0; i < ITER8; ++i) {
            const int tok = min(t * kBN + (i * kThreads + tid) / CPR8, k_len - 1);
            ids[i] = bt[tok >> a.bs_log2];
        }
    };
    auto load_tile8 = [&](int t, const int (&ids)[ITER8 > 0 ? ITER8 : 1],
                          u32x4_t (&kr)[ITER8 > 0 ? ITER8 : 1], u32x4_t (&vr)[ITER8 > 0 ? ITER8 : 1]) {
#pragma unroll
        for (int i = 0; i < ITER8; ++i) {
            const int c = i * kThreads + tid;
            const int tok = min(t * kBN + c / CPR8, k_len - 1);
            const long off = (long)ids[i] * a.cs_b + (long)(tok & bs_mask) * a.k_stride_t + head_off
                             + (c % CPR8) * 16;                      // bytes = elements
            kr[i] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint8_t*>(a.k) + off);
            vr[i] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint8_t*>(a.v) + off);
        }
    };
    auto write_tile8 = [&](int t, const u32x4_t (&kr)[ITER8 > 0 ? ITER8 : 1],
                           const u32x4_t (&vr)[ITER8 > 0 ? ITER8 : 1]) {
        char* kb = smem + (t & 1) * (2 * TILE_BYTES);
#pragma unroll
        for (int i = 0; i < ITER8; ++i) {
            const int c = i * kThreads + tid;
            const int row = c / CPR8, ch = 2 * (c % CPR8);
            u32x4_t lo, hi;
            widen_fp8x16<CODE>(kr[i], lo, hi);
            *reinterpret_cast<u32x4_t*>(kb + tile_off<CPR>(row, ch)) = lo;
            *reinterpret_cast<u32x4_t*>(kb + tile_off<CPR>(row, ch + 1)) = hi;
            widen_fp8x16<CODE>(vr[i], lo, hi);
            *reinterpret_cast<u32x4_t*>(kb + TILE_BYTES + tile_off<CPR>(row, ch)) = lo;
            *reinterpret_cast<u32x4_t*>(kb + TILE_BYTES + tile_off<CPR>(row, ch + 1)) = hi;
        }
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x16_t acc_o[NDB];
#pragma unroll
    for (int n = 0; n < NDB; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc_o[n][e] = 0.f;

    int ids_next[ITER];
    int ids8_next[ITER8 > 0 ? ITER8 : 1];
    u32x4_t kraw[ITER8 > 0 ? ITER8 : 1], vraw[ITER8 > 0 ? ITER8 : 1];
    if constexpr (KV8) {
        int ids[ITER8 > 0 ? ITER8 : 1];
        load_block_ids8(0, ids);
        load_tile8(0, ids, kraw, vraw);
        if (n_tiles > 1) load_block_ids8(1, ids8_next);
    } else if constexpr (UNI) {
        stage_tile_uni(0, tile_block(0));
        if (n_tiles > 1) blk_next = tile_block(1);
    } else {
        int ids[ITER];
        load_block_ids(0, ids);
        stage_tile(0, ids);
        if (n_tiles > 1) load_block_ids(1, ids_next);
    }
    // nothing that waits for a load (the table address needs the row's position) moves above
    // the first tile's requests
    __builtin_amdgcn_sched_barrier(0);
    // ---- q-norm + RoPE of the Q rows, fused here: a Q row is read by exactly one workgroup,
    //      so the projection's output never has to be rewritten in place and read again (K is
    //      shared by many workgroups and is transformed by the store kernel beforehand).  Pairs
    //      (e, e + D/2) of a row sit in one lane: fragments s and s + KSTEPS/2.  The table row
    //      was requested before the first tile's DMA; the arithmetic runs in that DMA's shadow.
    if (a.cos_sin != nullptr) {
        if (qw_pending) {                                // uniform: the workgroup's first Q tile
            if (threadIdx.x < D / 8) s_qw[threadIdx.x] = qw_mine;
            __syncthreads();
            qw_pending = false;
        }
        // the rotary table row of the lane's query: requested here, after the first tile's DMA,
        // where few other values are live (requested earlier, the 64 registers were spilled one
        // load at a time, each behind its own vmcnt(0)); the sum of squares below covers part
        // of the latency
        f32x4_t cs_raw[KSTEPS / 2][4];
        {
            const float* cs = a.cos_sin + (long)my_rope_pos * D;
#pragma unroll
            for (int s = 0; s < KSTEPS / 2; ++s) {
                cs_raw[s][0] = *reinterpret_cast<const f32x4_t*>(cs + 16 * s + 8 * hh);
                cs_raw[s][1] = *reinterpret_cast<const f32x4_t*>(cs + 16 * s + 8 * hh + 4);
                cs_raw[s][2] = *reinterpret_cast<const f32x4_t*>(cs + D / 2 + 16 * s + 8 * hh);
                cs_raw[s][3] = *reinterpret_cast<const f32x4_t*>(cs + D / 2 + 16 * s + 8 * hh + 4);
            }
        }
        // two passes over the packed fragments (sum of squares, then scale + rotate one pair of
        // fragments at a time) instead of one over an unpacked fp32 copy of the row: with the
        // 64 table registers in flight that copy did not fit, and the compiler spilled the
        // table loads one by one, each behind its own vmcnt(0)
        float ss = 0.f;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            float f[8];
            unpack8<CODE>(__builtin_bit_cast(u32x4_t, qfrag[s]), f);
#pragma unroll
            for (int i = 0; i < 8; ++i) ss += f[i] * f[i];
        }
        ss += __shfl_xor(ss, 32, 64);                    // the other half of the row
        const float rstd = a.q_w != nullptr ? rsqrtf(ss / (float)D + a.eps) : 1.f;
#pragma unroll
        for (int s = 0; s < KSTEPS / 2; ++s) {
            float f1[8], f2[8];
            unpack8<CODE>(__builtin_bit_cast(u32x4_t, qfrag[s]), f1);
            unpack8<CODE>(__builtin_bit_cast(u32x4_t, qfrag[s + KSTEPS / 2]), f2);
            if (a.q_w != nullptr) {
                float w1[8], w2[8];
                unpack8<CODE>(s_qw[2 * s + hh], w1);
                unpack8<CODE>(s_qw[2 * (s + KSTEPS / 2) + hh], w2);
#pragma unroll
                for (int i = 0; i < 8; ++i) { f1[i] *= rstd * w1[i]; f2[i] *= rstd * w2[i]; }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float c = cs_raw[s][i >> 2][i & 3], sn = cs_raw[s][2 + (i >> 2)][i & 3];
                const float x1 = f1[i], x2 = f2[i];
                f1[i] = x1 * c - x2 * sn;
                f2[i] = x1 * sn + x2 * c;
            }
            qfrag[s] = __builtin_bit_cast(vec8, pack8<CODE>(f1));
            qfrag[s + KSTEPS / 2] = __builtin_bit_cast(vec8, pack8<CODE>(f2));
        }
    }
    if constexpr (KV8) write_tile8(0, kraw, vraw);
    __syncthreads();                         // drains the LDS-DMA (vmcnt) and meets the waves
    MVL_PROBE(1, wall_clock64());
    MVL_PROBE(4, (unsigned long long)n_tiles);
    MVL_PROBE(5, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4));    // HW_ID
    MVL_PROBE(6, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20));   // XCC_ID

    // transposed-read lane constants: 16-lane group gq = lane >> 4; (gq & 1) picks the
    // 16-column half of a 32-wide d block, hh = gq >> 1 picks the key rows
    const int i16 = lane & 15;
    const int tr_row0 = 4 * hh + (i16 >> 2);
    const int tr_chunk0 = 2 * ((lane >> 4) & 1) + ((i16 & 3) >> 1);
    const int tr_sub = (i16 & 1) * 8;
    // LDS offsets of the lane's V fragments inside a {K, V} buffer, for d block n and the two
    // 8-row halves of a k-slot group; the 16 (s) and 32 (sub) row steps leave the swizzle term
    // unchanged and go into the reads' immediate offsets
    int v_frag_off[NDB][2];
#pragma unroll
    for (int n = 0; n < NDB; ++n)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi)
            v_frag_off[n][hi] = TILE_BYTES + tile_off<CPR>(tr_row0 + 8 * hi, 4 * n + tr_chunk0) + tr_sub;

    // One KV tile for this wave: S^T = K.Q^T, online softmax, O^T += V^T.P^T.
    // m_run is the reference point of the exponentials, not necessarily the true row
    // maximum: it is only moved (and O, l rescaled) when some row's maximum outgrows it by
    // more than kRescaleThr (log2 units), so most tiles skip the O-wide rescale; P stays
    // <= 2^kRescaleThr, far inside bf16/fp16 range.
    // stage_late(): the request for the next tile, issued between the softmax and the PV
    // products (UNI only; elsewhere a no-op and the request precedes the tile)
    auto process_tile = [&](auto mask_tag, int kv0, const char* k_lds, const char* v_lds,
                            auto&& stage_late) {
        constexpr bool MASK = decltype(mask_tag)::value;
        // the two MFMA phases of a tile run at priority 1, its softmax at 0: when the two waves of
        // a SIMD compete, the one feeding the matrix pipe issues first and the other's vector-ALU
        // work takes the slots in between (+0.5 % at 16k tokens, +4 % at 512; the opposite
        // assignment loses 2-3 %)
        __builtin_amdgcn_s_setprio(1);
        f32x16_t s_acc[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s_acc[sub][e] = 0.f;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const u32x4_t kv = *reinterpret_cast<const u32x4_t*>(
                    k_lds + tile_off<CPR>(32 * sub + r32, 2 * s + hh));
                s_acc[sub] = DT::mfma32(__builtin_bit_cast(vec8, kv), qfrag[s], s_acc[sub]);
            }
        }
        // lane (r32, hh) holds, for query row r32, keys kv0 + 32 sub + (e&3) + 8 (e>>2) + 4 hh
        __builtin_amdgcn_s_setprio(0);
        float m_tile = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (MASK) {
                    const int kv = kv0 + 32 * sub + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (kv > my_pos) s_acc[sub][e] = -INFINITY;
                }
                m_tile = fmaxf(m_tile, s_acc[sub][e]);
            }
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64)) * a.scale_log2;   // scale > 0
        // rescale only when needed (wave-uniform decision; NaN-safe: -inf - -inf counts)
        if (!__all(m_tile - m_run <= kRescaleThr)) {
            const float m_new = fmaxf(m_run, m_tile);
            // a row with nothing visible yet keeps m = -inf; exp2 then sees -inf -> 0
            const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int n = 0; n < NDB; ++n) acc_o[n] *= alpha;
            m_run = m_new;
        }
        const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
        // the softmax arithmetic shares the SIMD's issue slots with the MFMAs (the kernel's time
        // is the sum of the two), so it runs two values per instruction where the ISA allows:
        // packed FMA for the exponent arguments, packed adds for the row sums
        typedef __attribute__((ext_vector_type(2))) float f32x2_t;
        const f32x2_t scale2 = {a.scale_log2, a.scale_log2}, negm2 = {-m_use, -m_use};
        f32x2_t psum2 = {0.f, 0.f};
        vec8 pfrag[2][2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float p[8];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const f32x2_t sv = {s_acc[sub][8 * s + j], s_acc[sub][8 * s + j + 1]};
                    const f32x2_t arg = __builtin_elementwise_fma(sv, scale2, negm2);
                    const f32x2_t pv = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
                    psum2 += pv;
                    p[j] = pv[0];
                    p[j + 1] = pv[1];
                }
                pfrag[sub][s] = __builtin_bit_cast(vec8, pack8<CODE>(p));
            }
        l_run += psum2[0] + psum2[1];
        stage_late();
        __builtin_amdgcn_s_setprio(1);
        {
            // Through the builtin the compiler waits for EVERY LDS-DMA request in flight
            // (vmcnt(0)) before the first transposed read, i.e. for the next tile, requested at
            // the top of this iteration and needed only after its closing barrier.  As inline asm
            // the reads carry no such wait; their own lgkmcnt waits are written out: fragments
            // of step i + 2 are requested before step i's MFMA, and LDS returns in order, so "at
            // most 4 outstanding" means the pair of step i has arrived.
            constexpr int NPV = 4 * NDB;
            const uint32_t kb32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)k_lds;
            uint32_t va[NDB][2];
#pragma unroll
            for (int n = 0; n < NDB; ++n) {
                va[n][0] = kb32 + v_frag_off[n][0];
                va[n][1] = kb32 + v_frag_off[n][1];
            }
            s16x4_t vlo[NPV], vhi[NPV];
            auto v_read = [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int n = i / 4, sub = (i >> 1) & 1, s = i & 1;
                constexpr int off = (32 * sub + 16 * s) * CPR * 16;
                // (asm operands do not count as uses that capture: go through named locals)
                const uint32_t a0 = va[n][0], a1 = va[n][1];
                s16x4_t lo, hi;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a0), "n"(off));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a1), "n"(off));
                vlo[i] = lo;
                vhi[i] = hi;
            };
            v_read(std::integral_constant<int, 0>{});
            v_read(std::integral_constant<int, 1>{});
            static_for<0, NPV>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i + 2 < NPV) v_read(std::integral_constant<int, i + 2>{});
                s16x4_t lo = vlo[i], hi = vhi[i];
                if constexpr (i + 2 < NPV) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(lo), "+v"(hi));
                else if constexpr (i + 1 < NPV) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(lo), "+v"(hi));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi));
                const s16x8_t av = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                acc_o[i / 4] = DT::mfma32(__builtin_bit_cast(vec8, av), pfrag[(i >> 1) & 1][i & 1], acc_o[i / 4]);
            });
        }
    };

    // Two loops instead of a per-tile branch (one copy of the tile body per loop keeps the
    // register allocation at the single-variant level): tiles wholly below the workgroup's
    // first query row need no mask; the last few (diagonal) tiles apply it.
    auto run_tiles = [&](auto mask_tag, int t_begin, int t_end) {
        for (int t = t_begin; t < t_end; ++t) {
            const int kv0 = t * kBN;
            const bool run = wave_live && kv0 <= wave_max_pos;
            auto stage_next = [&]() {
                if (t + 1 < n_tiles) {
                    // buffer (t+1)&1 was last read in iteration t-1; the barrier that closed it
                    // ordered those reads before this DMA (KV8: before the writes further down)
                    if constexpr (KV8) {
                        load_tile8(t + 1, ids8_next, kraw, vraw);
                        if (t + 2 < n_tiles) load_block_ids8(t + 2, ids8_next);
                    } else if constexpr (UNI) {
                        stage_tile_uni(t + 1, blk_next);
                        if (t + 2 < n_tiles) blk_next = tile_block(t + 2);
                    } else {
                        stage_tile(t + 1, ids_next);
                        if (t + 2 < n_tiles) load_block_ids(t + 2, ids_next);
                    }
                }
            };
            // UNI: a wave that runs the tile issues its requests after the tile's softmax, where
            // they share their issue slots with vector-ALU work only; next to the 16 K fragment
            // reads at the top of the tile they cost more (2-3 % of the kernel at >= 4k tokens).
            // They still have the PV products and the barrier's wait to land.
            if (!UNI || !run) stage_next();
            if (run) {
                const char* k_lds = smem + (t & 1) * (2 * TILE_BYTES);
                if constexpr (UNI) process_tile(mask_tag, kv0, k_lds, k_lds + TILE_BYTES, stage_next);
                else process_tile(mask_tag, kv0, k_lds, k_lds + TILE_BYTES, []() {});
            }
            if (t + 1 < n_tiles) {
                if constexpr (KV8) write_tile8(t + 1, kraw, vraw);
                __syncthreads();                     // tile t+1 has landed, tile t is free
            }
        }
    };
    const int t_split = min(n_tiles, (q0 + pos_off + 1) / kBN);   // uniform for the workgroup
    run_tiles(std::false_type{}, 0, t_split);
    MVL_PROBE(2, wall_clock64());
    run_tiles(std::true_type{}, t_split, n_tiles);
    MVL_PROBE(3, wall_clock64());

    __builtin_amdgcn_s_setprio(0);
    // ---- epilogue: O[q][d] = O^T / l --------------------------------------------------
    l_run += __shfl_xor(l_run, 32, 64);
    {
        // A row's 8 d-values of group k = 4 n + g8 are split over the two lane halves (4 + 4), so
        // the natural store is 8 bytes per lane.  One half exchange per dword between groups k
        // and k + 1 (v_permlane32_swap: the upper half of the first operand swaps with the lower
        // half of the second) leaves the lower lanes with the 16 contiguous bytes of group k and
        // the upper lanes with those of group k + 1: half as many store instructions for the
        // same bytes (the tail of a workgroup is bound by the issue of its stores).
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        const bool live = my_q < q_len;
        uint16_t* op = a.out + (long)(q_start + (live ? my_q : 0)) * a.o_stride_t + (long)head * a.o_stride_h
                       + 8 * hh;
#pragma unroll
        for (int n = 0; n < NDB; ++n)
#pragma unroll
            for (int g8 = 0; g8 < 4; g8 += 2) {
                // registers 4 g8 .. 4 g8 + 3 are d = 32 n + 8 g8 + 4 hh + (0..3)
                uint32_t ax = pack2<CODE>(acc_o[n][4 * g8 + 0] * inv, acc_o[n][4 * g8 + 1] * inv);
                uint32_t ay = pack2<CODE>(acc_o[n][4 * g8 + 2] * inv, acc_o[n][4 * g8 + 3] * inv);
                uint32_t bx = pack2<CODE>(acc_o[n][4 * g8 + 4] * inv, acc_o[n][4 * g8 + 5] * inv);
                uint32_t by = pack2<CODE>(acc_o[n][4 * g8 + 6] * inv, acc_o[n][4 * g8 + 7] * inv);
                const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                if (live)
                    *reinterpret_cast<u32x4_t*>(op + 32 * n + 8 * g8) = u32x4_t{rx[0], ry[0], rx[1], ry[1]};
            }
    }
    }   // pass
}

template <int CODE, int D, bool PAGED, bool KV8 = false, bool UNI = false>
int launch_prefill(PrefillArgs a, hipStream_t st) {
    const size_t lds = 2 * 2 * (size_t)kBN * D * 2;      // double-buffered K and V tiles
    flash_prefill_kernel<CODE, D, PAGED, KV8, UNI><<<a.total_wg, kThreads, lds, st>>>(a);
    return MVL_CHECK_LAUNCH();
}

}  // namespace

// positions / cos_sin / q_w / eps: see PrefillArgs (cos_sin null: q is already transformed).
// paged == 0: k/v are packed [T, Hkv, D] tensors (token stride k_stride_t, head stride
// k_stride_h), cu_k may equal cu_q.  paged == 1: k/v are caches addressed by
// (block stride cs_b, token stride k_stride_t, head stride k_stride_h) through block_tables;
// kv_fp8 != 0: the caches hold OCP fp8 (e4m3) bytes (strides in bytes).
// items (optional, n_item_slots entries): the step's work list, see PrefillArgs; max_q_len is
// then only required to be positive.
extern "C" int mvl_flash_attention_prefill(
        const void* q, long q_stride_t, long q_stride_h, const void* k, const void* v,
        long k_stride_t, long k_stride_h, long cs_b, void* out, long o_stride_t, long o_stride_h,
        const int* cu_q, const int* cu_k, const int* block_tables, int bt_stride, int block_size,
        float scale, int B, int Hq, int Hkv, int D, int max_q_len, int paged, int dtype,
        const int* positions, const float* cos_sin, const void* q_w, float eps, int kv_fp8,
        const int* items, int n_item_slots, void* stream) {
    if (B == 0 || max_q_len <= 0) return 0;
    if (items != nullptr && n_item_slots == 0) return 0;
    if (items != nullptr && (n_item_slots < 0 || n_item_slots % 32 != 0 || B > 32767)) return -1;
    if ((D != 64 && D != 128) || Hq % Hkv != 0) return -1;
    if (paged && (block_size & (block_size - 1))) return -1;
    if (kv_fp8 && !paged) return -1;
    // the output is written 16 bytes at a time
    if (((uintptr_t)out & 15) || (o_stride_t & 7) || (o_stride_h & 7)) return -1;
    PrefillArgs a;
    a.q = (const uint16_t*)q; a.q_stride_t = q_stride_t; a.q_stride_h = q_stride_h;
    a.k = (const uint16_t*)k; a.v = (const uint16_t*)v;
    a.k_stride_t = k_stride_t; a.k_stride_h = k_stride_h; a.cs_b = cs_b;
    a.out = (uint16_t*)out; a.o_stride_t = o_stride_t; a.o_stride_h = o_stride_h;
    a.cu_q = cu_q; a.cu_k = cu_k; a.block_tables = block_tables; a.bt_stride = bt_stride;
    int l2 = 0;
    while (paged && (1 << l2) < block_size) ++l2;
    a.bs_log2 = l2;
    a.scale_log2 = scale * kLog2e;
    a.B = B; a.Hq = Hq; a.Hkv = Hkv;
    a.n_qtiles = (max_q_len + kBM - 1) / kBM;
    static const char* pair_env = getenv("MVL_PREFILL_PAIR");
    // one heavy / light pair per workgroup (MVL_PREFILL_PAIR: 0 = one Q tile per workgroup, g = g
    // pairs).  Several pairs per workgroup once paid for the workgroup's fixed cost; since the
    // prologue got short, more and smaller workgroups balance better (32 x 512, 16 q heads:
    // 0.073 ms with two pairs, 0.068 ms with one)
    const int n_pairs = (a.n_qtiles + 1) / 2;
    a.pair = pair_env ? atoi(pair_env) : 1;
    a.n_items = a.pair ? (n_pairs + a.pair - 1) / a.pair : a.n_qtiles;
    a.total_wg = B * Hq * a.n_items;
    a.items = items;
    if (items != nullptr) {          // one heavy / light pair of one sequence per slot
        a.pair = 1;
        a.n_items = n_item_slots;
        a.total_wg = Hq * n_item_slots;
    }
    a.positions = positions; a.cos_sin = cos_sin; a.q_w = (const uint16_t*)q_w; a.eps = eps;
    // one uniform base per K/V tile: blocks of a multiple of 64 tokens, tile strides within 31 bits
    static const char* uni_env = getenv("MVL_PREFILL_UNI");
    const bool uni = (uni_env ? atoi(uni_env) != 0 : true) && block_size % kBN == 0
                     && k_stride_t * 2 * kBN < (1l << 31);
    auto st = (hipStream_t)stream;
#define MVL_PRE(CODE, DD)                                                         \
    return kv_fp8 ? launch_prefill<CODE, DD, true, true>(a, st)                   \
         : !paged ? launch_prefill<CODE, DD, false, false, true>(a, st)           \
         : uni    ? launch_prefill<CODE, DD, true, false, true>(a, st)            \
                  : launch_prefill<CODE, DD, true>(a, st)
    if (dtype == 0) { if (D == 128) MVL_PRE(0, 128); else MVL_PRE(0, 64); }
    else            { if (D == 128) MVL_PRE(1, 128); else MVL_PRE(1, 64); }
#undef MVL_PRE
}
