// Compute-bound GEMM for prefill-sized row counts:  out[M, N] = x[M, K] . W[N, K]^T, optionally
// with the SwiGLU epilogue  out[M, N/2] = silu(x . Wg^T) * (x . Wu^T)  for W = [Wg; Wu], which
// saves the pass of the activation kernel over the [M, N] intermediate (for Qwen3-8B at 16384
// tokens: 805 MB written and read back per layer).
//
// Shape of the kernel (gfx950, wave64):
//  * one workgroup of 8 waves owns a 256 x 256 output tile and walks K in 64-element tiles;
//    LDS holds two K-tiles, each as four half-tiles of 128 rows x 128 bytes:
//    A0 / A1 = x rows 0..127 / 128..255 of the tile; B0 / B1 = for each of the four wave columns
//    the first / last 32 of its 64 consecutive output columns (so that a wave stores whole
//    128-byte rows); SwiGLU: B0 = 128 gate columns, B1 = the matching 128 up columns;
//  * wave (wr, wc), wr = 0..1, wc = 0..3, owns x rows {64 wr .. +63} of A0 ("top") and of A1
//    ("bottom") and W rows {32 wc .. +31} of B0 ("left") and of B1 ("right"): four 64 x 32
//    quadrants, 128 accumulator registers.  A K-tile is four phases, one quadrant each, in the
//    order (top,left) (top,right) (bottom,right) (bottom,left), so the fragment reads are
//    12 / 4 / 8 / 0 ds_read_b128 and the half-tiles are first needed in the order A0 B0 B1 A1;
//  * every phase requests one half-tile (2 LDS-DMA instructions per wave, buffer form: 32-bit
//    per-lane offsets, range-checked) and waits with a
//    counted vmcnt for the half-tile requested four phases earlier; a half-tile is read no
//    earlier than the phase after that wait and overwritten no earlier than two phases after
//    its last read.  Nothing in the loop drains the vector-memory queue;
//  * the two wave rows run one barrier apart (wr = 1 takes one extra barrier before the loop,
//    wr = 0 one after it): a phase is  reads + request | barrier | 16 MFMAs | barrier, so on
//    every SIMD one wave issues MFMAs while the other reads LDS and requests memory;
//  * LDS image rows are 128 bytes with 16-byte chunk' = chunk ^ (row & 7), applied on the source
//    address of the DMA (its destination is lane-linear) and on the fragment reads;
//  * the product is computed transposed (MFMA A operand = W rows, B operand = x rows): a lane
//    holds 4 consecutive output columns of a row;
//  * the grid is one workgroup per CU; a workgroup walks tiles g, g + G, ... of a logical order
//    (panels of 8 row tiles walked column by column) in which the workgroups that share an XCD
//    (id mod 8) hold a compact block of 32 tiles at any time, so each XCD's L2 serves an x panel
//    to 4 column tiles and a W panel to 8 row tiles.  The half-tile pipeline runs on across tile
//    boundaries: the last two K-tiles of a tile already request the first K-tiles of the next
//    one, and the stores of a finished tile drain under the next tile's MFMAs (rows beyond M
//    are dropped by the range check of the buffer stores).
//
// Two optional epilogue duties take the (residual add + RMSNorm) passes out of a prefill layer
// (reference: src/myvllm/layers/layernorm.py:19-22, src/myvllm/models/qwen3.py:191-195,216):
//  * RES (o / down projections): out is the residual stream, updated in place,
//    r = round(round(acc + bias) + residual);
//  * RS, row scale (qkv / gate_up projections, whose weights carry the norm's gamma): every output
//    row is multiplied by rsqrt(sum(x[row]^2) / K + eps) before bias / SwiGLU, which is the RMSNorm
//    of the x row moved behind the GEMM:  (x rstd) . (W gamma)^T = rstd (x . (W gamma)^T).  The
//    sums of squares come from the x fragments the waves hold for their MFMAs anyway (one v_dot2c
//    per register, a quarter of a wave row's x rows per wave, issued where the wave otherwise
//    only waits for the other wave row's MFMAs): no second pass over x, no buffer, nothing
//    shared between workgroups.
//
// Call sites: the MLP's gate_up projection on prefill steps (the reference's
// F.linear + SiluAndMul, src/myvllm/layers/linear.py:109, src/myvllm/layers/activation.py:8-16),
// and any projection for which the start-up measurement prefers it to the library.
#include "common.h"
#include <type_traits>
#include <cstdlib>

using namespace mvl;

namespace {

#ifndef MVL_BIG_ABLATE
#define MVL_BIG_ABLATE 0     // measurement builds only; bits: 1 no fragment reads in the loop,
#endif                       // 2 no DMA requests in the loop, 4 no barriers in the loop
#ifndef MVL_BIG_STAMP
#define MVL_BIG_STAMP 0      // measurement builds only: clock stamps of one K-tile's phases
#endif                       // (waves 0 and 4 of workgroup 0) overwrite the start of `out`
constexpr int kBigThreads = 512;
constexpr int kHalf = 128 * 128;            // bytes of a half-tile
constexpr int kBuf = 4 * kHalf;             // one K-tile: A0 B0 B1 A1
constexpr int kBigBK = 64;
constexpr int kBigLds = 2 * kBuf + 8 * 4096;   // two K-tiles + a 4 KiB output block per wave

struct BigArgs {
    const uint16_t* x; long x_stride;       // [M, K], row stride in elements
    const uint16_t* w;                      // [N, K] row-major
    const uint16_t* bias;                   // may be null
    uint16_t* out; long o_stride;
    unsigned out_bytes;                     // extent of `out` for the store range check
    unsigned x_bytes, w_bytes;              // extents of x and W (buffer descriptors)
    int M, N, K;
    int up_off;                             // SwiGLU: row of W where the up projection starts
    int tiles_m, tiles_n, tiles, ktiles;
    int probe;                              // measurement only: 1 every tile loads tile 0's operands, 2 no stores
    float rs_eps, rs_inv_k;                 // RS: rstd = rsqrt(sum(x[row]^2) * rs_inv_k + rs_eps)
};

template <int N>
__device__ __forceinline__ void big_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void big_barrier() {
    if (MVL_BIG_ABLATE & 4) { __builtin_amdgcn_sched_barrier(0); return; }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// per-lane DMA source offsets of one output tile (32-bit byte offsets from x / W)
struct BigSrc {
    uint32_t a[2][2], b[2][2];              // [half][piece]
};

template <int CODE, bool SWI, bool RES = false, bool RS = false>
__global__ __launch_bounds__(kBigThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void big_gemm_kernel(BigArgs a) {
    static_assert(!(SWI && RES), "the residual epilogue belongs to the plain projections");
    using DT = DType<CODE>;
    using vec8 = typename DT::vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wv >> 2, wc = wv & 3;
    const int G = gridDim.x, bid = blockIdx.x;

    // ---- tile walk.  Round rd covers logical tiles [rd G, rd G + cnt); inside a round the
    // workgroups with equal (id mod 8) take consecutive logical tiles.
    auto tile_id = [&](int rd) -> int {
        const int base = rd * G;
        const int cnt = min(G, a.tiles - base);
        if (bid >= cnt) return -1;
        const int q = cnt >> 3, r = cnt & 7, xcd = bid & 7;
        return base + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    };
    auto tile_pos = [&](int id, int& tm, int& tn) {
        constexpr int PM = 8;
        const int per_panel = PM * a.tiles_n;
        const int panel = id / per_panel;
        const int rem = id - panel * per_panel;
        const int rows = min(PM, a.tiles_m - panel * PM);
        tn = rem / rows;
        tm = panel * PM + rem - tn * rows;
    };

    // ---- DMA sources.  A piece is 8 rows x 128 bytes; lane l moves chunk (l & 7) of image row
    // (l >> 3), which holds global chunk (l & 7) ^ (l >> 3).  Wave w moves pieces 2w, 2w + 1 of
    // each half-tile.
    const int prow = lane >> 3;
    const int pchunk = (lane & 7) ^ prow;
    auto tile_src = [&](int tm, int tn, BigSrc& s) {
        if (a.probe & 1) tm = tn = 0;
        const int m0 = tm * 256;
        const int n0 = SWI ? tn * 128 : tn * 256;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = h * 128 + (wv * 2 + i) * 8 + prow;
                const int xr = min(m0 + r, a.M - 1);                    // rows >= M: duplicates
                s.a[h][i] = (uint32_t)xr * (uint32_t)(a.x_stride * 2) + pchunk * 16;
                // image row r of half h <-> W row: SwiGLU gate / up column r; plain: wave wc's 64
                // consecutive columns are rows 32 wc .. +31 of B0 (first 32) and of B1 (last 32)
                const int r7 = r & 127;
                const int wrow = SWI ? (h ? a.up_off : 0) + n0 + r7
                                     : n0 + (r7 >> 5) * 64 + h * 32 + (r7 & 31);
                s.b[h][i] = (uint32_t)wrow * (uint32_t)(a.K * 2) + pchunk * 16;
            }
    };
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(a.w), 0, (int)a.w_bytes, 0x00020000);

    // request half-tile J (0 A0, 1 B0, 2 B1, 3 A1) of K-tile k of this workgroup's current tile
    // (or, with `next`, of its next tile) into the buffer at LDS byte offset bofs; bofs, k and
    // next are wave-uniform
    BigSrc cur, nxt;
    auto stage = [&](int bofs, auto j_c, bool next, int k) {
        constexpr int J = decltype(j_c)::value;
        constexpr bool is_a = (J == 0 || J == 3);
        constexpr int h = (J == 2 || J == 3) ? 1 : 0;
        if (MVL_BIG_ABLATE & 2) return;
        char* dst = smem + bofs + J * kHalf + wv * 2048;
        const int kb = k * (kBigBK * 2);                 // uniform byte offset along K
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t off = is_a ? (next ? nxt.a[h][i] : cur.a[h][i])
                                      : (next ? nxt.b[h][i] : cur.b[h][i]);
            // buffer form of the LDS-DMA: per-lane offset in a VGPR, the K offset in an SGPR,
            // no address arithmetic on the vector ALU
            __builtin_amdgcn_raw_ptr_buffer_load_lds(is_a ? xrsrc : wrsrc,
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024),
                                                 16, (int)off, kb, 0, 0);
        }
    };

    // ---- fragment reads: rows (lane & 15) of a 16-row block, k-chunk 4 ks + (lane >> 4)
    const int frow = lane & 15, fq = lane >> 4;
    const int lane_off0 = frow * 128 + ((fq ^ (frow & 7)) << 4);
    const int a_off = lane_off0 + wr * (64 * 128);
    const int b_off = lane_off0 + wc * (32 * 128);

    f32x4_t acc[4][2][4];                    // [quadrant TL TR BR BL][W block][x block]
    vec8 af[2][4], bl[2][2], br[2][2];       // [k half][block]
    if (MVL_BIG_ABLATE & 1) {
        for (int ks = 0; ks < 2; ++ks) {
            for (int m = 0; m < 4; ++m) af[ks][m] = __builtin_bit_cast(vec8, u32x4_t{(uint32_t)lane, 1u, 2u, 3u});
            for (int n = 0; n < 2; ++n) { bl[ks][n] = af[ks][n]; br[ks][n] = af[ks][n + 2]; }
        }
    }

    auto read_a = [&](int bofs, auto j_c) {
        if (MVL_BIG_ABLATE & 1) return;
        constexpr int J = decltype(j_c)::value;
        const char* base = smem + bofs + J * kHalf;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                af[ks][m] = __builtin_bit_cast(
                    vec8, *reinterpret_cast<const u32x4_t*>(base + m * 2048 + (a_off ^ (ks << 6))));
    };
    auto read_b = [&](int bofs, auto j_c, vec8 (&bf)[2][2]) {
        if (MVL_BIG_ABLATE & 1) return;
        constexpr int J = decltype(j_c)::value;
        const char* base = smem + bofs + J * kHalf;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                bf[ks][n] = __builtin_bit_cast(
                    vec8, *reinterpret_cast<const u32x4_t*>(base + n * 2048 + (b_off ^ (ks << 6))));
    };
    // RS: sums of squares of the x rows.  The four waves of a wave row hold the same x fragments;
    // wave wc sums x block wc, a quarter of the work each, and the waves exchange the results
    // through LDS before the epilogue.  Which of its four fragment registers that block is
    // depends on the wave, so the wave reads its block once more into registers of its own (two
    // ds_read_b128 per half-tile; selecting among the MFMA operands instead costs a register
    // copy per fragment).  ssq[top / bottom][k half]: the lane's k-chunks of row 16 wc + frow.
    float ssq[2][2];
    const int duty_off = a_off + wc * 2048;
    auto sum_squares = [&](int bofs, auto j_c, int hb) {
        if constexpr (RS) {
            constexpr int J = decltype(j_c)::value;
            const char* base = smem + bofs + J * kHalf;
            u32x4_t v[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                v[ks] = *reinterpret_cast<const u32x4_t*>(base + (duty_off ^ (ks << 6)));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) ssq[hb][ks] = sumsq2<CODE>(v[ks][e], ssq[hb][ks]);
        }
    };
    auto mfma_quadrant = [&](auto q_c, vec8 (&bf)[2][2]) {
        constexpr int Q = decltype(q_c)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[Q][n][m] = DT::mfma16(bf[ks][n], af[ks][m], acc[Q][n][m]);
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // One K-tile, the g-th of this workgroup (buffer g & 1); u = its index inside the tile.
    // It requests the halves B1 / A1 of the K-tile that follows it in the workgroup's sequence
    // and A0 / B0 of the one after that (both may belong to the next tile).  A wait names the
    // half-tile the NEXT phase reads: 4 half-tiles (8 instructions) were requested after it, and
    // loads complete in order, so at most 8 outstanding instructions means it has landed
    // (whatever the epilogue's stores do: a store still pending only makes the wait stricter).
    // FIRST: the tile's first K-tile.  Every earlier request was waited for before the previous
    // tile's epilogue, so this K-tile waits for nothing: the first counted wait after the
    // epilogue's stores (which it would force to complete) comes a whole K-tile later.
    uint64_t stamps[16];
#define MVL_STAMP(i) do { if (MVL_BIG_STAMP) stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
    auto ktile = [&](auto first_c, int g, int u, int nk) {
        constexpr bool FIRST = decltype(first_c)::value != 0;
        const int B = (g & 1) * kBuf, NB = kBuf - B;
        const bool n1 = u + 1 >= nk, n2 = u + 2 >= nk;
        const int k1 = n1 ? u + 1 - nk : u + 1, k2 = n2 ? u + 2 - nk : u + 2;
        // phase 1: (top, left)
        MVL_STAMP(0);
        read_b(B, I1{}, bl);
        read_a(B, I0{});
        stage(NB, I2{}, n1, k1);
        if (!FIRST) big_wait_vmcnt<8>();
        MVL_STAMP(1);
        big_barrier();
        MVL_STAMP(2);
        mfma_quadrant(I0{}, bl);
        MVL_STAMP(3);
        big_barrier();
        // phase 2: (top, right)
        MVL_STAMP(4);
        read_b(B, I2{}, br);
        stage(NB, I3{}, n1, k1);
        sum_squares(B, I0{}, 0);             // the top x rows (A0: not overwritten before phase 3)
        if (!FIRST) big_wait_vmcnt<8>();
        MVL_STAMP(5);
        big_barrier();
        MVL_STAMP(6);
        mfma_quadrant(I1{}, br);
        MVL_STAMP(7);
        big_barrier();
        // phase 3: (bottom, right)
        MVL_STAMP(8);
        read_a(B, I3{});
        stage(B, I0{}, n2, k2);
        MVL_STAMP(9);
        big_barrier();
        MVL_STAMP(10);
        mfma_quadrant(I2{}, br);
        MVL_STAMP(11);
        big_barrier();
        // phase 4: (bottom, left), no reads
        MVL_STAMP(12);
        stage(B, I1{}, n2, k2);
        sum_squares(B, I3{}, 1);             // the bottom x rows (A1: safe until the next K-tile)
        if (!FIRST) big_wait_vmcnt<8>();
        MVL_STAMP(13);
        big_barrier();
        MVL_STAMP(14);
        mfma_quadrant(I3{}, bl);
        MVL_STAMP(15);
        big_barrier();
    };

    const __amdgpu_buffer_rsrc_t orsrc =
        __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);

    // acc[q][n][m][e] = out[row 16 m + frow][col 16 n + 4 fq + e] of quadrant q: 0 (top, left)
    // 1 (top, right) 2 (bottom, right) 3 (bottom, left).  A wave's output is 128 rows x 64
    // consecutive columns (SwiGLU: 32, gate quadrant 0 / 3 paired with up quadrant 1 / 2).  It
    // goes through a wave-private 4 KiB LDS block (32 rows x 128 B, SwiGLU 64 rows x 64 B; 16-byte
    // chunks XOR-ed with the row so that neither the 8-byte writes nor the 16-byte reads pile up
    // on a bank) and leaves as 16-byte stores that cover whole rows of the block: 8-byte stores
    // straight from the accumulator layout cost 4x the write requests, which the L2 channels
    // could not absorb (0.6B qkv at 16384 rows: 127 us with them, 98 us without any store).
    char* const ep = smem + 2 * kBuf + wv * 4096;
    // RES: the 16 residual chunks the lane will add to its 16 output chunks are requested BEFORE
    // the wait that precedes the epilogue (the fragment registers are dead by then), so their
    // latency is paid once per tile, together with the drain of the DMA queue.
    u32x4_t resv[2][2][4];                   // [top / bottom][x block pair][store]
    auto out_voff = [&](int tm, int tn, int hb, int blk, int i) -> uint32_t {
        constexpr int LPRW = SWI ? 4 : 8, RPI = 64 / LPRW;
        const int col_base = SWI ? tn * 128 + wc * 32 : tn * 256 + wc * 64;
        const int rr = i * RPI + lane / LPRW;
        const int row = tm * 256 + hb * 128 + wr * 64 + (SWI ? 0 : blk * 32) + rr;
        return (uint32_t)row * (uint32_t)(a.o_stride * 2) + (uint32_t)(col_base * 2 + (lane % LPRW) * 16);
    };
    auto pre_epilogue = [&](int tm, int tn) {
        if constexpr (RES) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int i = 0; i < 4; ++i)     // rows >= M: beyond out_bytes, read as 0
                        resv[hb][blk][i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(
                            orsrc, (int)out_voff(tm, tn, hb, blk, i), 0, 0));
        }
    };
    // RS: every wave needs the sums of all 64 top and 64 bottom rows of its wave row.  They pass
    // through the first 512 bytes of the output block of the wave row's first wave, which is
    // idle until the epilogue: write, barrier, read, barrier (both wave rows take both barriers,
    // so the one-barrier distance between them is kept).
    float rsv[2][4];
    auto exchange_sums = [&]() {
        if constexpr (RS) {
            float* xch = reinterpret_cast<float*>(smem + 2 * kBuf + (wr * 4) * 4096);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                float v = ssq[hb][0] + ssq[hb][1];           // a row's k-chunks: lanes frow + 16 fq
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (fq == 0) xch[(hb * 4 + wc) * 16 + frow] = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            big_barrier();
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int m = 0; m < 4; ++m) rsv[hb][m] = xch[(hb * 4 + m) * 16 + frow];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            big_barrier();
        }
    };
    auto epilogue = [&](int tm, int tn) {
        if (a.probe & 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int m = 0; m < 4; ++m) asm volatile("" ::"v"(acc[q][n][m]));
            return;
        }
        const int m0 = tm * 256;
        constexpr int ROWB = SWI ? 64 : 128;                 // bytes per row of the block
        constexpr int LPRW = ROWB / 16;                      // lanes per row on the way out
        constexpr int RPI = 64 / LPRW;                       // rows per store instruction
        const int col_base = SWI ? tn * 128 + wc * 32 : tn * 256 + wc * 64;
        // bias of this lane's columns: [side][n][e]
        float bias_v[2][2][4];
#pragma unroll
        for (int sd = 0; sd < 2; ++sd)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bias_v[sd][n][e] = 0.f;
                if (a.bias != nullptr) {
                    const int col = SWI ? (sd ? a.up_off : 0) + col_base + n * 16 + 4 * fq
                                        : col_base + sd * 32 + n * 16 + 4 * fq;
                    const u32x2_t b = *reinterpret_cast<const u32x2_t*>(a.bias + col);
                    unpack2<CODE>(b[0], bias_v[sd][n][0], bias_v[sd][n][1]);
                    unpack2<CODE>(b[1], bias_v[sd][n][2], bias_v[sd][n][3]);
                }
            }
        const int orow = lane / LPRW, ochunk = lane % LPRW;
        float rs[2][4];                                      // [top / bottom][x block]
        if constexpr (RS) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int m = 0; m < 4; ++m) rs[hb][m] = rsqrtf(rsv[hb][m] * a.rs_inv_k + a.rs_eps);
        }
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {                     // top / bottom rows of the wave
            const int ql = hb ? 3 : 0, qr = hb ? 2 : 1;      // left (gate) / right (up) quadrant
#pragma unroll
            for (int blk = 0; blk < (SWI ? 1 : 2); ++blk) {  // plain: x blocks {0,1} then {2,3}
                // ---- accumulators -> block
#pragma unroll
                for (int mi = 0; mi < (SWI ? 4 : 2); ++mi) {
                    const int m = SWI ? mi : blk * 2 + mi;
                    const int rr = mi * 16 + frow;           // row of the block
#pragma unroll
                    for (int sd = 0; sd < (SWI ? 1 : 2); ++sd)
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            float r[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float a0 = SWI ? acc[ql][n][m][e] : acc[sd ? qr : ql][n][m][e];
                                float a1 = SWI ? acc[qr][n][m][e] : 0.f;
                                if constexpr (RS) { a0 *= rs[hb][m]; a1 *= rs[hb][m]; }
                                if (SWI) {
                                    const float g = a0 + bias_v[0][n][e];
                                    const float up = a1 + bias_v[1][n][e];
                                    r[e] = g * __builtin_amdgcn_rcpf(1.f + __expf(-g)) * up;
                                } else {
                                    r[e] = a0 + bias_v[sd][n][e];
                                }
                            }
                            u32x2_t o;
                            o[0] = pack2<CODE>(r[0], r[1]);
                            o[1] = pack2<CODE>(r[2], r[3]);
                            const int cb = (sd * 32 + n * 16 + 4 * fq) * 2;      // byte column
                            const int chunk = (cb >> 4) ^ (SWI ? ((rr >> 1) & 3) : (rr & 7));
                            *reinterpret_cast<u32x2_t*>(ep + rr * ROWB + chunk * 16 + (cb & 15)) = o;
                        }
                }
                // ---- block -> memory, RPI whole rows per instruction
#pragma unroll
                for (int i = 0; i < 4096 / ROWB / RPI; ++i) {
                    const int rr = i * RPI + orow;
                    const int chunk = ochunk ^ (SWI ? ((rr >> 1) & 3) : (rr & 7));
                    u32x4_t v = *reinterpret_cast<const u32x4_t*>(ep + rr * ROWB + chunk * 16);
                    const int row = m0 + hb * 128 + wr * 64 + (SWI ? 0 : blk * 32) + rr;
                    // rows >= M lie beyond out_bytes: the range check drops them
                    const uint32_t voff = (uint32_t)row * (uint32_t)(a.o_stride * 2)
                                          + (uint32_t)(col_base * 2 + ochunk * 16);
                    if constexpr (RES) {
                        // the projection was rounded to the storage type (as a stand-alone GEMM
                        // rounds its output); + residual, rounded again: the new residual
                        float p[8], q[8];
                        unpack8<CODE>(v, p);
                        unpack8<CODE>(resv[hb][blk][i], q);
#pragma unroll
                        for (int e = 0; e < 8; ++e) p[e] += q[e];
                        v = pack8<CODE>(p);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, (int)voff, 0, 0);
                }
            }
        }
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[q][n][m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if constexpr (RS) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) ssq[hb][0] = ssq[hb][1] = 0.f;
        }
    };

    // ---- this workgroup's tiles
    int id = tile_id(0);
    if (id < 0) return;                      // uniform over the workgroup
    int tm, tn;
    tile_pos(id, tm, tn);
    tile_src(tm, tn, cur);
    const int nk = a.ktiles;                 // >= 4

    // prologue: K-tile 0 complete, A0 / B0 of K-tile 1 on their way
    stage(0, I0{}, false, 0); stage(0, I1{}, false, 0); stage(0, I2{}, false, 0); stage(0, I3{}, false, 0);
    stage(kBuf, I0{}, false, 1); stage(kBuf, I1{}, false, 1);
    big_wait_vmcnt<0>();
    big_barrier();
    if (wr == 1) big_barrier();              // the second wave row runs one barrier behind

    int g = 0;                               // K-tiles done by this workgroup (buffer parity)
    for (int rd = 0;; ++rd) {
        const int nid = tile_id(rd + 1);
        int ntm = tm, ntn = tn;
        if (nid >= 0) tile_pos(nid, ntm, ntn);
        tile_src(ntm, ntn, nxt);             // the last tile requests its own first K-tiles again
        zero_acc();
        ktile(I1{}, g, 0, nk);
        ++g;
        for (int u = 1; u < nk; ++u, ++g) ktile(I0{}, g, u, nk);
        // every request so far (up to A0 / B0 of the next tile's second K-tile) lands before
        // the stores are issued: loads and stores do not complete in a common order, so a
        // counted wait behind the stores would have to wait for all of them
        if constexpr (RES) pre_epilogue(tm, tn);
        exchange_sums();
        big_wait_vmcnt<0>();
        epilogue(tm, tn);
        if (nid < 0) break;
        cur = nxt; tm = ntm; tn = ntn;
    }
    if (wr == 0) big_barrier();
    big_wait_vmcnt<0>();
    if (MVL_BIG_STAMP && bid == 0 && (wv & 3) == 0 && lane < 16) {
        uint64_t v = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) v = lane == i ? stamps[i] : v;
        reinterpret_cast<uint64_t*>(a.out)[wv * 4 + lane] = v;       // wave 0: slots 0..15, wave 4: 16..31
    }
}

template <int CODE, bool SWI, bool RES = false, bool RS = false>
int big_launch(BigArgs a, hipStream_t st) {
    auto k = big_gemm_kernel<CODE, SWI, RES, RS>;
    static const hipError_t once = hipFuncSetAttribute(
        (const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kBigLds);
    if (once != hipSuccess) return (int)once;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -2;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = a.tiles < cus ? a.tiles : cus;
    k<<<grid, kBigThreads, kBigLds, st>>>(a);
    return MVL_CHECK_LAUNCH();
}

}  // namespace

// out[M, N_out] = x . W^T (+ bias)  [swiglu: W = [gate; up], N_out = N / 2].
// K a multiple of 64 and >= 256; N a multiple of 256 (swiglu: N / 2 a multiple of 128); x rows
// and out rows 16-byte aligned; x, W and out below 2 GiB each (32-bit buffer offsets).
// row_scale: every output row is scaled by rsqrt(sum(x[row]^2) / K + eps) before bias / SwiGLU.
// add_residual (plain epilogue only, not together with row_scale): `out` holds the residual
// stream on entry, out = round(round(x . W^T + bias) + out) in place.
extern "C" int mvl_big_gemm(const void* x, long x_stride, const void* w, const void* bias,
                            void* out, long o_stride, int M, int N, int K, int swiglu, int dtype,
                            int row_scale, float eps, int add_residual, void* stream) {
    if (M == 0) return 0;
    if (add_residual && (swiglu || row_scale)) return -1;
    if (M < 0 || K % kBigBK != 0 || K < 4 * kBigBK || (dtype != 0 && dtype != 1)) return -1;
    if (swiglu ? (N % 2 != 0 || (N / 2) % 128 != 0) : (N % 256 != 0)) return -1;
    if (x_stride % 8 != 0 || o_stride % 8 != 0) return -1;
    const int n_out = swiglu ? N / 2 : N;
    const double out_bytes = ((double)(M - 1) * (double)o_stride + n_out) * 2.0;
    if ((double)M * (double)x_stride * 2.0 >= 2147483648.0 || (double)N * (double)K * 2.0 >= 2147483648.0
        || out_bytes >= 2147483648.0)
        return -1;
    BigArgs a;
    a.x = (const uint16_t*)x; a.x_stride = x_stride; a.w = (const uint16_t*)w;
    a.bias = (const uint16_t*)bias; a.out = (uint16_t*)out; a.o_stride = o_stride;
    a.out_bytes = (unsigned)out_bytes;
    a.x_bytes = (unsigned)(((double)(M - 1) * (double)x_stride + K) * 2.0);
    a.w_bytes = (unsigned)((double)N * (double)K * 2.0);
    a.M = M; a.N = N; a.K = K; a.up_off = N / 2;
    a.tiles_m = (M + 255) / 256;
    a.tiles_n = swiglu ? (N / 2) / 128 : N / 256;
    a.tiles = a.tiles_m * a.tiles_n;
    a.ktiles = K / kBigBK;
    static const char* probe_env = getenv("MVL_BIG_PROBE");
    a.probe = probe_env ? atoi(probe_env) : 0;
    a.rs_eps = eps; a.rs_inv_k = 1.0f / (float)K;
    auto st = (hipStream_t)stream;
    if (add_residual)
        return dtype == 0 ? big_launch<0, false, true>(a, st) : big_launch<1, false, true>(a, st);
    if (row_scale) {
        if (dtype == 0) return swiglu ? big_launch<0, true, false, true>(a, st) : big_launch<0, false, false, true>(a, st);
        return swiglu ? big_launch<1, true, false, true>(a, st) : big_launch<1, false, false, true>(a, st);
    }
    if (dtype == 0) return swiglu ? big_launch<0, true>(a, st) : big_launch<0, false>(a, st);
    return swiglu ? big_launch<1, true>(a, st) : big_launch<1, false>(a, st);
}
