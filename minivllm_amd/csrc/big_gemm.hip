// Compute-bound GEMM for prefill-sized row counts:  out[M, N] = x[M, K] . W[N, K]^T, optionally
// with the SwiGLU epilogue  out[M, N/2] = silu(x . Wg^T) * (x . Wu^T)  for W = [Wg; Wu], which
// saves the pass of the activation kernel over the [M, N] intermediate (for Qwen3-8B at 16384
// tokens: 805 MB written and read back per layer).
//
// Shape of the kernel (gfx950, wave64):
//  * one workgroup of 8 waves owns a 256 x 256 output tile and walks K in 64-element tiles;
//    LDS holds two K-tiles, each as four half-tiles of 128 rows x 128 bytes:
//    A0 / A1 = x rows 0..127 / 128..255, B0 / B1 = W rows 0..127 / 128..255 of the tile
//    (SwiGLU: B0 = 128 gate columns, B1 = the matching 128 up columns);
//  * wave (wr, wc), wr = 0..1, wc = 0..3, owns x rows {64 wr .. +63} of A0 ("top") and of A1
//    ("bottom") and W rows {32 wc .. +31} of B0 ("left") and of B1 ("right"): four 64 x 32
//    quadrants, 128 accumulator registers.  A K-tile is four phases, one quadrant each, in the
//    order (top,left) (top,right) (bottom,right) (bottom,left), so the fragment reads are
//    12 / 4 / 8 / 0 ds_read_b128 and the half-tiles are first needed in the order A0 B0 B1 A1;
//  * every phase requests one half-tile (2 LDS-DMA instructions per wave) and waits with a
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
//  * workgroup ids are remapped so that the ids sharing an XCD (id mod 8) cover a compact block
//    of tiles (panels of 8 row tiles walked column by column): each XCD's L2 then serves an x
//    panel to 4 column tiles and a W panel to 8 row tiles at a time.
//
// Call sites: the MLP's gate_up projection on prefill steps (the reference's
// F.linear + SiluAndMul, src/myvllm/layers/linear.py:109, src/myvllm/layers/activation.py:8-16),
// and any projection for which the start-up measurement prefers it to the library.
#include "common.h"
#include <type_traits>

using namespace mvl;

namespace {

constexpr int kBigThreads = 512;
constexpr int kHalf = 128 * 128;            // bytes of a half-tile
constexpr int kBuf = 4 * kHalf;             // one K-tile: A0 B0 B1 A1
constexpr int kBigBK = 64;

struct BigArgs {
    const uint16_t* x; long x_stride;       // [M, K], row stride in elements
    const uint16_t* w;                      // [N, K] row-major
    const uint16_t* bias;                   // may be null
    uint16_t* out; long o_stride;
    int M, N, K;
    int up_off;                             // SwiGLU: row of W where the up projection starts
    int tiles_m, tiles_n, ktiles;
};

template <int N>
__device__ __forceinline__ void big_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void big_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int CODE, bool SWI>
__global__ __launch_bounds__(kBigThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void big_gemm_kernel(BigArgs a) {
    using DT = DType<CODE>;
    using vec8 = typename DT::vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wv >> 2, wc = wv & 3;

    // ---- tile of this workgroup
    int tm, tn;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        constexpr int PM = 8;
        const int per_panel = PM * a.tiles_n;
        const int panel = id / per_panel;
        const int rem = id - panel * per_panel;
        const int rows = min(PM, a.tiles_m - panel * PM);
        tn = rem / rows;
        tm = panel * PM + rem - tn * rows;
    }
    const int m0 = tm * 256;
    const int n0 = SWI ? tn * 128 : tn * 256;          // first W row of half B0

    // ---- DMA sources.  A piece is 8 rows x 128 bytes; lane l moves chunk (l & 7) of image row
    // (l >> 3), which holds global chunk (l & 7) ^ (l >> 3).  Wave w moves pieces 2w, 2w + 1 of
    // each half-tile.  Offsets are 32-bit byte offsets from a uniform base that walks along K.
    const int prow = lane >> 3;
    const int pchunk = (lane & 7) ^ prow;
    uint32_t off_a[2][2], off_b[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = h * 128 + (wv * 2 + i) * 8 + prow;
            const int xr = min(m0 + r, a.M - 1);                        // rows >= M: duplicates
            off_a[h][i] = (uint32_t)xr * (uint32_t)(a.x_stride * 2) + pchunk * 16;
            int wrow = SWI ? (h ? a.up_off : 0) + n0 + (r & 127) : n0 + r;
            wrow = min(wrow, a.N - 1);
            off_b[h][i] = (uint32_t)wrow * (uint32_t)(a.K * 2) + pchunk * 16;
        }
    const char* xk = reinterpret_cast<const char*>(a.x);
    const char* wk = reinterpret_cast<const char*>(a.w);

    // request half-tile J (0 A0, 1 B0, 2 B1, 3 A1) of the K-tile at byte offset kb into the
    // buffer at LDS byte offset bofs (wave-uniform)
    auto stage = [&](int bofs, auto j_c, long kb) {
        constexpr int J = decltype(j_c)::value;
        constexpr bool is_a = (J == 0 || J == 3);
        constexpr int h = (J == 2 || J == 3) ? 1 : 0;
        char* dst = smem + bofs + J * kHalf + wv * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* src = (is_a ? xk : wk) + kb + (is_a ? off_a[h][i] : off_b[h][i]);
            glds16(src, dst + i * 1024);
        }
    };

    // ---- fragment reads: rows (lane & 15) of a 16-row block, k-chunk 4 ks + (lane >> 4)
    const int frow = lane & 15, fq = lane >> 4;
    const int lane_off0 = frow * 128 + ((fq ^ (frow & 7)) << 4);
    const int a_off = lane_off0 + wr * (64 * 128);
    const int b_off = lane_off0 + wc * (32 * 128);

    f32x4_t acc[4][2][4];                    // [quadrant TL TR BR BL][W block][x block]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[q][n][m] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    vec8 af[2][4], bl[2][2], br[2][2];       // [k half][block]

    auto read_a = [&](int bofs, auto j_c) {
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
        constexpr int J = decltype(j_c)::value;
        const char* base = smem + bofs + J * kHalf;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                bf[ks][n] = __builtin_bit_cast(
                    vec8, *reinterpret_cast<const u32x4_t*>(base + n * 2048 + (b_off ^ (ks << 6))));
    };
    auto mfma_quadrant = [&](auto q_c, vec8 (&bf)[2][2]) {
        constexpr int Q = decltype(q_c)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[Q][n][m] = DT::mfma16(bf[ks][n], af[ks][m], acc[Q][n][m]);
        __builtin_amdgcn_s_setprio(0);
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // One K-tile (index u, in buffer u & 1).  MODE 0: tiles u + 1 and u + 2 exist; 1: u + 1 is
    // the last tile; 2: u is the last tile.  The vmcnt values count 2 instructions per half-tile
    // requested after the one that the next phase reads.
    auto ktile = [&](auto mode_c, int u) {
        constexpr int MODE = decltype(mode_c)::value;
        const int B = (u & 1) * kBuf, NB = kBuf - B;
        const long kb1 = (long)(u + 1) * (kBigBK * 2), kb2 = (long)(u + 2) * (kBigBK * 2);
        // phase 1: (top, left)
        read_b(B, I1{}, bl);
        read_a(B, I0{});
        if (MODE < 2) stage(NB, I2{}, kb1);
        if (MODE < 2) big_wait_vmcnt<8>(); else big_wait_vmcnt<2>();
        big_barrier();
        mfma_quadrant(I0{}, bl);
        big_barrier();
        // phase 2: (top, right)
        read_b(B, I2{}, br);
        if (MODE < 2) stage(NB, I3{}, kb1);
        if (MODE < 2) big_wait_vmcnt<8>(); else big_wait_vmcnt<0>();
        big_barrier();
        mfma_quadrant(I1{}, br);
        big_barrier();
        // phase 3: (bottom, right)
        read_a(B, I3{});
        if (MODE == 0) stage(B, I0{}, kb2);
        big_barrier();
        mfma_quadrant(I2{}, br);
        big_barrier();
        // phase 4: (bottom, left), no reads
        if (MODE == 0) stage(B, I1{}, kb2);
        if (MODE == 0) big_wait_vmcnt<8>(); else if (MODE == 1) big_wait_vmcnt<4>();
        big_barrier();
        mfma_quadrant(I3{}, bl);
        big_barrier();
    };

    // ---- prologue: K-tile 0 complete, A0 / B0 of K-tile 1
    stage(0, I0{}, 0); stage(0, I1{}, 0); stage(0, I2{}, 0); stage(0, I3{}, 0);
    stage(kBuf, I0{}, kBigBK * 2); stage(kBuf, I1{}, kBigBK * 2);
    big_wait_vmcnt<8>();
    big_barrier();
    if (wr == 1) big_barrier();              // the second wave row runs one barrier behind

    const int nk = a.ktiles;
    for (int u = 0; u < nk - 2; ++u) ktile(I0{}, u);
    ktile(I1{}, nk - 2);
    ktile(I2{}, nk - 1);
    if (wr == 0) big_barrier();

    // ---- epilogue: acc[q][n][m][e] = out[row of x block m][col 16 n + 4 fq + e of the quadrant]
    // quadrants: 0 (top, left) 1 (top, right) 2 (bottom, right) 3 (bottom, left)
#pragma unroll
    for (int q = 0; q < (SWI ? 2 : 4); ++q) {
        // SwiGLU pairs quadrant 0 with 1 (top rows) and 3 with 2 (bottom rows)
        const int qg = SWI ? (q == 0 ? 0 : 3) : q;          // gate / plain quadrant
        const int qu = SWI ? (q == 0 ? 1 : 2) : q;          // up quadrant
        const bool bottom = (qg == 2 || qg == 3);
        const bool right = (qg == 1 || qg == 2);
        const int row0 = m0 + (bottom ? 128 : 0) + wr * 64 + frow;
        const int col0 = SWI ? tn * 128 + wc * 32 + 4 * fq
                             : tn * 256 + (right ? 128 : 0) + wc * 32 + 4 * fq;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = col0 + n * 16;
            float bg[4] = {0.f, 0.f, 0.f, 0.f}, bu[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias != nullptr) {
                const u32x2_t b = *reinterpret_cast<const u32x2_t*>(a.bias + col);
                unpack2<CODE>(b[0], bg[0], bg[1]); unpack2<CODE>(b[1], bg[2], bg[3]);
                if (SWI) {
                    const u32x2_t c = *reinterpret_cast<const u32x2_t*>(a.bias + a.up_off + col);
                    unpack2<CODE>(c[0], bu[0], bu[1]); unpack2<CODE>(c[1], bu[2], bu[3]);
                }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int row = row0 + m * 16;
                if (row >= a.M) continue;
                float r[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = acc[qg][n][m][e] + bg[e];
                    if (SWI) {
                        const float up = acc[qu][n][m][e] + bu[e];
                        r[e] = g / (1.f + __expf(-g)) * up;
                    } else {
                        r[e] = g;
                    }
                }
                u32x2_t o;
                o[0] = pack2<CODE>(r[0], r[1]);
                o[1] = pack2<CODE>(r[2], r[3]);
                *reinterpret_cast<u32x2_t*>(a.out + (long)row * a.o_stride + col) = o;
            }
        }
    }
}

template <int CODE, bool SWI>
int big_launch(BigArgs a, hipStream_t st) {
    auto k = big_gemm_kernel<CODE, SWI>;
    static const hipError_t once = hipFuncSetAttribute(
        (const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kBuf);
    if (once != hipSuccess) return (int)once;
    k<<<a.tiles_m * a.tiles_n, kBigThreads, 2 * kBuf, st>>>(a);
    return MVL_CHECK_LAUNCH();
}

}  // namespace

// out[M, N_out] = x . W^T (+ bias)  [swiglu: W = [gate; up], N_out = N / 2].
// K a multiple of 64 and >= 128; N (N / 2 with swiglu) a multiple of 128... see the checks.
extern "C" int mvl_big_gemm(const void* x, long x_stride, const void* w, const void* bias,
                            void* out, long o_stride, int M, int N, int K, int swiglu, int dtype,
                            void* stream) {
    if (M == 0) return 0;
    if (M < 0 || K % kBigBK != 0 || K < 2 * kBigBK || (dtype != 0 && dtype != 1)) return -1;
    if (swiglu ? (N % 2 != 0 || (N / 2) % 128 != 0) : (N % 256 != 0)) return -1;
    if (x_stride % 8 != 0 || o_stride % 4 != 0) return -1;
    // 32-bit byte offsets inside the operands
    if ((double)M * (double)x_stride * 2.0 >= 4294967296.0 || (double)N * (double)K * 2.0 >= 4294967296.0)
        return -1;
    BigArgs a;
    a.x = (const uint16_t*)x; a.x_stride = x_stride; a.w = (const uint16_t*)w;
    a.bias = (const uint16_t*)bias; a.out = (uint16_t*)out; a.o_stride = o_stride;
    a.M = M; a.N = N; a.K = K; a.up_off = N / 2;
    a.tiles_m = (M + 255) / 256;
    a.tiles_n = swiglu ? (N / 2) / 128 : N / 256;
    a.ktiles = K / kBigBK;
    auto st = (hipStream_t)stream;
    if (dtype == 0) return swiglu ? big_launch<0, true>(a, st) : big_launch<0, false>(a, st);
    return swiglu ? big_launch<1, true>(a, st) : big_launch<1, false>(a, st);
}
