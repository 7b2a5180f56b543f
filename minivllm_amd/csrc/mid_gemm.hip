// Weight-streaming GEMM for decode batches of 17..256 rows:  out[M, N] = x[M, K] . W[N, K]^T.
//
// At these sizes a projection sits at the ridge of the roofline (M flops per weight byte): the
// weights must stream from HBM at full rate on all 256 CUs while every CU also re-reads its
// share of x from L2, and the MFMA work has to hide behind both.  What bounds a CU is how many
// bytes it can keep in flight, so the kernel is built around the loader:
//
//  * one workgroup (8 waves) owns a BM x BN output tile and a slice of K; BM covers every row
//    of the batch (64 / 128 / 256), so each weight byte is fetched from HBM exactly once, and
//    BN x split-K is chosen on the host so that the grid fills the chip;
//  * x and W tiles go global -> LDS by LDS-DMA in pieces of 8 rows x 128 bytes (whole cache
//    lines), into a ring of D stages of 64 k-elements; the loads of the next D - 1 k-steps
//    stay in flight across the one barrier per k-step (counted vmcnt, raw s_barrier);
//  * the LDS image is [row][8 chunks of 16 B] with chunk' = chunk ^ (row & 7), applied on the
//    SOURCE address of the DMA (the destination of a DMA is lane-linear); both operands are
//    read back as MFMA 16x16x32 fragments with conflict-free ds_read_b128;
//  * the product is computed transposed (A operand = W rows, B operand = x rows): a lane then
//    holds 4 consecutive output columns of one row, i.e. 8-byte bf16 / 16-byte fp32 stores;
//  * epilogues: plain (+ bias), SwiGLU (a wave's W rows are half gate, half the matching up
//    columns; stores silu(gate) * up), or fp32 partial sums per K slice for the reduce kernel
//    below (which applies bias / SwiGLU).
//
// Call sites replaced: the reference's F.linear in every projection
// (reference: src/myvllm/layers/linear.py:76,109,217).
#include "common.h"

using namespace mvl;

namespace {

#ifndef MVL_MID_ABLATE
#define MVL_MID_ABLATE 0    // measurement builds only: 1 no MFMA, 2 no DMA in the loop,
#endif                      // 3 DMA only, 4 x DMA only, 5 W DMA only
#ifndef MVL_MID_ROT_WINDOW
#define MVL_MID_ROT_WINDOW 8
#endif
constexpr int kBK = 64;                 // k-elements per ring stage (128 bytes per row)
constexpr int kThreads = 512;

struct MidArgs {
    const uint16_t* x; long x_stride;
    const uint16_t* w;                  // [N, K] row-major
    const uint16_t* bias;               // may be null
    uint16_t* out; long o_stride;       // EPI 0/1
    float* ws;                          // EPI 2: [splits][M][N]
    int M, N, K;
    int up_off;                         // SwiGLU: row offset of the up projection (= N / 2)
    int ksteps_per_split, ksteps;       // in units of kBK
};

// Issue order handed to the instruction scheduler for one stage: R fragment reads of the
// second k-half spread evenly between the Q MFMAs of the first.
template <int I, int R, int Q>
struct Interleave {
    static __device__ __forceinline__ void run() {
        constexpr int n_mfma = (Q * (I + 1)) / R - (Q * I) / R;
        if (n_mfma > 0) __builtin_amdgcn_sched_group_barrier(0x008, n_mfma, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        Interleave<I + 1, R, Q>::run();
    }
};
template <int R, int Q>
struct Interleave<R, R, Q> {
    static __device__ __forceinline__ void run() {}
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// EPI: 0 plain, 1 SwiGLU, 2 fp32 partials (SWI says how the W tile is assembled).
// WM waves along M (x rows) x 8 / WM along N (W rows); D: stages of the ring.
template <int CODE, int BM, int BN, int WM, int D, int EPI, bool SWI, bool NT>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mid_gemm_kernel(MidArgs a) {
    using DT = DType<CODE>;
    using vec8 = typename DT::vec8;
    constexpr int WN = 8 / WM;
    constexpr int TM = BM / WM, TN = BN / WN;          // wave tile: TM x-rows x TN W-rows
    constexpr int MT = TM / 16, NTL = TN / 16;
    constexpr int XPIECES = BM / 8, NPIECES = (BM + BN) / 8;   // 1 KiB DMA pieces per stage
    constexpr int PFULL = NPIECES / 8, PREM = NPIECES % 8;     // wave w moves pieces w, w + 8, ...
    constexpr int PMAX = PFULL + (PREM ? 1 : 0);
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int A = D - 1;                                   // stages requested ahead
    static_assert(TM % 16 == 0 && TN % 16 == 0 && BM % 64 == 0 && BN % 8 == 0, "tile shape");
    static_assert(!SWI || (TN / 2) % 16 == 0, "SwiGLU needs 16-column gate and up halves per wave");
    static_assert(D >= 2 && D * STAGE <= 160 * 1024, "LDS budget");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wv / WN, wn = wv % WN;
    const bool extra = wv < PREM;                      // this wave moves PFULL + 1 pieces
    const int n_tile = blockIdx.x, split = blockIdx.y;
    const int ks0 = split * a.ksteps_per_split;
    const int nsteps = min(a.ksteps_per_split, a.ksteps - ks0);

    // ---- DMA sources: a piece = 8 rows x 128 B; lane l moves chunk (l & 7) of the image row
    // (l >> 3), which holds global chunk (l & 7) ^ (l >> 3).  Pieces 0 .. BM/8-1 are x rows,
    // the rest W rows; BM/8 is a multiple of 8, so round i is x for every wave or W for every
    // wave.
    const int prow = lane >> 3;
    const int pchunk = (lane & 7) ^ prow;
    const uint16_t* src[PMAX];
#pragma unroll
    for (int i = 0; i < PMAX; ++i) {
        const int p = i * 8 + wv;
        if (i * 8 < XPIECES) {
            const int row = min(p * 8 + prow, a.M - 1);               // rows >= M: duplicates
            src[i] = a.x + (long)row * a.x_stride + (long)ks0 * kBK + pchunk * 8;
        } else {
            const int j = min(p - XPIECES, BN / 8 - 1) * 8 + prow;    // row of the W tile image
            long grow;
            if (SWI) {
                // per wave column: TN / 2 gate columns, then the matching TN / 2 up columns
                const int wn2 = j / TN, jj = j % TN;
                const int col = n_tile * (BN / 2) + wn2 * (TN / 2) + (jj % (TN / 2));
                grow = (jj >= TN / 2 ? a.up_off : 0) + col;
            } else {
                grow = (long)n_tile * BN + j;
            }
            src[i] = a.w + grow * a.K + (long)ks0 * kBK + pchunk * 8;
        }
    }
    // Workgroups that run at the same time start at slightly different k offsets (a window of a
    // few steps, then wrap around): with a power-of-two row pitch every workgroup otherwise
    // asks the same memory channels for its W rows at any moment (W alone streams at 4.5 TB/s
    // in lockstep, 6.0 TB/s staggered).  The window stays small because x is shared: workgroups
    // far apart in k no longer find each other's x lines in L2 once x exceeds it (K = 12288).
#ifdef MVL_MID_NO_ROTATE
    int kpos = 0;
#else
    int kpos = (int)((blockIdx.x * 37u + blockIdx.y * 11u) % (unsigned)min(nsteps, MVL_MID_ROT_WINDOW));
#endif
#pragma unroll
    for (int i = 0; i < PMAX; ++i) src[i] += (long)kpos * kBK;
    auto issue_stage = [&](int slot) {
        char* base = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            if (i == PFULL && !extra) break;                          // wave-uniform
            const bool is_w = i * 8 >= XPIECES;
            if ((MVL_MID_ABLATE == 4 && is_w) || (MVL_MID_ABLATE == 5 && !is_w)) continue;
            char* dst = base + (i * 8 + wv) * 1024;
            if (NT && is_w)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)src[i],
                    (__attribute__((address_space(3))) void*)dst, 16, 0, 2);
            else
                glds16(src[i], dst);
            src[i] += kBK;
        }
        // the K slice is walked from a per-workgroup starting step and wraps around
        if (++kpos == nsteps) {
            kpos = 0;
#pragma unroll
            for (int i = 0; i < PMAX; ++i) src[i] -= (long)nsteps * kBK;
        }
    };

    // ---- fragment reads: rows (lane & 15) of a 16-row block, chunk 4 ks + (lane >> 4)
    const int frow = lane & 15, fq = lane >> 4;
    const int lane_off0 = frow * 128 + ((fq ^ (frow & 7)) << 4);      // ks = 1: ^ 64
    const int x_base = wm * TM * 128, w_base = BM * 128 + wn * TN * 128;

    f32x4_t acc[NTL][MT];
#pragma unroll
    for (int n = 0; n < NTL; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto compute_stage = [&](int slot) {
        if (MVL_MID_ABLATE >= 3) return;
        const char* xb = smem + slot * STAGE + x_base;
        const char* wb = smem + slot * STAGE + w_base;
        // every fragment of the stage is requested before the first MFMA: the reads of the
        // second k-half return under the MFMAs of the first
        vec8 xf[2][MT], wf[2][NTL];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int lo = lane_off0 ^ (ks << 6);
#pragma unroll
            for (int n = 0; n < NTL; ++n)
                wf[ks][n] = __builtin_bit_cast(vec8, *reinterpret_cast<const u32x4_t*>(wb + n * 2048 + lo));
#pragma unroll
            for (int m = 0; m < MT; ++m)
                xf[ks][m] = __builtin_bit_cast(vec8, *reinterpret_cast<const u32x4_t*>(xb + m * 2048 + lo));
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int n = 0; n < NTL; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    if (MVL_MID_ABLATE == 1) {
                        asm volatile("" ::"v"(wf[ks][n]), "v"(xf[ks][m]));
                        continue;
                    }
                    acc[n][m] = DT::mfma16(wf[ks][n], xf[ks][m], acc[n][m]);
                }
        // without this the scheduler serialises read -> wait -> MFMA to save registers
        constexpr int R = MT + NTL, Q = MT * NTL;
        __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
        Interleave<0, R, Q>::run();
        __builtin_amdgcn_sched_group_barrier(0x008, Q, 0);
    };

    // ---- pipeline.  Step t computes on stage t; its loads were requested A steps earlier and
    // stage t + A is requested right after the barrier of step t, which also tells this wave
    // that every wave is done with the slot of step t - 1.
#pragma unroll
    for (int s = 0; s < A; ++s)
        if (s < nsteps) issue_stage(s);
    int sc = 0, si = A;                              // ring slots: computed / requested in step t
    for (int t = 0; t < nsteps; ++t) {
        // stage t has landed once all but the requests of the A - 1 younger stages are done
        // (the count only holds while each of them is a real stage; at the end just drain)
        if (t + A <= nsteps) {
            if (extra) wait_vmcnt<(PFULL + 1) * (A - 1)>(); else wait_vmcnt<PFULL * (A - 1)>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (MVL_MID_ABLATE != 2 && t + A < nsteps) issue_stage(si);
        compute_stage(sc);
        if (++sc == D) sc = 0;
        if (++si == D) si = 0;
    }

    // ---- epilogue: acc[n][m][e] = out[row 16 m + frow][col 16 n + 4 fq + e]
    // column of accumulator block n (first of the lane's 4), in the full [N] numbering
    int col_of[NTL];
#pragma unroll
    for (int n = 0; n < NTL; ++n) {
        if (SWI) {
            constexpr int half = NTL / 2;
            col_of[n] = (n >= half ? a.up_off : 0) + n_tile * (BN / 2) + wn * (TN / 2)
                        + (n % half) * 16 + 4 * fq;
        } else {
            col_of[n] = n_tile * BN + wn * TN + n * 16 + 4 * fq;
        }
    }
    if (EPI != 2 && a.bias != nullptr) {                       // uniform branch, 8-byte loads
#pragma unroll
        for (int n = 0; n < NTL; ++n) {
            const u32x2_t b = *reinterpret_cast<const u32x2_t*>(a.bias + col_of[n]);
            float bv[4];
            unpack2<CODE>(b[0], bv[0], bv[1]);
            unpack2<CODE>(b[1], bv[2], bv[3]);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[n][m][e] += bv[e];
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = wm * TM + m * 16 + frow;
        if (row >= a.M) continue;
        if (EPI == 2) {
            float* wp = a.ws + ((long)split * a.M + row) * a.N;
#pragma unroll
            for (int n = 0; n < NTL; ++n) *reinterpret_cast<f32x4_t*>(wp + col_of[n]) = acc[n][m];
        } else if (EPI == 1) {
            constexpr int half = NTL / 2;
            uint16_t* op = a.out + (long)row * a.o_stride;
#pragma unroll
            for (int n = 0; n < half; ++n) {
                float r[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = acc[n][m][e], u = acc[n + half][m][e];
                    r[e] = g / (1.f + __expf(-g)) * u;
                }
                u32x2_t o;
                o[0] = pack2<CODE>(r[0], r[1]);
                o[1] = pack2<CODE>(r[2], r[3]);
                *reinterpret_cast<u32x2_t*>(op + col_of[n]) = o;
            }
        } else {
            uint16_t* op = a.out + (long)row * a.o_stride;
#pragma unroll
            for (int n = 0; n < NTL; ++n) {
                u32x2_t o;
                o[0] = pack2<CODE>(acc[n][m][0], acc[n][m][1]);
                o[1] = pack2<CODE>(acc[n][m][2], acc[n][m][3]);
                *reinterpret_cast<u32x2_t*>(op + col_of[n]) = o;
            }
        }
    }
}

// Sum of the K-slice partials (+ bias, SwiGLU): ws [S][M][N] fp32 -> out [M, N or N/2].
// One thread per 4 output columns.
template <int CODE, bool SWI>
__global__ __launch_bounds__(256) void mid_reduce_kernel(const float* __restrict__ ws,
                                                         const uint16_t* __restrict__ bias,
                                                         uint16_t* __restrict__ out, long o_stride,
                                                         int M, int N, int S, int n_out) {
    using DT = DType<CODE>;
    const int per_row = n_out / 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)M * per_row) return;
    const int row = (int)(idx / per_row), col = (int)(idx % per_row) * 4;
    const long slab = (long)M * N;
    const float* p = ws + (long)row * N + col;
    f32x4_t g = *reinterpret_cast<const f32x4_t*>(p), u = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (SWI) u = *reinterpret_cast<const f32x4_t*>(p + n_out);
    for (int s = 1; s < S; ++s) {
        g += *reinterpret_cast<const f32x4_t*>(p + s * slab);
        if (SWI) u += *reinterpret_cast<const f32x4_t*>(p + s * slab + n_out);
    }
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float gv = g[e];
        if (bias != nullptr) gv += DT::to_float(bias[col + e]);
        if (SWI) {
            float uv = u[e];
            if (bias != nullptr) uv += DT::to_float(bias[n_out + col + e]);
            r[e] = gv / (1.f + __expf(-gv)) * uv;
        } else {
            r[e] = gv;
        }
    }
    u32x2_t o;
    o[0] = pack2<CODE>(r[0], r[1]);
    o[1] = pack2<CODE>(r[2], r[3]);
    *reinterpret_cast<u32x2_t*>(out + (long)row * o_stride + col) = o;
}

// Consumer of the K-slice partials when the projection feeds (residual add +) RMSNorm:
// r = sum_s ws[s][row] (+ bias) + residual[row], rounded to the storage dtype and written back
// to `residual`; out[row] = r * rsqrt(mean(r^2) + eps) * weight.  One workgroup per row, 8
// columns per thread and pass; the row stays in registers between the two phases.  Replaces
// the partial-sum kernel above and the add + norm kernel that would follow it.
template <int CODE, int MAXC>
__global__ void splitk_add_rms_norm_kernel(const float* __restrict__ ws,
                                           const uint16_t* __restrict__ bias,
                                           uint16_t* residual,
                                           const uint16_t* __restrict__ weight,
                                           uint16_t* __restrict__ out, int M, int N, int S,
                                           long r_stride, long o_stride, float eps) {
    __shared__ float scratch[16];
    const int row = blockIdx.x;
    const int nchunk = N >> 3;
    const long slab = (long)M * N;
    const float* p = ws + (long)row * N;
    u32x4_t* rr = reinterpret_cast<u32x4_t*>(residual + (long)row * r_stride);
    const u32x4_t* wr = reinterpret_cast<const u32x4_t*>(weight);
    u32x4_t wv[MAXC];
    float v[MAXC][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = threadIdx.x + i * blockDim.x;
        if (c < nchunk) {
            wv[i] = wr[c];
            const u32x4_t rv = rr[c];
            f32x4_t lo = *reinterpret_cast<const f32x4_t*>(p + c * 8);
            f32x4_t hi = *reinterpret_cast<const f32x4_t*>(p + c * 8 + 4);
#pragma unroll 4
            for (int s = 1; s < S; ++s) {
                lo += *reinterpret_cast<const f32x4_t*>(p + s * slab + c * 8);
                hi += *reinterpret_cast<const f32x4_t*>(p + s * slab + c * 8 + 4);
            }
            float r[8];
            unpack8<CODE>(rv, r);
            if (bias != nullptr) {
                float b[8];
                unpack8<CODE>(reinterpret_cast<const u32x4_t*>(bias)[c], b);
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] += b[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = (j < 4 ? lo[j] : hi[j - 4]) + r[j];
                v[i][j] = DType<CODE>::to_float(DType<CODE>::from_float(t));
                ss += v[i][j] * v[i][j];
            }
            rr[c] = pack8<CODE>(v[i]);
        } else {
            wv[i] = u32x4_t{0, 0, 0, 0};
        }
    }
    ss = block_sum(ss, scratch);
    const float rstd = rsqrtf(ss / (float)N + eps);
    u32x4_t* orow = reinterpret_cast<u32x4_t*>(out + (long)row * o_stride);
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

template <int CODE, int BM, int BN, int WM, int D, bool SWI, bool NT>
int launch_tile(MidArgs a, int splits, hipStream_t st) {
    const int tiles = a.N / BN;
    const size_t lds = (size_t)D * (BM + BN) * 128;
    dim3 grid(tiles, splits);
    if (splits > 1) {
        auto k = mid_gemm_kernel<CODE, BM, BN, WM, D, 2, SWI, NT>;
        static const hipError_t once = hipFuncSetAttribute(
            (const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (once != hipSuccess) return (int)once;
        k<<<grid, kThreads, lds, st>>>(a);
    } else {
        auto k = mid_gemm_kernel<CODE, BM, BN, WM, D, SWI ? 1 : 0, SWI, NT>;
        static const hipError_t once = hipFuncSetAttribute(
            (const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (once != hipSuccess) return (int)once;
        k<<<grid, kThreads, lds, st>>>(a);
    }
    return MVL_CHECK_LAUNCH();
}

// (row tile, column tile) -> wave layout and ring depth; LDS = D (BM + BN) 128 bytes
template <int CODE, bool SWI, bool NT>
int launch_shape(MidArgs a, int bm, int bn, int splits, hipStream_t st) {
#define MVL_MID(BM_, BN_, WM_, D_) \
    if (bm == BM_ && bn == BN_) return launch_tile<CODE, BM_, BN_, WM_, D_, SWI, NT>(a, splits, st)
    MVL_MID(256, 128, 4, 3);            // 144 KB
    MVL_MID(256, 96, 8, 3);             // 132 KB
    MVL_MID(256, 64, 4, 3);             // 120 KB
    MVL_MID(128, 128, 4, 4);            // 128 KB
    MVL_MID(128, 64, 4, 4);             //  96 KB
    MVL_MID(64, 128, 4, 4);             //  96 KB
    MVL_MID(64, 64, 4, 5);              //  80 KB
#undef MVL_MID
    return -1;
}

}  // namespace

// out[M, N_out] = x . W^T (+ bias) [SwiGLU: W = [gate; up], N_out = N / 2].
// bm in {64, 128, 256} >= M; bn in {64, 96 (bm 256 only), 128} with N % bn == 0; splits >= 1 K slices (workspace
// `ws` of splits * M * N floats when splits > 1); nt: non-temporal weight loads.
extern "C" int mvl_mid_gemm(const void* x, long x_stride, const void* w, const void* bias,
                            void* out, long o_stride, void* ws, int M, int N, int K, int swiglu,
                            int bm, int bn, int splits, int nt, int dtype, void* stream) {
    if (M == 0) return 0;
    if (M > bm || K % kBK != 0 || N % bn != 0 || splits < 1 || (dtype != 0 && dtype != 1)) return -1;
    if (swiglu && (N % 2 != 0 || (N / 2) % (bn / 2) != 0)) return -1;
    if (splits > 1 && ws == nullptr) return -1;
    MidArgs a;
    a.x = (const uint16_t*)x; a.x_stride = x_stride; a.w = (const uint16_t*)w;
    a.bias = (const uint16_t*)bias; a.out = (uint16_t*)out; a.o_stride = o_stride;
    a.ws = (float*)ws; a.M = M; a.N = N; a.K = K; a.up_off = N / 2;
    a.ksteps = K / kBK;
    a.ksteps_per_split = (a.ksteps + splits - 1) / splits;
    splits = (a.ksteps + a.ksteps_per_split - 1) / a.ksteps_per_split;     // no empty slices
    if (out == nullptr && (splits == 1 || swiglu)) return -1;              // partials only: needs a K split
    auto st = (hipStream_t)stream;
    int rc;
#define MVL_DISPATCH(CODE)                                                                     \
    rc = swiglu ? (nt ? launch_shape<CODE, true, true>(a, bm, bn, splits, st)                  \
                      : launch_shape<CODE, true, false>(a, bm, bn, splits, st))                \
                : (nt ? launch_shape<CODE, false, true>(a, bm, bn, splits, st)                 \
                      : launch_shape<CODE, false, false>(a, bm, bn, splits, st))
    if (dtype == 0) { MVL_DISPATCH(0); } else { MVL_DISPATCH(1); }
#undef MVL_DISPATCH
    if (rc != 0 || splits == 1 || out == nullptr) return rc;      // out == nullptr: partials only
    const int n_out = swiglu ? N / 2 : N;
    const long threads = (long)M * (n_out / 4);
    const int blocks = (int)((threads + 255) / 256);
    auto B = (const uint16_t*)bias; auto O = (uint16_t*)out;
    if (dtype == 0) {
        if (swiglu) mid_reduce_kernel<0, true><<<blocks, 256, 0, st>>>(a.ws, B, O, o_stride, M, N, splits, n_out);
        else mid_reduce_kernel<0, false><<<blocks, 256, 0, st>>>(a.ws, B, O, o_stride, M, N, splits, n_out);
    } else {
        if (swiglu) mid_reduce_kernel<1, true><<<blocks, 256, 0, st>>>(a.ws, B, O, o_stride, M, N, splits, n_out);
        else mid_reduce_kernel<1, false><<<blocks, 256, 0, st>>>(a.ws, B, O, o_stride, M, N, splits, n_out);
    }
    return MVL_CHECK_LAUNCH();
}

// Number of K slices mvl_mid_gemm really writes for a requested `splits` (no empty slices).
extern "C" int mvl_mid_effective_splits(int K, int splits) {
    if (K <= 0 || K % kBK != 0 || splits < 1) return -1;
    const int ksteps = K / kBK, per = (ksteps + splits - 1) / splits;
    return (ksteps + per - 1) / per;
}

// ws [S][M][N] fp32 (K-slice partials from mvl_mid_gemm with out == nullptr) -> out [M, N]
extern "C" int mvl_mid_reduce(const void* ws, const void* bias, void* out, long o_stride, int M,
                              int N, int S, int dtype, void* stream) {
    if (M == 0) return 0;
    if (N % 4 != 0 || S < 1 || (dtype != 0 && dtype != 1)) return -1;
    const long threads = (long)M * (N / 4);
    const int blocks = (int)((threads + 255) / 256);
    auto st = (hipStream_t)stream;
    auto W = (const float*)ws; auto B = (const uint16_t*)bias; auto O = (uint16_t*)out;
    if (dtype == 0) mid_reduce_kernel<0, false><<<blocks, 256, 0, st>>>(W, B, O, o_stride, M, N, S, N);
    else mid_reduce_kernel<1, false><<<blocks, 256, 0, st>>>(W, B, O, o_stride, M, N, S, N);
    return MVL_CHECK_LAUNCH();
}

// residual <- sum_s ws[s] (+ bias) + residual; out = rms_norm(residual) * weight.  N % 8 == 0,
// N <= 32768; residual and out rows 16-byte aligned.
extern "C" int mvl_splitk_add_rms_norm(const void* ws, const void* bias, void* residual,
                                       const void* weight, void* out, int M, int N, int S,
                                       long r_stride, long o_stride, float eps, int dtype,
                                       void* stream) {
    if (M == 0) return 0;
    if (N % 8 != 0 || N > 8 * 1024 * 4 || S < 1 || (dtype != 0 && dtype != 1)
        || r_stride % 8 != 0 || o_stride % 8 != 0) return -1;
    const int nchunk = N / 8;
    int threads = 64;
    while (threads < nchunk && threads < 1024) threads <<= 1;
    const int per = (nchunk + threads - 1) / threads;
    auto st = (hipStream_t)stream;
    auto W = (const float*)ws; auto B = (const uint16_t*)bias; auto R = (uint16_t*)residual;
    auto G = (const uint16_t*)weight; auto O = (uint16_t*)out;
#define MVL_SK(CODE, MAXC) \
    splitk_add_rms_norm_kernel<CODE, MAXC><<<M, threads, 0, st>>>(W, B, R, G, O, M, N, S, r_stride, o_stride, eps)
    if (dtype == 0) {
        if (per <= 1) MVL_SK(0, 1); else if (per <= 2) MVL_SK(0, 2); else MVL_SK(0, 4);
    } else {
        if (per <= 1) MVL_SK(1, 1); else if (per <= 2) MVL_SK(1, 2); else MVL_SK(1, 4);
    }
#undef MVL_SK
    return MVL_CHECK_LAUNCH();
}
