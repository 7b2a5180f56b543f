// Weight-streaming GEMM for decode-sized inputs:  out[M, N] = x[M, K] . W[N, K]^T,  M <= 16.
//
// At these sizes a projection is a pure stream of the weight matrix (every weight byte is
// used once, 2 M flops per byte), so the kernel is organised like the paged-decode attention
// kernel: keep as many 16-byte-per-lane loads in flight as the CU allows and let MFMA
// 16x16x32 do the (nearly free) arithmetic.
//
//  * one workgroup per 16-column tile of the output; its waves split K, so the whole chip is
//    busy even for N = 4096 (256 tiles x up to 16 waves) and the split-K reduction stays
//    inside the workgroup (LDS), never across workgroups;
//  * W is the MFMA B operand straight from global memory: lane (n = lane & 15, q = lane >> 4)
//    holds W[n0 + n][k + 8 q .. +7], i.e. 16 contiguous bytes, 64 contiguous bytes per row and
//    instruction; the loop is unrolled 8 deep so each row is read in 512-byte bursts;
//  * x (at most 16 rows, L2 resident) is the A operand, loaded the same way; rows >= M are
//    clamped duplicates whose results are dropped;
//  * projections with few column tiles (N = 4096: one workgroup per CU) can also split K over
//    workgroups (grid.y): each writes its fp32 partial tile into a slab [split][M][N] that the
//    consumer sums - the residual-add + RMSNorm kernel that follows an o / down projection
//    does that in passing, so the split costs no launch;
//  * epilogues: plain (+ optional bias), or SwiGLU: the tile computes the gate and the up
//    projection of the same 16 columns (rows n and I + n of the fused weight) with one shared
//    x fragment and stores silu(gate) * up, which removes the activation kernel and the
//    [M, 2I] intermediate.
#include "common.h"
#include <type_traits>

using namespace mvl;

namespace {

constexpr int kUnroll = 8;

// ws != nullptr (EPI 0 only): blockIdx.y selects the K range [y k_len, (y + 1) k_len) and the
// fp32 sums go to ws[y][M][16 gridDim.x] instead of out.
template <int CODE, int EPI, int WAVES, bool XL = false>
__global__ __launch_bounds__(WAVES * 64) void skinny_gemm_kernel(
        const uint16_t* __restrict__ x, long x_stride, const uint16_t* __restrict__ w,
        const uint16_t* __restrict__ bias, uint16_t* __restrict__ out, long o_stride, int M,
        int K, int up_row_offset, float* __restrict__ ws, int k_len) {
    using DT = DType<CODE>;
    using vec8 = typename DT::vec8;
    constexpr int NB = EPI == 1 ? 2 : 1;
    __shared__ float red[WAVES][NB][16][17];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c16 = lane & 15, q4 = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int kslice = k_len / WAVES;
    const long koff = (long)blockIdx.y * k_len + (long)wv * kslice + 8 * q4;
    const uint16_t* wp0 = w + (long)(n0 + c16) * K + koff;
    const uint16_t* wp1 = w + (long)(up_row_offset + n0 + c16) * K + koff;
    const uint16_t* xp = x + (long)min(c16, M - 1) * x_stride + koff;
    // XL: every wave first copies its K slice of the x rows into LDS and takes its A fragments
    // from there, which leaves the vector-memory queue to the weight stream
    extern __shared__ __attribute__((aligned(16))) uint16_t xs_all[];
    const int xs_pitch = kslice + 8;                         // elements; rows 16 B apart in banks
    uint16_t* xs = xs_all + (long)wv * M * xs_pitch;
    if (XL) {
        for (int r = 0; r < M; ++r) {
            const uint16_t* src = x + (long)r * x_stride + (long)blockIdx.y * k_len + (long)wv * kslice;
            for (int c = lane * 8; c < kslice; c += 512)
                *reinterpret_cast<u32x4_t*>(xs + r * xs_pitch + c) =
                    *reinterpret_cast<const u32x4_t*>(src + c);
        }
        __syncthreads();
    }
    const uint16_t* xl = xs + min(c16, M - 1) * xs_pitch + 8 * q4;

    f32x4_t acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // k-loop in blocks of U steps: all 2 U (3 U with SwiGLU) loads of a block are issued before
    // the first MFMA waits for one.  A slice that is not a multiple of 8 steps (K = 5120: 20
    // steps per wave) finishes with blocks of 4, 2 and 1 steps instead of dependent single steps.
    int k = 0;
    auto run = [&](auto u_tag) {
        constexpr int U = decltype(u_tag)::value;
        for (; k + 32 * U <= kslice; k += 32 * U) {
            u32x4_t bw[NB][U], ax[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                bw[0][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wp0 + k + 32 * u));
                if (NB == 2)
                    bw[NB - 1][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wp1 + k + 32 * u));
                ax[u] = XL ? *reinterpret_cast<const u32x4_t*>(xl + k + 32 * u)
                           : *reinterpret_cast<const u32x4_t*>(xp + k + 32 * u);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const vec8 a = __builtin_bit_cast(vec8, ax[u]);
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    acc[b] = DT::mfma16(a, __builtin_bit_cast(vec8, bw[b][u]), acc[b]);
            }
        }
    };
    run(std::integral_constant<int, kUnroll>{});
    run(std::integral_constant<int, 4>{});
    run(std::integral_constant<int, 2>{});
    run(std::integral_constant<int, 1>{});

    // split-K reduction across the waves; accumulator layout: col n = lane & 15, row m = 4 q + r
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv][b][4 * q4 + r][c16] = acc[b][r];
    __syncthreads();
    if (threadIdx.x < 256) {
        const int m = threadIdx.x >> 4, n = threadIdx.x & 15;
        if (m < M) {
            float s[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                s[b] = 0.f;
#pragma unroll
                for (int ww = 0; ww < WAVES; ++ww) s[b] += red[ww][b][m][n];
            }
            if (EPI == 0 && ws != nullptr) {
                ws[((long)blockIdx.y * M + m) * (16L * gridDim.x) + n0 + n] = s[0];
                return;
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

int g_force_waves = 0;        // tuning aid (mvl_skinny_gemm_force_waves): 0 = automatic choice
int g_force_xl = 0;           // x staged in LDS: 0 = automatic, 1 = whenever it fits, -1 = never

template <int CODE, int EPI>
int launch(const void* x, long xs, const void* w, const void* bias, void* out, long os, int M,
           int N_out, int K, int up_off, hipStream_t st, float* ws = nullptr, int splits = 1) {
    auto X = (const uint16_t*)x; auto W = (const uint16_t*)w; auto B = (const uint16_t*)bias;
    auto O = (uint16_t*)out;
    const dim3 tiles(N_out / 16, splits);
    const int KT = K;                            // row pitch of W and length of x rows
    K /= splits;                                 // what one workgroup covers
    // 8 waves split K (4 when K is not a multiple of 256).  Measured on Qwen3-8B shapes with
    // cold weights: 8 waves beat 16 everywhere (o-projection 8.3 vs 9.1 us, fused SwiGLU 35.9
    // vs 40.5 us) - fewer, longer streams per wave and half the LDS reduction.
    const int waves = g_force_waves ? g_force_waves : (K % 256 == 0 ? 8 : 4);
    if (waves == 16 && K % 512 == 0)
        skinny_gemm_kernel<CODE, EPI, 16><<<tiles, 16 * 64, 0, st>>>(X, xs, W, B, O, os, M, KT, up_off, ws, K);
    else if (waves == 8 && K % 256 == 0) {
        // x staged in LDS (plain epilogue) when the rows fit: <= 80 KB keeps two workgroups on
        // a CU; a grid of at most one workgroup per CU may take up to 136 KB.  Measured on the
        // Qwen3-8B projections at 1-4 rows: qkv 13.8 -> 13.2 us, o 9.5 -> 8.9, down 25.6 -> 23.3
        // (the x loads no longer take places in the vector-memory queue); the SwiGLU form,
        // which has three times the workgroups, gains nothing and keeps its x in registers.
        const size_t lds = (size_t)M * (K + 64) * 2;
        const bool one_per_cu = (long)tiles.x * tiles.y <= 256;
        const bool xl = g_force_xl >= 0 && (EPI == 0 || g_force_xl > 0)
                        && lds <= (size_t)(one_per_cu ? 136 : 80) * 1024;
        if (xl) {
            auto kern = skinny_gemm_kernel<CODE, EPI, 8, true>;
            static bool attr_set = false;
            if (!attr_set) {
                if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        136 * 1024) != hipSuccess) return -1;
                attr_set = true;
            }
            kern<<<tiles, 8 * 64, lds, st>>>(X, xs, W, B, O, os, M, KT, up_off, ws, K);
        } else {
            skinny_gemm_kernel<CODE, EPI, 8><<<tiles, 8 * 64, 0, st>>>(X, xs, W, B, O, os, M, KT, up_off, ws, K);
        }
    }
    else if (K % 128 == 0)
        skinny_gemm_kernel<CODE, EPI, 4><<<tiles, 4 * 64, 0, st>>>(X, xs, W, B, O, os, M, KT, up_off, ws, K);
    else
        return -1;
    return MVL_CHECK_LAUNCH();
}

}  // namespace

extern "C" void mvl_skinny_gemm_force_waves(int waves) { g_force_waves = waves; }
extern "C" void mvl_skinny_gemm_force_x_lds(int mode) { g_force_xl = mode; }

// epilogue 0: out[M, N] = x . W^T (+ bias);  W is [N, K] row-major, N % 16 == 0.
// epilogue 1: W is [2 I, K] (gate rows then up rows), out[M, I] = silu(x.Wg^T) * (x.Wu^T).
extern "C" int mvl_skinny_gemm(const void* x, long x_stride, const void* w, const void* bias,
                               void* out, long o_stride, int M, int N, int K, int epilogue,
                               int dtype, void* stream) {
    if (M == 0) return 0;
    if (M > 16 || K % 128 != 0) return -1;
    auto st = (hipStream_t)stream;
    if (epilogue == 0) {
        if (N % 16 != 0) return -1;
        return dtype == 0 ? launch<0, 0>(x, x_stride, w, bias, out, o_stride, M, N, K, 0, st)
                          : launch<1, 0>(x, x_stride, w, bias, out, o_stride, M, N, K, 0, st);
    }
    if (N % 32 != 0) return -1;
    const int inter = N / 2;
    return dtype == 0 ? launch<0, 1>(x, x_stride, w, bias, out, o_stride, M, inter, K, inter, st)
                      : launch<1, 1>(x, x_stride, w, bias, out, o_stride, M, inter, K, inter, st);
}

// K split over ``splits`` workgroups per column tile: ws[splits][M][N] fp32 receives the
// partial products (no bias); K / splits must be a multiple of 128.
extern "C" int mvl_skinny_gemm_splitk(const void* x, long x_stride, const void* w, void* ws, int M,
                                      int N, int K, int splits, int dtype, void* stream) {
    if (M == 0) return 0;
    if (M > 16 || N % 16 != 0 || splits < 1 || splits > 16 || ws == nullptr) return -1;
    if (K % splits != 0 || (K / splits) % 128 != 0) return -1;
    auto st = (hipStream_t)stream;
    return dtype == 0
        ? launch<0, 0>(x, x_stride, w, nullptr, nullptr, 0, M, N, K, 0, st, (float*)ws, splits)
        : launch<1, 0>(x, x_stride, w, nullptr, nullptr, 0, M, N, K, 0, st, (float*)ws, splits);
}
