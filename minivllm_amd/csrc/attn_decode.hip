// Paged-attention decode for gfx950: one query token per sequence against a block-table
// addressed KV cache.
//
// The reference kernel runs one Triton program per (sequence, QUERY head) with a scalar
// per-token loop: a block-table lookup and a D-vector load per token, K and V in separate
// passes, no sharing between the query heads of one KV head and no split over the context
// (reference: src/myvllm/layers/attention.py:283-473).  This kernel is organised around the
// KV stream instead, which is what bounds decode on HBM3E:
//
//  * one workgroup (4 waves) serves ALL query heads that share a KV head (up to 16 per
//    group), so every K/V byte is fetched once per sequence, not once per query head;
//  * grid = (context splits, KV heads, sequences); each wave walks its own 32-token tiles
//    and keeps a private online-softmax state, the 4 waves and then the splits are merged
//    flash-decoding style (the splits by whichever of their workgroups finishes last, inside
//    this kernel), so small batches still fill 256 CUs;
//  * a tile is fetched with 16-byte-per-lane coalesced loads (a (block, kv-head) slab is
//    contiguous in the engine's cache layout), staged through wave-private LDS (no
//    workgroup barrier in the loop) and prefetched one tile ahead in registers, block-table
//    entries two tiles ahead;
//  * both products run on MFMA 16x16x32: S^T = K . Q^T puts the query head on the lane and
//    the token on the accumulator rows, which is already the B-operand layout of
//    O^T = V^T . P^T, and V^T fragments come from ds_read_b64_tr_b16 (hardware transpose);
//  * the grid does not depend on context lengths (rows with length 0 exit immediately), so
//    a launch can be captured in a hipGraph and replayed for any batch content;
//  * FUSE (16-bit cache): q, and the new token's k and v, arrive as projected.  The kernel applies
//    the per-head RMSNorm and the rotary embedding to its query heads while it loads them, and
//    the wave that owns the context's last tile does the same for the new key, writes key and
//    value into their cache slot and patches them into its LDS image of that tile (the slot's
//    old content is what the tile load fetched) - which is the work of the separate
//    norm + rotate + store kernel (reference: src/myvllm/models/qwen3.py:97-116,
//    src/myvllm/layers/attention.py:63-108) without its launch and its pass over q;
//  * KV8: the cache holds OCP fp8 (e4m3) instead of 16-bit values.  A tile is then half the
//    bytes (the kernel is bound by exactly those bytes); the registers -> LDS staging step
//    widens it to the query's 16-bit type, so the LDS image and both MFMA products are the same
//    code as for a 16-bit cache.
#include "common.h"

using namespace mvl;

typedef __attribute__((ext_vector_type(8))) short s16x8_t;

namespace {

constexpr int kTile = 32;          // tokens per wave iteration
constexpr int kWaves = 4;

// Device-scope ("agent") relaxed accesses to global memory: stores are written through to
// where every XCD sees them, loads are served from there.
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) int gint;
__device__ __forceinline__ void st_agent(float* p, float v) {
    __hip_atomic_store((gfloat*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
    return __hip_atomic_load((gfloat*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Byte offset of 16-byte chunk `ch` of row `row` in a wave's [32 rows][CPR chunks] tile image.
// The XOR pattern is chosen for the lane groups in which the LDS services this kernel's
// accesses (all free of bank conflicts for 256-byte rows):
//   * ds_read_b128 of K fragments (lane = row c16, chunk 4 s + q4) is serviced in the groups
//     {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...: rows 0-3 and 12-15 of one chunk column
//     meet rows 4-11 of the next one, so row -> x(row) ^ [4 <= row <= 11] must be a bijection
//     of 16 rows onto the 16 chunk positions;
//   * ds_read_b64_tr_b16 of V fragments is serviced per 32 lanes = 8 rows x 2 chunks x 2 halves:
//     x >> 1 must take 8 values over 8 consecutive rows;
//   * staging stores of an fp8 tile pair rows r and r + 4 in a group of 8 lanes: their x must
//     differ in bit 0.
template <int CPR>
__device__ __forceinline__ int dec_tile_off(int row, int ch) {
    const int x = (((row & 7) << 1) | ((row >> 2) & 1)) & (CPR - 1);
    return row * (CPR * 16) + ((ch ^ x) << 4);
}

// the raw projections and what turns them into queries / keys (FUSE only)
struct DecodeFuse {
    const uint16_t* k_new; const uint16_t* v_new;   // [B, Hkv, D] with q's batch stride
    const float* cos_sin;                           // fp32 [positions][D]: cos of the D/2 pairs, then sin
    const uint16_t* q_w; const uint16_t* k_w;       // norm weights [D], null: no norm
    float eps;
};

template <int CODE, int D, bool KV8, bool FUSE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void paged_decode_kernel(
        const uint16_t* __restrict__ q, long q_stride_b, long q_stride_h,
        char* __restrict__ k_cache, char* __restrict__ v_cache,
        long cs_b, long cs_t, long cs_h, const int* __restrict__ block_tables, int bt_stride,
        const int* __restrict__ context_lens, uint16_t* __restrict__ out, long o_stride_b,
        long o_stride_h, float* part_acc, float* part_ml, int* tickets,
        float scale_log2, int Hq, int Hkv, int bs_log2, int nsplit, DecodeFuse fu) {
    static_assert(!(FUSE && KV8), "the fused form writes a 16-bit cache");
    using DT = DType<CODE>;
    using vec8 = typename DT::vec8;
    constexpr int CPR = D / 8;                 // 16-byte chunks per row of the LDS image
    constexpr int EB = KV8 ? 1 : 2;            // bytes per cache element
    constexpr int GCPR = D * EB / 16;          // 16-byte chunks per row in the cache
    constexpr int ITER = kTile * GCPR / 64;    // staging loads per lane per tile
    constexpr int KSTEPS = D / 32;
    constexpr int NDB = D / 16;
    constexpr int TILE_BYTES = kTile * D * 2;
    constexpr int WAVE_LDS = 2 * TILE_BYTES;
    constexpr int ACC_LD = D + 4;              // padded row of the merge buffer (floats)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c16 = lane & 15;                 // query head within the group / token row
    const int q4 = lane >> 4;
    char* k_lds = smem + w * WAVE_LDS;
    char* v_lds = k_lds + TILE_BYTES;

    const int split = blockIdx.x;
    const int G = Hq / Hkv;
    const int ngrp = (G + 15) >> 4;
    const int kvh = blockIdx.y / ngrp;
    const int hg = blockIdx.y - kvh * ngrp;
    const int b = blockIdx.z;
    const int head0 = kvh * G + hg * 16;
    const int nheads = min(16, G - hg * 16);
    const int L = context_lens[b];
    const int bs_mask = (1 << bs_log2) - 1;

    const int total_tiles = (L + kTile - 1) / kTile;
    const int tps = (total_tiles + nsplit - 1) / nsplit;
    const int t0 = split * tps;
    const int t1 = min(t0 + tps, total_tiles);

    // ---- Q^T fragments (B operand): lane holds Q[head c16][32 s + 8 q4 .. +7]
    vec8 qfrag[KSTEPS];
    {
        const uint16_t* qp = q + b * q_stride_b + (long)(head0 + (c16 < nheads ? c16 : 0)) * q_stride_h;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            u32x4_t v = {0, 0, 0, 0};
            if (c16 < nheads) v = *reinterpret_cast<const u32x4_t*>(qp + 32 * s + 8 * q4);
            qfrag[s] = __builtin_bit_cast(vec8, v);
        }
    }

    const int* bt = block_tables + (long)b * bt_stride;
    // ---- FUSE: norm + rotary embedding of the query heads (position = L - 1, the new token's),
    //      and in the wave that owns the last tile the new key / value.  A lane holds elements
    //      32 s + 8 q4 .. +7 of its row; the pair (e, e + D/2) is fragments s and s + KSTEPS/2.
    //      What it needs from memory is requested here, ahead of the first K/V tile; the
    //      arithmetic follows those requests and runs while the tile is on its way.
    constexpr int HS = KSTEPS / 2;
    const int tile_last = total_tiles - 1;
    const bool own = FUSE && L > 0 && tile_last >= t0 && tile_last < t1 && ((tile_last - t0) % kWaves) == w;
    u32x4_t knew[KSTEPS], vnew[KSTEPS];
    f32x4_t cs_raw[FUSE ? HS : 1][4];                        // [s][cos lo, cos hi, sin lo, sin hi]
    u32x4_t qw_raw[FUSE ? KSTEPS : 1], kw_raw[FUSE ? KSTEPS : 1];
    if constexpr (FUSE) {
        if (fu.q_w != nullptr) {
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s)
                qw_raw[s] = *reinterpret_cast<const u32x4_t*>(fu.q_w + 32 * s + 8 * q4);
        }
        const float* cs = fu.cos_sin + (long)max(L - 1, 0) * D;
#pragma unroll
        for (int s = 0; s < HS; ++s) {
            cs_raw[s][0] = *reinterpret_cast<const f32x4_t*>(cs + 32 * s + 8 * q4);
            cs_raw[s][1] = *reinterpret_cast<const f32x4_t*>(cs + 32 * s + 8 * q4 + 4);
            cs_raw[s][2] = *reinterpret_cast<const f32x4_t*>(cs + D / 2 + 32 * s + 8 * q4);
            cs_raw[s][3] = *reinterpret_cast<const f32x4_t*>(cs + D / 2 + 32 * s + 8 * q4 + 4);
        }
        if (own) {                                           // wave-uniform
            const long at = b * q_stride_b + (long)kvh * D + 8 * q4;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                knew[s] = *reinterpret_cast<const u32x4_t*>(fu.k_new + at + 32 * s);
                vnew[s] = *reinterpret_cast<const u32x4_t*>(fu.v_new + at + 32 * s);
                if (fu.k_w != nullptr)
                    kw_raw[s] = *reinterpret_cast<const u32x4_t*>(fu.k_w + 32 * s + 8 * q4);
            }
        }
    }
    // Address arithmetic in 32 bits wherever the launcher's range checks allow: only the block
    // base is a 64-bit product (with the fp8 cache the waves issue instructions nearly half of
    // their cycles, and the addresses of a tile were a quarter of them).
    const unsigned cs_b_bytes = (unsigned)(cs_b * EB);
    const unsigned cs_t_bytes = (unsigned)(cs_t * EB);
    const unsigned head_off_bytes = (unsigned)((long)kvh * cs_h * EB);

    // per-lane staging coordinates: chunk id = i*64 + lane -> (row, ch)
    int st_row[ITER], st_ch[ITER];
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        if constexpr (!KV8) {
            const int c = i * 64 + lane;
            st_row[i] = c / GCPR;
            st_ch[i] = c % GCPR;
        } else if constexpr (D == 128) {
            // An fp8 chunk widens into LDS chunks 2 ch and 2 ch + 1, so 8 lanes on one row would
            // store to even chunks only: half the 32 banks, twice (ds_write_b128 is banked per
            // 8 consecutive lanes).  Rows r and r + 4 differ in the low bit of their XOR
            // pattern, so 4 chunks of each per lane group cover all banks.  One load instruction still
            // reads 8 whole rows (1 KB contiguous).
            st_ch[i] = ((lane >> 3) & 1) * 4 + (lane & 3);
            st_row[i] = i * 8 + ((lane >> 2) & 1) * 4 + ((lane >> 4) & 3);
        } else {
            st_ch[i] = lane & 3;
            st_row[i] = i * 16 + ((lane >> 2) & 1) * 4 + ((lane >> 3) & 3) + ((lane >> 5) & 1) * 8;
        }
    }
    unsigned st_const[ITER];                   // byte offset of the lane's chunk inside a block
#pragma unroll
    for (int i = 0; i < ITER; ++i) st_const[i] = head_off_bytes + st_ch[i] * 16;

    auto load_block_ids = [&](int tile, int (&ids)[ITER]) {
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const unsigned tok = (unsigned)min(tile * kTile + st_row[i], L - 1);
            ids[i] = bt[tok >> bs_log2];
        }
    };
    auto issue_loads = [&](int tile, const int (&ids)[ITER], u32x4_t (&kr)[ITER], u32x4_t (&vr)[ITER]) {
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const unsigned tok = (unsigned)min(tile * kTile + st_row[i], L - 1);
            const unsigned in_block = __umul24(tok & (unsigned)bs_mask, cs_t_bytes) + st_const[i];
            const unsigned long off = (unsigned long)(unsigned)ids[i] * cs_b_bytes + in_block;   // bytes
            // every KV byte is used once per step: keep it from displacing weights in L2 / MALL
            kr[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(k_cache + off));
            vr[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(v_cache + off));
        }
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x4_t acc_o[NDB];
#pragma unroll
    for (int n = 0; n < NDB; ++n) acc_o[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    int tile = t0 + w;
    u32x4_t kreg[ITER], vreg[ITER];
    int ids_next[ITER];
    if (tile < t1) {
        int ids[ITER];
        load_block_ids(tile, ids);
        issue_loads(tile, ids, kreg, vreg);
        if (tile + kWaves < t1) load_block_ids(tile + kWaves, ids_next);
    }

    // same arithmetic as the stand-alone kernel: x * (rstd * w), rotate in fp32, round once
        auto transform = [&](auto& f, const uint16_t* wgt, bool is_q) {   // generic: FUSE only
        float x[KSTEPS][8];
        float ss = 0.f;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            unpack8<CODE>(f[s], x[s]);
#pragma unroll
            for (int i = 0; i < 8; ++i) ss += x[s][i] * x[s][i];
        }
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        if (wgt != nullptr) {
            const float rstd = rsqrtf(ss / (float)D + fu.eps);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                float wv[8];
                unpack8<CODE>(is_q ? qw_raw[s] : kw_raw[s], wv);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[s][i] *= rstd * wv[i];
            }
        }
#pragma unroll
        for (int s = 0; s < HS; ++s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float c = cs_raw[s][i >> 2][i & 3], sn = cs_raw[s][2 + (i >> 2)][i & 3];
                const float x1 = x[s][i], x2 = x[s + HS][i];
                x[s][i] = x1 * c - x2 * sn;
                x[s + HS][i] = x1 * sn + x2 * c;
            }
        }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) f[s] = pack8<CODE>(x[s]);
    };
    if constexpr (FUSE) {
        {
            u32x4_t qraw[KSTEPS];
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) qraw[s] = __builtin_bit_cast(u32x4_t, qfrag[s]);
            transform(qraw, fu.q_w, true);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) qfrag[s] = __builtin_bit_cast(vec8, qraw[s]);
        }
        if (own) transform(knew, fu.k_w, false);
    }

    for (; tile < t1; tile += kWaves) {
        // registers -> wave-private LDS image (swizzled); same-wave DS ops stay in order
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            if constexpr (KV8) {
                u32x4_t lo, hi;
                const int o0 = dec_tile_off<CPR>(st_row[i], 2 * st_ch[i]);
                const int o1 = dec_tile_off<CPR>(st_row[i], 2 * st_ch[i] + 1);
                widen_fp8x16<CODE>(kreg[i], lo, hi);
                *reinterpret_cast<u32x4_t*>(k_lds + o0) = lo;
                *reinterpret_cast<u32x4_t*>(k_lds + o1) = hi;
                widen_fp8x16<CODE>(vreg[i], lo, hi);
                *reinterpret_cast<u32x4_t*>(v_lds + o0) = lo;
                *reinterpret_cast<u32x4_t*>(v_lds + o1) = hi;
            } else {
                const int o = dec_tile_off<CPR>(st_row[i], st_ch[i]);
                *reinterpret_cast<u32x4_t*>(k_lds + o) = kreg[i];
                *reinterpret_cast<u32x4_t*>(v_lds + o) = vreg[i];
            }
        }
        if constexpr (FUSE) {
            if (own && tile == tile_last) {
                // The new token's slot has not been written yet: row (L - 1) % 32 of the image
                // and the rows behind it (the loads of a tile's tail repeat token L - 1) hold
                // whatever the slot held.  All of them get the new key / value (the tail rows
                // are masked out of the softmax, but their V rows must be finite); the first
                // head group's lanes c16 == 0 also write the slot itself.
                const int r = (L - 1) & (kTile - 1);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int row = r + c16 + 16 * half;
                    if (row < kTile) {
#pragma unroll
                        for (int s = 0; s < KSTEPS; ++s) {
                            const int o = dec_tile_off<CPR>(row, 4 * s + q4);
                            *reinterpret_cast<u32x4_t*>(k_lds + o) = knew[s];
                            *reinterpret_cast<u32x4_t*>(v_lds + o) = vnew[s];
                        }
                    }
                }
                if (c16 == 0 && hg == 0) {
                    const unsigned tok = (unsigned)(L - 1);
                    const unsigned long off = (unsigned long)(unsigned)bt[tok >> bs_log2] * cs_b_bytes
                        + __umul24(tok & (unsigned)bs_mask, cs_t_bytes) + head_off_bytes + q4 * 16;
#pragma unroll
                    for (int s = 0; s < KSTEPS; ++s) {
                        *reinterpret_cast<u32x4_t*>(k_cache + off + 64 * s) = knew[s];
                        *reinterpret_cast<u32x4_t*>(v_cache + off + 64 * s) = vnew[s];
                    }
                }
            }
        }
        if (tile + kWaves < t1) {
            issue_loads(tile + kWaves, ids_next, kreg, vreg);
            if (tile + 2 * kWaves < t1) load_block_ids(tile + 2 * kWaves, ids_next);
        }

        // ---- S^T[token][head] = K . Q^T : two 16-token sub-tiles
        f32x4_t s_acc[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            s_acc[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const u32x4_t kv = *reinterpret_cast<const u32x4_t*>(
                    k_lds + dec_tile_off<CPR>(16 * sub + c16, 4 * s + q4));
                s_acc[sub] = DT::mfma16(__builtin_bit_cast(vec8, kv), qfrag[s], s_acc[sub]);
            }
        }
        // lane (c16, q4) holds tokens 16 sub + 4 q4 + r of query head c16
        float p[8];
        float m_tile = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tok = tile * kTile + 16 * sub + 4 * q4 + r;
                const float v = tok < L ? s_acc[sub][r] * scale_log2 : -INFINITY;
                p[4 * sub + r] = v;
                m_tile = fmaxf(m_tile, v);
            }
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 16, 64));
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
        const float m_new = fmaxf(m_run, m_tile);          // finite: the tile has >= 1 token
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            p[j] = __builtin_amdgcn_exp2f(p[j] - m_new);
            psum += p[j];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        // P^T as B operand: k-slot j <-> token (j < 4 ? 4 q4 + j : 16 + 4 q4 + j - 4)
        const vec8 pfrag = __builtin_bit_cast(vec8, pack8<CODE>(p));

        // ---- O^T[d][head] += V^T . P^T
        const int tr_row = 4 * q4 + (c16 >> 2);
        const int tr_sub = (c16 & 1) * 8;
        // the running maximum settles after the first tiles of a context: skip the rescale of
        // the 32 accumulators (a round trip through the VALU) whenever no lane's maximum moved
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
            for (int n = 0; n < NDB; ++n) acc_o[n] *= alpha;
        }
#pragma unroll
        for (int n = 0; n < NDB; ++n) {
            const int ch = 2 * n + ((c16 & 3) >> 1);
            const s16x4_t lo = lds_read_tr16(v_lds, dec_tile_off<CPR>(tr_row, ch) + tr_sub);
            const s16x4_t hi = lds_read_tr16(v_lds, dec_tile_off<CPR>(16 + tr_row, ch) + tr_sub);
            const s16x8_t a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            acc_o[n] = DT::mfma16(__builtin_bit_cast(vec8, a), pfrag, acc_o[n]);
        }
    }

    // ---- merge the 4 waves through LDS (each wave reuses its own staging region)
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    float* wbuf = reinterpret_cast<float*>(smem + w * WAVE_LDS);
    if (q4 == 0) { wbuf[c16] = m_run; wbuf[16 + c16] = l_run; }
#pragma unroll
    for (int n = 0; n < NDB; ++n)
        *reinterpret_cast<f32x4_t*>(wbuf + 32 + c16 * ACC_LD + 16 * n + 4 * q4) = acc_o[n];
    __syncthreads();

    for (int e = threadIdx.x; e < 16 * D; e += 256) {
        const int c = e / D, d = e - c * D;
        if (c >= nheads) continue;
        float M = -INFINITY;
#pragma unroll
        for (int ww = 0; ww < kWaves; ++ww)
            M = fmaxf(M, reinterpret_cast<const float*>(smem + ww * WAVE_LDS)[c]);
        float o = 0.f, l = 0.f;
#pragma unroll
        for (int ww = 0; ww < kWaves; ++ww) {
            const float* wb = reinterpret_cast<const float*>(smem + ww * WAVE_LDS);
            const float f = (wb[c] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(wb[c] - M);
            o += wb[32 + c * ACC_LD + d] * f;
            l += wb[16 + c] * f;
        }
        const int head = head0 + c;
        if (nsplit == 1) {
            const float r = l > 0.f ? o / l : 0.f;
            out[b * o_stride_b + head * o_stride_h + d] = DT::from_float(r);
        } else {
            // partial results cross to another workgroup, possibly on another XCD (private
            // L2): device-scope (write-through) stores, read back below by device-scope loads
            const long pi = ((long)b * Hq + head) * nsplit + split;
            st_agent(part_acc + pi * D + d, o);
            if (d == 0) { st_agent(part_ml + pi * 2, M); st_agent(part_ml + pi * 2 + 1, l); }
        }
    }
    if (nsplit == 1) return;

    // ---- merge of the context splits by the workgroup that finishes last.  Every split of a
    // (sequence, head group) takes a ticket once its partial results are out: each wave waits
    // for its write-through stores, the workgroup meets, one lane adds to the counter.  The
    // holder of the last ticket resets the counter for the next launch and combines all
    // partials, which it reads with device-scope loads (they bypass this CU's L1 and this
    // XCD's L2, so no cache invalidate is needed; a full release/acquire fence pair here cost
    // more than the separate merge kernel it replaces).
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        gint* ctr = (gint*)(tickets + (long)b * gridDim.y + blockIdx.y);
        const int t = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == nsplit - 1;
        if (s_last) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     // no instruction: keeps the loads below

    float* s_f = reinterpret_cast<float*>(smem);       // [16 heads][64 splits] weights of the splits
    float* s_l = s_f + 16 * 64;                        // [16 heads] merged normalisers
    for (int c = w; c < nheads; c += kWaves) {         // lane = split (nsplit <= 64)
        const long base = ((long)b * Hq + head0 + c) * nsplit;
        const float ms = lane < nsplit ? ld_agent(part_ml + (base + lane) * 2) : -INFINITY;
        const float ls = lane < nsplit ? ld_agent(part_ml + (base + lane) * 2 + 1) : 0.f;
        const float M = wave_max(ms);
        const float f = (ms == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(ms - M);
        s_f[c * 64 + lane] = f;
        const float l = wave_sum(ls * f);
        if (lane == 0) s_l[c] = l;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nheads * D; e += 256) {
        const int c = e / D, d = e - c * D;
        const long base = ((long)b * Hq + head0 + c) * nsplit;
        const float* sf = s_f + c * 64;
        float o = 0.f;
        int s = 0;
        for (; s + 8 <= nsplit; s += 8) {              // 8 loads in flight: a chain of round trips otherwise
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ld_agent(part_acc + (base + s + j) * D + d);
#pragma unroll
            for (int j = 0; j < 8; ++j) o += v[j] * sf[s + j];
        }
        for (; s < nsplit; ++s) o += ld_agent(part_acc + (base + s) * D + d) * sf[s];
        const float l = s_l[c];
        out[b * o_stride_b + (head0 + c) * o_stride_h + d] = DT::from_float(l > 0.f ? o / l : 0.f);
    }
}

template <int CODE, int D, bool KV8>
int launch_decode(const void* q, long qsb, long qsh, void* kc, void* vc, long csb,
                  long cst, long csh, const int* bt, int bt_stride, const int* ctx, void* out,
                  long osb, long osh, float* part_acc, float* part_ml, int* tickets, float scale,
                  int B, int Hq, int Hkv, int block_size, int nsplit, DecodeFuse fu, hipStream_t st) {
    int bs_log2 = 0;
    while ((1 << bs_log2) < block_size) ++bs_log2;
    // the kernel forms in-block byte offsets in 32 bits (token stride through a 24-bit multiply)
    constexpr long EB = KV8 ? 1 : 2;
    if (csb <= 0 || cst <= 0 || csh < 0 || csb * EB >= (1L << 32) || cst * EB >= (1L << 24) ||
        ((long)block_size * cst + (long)Hkv * csh) * EB + D * EB >= (1L << 31))
        return -1;
    const int G = Hq / Hkv;
    const int ngrp = (G + 15) / 16;
    dim3 grid(nsplit, Hkv * ngrp, B);
    const size_t lds = (size_t)kWaves * 2 * kTile * D * 2;
    if constexpr (!KV8) {
        if (fu.k_new != nullptr) {
            paged_decode_kernel<CODE, D, KV8, true><<<grid, 256, lds, st>>>(
                (const uint16_t*)q, qsb, qsh, (char*)kc, (char*)vc, csb, cst, csh, bt,
                bt_stride, ctx, (uint16_t*)out, osb, osh, part_acc, part_ml, tickets, scale * kLog2e,
                Hq, Hkv, bs_log2, nsplit, fu);
            return MVL_CHECK_LAUNCH();
        }
    }
    if (fu.k_new != nullptr) return -1;
    paged_decode_kernel<CODE, D, KV8><<<grid, 256, lds, st>>>(
        (const uint16_t*)q, qsb, qsh, (char*)kc, (char*)vc, csb, cst, csh, bt,
        bt_stride, ctx, (uint16_t*)out, osb, osh, part_acc, part_ml, tickets, scale * kLog2e, Hq,
        Hkv, bs_log2, nsplit, fu);
    return MVL_CHECK_LAUNCH();
}

}  // namespace

// q [B, Hq, D] (strides in elements), caches addressed by (block, token, kv head) strides in
// cache elements, block_size a power of two, workspaces needed only when nsplit > 1:
// part_acc [B*Hq*nsplit*D] fp32, part_ml [B*Hq*nsplit*2] fp32, tickets [B * Hkv * ceil(G/16)]
// ints that are zero at the first launch and shared with no launch that may run at the same
// time (the kernel leaves them zero).  kv_fp8 != 0: the caches hold OCP fp8 e4m3 bytes instead
// of values of the query's dtype.
// k_new != null (16-bit cache only): q, k_new [B, Hkv, D] and v_new (q's batch stride) are the raw
// projections of the step's new tokens, whose positions are context_lens - 1; the kernel norms
// (q_w / k_w, each may be null) and rotates (cos_sin) q and k_new itself and writes k / v into
// the caches at those positions before the step's attention reads them.
extern "C" int mvl_paged_attention_decode(
        const void* q, long q_stride_b, long q_stride_h, void* k_cache, void* v_cache,
        long cs_b, long cs_t, long cs_h, const int* block_tables, int bt_stride,
        const int* context_lens, void* out, long o_stride_b, long o_stride_h, float* part_acc,
        float* part_ml, int* tickets, float scale, int B, int Hq, int Hkv, int D, int block_size,
        int nsplit, int dtype, int kv_fp8, const void* k_new, const void* v_new,
        const float* cos_sin, const void* q_w, const void* k_w, float eps, void* stream) {
    if (B == 0) return 0;
    if (k_new != nullptr && (v_new == nullptr || cos_sin == nullptr || kv_fp8)) return -1;
    DecodeFuse fu;
    fu.k_new = (const uint16_t*)k_new; fu.v_new = (const uint16_t*)v_new; fu.cos_sin = cos_sin;
    fu.q_w = (const uint16_t*)q_w; fu.k_w = (const uint16_t*)k_w; fu.eps = eps;
    if ((D != 64 && D != 128) || Hq % Hkv != 0 || (block_size & (block_size - 1)) || nsplit < 1 ||
        nsplit > 64)
        return -1;
    if (nsplit > 1 && (part_acc == nullptr || part_ml == nullptr || tickets == nullptr)) return -2;
    auto st = (hipStream_t)stream;
#define MVL_DEC(CODE, DD, K8)                                                                   \
    return launch_decode<CODE, DD, K8>(q, q_stride_b, q_stride_h, k_cache, v_cache, cs_b, cs_t,    \
                                       cs_h, block_tables, bt_stride, context_lens, out,           \
                                       o_stride_b, o_stride_h, part_acc, part_ml, tickets, scale,  \
                                       B, Hq, Hkv, block_size, nsplit, fu, st)
    if (kv_fp8) {
        if (dtype == 0) { if (D == 128) MVL_DEC(0, 128, true); else MVL_DEC(0, 64, true); }
        else            { if (D == 128) MVL_DEC(1, 128, true); else MVL_DEC(1, 64, true); }
    }
    if (dtype == 0) { if (D == 128) MVL_DEC(0, 128, false); else MVL_DEC(0, 64, false); }
    else            { if (D == 128) MVL_DEC(1, 128, false); else MVL_DEC(1, 64, false); }
#undef MVL_DEC
}
