// Per-workgroup phase timestamps of the flash prefill kernel (csrc/attn_prefill.hip built with
// -DMVL_PREFILL_PROBE): entry, after the first K/V tile has landed, after the unmasked tiles,
// after the diagonal tiles; plus the tile count and the CU the workgroup ran on.  Answers where
// the time of short-sequence prefill goes (latency before the first tile, diagonal tiles,
// idle slots at the end of the grid).  Output: CSV on stdout, one line per workgroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMVL_PREFILL_PROBE -I minivllm_amd/csrc \
//         tools/native/prefill_phase_probe.hip -o tools/native/prefill_phase_probe
//   ./prefill_phase_probe B S [Hq Hkv]
#define MVL_PREFILL_PROBE 1
#include "../../minivllm_amd/csrc/attn_prefill.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16, S = argc > 2 ? atoi(argv[2]) : 1024;
    const int Hq = argc > 3 ? atoi(argv[3]) : 32, Hkv = argc > 4 ? atoi(argv[4]) : 8, D = 128;
    const long T = (long)B * S;
    std::vector<uint16_t> hq((size_t)T * Hq * D), hk((size_t)T * Hkv * D);
    unsigned s = 12345;
    auto rnd = [&]() {                       // bf16 in about [-1, 1]
        s = s * 1664525u + 1013904223u;
        const float f = ((s >> 8) & 0xffff) / 32768.f - 1.f;
        unsigned u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16);
    };
    for (auto& v : hq) v = rnd();
    for (auto& v : hk) v = rnd();
    std::vector<int> cu(B + 1);
    for (int i = 0; i <= B; ++i) cu[i] = i * S;
    uint16_t *q, *k, *v, *o; int* dcu; unsigned long long* probe;
    CK(hipMalloc(&q, hq.size() * 2)); CK(hipMalloc(&o, hq.size() * 2));
    CK(hipMalloc(&k, hk.size() * 2)); CK(hipMalloc(&v, hk.size() * 2));
    CK(hipMalloc(&dcu, (B + 1) * 4));
    CK(hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(k, hk.data(), hk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(v, hk.data(), hk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcu, cu.data(), (B + 1) * 4, hipMemcpyHostToDevice));
    PrefillArgs a;
    a.q = q; a.q_stride_t = (long)Hq * D; a.q_stride_h = D;
    // K/V head-major ([Hkv][T][D]): a 64-token tile is one contiguous 16 KiB piece, as in the
    // paged cache with blocks of >= 64 tokens (token-major [T][Hkv][D] rows, 2 KiB apart, run
    // the kernel about half as fast: MVL_PROBE_TOKEN_MAJOR=1)
    const bool token_major = getenv("MVL_PROBE_TOKEN_MAJOR") != nullptr;
    a.k = k; a.v = v; a.cs_b = 0;
    a.k_stride_t = token_major ? (long)Hkv * D : D;
    a.k_stride_h = token_major ? D : T * D;
    a.out = o; a.o_stride_t = (long)Hq * D; a.o_stride_h = D;
    a.cu_q = dcu; a.cu_k = dcu; a.block_tables = nullptr; a.bt_stride = 0; a.bs_log2 = 0;
    a.scale_log2 = kLog2e / sqrtf((float)D);
    a.B = B; a.Hq = Hq; a.Hkv = Hkv;
    a.positions = nullptr; a.cos_sin = nullptr; a.q_w = nullptr; a.eps = 0.f;
    a.n_qtiles = (S + kBM - 1) / kBM;
    // default: one Q tile per workgroup (the stamps are per tile count); argv[5] = pairs per workgroup
    a.pair = argc > 5 ? atoi(argv[5]) : 0;
    a.n_items = a.pair ? ((a.n_qtiles + 1) / 2 + a.pair - 1) / a.pair : a.n_qtiles;
    a.total_wg = B * Hq * a.n_items;
    CK(hipMalloc(&probe, (size_t)a.total_wg * 8 * 8));
    CK(hipMemset(probe, 0, (size_t)a.total_wg * 8 * 8));
    a.probe = probe;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0.f;
    const bool quiet = argc > 6;                 // argv[6]: timing only, no CSV
    // the clocks of an idle GPU ramp up over the first milliseconds of work
    const int warm = getenv("MVL_PROBE_WARMUP") ? atoi(getenv("MVL_PROBE_WARMUP")) : 0;
    for (int rep = 0; rep < warm; ++rep)
        if (launch_prefill<0, 128, false, false, true>(a, 0)) { fprintf(stderr, "launch failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 5; ++rep) {      // the last launch's timestamps are the ones kept
        CK(hipEventRecord(e0, 0));
        if (launch_prefill<0, 128, false, false, true>(a, 0)) { fprintf(stderr, "launch failed\n"); return 1; }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        fprintf(stderr, "B=%d S=%d Hq=%d Hkv=%d: %.1f us\n", B, S, Hq, Hkv, ms * 1e3f);
    }
    if (quiet) return 0;
    std::vector<unsigned long long> h((size_t)a.total_wg * 8);
    CK(hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost));
    printf("wg,t_entry,t_first_tile,t_unmasked_done,t_end,n_tiles,hw_id,xcc_id\n");
    for (int i = 0; i < a.total_wg; ++i)
        printf("%d,%llu,%llu,%llu,%llu,%llu,%llu,%llu\n", i, h[8 * i], h[8 * i + 1], h[8 * i + 2],
               h[8 * i + 3], h[8 * i + 4], h[8 * i + 5], h[8 * i + 6]);
    return 0;
}
