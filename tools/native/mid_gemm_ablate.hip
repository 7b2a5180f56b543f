// Timing harness for csrc/mid_gemm.hip built with -DMVL_MID_ABLATE=k (0 full kernel, 1 without
// the MFMAs, 2 without the in-loop DMA, 3 DMA only): which of the three streams of work bounds
// the kernel.  Weights rotate over > 640 MB so that every launch streams from HBM.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMVL_MID_ABLATE=1 -I minivllm_amd/csrc \
//         tools/native/mid_gemm_ablate.hip -o tools/native/mid_gemm_ablate_1
#include "../../minivllm_amd/csrc/mid_gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    struct Shape { const char* name; int N, K, swiglu, bn, splits; };
    std::vector<Shape> shapes = {{"gate_up", 24576, 4096, 1, 128, 1}, {"qkv", 6144, 4096, 0, 128, 5},
                                 {"o", 4096, 4096, 0, 64, 4}, {"down", 4096, 12288, 0, 128, 8}};
    const int M = argc > 1 ? atoi(argv[1]) : 256;
    const int nt = argc > 2 ? atoi(argv[2]) : 1;
    // optional overrides: <projection> <bn> <splits> replaces the table with that one case
    const int depth_cfg = argc > 6 ? atoi(argv[6]) : 0;
    if (argc > 7) shapes = {{"custom", atoi(argv[7]), 4096, 0, 128, 1}};      // N override, K = 4096
    if (argc > 5) {
        for (auto& sh : shapes)
            if (!strcmp(sh.name, argv[3]) || argc > 7) { Shape one = sh; one.bn = atoi(argv[4]); one.splits = atoi(argv[5]); shapes = {one}; break; }
    }
    for (auto& sh : shapes) {
        const size_t wbytes = (size_t)sh.N * sh.K * 2;
        const int copies = (int)((640ull << 20) / wbytes) + 1;
        uint16_t *w, *x, *out; float* ws;
        CK(hipMalloc(&w, wbytes * copies));
        CK(hipMalloc(&x, (size_t)M * sh.K * 2));
        CK(hipMalloc(&out, (size_t)M * sh.N * 2));
        CK(hipMalloc(&ws, (size_t)sh.splits * M * sh.N * 4));
        CK(hipMemset(w, 0x11, wbytes * copies));        // small finite bf16 values
        CK(hipMemset(x, 0x11, (size_t)M * sh.K * 2));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 3 * copies;
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) {
                int rc;
                if (depth_cfg) {
                    MidArgs a; a.x = x; a.x_stride = sh.K; a.w = w + (size_t)(i % copies) * sh.N * sh.K;
                    a.bias = nullptr; a.out = out; a.o_stride = sh.swiglu ? sh.N / 2 : sh.N; a.ws = ws;
                    a.M = M; a.N = sh.N; a.K = sh.K; a.up_off = sh.N / 2; a.ksteps = sh.K / 64;
                    a.ksteps_per_split = (a.ksteps + sh.splits - 1) / sh.splits;
                    switch (depth_cfg) {       // 256 x 128 tile, ring depth 2 / 3
                    case 2: rc = launch_tile<0, 256, 128, 4, 2, false, true>(a, sh.splits, 0); break;
                    case 3: rc = launch_tile<0, 256, 128, 4, 3, false, true>(a, sh.splits, 0); break;
                    default: rc = -1;
                    }
                } else
                rc = mvl_mid_gemm(x, sh.K, w + (size_t)(i % copies) * sh.N * sh.K, nullptr, out,
                                      sh.swiglu ? sh.N / 2 : sh.N, ws, M, sh.N, sh.K, sh.swiglu,
                                      M <= 64 ? 64 : M <= 128 ? 128 : 256, sh.bn, sh.splits, nt, 0, nullptr);
                if (rc) { printf("launch failed %d\n", rc); return 1; }
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("ablate %d M %d %-8s bn %d splits %d: %.1f us\n", MVL_MID_ABLATE, M, sh.name, sh.bn,
               sh.splits, ms * 1e3 / reps);
        CK(hipFree(w)); CK(hipFree(x)); CK(hipFree(out)); CK(hipFree(ws));
    }
    return 0;
}
