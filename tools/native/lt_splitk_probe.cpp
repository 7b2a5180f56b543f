// hipBLASLt extension API probe: for the M = 128/256 decode projections of Qwen3-8B, does any
// (solution, split-K) pair beat the best plain solution?  PyTorch's TunableOp searches solutions
// but leaves split-K at each solution's built-in value; GemmTuning::setSplitK makes it a knob.
//
//   out[M, N] = x[M, K] . W[N, K]^T   (row-major)  ==  column-major  D[N, M] = W^T-op(A) * B
//   with A = W (K x N column-major, op T), B = x (K x M column-major, op N).
//
// Weights rotate through copies larger than the 256 MB Infinity Cache, so every call streams
// them from HBM as in the decode step.  Standalone (no PyTorch); build:
//   hipcc --offload-arch=gfx950 -O2 tools/native/lt_splitk_probe.cpp -lhipblaslt -o lt_splitk_probe
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { std::fprintf(stderr, "%s failed: %d (line %d)\n", #x, (int)e_, __LINE__); std::exit(1); } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? std::atoi(argv[1]) : 256;
    struct Shape { const char* name; int N, K; };
    const Shape shapes[] = {{"qkv", 6144, 4096}, {"o", 4096, 4096}, {"down", 4096, 12288}, {"gate_up", 24576, 4096}};
    hipblasLtHandle_t handle;
    CK(hipblasLtCreate(&handle));
    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    const size_t ws_bytes = 256u << 20;
    void* ws;
    CK(hipMalloc(&ws, ws_bytes));
    std::vector<hipblasLtMatmulHeuristicResult_t> all;
    CK(hipblaslt_ext::getAllAlgos(handle, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, HIPBLAS_OP_T, HIPBLAS_OP_N,
                                  HIP_R_16BF, HIP_R_16BF, HIP_R_16BF, HIP_R_16BF, HIPBLAS_COMPUTE_32F, all));
    std::printf("M=%d: %zu solutions for bf16 TN\n", M, all.size());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float alpha = 1.f, beta = 0.f;
    for (const Shape& s : shapes) {
        const size_t wbytes = (size_t)s.N * s.K * 2;
        const int copies = (int)std::max<size_t>(4, ((640u << 20) + wbytes - 1) / wbytes);
        void *w, *x, *d;
        CK(hipMalloc(&w, wbytes * copies));
        CK(hipMalloc(&x, (size_t)M * s.K * 2));
        CK(hipMalloc(&d, (size_t)M * s.N * 2));
        CK(hipMemset(w, 0x11, wbytes * copies));       // small finite bf16 values
        CK(hipMemset(x, 0x11, (size_t)M * s.K * 2));
        struct Res { float us; int idx; int splitk; };
        std::vector<Res> results;
        size_t visited = 0;
        for (auto& h : all) {
            if (++visited % 200 == 0) { std::printf("  ... %s: %zu / %zu solutions\n", s.name, visited, all.size()); std::fflush(stdout); }
            for (int sk : {0, 2, 3, 4, 6, 8}) {
                hipblaslt_ext::Gemm gemm(handle, HIPBLAS_OP_T, HIPBLAS_OP_N, HIP_R_16BF, HIP_R_16BF, HIP_R_16BF,
                                         HIP_R_16BF, HIPBLAS_COMPUTE_32F);
                hipblaslt_ext::GemmEpilogue epi;
                hipblaslt_ext::GemmInputs in;
                in.setA(w); in.setB(x); in.setC(d); in.setD(d); in.setAlpha(&alpha); in.setBeta(&beta);
                if (gemm.setProblem(s.N, M, s.K, 1, epi, in) != HIPBLAS_STATUS_SUCCESS) continue;
                hipblaslt_ext::GemmTuning tune;
                tune.setSplitK((uint16_t)sk);
                size_t need = 0;
                if (gemm.isAlgoSupported(h.algo, tune, need) != HIPBLAS_STATUS_SUCCESS || need > ws_bytes) continue;
                // one instance per weight copy (pointers are baked in at initialize)
                std::vector<hipblaslt_ext::Gemm> inst;
                bool ok = true;
                for (int c = 0; c < copies && ok; ++c) {
                    inst.emplace_back(handle, HIPBLAS_OP_T, HIPBLAS_OP_N, HIP_R_16BF, HIP_R_16BF, HIP_R_16BF,
                                      HIP_R_16BF, HIPBLAS_COMPUTE_32F);
                    hipblaslt_ext::GemmInputs ic;
                    ic.setA((char*)w + (size_t)c * wbytes); ic.setB(x); ic.setC(d); ic.setD(d);
                    ic.setAlpha(&alpha); ic.setBeta(&beta);
                    ok = inst.back().setProblem(s.N, M, s.K, 1, epi, ic) == HIPBLAS_STATUS_SUCCESS
                         && inst.back().isAlgoSupported(h.algo, tune, need) == HIPBLAS_STATUS_SUCCESS
                         && inst.back().initialize(h.algo, tune, ws, false, stream) == HIPBLAS_STATUS_SUCCESS;
                }
                if (!ok) continue;
                for (auto& g : inst) if (g.run(stream) != HIPBLAS_STATUS_SUCCESS) { ok = false; break; }
                if (!ok || hipStreamSynchronize(stream) != hipSuccess) continue;
                // screen with one pass; the slow majority is dropped early
                CK(hipEventRecord(e0, stream));
                for (auto& g : inst) g.run(stream);
                CK(hipEventRecord(e1, stream));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                float us = ms * 1e3f / copies;
                if (!results.empty() && us > 1.5f * results.front().us) { results.push_back({us, hipblaslt_ext::getIndexFromAlgo(h.algo), sk}); continue; }
                CK(hipEventRecord(e0, stream));
                for (int r = 0; r < 4; ++r) for (auto& g : inst) g.run(stream);
                CK(hipEventRecord(e1, stream));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                us = ms * 1e3f / (4 * copies);
                results.push_back({us, hipblaslt_ext::getIndexFromAlgo(h.algo), sk});
                std::sort(results.begin(), results.end(), [](const Res& a, const Res& b) { return a.us < b.us; });
            }
        }
        std::sort(results.begin(), results.end(), [](const Res& a, const Res& b) { return a.us < b.us; });
        float best_plain = 1e30f;
        for (auto& r : results) if (r.splitk == 0) best_plain = std::min(best_plain, r.us);
        std::printf("M=%d %-8s N=%d K=%d: best plain %.1f us | top:", M, s.name, s.N, s.K, best_plain);
        for (size_t i = 0; i < std::min<size_t>(5, results.size()); ++i)
            std::printf("  %.1f us (sol %d, splitK %d)", results[i].us, results[i].idx, results[i].splitk);
        std::printf("\n");
        std::fflush(stdout);
        CK(hipFree(w)); CK(hipFree(x)); CK(hipFree(d));
    }
    return 0;
}
