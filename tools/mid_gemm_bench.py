"""Mid-batch weight-streaming GEMM (csrc/mid_gemm.hip) against the library GEMM on the decode
projections of a model at 32..256 rows, weights rotating so that each call streams from HBM.
Prints every candidate configuration; `--json` writes the table.

    python tools/mid_gemm_bench.py [--model qwen3-8b] [--rows 256] [--json out.json]
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from minivllm_amd.ops import hip
from minivllm_amd.utils.gemm_tuning import enable_tuned_gemms

SHAPES = {
    "qwen3-8b": {"qkv": (6144, 4096, False), "o": (4096, 4096, False),
                 "gate_up": (24576, 4096, True), "down": (4096, 12288, False)},
    "qwen3-0.6b": {"qkv": (4096, 1024, False), "o": (1024, 2048, False),
                   "gate_up": (6144, 1024, True), "down": (1024, 3072, False)},
    "qwen3-32b": {"qkv": (10240, 5120, False), "o": (5120, 8192, False),
                  "gate_up": (51200, 5120, True), "down": (5120, 25600, False)},
    "qwen3-8b-tp8": {"qkv": (768, 4096, False), "o": (4096, 512, False),
                     "gate_up": (3072, 4096, True), "down": (4096, 1536, False)},
}

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="qwen3-8b")
ap.add_argument("--rows", default="32,64,128,256")
ap.add_argument("--json", default=None)
ap.add_argument("--only", default=None, help="comma list of projections")
args = ap.parse_args()
print("tuned table:", enable_tuned_gemms(), flush=True)
result = {}
for M in [int(r) for r in args.rows.split(",")]:
    tot_lib = tot_best = 0.0
    for name, (N, K, swiglu) in SHAPES[args.model].items():
        if args.only and name not in args.only.split(","):
            continue
        copies = max(4, -(-(640 << 20) // (N * K * 2)))
        ws = [(torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16) for _ in range(copies)]
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        lib = lambda xx, w: hip.silu_and_mul(F.linear(xx, w)) if swiglu else F.linear(xx, w)
        t_lib = hip._graph_time_ms(lib, x, ws, 2) * 1e3
        row = {"lib_us": round(t_lib, 1)}
        best = None
        for bn, sp in hip.mid_gemm_candidates(M, N, K, swiglu):
            for nt in (True, False):
                t = hip._graph_time_ms(
                    lambda xx, w: hip.mid_linear(xx, w, None, swiglu, bn, sp, nt), x, ws, 2) * 1e3
                row[f"bn{bn}_s{sp}_{'nt' if nt else 'ld'}"] = round(t, 1)
                if best is None or t < best[0]:
                    best = (t, bn, sp, nt)
        tot_lib += t_lib
        tot_best += min(best[0], t_lib)
        fl = 2 * M * N * K
        print(f"M={M:3d} {name:8s} lib {t_lib:6.1f} us | best kernel {best[0]:6.1f} us "
              f"(bn {best[1]}, splits {best[2]}, nt {best[3]}) {N*K*2/best[0]/1e6:4.2f} TB/s "
              f"{fl/best[0]/1e9:4.2f} PF/s | {row}", flush=True)
        result[f"{M}:{name}"] = row
        del ws
    print(f"M={M:3d} layer: library {tot_lib:6.1f} us, best-of {tot_best:6.1f} us", flush=True)
if args.json:
    json.dump(result, open(args.json, "w"), indent=1)
