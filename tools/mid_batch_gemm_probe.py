"""Library GEMM (with the shipped TunableOp table) on the decode projections at 32..256 rows,
weights rotating so that each call streams from HBM: time, weight bandwidth and MFMA rate per
shape.  Shows how far the mid-batch decode GEMMs are from the HBM and MFMA limits."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivllm_amd.utils.gemm_tuning import enable_tuned_gemms

SHAPES = {
    "qwen3-8b": {"qkv": (6144, 4096), "o": (4096, 4096), "gate_up": (24576, 4096), "down": (4096, 12288)},
    "qwen3-0.6b": {"qkv": (4096, 1024), "o": (1024, 2048), "gate_up": (6144, 1024), "down": (1024, 3072)},
}
shapes = SHAPES[os.environ.get("MODEL", "qwen3-8b")]
print("tuned table:", enable_tuned_gemms(), flush=True)


def bench(fn, x, ws):
    for w in ws:
        fn(x, w)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(5):
            for w in ws:
                fn(x, w)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (25 * len(ws))


for M in (32, 64, 128, 256):
    total = 0.0
    for name, (N, K) in shapes.items():
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        copies = max(4, -(-(640 << 20) // (N * K * 2)))
        ws = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) for _ in range(copies)]
        t = bench(torch.nn.functional.linear, x, ws)
        total += t
        print(f"M={M:3d} {name:8s} {t*1e6:6.1f} us  {N*K*2/t/1e12:4.2f} TB/s  "
              f"{2*M*N*K/t/1e15:4.2f} PF/s", flush=True)
        del ws
    print(f"M={M:3d} layer total {total*1e6:6.1f} us", flush=True)
