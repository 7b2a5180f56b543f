"""Does a weight matrix that another kernel has just read (so that it sits in the 256 MB
memory-side cache) stream faster through the weight-streaming GEMM than one that comes from HBM?
Times the 1-row skinny GEMM on Qwen3-8B projections: cold (the matrix was last touched > 1 GB of
traffic ago) against warm (a reduction kernel read exactly this matrix right before)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivllm_amd.ops import hip

dev = "cuda"
torch.manual_seed(0)
SHAPES = [("o", 4096, 4096, False), ("qkv", 6144, 4096, False), ("down", 4096, 12288, False),
          ("gate_up", 24576, 4096, True)]
filler = torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def flush():
    filler.view(torch.int32).sum()          # 1 GB of reads: nothing of the weights stays cached


def touch(w):
    return w.view(torch.int32).sum()        # streams the matrix once


for name, N, K, swi in SHAPES:
    ws = [(torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.03) for _ in range(3)]
    x = torch.randn(1, K, device=dev, dtype=torch.bfloat16)
    for w in ws:
        hip.skinny_linear(x, w, None, swiglu=swi)
    res = {}
    for mode in ("cold", "warm"):
        ts = []
        for rep in range(6):
            w = ws[rep % 3]
            flush()
            if mode == "warm":
                touch(w)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            hip.skinny_linear(x, w, None, swiglu=swi)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res[mode] = sorted(ts)[len(ts) // 2]
    mb = N * K * 2 / 1e6
    print(f"{name:8s} {mb:6.1f} MB: cold {res['cold']:6.1f} us ({mb / res['cold'] / 1e6 * 1e6:.2f} TB/s)   "
          f"warm {res['warm']:6.1f} us ({mb / res['warm']:.2f} TB/s)", flush=True)
