"""Reads the clock stamps a -DMVL_BIG_STAMP=1 build of csrc/big_gemm.hip leaves at the start of
its output (run with MINIVLLM_HIP_LIB pointing at that build): per phase of the last K-tile
before a tile's final one, for one wave of each wave row of workgroup 0:
load segment | wait at barrier | MFMA cluster | wait at barrier."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivllm_amd.ops import hip

M, N, K = 16384, 4096, 4096
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.03
for rep in range(3):
    out = hip.big_linear(x, w)
    torch.cuda.synchronize()
st = out.view(-1)[:128].view(torch.int64).cpu().tolist()
for wave, base in ((0, 0), (4, 16)):
    t = st[base:base + 16]
    print(f"wave {wave}: stamps relative to the first (100 MHz ticks unless the counter is the shader clock)")
    print("   ", [v - t[0] for v in t])
    for p in range(4):
        a, b, c, d = t[4 * p:4 * p + 4]
        nxt = t[4 * p + 4] if p < 3 else None
        print(f"    phase {p + 1}: loads {b - a:5d}  barrier {c - b:5d}  mfma {d - c:5d}  barrier {'' if nxt is None else nxt - d}")
t0, t4 = st[0:16], st[16:32]
print("wave 4 minus wave 0 at each stamp:", [b - a for a, b in zip(t0, t4)])
