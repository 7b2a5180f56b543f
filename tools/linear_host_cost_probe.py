#!/usr/bin/env python3
"""Host-side cost of one ``F.linear`` call with and without the TunableOp table (a small GEMM,
so that the loop is bound by the host, not the GPU)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F


def cost(x, w, n=400):
    for _ in range(20):
        F.linear(x, w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        F.linear(x, w)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6, (time.perf_counter() - t0) / n * 1e6


w = torch.randn(1024, 1024, device="cuda", dtype=torch.bfloat16)
for M in (64, 500, 512):
    x = torch.randn(M, 1024, device="cuda", dtype=torch.bfloat16)
    print(f"M={M:4d} plain          host {cost(x, w)[0]:6.1f} us/call", flush=True)
from minivllm_amd.utils.gemm_tuning import enable_tuned_gemms
print("table loaded:", enable_tuned_gemms(), flush=True)
for M in (64, 500, 512):
    x = torch.randn(M, 1024, device="cuda", dtype=torch.bfloat16)
    print(f"M={M:4d} with TunableOp host {cost(x, w)[0]:6.1f} us/call", flush=True)
