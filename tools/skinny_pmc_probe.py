#!/usr/bin/env python3
"""A handful of weight-streaming GEMM launches for counter collection (run under rocprofv3
--pmc): Qwen3-8B o-projection and fused gate_up+SwiGLU at 1 and 8 rows, rotating weights so
every launch streams from HBM."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from minivllm_amd.ops import hip

torch.manual_seed(0)
for M in (1, 8):
    x = torch.randn(M, 4096, device="cuda", dtype=torch.bfloat16)
    wo = [torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16) * 0.02 for _ in range(6)]
    wg = [torch.randn(24576, 4096, device="cuda", dtype=torch.bfloat16) * 0.02 for _ in range(3)]
    for w in wo:
        hip.skinny_linear(x, w)
    for w in wg:
        hip.skinny_linear(x, w, swiglu=True)
torch.cuda.synchronize()
print("done")
