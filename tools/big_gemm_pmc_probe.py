"""A few launches of the prefill-sized GEMM kernel for counter collection (run under rocprofv3
--pmc): Qwen3-8B o-projection shape (16384 x 4096 x 4096) and the fused gate_up + SwiGLU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivllm_amd.ops import hip

torch.manual_seed(0)
M = 16384
for N, K, swi in ((4096, 4096, False), (24576, 4096, True)):
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.03
    for _ in range(6):
        hip.big_linear(x, w, None, swi)
    torch.cuda.synchronize()
