"""Prefill MLP (gate_up GEMM -> SwiGLU -> down GEMM) over all tokens at once vs in token chunks
small enough for the [chunk, 2I] intermediate to stay in the 256 MB Infinity Cache."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivllm_amd.ops import hip
from minivllm_amd.utils.gemm_tuning import enable_tuned_gemms

print("tuned table:", enable_tuned_gemms(), flush=True)
H, I, T = 4096, 12288, 16384
x = torch.randn(T, H, device="cuda", dtype=torch.bfloat16)
layers = [(torch.randn(2 * I, H, device="cuda", dtype=torch.bfloat16) * 0.02,
           torch.randn(H, I, device="cuda", dtype=torch.bfloat16) * 0.02) for _ in range(6)]
out = torch.empty(T, H, device="cuda", dtype=torch.bfloat16)


def mlp(wgu, wd, chunk):
    for lo in range(0, T, chunk):
        a = hip.silu_and_mul(torch.nn.functional.linear(x[lo:lo + chunk], wgu))
        torch.mm(a, wd.t(), out=out[lo:lo + chunk])


for chunk in (16384, 8192, 4096, 2048, 1024):
    for wgu, wd in layers:
        mlp(wgu, wd, chunk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        for wgu, wd in layers:
            mlp(wgu, wd, chunk)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (3 * len(layers))
    flops = 2 * T * H * 3 * I
    print(f"chunk {chunk:6d}: {dt*1e3:7.3f} ms per layer  ({flops/dt/1e15:.2f} PF/s incl. SwiGLU)", flush=True)
