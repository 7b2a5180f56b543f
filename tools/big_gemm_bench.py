"""Prefill-sized GEMM: the hand-written 256 x 256-tile kernel (csrc/big_gemm.hip) against the
library path (F.linear with the shipped tuned table, + the SiluAndMul kernel for gate_up), on the
projection shapes of Qwen3-0.6B and Qwen3-8B at 16384 tokens.  Random data; the two sides are
timed interleaved in one process; every kernel result is checked against an fp32 product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from minivllm_amd.ops import hip
from minivllm_amd.utils.gemm_tuning import enable_tuned_gemms

print("tuned table:", enable_tuned_gemms(), flush=True)
torch.manual_seed(0)
dev = "cuda"
M = int(os.environ.get("ROWS", 16384))
SHAPES = [  # name, N, K, swiglu
    ("0.6b qkv", 4096, 1024, False), ("0.6b o", 1024, 2048, False),
    ("0.6b gate_up", 6144, 1024, True), ("0.6b down", 1024, 3072, False),
    ("8b qkv", 6144, 4096, False), ("8b o", 4096, 4096, False),
    ("8b gate_up", 24576, 4096, True), ("8b down", 4096, 12288, False),
]
only = os.environ.get("ONLY")


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, N, K, swi in SHAPES:
    if only and only not in name:
        continue
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.03
    lib = (lambda: hip.silu_and_mul(F.linear(x, w))) if swi else (lambda: F.linear(x, w))
    ker = lambda: hip.big_linear(x, w, None, swi)
    # correctness on a band of rows (the fp32 product of everything would take 3 GB at the largest shape)
    got = ker()
    rows = torch.cat([torch.arange(0, 300), torch.arange(M // 2 - 130, M // 2 + 130), torch.arange(M - 300, M)]).to(dev)
    ref = x[rows].float() @ w.float().t()
    if swi:
        ref = F.silu(ref[:, :N // 2]) * ref[:, N // 2:]
    err = (got[rows].float() - ref).abs().max().item()
    scale = ref.abs().max().item()
    lib_err = (lib()[rows].float() - ref).abs().max().item()
    for f in (lib, ker):
        f(); f()
    torch.cuda.synchronize()
    n = 10 if N * K > 5e7 else 30
    tl, tk = [], []
    for _ in range(5):
        tl.append(timed(lib, n)); tk.append(timed(ker, n))
    tl, tk = sorted(tl)[2], sorted(tk)[2]
    fl = 2.0 * M * N * K
    print(f"{name:13s} M {M} N {N:5d} K {K:5d}: library {tl:7.1f} us ({fl/tl/1e9:5.2f} PF/s)   kernel {tk:7.1f} us "
          f"({fl/tk/1e9:5.2f} PF/s)   ratio {tl/tk:4.2f}   max err {err:.3g} (library {lib_err:.3g}, |ref| max {scale:.3g})",
          flush=True)
