"""Race screen for csrc/big_gemm.hip: the kernel is deterministic (fixed accumulation order), so
any run whose output differs from the first run's in a single bit read a half-tile before it had
landed or overwrote one that was still being read.  Runs each shape many times, with and without a
second stream that keeps HBM and L2 busy (which moves the DMA latencies around), and compares."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivllm_amd.ops import hip

torch.manual_seed(0)
dev = "cuda"
REPS = int(os.environ.get("REPS", 150))
SHAPES = [(16384, 4096, 4096, False), (16384, 6144, 1024, True), (4160, 8192, 320, False),
          (16384, 1024, 3072, False), (8192, 24576, 4096, True), (1000, 5120, 8192, False)]
noise_src = torch.empty(1 << 28, dtype=torch.int32, device=dev)       # 1 GiB
noise_dst = torch.empty_like(noise_src)
side = torch.cuda.Stream()
bad = 0
for M, N, K, swi in SHAPES:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.03
    ref = hip.big_linear(x, w, None, swi).clone()
    torch.cuda.synchronize()
    for load in (False, True):
        mism = 0
        for r in range(REPS):
            if load:
                with torch.cuda.stream(side):
                    noise_dst.copy_(noise_src, non_blocking=True)
            out = hip.big_linear(x, w, None, swi)
            if not torch.equal(out, ref):
                mism += 1
        torch.cuda.synchronize()
        bad += mism
        print(f"M {M} N {N} K {K} swiglu {swi} concurrent copy {load}: {mism} of {REPS} runs differ", flush=True)
print("RACE SCREEN", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
