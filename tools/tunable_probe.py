"""Probe what PyTorch TunableOp (hipBLASLt / rocBLAS solution search) buys for the decode GEMM
shapes of Qwen3-8B.  Prints default vs tuned time per shape."""
import os, sys, time
import torch
import torch.cuda.tunable as tunable

shapes = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 24576), "down": (12288, 4096)}
Ms = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["256"])]


def bench(M, K, N):
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    ws = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) for _ in range(4)]
    g = torch.cuda.CUDAGraph()
    for w in ws:
        torch.nn.functional.linear(x, w)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(5):
            for w in ws:
                torch.nn.functional.linear(x, w)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 100


base = {(M, n): bench(M, *s) for M in Ms for n, s in shapes.items()}
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_max_tuning_duration(30)
tunable.set_max_tuning_iterations(100)
tunable.set_rotating_buffer_size(1024)
tunable.set_filename("gpurun_out/tunableop_probe.csv")
t0 = time.time()
for M in Ms:
    for n, (K, N) in shapes.items():
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
        torch.nn.functional.linear(x, w)
        torch.cuda.synchronize()
        print(f"tuned M={M} {n} at {time.time() - t0:.1f}s", flush=True)
tunable.tuning_enable(False)
tuned = {(M, n): bench(M, *s) for M in Ms for n, s in shapes.items()}
pass
for k in base:
    K, N = shapes[k[1]]
    print(f"M={k[0]:4d} {k[1]:8s} default {base[k]*1e6:7.1f} us ({N*K*2/base[k]/1e12:4.2f} TB/s)  "
          f"tuned {tuned[k]*1e6:7.1f} us ({N*K*2/tuned[k]/1e12:4.2f} TB/s)  x{base[k]/tuned[k]:.2f}")
