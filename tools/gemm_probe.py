"""Library GEMM time and weight bandwidth on the Qwen3-8B projections for M = 1..512 rows
(default hipBLASLt heuristics, no tuned table): the first look at where decode GEMMs stand."""
import torch, time
torch.manual_seed(0)
shapes = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 24576), "down": (12288, 4096)}
print("M", *shapes.keys(), "(us | TB/s weights)")
for M in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
    row = []
    for name, (K, N) in shapes.items():
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        ws = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) for _ in range(6)]  # defeat L2/MALL
        for w in ws: torch.nn.functional.linear(x, w)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            for w in ws: torch.nn.functional.linear(x, w)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 30
        row.append(f"{t*1e6:7.1f}|{N*K*2/t/1e12:4.2f}")
    print(M, *row)
