"""Flash prefill at 16 x 1024 (32q/8kv): packed vs paged K/V, and how much the data (normal, uniform, zeros) moves the time."""
import sys, math, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minivllm_amd.ops import hip
B, S, hq, hkv, d, bs = 16, 1024, 32, 8, 128, 256
scale = 1 / math.sqrt(d)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for dist in ("randn", "uniform", "zeros"):
    gen = {"randn": lambda *s: torch.randn(*s, device="cuda", dtype=torch.bfloat16),
           "uniform": lambda *s: (torch.rand(*s, device="cuda") * 2 - 1).to(torch.bfloat16),
           "zeros": lambda *s: torch.zeros(*s, device="cuda", dtype=torch.bfloat16)}[dist]
    q = gen(B * S, hq, d); k = gen(B * S, hkv, d); v = gen(B * S, hkv, d)
    cu = torch.arange(0, (B + 1) * S, S, device="cuda", dtype=torch.int32)
    nblk = S // bs
    kc = k.view(B * nblk, bs, hkv, d).permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
    vc = v.view(B * nblk, bs, hkv, d).permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
    bt = torch.arange(B * nblk, device="cuda", dtype=torch.int32).view(B, nblk)
    tp = timeit(lambda: hip.flash_attention_prefill(q, k, v, cu, scale, S))
    tg = timeit(lambda: hip.paged_attention_prefill(q, kc, vc, cu, cu, bt, scale, S))
    print(f"{dist:8s}: packed {tp:6.1f} us   paged {tg:6.1f} us", flush=True)
