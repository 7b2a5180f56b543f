#!/usr/bin/env python3
"""Flash prefill over paged caches the uniform-base staging does not cover (blocks of 16 and 32
tokens: per-lane block-table lookups) and over the fp8 cache (tiles staged through registers),
beside the default block of 256 tokens.  Qwen3-8B heads, bf16 queries; the first case is measured
twice (clocks).

    python tools/flash_prefill_small_block_probe.py
"""
import math, sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from minivllm_amd.ops import hip
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters
hq, hkv, d = 32, 8, 128
for bs in (16, 16, 32, 256):
    for fp8 in ((False, True) if bs == 256 else (False,)):
        for B, S in ((16, 1024), (4, 4096)):
            nblk = (S + bs - 1) // bs
            kc = torch.randn(B * nblk, hkv, bs, d, device="cuda", dtype=torch.bfloat16).permute(0, 2, 1, 3)
            vc = torch.randn(B * nblk, hkv, bs, d, device="cuda", dtype=torch.bfloat16).permute(0, 2, 1, 3)
            if fp8: kc, vc = kc.to(torch.float8_e4m3fn), vc.to(torch.float8_e4m3fn)
            q = torch.randn(B * S, hq, d, device="cuda", dtype=torch.bfloat16)
            cu = torch.arange(0, (B + 1) * S, S, device="cuda", dtype=torch.int32)
            bt = torch.arange(B * nblk, device="cuda", dtype=torch.int32).view(B, nblk)
            t = timeit(lambda: hip.paged_attention_prefill(q, kc, vc, cu, cu, bt, 1 / math.sqrt(d), S))
            fl = 4 * hq * d * B * S * (S + 1) / 2
            print(f"block {bs} fp8={fp8} B={B} S={S}: {t*1e3:.3f} ms {fl/t/1e12:.1f} TFLOP/s", flush=True)
