#!/usr/bin/env python3
"""Runs ONLY the two attention kernels on serving-sized shapes (for rocprofv3 passes and
quick timing): prefill B x S causal (paged, block 256) and decode B x ctx (paged, block 256),
Qwen3-8B heads (32/8/128), bf16.

    python tools/attn_kernel_bench.py [--mode prefill|decode|both] [--iters N]
"""
import argparse
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from minivllm_amd.ops import hip


def cache(nb, bs, hkv, d):
    return torch.randn(nb, hkv, bs, d, device="cuda", dtype=torch.bfloat16).permute(0, 2, 1, 3)


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="both")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--hq", type=int, default=32)
    ap.add_argument("--hkv", type=int, default=8)
    ap.add_argument("--prefill", default="16x1024,4x4096,1x16384,1x32768")
    ap.add_argument("--decode", default="256x1024,64x4096,8x16384,1x32768")
    ap.add_argument("--kv-fp8", action="store_true", help="fp8 (e4m3) KV cache")
    ap.add_argument("--work-list", action="store_true",
                    help="prefill: the engine's form (work list in device memory, sized for the "
                         "step budget)")
    ap.add_argument("--implicit-positions", action="store_true",
                    help="with --q-rope: no position array, the kernel uses k_len - q_len + i")
    ap.add_argument("--q-rope", action="store_true",
                    help="prefill: q-norm + RoPE of the Q rows inside the kernel (the engine's pure "
                         "prefill steps)")
    ap.add_argument("--splits", type=int, default=0, help="decode: context splits (0 = the library's rule)")
    ap.add_argument("--fused-store", action="store_true",
                    help="decode: the engine's form (raw q / k / v of the new token in; the kernel "
                         "norms, rotates and stores them itself)")
    a = ap.parse_args()
    eb = 1 if a.kv_fp8 else 2
    q8 = (lambda c: c.to(torch.float8_e4m3fn)) if a.kv_fp8 else (lambda c: c)
    hq, hkv, d, bs = a.hq, a.hkv, 128, 256
    scale = 1 / math.sqrt(d)
    if a.mode in ("prefill", "both"):
        for spec in a.prefill.split(","):
            B, S = map(int, spec.split("x"))
            nblk = (S + bs - 1) // bs
            kc, vc = q8(cache(B * nblk, bs, hkv, d)), q8(cache(B * nblk, bs, hkv, d))
            q = torch.randn(B * S, hq, d, device="cuda", dtype=torch.bfloat16)
            cu = torch.arange(0, (B + 1) * S, S, device="cuda", dtype=torch.int32)
            bt = torch.arange(B * nblk, device="cuda", dtype=torch.int32).view(B, nblk)
            items = q_rope = None
            if a.work_list:
                import numpy as np
                lens = np.full(B, S)
                items = torch.from_numpy(hip.prefill_work_items(
                    lens, lens, slots=hip.prefill_item_slots(B * S, max(B, 16)))).cuda()
            if a.q_rope:
                inv = 1.0 / (1e6 ** (torch.arange(0, d, 2, device="cuda").float() / d))
                ang = torch.arange(S, device="cuda").float()[:, None] * inv[None]
                q_rope = (None if a.implicit_positions else
                          torch.arange(S, device="cuda", dtype=torch.int32).repeat(B),
                          torch.cat([ang.cos(), ang.sin()], 1).contiguous(),
                          torch.rand(d, device="cuda", dtype=torch.bfloat16) + 0.5, 1e-6)
            t = timeit(lambda: hip.paged_attention_prefill(q, kc, vc, cu, cu, bt, scale, S,
                                                           q_rope=q_rope, work_items=items), a.iters)
            fl = 4 * hq * d * B * S * (S + 1) / 2
            print(f"prefill B={B} S={S}: {t*1e3:8.3f} ms  {fl/t/1e12:7.1f} TFLOP/s (causal)", flush=True)
            del kc, vc, q
    if a.mode in ("decode", "both"):
        for spec in a.decode.split(","):
            B, S = map(int, spec.split("x"))
            nblk = (S + bs - 1) // bs
            kc, vc = q8(cache(B * nblk, bs, hkv, d)), q8(cache(B * nblk, bs, hkv, d))
            q = torch.randn(B, hq, d, device="cuda", dtype=torch.bfloat16)
            ctx = torch.full((B,), S, device="cuda", dtype=torch.int32)
            bt = torch.arange(B * nblk, device="cuda", dtype=torch.int32).view(B, nblk)
            new_kv = None
            if a.fused_store:
                qkv = torch.randn(B, (hq + 2 * hkv) * d, device="cuda", dtype=torch.bfloat16)
                q = qkv[:, :hq * d].view(B, hq, d)
                inv = 1.0 / (1e6 ** (torch.arange(0, d, 2, device="cuda").float() / d))
                ang = torch.arange(S, device="cuda").float()[:, None] * inv[None]
                w = lambda: torch.rand(d, device="cuda", dtype=torch.bfloat16) + 0.5
                new_kv = (qkv[:, hq * d:(hq + hkv) * d].view(B, hkv, d), qkv[:, (hq + hkv) * d:].view(B, hkv, d),
                          torch.cat([ang.cos(), ang.sin()], 1).contiguous(), w(), w(), 1e-6)
            t = timeit(lambda: hip.paged_attention_decode(q, kc, vc, bt, ctx, scale, S, new_kv=new_kv,
                                                          num_splits=a.splits or None), a.iters)
            by = 2 * B * S * hkv * d * eb
            print(f"decode  B={B} ctx={S}: {t*1e3:8.3f} ms  {by/t/1e12:6.2f} TB/s KV stream", flush=True)
            del kc, vc, q


if __name__ == "__main__":
    main()
