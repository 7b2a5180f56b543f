#!/usr/bin/env python3
"""Flash prefill kernel on the K/V sources it can meet: the paged cache (block 256, a 64-token
tile is one contiguous 16 KiB piece per head), packed token-major K/V ([T, Hkv, D] rows 2 KiB
apart, or slices of the fused qkv projection, 12 KiB apart) and packed head-major K/V.

    python tools/flash_prefill_layout_probe.py [--shapes 16x1024,1x16384] [--hq 32]
"""
import argparse
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from minivllm_amd.ops import hip


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="16x1024,4x4096,1x16384")
    ap.add_argument("--hq", type=int, default=32)
    ap.add_argument("--hkv", type=int, default=8)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    hq, hkv, d, bs = a.hq, a.hkv, 128, 256
    scale = 1 / math.sqrt(d)
    dev, dt = "cuda", torch.bfloat16
    for spec in a.shapes.split(","):
        B, S = map(int, spec.split("x"))
        T = B * S
        fl = 4 * hq * d * B * S * (S + 1) / 2
        cu = torch.arange(0, (B + 1) * S, S, device=dev, dtype=torch.int32)
        q = torch.randn(T, hq, d, device=dev, dtype=dt)
        res = {}
        nblk = (S + bs - 1) // bs
        kc = torch.randn(B * nblk, hkv, bs, d, device=dev, dtype=dt).permute(0, 2, 1, 3)
        vc = torch.randn(B * nblk, hkv, bs, d, device=dev, dtype=dt).permute(0, 2, 1, 3)
        bt = torch.arange(B * nblk, device=dev, dtype=torch.int32).view(B, nblk)
        res["paged cache"] = timeit(lambda: hip.paged_attention_prefill(q, kc, vc, cu, cu, bt, scale, S), a.iters)
        k = torch.randn(T, hkv, d, device=dev, dtype=dt)
        v = torch.randn(T, hkv, d, device=dev, dtype=dt)
        res["packed [T,Hkv,D]"] = timeit(lambda: hip.flash_attention_prefill(q, k, v, cu, scale, S), a.iters)
        qkv = torch.randn(T, hq + 2 * hkv, d, device=dev, dtype=dt)
        qs, ks, vs = qkv[:, :hq], qkv[:, hq:hq + hkv], qkv[:, hq + hkv:]
        res["slices of fused qkv"] = timeit(lambda: hip.flash_attention_prefill(qs, ks, vs, cu, scale, S), a.iters)
        kh = torch.randn(hkv, T, d, device=dev, dtype=dt).permute(1, 0, 2)
        vh = torch.randn(hkv, T, d, device=dev, dtype=dt).permute(1, 0, 2)
        res["packed head-major"] = timeit(lambda: hip.flash_attention_prefill(q, kh, vh, cu, scale, S), a.iters)
        # the same head-major memory addressed through a block table (identity): separates the
        # paged code path from the placement of the blocks
        kp = kh.permute(1, 0, 2).as_strided((T // bs, bs, hkv, d), (bs * d, d, T * d, 1))
        vp = vh.permute(1, 0, 2).as_strided((T // bs, bs, hkv, d), (bs * d, d, T * d, 1))
        res["head-major, paged"] = timeit(lambda: hip.paged_attention_prefill(q, kp, vp, cu, cu, bt, scale, S), a.iters)
        for name, t in res.items():
            print(f"B={B} S={S} hq={hq} {name:22s} {t*1e3:8.3f} ms  {fl/t/1e12:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
