#!/usr/bin/env python3
"""Paged decode attention at small batch and long context against the number of context splits
(the default rule gives min(32, 512 / (batch x KV heads)) splits).  Qwen3-8B heads by default, bf16,
block 256.

    python tools/decode_splits_probe.py [--hq 4 --hkv 1 --cases 256x1024,64x1024 --splits 1,2,3,4,6,8]
"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from minivllm_amd.ops import hip


def timeit(fn, iters=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--hq", type=int, default=32)
    ap.add_argument("--hkv", type=int, default=8)
    ap.add_argument("--cases", default="1x32768,1x32768,1x8192,2x16384,4x8192,8x16384,16x4096",
                    help="comma list of BATCHxCONTEXT")
    ap.add_argument("--splits", default="8,16,32,48,64")
    a = ap.parse_args()
    hq, hkv, d, bs = a.hq, a.hkv, 128, 256
    scale = 1 / math.sqrt(d)
    cases = [tuple(int(v) for v in c.split("x")) for c in a.cases.split(",")]
    extra = {int(v) for v in a.splits.split(",")}
    for B, ctx in cases:
        nblk = (ctx + bs - 1) // bs
        kc = torch.randn(B * nblk, hkv, bs, d, device="cuda", dtype=torch.bfloat16).permute(0, 2, 1, 3)
        vc = torch.randn(B * nblk, hkv, bs, d, device="cuda", dtype=torch.bfloat16).permute(0, 2, 1, 3)
        q = torch.randn(B, hq, d, device="cuda", dtype=torch.bfloat16)
        bt = torch.arange(B * nblk, device="cuda", dtype=torch.int32).view(B, nblk)
        cl = torch.full((B,), ctx, device="cuda", dtype=torch.int32)
        by = 2 * B * ctx * hkv * d * 2
        dflt = hip.decode_num_splits(B, hkv, ctx)
        row = []
        for ns in sorted({dflt} | extra):
            if ns > (ctx + 127) // 128:
                continue
            # replayed from a graph: at these sizes the launch path would dominate
            g = torch.cuda.CUDAGraph()
            out = torch.empty_like(q)
            hip.paged_attention_decode(q, kc, vc, bt, cl, scale, ctx, num_splits=ns, out=out)
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                for _ in range(8):
                    hip.paged_attention_decode(q, kc, vc, bt, cl, scale, ctx, num_splits=ns, out=out)
            t = timeit(g.replay) / 8
            row.append(f"{ns}{'*' if ns == dflt else ''}: {t*1e6:6.1f} us {by/t/1e12:4.2f} TB/s")
        print(f"B={B} ctx={ctx}  " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
