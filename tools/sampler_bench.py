"""Sampler timings: plain one-pass kernel, top-k / top-p kernel (no sort), sort-based PyTorch path.

    python tools/sampler_bench.py [--vocab 151936] [--batch 1 16 256] [--sigma 3.0]

Logits are N(0, sigma^2) in bf16: sigma ~3 has a few dominant tokens per row like a trained model,
sigma 0.02 is the near-flat row of a random-init model (worst case for the rejection rounds).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minivllm_amd.ops import hip                       # noqa: E402


def timed(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters * 1e3


def sort_path(logits, temps, k, p):
    lg = logits.float() / temps[:, None]
    vals, order = lg.sort(dim=-1, descending=True)
    rank_of = torch.arange(vals.shape[1], device=lg.device)[None]
    drop = (k[:, None] > 0) & (rank_of >= k[:, None])
    probs = torch.softmax(vals, dim=-1)
    drop |= (probs.cumsum(-1) - probs) >= p[:, None]
    vals = vals.masked_fill(drop, float("-inf"))
    u = torch.rand(vals.shape, device=lg.device).clamp_(1e-10, 1.0 - 1e-7)
    choice = (vals - torch.log(-torch.log(u))).argmax(dim=-1)
    return order.gather(1, choice[:, None]).squeeze(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vocab", type=int, default=151936)
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 16, 256])
    ap.add_argument("--sigma", type=float, nargs="+", default=[3.0, 0.02])
    a = ap.parse_args()
    dev = "cuda"
    for sigma in a.sigma:
        for B in a.batch:
            torch.manual_seed(B)
            logits = (torch.randn(B, a.vocab, device=dev) * sigma).to(torch.bfloat16)
            temps = torch.full((B,), 0.8, device=dev)
            rows = torch.arange(B, device=dev, dtype=torch.int32)
            out = torch.empty(B, device=dev, dtype=torch.int32)
            rounds = torch.zeros(B, device=dev, dtype=torch.int32)
            step = torch.zeros(1, device=dev, dtype=torch.int32)
            line = [f"sigma {sigma:<5} B {B:3d}: plain {timed(lambda: hip.sample_local(logits, temps, 1, step)):7.1f} us"]
            for k, p in ((50, 1.0), (0, 0.95), (50, 0.95)):
                tk = torch.full((B,), k, device=dev, dtype=torch.int32)
                tp = torch.full((B,), p, device=dev)
                t = timed(lambda: hip.sample_filtered(logits, rows, rows, temps, tk, tp, 1, step, out,
                                                      rounds=rounds))
                t_sort = timed(lambda: sort_path(logits, temps, tk, tp), iters=5)
                line.append(f"k={k} p={p}: kernel {t:7.1f} us (rounds mean {rounds.float().mean():.1f} "
                            f"max {int(rounds.max())}) sort {t_sort:8.1f} us")
            print("  ".join(line), flush=True)


if __name__ == "__main__":
    main()
