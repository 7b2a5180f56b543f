#!/usr/bin/env python3
"""SiluAndMul micro-benchmark: the one measurement the reference publishes numbers for.

Shapes, precision (fp32) and method are the reference's (src/myvllm/layers/activation.py:20-37,
HowToApproachvLLM.md:18-42): 10 warm-up calls, then 100 calls each bracketed by a device
synchronise and timed with ``time.time()``; the mean is reported.  The published values were
taken on one NVIDIA A6000 with the op under ``torch.compile``; here the op is the hand-written
gfx950 kernel (``csrc/elementwise.hip``).  bf16 rows are measured as well because that is what
the engine feeds it.

    python benchmark_activation.py [--json out.json]
"""
from __future__ import annotations

import argparse
import json
import time

import torch

from minivllm_amd.layers.activation import SiluAndMul

# ms, best of (torch.compile, eager) as published by the reference for 1x A6000, fp32
PUBLISHED_A6000_MS = {(400, 800): 0.0823, (4000, 8000): 0.4494, (8, 4000, 8000): 2.3865}


def time_layer(layer, x, warmup=10, iters=100):
    for _ in range(warmup):
        layer(x)
    times = []
    for _ in range(iters):
        torch.cuda.synchronize()
        t0 = time.time()
        layer(x)
        torch.cuda.synchronize()
        times.append(time.time() - t0)
    return sum(times) / len(times) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("benchmark_activation.py needs a GPU")
    layer = SiluAndMul()
    rows = []
    for shape, ref_ms in PUBLISHED_A6000_MS.items():
        for dtype in (torch.float32, torch.bfloat16):
            x = torch.randn(*shape, device="cuda", dtype=dtype)
            ms = time_layer(layer, x)
            nbytes = x.numel() * x.element_size() * 1.5          # read 2 halves, write 1
            row = {"shape": list(shape), "dtype": str(dtype).replace("torch.", ""),
                   "ms": round(ms, 4), "TB_per_s": round(nbytes / ms / 1e9, 2)}
            if dtype == torch.float32:
                row["published_a6000_ms"] = ref_ms
                row["speedup_vs_published"] = round(ref_ms / ms, 2)
            rows.append(row)
            print(f"SiluAndMul {str(tuple(shape)):18s} {row['dtype']:8s} {ms:8.4f} ms  "
                  f"{row['TB_per_s']:5.2f} TB/s"
                  + (f"   reference (A6000, fp32): {ref_ms:.4f} ms -> x{ref_ms / ms:.1f}"
                     if dtype == torch.float32 else ""), flush=True)
            del x
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
