#!/usr/bin/env python3
"""Time of the sampling kernel against the number of workgroups a logits row is split over
(ops.hip.sample_local(chunks=...)), bf16 logits of a 151936-entry vocabulary.
    python tools/sampler_chunks_probe.py [--rows 256,64,16,1]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from minivllm_amd.ops import hip

ap = argparse.ArgumentParser()
ap.add_argument("--rows", default="256,64,16,1")
ap.add_argument("--vocab", type=int, default=151936)
a = ap.parse_args()
for B in map(int, a.rows.split(",")):
    logits = torch.randn(B, a.vocab, device="cuda", dtype=torch.bfloat16)
    for temp in (0.0, 0.8):
        t = torch.full((B,), temp, device="cuda")
        step = torch.zeros(1, dtype=torch.int32, device="cuda")
        row = []
        for chunks in (1, 2, 4, 8, 16):
            fn = lambda: hip.sample_local(logits, t, 1, step, chunks=chunks)
            for _ in range(3):
                fn()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            row.append(f"{chunks}: {e0.elapsed_time(e1) / 20 * 1e3:6.1f}")
        print(f"rows {B:4d} temperature {temp}: us per call by chunks  " + "  ".join(row), flush=True)
