#!/usr/bin/env python3
"""What splitting a row-parallel projection into token chunks (parallel/overlap.py pipelines chunk
i's collective under chunk i + 1's GEMM) costs in GEMM time on one GPU: the per-rank o / down
projections of Qwen3-8B at TP = 2, 4, 8, 16384 rows as one GEMM and as 2 / 4 / 8 chunks
(ops.linear: the same kernel choice as the engine, untuned tables = library GEMMs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from minivllm_amd import ops
from minivllm_amd.ops import hip
from minivllm_amd.utils.gemm_tuning import enable_tuned_gemms

enable_tuned_gemms()
M, H = 16384, 4096
for tp in (2, 4, 8):
    for name, k_full in (("o", 4096), ("down", 12288)):
        K = k_full // tp
        ws = [(torch.randn(H, K, device="cuda") * 0.03).to(torch.bfloat16) for _ in range(4)]
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        row = []
        for n in (1, 2, 4, 8):
            step = M // n

            def fn(xx, w, _n=n, _s=step):
                return [ops.linear(xx[i * _s:(i + 1) * _s], w) for i in range(_n)]
            row.append(f"{n}: {hip._graph_time_ms(fn, x, ws, 4) * 1e3:7.1f}")
        fl = 2.0 * M * H * K
        print(f"tp {tp} {name:5s} [4096 x {K:5d}] us by number of chunks  " + "  ".join(row)
              + f"   (one GEMM = {fl / 1e9 / float(row[0].split(':')[1]):.2f} PF/s)", flush=True)
