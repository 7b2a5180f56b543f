#!/usr/bin/env python3
"""Build the TunableOp table for the serving GEMM shapes on this GPU.

    python tools/tune_gemms.py --models qwen3-8b,qwen3-0.6b --tp 1,2,4,8 \
        --out minivllm_amd/tuned/tunableop_gfx950.csv

For every (model, TP degree, M) it runs ``F.linear`` once with tuning enabled; M covers the
decode graph buckets and a few prefill token counts.  Rotating buffers make every timed call
stream its weights from HBM.  Progress goes to stdout (one line per shape).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.cuda.tunable as tunable

from minivllm_amd.config import model_preset
from minivllm_amd.engine.model_runner import default_graph_batch_sizes
from minivllm_amd.utils.gemm_tuning import model_gemm_shapes, write_table


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="qwen3-8b,qwen3-0.6b")
    ap.add_argument("--tp", default="1,2,4,8")
    ap.add_argument("--max-batch", type=int, default=256)
    ap.add_argument("--prefill-ms", default="2048,4096,8192,16384")
    ap.add_argument("--out", default="minivllm_amd/tuned/tunableop_gfx950.csv")
    ap.add_argument("--min-m", type=int, default=1)
    ap.add_argument("--base", default="minivllm_amd/tuned/tunableop_gfx950.csv",
                    help="existing table to extend")
    a = ap.parse_args()
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_rotating_buffer_size(1024)
    tunable.set_max_tuning_duration(30)
    tunable.set_max_tuning_iterations(30)
    tunable.set_filename(a.out + ".autosave")
    for prev in (a.base, a.out):
        if prev and os.path.exists(prev):
            tunable.read_file(prev)         # extend an existing table
    decode_ms = [m for m in default_graph_batch_sizes(a.max_batch) if m >= a.min_m]
    prefill_ms = [int(x) for x in a.prefill_ms.split(",") if x]
    seen = set()
    t0 = time.time()
    for name in a.models.split(","):
        model = model_preset(name)
        for tp in [int(x) for x in a.tp.split(",")]:
            if model.num_kv_heads % tp:
                continue
            for gname, N, K in model_gemm_shapes(model, tp):
                ms = decode_ms + ([] if gname == "lm_head" else prefill_ms)
                w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
                for M in ms:
                    if (M, N, K) in seen:
                        continue
                    seen.add((M, N, K))
                    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
                    torch.nn.functional.linear(x, w)
                    torch.cuda.synchronize()
                    print(f"[{time.time() - t0:7.1f}s]   {gname} M={M}", flush=True)
                print(f"[{time.time() - t0:7.1f}s] {name} tp{tp} {gname} N={N} K={K}: "
                      f"{len(ms)} M values", flush=True)
                del w
    n = write_table(a.out)
    print(f"wrote {a.out}: {n} entries ({len(seen)} shapes visited) in {time.time() - t0:.0f}s",
          flush=True)


if __name__ == "__main__":
    main()
