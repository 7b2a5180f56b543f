#!/usr/bin/env python3
"""Decode step time of the engine (bench.py --mode decode) for Qwen3-8B and Qwen3-0.6B at batch 256
and batch 1, each in its own process.    python tools/decode_batch_sweep.py"""
import os, sys, json, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for model in ("qwen3-8b", "qwen3-0.6b"):
    for batch in (256, 1):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--model", model, "--second-model", "", "--mode", "decode", "--steps", "30", "--warmup", "5"]
        if batch != 256: cmd += ["--batch", str(batch)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=400)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(model, "batch", batch, "decode ms/step", d.get("decode_ms_per_step"), flush=True)
        except Exception as e:
            print("failed", model, batch, r.stderr[-500:]); sys.exit(1)
