#!/usr/bin/env python3
"""Prefill GEMM chains of one decoder layer at a row count: each projection on the library and on
the 256 x 256-tile kernel (plain / SwiGLU / residual epilogue / row-scaled), the norm kernel, and
the two chains o -> norm -> gate_up and down -> norm -> qkv unfused against fused.

    python tools/prefill_chain_probe.py [--model qwen3-0.6b] [--rows 16384]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from minivllm_amd.config import model_preset
from minivllm_amd.ops import hip
from minivllm_amd.utils.gemm_tuning import enable_tuned_gemms

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="qwen3-0.6b")
ap.add_argument("--rows", type=int, default=16384)
a = ap.parse_args()
enable_tuned_gemms()
m = model_preset(a.model)
H, I = m.hidden_size, m.intermediate_size
shapes = {"qkv": ((m.num_heads + 2 * m.num_kv_heads) * m.head_dim, H), "o": (H, m.num_heads * m.head_dim),
          "gate_up": (2 * I, H), "down": (H, I)}
M = a.rows
dev, dt = "cuda", torch.bfloat16
W = {k: [(torch.randn(n, kk, device=dev) * 0.03).to(dt) for _ in range(4)] for k, (n, kk) in shapes.items()}
X = {k: torch.randn(M, kk, device=dev, dtype=dt) for k, (n, kk) in shapes.items()}
ones = torch.ones(H, device=dev, dtype=dt)
resid = torch.zeros(M, H, device=dev, dtype=dt)


def t(fn, x, ws, reps=2):
    return hip._graph_time_ms(fn, x, ws, reps) * 1e3


print(f"{a.model} rows {M}")
for k in shapes:
    sw = k == "gate_up"
    lib = (lambda xx, w: hip.silu_and_mul(F.linear(xx, w))) if sw else (lambda xx, w: F.linear(xx, w))
    row = {"lib": t(lib, X[k], W[k]), "big": t(lambda xx, w: hip.big_linear(xx, w, None, sw), X[k], W[k])}
    if k in ("qkv", "gate_up"):
        row["big+rowscale"] = t(lambda xx, w: hip.big_linear(xx, w, None, sw, row_scale=True, eps=1e-6), X[k], W[k])
    else:
        row["big+resid"] = t(lambda xx, w: hip.big_linear_resid(xx, w, None, resid), X[k], W[k])
    n, kk = shapes[k]
    fl = 2.0 * M * n * kk
    print(f"  {k:8s} [{n} x {kk}] " + "  ".join(f"{nm} {v:7.1f} us ({fl / v / 1e9:5.2f} PF/s)" for nm, v in row.items()), flush=True)
xx = torch.randn(M, H, device=dev, dtype=dt)
print(f"  add_rms_norm {t(lambda x_, w: hip.add_rms_norm(x_, resid, ones, 1e-6), xx, [None]):7.1f} us")
for prod, cons, sw in (("o", "gate_up", True), ("down", "qkv", False)):
    r = hip.tune_big_resid(W[prod], W[cons], sw, rows=[M])
    print(f"  chain {prod} -> norm -> {cons}: unfused {r[M][0]} us, fused {r[M][1]} us", flush=True)
