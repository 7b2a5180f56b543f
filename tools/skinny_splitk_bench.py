"""Weight-streaming GEMM for <= 16 rows: library vs kernel vs kernel with K split over workgroups.

    python tools/skinny_gemm_bench.py [--model qwen3-8b] [--rows 1 8 16]

Per projection of the model (o and down together with the residual-add + RMSNorm kernel that
consumes them, which also sums the K slices; qkv with the separate reduce kernel), microseconds
per call over weights that rotate through > 640 MB (cold reads, as in a decode step), measured
on replays of a captured graph.
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minivllm_amd.config import model_preset          # noqa: E402
from minivllm_amd.ops import hip                       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-8b")
    ap.add_argument("--rows", type=int, nargs="+", default=[1, 8, 16])
    a = ap.parse_args()
    m = model_preset(a.model)
    H, D = m.hidden_size, m.head_dim
    shapes = {"qkv": ((m.num_heads + 2 * m.num_kv_heads) * D, H, False),
              "o": (H, m.num_heads * D, True),
              "down": (H, m.intermediate_size, True)}
    dev, dt = "cuda", torch.bfloat16
    for name, (n, k, to_norm) in shapes.items():
        count = max(2, -(-(640 << 20) // (n * k * 2)))
        weights = [(torch.randn(n, k, device=dev) * 0.02).to(dt) for _ in range(min(count, 16))]
        reps = max(1, -(-count // len(weights)))
        gamma = torch.ones(n, dtype=dt, device=dev)
        for rows in a.rows:
            x = torch.randn(rows, k, device=dev, dtype=dt)
            resid = torch.zeros(rows, n, device=dev, dtype=dt)

            def fin(y):
                if not to_norm:
                    return y.materialize() if isinstance(y, hip.SplitKPartials) else y
                if isinstance(y, hip.SplitKPartials):
                    return y.add_rms_norm(resid, gamma, 1e-6)
                return hip.add_rms_norm(y, resid, gamma, 1e-6)
            line = [f"{name:5s} {n}x{k} rows {rows:2d}:"]
            t = hip._graph_time_ms(lambda xx, w: fin(F.linear(xx, w)), x, weights, reps)
            line.append(f"lib {t * 1e3:6.1f}")
            for sp in (1, 2, 3, 4, 6, 8):
                if not hip.skinny_splits_ok(k, sp):
                    continue
                t = hip._graph_time_ms(
                    lambda xx, w: fin(hip.skinny_linear(xx, w, splits=sp, defer=True)), x, weights, reps)
                line.append(f"s{sp} {t * 1e3:6.1f}")
            print("  ".join(line), flush=True)
        del weights
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
