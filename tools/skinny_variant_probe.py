"""Weight-streaming GEMM (<= 16 rows): x rows staged in LDS (+1 forced, 0 automatic, -1 never) vs the library,
Qwen3-8B projections, rotating cold weights, graph replay; d = largest difference to the first variant."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
import torch.nn.functional as F
from minivllm_amd.ops import hip
hip._lib.mvl_skinny_gemm_force_x_lds.restype = None
shapes = {"qkv": (6144, 4096, False), "o": (4096, 4096, False), "down": (4096, 12288, False), "gate_up": (24576, 4096, True)}
for name, (n, k, sw) in shapes.items():
    cnt = max(2, -(-(640 << 20) // (n * k * 2)))
    ws = [(torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16) for _ in range(min(cnt, 16))]
    reps = max(1, -(-cnt // len(ws)))
    for rows in (1, 2, 4, 8, 16):
        x = torch.randn(rows, k, device="cuda", dtype=torch.bfloat16)
        line = [f"{name:7s} rows {rows}: lib {hip._graph_time_ms(lambda a, w: (hip.silu_and_mul(F.linear(a, w)) if sw else F.linear(a, w)), x, ws, reps)*1e3:6.1f}"]
        ref = None
        for xl in (-1, 0, 1):
            hip._lib.mvl_skinny_gemm_force_x_lds(xl)
            t = hip._graph_time_ms(lambda a, w: hip.skinny_linear(a, w, swiglu=sw), x, ws, reps)
            y = hip.skinny_linear(x, ws[0], swiglu=sw).float()
            if ref is None: ref = y
            err = (y - ref).abs().max().item()
            line.append(f"xl{xl:+d} {t*1e3:6.1f} ({n*k*2/t/1e9:4.2f} TB/s, d {err:.1e})")
        hip._lib.mvl_skinny_gemm_force_x_lds(0)
        print("  ".join(line), flush=True)
    del ws; torch.cuda.empty_cache()
