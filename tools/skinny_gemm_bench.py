"""Weight-streaming GEMM kernel vs the library GEMM on the decode projections of Qwen3-8B
(graph-replayed, rotating weights so that every call streams from HBM)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivllm_amd.ops import hip

SHAPES = {
    "qwen3-8b": {"qkv": (6144, 4096), "o": (4096, 4096), "gate_up": (24576, 4096), "down": (4096, 12288)},
    "qwen3-14b": {"qkv": (7168, 5120), "o": (5120, 5120), "gate_up": (34816, 5120), "down": (5120, 17408)},
    "qwen3-32b": {"qkv": (10240, 5120), "o": (5120, 8192), "gate_up": (51200, 5120), "down": (5120, 25600)},
}
shapes = SHAPES[os.environ.get("MODEL", "qwen3-8b")]


def bench(fn, x, ws):
    for w in ws:
        fn(x, w)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(5):
            for w in ws:
                fn(x, w)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (25 * len(ws))


hip._lib.mvl_skinny_gemm_force_waves.restype = None
for M in (1, 8, 16):
    for name, (N, K) in shapes.items():
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        # enough copies that a pass over them cannot stay in the 256 MB Infinity Cache
        copies = max(4, -(-(640 << 20) // (N * K * 2)))
        ws = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) for _ in range(copies)]
        tl = bench(torch.nn.functional.linear, x, ws)
        line = f"M={M:2d} {name:8s} library {tl*1e6:6.1f} us ({N*K*2/tl/1e12:4.2f} TB/s)  kernel"
        for waves in (0, 16, 8, 4):
            hip._lib.mvl_skinny_gemm_force_waves(waves)
            tk = bench(hip.skinny_linear, x, ws)
            line += f"  w{waves or 'auto'} {tk*1e6:6.1f} us ({N*K*2/tk/1e12:4.2f})"
        if name == "gate_up":
            tlf = bench(lambda a, w: hip.silu_and_mul(torch.nn.functional.linear(a, w)), x, ws)
            line += f" | fused SwiGLU: library+act {tlf*1e6:6.1f} us, kernel"
            for waves in (0, 16, 8, 4):
                hip._lib.mvl_skinny_gemm_force_waves(waves)
                tf = bench(lambda a, w: hip.skinny_linear(a, w, swiglu=True), x, ws)
                line += f"  w{waves or 'auto'} {tf*1e6:6.1f}"
        hip._lib.mvl_skinny_gemm_force_waves(0)
        print(line, flush=True)
        del ws
