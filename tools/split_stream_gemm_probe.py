"""Would the K-deep decode projections (down: K = 12288) run faster as S concurrent GEMMs over K
slices on S streams (more workgroups in flight per CU) plus one add?  Library GEMMs only, shipped
TunableOp table loaded read-only (no tuning in this process), weights rotating so that every call
streams from HBM; everything inside one graph as it would be in the decode step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivllm_amd.utils.gemm_tuning import enable_tuned_gemms

print("tuned table:", enable_tuned_gemms(), flush=True)
SHAPES = {"down": (4096, 12288), "o": (4096, 4096), "qkv": (6144, 4096)}


def make(S, streams):
    def f(x, w, outs):
        K = x.shape[1]
        ks = K // S
        cur = torch.cuda.current_stream()
        ev0 = torch.cuda.Event()
        ev0.record(cur)
        evs = []
        for s in range(S):
            st = cur if s == 0 else streams[s - 1]
            if s:
                st.wait_event(ev0)
            with torch.cuda.stream(st):
                torch.mm(x[:, s * ks:(s + 1) * ks], w[:, s * ks:(s + 1) * ks].t(), out=outs[s])
                if s:
                    e = torch.cuda.Event()
                    e.record(st)
                    evs.append(e)
        for e in evs:
            cur.wait_event(e)
        acc = outs[0]
        for s in range(1, S):
            acc = acc.add_(outs[s])
        return acc
    return f


def bench(fn, x, ws, outs):
    for w in ws:
        fn(x, w, outs)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(3):
            for w in ws:
                fn(x, w, outs)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (15 * len(ws))


streams = [torch.cuda.Stream() for _ in range(3)]
for M in (128, 256):
    for name, (N, K) in SHAPES.items():
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        copies = max(4, -(-(640 << 20) // (N * K * 2)))
        ws = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) for _ in range(copies)]
        outs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(4)]
        base = bench(lambda a, w, o: torch.mm(a, w.t(), out=o[0]), x, ws, outs)
        line = f"M={M:3d} {name:5s} one GEMM {base*1e6:6.1f} us"
        for S in (2, 3, 4):
            if K % (S * 256):
                continue
            t = bench(make(S, streams), x, ws, outs)
            line += f" | {S} streams {t*1e6:6.1f} us"
        print(line, flush=True)
        del ws
