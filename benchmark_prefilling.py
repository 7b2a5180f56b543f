#!/usr/bin/env python3
"""Prefill-attention micro-benchmark with the reference script's configurations and printed
metrics (reference: benchmark_prefilling.py:294-502): O(N^2) PyTorch attention vs a
full-matrix "naive" fused attention vs flash attention, fp16, Hq=32 Hkv=8 D=128, plus the
crossover sweep and the launch-count arithmetic.

The flash kernel is the hand-written gfx950 MFMA kernel of ``minivllm_amd``; the "naive" arm is
a deliberately simple HIP kernel that materialises the whole score matrix of a sequence in LDS
(sequences of at most 128 tokens), the counterpart of the reference's naive Triton kernel.
Outputs are checked against the PyTorch oracle; ``--json FILE`` dumps the timings.
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).parent))

from minivllm_amd.ops import torch_ref  # noqa: E402


def pytorch_standard_attention(q, k, v, cu_seqlens, scale, num_heads, num_kv_heads, head_dim):
    """Standard attention that materialises the score matrix per sequence."""
    return torch_ref.flash_attention_prefill(q, k, v, cu_seqlens, scale)


def flash_attention(q, k, v, cu_seqlens, scale, num_heads, num_kv_heads, head_dim):
    from minivllm_amd.layers.attention import flash_attention_prefill
    return flash_attention_prefill(q, k, v, cu_seqlens, scale, num_heads, num_kv_heads, head_dim)


def naive_triton_attention(q, k, v, cu_seqlens, scale, num_heads, num_kv_heads, head_dim,
                           max_seq_len):
    """Full-matrix attention: one workgroup per (sequence, head) holds the whole score matrix in
    LDS (HIP kernel ``attn_naive.hip``; the reference's counterpart is its naive Triton kernel).
    Name kept from the reference script."""
    from minivllm_amd.ops import hip
    return hip.naive_attention(q, k, v, cu_seqlens, scale, max_seq_len)


naive_attention = naive_triton_attention


def setup_data(num_seqs, seq_len, num_heads, num_kv_heads, head_dim, dtype=torch.float16):
    T = num_seqs * seq_len
    q = torch.randn(T, num_heads, head_dim, device="cuda", dtype=dtype)
    k = torch.randn(T, num_kv_heads, head_dim, device="cuda", dtype=dtype)
    v = torch.randn(T, num_kv_heads, head_dim, device="cuda", dtype=dtype)
    cu = torch.tensor([i * seq_len for i in range(num_seqs + 1)], device="cuda", dtype=torch.int32)
    return q, k, v, cu, 1.0 / (head_dim ** 0.5)


def _time(fn, args, warmup, iters):
    for _ in range(warmup):
        out = fn(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = fn(*args)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters, out


def benchmark(num_seqs, seq_len, num_heads=32, num_kv_heads=8, head_dim=128, num_iter=50):
    print(f"\n{'=' * 80}")
    print(f"Benchmark: {num_seqs} seqs × {seq_len} tokens (total: {num_seqs * seq_len} tokens)")
    print(f"Heads: {num_heads}/{num_kv_heads}, Dim: {head_dim}")
    print(f"{'=' * 80}")
    q, k, v, cu, scale = setup_data(num_seqs, seq_len, num_heads, num_kv_heads, head_dim)
    args = (q, k, v, cu, scale, num_heads, num_kv_heads, head_dim)
    results, outputs = {}, {}
    print("\n[1/3] PyTorch Standard (O(N²) memory)...")
    t, outputs["pytorch"] = _time(pytorch_standard_attention, args, 5, num_iter)
    results["PyTorch (O(N²))"] = t
    print(f"      {t * 1000:.3f} ms")
    max_safe_seq = 128
    if seq_len <= max_safe_seq:
        print("\n[2/3] Naive HIP (O(N²), materializes full attention)...")
        t, outputs["naive"] = _time(naive_triton_attention, args + (seq_len,), 5, num_iter)
        results["Naive HIP (O(N²))"] = t
        print(f"      {t * 1000:.3f} ms")
    else:
        print(f"\n[2/3] Naive HIP: SKIPPED (seq_len={seq_len} > {max_safe_seq}, score matrix would not fit LDS)")
    print("\n[3/3] Flash Attention (O(N), online softmax, gfx950 MFMA)...")
    t, outputs["flash"] = _time(flash_attention, args, 5, num_iter)
    results["Flash Attention (O(N))"] = t
    flops = 2 * 2 * num_heads * head_dim * num_seqs * seq_len * (seq_len + 1) / 2
    print(f"      {t * 1000:.3f} ms   ({flops / t / 1e12:.1f} TFLOP/s causal)")
    err = (outputs["flash"].float() - outputs["pytorch"].float()).abs().max().item()
    print(f"      max |flash - pytorch| = {err:.2e}")
    assert err < 2e-2, "kernel disagrees with the PyTorch oracle"
    return results


def find_crossover_point():
    print("\n" + "=" * 80)
    print("FINDING CROSSOVER POINT: When does Flash beat Naive?")
    print("=" * 80)
    H, HKV, D, B = 32, 8, 128, 2
    rows = []
    for L in [16, 32, 48, 64, 80, 96, 112, 128, 192, 256, 512, 1024]:
        q, k, v, cu, scale = setup_data(B, L, H, HKV, D)
        args = (q, k, v, cu, scale, H, HKV, D)
        naive = _time(naive_triton_attention, args + (L,), 10, 50)[0] if L <= 128 else None
        flash, _ = _time(flash_attention, args, 10, 50)
        rows.append((L, naive, flash))
        if naive is None:
            print(f"  seq_len={L:5d}  Naive: SKIPPED | Flash: {flash * 1000:.3f}ms | Winner: Flash (by default)")
        else:
            w = "Flash" if flash < naive else "Naive"
            print(f"  seq_len={L:5d}  Naive: {naive * 1000:.3f}ms | Flash: {flash * 1000:.3f}ms | Winner: {w}")
    print(f"\n{'Seq Len':>10} | {'Naive (ms)':>12} | {'Flash (ms)':>12} | {'Winner':>10} | {'Speedup':>10}")
    print("-" * 80)
    for L, n, f in rows:
        if n is None:
            print(f"{L:>10} | {'n/a':>12} | {f * 1000:>12.3f} | {'Flash':>10} | {'N/A':>10}")
            continue
        w = "Flash" if f < n else "Naive"
        print(f"{L:>10} | {n * 1000:>12.3f} | {f * 1000:>12.3f} | {w:>10} | {max(n, f) / min(n, f):>9.2f}x")
    return rows


def analyze_kernel_launches():
    print("\n" + "=" * 80)
    print("KERNEL LAUNCH ANALYSIS")
    print("=" * 80)
    num_seqs, seq_len, num_heads, BLOCK_M = 2, 60, 32, 128
    tiles = (seq_len + BLOCK_M - 1) // BLOCK_M
    print(f"\nFor {num_seqs} sequences × {seq_len} tokens:")
    print(f"  Flash attention: ONE launch of {tiles * num_heads * num_seqs} workgroups "
          f"({tiles} Q tile(s) of {BLOCK_M} rows × {num_heads} heads × {num_seqs} seqs), 256 threads each")
    print("  The 1-D grid is remapped so that workgroups sharing an XCD work on one KV head")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--no-sweep", action="store_true")
    a = ap.parse_args()
    print("\n" + "=" * 80)
    print("PREFILL ATTENTION BENCHMARK")
    print("Comparing: PyTorch (O(N²)) | Naive fused (O(N²)) | Flash (O(N), gfx950)")
    print("=" * 80)
    allres = []
    for n, L, it in [(2, 60, 100), (4, 64, 100), (2, 1024, 30), (1, 4096, 10)]:
        r = benchmark(num_seqs=n, seq_len=L, num_iter=it)
        allres.append({"num_seqs": n, "seq_len": L, **{k: v * 1e3 for k, v in r.items()}})
    if not a.no_sweep:
        sweep = find_crossover_point()
        allres.append({"crossover": [{"seq_len": L, "naive_ms": None if n is None else n * 1e3,
                                      "flash_ms": f * 1e3} for L, n, f in sweep]})
    analyze_kernel_launches()
    if a.json:
        Path(a.json).write_text(json.dumps(allres, indent=1))
