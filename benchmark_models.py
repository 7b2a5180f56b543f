#!/usr/bin/env python3
"""Attention kernels at large-model head shapes (same CLI, shapes and printed metrics as the
reference, reference: benchmark_models.py:10-43,46-179,182-224): flash prefill over batch x
sequence length and paged decode over batch x context length, for the head layouts of
Qwen3-8B, Qwen3-30B, Llama-3-70B and Qwen3-235B-MoE.

    python benchmark_models.py --config qwen3-8b --mode all [--quick] [--json out.json]

Prefill reports ms, M tok/s and TFLOPS = 2*T*S*Hq*D / t (the reference's formula, which counts
the full S x S rectangle); decode reports ms, tokens/s = batch / t, and additionally the
achieved KV-streaming bandwidth, which is the quantity the MI355X kernel is bound by.
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).parent))

#            name              (Hq, Hkv,  D, hidden, layers, description)
MODEL_CONFIGS = {
    name: dict(num_heads=hq, num_kv_heads=hkv, head_dim=d, hidden_size=hid, num_layers=nl,
               description=desc)
    for name, (hq, hkv, d, hid, nl, desc) in {
        "qwen3-8b": (32, 8, 128, 4096, 36, "Qwen3-8B (GQA 32/8)"),
        "qwen3-30b": (40, 8, 128, 5120, 48, "Qwen3-30B (GQA 40/8)"),
        "llama3-70b": (64, 8, 128, 8192, 80, "Llama-3-70B (GQA 64/8)"),
        "qwen3-235b-moe": (64, 4, 128, 8192, 94, "Qwen3-235B-MoE (GQA 64/4, 128 experts)"),
    }.items()
}


def _timeit(fn, warmup, iters):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def benchmark_prefill_at_scale(config_name, seq_lengths, batch_sizes):
    from benchmark_prefilling import flash_attention, setup_data
    c = MODEL_CONFIGS[config_name]
    hq, hkv, d = c["num_heads"], c["num_kv_heads"], c["head_dim"]
    print(f"\n{'=' * 70}\nPrefilling Benchmark: {c['description']}")
    print(f"num_heads={hq}, num_kv_heads={hkv}, head_dim={d}\n{'=' * 70}")
    out = []
    for bsz in batch_sizes:
        for S in seq_lengths:
            T = bsz * S
            print(f"\nbatch_size={bsz}, seq_len={S}, total_tokens={T}")
            q, k, v, cu, scale = setup_data(bsz, S, hq, hkv, d)
            iters = max(10, 100 // (S // 256 + 1))
            t = _timeit(lambda: flash_attention(q, k, v, cu, scale, hq, hkv, d), 3, iters)
            tflops = 2 * T * S * hq * d / t / 1e12
            print(f"  Flash Attention: {t * 1e3:.3f}ms | {T / t / 1e6:.2f}M tok/s | {tflops:.2f} TFLOPS")
            out.append(dict(config=config_name, batch_size=bsz, seq_len=S, time_ms=t * 1e3,
                            tokens_per_sec=T / t, tflops=tflops))
            del q, k, v
    return out


def benchmark_decode_at_scale(config_name, context_lengths, batch_sizes, block_size=16):
    from benchmark_decoding import paged_attention_decode_hip, setup_test_data
    c = MODEL_CONFIGS[config_name]
    hq, hkv, d = c["num_heads"], c["num_kv_heads"], c["head_dim"]
    print(f"\n{'=' * 70}\nDecoding Benchmark: {c['description']}\n{'=' * 70}")
    out = []
    for bsz in batch_sizes:
        for ctx in context_lengths:
            print(f"\nbatch_size={bsz}, context_len={ctx}")
            q, kc, vc, bt, cl, scale = setup_test_data(bsz, ctx, hq, hkv, d, block_size)
            t = _timeit(lambda: paged_attention_decode_hip(q, kc, vc, bt, cl, scale, hq, hkv, d,
                                                           block_size), 5, 100)
            bw = 2 * bsz * ctx * hkv * d * 2 / t / 1e12
            print(f"  Paged Attention: {t * 1e3:.3f}ms | {bsz / t:.0f} tokens/s | KV stream {bw:.2f} TB/s")
            out.append(dict(config=config_name, batch_size=bsz, context_len=ctx, time_ms=t * 1e3,
                            tokens_per_sec=bsz / t, kv_tb_s=bw))
            del q, kc, vc
    return out


def main():
    ap = argparse.ArgumentParser(description="Large Model Attention Benchmark")
    ap.add_argument("--config", default="qwen3-8b", choices=list(MODEL_CONFIGS) + ["all"],
                    help="head geometry to sweep ('all': every configuration of the table in turn)")
    ap.add_argument("--mode", default="all", choices=["prefill", "decode", "all"])
    ap.add_argument("--quick", action="store_true", help="fewer configurations")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    print("=" * 70 + "\nLarge Model Attention Benchmark\n" + "=" * 70)
    if not torch.cuda.is_available():
        print("WARNING: no GPU available, benchmarks need an MI355X")
        return
    props = torch.cuda.get_device_properties(0)
    print(f"GPU: {torch.cuda.get_device_name(0)}\nMemory: {props.total_memory / 1e9:.1f} GB")
    print(f"Arch: {getattr(props, 'gcnArchName', 'n/a')}, CUs: {props.multi_processor_count}")
    if args.quick:
        seqs, batches, ctxs = [512, 2048], [1, 4], [1024, 4096]
    else:
        seqs, batches = [512, 1024, 2048, 4096, 8192], [1, 2, 4, 8, 16]
        ctxs = [512, 1024, 2048, 4096, 8192, 16384]
    res = {}
    for config in (list(MODEL_CONFIGS) if args.config == "all" else [args.config]):
        r = res.setdefault(config, {})
        if args.mode in ("prefill", "all"):
            r["prefill"] = benchmark_prefill_at_scale(config, seqs, batches)
        if args.mode in ("decode", "all"):
            r["decode"] = benchmark_decode_at_scale(config, ctxs, batches)
    if args.config != "all":
        res = res[args.config]
    print("\n" + "=" * 70 + "\nBenchmark Complete\n" + "=" * 70)
    if args.json:
        Path(args.json).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
