#!/usr/bin/env python3
"""Online serving latency: requests arrive over time (Poisson process at ``--rate`` requests/s)
at the background engine loop of ``minivllm_amd.serve.AsyncLLMEngine``; reports the achieved
throughput and the percentiles of time to first token (TTFT), time per output token after the
first (TPOT) and end-to-end latency.

    python benchmark_serving.py --model qwen3-8b --requests 512 --rate 32 --input-len 512 --output-len 128

``benchmark_throughput.py`` answers "how fast does a batch job finish"; this answers "what does
a user see at a given load".  ``--rate 0`` submits everything at once.  Random-init weights and
synthetic token prompts (the reference has no serving benchmark; its per-step print of
tokens/s, src/myvllm/engine/llm_engine.py:76-83, is the closest thing).
"""
from __future__ import annotations

import argparse
import json
import time

import numpy as np
import torch

from minivllm_amd.config import EngineConfig, model_preset
from minivllm_amd.sampling_parameters import SamplingParams
from minivllm_amd.serve import AsyncLLMEngine


def pct(xs, qs=(50, 90, 99)):
    xs = np.asarray(xs, dtype=np.float64)
    return {f"p{q}": round(float(np.percentile(xs, q)) * 1e3, 2) for q in qs} | {
        "mean": round(float(xs.mean()) * 1e3, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-8b")
    ap.add_argument("--requests", type=int, default=512)
    ap.add_argument("--rate", type=float, default=32.0, help="mean arrivals per second (0 = all at once)")
    ap.add_argument("--input-len", type=int, default=512)
    ap.add_argument("--output-len", type=int, default=128)
    ap.add_argument("--jitter", type=float, default=0.5)
    ap.add_argument("--max-seqs", type=int, default=256)
    ap.add_argument("--max-batched-tokens", type=int, default=16384)
    ap.add_argument("--block-size", type=int, default=256)
    ap.add_argument("--kv-cache-dtype", choices=["auto", "fp8"], default="auto")
    ap.add_argument("--min-prefill-tokens", type=int, default=0)
    ap.add_argument("--max-prefill-delay-ms", type=float, default=0.0,
                    help="hold a small prefill step this long so that arrivals share it")
    ap.add_argument("--mixed", action="store_true",
                    help="running sequences take a decode token inside every prefill step")
    ap.add_argument("--data-parallel-size", type=int, default=1,
                    help="engine replicas behind the front end (one GPU each; on a one-GPU box "
                         "they share it, for rehearsal)")
    ap.add_argument("--gpu-memory-utilization", type=float, default=0.9)
    ap.add_argument("--kv-blocks", type=int, default=None,
                    help="cap the KV blocks per engine (replicas sharing one GPU: equal shares)")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--warmup-requests", type=int, default=0,
                    help="requests served (at the same rate and lengths) before the measured ones: "
                         "step shapes seen for the first time capture their hipGraphs here")
    ap.add_argument("--no-lazy-prefill-graphs", action="store_true",
                    help="prefill steps beyond the start-up graphs launch eagerly (A/B)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()

    rng = np.random.default_rng(a.seed)
    model = model_preset(a.model)

    def lengths(mean):
        lo, hi = max(1, int(mean * (1 - a.jitter))), max(2, int(mean * (1 + a.jitter)) + 1)
        return rng.integers(lo, hi, size=a.requests)

    in_lens, out_lens = lengths(a.input_len), lengths(a.output_len)
    prompts = [rng.integers(0, model.vocab_size, size=int(n)).tolist() for n in in_lens]
    gaps = rng.exponential(1.0 / a.rate, size=a.requests) if a.rate > 0 else np.zeros(a.requests)
    arrivals = np.cumsum(gaps)
    on_gpu = a.device == "cuda"
    cfg = EngineConfig(model=model, max_num_sequences=a.max_seqs, kv_cache_dtype=a.kv_cache_dtype,
                       max_num_batched_tokens=a.max_batched_tokens, block_size=a.block_size,
                       eos=-1, min_prefill_tokens=a.min_prefill_tokens,
                       enable_mixed_batches=a.mixed,
                       max_prefill_delay_ms=a.max_prefill_delay_ms,
                       enforce_eager=a.eager or not on_gpu,
                       max_model_length=int(in_lens.max() + out_lens.max()) + 8,
                       dtype=torch.bfloat16 if on_gpu else torch.float32, device=a.device,
                       gpu_memory_utilization=a.gpu_memory_utilization,
                       max_cached_blocks=a.kv_blocks, seed=a.seed,
                       prefill_graph_lazy=not a.no_lazy_prefill_graphs)
    if a.data_parallel_size > 1:
        from minivllm_amd.serve import DataParallelAsyncEngine
        eng = DataParallelAsyncEngine(cfg, a.data_parallel_size)
    else:
        eng = AsyncLLMEngine(cfg)
    # one request through the whole path before the clock starts
    eng.generate(prompts[0][:16], SamplingParams(temperature=0.8, max_tokens=4, ignore_eos=True))

    if a.warmup_requests:
        wrng = np.random.default_rng(a.seed + 1)
        warm, tw = [], time.perf_counter()
        gaps = wrng.exponential(1.0 / a.rate, a.warmup_requests).cumsum() if a.rate > 0 else np.zeros(a.warmup_requests)
        for i in range(a.warmup_requests):
            wait = tw + gaps[i] - time.perf_counter()
            if wait > 0:
                time.sleep(wait)
            j = int(wrng.integers(0, a.requests))
            warm.append(eng.submit(prompts[j], SamplingParams(temperature=0.8, max_tokens=int(out_lens[j]),
                                                              ignore_eos=True)))
        for h in warm:
            h.result(timeout=1800)

    t0 = time.perf_counter()
    handles = []
    for i in range(a.requests):
        wait = t0 + arrivals[i] - time.perf_counter()
        if wait > 0:
            time.sleep(wait)
        handles.append(eng.submit(prompts[i], SamplingParams(temperature=0.8, max_tokens=int(out_lens[i]),
                                                             ignore_eos=True)))
    for h in handles:
        h.result(timeout=1800)
    wall = time.perf_counter() - t0
    st = eng.stats()
    eng.shutdown()

    assert [len(h.token_ids) for h in handles] == out_lens.tolist()
    ttft = [h.first_token_at - h.submitted_at for h in handles]
    e2e = [h.finished_at - h.submitted_at for h in handles]
    tpot = [(h.finished_at - h.first_token_at) / (len(h.token_ids) - 1)
            for h in handles if len(h.token_ids) > 1]
    n_in, n_out = int(in_lens.sum()), int(out_lens.sum())
    res = {"model": a.model, "requests": a.requests, "offered_rate_req_s": a.rate,
           "achieved_req_s": round(a.requests / wall, 2), "wall_s": round(wall, 2),
           "prompt_tokens": n_in, "generated_tokens": n_out,
           "total_tok_s": round((n_in + n_out) / wall, 1), "generated_tok_s": round(n_out / wall, 1),
           "ttft_ms": pct(ttft), "tpot_ms": pct(tpot), "e2e_ms": pct(e2e),
           "prefill_steps": st["prefill_steps"], "decode_steps": st["decode_steps"],
           "preemptions": st["preemptions"],
           "prefill_graphs": st.get("prefill_graphs"), "prefill_graph_replays": st.get("prefill_graph_replays"),
           "warmup_requests": a.warmup_requests,
           "kv_blocks": st["kv_blocks"], "kv_blocks_free_at_end": st["kv_blocks_free"],
           "config": {"kv_cache_dtype": a.kv_cache_dtype, "input_len": a.input_len, "output_len": a.output_len, "jitter": a.jitter,
                      "data_parallel_size": a.data_parallel_size,
                      "max_seqs": a.max_seqs, "max_batched_tokens": a.max_batched_tokens,
                      "min_prefill_tokens": a.min_prefill_tokens, "mixed_batches": a.mixed,
                      "max_prefill_delay_ms": a.max_prefill_delay_ms,
                      "dtype": str(cfg.dtype).replace("torch.", ""), "device": a.device}}
    print(json.dumps(res), flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
