#!/usr/bin/env python3
"""Offline serving throughput: N requests with random prompt / output lengths pushed through
``LLMEngine.generate`` at once (continuous batching, prefix cache, chunked prefill and
preemption all in play), the way a batch-inference job uses the engine.

    python benchmark_throughput.py --model qwen3-8b --requests 1024 --input-len 512 --output-len 128

Unlike ``bench.py`` (which times K pure prefill steps and K pure decode steps) this measures the
whole run, wall clock from the first ``add_prompt`` to the last finished sequence, and reports
requests/s, total tokens/s and generated tokens/s together with the engine's own per-phase
counters (reference metric definitions: src/myvllm/engine/llm_engine.py:48-60,76-83).
Random-init weights and synthetic prompts; ``--shared-prefix`` prepends a common system prompt
to every request so that prefix caching has something to hit.
"""
from __future__ import annotations

import argparse
import json
import time

import numpy as np
import torch

from minivllm_amd.config import EngineConfig, model_preset
from minivllm_amd.engine.llm_engine import LLMEngine
from minivllm_amd.sampling_parameters import SamplingParams


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-8b")
    ap.add_argument("--requests", type=int, default=1024)
    ap.add_argument("--input-len", type=int, default=512, help="mean prompt length")
    ap.add_argument("--output-len", type=int, default=128, help="mean generation length")
    ap.add_argument("--jitter", type=float, default=0.5,
                    help="lengths are uniform in mean * [1 - jitter, 1 + jitter]")
    ap.add_argument("--shared-prefix", type=int, default=0, help="tokens of common prefix")
    ap.add_argument("--max-seqs", type=int, default=256)
    ap.add_argument("--max-batched-tokens", type=int, default=16384)
    ap.add_argument("--block-size", type=int, default=256)
    ap.add_argument("--kv-cache-dtype", choices=["auto", "fp8"], default="auto")
    ap.add_argument("--min-prefill-tokens", type=int, default=0,
                    help="defer prefill steps that would carry fewer prompt tokens (0 = never)")
    ap.add_argument("--mixed", action="store_true",
                    help="running sequences take a decode token inside every prefill step")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()

    rng = np.random.default_rng(a.seed)
    model = model_preset(a.model)

    def lengths(mean):
        lo, hi = max(1, int(mean * (1 - a.jitter))), max(2, int(mean * (1 + a.jitter)) + 1)
        return rng.integers(lo, hi, size=a.requests)

    in_lens, out_lens = lengths(a.input_len), lengths(a.output_len)
    prefix = rng.integers(0, model.vocab_size, size=a.shared_prefix).tolist()
    prompts = [prefix + rng.integers(0, model.vocab_size, size=int(n)).tolist() for n in in_lens]
    params = [SamplingParams(temperature=0.8, max_tokens=int(n), ignore_eos=True) for n in out_lens]
    longest = a.shared_prefix + int(in_lens.max()) + int(out_lens.max())
    on_gpu = a.device == "cuda"
    cfg = EngineConfig(model=model, max_num_sequences=a.max_seqs, kv_cache_dtype=a.kv_cache_dtype,
                       max_num_batched_tokens=a.max_batched_tokens, block_size=a.block_size,
                       eos=-1, min_prefill_tokens=a.min_prefill_tokens,
                       enable_mixed_batches=a.mixed,
                       enforce_eager=a.eager or not on_gpu, max_model_length=longest + 8,
                       dtype=torch.bfloat16 if on_gpu else torch.float32, device=a.device, seed=a.seed)
    eng = LLMEngine(cfg)
    t0 = time.perf_counter()
    out = eng.generate(prompts, params, verbose=False)
    if on_gpu:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert [len(t) for t in out["token_ids"]] == out_lens.tolist()
    n_in = int(sum(len(p) for p in prompts))
    n_out = int(out_lens.sum())
    st, bm = eng.stats.as_dict(), eng.scheduler.block_manager.stats()
    res = {"model": a.model, "requests": a.requests, "prompt_tokens": n_in, "generated_tokens": n_out,
           "wall_s": round(dt, 3), "requests_per_s": round(a.requests / dt, 2),
           "total_tok_s": round((n_in + n_out) / dt, 1), "generated_tok_s": round(n_out / dt, 1),
           "engine_prefill_tok_s": round(st["prefill_tok_s"], 1),
           "engine_decode_tok_s": round(st["decode_tok_s"], 1),
           "prefill_steps": st["prefill_steps"], "decode_steps": st["decode_steps"],
           "prefill_tokens_computed": st["prefill_tokens"],
           "prefill_graph_replays": eng.model_runner.prefill_graph_replays,
           "prefill_graphs": len(eng.model_runner.prefill_graphs),
           "preemptions": eng.scheduler.num_preemptions,
           "prefix_hit_rate": round(bm["prefix_hit_rate"], 3), "kv_blocks": bm["num_blocks"],
           "config": {"kv_cache_dtype": a.kv_cache_dtype, "input_len": a.input_len, "output_len": a.output_len, "jitter": a.jitter,
                      "shared_prefix": a.shared_prefix, "max_seqs": a.max_seqs,
                      "max_batched_tokens": a.max_batched_tokens, "block_size": a.block_size,
                      "min_prefill_tokens": a.min_prefill_tokens, "mixed_batches": a.mixed,
                      "dtype": str(cfg.dtype).replace("torch.", ""), "device": a.device}}
    eng.exit()
    print(json.dumps(res), flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
