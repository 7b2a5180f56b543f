#!/usr/bin/env python3
"""Headline benchmark: engine-level prefill tok/s and decode tok/s of Qwen3 (random-init
weights, synthetic prompts, bf16) on N MI355X GPUs of one node.

    python bench.py                       # 1 GPU, Qwen3-8B flagship + Qwen3-0.6B secondary
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

What is timed is the real serving loop (``LLMEngine.step()``: schedule -> batch prep -> H2D ->
model -> sample -> D2H -> postprocess), not a kernel in isolation.  Metric definitions follow
the reference engine (src/myvllm/engine/llm_engine.py:48-60,76-83): a prefill step processes
sum(len(prompt)) tokens, a decode step one token per running sequence.

* decode : B sequences with S-token contexts; W untimed steps, then exactly K timed steps.
* prefill: steps of ``max_num_batched_tokens`` tokens (S-token prompts, generation length 1 so
  every engine step is a prefill step); W untimed, then exactly K timed steps.

Each timed region is bracketed by device synchronise + barrier on every rank and the elapsed
time is the MAX over ranks.  Rank 0 prints ONE JSON line; ``value`` is the whole-job decode
tok/s of the flagship model, the prefill number and the second model are extra keys.
With N > 1 the model is tensor-parallel over the N GPUs (RCCL over xGMI) and the global batch
is the same as with 1 GPU (strong scaling); that is the headline ``value``.  The same N GPUs are
then also measured as N independent single-GPU replicas (data parallel, weak scaling) and
reported under ``data_parallel_replicas``.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from minivllm_amd.config import EngineConfig, model_preset
from minivllm_amd.parallel.group import env_rank_world


def make_config(args, model_name: str, world: int) -> EngineConfig:
    model = model_preset(model_name)
    if args.layers:
        model.num_layers = args.layers          # debugging only; reported in the JSON
    port = int(os.environ.get("MASTER_PORT", args.port))
    return EngineConfig(
        model=model, max_num_sequences=args.batch, max_num_batched_tokens=args.prefill_tokens,
        block_size=args.block_size, eos=-1, world_size=world, enforce_eager=args.eager,
        max_model_length=max(args.seq_len + args.steps + args.warmup + 64, 2048),
        gpu_memory_utilization=0.85, dtype=torch.bfloat16, device="cuda", seed=0,
        master_addr=os.environ.get("MASTER_ADDR", "127.0.0.1"), master_port=port,
        # MINIVLLM_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal only)
        dist_backend=os.environ.get("MINIVLLM_DIST_BACKEND", "auto"),
        max_cached_blocks=args.kv_blocks).validate()


class _ReplicaTimer:
    """Timing bracket for data-parallel replicas: every rank drains its device, meets the
    others at a barrier of the job-wide process group, and the elapsed time is the MAX."""

    def call(self, what: str):
        import torch.distributed as dist
        torch.cuda.synchronize()
        dist.barrier()
        if what == "timer_start":
            self.t0 = time.perf_counter()
            return None
        t = torch.tensor([time.perf_counter() - self.t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())


def bench_model(args, model_name: str, rank: int, world: int, replicas: bool = False):
    """``replicas=False``: one tensor-parallel engine over all ranks (rank 0 drives).
    ``replicas=True``: every rank runs its own single-GPU engine on the same workload
    (data parallel); the returned numbers are per replica."""
    from minivllm_amd.engine.llm_engine import LLMEngine, run_worker
    from minivllm_amd.sampling_parameters import SamplingParams
    cfg = make_config(args, model_name, 1 if replicas else world)
    if rank != 0 and not replicas:
        run_worker(cfg, rank)                    # serves rank 0 until 'exit'
        return None
    t_init = time.perf_counter()
    eng = LLMEngine(cfg, spawn_workers=False)
    init_s = time.perf_counter() - t_init
    runner = _ReplicaTimer() if replicas else eng.model_runner
    rng = np.random.default_rng(0)
    vocab = cfg.model.vocab_size
    W, K, S, B = args.warmup, args.steps, args.seq_len, args.batch

    def prompts(n, length):
        return rng.integers(0, vocab, size=(n, length)).tolist()

    res = {"init_s": round(init_s, 1), "kv_blocks": eng.model_runner.num_kv_blocks}

    # ------------------------------------------------------------ prefill phase
    if args.mode in ("both", "prefill"):
        per_step = max(1, min(args.prefill_tokens // S, B))
        sp = SamplingParams(temperature=0.6, max_tokens=1, ignore_eos=True)
        for p in prompts(per_step * (W + K), S):
            eng.add_prompt(p, sp)
        for _ in range(W):
            _, n, is_prefill = eng.step()
            assert is_prefill and n == per_step * S
        runner.call("timer_start")
        tokens = 0
        for _ in range(K):
            _, n, is_prefill = eng.step()
            assert is_prefill
            tokens += n
        dt = runner.call("timer_stop")
        assert eng.is_finished()
        res.update(prefill_tok_s=tokens / dt, prefill_ms_per_step=dt / K * 1e3,
                   prefill_tokens_per_step=per_step * S)

    # ------------------------------------------------------------- decode phase
    if args.mode in ("both", "decode"):
        sp = SamplingParams(temperature=0.6, max_tokens=W + K + 1, ignore_eos=True)
        for p in prompts(B, S):
            eng.add_prompt(p, sp)
        while eng.scheduler.waiting:             # untimed: prefill all B prompts
            _, _, is_prefill = eng.step()
            assert is_prefill
        for _ in range(W):
            _, n, is_prefill = eng.step()
            assert not is_prefill and n == B
        runner.call("timer_start")
        tokens = 0
        for _ in range(K):
            _, n, is_prefill = eng.step()
            assert not is_prefill
            tokens += n
        dt = runner.call("timer_stop")
        res.update(decode_tok_s=tokens / dt, decode_ms_per_step=dt / K * 1e3)
        while not eng.is_finished():
            eng.step()
    eng.exit()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--model", default="qwen3-8b")
    ap.add_argument("--second-model", default="qwen3-0.6b",
                    help="also measured and reported under 'secondary' ('' to skip)")
    ap.add_argument("--batch", type=int, default=256, help="decode batch (sequences)")
    ap.add_argument("--seq-len", type=int, default=1024, help="prompt / context length")
    ap.add_argument("--prefill-tokens", type=int, default=16384, help="tokens per prefill step")
    ap.add_argument("--block-size", type=int, default=256)
    ap.add_argument("--mode", choices=["both", "prefill", "decode"], default="both")
    ap.add_argument("--eager", action="store_true", help="no hipGraph decode")
    ap.add_argument("--layers", type=int, default=0, help="override layer count (debug)")
    ap.add_argument("--kv-blocks", type=int, default=None)
    ap.add_argument("--port", type=int, default=29533)
    ap.add_argument("--no-dp", action="store_true",
                    help="with N > 1, skip the extra data-parallel (N replicas) measurement")
    args = ap.parse_args()

    if os.environ.get("MINIVLLM_HANG_DUMP"):     # debugging aid: dump all stacks if stuck
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MINIVLLM_HANG_DUMP"]), repeat=False)
    rank, local_rank, world = env_rank_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run "
                     "(one rank per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    os.environ["MINIVLLM_KEEP_PG"] = "1"     # two engines in a row share one process group

    main_res = bench_model(args, args.model, rank, world)
    second = None
    if args.second_model and args.second_model != args.model:
        second = bench_model(args, args.second_model, rank, world)
    dp = None
    if world > 1 and not args.no_dp:
        # same GPUs as N independent replicas (each serves its own batch of B sequences)
        dp = bench_model(args, args.model, rank, world, replicas=True)
    if rank != 0:
        return
    published = None
    try:
        # the only numbers the reference publishes: SiluAndMul fp32 timings (1x A6000)
        from benchmark_activation import PUBLISHED_A6000_MS, time_layer
        from minivllm_amd.layers.activation import SiluAndMul
        shape = (4000, 8000)
        ms = time_layer(SiluAndMul(), torch.randn(*shape, device="cuda", dtype=torch.float32))
        published = {"op": "SiluAndMul forward 4000x8000 fp32 (reference method)",
                     "ms": round(ms, 4), "reference_a6000_ms": PUBLISHED_A6000_MS[shape],
                     "speedup": round(PUBLISHED_A6000_MS[shape] / ms, 2)}
    except Exception as e:                       # never lose the headline over the side check
        published = {"error": repr(e)}
    value = main_res.get("decode_tok_s", main_res.get("prefill_tok_s"))
    ms = main_res.get("decode_ms_per_step", main_res.get("prefill_ms_per_step"))
    out = {
        "metric": f"decode tok/s ({args.model}, engine step); prefill tok/s in prefill_tok_s",
        "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic random token prompts, random-init weights",
        "config": {"model": args.model, "global_batch": args.batch, "seq_len": args.seq_len,
                   "parallelism": f"tp{world}", "prefill_tokens_per_step": args.prefill_tokens,
                   "kv_block_size": args.block_size, "hipgraph_decode": not args.eager,
                   **({"layers_override": args.layers} if args.layers else {})},
        **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in main_res.items()},
    }
    if dp is not None:
        out["data_parallel_replicas"] = {
            "parallelism": f"dp{world}", "scaling": "weak", "global_batch": args.batch * world,
            "decode_tok_s": round(dp.get("decode_tok_s", 0.0) * world, 1),
            "prefill_tok_s": round(dp.get("prefill_tok_s", 0.0) * world, 1),
            "decode_ms_per_step": round(dp.get("decode_ms_per_step", 0.0), 3)}
    out["published_reference_check"] = published
    if second is not None:
        out["secondary"] = {"model": args.second_model,
                            **{k: (round(v, 3) if isinstance(v, float) else v)
                               for k, v in second.items()}}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
