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
time is the MAX over ranks.  Rank 0 prints ONE JSON line.

With N > 1 the headline is the reference's own (and only) parallelism, tensor parallelism
(reference: src/myvllm/engine/model_runner.py:20-25, layers/linear.py:195-221):

* ``value`` / ``ms_per_step`` / ``prefill_tok_s``: ONE engine sharded over the N GPUs (RCCL
  all-reduce over xGMI inside the hipGraph-captured decode step), global batch B whatever N is
  (strong scaling).  The JSON records the process-group size the engine really saw, whether
  the decode graphs were captured and whether the peer-to-peer all-reduce kernel was active.
* ``tensor_parallel_p2p_allreduce``: the same with decode-sized all-reduces done by the
  peer-to-peer kernel of ``parallel/custom_allreduce.py`` instead of RCCL.
* ``qwen3_32b_tp8_32k_prefill`` (N = 8 only): Qwen3-32B sharded over the 8 GPUs, 32768-token
  prompts - the long-context configuration.
* ``data_parallel_replicas``: a side result, NOT the reference's parallelism - N single-GPU
  replicas each serving its own batch of B sequences (weak scaling, global batch N x B).

Under a launcher each rank's bench.py process is only an orchestrator: every phase runs in a
child process of its own (fresh HIP context, fresh rendezvous store, a time limit), so a phase
that fails or hangs is reported in the JSON instead of taking the other phase's result with it.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from minivllm_amd.config import EngineConfig, model_preset
from minivllm_amd.parallel.group import env_rank_world

PHASE_PORT_OFFSET = {"dp": 101, "tp": 202, "tpc": 303, "l32k": 404}
LONG_CONTEXT = dict(model="qwen3-32b", seq_len=32768, prefill_tokens=32768, batch=8)


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
        max_cached_blocks=args.kv_blocks, kv_cache_dtype=args.kv_cache_dtype,
        tp_shard=args.tp_shard if world == 1 else 0,
        # every step of a phase has the same shape: capture it at its first (warm-up) occurrence
        prefill_graph_capture_after=1,
        custom_all_reduce=bool(args.tp_shard and args.p2p)).validate()


def _condense_smi(doc):
    """Socket power and, per clock domain, (min, max) over the instances the tool lists."""
    found = {}

    def walk(node, path):
        if isinstance(node, dict):
            if "value" in node and isinstance(node["value"], (int, float)):
                key = path.rstrip("_0123456789") if path[-1:].isdigit() else path
                found.setdefault(key + (f" [{node.get('unit')}]" if node.get("unit") else ""), []).append(node["value"])
                return
            for k, v in node.items():
                if k in ("min_clk", "max_clk"):
                    continue
                walk(v, k if k not in ("clk",) else path)
        elif isinstance(node, list):
            for v in node:
                walk(v, path)
    try:
        walk(doc, "")
        return {k: (v[0] if len(v) == 1 else [min(v), max(v)]) for k, v in found.items()}
    except Exception:
        return str(doc)[:600]


def gpu_state() -> dict:
    """Clock / power readings of GPU 0 as the SMI tools report them, so that a slow number can be
    told from a slow box (boxes of one pool differ by 5-15 %).  Best effort: absent tools or
    fields simply leave entries out."""
    import shutil
    out = {}
    try:
        out["device"] = torch.cuda.get_device_name(0)
        out["cu_count"] = torch.cuda.get_device_properties(0).multi_processor_count
    except Exception:
        pass
    for tool, argv in (("amd-smi", ["amd-smi", "metric", "-g", "0", "--clock", "--power", "--json"]),
                       ("rocm-smi", ["rocm-smi", "-d", "0", "--showclocks", "--showpower",
                                     "--showmaxpower", "--showperflevel", "--json"])):
        if not shutil.which(argv[0]):
            continue
        try:
            r = subprocess.run(argv, capture_output=True, text=True, timeout=20)
            if r.returncode == 0 and r.stdout.strip():
                try:
                    out[tool] = _condense_smi(json.loads(r.stdout))
                except ValueError:
                    out[tool] = r.stdout.strip()[:600]
                break
        except Exception:
            continue
    return out


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
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([time.perf_counter() - self.t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())


class _LocalTimer:
    """Single replica, no process group: drain the device around the timed region."""

    def call(self, what: str):
        torch.cuda.synchronize()
        if what == "timer_start":
            self.t0 = time.perf_counter()
            return None
        return time.perf_counter() - self.t0


def bench_model(args, model_name: str, rank: int, world: int, replicas: bool = False):
    """``replicas=False``: one tensor-parallel engine over all ranks (rank 0 drives).
    ``replicas=True``: every rank runs its own single-GPU engine on its own prompts
    (data parallel); the returned numbers are per replica."""
    import torch.distributed as dist
    from minivllm_amd.engine.llm_engine import LLMEngine, run_worker
    from minivllm_amd.sampling_parameters import SamplingParams
    cfg = make_config(args, model_name, 1 if replicas else world)
    if rank != 0 and not replicas:
        run_worker(cfg, rank)                    # serves rank 0 until 'exit'
        return None
    t_init = time.perf_counter()
    eng = LLMEngine(cfg, spawn_workers=False)
    init_s = time.perf_counter() - t_init
    if (os.environ.get("MINIVLLM_BENCH_INJECT_TP_FAILURE") == "1" and world > 1 and not replicas
            and os.environ.get("MINIVLLM_TP_CONSERVATIVE") != "1"):
        raise RuntimeError("injected failure of the first tensor-parallel attempt (test hook)")
    if replicas:
        runner = _ReplicaTimer() if dist.is_initialized() else _LocalTimer()
    else:
        runner = eng.model_runner
    rng = np.random.default_rng(rank if replicas else 0)     # every replica has its own prompts
    vocab = cfg.model.vocab_size
    W, K, S, B = args.warmup, args.steps, args.seq_len, args.batch

    def prompts(n, length):
        return rng.integers(0, vocab, size=(n, length)).tolist()

    res = {"init_s": round(init_s, 1), "kv_blocks": eng.model_runner.num_kv_blocks,
           "hipgraph_captured": bool(eng.model_runner.graphs),
           "gemm_choice_source": getattr(eng.model_runner, "gemm_choice_source", None),
           "gemm_choices": getattr(eng.model_runner, "gemm_choices", None)}
    if cfg.tp_shard:
        from minivllm_amd.parallel import group as pg
        res["p2p_allreduce"] = pg.custom_all_reduce() is not None
    if not replicas:
        from minivllm_amd.parallel import group as pg
        res["tp_world_size_seen"] = (dist.get_world_size(pg.tp_group())
                                     if dist.is_initialized() and world > 1 else 1)
        res["dist_backend"] = dist.get_backend(pg.tp_group()) if dist.is_initialized() and world > 1 else None
        res["p2p_allreduce"] = pg.custom_all_reduce() is not None

    # ------------------------------------------------------------ prefill phase
    if args.mode in ("both", "prefill"):
        per_step = max(1, min(args.prefill_tokens // S, B))
        sp = SamplingParams(temperature=0.6, max_tokens=1, ignore_eos=True)
        # Consecutive prefill steps are launched one ahead: a step is in flight when the window
        # opens (timer_start drains it) and the last timed call launches one more, which timer_stop
        # drains, so the window holds exactly K steps of GPU work.  The look-ahead starts with the
        # step after the one that captures the shape's graph: if the warm-up was too short for
        # that (W = 1), up to two more untimed steps bring the pipeline into its steady state -
        # a window that opens with nothing in flight but closes on a launched-ahead step would
        # hold K + 1 steps of work.
        extra = 2
        for p in prompts(per_step * (W + extra + K + 1), S):
            eng.add_prompt(p, sp)
        for _ in range(W):
            _, n, is_prefill = eng.step()
            assert is_prefill
        mr = eng.model_runner
        while extra and cfg.prefill_prelaunch and not mr.prefill_ahead_in_flight():
            _, n, is_prefill = eng.step()
            assert is_prefill
            extra -= 1
        res["prefill_steps_launched_ahead"] = mr.prefill_ahead_in_flight()
        runner.call("timer_start")
        tokens = 0
        for _ in range(K):
            _, n, is_prefill = eng.step()
            assert is_prefill
            tokens += n
        dt = runner.call("timer_stop")
        while not eng.is_finished():             # a budget that the prompt length does not divide
            eng.step()
        res.update(prefill_tok_s=tokens / dt, prefill_ms_per_step=dt / K * 1e3,
                   prefill_tokens_per_step=tokens // K)

    # ------------------------------------------------------------- decode phase
    if args.mode in ("both", "decode"):
        # the sequences outlive the timed window by a few tokens: with launch-ahead decode the step
        # after the last timed one must exist, so that the window starts and ends in the same
        # pipeline state (one step in flight) and contains exactly K steps of GPU work
        sp = SamplingParams(temperature=0.6, max_tokens=W + K + 4, ignore_eos=True)
        for p in prompts(B, S):
            eng.add_prompt(p, sp)
        while (eng.scheduler.waiting or eng.scheduler.prefilling      # untimed: prefill all B prompts
               or eng.model_runner.has_prefill_ahead()):
            _, _, is_prefill = eng.step()
            assert is_prefill
        for _ in range(W):
            _, n, is_prefill = eng.step()
            assert not is_prefill and n == B
        # with launch-ahead decode one step is already in flight here and another one is when the
        # timed loop ends: timer_start / timer_stop drain the device, so the K timed steps contain
        # exactly K steps of GPU work (one carried in, one carried out)
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


def _round(d):
    return None if d is None else {k: (round(v, 3) if isinstance(v, float) else v)
                                   for k, v in d.items()}


# ------------------------------------------------------------------ phases (run in a child)
def run_phase(args, phase: str, rank: int, local_rank: int, world: int):
    """Body of one phase.  Returns {'main': ..., 'second': ...} on rank 0, None elsewhere."""
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    os.environ["MINIVLLM_KEEP_PG"] = "1"         # two engines in a row share one process group
    if phase == "l32k":                          # Qwen3-32B, 32768-token prompts, few steps
        args = argparse.Namespace(**vars(args))
        for k, v in LONG_CONTEXT.items():
            setattr(args, k, v)
        args.steps, args.warmup, args.second_model = min(args.steps, 4), 1, ""
    models = [args.model] + ([args.second_model] if args.second_model
                             and args.second_model != args.model and phase != "tpc" else [])
    if phase == "dp" and (world > 1 or os.environ.get("MINIVLLM_FORCE_PG") == "1"):
        import torch.distributed as dist
        backend = os.environ.get("MINIVLLM_DIST_BACKEND", "nccl")
        kw = ({"device_id": torch.device("cuda", torch.cuda.current_device())}
              if backend == "nccl" else {})
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, init_method=f"tcp://{addr}:{os.environ['MASTER_PORT']}",
                                rank=rank, world_size=world, **kw)
    out = [bench_model(args, m, rank, world, replicas=(phase not in ("tp", "tpc", "l32k")))
           for m in models]
    import torch.distributed as dist
    if dist.is_initialized():                    # kept alive across the engines of this phase
        try:
            dist.destroy_process_group()
        except Exception:
            pass
    if rank != 0:
        return None
    res = {"main": out[0], "second": out[1] if len(out) > 1 else None}
    if phase == "l32k":
        res["config"] = dict(LONG_CONTEXT, steps=args.steps, warmup=args.warmup)
    return res


def spawn_phase(args, phase: str, rank: int, local_rank: int, world: int, conservative: bool = False,
                port_shift: int = 0, limit: int = 0):
    """Run ``phase`` in a child process of this rank (the children of all ranks form the job of
    that phase).  Returns rank 0's result dict, or {'error': ...}.  ``conservative``: the engine's
    newer multi-rank paths off (MINIVLLM_TP_CONSERVATIVE, see ModelRunner).  ``limit``: seconds
    after which the child is stopped (0 = the phase's default)."""
    env = dict(os.environ)
    if conservative:
        env["MINIVLLM_TP_CONSERVATIVE"] = "1"
    for k in [k for k in env if k.startswith("TORCHELASTIC_")]:
        env.pop(k)                               # the child hosts its own rendezvous store
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", args.port))
                             + PHASE_PORT_OFFSET[phase] + port_shift)
    env.update(RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world),
               MINIVLLM_BENCH_PHASE=phase)
    if phase == "tpc":                           # TP again, all-reduce by the peer-to-peer kernel
        env["MINIVLLM_CUSTOM_AR"] = "1"
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    # the headline (tp) and the long-context phase get the full limit; side results less, so
    # that a stuck one does not hold the JSON line back long
    limit = limit or phase_limit(args, phase)
    t0 = time.time()
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    stdout = None
    while stdout is None:
        try:
            stdout, _ = proc.communicate(timeout=30)
        except subprocess.TimeoutExpired:
            waited = time.time() - t0
            if waited > limit:
                proc.kill()                      # exactly this rank's child, by pid
                proc.communicate()
                return {"error": f"{phase} phase exceeded {limit}s and was stopped"}
            if rank == 0:
                print(f"[bench] {phase} phase running for {waited:.0f}s", file=sys.stderr,
                      flush=True)
    if proc.returncode != 0:
        return {"error": f"{phase} phase child exited with code {proc.returncode}"}
    if rank != 0:
        return None
    for line in reversed(stdout.strip().splitlines()):
        if line.startswith("{"):
            res = json.loads(line)
            res["phase_wall_s"] = round(time.time() - t0, 1)
            return res
    return {"error": f"{phase} phase printed no result"}


def phase_limit(args, phase: str) -> int:
    return args.phase_timeout if phase in ("tp", "l32k") else min(args.phase_timeout, 360)


class JobClock:
    """Keeps an N > 1 run inside ``--total-timeout``: before each phase after the first, rank 0
    decides how long the phase may take (what is left of the budget, capped by the phase's own
    limit) or that it is skipped, and publishes the decision through the key-value store, so
    that the ranks never disagree about whether a phase runs.  Without a store (it could not be
    set up) every phase runs with its own limit."""

    MIN_USEFUL_S = 90                            # an engine start alone takes about that long

    def __init__(self, args, rank: int, world: int):
        self.args, self.rank, self.t0, self.store = args, rank, time.time(), None
        try:
            import datetime
            from torch.distributed import TCPStore
            self.store = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"),
                                  int(os.environ.get("MASTER_PORT", args.port)) + 555, world,
                                  rank == 0,
                                  timeout=datetime.timedelta(seconds=args.phase_timeout + 90))
        except Exception as e:
            print(f"[bench] rank {rank}: no key-value store between the ranks ({e!r}): no retry "
                  f"of a failed phase, no overall time limit", file=sys.stderr, flush=True)

    def agree(self, key: str, value) -> str:
        """Rank 0's ``value`` (a string), on every rank; None without a store."""
        if self.store is None:
            return None
        try:
            if self.rank == 0:
                self.store.set(key, str(value))
            return self.store.get(key).decode()
        except Exception as e:
            print(f"[bench] rank {self.rank}: store failed on {key!r} ({e!r})", file=sys.stderr,
                  flush=True)
            self.store = None
            return None

    def limit_for(self, phase: str, tag: str = "") -> int:
        """Seconds the phase may run (the same number on every rank); 0 = skip it."""
        own = phase_limit(self.args, phase)
        left = int(self.args.total_timeout - (time.time() - self.t0)) - 20
        got = self.agree(f"limit_{phase}{tag}", min(own, left) if left >= self.MIN_USEFUL_S else 0)
        return own if got is None else int(got)


SKIPPED = {"error": "skipped: the run's overall time limit (--total-timeout) was nearly used up"}

_RESULT_OUT = None


def claim_stdout() -> None:
    """The one JSON line goes to the process's original stdout; from here on file descriptor 1
    is stderr, so that nothing a library prints there (RCCL's version banner when its first
    communicator comes up) can end up next to the result."""
    global _RESULT_OUT
    sys.stdout.flush()
    _RESULT_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)


def emit(obj) -> None:
    out = _RESULT_OUT or sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


def published_reference_check():
    """The only numbers the reference publishes: SiluAndMul fp32 timings on one A6000."""
    try:
        from benchmark_activation import PUBLISHED_A6000_MS, time_layer
        from minivllm_amd.layers.activation import SiluAndMul
        shape = (4000, 8000)
        ms = time_layer(SiluAndMul(), torch.randn(*shape, device="cuda", dtype=torch.float32))
        return {"op": "SiluAndMul forward 4000x8000 fp32 (reference method)",
                "ms": round(ms, 4), "reference_a6000_ms": PUBLISHED_A6000_MS[shape],
                "speedup": round(PUBLISHED_A6000_MS[shape] / ms, 2)}
    except Exception as e:                       # never lose the headline over the side check
        return {"error": repr(e)}


def phase_ok(r) -> bool:
    return bool(r is not None and "error" not in r and r.get("main"))


def headline(m):
    return (round(m.get("decode_tok_s", m.get("prefill_tok_s")), 1),
            round(m.get("decode_ms_per_step", m.get("prefill_ms_per_step")), 3))


def assemble_multi_gpu(args, world, dp, tp, tpc, metric, base, cfg_common, long_ctx=None):
    """The one JSON object of an N > 1 run from the phase results (each a dict from
    ``spawn_phase``, ``{'error': ...}`` or None when the phase was skipped).  Returns
    (object, whether any phase delivered a number).  Headline = ONE tensor-parallel engine over
    the N GPUs at the global batch of the 1-GPU run (strong scaling).  Only if that phase
    failed does the data-parallel side result stand in, marked as a fallback; if both failed,
    value 0 with the errors."""
    B = args.batch

    def scaled(m):                               # per-replica numbers -> whole job
        d = dict(m)
        for k in ("decode_tok_s", "prefill_tok_s", "prefill_tokens_per_step"):
            if k in d:
                d[k] = d[k] * world
        return d

    def tp_section(r):
        if not phase_ok(r):
            return r
        sec = {"parallelism": f"tp{world}", "scaling": "strong", "global_batch": B,
               **_round(r["main"]), "phase_wall_s": r.get("phase_wall_s")}
        if r.get("second"):
            sec["secondary"] = {"model": args.second_model, **_round(r["second"])}
        if r.get("config"):
            sec.update(r["config"])
            sec["global_batch"] = r["config"].get("batch", B)
        return sec

    def dp_section(r):
        if not phase_ok(r):
            return r
        sec = {"note": "side result, not the reference's parallelism: N single-GPU replicas, "
                       "each with its own batch", "parallelism": f"dp{world}", "scaling": "weak",
               "global_batch": B * world, **_round(scaled(r["main"])),
               "phase_wall_s": r.get("phase_wall_s")}
        if r.get("second"):
            sec["secondary"] = {"model": args.second_model, **_round(scaled(r["second"]))}
        return sec

    ref_check = (dp or {}).get("published_reference_check") or (tp or {}).get("published_reference_check")
    # The same TP engine is measured with both all-reduce data planes (RCCL; peer-to-peer
    # kernels over xGMI, which check themselves at start-up and poison the step on a timeout):
    # the headline is the faster one that completed, and says which it is.
    best, plane = tp, "rccl"
    if phase_ok(tpc) and (not phase_ok(tp) or headline(tpc["main"])[0] > headline(tp["main"])[0]):
        best, plane = tpc, "p2p"
    if phase_ok(best):
        value, ms = headline(best["main"])
        out = {"metric": metric, "value": value, **base, "ms_per_step": ms, "scaling": "strong",
               "config": {**cfg_common, "global_batch": B, "parallelism": f"tp{world}",
                          "all_reduce": plane},
               **_round(best["main"]), "phase_wall_s": best.get("phase_wall_s"),
               "published_reference_check": ref_check}
        if best.get("second"):
            out["secondary"] = {"model": args.second_model, **_round(best["second"])}
        if plane == "p2p":
            out["tensor_parallel_rccl_allreduce"] = tp_section(tp)
        if dp is not None:
            out["data_parallel_replicas"] = dp_section(dp)
    elif phase_ok(dp):
        m = scaled(dp["main"])
        value, ms = headline(m)
        out = {"metric": metric, "value": value, **base, "ms_per_step": ms, "scaling": "weak",
               "headline_fallback": "the tensor-parallel phase failed: value is N data-parallel "
                                    "single-GPU replicas (weak scaling), not the TP metric",
               "config": {**cfg_common, "global_batch": B * world, "parallelism": f"dp{world}",
                          "replica": "tp1, one engine per GPU"},
               **_round(m), "phase_wall_s": dp.get("phase_wall_s"),
               "published_reference_check": ref_check, "tensor_parallel": tp}
        if dp.get("second"):
            out["secondary"] = {"model": args.second_model, **_round(scaled(dp["second"]))}
    else:
        out = {"metric": metric, "value": 0.0, **base, "ms_per_step": None, "scaling": "strong",
               "config": {**cfg_common, "global_batch": B, "parallelism": f"tp{world}"},
               "tensor_parallel": tp, "data_parallel_replicas": dp}
    if tpc is not None and not (phase_ok(best) and plane == "p2p"):
        out["tensor_parallel_p2p_allreduce"] = tp_section(tpc)
    if long_ctx is not None:
        out["qwen3_32b_tp8_32k_prefill"] = tp_section(long_ctx)
    return out, phase_ok(dp) or phase_ok(tp) or phase_ok(tpc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--model", default="qwen3-8b")
    ap.add_argument("--second-model", default="qwen3-0.6b",
                    help="also measured and reported under 'secondary' ('' to skip)")
    ap.add_argument("--batch", type=int, default=256, help="decode batch (sequences) per engine")
    ap.add_argument("--seq-len", type=int, default=1024, help="prompt / context length")
    ap.add_argument("--prefill-tokens", type=int, default=16384, help="tokens per prefill step")
    ap.add_argument("--block-size", type=int, default=256)
    ap.add_argument("--mode", choices=["both", "prefill", "decode"], default="both")
    ap.add_argument("--eager", action="store_true", help="no hipGraph decode")
    ap.add_argument("--layers", type=int, default=0, help="override layer count (debug)")
    ap.add_argument("--kv-blocks", type=int, default=None)
    ap.add_argument("--kv-cache-dtype", choices=["auto", "fp8"], default="auto",
                    help="'fp8' stores K/V as e4m3 (not the headline configuration; reported)")
    ap.add_argument("--port", type=int, default=29533)
    ap.add_argument("--no-dp", action="store_true", help="with N > 1, skip the data-parallel side phase")
    ap.add_argument("--long-context", choices=["auto", "on", "off"], default="auto",
                    help="Qwen3-32B / 32768-token-prompt phase: 'auto' runs it with 8 GPUs")
    ap.add_argument("--no-tp", action="store_true", help="with N > 1, skip the tensor-parallel phase")
    ap.add_argument("--no-p2p", action="store_true",
                    help="with N > 1, skip the TP phase that uses the peer-to-peer all-reduce kernel")
    ap.add_argument("--phase-timeout", type=int, default=600,
                    help="time limit in seconds of one phase (N > 1)")
    ap.add_argument("--total-timeout", type=int, default=1500,
                    help="N > 1: side phases are shortened or skipped so that the whole run ends "
                         "within about this many seconds even if phases hang")
    ap.add_argument("--tp-shard", type=int, default=0,
                    help="ONE GPU: run rank 0's shard of an N-way tensor-parallel engine with every "
                         "collective issued through a single-member RCCL group (per-rank compute + "
                         "launch floor of a real N-GPU run; NOT a scaling number)")
    ap.add_argument("--p2p", action="store_true",
                    help="with --tp-shard: all-reduces by the peer-to-peer kernels in loop-back form")
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

    claim_stdout()
    phase = os.environ.get("MINIVLLM_BENCH_PHASE")
    if phase:                                    # child of an orchestrating rank
        res = run_phase(args, phase, rank, local_rank, world)
        if rank == 0:
            if phase in ("dp", "tp"):
                res["published_reference_check"] = published_reference_check()
            emit(res)
        return

    B = args.batch
    base = {"unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic random token prompts, random-init weights"}
    cfg_common = {"model": args.model, "seq_len": args.seq_len,
                  "prefill_tokens_per_step": args.prefill_tokens, "kv_block_size": args.block_size,
                  "hipgraph_decode": not args.eager,
                  **({"kv_cache_dtype": args.kv_cache_dtype} if args.kv_cache_dtype != "auto" else {}),
                  **({"layers_override": args.layers} if args.layers else {})}
    metric = f"decode tok/s ({args.model}, engine step); prefill tok/s in prefill_tok_s"

    if world == 1:
        res = run_phase(args, "single", rank, local_rank, world)
        m, second = res["main"], res["second"]
        value, ms = headline(m)
        par = "tp1"
        if args.tp_shard > 1:
            par = f"ONE rank's shard of tp{args.tp_shard}, no peers (rehearsal)"
            metric = (f"REHEARSAL, not a throughput: step time of one rank of a tp{args.tp_shard} "
                      f"engine with nobody at the other end of its collectives ({args.model})")
            base = dict(base, n_gpus=1, rehearsal=True,
                        all_reduce="p2p kernels, loop-back" if args.p2p else "rccl, single-member group")
        out = {"metric": metric, "value": value, **base, "ms_per_step": ms, "scaling": "weak",
               "config": {**cfg_common, "global_batch": B, "parallelism": par}, **_round(m),
               "gpu_state": gpu_state(),
               "published_reference_check": published_reference_check()}
        if second is not None:
            out["secondary"] = {"model": args.second_model, **_round(second)}
        emit(out)
        return

    # ---- N > 1: orchestrate one child per phase --------------------------------------
    if rank == 0:
        print(f"[bench] {world} ranks up, starting the tensor-parallel phase", file=sys.stderr, flush=True)
    clock = JobClock(args, rank, world)

    def side_phase(phase):
        limit = clock.limit_for(phase)
        return spawn_phase(args, phase, rank, local_rank, world, limit=limit) if limit else dict(SKIPPED)

    # headline first: whatever happens to the side phases, its number exists by then
    tp = None if args.no_tp else spawn_phase(args, "tp", rank, local_rank, world)
    tp_first = None
    if not args.no_tp:
        # a failed tensor-parallel phase is tried once more with the engine's newer multi-rank
        # paths off.  Rank 0's verdict decides for everybody (only its child prints the numbers).
        if clock.agree("tp_retry", "0" if phase_ok(tp) else "1") == "1":
            limit = clock.limit_for("tp", "_retry")
            if limit:
                tp_first = tp
                tp = spawn_phase(args, "tp", rank, local_rank, world, conservative=True,
                                 port_shift=37, limit=limit)
    want_long = args.long_context == "on" or (args.long_context == "auto" and world == 8)
    long_ctx = side_phase("l32k") if want_long and not args.no_tp else None
    tpc = None if args.no_tp or args.no_p2p or world not in (2, 4, 8) else side_phase("tpc")
    dp = None if args.no_dp else side_phase("dp")
    if rank != 0:
        return
    out, good = assemble_multi_gpu(args, world, dp, tp, tpc, metric, base, cfg_common, long_ctx)
    if tp_first is not None:
        out["tensor_parallel_first_attempt"] = tp_first
        out["tensor_parallel_retry"] = "conservative engine settings (MINIVLLM_TP_CONSERVATIVE=1)"
    out["gpu_state"] = gpu_state()
    emit(out)
    if not good:
        sys.exit(1)


if __name__ == "__main__":
    main()
