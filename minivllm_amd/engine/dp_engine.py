"""Data-parallel serving: N independent engine replicas behind one ``generate()``.

The reference scales only by tensor parallelism (reference: src/myvllm/engine/llm_engine.py:22-31,
one scheduler, ``world_size`` == TP degree).  On MI355X a model that fits the 288 GB of one GPU is
served fastest by NOT sharding it: every replica keeps whole weights, runs the full-width GEMMs
and never waits for an xGMI collective, so aggregate throughput grows linearly with the number of
GPUs.  Tensor parallelism stays the tool for models that do not fit (or for latency); both
compose: ``data_parallel_size`` replicas of ``world_size`` GPUs each.

Two ways to run, same class:

* self-spawned: the calling process is a thin front end; it starts one process per replica
  (replica i drives GPUs ``[i * world_size, (i + 1) * world_size)``), balances the prompts over
  them by token count and merges the results back into request order.
* launcher-started (``torchrun``: RANK / WORLD_SIZE set, one rank per replica): every rank builds
  its own replica in-process; ``generate()`` is a collective call — rank 0 passes the prompts,
  they are scattered, and the results are gathered back on rank 0 (other ranks return None).
"""
from __future__ import annotations

import os
import traceback
from dataclasses import replace
from typing import Dict, List, Optional, Sequence as Seq, Union

import torch.multiprocessing as mp

from minivllm_amd.config import EngineConfig, as_engine_config
from minivllm_amd.parallel.group import env_rank_world
from minivllm_amd.sampling_parameters import SamplingParams


def replica_config(cfg: EngineConfig, replica: int) -> EngineConfig:
    """Config of replica ``replica``: its own device range and its own rendezvous port."""
    return replace(cfg, data_parallel_size=1, device_base=cfg.device_base + replica * cfg.world_size,
                   master_port=cfg.master_port + 16 * replica)


def balance(lengths: Seq[int], parts: int) -> List[List[int]]:
    """Longest-first greedy partition of request indices over ``parts`` replicas by token
    count; every part keeps its indices in request order."""
    load = [0] * parts
    out: List[List[int]] = [[] for _ in range(parts)]
    for i in sorted(range(len(lengths)), key=lambda j: (-lengths[j], j)):
        p = min(range(parts), key=lambda q: (load[q], q))
        out[p].append(i)
        load[p] += lengths[i] + 1
    return [sorted(p) for p in out]


def _expand(prompts, sampling_params):
    if sampling_params is None or isinstance(sampling_params, SamplingParams):
        sampling_params = [sampling_params] * len(prompts)
    if len(sampling_params) != len(prompts):
        raise ValueError("one SamplingParams per prompt (or a single one for all)")
    return list(sampling_params)


def _replica_main(cfg: EngineConfig, conn) -> None:
    """Process target of a self-spawned replica: serve requests from the pipe until 'exit'."""
    eng = None
    try:
        from minivllm_amd.engine.llm_engine import LLMEngine
        os.environ.pop("LOCAL_RANK", None)          # devices come from cfg.device_base
        eng = LLMEngine(cfg)
        conn.send(("ready", eng.model_runner.num_kv_blocks))
        while True:
            cmd, payload = conn.recv()
            if cmd == "generate":
                prompts, sps = payload
                conn.send(("ok", eng.generate(prompts, sps, verbose=False)))
            elif cmd == "stats":
                conn.send(("ok", eng.stats.as_dict()))
            elif cmd == "exit":
                break
            else:
                conn.send(("error", f"unknown command {cmd!r}"))
    except Exception:
        try:
            conn.send(("error", traceback.format_exc()))
        except Exception:
            pass
    finally:
        if eng is not None:
            eng.exit()
        conn.close()


class DataParallelEngine:
    def __init__(self, config: Union[dict, EngineConfig], data_parallel_size: Optional[int] = None):
        cfg = as_engine_config(config)
        rank, _, world = env_rank_world()
        self.launcher_mode = world > 1
        self.dp_size = data_parallel_size or (world if self.launcher_mode else cfg.data_parallel_size)
        self.cfg = cfg
        self._procs: List[mp.Process] = []
        self._conns = []
        self._exited = False
        self.engine = None
        if self.launcher_mode:
            if cfg.world_size != 1:
                raise ValueError("launcher-started replicas are single-GPU (world_size 1); "
                                 "use the self-spawned mode to combine DP with TP")
            if self.dp_size != world:
                raise ValueError(f"data_parallel_size={self.dp_size} but WORLD_SIZE={world}")
            import torch.distributed as dist
            from minivllm_amd.engine.llm_engine import LLMEngine
            self.rank = rank
            if not dist.is_initialized():
                # gloo is enough: only prompt lists and results travel between replicas
                dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
                self._owns_pg = True
            else:
                self._owns_pg = False
            self.engine = LLMEngine(replace(cfg, data_parallel_size=1), spawn_workers=False)
            self.tokenizer = self.engine.tokenizer
            return
        self.rank = 0
        from minivllm_amd.utils.tokenizer import load_tokenizer
        self.tokenizer = load_tokenizer(cfg.model_name_or_path, cfg.model.vocab_size)
        ctx = mp.get_context("spawn")
        for r in range(self.dp_size):
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_replica_main, args=(replica_config(cfg, r), child), daemon=False)
            p.start()
            child.close()
            self._procs.append(p)
            self._conns.append(parent)
        self.kv_blocks = [self._recv(r, "ready") for r in range(self.dp_size)]

    # ---------------------------------------------------------------- plumbing
    def _recv(self, r: int, expect: str = "ok"):
        conn, proc = self._conns[r], self._procs[r]
        while not conn.poll(1.0):
            if not proc.is_alive():
                raise RuntimeError(f"data-parallel replica {r} (pid {proc.pid}) died "
                                   f"with exit code {proc.exitcode}")
        tag, payload = conn.recv()
        if tag == "error":
            raise RuntimeError(f"data-parallel replica {r} failed:\n{payload}")
        if tag != expect:
            raise RuntimeError(f"replica {r}: expected {expect!r}, got {tag!r}")
        return payload

    def _token_lengths(self, prompts) -> List[int]:
        return [len(self.tokenizer.encode(p)) if isinstance(p, str) else len(p) for p in prompts]

    # ------------------------------------------------------------------ serving
    def generate(self, prompts: Optional[Seq[Union[str, Seq[int]]]],
                 sampling_params: Union[SamplingParams, Seq[SamplingParams], None] = None
                 ) -> Optional[Dict[str, list]]:
        if self.launcher_mode:
            return self._generate_collective(prompts, sampling_params)
        sps = _expand(prompts, sampling_params)
        parts = balance(self._token_lengths(prompts), self.dp_size)
        for r, idx in enumerate(parts):
            self._conns[r].send(("generate", ([prompts[i] for i in idx], [sps[i] for i in idx])))
        text: List[Optional[str]] = [None] * len(prompts)
        toks: List[Optional[List[int]]] = [None] * len(prompts)
        for r, idx in enumerate(parts):
            res = self._recv(r)
            for j, i in enumerate(idx):
                text[i], toks[i] = res["text"][j], res["token_ids"][j]
        return {"text": text, "token_ids": toks}

    def _generate_collective(self, prompts, sampling_params):
        import torch.distributed as dist
        box = [None]
        if self.rank == 0:
            sps = _expand(prompts, sampling_params)
            parts = balance(self._token_lengths(prompts), self.dp_size)
            box = [[([prompts[i] for i in idx], [sps[i] for i in idx], idx) for idx in parts]]
        dist.broadcast_object_list(box, src=0)
        mine_prompts, mine_sps, _ = box[0][self.rank]
        res = self.engine.generate(mine_prompts, mine_sps, verbose=False)
        gathered = [None] * self.dp_size if self.rank == 0 else None
        dist.gather_object(res, gathered, dst=0)
        if self.rank != 0:
            return None
        n = sum(len(p[2]) for p in box[0])
        text, toks = [None] * n, [None] * n
        for (_, _, idx), part in zip(box[0], gathered):
            for j, i in enumerate(idx):
                text[i], toks[i] = part["text"][j], part["token_ids"][j]
        return {"text": text, "token_ids": toks}

    def stats(self) -> List[dict]:
        """Per-replica throughput counters (launcher mode: this rank's replica only)."""
        if self.launcher_mode:
            return [self.engine.stats.as_dict()]
        for c in self._conns:
            c.send(("stats", None))
        return [self._recv(r) for r in range(self.dp_size)]

    # ---------------------------------------------------------------- lifecycle
    def exit(self) -> None:
        if self._exited:
            return
        self._exited = True
        if self.launcher_mode:
            self.engine.exit()
            if self._owns_pg:
                import torch.distributed as dist
                if dist.is_initialized():
                    dist.destroy_process_group()
            return
        for c in self._conns:
            try:
                c.send(("exit", None))
            except Exception:
                pass
        for p in self._procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.exit()
