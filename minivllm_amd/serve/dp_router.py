"""Online serving over N engine replicas: one front end, one process per replica.

``engine/dp_engine.py`` spreads a closed list of prompts over replicas; a server needs the same
deployment for requests that arrive one at a time.  ``DataParallelAsyncEngine`` has the interface
of ``AsyncLLMEngine`` (``submit`` -> ``RequestHandle``, ``stats``, ``shutdown``) so the HTTP front
end runs on either.  Every replica process hosts its own ``AsyncLLMEngine`` (whole weights on its
own GPU range, its own scheduler and KV cache: the throughput-optimal MI355X deployment of a model
that fits 288 GB, no collective on the critical path); a request goes to the replica with the
fewest outstanding tokens - unless another replica has recently served a prompt with the same
first KV block and is not much busier: its prefix cache probably still holds that prefix (system
prompts, chat histories), so the request is sent there and skips that part of the prefill.  Tokens
come back over one shared result queue and a dispatcher thread hands them to the caller's handle.  The reference has neither replicas nor a server
(reference: src/myvllm/engine/llm_engine.py:13-42).
"""
from __future__ import annotations

import asyncio
import itertools
import queue
import threading
import traceback
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence as Seq, Union

import torch.multiprocessing as mp

from minivllm_amd.config import EngineConfig, as_engine_config
from minivllm_amd.engine.dp_engine import replica_config
from minivllm_amd.sampling_parameters import SamplingParams
from minivllm_amd.serve.async_engine import AsyncLLMEngine, RequestHandle


def _replica_serve(cfg: EngineConfig, replica: int, inq, outq) -> None:
    """Process target: an AsyncLLMEngine fed from ``inq``; every token list and every completion
    goes to ``outq`` tagged with the front end's request id."""
    import os
    os.environ.pop("LOCAL_RANK", None)               # devices come from cfg.device_base
    try:
        eng = AsyncLLMEngine(cfg)
    except BaseException:
        outq.put(("failed", replica, traceback.format_exc()))
        return
    outq.put(("ready", replica, eng.engine.model_runner.num_kv_blocks))
    handles: Dict[int, RequestHandle] = {}
    lock = threading.Lock()

    def sink(h: RequestHandle, tokens) -> None:      # engine thread
        if tokens is not None:
            outq.put(("tok", h.tag, tokens))
            return
        with lock:
            handles.pop(h.tag, None)
        err = h.error
        outq.put(("done", h.tag, h.finish_reason, None if err is None else str(err),
                  None if err is None else type(err).__name__, h.num_prompt_tokens))

    try:
        while True:
            msg = inq.get()
            kind = msg[0]
            if kind == "add":
                _, rid, prompt, sp = msg
                h = RequestHandle(-1, eng, None, sink)
                h.tag = rid
                with lock:
                    handles[rid] = h
                eng._submit_handle(h, prompt, sp)
            elif kind == "abort":
                with lock:
                    h = handles.get(msg[1])
                if h is not None:
                    h.cancel()
            elif kind == "stats":
                outq.put(("stats", replica, msg[1], eng.stats()))
            elif kind == "stop":
                eng.shutdown(drain=bool(msg[1]))
                break
    finally:
        eng.shutdown(drain=False)
        outq.put(("stopped", replica, None))


class DataParallelAsyncEngine:
    def __init__(self, config, data_parallel_size: Optional[int] = None,
                 start_timeout_s: float = 600.0):
        cfg = as_engine_config(config)
        self._cfg = cfg
        self.dp_size = data_parallel_size or cfg.data_parallel_size
        if self.dp_size < 1:
            raise ValueError("data_parallel_size must be >= 1")
        from minivllm_amd.utils.tokenizer import load_tokenizer
        self.tokenizer = load_tokenizer(cfg.model_name_or_path, cfg.model.vocab_size)
        ctx = mp.get_context("spawn")
        self._outq = ctx.Queue()
        self._inqs = [ctx.Queue() for _ in range(self.dp_size)]
        self._procs = [ctx.Process(target=_replica_serve,
                                   args=(replica_config(cfg, r), r, self._inqs[r], self._outq))
                       for r in range(self.dp_size)]
        for p in self._procs:
            p.start()
        self._ids = itertools.count()
        self._lock = threading.Lock()
        self._handles: Dict[int, RequestHandle] = {}
        self._where: Dict[int, int] = {}             # request id -> replica
        self._cost: Dict[int, int] = {}
        self._load = [0] * self.dp_size              # outstanding tokens per replica
        self._stats_wait: Dict[int, list] = {}
        self._stats_ids = itertools.count()
        self._stopped = threading.Event()
        self._dead: set = set()                      # replicas whose process has gone away
        # prefix affinity: hash of a prompt's first KV block -> replica that served it last
        self._affinity: "OrderedDict[int, int]" = OrderedDict()
        self._affinity_cap = 65536
        # follow the affinity only while that replica's backlog exceeds the lightest one's by
        # less than this many tokens
        self.affinity_slack_tokens = 4 * cfg.max_num_batched_tokens
        self.affinity_hits = 0
        self.kv_blocks = [0] * self.dp_size
        ready = 0
        while ready < self.dp_size:
            try:
                msg = self._outq.get(timeout=1.0)
            except queue.Empty:
                dead = [p for p in self._procs if not p.is_alive()]
                start_timeout_s -= 1.0
                if dead or start_timeout_s <= 0:
                    self._kill()
                    raise RuntimeError("a data-parallel replica did not come up")
                continue
            if msg[0] == "failed":
                self._kill()
                raise RuntimeError(f"data-parallel replica {msg[1]} failed:\n{msg[2]}")
            if msg[0] == "ready":
                self.kv_blocks[msg[1]] = msg[2]
                ready += 1
        self._dispatcher = threading.Thread(target=self._dispatch, name="minivllm-dp-router",
                                            daemon=True)
        self._dispatcher.start()

    # ------------------------------------------------------------ interface
    @property
    def cfg(self) -> EngineConfig:
        return self._cfg

    def submit(self, prompt: Union[str, Seq[int]], sampling_params: Optional[SamplingParams] = None,
               loop: Optional[asyncio.AbstractEventLoop] = None) -> RequestHandle:
        h = RequestHandle(next(self._ids), self, loop)
        if self._stopped.is_set():
            h._finish("error", RuntimeError("the serving loop is not running"))
            return h
        ids = self.tokenizer.encode(prompt) if isinstance(prompt, str) else list(prompt)
        sp = sampling_params or SamplingParams()
        cost = len(ids) + sp.max_tokens
        with self._lock:
            alive = [q for q in range(self.dp_size) if q not in self._dead]
            if not alive:
                h._finish("error", RuntimeError("no data-parallel replica is alive"))
                return h
            r = min(alive, key=lambda q: (self._load[q], q))
            key = self._prefix_key(ids)
            if key is not None:
                sticky = self._affinity.get(key)
                if (sticky is not None and sticky not in self._dead
                        and self._load[sticky] - self._load[r] <= self.affinity_slack_tokens):
                    self.affinity_hits += 1
                    r = sticky
                self._affinity[key] = r
                self._affinity.move_to_end(key)
                if len(self._affinity) > self._affinity_cap:
                    self._affinity.popitem(last=False)
            self._load[r] += cost
            self._handles[h.request_id] = h
            self._where[h.request_id] = r
            self._cost[h.request_id] = cost
        self._inqs[r].put(("add", h.request_id, ids, sp))
        return h

    def generate(self, prompt, sampling_params: Optional[SamplingParams] = None,
                 timeout: Optional[float] = None) -> List[int]:
        return self.submit(prompt, sampling_params).result(timeout)

    def _prefix_key(self, ids: List[int]) -> Optional[int]:
        """Hash of the first full KV block of the prompt (the unit the replicas' prefix caches
        work in); None for prompts shorter than a block."""
        bs = self._cfg.block_size
        if len(ids) <= bs:
            return None
        from minivllm_amd.engine.block_manager import BlockManager
        return BlockManager.compute_hash(ids[:bs])

    def _cancel(self, handle: RequestHandle) -> None:
        with self._lock:
            r = self._where.get(handle.request_id)
        if r is not None:
            self._inqs[r].put(("abort", handle.request_id))

    def stats(self, timeout: float = 10.0) -> dict:
        """Counters summed over the replicas (rates are sums too: the replicas run side by
        side); ``replicas`` holds the per-replica dictionaries."""
        sid = next(self._stats_ids)
        box: list = []
        with self._lock:
            self._stats_wait[sid] = box
        for q in self._inqs:
            q.put(("stats", sid))
        import time
        deadline = time.monotonic() + timeout
        # only replicas that are alive can answer: a dead one must not hold every caller for
        # the full timeout
        while (len(box) < self.dp_size - len(self._dead) and time.monotonic() < deadline
               and not self._stopped.is_set()):
            time.sleep(0.002)
        with self._lock:
            self._stats_wait.pop(sid, None)
        per = [s for _, s in sorted(box, key=lambda t: t[0])]
        out: dict = {"replicas": per, "outstanding_tokens": list(self._load)}
        for s in per:
            for k, v in s.items():
                if isinstance(v, (int, float)) and v is not None:
                    out[k] = out.get(k, 0) + v
        return out

    def shutdown(self, drain: bool = True, timeout: float = 120.0) -> None:
        if self._stopped.is_set():
            return
        for q in self._inqs:
            q.put(("stop", drain))
        for p in self._procs:
            p.join(timeout)
        self._stopped.set()
        self._kill()
        self._dispatcher.join(5.0)
        with self._lock:
            left = list(self._handles.values())
            self._handles.clear()
        for h in left:
            h._finish("abort")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.shutdown(drain=False)

    # -------------------------------------------------------------- internals
    def _kill(self) -> None:
        for p in self._procs:
            if p.is_alive():
                p.terminate()                        # this replica's process, by handle

    def _reap_dead_replicas(self) -> None:
        """A replica process that went away takes its requests with it: fail them (the caller
        can retry) and stop routing to it."""
        for r, p in enumerate(self._procs):
            if r in self._dead or p.is_alive():
                continue
            with self._lock:
                self._dead.add(r)
                lost = [rid for rid, where in self._where.items() if where == r]
                hs = [self._handles.pop(rid) for rid in lost if rid in self._handles]
                for rid in lost:
                    self._where.pop(rid, None)
                    self._cost.pop(rid, None)
                self._load[r] = 0
            err = RuntimeError(f"data-parallel replica {r} (pid {p.pid}) exited with code {p.exitcode}")
            for h in hs:
                h._finish("error", err)

    def _dispatch(self) -> None:
        import time
        stopped = 0
        next_reap = time.monotonic() + 0.5
        while stopped < self.dp_size:
            # liveness is checked on a timer, not only when the queue runs dry: with other
            # replicas streaming tokens the queue is rarely empty
            if time.monotonic() >= next_reap:
                next_reap = time.monotonic() + 0.5
                if not self._stopped.is_set():
                    self._reap_dead_replicas()
                    if len(self._dead) == self.dp_size:
                        break
            try:
                msg = self._outq.get(timeout=0.5)
            except queue.Empty:
                if self._stopped.is_set():
                    break
                continue
            kind = msg[0]
            if kind == "tok":
                h = self._handles.get(msg[1])
                if h is not None:
                    h._emit(msg[2])
            elif kind == "done":
                _, rid, reason, err, err_type, n_prompt = msg
                with self._lock:
                    h = self._handles.pop(rid, None)
                    r = self._where.pop(rid, None)
                    if r is not None:
                        self._load[r] -= self._cost.pop(rid, 0)
                if h is not None:
                    h.num_prompt_tokens = n_prompt
                    e = None
                    if err is not None:
                        e = ValueError(err) if err_type == "ValueError" else RuntimeError(err)
                    h._finish(reason, e)
            elif kind == "stats":
                with self._lock:
                    box = self._stats_wait.get(msg[2])
                if box is not None:
                    box.append((msg[1], msg[3]))
            elif kind == "stopped":
                stopped += 1
                self._dead.add(msg[1])
        with self._lock:
            left = list(self._handles.values())
            self._handles.clear()
        for h in left:                               # replicas gone with requests in flight
            h._finish("abort")
