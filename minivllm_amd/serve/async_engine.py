"""Continuous serving loop around ``LLMEngine``.

``LLMEngine.generate`` takes a closed list of prompts and returns when all are done (reference:
src/myvllm/engine/llm_engine.py:75-88).  A server needs the opposite: requests arrive while others
are mid-generation, every request wants its tokens as they are sampled, and a client that goes
away must give its KV blocks back.  ``AsyncLLMEngine`` owns one ``LLMEngine`` in a dedicated
thread (every GPU call of the engine is made from that thread) and runs ``step()`` for as long as
there is work; between steps it admits new requests from an inbox, drops aborted ones and
publishes newly sampled tokens to the per-request ``RequestHandle``.

A handle can be consumed from plain threads (``for tokens in handle`` / ``handle.result()``) or
from asyncio (``async for tokens in handle``; items are handed to the event loop with
``call_soon_threadsafe``, the engine thread never blocks on a consumer).
"""
from __future__ import annotations

import asyncio
import itertools
import queue
import threading
import time
from typing import Dict, List, Optional, Sequence as Seq, Union

from minivllm_amd.sampling_parameters import SamplingParams

_DONE = object()


class RequestHandle:
    """One in-flight request.  Yields lists of newly generated token ids (one list per engine
    step that produced tokens for it) and then stops; ``finish_reason`` is ``"stop"`` (EOS),
    ``"length"`` (``max_tokens`` or the context limit), ``"abort"`` or ``"error"``."""

    def __init__(self, request_id: int, owner, loop: Optional[asyncio.AbstractEventLoop] = None,
                 sink=None):
        """``owner`` implements ``_cancel(handle)``.  ``sink(handle, tokens_or_None)``, when
        given, receives every item instead of a queue (None = finished); it is called on the
        producing thread and must not block."""
        self.request_id = request_id
        self._owner = owner
        self._loop = loop
        self._sink = sink
        self._q = None if sink is not None else (asyncio.Queue() if loop is not None
                                                 else queue.Queue())
        self.token_ids: List[int] = []           # everything generated so far
        self.num_prompt_tokens = 0
        self.finish_reason: Optional[str] = None
        # set by a front end that cut the request short at a stop string: the text up to it
        self.stop_text: Optional[str] = None
        self.error: Optional[BaseException] = None
        self.submitted_at = time.perf_counter()
        self.first_token_at: Optional[float] = None
        self.finished_at: Optional[float] = None
        self._done = threading.Event()
        # engine-thread state
        self._seq = None
        self._sent = 0

    # ------------------------------------------------ engine thread side
    def _put(self, item) -> None:
        if self._sink is not None:
            self._sink(self, None if item is _DONE else item)
        elif self._loop is not None:
            try:
                self._loop.call_soon_threadsafe(self._q.put_nowait, item)
            except RuntimeError:                  # the consumer's loop is gone
                pass
        else:
            self._q.put(item)

    def _emit(self, tokens: List[int]) -> None:
        if self.first_token_at is None:
            self.first_token_at = time.perf_counter()
        self.token_ids.extend(tokens)
        self._put(tokens)

    def _finish(self, reason: str, error: Optional[BaseException] = None) -> None:
        if self.finish_reason is not None:
            return
        self.finish_reason, self.error = reason, error
        self.finished_at = time.perf_counter()
        self._done.set()
        self._put(_DONE)

    # ------------------------------------------------------ consumer side
    @property
    def finished(self) -> bool:
        return self._done.is_set()

    def cancel(self) -> None:
        """Ask the engine to drop the request (no-op once it has finished)."""
        if not self.finished:
            self._owner._cancel(self)

    def __iter__(self):
        assert self._loop is None, "created for asyncio: use 'async for'"
        while True:
            item = self._q.get()
            if item is _DONE:
                break
            yield item
        if self.error is not None:
            raise self.error

    def __aiter__(self):
        assert self._loop is not None, "created without an event loop: use 'for'"
        return self._agen()

    async def _agen(self):
        while True:
            item = await self._q.get()
            if item is _DONE:
                break
            yield item
        if self.error is not None:
            raise self.error

    def result(self, timeout: Optional[float] = None) -> List[int]:
        """Block until the request has finished; returns all generated token ids."""
        if not self._done.wait(timeout):
            raise TimeoutError(f"request {self.request_id} still running after {timeout} s")
        if self.error is not None:
            raise self.error
        return list(self.token_ids)


class AsyncLLMEngine:
    def __init__(self, config, start: bool = True):
        """``config``: whatever ``LLMEngine`` accepts (flat dict or ``EngineConfig``)."""
        self._config = config
        self._inbox: "queue.Queue" = queue.Queue()
        self._ids = itertools.count()
        self._active: Dict[int, RequestHandle] = {}        # seq_id -> handle (engine thread)
        self._ready = threading.Event()
        self._stopped = threading.Event()                  # set when the loop has ended
        self._startup_error: Optional[BaseException] = None
        self._thread: Optional[threading.Thread] = None
        self.engine = None
        self._stats_snapshot: dict = {}
        self.requests_finished = 0
        self.requests_aborted = 0
        if start:
            self.start()

    # ----------------------------------------------------------- lifecycle
    def start(self, timeout: Optional[float] = None) -> None:
        if self._thread is not None:
            return
        self._thread = threading.Thread(target=self._run, name="minivllm-engine", daemon=True)
        self._thread.start()
        if not self._ready.wait(timeout):
            raise TimeoutError("engine did not come up in time")
        if self._startup_error is not None:
            raise self._startup_error

    def shutdown(self, drain: bool = True, timeout: Optional[float] = 60.0) -> None:
        """Stop the loop; with ``drain`` the requests in flight finish first, otherwise they end
        with ``finish_reason == "abort"``."""
        if self._thread is None:
            return
        self._inbox.put(("stop", drain))
        self._thread.join(timeout)
        self._thread = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.shutdown(drain=False)

    @property
    def tokenizer(self):
        return self.engine.tokenizer

    # ------------------------------------------------------------ requests
    @property
    def cfg(self):
        return self.engine.cfg

    def _cancel(self, handle: RequestHandle) -> None:
        self._inbox.put(("abort", handle))

    def submit(self, prompt: Union[str, Seq[int]], sampling_params: Optional[SamplingParams] = None,
               loop: Optional[asyncio.AbstractEventLoop] = None, sink=None) -> RequestHandle:
        """Queue a request (any thread).  Pass the running event loop to consume the handle with
        ``async for``, or a ``sink`` callable to be called with every item (see RequestHandle)."""
        h = RequestHandle(next(self._ids), self, loop, sink)
        return self._submit_handle(h, prompt, sampling_params)

    def _submit_handle(self, h: RequestHandle, prompt, sampling_params) -> RequestHandle:
        if self._stopped.is_set():                         # nobody would ever read the inbox
            h._finish("error", RuntimeError("the engine loop is not running"))
            return h
        self._inbox.put(("add", h, prompt, sampling_params))
        if self._stopped.is_set():                         # lost the race with a dying loop
            h._finish("error", RuntimeError("the engine loop is not running"))
        return h

    def generate(self, prompt, sampling_params: Optional[SamplingParams] = None,
                 timeout: Optional[float] = None) -> List[int]:
        return self.submit(prompt, sampling_params).result(timeout)

    def stats(self) -> dict:
        """The counters as of the engine thread's last step.  Other threads never touch the
        scheduler or the (native) block manager themselves: the engine thread takes the
        snapshot between steps and readers get a copy of it."""
        return dict(self._stats_snapshot)

    def _snapshot_stats(self) -> None:                     # engine thread only
        eng = self.engine
        sch = eng.scheduler
        out = eng.stats.as_dict()
        out.update(running=len(sch.running) + len(sch.prefilling), waiting=len(sch.waiting),
                   requests_finished=self.requests_finished, requests_aborted=self.requests_aborted,
                   preemptions=sch.num_preemptions,
                   kv_blocks_free=sch.block_manager.num_free_blocks,
                   kv_blocks=sch.block_manager.num_blocks,
                   prefix_hits=sch.block_manager.num_prefix_hits,
                   prefill_graphs=len(eng.model_runner.prefill_graphs),
                   prefill_graph_replays=eng.model_runner.prefill_graph_replays)
        self._stats_snapshot = out                         # one reference assignment

    # ------------------------------------------------------- engine thread
    def _handle(self, item) -> Optional[bool]:
        kind = item[0]
        if kind == "add":
            _, h, prompt, sp = item
            try:
                seq = self.engine.add_prompt(prompt, sp)
            except Exception as e:                         # e.g. prompt longer than the context
                h._finish("error", e)
                return None
            h._seq, h.num_prompt_tokens = seq, seq.num_prompt_tokens
            self._active[seq.seq_id] = h
        elif kind == "abort":
            h = item[1]
            if h._seq is not None and self._active.pop(h._seq.seq_id, None) is not None:
                self.engine.scheduler.abort(h._seq)
                self.requests_aborted += 1
                self._snapshot_stats()
                h._finish("abort")
        elif kind == "stop":
            return bool(item[1])
        return None

    def _publish(self, finished) -> None:
        for h in self._active.values():
            seq = h._seq
            n = seq.num_completion_tokens
            if n > h._sent:
                start = seq.num_prompt_tokens + h._sent
                h._sent = n
                h._emit(seq.token_ids[start:seq.num_tokens])
        done = []
        for seq_id, _ in finished:
            h = self._active.pop(seq_id, None)
            if h is None:
                continue
            seq = h._seq
            eos = ((not seq.ignore_eos and seq.last_token == self.engine.scheduler.eos)
                   or seq.last_token in seq.stop_token_ids)
            self.requests_finished += 1
            done.append((h, "stop" if eos else "length"))
        # a client that sees its request complete must also see it in the counters
        self._snapshot_stats()
        for h, reason in done:
            h._finish(reason)

    def _fail_all(self, err: BaseException) -> None:
        # the step raised: whatever KV it was writing does not count as computed
        for h in list(self._active.values()):
            self.engine.scheduler.abort(h._seq, step_failed=True)
            h._finish("error", err)
        self._active.clear()

    def _run(self) -> None:
        try:
            from minivllm_amd.engine.llm_engine import LLMEngine
            self.engine = LLMEngine(self._config)
        except BaseException as e:
            self._startup_error = e
            self._stopped.set()
            self._ready.set()
            return
        self._snapshot_stats()
        self._ready.set()
        eng, st = self.engine, self.engine.stats
        stopping, drain = False, True
        try:
            while True:
                idle = eng.is_finished()
                if stopping and (idle or not drain):
                    break
                try:
                    item = self._inbox.get(timeout=0.05) if idle else self._inbox.get_nowait()
                except queue.Empty:
                    item = None
                while item is not None:
                    stop = self._handle(item)
                    if stop is not None:
                        stopping, drain = True, stop
                    try:
                        item = self._inbox.get_nowait()
                    except queue.Empty:
                        item = None
                if eng.is_finished() or (stopping and not drain):
                    self._snapshot_stats()
                    continue
                t0 = time.perf_counter()
                try:
                    finished, num_tokens, is_prefill = eng.step()
                except Exception as e:                     # keep serving: fail what was in flight
                    self._fail_all(e)
                    continue
                dt = time.perf_counter() - t0
                if is_prefill:
                    st.prefill_tokens += num_tokens; st.prefill_time += dt; st.prefill_steps += 1
                else:
                    st.decode_tokens += num_tokens; st.decode_time += dt; st.decode_steps += 1
                self._publish(finished)
        finally:
            self._stopped.set()
            for h in list(self._active.values()):
                self.engine.scheduler.abort(h._seq)
                h._finish("abort")
            self._active.clear()
            while True:                                    # requests that never reached the engine
                try:
                    item = self._inbox.get_nowait()
                except queue.Empty:
                    break
                if item[0] == "add":
                    item[1]._finish("abort")
            eng.exit()
