"""Engine façade: tokenizer + scheduler + model runner(s).

API follows the reference's ``LLMEngine`` (reference: src/myvllm/engine/llm_engine.py:13-88):
``LLMEngine(config: dict)``, ``add_prompt``, ``step() -> (finished, num_tokens, is_prefill)``,
``generate(prompts, sampling_params) -> {'text': [...], 'token_ids': [...]}``, ``exit``.

Differences: the runner is built first and the scheduler's KV block count is the number the
runner really allocated (the reference trusts an unrelated ``max_cached_blocks``); ``step``
always returns a 3-tuple; prompts may be strings or token-id lists; throughput is
aggregated per phase (prefill tok/s, decode tok/s) over the whole run in ``self.stats``
besides the per-step print; worker processes are checked for liveness each step; and when the
process was started by ``torchrun`` (RANK/WORLD_SIZE set) no workers are spawned — the other
ranks are expected to call ``run_worker``.
"""
from __future__ import annotations

import atexit
import os
import sys
import threading
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence as Seq, Tuple, Union

import torch.multiprocessing as mp

from minivllm_amd.config import EngineConfig, as_engine_config
from minivllm_amd.engine.model_runner import ModelRunner
from minivllm_amd.engine.scheduler import Scheduler
from minivllm_amd.engine.sequence import Sequence, SequenceStatus
from minivllm_amd.parallel.group import env_rank_world
from minivllm_amd.sampling_parameters import SamplingParams
from minivllm_amd.utils.tokenizer import load_tokenizer


@dataclass
class EngineStats:
    prefill_tokens: int = 0
    prefill_time: float = 0.0
    prefill_steps: int = 0
    decode_tokens: int = 0
    decode_time: float = 0.0
    decode_steps: int = 0

    @property
    def prefill_tok_s(self) -> float:
        return self.prefill_tokens / self.prefill_time if self.prefill_time else 0.0

    @property
    def decode_tok_s(self) -> float:
        return self.decode_tokens / self.decode_time if self.decode_time else 0.0

    def as_dict(self) -> dict:
        return dict(prefill_tokens=self.prefill_tokens, prefill_steps=self.prefill_steps,
                    prefill_tok_s=self.prefill_tok_s, decode_tokens=self.decode_tokens,
                    decode_steps=self.decode_steps, decode_tok_s=self.decode_tok_s)


def _worker_main(config: EngineConfig, rank: int) -> None:
    ModelRunner(config, rank)            # enters loop() and returns after 'exit'


def run_worker(config, rank: int) -> None:
    """Entry point for a non-zero rank launched independently (e.g. by torchrun)."""
    ModelRunner(as_engine_config(config), rank)


class LLMEngine:
    def __init__(self, config: Union[dict, EngineConfig], spawn_workers: Optional[bool] = None):
        self.cfg = as_engine_config(config)
        cfg = self.cfg
        self.workers: List[mp.Process] = []
        env_rank, _, env_world = env_rank_world()
        if spawn_workers is None:
            spawn_workers = not (env_world == cfg.world_size and cfg.world_size > 1)
        if cfg.world_size > 1 and spawn_workers:
            ctx = mp.get_context("spawn")
            for r in range(1, cfg.world_size):
                p = ctx.Process(target=_worker_main, args=(cfg, r), daemon=True)
                p.start()
                self.workers.append(p)
        self.model_runner = ModelRunner(cfg, 0)
        if self.workers and self.model_runner.channel is not None:
            self.model_runner.channel.liveness = self._workers_alive
        self._step_seq = self._in_step = 0
        self._watchdog_stop = threading.Event()
        if self.workers and os.environ.get("MINIVLLM_NO_WATCHDOG") != "1":
            threading.Thread(target=self._watchdog, name="tp-worker-watchdog",
                             daemon=True).start()
        self.scheduler = Scheduler(cfg.max_num_sequences, cfg.max_num_batched_tokens,
                                   self.model_runner.num_kv_blocks, cfg.block_size, cfg.eos,
                                   max_model_length=cfg.max_model_length,
                                   enable_chunked_prefill=cfg.enable_chunked_prefill,
                                   min_prefill_tokens=cfg.min_prefill_tokens,
                                   max_prefill_delay_s=cfg.max_prefill_delay_ms / 1e3,
                                   enable_mixed_batches=cfg.enable_mixed_batches,
                                   ahead_over_block_edges=(cfg.decode_prelaunch_edges and os.environ.get(
                                       "MINIVLLM_NO_EDGE_PRELAUNCH") != "1"))
        self.tokenizer = load_tokenizer(cfg.model_name_or_path, cfg.model.vocab_size)
        self.stats = EngineStats()
        self._exited = False
        atexit.register(self.exit)

    # ------------------------------------------------------------ lifecycle
    def exit(self) -> None:
        if self._exited:
            return
        self._exited = True
        self._watchdog_stop.set()
        atexit.unregister(self.exit)
        peers_dead = not self._workers_alive()
        try:
            if peers_dead:
                # no 'exit' broadcast (nobody would acknowledge it), no closing barrier
                self.model_runner.exit(peers_dead=True)
            else:
                self.model_runner.call("exit")
        except Exception:
            peers_dead = True
            self.model_runner.exit(peers_dead=True)
        finally:
            for p in self.workers:
                if not peers_dead:
                    p.join(timeout=30)
                if p.is_alive():
                    p.terminate()                # this engine's own worker, by handle

    def _workers_alive(self) -> bool:
        return all(p.is_alive() for p in self.workers)

    def _check_workers(self) -> None:
        for p in self.workers:
            if not p.is_alive():
                raise RuntimeError(f"tensor-parallel worker pid={p.pid} died "
                                   f"(exit code {p.exitcode})")

    WATCHDOG_GRACE_S = 8.0

    def _watchdog(self) -> None:
        """A worker that dies while rank 0 sits inside a collective leaves rank 0 waiting for
        the backend's own time-out (minutes), in code no Python exception can reach.  This
        thread notices the death; if the main thread has not left the engine step on its own
        within a few seconds (``step`` raises when it gets the chance), the remaining workers
        are stopped and the process ends with a non-zero code."""
        while not self._watchdog_stop.wait(0.5):
            if self._workers_alive():
                continue
            dead = [p for p in self.workers if not p.is_alive()]
            stuck_in = self._in_step                 # 0 = the main thread is not inside a step
            if self._watchdog_stop.wait(self.WATCHDOG_GRACE_S):
                return                           # exit() ran: the main thread had its say
            if stuck_in == 0 or self._in_step != stuck_in:
                continue                         # it moved on: step() raises on its next call
            print(f"[engine] tensor-parallel worker pid={dead[0].pid} died (exit code "
                  f"{dead[0].exitcode}) and the engine is stuck waiting for it: stopping",
                  file=sys.stderr, flush=True)
            for p in self.workers:
                if p.is_alive():
                    p.terminate()
            os._exit(70)

    # --------------------------------------------------------------- requests
    def add_prompt(self, prompt: Union[str, Seq[int]],
                   sampling_params: Optional[SamplingParams] = None) -> Sequence:
        if isinstance(prompt, str):
            token_ids = self.tokenizer.encode(prompt)
        else:
            token_ids = list(prompt)
        vocab = self.cfg.model.vocab_size
        if any(t < 0 or t >= vocab for t in token_ids):
            raise ValueError(f"prompt token id outside the model vocabulary [0, {vocab})")
        seq = Sequence(token_ids, sampling_params, block_size=self.cfg.block_size)
        self.scheduler.add_sequence(seq)
        return seq

    def step(self) -> Tuple[List[Tuple[int, List[int]]], int, bool]:
        """One engine iteration.  Returns (finished [(seq_id, completion ids)],
        tokens processed, is_prefill).  Tokens processed = newly computed prompt tokens
        for a prefill step, number of sequences for a decode step
        (reference: llm_engine.py:48-60)."""
        runner = self.model_runner
        if runner.has_prelaunched():
            # the previous decode step launched this one ahead: collect it for the sequences
            # that are still running (one that finished meanwhile just loses its row)
            ahead = runner.prelaunched_seqs()
            alive = [i for i, s in enumerate(ahead) if s.status == SequenceStatus.RUNNING]
            if not alive:
                runner.drop_prelaunched()
            else:
                self._check_workers()
                seqs = [ahead[i] for i in alive]
                self.scheduler.consume_decode_step_ahead(seqs)
                again = self.scheduler.next_decode_step_ahead() if len(alive) == len(ahead) else 0
                self._step_seq += 1
                self._in_step = self._step_seq
                try:
                    token_ids = runner.run_prelaunched(alive, allow_prelaunch=again)
                finally:
                    self._in_step = 0
                finished = self.scheduler.postprocess(seqs, token_ids)
                self._drop_stale_prelaunch()
                return ([(s.seq_id, s.completion_token_ids) for s in finished], len(seqs), False)
        ahead_fn = self.scheduler.schedule_prefill_ahead if self.cfg.prefill_prelaunch else None
        if runner.has_prefill_ahead():
            # the previous prefill step admitted (and possibly launched) this one: its sequences
            # are in the scheduler's running queue already, so it has to be the next step.  A
            # sequence that was aborted meanwhile just loses its token.
            ahead, num_tokens = runner.prefill_ahead_info()
            alive = [i for i, s in enumerate(ahead) if s.status == SequenceStatus.RUNNING]
            if not alive:
                runner.drop_prefill_ahead()
            else:
                self._check_workers()
                self._step_seq += 1
                self._in_step = self._step_seq
                try:
                    token_ids = runner.run_prefill_ahead(next_prefill=ahead_fn)
                finally:
                    self._in_step = 0
                finished = self.scheduler.postprocess([ahead[i] for i in alive],
                                                      [token_ids[i] for i in alive])
                return ([(s.seq_id, s.completion_token_ids) for s in finished], num_tokens, True)
        seqs, is_prefill = self.scheduler.schedule()
        if not seqs:
            if not self.scheduler.is_finished():
                raise RuntimeError("scheduler made no progress: KV cache too small for the "
                                   "oldest waiting request")
            return [], 0, False
        self._check_workers()
        riders = self.scheduler.last_num_decode if is_prefill else 0
        if is_prefill:
            # a mixed step counts its prompt tokens plus one token per riding decode sequence
            num_tokens = sum(s.num_scheduled_prefill_tokens
                             for s in seqs[:len(seqs) - riders]) + riders
        else:
            num_tokens = len(seqs)
        self._step_seq += 1
        self._in_step = self._step_seq
        try:
            ahead_ok = 0 if is_prefill else self.scheduler.next_decode_step_ahead()
            token_ids = self.model_runner.run(seqs, is_prefill, riders, allow_prelaunch=ahead_ok,
                                              next_prefill=ahead_fn if is_prefill and not riders else None)
        finally:
            self._in_step = 0
        finished = self.scheduler.postprocess(seqs, token_ids)
        self._drop_stale_prelaunch()
        return ([(s.seq_id, s.completion_token_ids) for s in finished], num_tokens, is_prefill)

    def _drop_stale_prelaunch(self) -> None:
        """A decode step launched ahead whose sequences have all just finished would otherwise
        outlive the loop that drives ``step()`` (``generate`` stops as soon as the scheduler is
        empty) and be found by the next, unrelated use of the runner."""
        runner = self.model_runner
        if runner.has_prelaunched() and self.scheduler.is_finished():
            runner.drop_prelaunched()

    def is_finished(self) -> bool:
        return self.scheduler.is_finished()

    def generate(self, prompts: Seq[Union[str, Seq[int]]],
                 sampling_params: Union[SamplingParams, Seq[SamplingParams], None] = None,
                 verbose: Optional[bool] = None) -> Dict[str, list]:
        verbose = self.cfg.log_stats if verbose is None else verbose
        if sampling_params is None or isinstance(sampling_params, SamplingParams):
            sampling_params = [sampling_params] * len(prompts)
        order = [self.add_prompt(p, sp).seq_id for p, sp in zip(prompts, sampling_params)]
        done: Dict[int, List[int]] = {}
        st = self.stats
        while not self.scheduler.is_finished():
            t0 = time.perf_counter()
            finished, num_tokens, is_prefill = self.step()
            dt = time.perf_counter() - t0
            if is_prefill:
                st.prefill_tokens += num_tokens
                st.prefill_time += dt
                st.prefill_steps += 1
            else:
                st.decode_tokens += num_tokens
                st.decode_time += dt
                st.decode_steps += 1
            if verbose:
                phase = "prefilling" if is_prefill else "decoding"
                print(f"{num_tokens / dt:.1f} tokens/sec during {phase} "
                      f"({num_tokens} tokens, {dt * 1e3:.2f} ms)")
            done.update(finished)
        token_ids = [done[i] for i in order]
        return {"text": [self.tokenizer.decode(t) for t in token_ids], "token_ids": token_ids}
