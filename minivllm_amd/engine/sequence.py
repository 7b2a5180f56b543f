"""One generation request: prompt + generated tokens + its KV block table.

Public surface follows the reference's ``Sequence`` (reference:
src/myvllm/engine/sequence.py:14-105): ``status``, ``token_ids``, ``last_token``,
``num_tokens``, ``num_prompt_tokens``, ``num_cached_tokens``, ``block_table``,
the derived ``num_blocks`` / ``last_block_num_tokens`` / ``block(i)`` helpers and
``append_token``.  Differences: ``block_size`` is per instance and comes from the
engine configuration (the reference hard-codes a class attribute of 256 that the
configuration never reaches), and the pickled state restores every field a
worker needs.
"""
from __future__ import annotations

from enum import Enum, auto
from itertools import count
from typing import List, Optional

import numpy as np

from minivllm_amd.sampling_parameters import SamplingParams


class SequenceStatus(Enum):
    WAITING = auto()
    RUNNING = auto()
    FINISHED = auto()


class Sequence:
    counter = count()
    block_size = 256          # default when none is given; the engine always passes its own
    # chunked prefill: the prefill step scheduled now covers tokens [num_cached_tokens,
    # prefill_end); None = up to the end of the sequence
    prefill_end: Optional[int] = None
    stop_token_ids: frozenset = frozenset()
    top_k: int = 0
    top_p: float = 1.0

    def __init__(self, token_ids: List[int], sampling_params: Optional[SamplingParams] = None,
                 block_size: Optional[int] = None):
        if not token_ids:
            raise ValueError("a sequence needs at least one prompt token")
        sp = sampling_params or SamplingParams()
        self.seq_id = next(Sequence.counter)
        self.status = SequenceStatus.WAITING
        self.token_ids = list(token_ids)
        self.last_token = self.token_ids[-1]
        self.num_tokens = len(self.token_ids)
        self.num_prompt_tokens = self.num_tokens
        self.num_cached_tokens = 0
        self.block_table: List[int] = []
        self.temperature = sp.temperature
        self.max_tokens = sp.max_tokens
        self.ignore_eos = sp.ignore_eos
        self.stop_token_ids = frozenset(sp.stop_token_ids)   # host-side only (never pickled)
        # rank 0 puts them into the step's batch arrays; greedy requests ignore both
        self.top_k, self.top_p = (sp.top_k, sp.top_p) if sp.filtered else (0, 1.0)
        if block_size is not None:
            self.block_size = block_size
        # int32 image of the tokens for batch preparation (built when the request arrives, off
        # the step loop: converting a 1024-token Python list costs ~40 us, which over the 16
        # prompts of a 16k-token prefill step was most of the gap between two steps)
        self._np_tokens: Optional[np.ndarray] = np.asarray(self.token_ids, dtype=np.int32)

    def token_array(self, start: int, end: int) -> np.ndarray:
        """``token_ids[start:end]`` as an int32 array (a view of a cached image that is extended
        when generated tokens fall into the range, e.g. a re-prefill after preemption)."""
        if end - start == 1 and end == self.num_tokens:
            # the newest token alone (a decode rider of a mixed step): no need to bring the
            # image up to date, which would convert the whole list for every rider every step
            return np.array([self.last_token], dtype=np.int32)
        img = getattr(self, "_np_tokens", None)
        if img is None or img.size < end:
            img = self._np_tokens = np.asarray(self.token_ids[:max(end, self.num_tokens)], dtype=np.int32)
        return img[start:end]

    def __len__(self) -> int:
        return self.num_tokens

    def __getitem__(self, idx):
        return self.token_ids[idx]

    @property
    def is_finished(self) -> bool:
        return self.status == SequenceStatus.FINISHED

    @property
    def num_completion_tokens(self) -> int:
        return self.num_tokens - self.num_prompt_tokens

    @property
    def prompt_token_ids(self) -> List[int]:
        return self.token_ids[:self.num_prompt_tokens]

    @property
    def completion_token_ids(self) -> List[int]:
        return self.token_ids[self.num_prompt_tokens:]

    @property
    def num_cached_blocks(self) -> int:
        return self.num_cached_tokens // self.block_size

    @property
    def num_blocks(self) -> int:
        return (self.num_tokens + self.block_size - 1) // self.block_size

    @property
    def last_block_num_tokens(self) -> int:
        return self.num_tokens - (self.num_blocks - 1) * self.block_size

    def block(self, i: int) -> List[int]:
        if not 0 <= i < self.num_blocks:
            raise IndexError(f"block index {i} out of range [0, {self.num_blocks})")
        return self.token_ids[i * self.block_size:(i + 1) * self.block_size]

    @property
    def num_scheduled_prefill_tokens(self) -> int:
        """Tokens the scheduled prefill step computes for this sequence."""
        end = self.num_tokens if self.prefill_end is None else self.prefill_end
        return end - self.num_cached_tokens

    def append_token(self, token_id: int) -> None:
        self.token_ids.append(token_id)
        self.last_token = token_id
        self.num_tokens += 1

    # Compact state for shipping to another process: the full token list is only
    # needed until the prompt has been prefetched once; afterwards a worker needs
    # just the last token (same idea as reference sequence.py:88-105, but complete).
    def __getstate__(self):
        body = self.token_ids if self.num_completion_tokens == 0 else self.last_token
        return (self.seq_id, self.status, self.num_tokens, self.num_prompt_tokens,
                self.num_cached_tokens, list(self.block_table), self.temperature,
                self.max_tokens, self.ignore_eos, self.block_size, body)

    def __setstate__(self, state):
        (self.seq_id, self.status, self.num_tokens, self.num_prompt_tokens,
         self.num_cached_tokens, self.block_table, self.temperature,
         self.max_tokens, self.ignore_eos, self.block_size, body) = state
        self._np_tokens = None           # rebuilt on demand
        if isinstance(body, list):
            self.token_ids = body
            self.last_token = body[-1]
        else:
            self.token_ids = []          # history is not needed by a worker in decode
            self.last_token = body

    def __repr__(self):
        return (f"Sequence(id={self.seq_id}, {self.status.name}, tokens={self.num_tokens}, "
                f"prompt={self.num_prompt_tokens}, cached={self.num_cached_tokens}, "
                f"blocks={self.block_table})")
