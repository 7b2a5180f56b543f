"""Per-step forward context shared between the runner and the layers.

Same role and field names as the reference's module-global dataclass
(reference: src/myvllm/utils/context.py:5-27).  Two additions remove the
per-layer device->host synchronisations the reference performs
(qwen3.py:199-206, attention.py:263-264): ``positions`` is computed once per
step on the host, and ``max_seqlen_*`` / ``max_context_len`` are plain ints.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class Context:
    is_prefill: bool = False
    cu_seqlens_q: Optional[torch.Tensor] = None   # [B+1] int32, new (uncached) tokens
    cu_seqlens_k: Optional[torch.Tensor] = None   # [B+1] int32, full lengths
    max_seqlen_q: int = 0
    max_seqlen_k: int = 0
    slot_mapping: Optional[torch.Tensor] = None   # [T] int32, -1 = do not store
    context_lens: Optional[torch.Tensor] = None   # [B] int32 (decode)
    block_tables: Optional[torch.Tensor] = None   # [B, max_blocks] int32, -1 padded
    positions: Optional[torch.Tensor] = None      # [T] int32 rotary positions
    max_context_len: int = 0                      # decode: upper bound of context_lens
    num_seqs: int = 0
    has_prefix: bool = False                      # prefill reads a cached prefix
    # mixed step (a prefill-form batch whose FIRST num_decode_seqs sequences are running
    # sequences contributing one decode row each): rows [0, num_decode_seqs) are decode tokens
    # whose lengths are context_lens, the prompt tokens follow (cu_seqlens_q holds their rows)
    num_decode_seqs: int = 0
    num_prefill_tokens: int = 0
    total_k_tokens: int = 0                       # prefill: sum of context lengths (0 = unknown)
    # GPU prefill: the attention kernel's work list (ops.hip.prefill_work_items), None = derive
    # the launch from max_seqlen_q
    prefill_items: Optional[torch.Tensor] = None
    # prefill: every query's position is its place in its context (k_len - q_len + i); the GPU
    # attention kernel then derives the rotary positions instead of loading them
    positions_in_context: bool = False


_CONTEXT = Context()


def get_context() -> Context:
    return _CONTEXT


def set_context(is_prefill: bool = False, **kw) -> Context:
    global _CONTEXT
    _CONTEXT = Context(is_prefill=is_prefill, **kw)
    return _CONTEXT


def reset_context() -> None:
    global _CONTEXT
    _CONTEXT = Context()
