"""Attention layer over a paged KV cache, plus the three kernel entry points the
reference exposes.

Reference (src/myvllm/layers/attention.py:67-108, 212-280, 418-473, 476-557):
``store_kvcache`` scatters new K/V rows into the cache, prefill runs a varlen
causal flash-attention over the *new* tokens only (so a cached prefix is silently
ignored), decode runs one-query paged attention with a ``block_size`` that is
never wired to the engine's.  Here the cache tensor itself defines the block
size, prefill attends through the block table whenever a cache is attached
(``q_len <= k_len``, so prefix-cache hits and re-prefill after preemption are
exact), and K/V are written by the fused norm+RoPE+store kernel upstream, not by
this layer.

Cache tensors are logically ``[num_blocks, block_size, num_kv_heads, head_dim]``;
the engine allocates them with a head-major physical order so that one
(block, kv-head) pair is one contiguous slab for the decode kernel.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn

from minivllm_amd import ops
from minivllm_amd.utils.context import get_context


# ----------------------------------------------- reference-signature functions
def store_kvcache(key: torch.Tensor, value: torch.Tensor, k_cache: torch.Tensor,
                  v_cache: torch.Tensor, slot_mapping: torch.Tensor,
                  block_size: Optional[int] = None) -> None:
    if block_size is not None and block_size != k_cache.shape[1]:
        raise ValueError(f"block_size={block_size} does not match the cache ({k_cache.shape[1]})")
    ops.store_kvcache(key, value, k_cache, v_cache, slot_mapping)


def flash_attention_prefill(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                            cu_seqlens: torch.Tensor, scale: float,
                            num_heads: Optional[int] = None, num_kv_heads: Optional[int] = None,
                            head_dim: Optional[int] = None,
                            max_seqlen: Optional[int] = None) -> torch.Tensor:
    """q [T,Hq,D], k/v [T,Hkv,D], causal, packed by ``cu_seqlens``.  ``scale`` is the
    full softmax scale (1/sqrt(D) already included)."""
    return ops.flash_attention_prefill(q, k, v, cu_seqlens, scale, max_seqlen)


def paged_attention_decode(query: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                           block_tables: torch.Tensor, context_lens: torch.Tensor, scale: float,
                           num_heads: Optional[int] = None, num_kv_heads: Optional[int] = None,
                           head_dim: Optional[int] = None, block_size: Optional[int] = None,
                           max_context_len: Optional[int] = None) -> torch.Tensor:
    if block_size is not None and block_size != k_cache.shape[1]:
        raise ValueError(f"block_size={block_size} does not match the cache ({k_cache.shape[1]})")
    return ops.paged_attention_decode(query, k_cache, v_cache, block_tables, context_lens, scale,
                                      max_context_len)


# --------------------------------------------------------------------- layer
# MINIVLLM_NO_Q_FUSE=1: q-norm + RoPE of the Q rows stay in the store kernel (for A/B timing)
_FUSE_Q_ROPE = os.environ.get("MINIVLLM_NO_Q_FUSE", "0") != "1"
# MINIVLLM_NO_DECODE_FUSE=1: decode steps keep the stand-alone norm + rotate + store kernel
_FUSE_DECODE_STORE = os.environ.get("MINIVLLM_NO_DECODE_FUSE", "0") != "1"


class Attention(nn.Module):
    def __init__(self, num_heads: int, head_dim: int, scale: float = 1.0,
                 num_kv_heads: Optional[int] = None, block_size: Optional[int] = None):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.num_kv_heads = num_kv_heads or num_heads
        # like the reference, ``scale`` multiplies 1/sqrt(head_dim) (attention.py:512)
        self.scale = scale / math.sqrt(head_dim)
        self.block_size = block_size
        self.k_cache: Optional[torch.Tensor] = None
        self.v_cache: Optional[torch.Tensor] = None

    @staticmethod
    def fuses_q_rope(ctx, q: torch.Tensor) -> bool:
        """True for the steps whose attention transforms raw Q rows itself (``q_rope``): pure
        prefill steps on the GPU.  Decode rows (decode steps, the riders of a mixed step) go
        through the decode kernel, which expects finished queries."""
        return bool(_FUSE_Q_ROPE and ctx.is_prefill and not ctx.num_decode_seqs and q.is_cuda)

    def fuses_decode_store(self, ctx, qkv: torch.Tensor) -> bool:
        """True for the steps whose attention takes the raw q / k / v projections and does the
        norm, the rotation and the cache write itself (``new_kv``): pure decode steps on the GPU
        with a 16-bit cache whose positions are ``context_lens - 1`` (the engine's are)."""
        return bool(_FUSE_DECODE_STORE and not ctx.is_prefill and qkv.is_cuda
                    and ctx.positions_in_context and self.k_cache is not None
                    and self.k_cache.dtype != torch.float8_e4m3fn and self.head_dim in (64, 128))

    def reads_k_from_cache(self, ctx) -> bool:
        """True for prefill steps whose attention goes through the block table of a 16-bit cache
        (the freshly projected ``k`` / ``v`` arguments of ``forward`` are then never read)."""
        return bool(ctx.is_prefill and self.k_cache is not None and ctx.block_tables is not None
                    and ctx.slot_mapping is not None and self.k_cache.dtype != torch.float8_e4m3fn)

    def forward(self, q: torch.Tensor, k: Optional[torch.Tensor] = None,
                v: Optional[torch.Tensor] = None, q_rope=None, new_kv=None) -> torch.Tensor:
        """q [T,Hq,D] (rows may be strided); k/v [T,Hkv,D] are only read when no cache
        is attached.  ``q_rope`` (prefill steps only): q holds the raw projection, see
        ``ops.flash_attention_prefill``.  ``new_kv`` = (cos_sin, q_norm weight, k_norm weight,
        eps) (decode steps only, ``fuses_decode_store``): q, k and v are raw projections, see
        ``ops.paged_attention_decode``.  Returns [T, Hq*D]."""
        ctx = get_context()
        T = q.shape[0]
        assert q_rope is None or self.fuses_q_rope(ctx, q)
        assert new_kv is None or self.fuses_decode_store(ctx, q)
        if ctx.is_prefill and ctx.num_decode_seqs:
            return self._mixed(q, ctx)
        if ctx.is_prefill:
            paged = self.k_cache is not None and ctx.block_tables is not None
            if paged and self.k_cache.dtype == torch.float8_e4m3fn and not ctx.has_prefix \
                    and k is not None:
                paged = False       # nothing cached yet: attend to the exact K/V just projected
            if paged:
                o = ops.paged_attention_prefill(q, self.k_cache, self.v_cache, ctx.cu_seqlens_q,
                                                ctx.cu_seqlens_k, ctx.block_tables, self.scale,
                                                ctx.max_seqlen_q, max_seqlen_k=ctx.max_seqlen_k,
                                                total_k=ctx.total_k_tokens, q_rope=q_rope,
                                                work_items=ctx.prefill_items,
                                                positions_in_context=ctx.positions_in_context)
            else:
                o = ops.flash_attention_prefill(q, k, v, ctx.cu_seqlens_q, self.scale,
                                                ctx.max_seqlen_q, q_rope=q_rope)
        else:
            if self.k_cache is None:
                raise RuntimeError("decode needs an allocated KV cache")
            o = ops.paged_attention_decode(q, self.k_cache, self.v_cache, ctx.block_tables,
                                           ctx.context_lens, self.scale, ctx.max_context_len,
                                           new_kv=(k, v) + tuple(new_kv) if new_kv is not None else None)
        return o.reshape(T, self.num_heads * self.head_dim)

    def _mixed(self, q: torch.Tensor, ctx) -> torch.Tensor:
        """The decode rows (the first ``num_decode_seqs`` rows and sequence slots) through the
        decode kernel, the prompt rows behind them through the prefill kernel (a one-row query
        in the prefill kernel would stream the KV of its sequence once per query head and fill
        1 of 128 tile rows), one output buffer.  Rows that belong to no sequence (padding of the
        token bucket, unused slots of a captured step) are left as they are."""
        if self.k_cache is None or ctx.block_tables is None:
            raise RuntimeError("a mixed prefill + decode step needs an allocated KV cache")
        T = q.shape[0]
        Bd = ctx.num_decode_seqs
        o = q.new_empty((T, self.num_heads, self.head_dim)) if q.is_cuda \
            else q.new_zeros((T, self.num_heads, self.head_dim))
        if ctx.num_seqs > Bd:
            # cu_seqlens hold absolute rows, so q and o go in whole
            ops.paged_attention_prefill(q, self.k_cache, self.v_cache,
                                        ctx.cu_seqlens_q[Bd:], ctx.cu_seqlens_k[Bd:],
                                        ctx.block_tables[Bd:], self.scale, ctx.max_seqlen_q,
                                        out=o, max_seqlen_k=ctx.max_seqlen_k,
                                        total_k=ctx.total_k_tokens, work_items=ctx.prefill_items)
        ops.paged_attention_decode(q[:Bd], self.k_cache, self.v_cache,
                                   ctx.block_tables[:Bd], ctx.context_lens, self.scale,
                                   ctx.max_context_len, out=o[:Bd])
        return o.reshape(T, self.num_heads * self.head_dim)
