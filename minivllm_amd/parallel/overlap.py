"""Row-parallel GEMM with its all-reduce pipelined over token chunks.

The reference all-reduces the whole ``[T, hidden]`` output on the compute stream after the GEMM
has finished (reference: src/myvllm/layers/linear.py:216-221), so the interconnect idles during
the GEMM and the matrix cores idle during the collective.  For prefill-sized inputs this splits
the tokens into chunks: chunk i is all-reduced (RCCL runs collectives on its own HIP stream,
ordered after the producing GEMM by an event) while the GEMM of chunk i+1 runs on the compute
stream.  xGMI is point-to-point and ring collectives are per-link bound, so chunks are kept
large (>= ``min_chunk_tokens`` rows, a few MiB each) — small chunks would turn a
bandwidth-bound collective into several latency-bound ones.

Decode-sized inputs (a few hundred rows) take the plain path: one GEMM, one all-reduce.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import group as pg
from .. import ops

# Rows per chunk.  The GEMM side is measured (one MI355X, per-rank o / down shapes of Qwen3-8B at
# TP = 2, 4, 8, 16384 rows; tools/overlap_chunk_cost_probe.py): chunks of 4096 rows or more cost
# the GEMMs nothing (-10 % ... +3 %; the 512-deep o projection of TP = 8 +15 % = 10 us), chunks
# of 2048 rows 25-60 % (the 256-row tiles no longer fill the chip evenly).  The collective side
# is an estimate until measured on a real xGMI node: MINIVLLM_OVERLAP_MIN_CHUNK_TOKENS /
# MINIVLLM_OVERLAP_MAX_CHUNKS (or ``configure``) override both; MAX_CHUNKS = 1 switches the
# pipelining off.
MIN_CHUNK_TOKENS = int(os.environ.get("MINIVLLM_OVERLAP_MIN_CHUNK_TOKENS", 4096))
MAX_CHUNKS = int(os.environ.get("MINIVLLM_OVERLAP_MAX_CHUNKS", 4))


def configure(min_chunk_tokens: Optional[int] = None, max_chunks: Optional[int] = None) -> None:
    global MIN_CHUNK_TOKENS, MAX_CHUNKS
    if min_chunk_tokens:
        MIN_CHUNK_TOKENS = int(min_chunk_tokens)
    if max_chunks:
        MAX_CHUNKS = int(max_chunks)


def plan_chunks(num_tokens: int, min_chunk_tokens: Optional[int] = None,
                max_chunks: Optional[int] = None) -> int:
    min_chunk_tokens = min_chunk_tokens or MIN_CHUNK_TOKENS
    max_chunks = max_chunks or MAX_CHUNKS
    if num_tokens < 2 * min_chunk_tokens:
        return 1
    return max(1, min(max_chunks, num_tokens // min_chunk_tokens))


def chunk_bounds(num_tokens: int, n: int) -> list:
    """Row boundaries of ``n`` chunks.  Every chunk but the last has a row count from the
    prefill token buckets (the sizes the GEMM table is tuned for and the library has seen
    before); the last takes the rest, which for a bucketed ``num_tokens`` is a multiple of the
    bucket granularity too."""
    from ..engine.model_runner import prefill_token_bucket
    size = prefill_token_bucket(-(-num_tokens // n))
    bounds = [min(i * size, num_tokens) for i in range(n)] + [num_tokens]
    return [b for i, b in enumerate(bounds) if i == 0 or b > bounds[i - 1]]


def row_parallel_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                        num_chunks: Optional[int] = None) -> torch.Tensor:
    """``all_reduce(x @ weight.T (+ bias))`` over the TP group, chunk-pipelined when it pays.
    ``bias`` must already be None on every rank but one."""
    group = pg.tp_group()
    if group is None:
        return F.linear(x, weight, bias)
    T = x.shape[0] if x.dim() == 2 else 0
    n = plan_chunks(T) if num_chunks is None else num_chunks
    if n <= 1 or x.dim() != 2 or pg.host_staged_collectives():
        return pg.tp_all_reduce(ops.linear(x, weight, bias))
    works, parts = [], []
    bounds = chunk_bounds(T, n)
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        # same entry point (and so the same kernel choice) as an unchunked projection
        y = ops.linear(x[lo:hi], weight, bias)
        works.append(dist.all_reduce(y, group=group, async_op=True))
        parts.append(y)
    for w in works:
        w.wait()              # makes the compute stream wait for the collective's event
    return torch.cat(parts, 0)


def row_parallel_linear_scatter(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                                num_chunks: Optional[int] = None):
    """``x @ weight.T (+ bias)`` summed over the TP group and left SCATTERED: every rank ends up
    with its share of the rows (``ops.ScatteredRows``), for a consumer that works row-wise
    (residual add + RMSNorm) and gathers afterwards.  Chunk i is reduce-scattered while the GEMM
    of chunk i + 1 runs; the collectives are still in flight when this returns.  Each GEMM goes
    through ``ops.linear`` (the hand-written prefill kernel where it was measured faster)."""
    from ..ops.partials import ScatteredRows
    P = pg.tp_size()
    T, H = x.shape[0], weight.shape[0]
    n = plan_chunks(T) if num_chunks is None else num_chunks
    bounds = chunk_bounds(T, n) if n > 1 else [0, T]
    if any((hi - lo) % P for lo, hi in zip(bounds[:-1], bounds[1:])):
        bounds = [0, T]                               # a chunk the ranks cannot share evenly
    pipelined = len(bounds) > 2 and not pg.host_staged_collectives()
    own = torch.empty((T // P, H), dtype=x.dtype, device=x.device)
    works, keep, off = [], [], 0
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        y = ops.linear(x[lo:hi], weight, bias)
        per = (hi - lo) // P
        works.append(pg.reduce_scatter_rows(own[off:off + per], y, async_op=pipelined))
        keep.append(y)
        off += per
    return ScatteredRows(own, works, bounds, T, keep)
