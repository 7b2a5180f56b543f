"""Values that stand in for a tensor between a projection and the norm that consumes it.

A decoder layer's row-parallel projections (o, down) feed ``residual add + RMSNorm`` and
nothing else (reference: src/myvllm/models/qwen3.py:191-195,216, layers/linear.py:216-221).
That lets the work between the GEMM's accumulators and the normalised row be rearranged:

* ``SplitKPartials`` - the fp32 K-slice partial products of a GEMM that split K over
  workgroups, not yet summed: the norm kernel sums them while it reads the row.
* ``PendingReduce``  - this rank's partial sum of a row-parallel projection (a tensor or
  ``SplitKPartials``) whose all-reduce over the TP group is left to the consumer, which runs
  reduce + residual add + norm as one peer-to-peer kernel when it can.
* ``ResidualSum``    - (one rank, prefill-sized) the projection's GEMM has already added the
  residual stream in its epilogue; the next projection applies the norm as a per-row factor
  behind ITS GEMM (its weights carry gamma), so the two (residual add + RMSNorm) passes of a
  layer disappear.
* ``ResidualShard``  - this rank's rows of the residual stream after a reduce-scatter shaped
  reduce + norm (the two-shot peer-to-peer kernel for decode-sized batches; RCCL reduce-scatter,
  norm on the rank's rows, all-gather for prefill-sized ones): only normalised rows are
  gathered, so the stream stays row-sharded until the next such norm consumes it and the norm
  itself costs 1 / tp of a replicated one.

They are real types, not attributes hung on tensors: a ``view`` / ``reshape`` / arithmetic on
one of them raises instead of silently dropping the pending work.  Only ``ops.rms_norm``,
``ops.add_rms_norm`` and ``ops.norm_linear`` accept them; ``Qwen3Model.forward`` asserts that
none survives the final norm.
"""
from __future__ import annotations

from typing import Optional, Union

import torch


class SplitKPartials:
    """``ws`` [S, M, N] fp32 (+ the bias still to be added).  Produced by
    ``hip.mid_linear(..., defer=True)`` / ``hip.skinny_linear(..., defer=True)``;
    ``add_rms_norm`` sums the slices inside the norm kernel, ``materialize`` gives the plain
    [M, N] tensor to any other consumer."""

    __slots__ = ("ws", "bias", "dtype")

    def __init__(self, ws: torch.Tensor, bias: Optional[torch.Tensor], dtype: torch.dtype):
        self.ws, self.bias, self.dtype = ws, bias, dtype

    @property
    def shape(self):
        return self.ws.shape[1:]

    @property
    def device(self):
        return self.ws.device

    @property
    def is_cuda(self) -> bool:
        return self.ws.is_cuda

    def materialize(self) -> torch.Tensor:
        from . import hip
        return hip.splitk_materialize(self)

    def add_rms_norm(self, residual: Optional[torch.Tensor], weight: torch.Tensor, eps: float):
        """``(rms_norm(r) * weight, r)`` with ``r = sum of the slices (+ bias) + residual``;
        ``residual`` is updated in place and returned as ``r``."""
        from . import hip
        return hip.splitk_add_rms_norm(self, residual, weight, eps)


class ResidualSum:
    """The residual stream after a projection has been added to it inside that projection's
    GEMM (``r`` [rows, hidden], updated in place by the kernel's epilogue); its RMSNorm is still
    to do.  A consumer whose weights carry ``gamma`` takes ``r`` as it is and applies
    ``rsqrt(mean(r[i]^2) + eps)`` as a row factor behind its own GEMM, which sums the squares
    from the rows it loads anyway, so no pass over ``r`` is needed to normalise it; any other
    consumer just norms ``r``."""

    __slots__ = ("r",)

    def __init__(self, r: torch.Tensor):
        self.r = r

    @property
    def shape(self):
        return self.r.shape


class ScatteredRows:
    """A row-parallel projection whose partial sums are being reduce-SCATTERED: ``own`` holds this
    rank's rows of the sum ([total_rows / tp, hidden]; for every token chunk ``bounds[i] ..
    bounds[i + 1]`` the rank owns an equal share, in rank order, and ``own`` concatenates the
    shares chunk by chunk), ``works`` the collectives still in flight (None entries: done),
    ``keep`` whatever must outlive them."""

    __slots__ = ("own", "works", "bounds", "total_rows", "keep")

    def __init__(self, own: torch.Tensor, works, bounds, total_rows: int, keep=None):
        self.own, self.works, self.bounds, self.total_rows, self.keep = own, works, tuple(bounds), total_rows, keep

    def wait(self) -> None:
        for w in self.works:
            if w is not None:
                w.wait()
        self.works, self.keep = [], None


class PendingReduce:
    """A row-parallel partial sum [rows, hidden] whose reduction over the TP group is still to be
    finished by its consumer: ``value`` (tensor or ``SplitKPartials``) when nothing has been
    sent yet, ``scattered`` when the sum is already on its way as a reduce-scatter."""

    __slots__ = ("value", "scattered")

    def __init__(self, value: Union[torch.Tensor, SplitKPartials, None] = None,
                 scattered: Optional[ScatteredRows] = None):
        assert (value is None) != (scattered is None)
        assert value is None or torch.is_tensor(value) or isinstance(value, SplitKPartials)
        self.value, self.scattered = value, scattered

    @property
    def shape(self):
        if self.scattered is not None:
            return (self.scattered.total_rows, self.scattered.own.shape[1])
        return self.value.shape


class ResidualShard:
    """``rows``: this rank's [total_rows / tp, hidden] share of the residual stream.  ``bounds``
    None: the share is rows [rank R, (rank + 1) R); a tuple of chunk bounds: an equal share of
    every chunk (see ``ScatteredRows``).  Producer and consumer must agree on it, which they do
    within a step: every layer sees the same row count and plans the same chunks."""

    __slots__ = ("rows", "total_rows", "bounds")

    def __init__(self, rows: torch.Tensor, total_rows: int, bounds=None):
        self.rows, self.total_rows, self.bounds = rows, total_rows, (None if bounds is None else tuple(bounds))


def is_plain(x) -> bool:
    return torch.is_tensor(x)
