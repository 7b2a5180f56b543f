"""Values that stand in for a tensor between a projection and the norm that consumes it.

A decoder layer's row-parallel projections (o, down) feed ``residual add + RMSNorm`` and
nothing else (reference: src/myvllm/models/qwen3.py:191-195,216, layers/linear.py:216-221).
That lets the work between the GEMM's accumulators and the normalised row be rearranged:

* ``SplitKPartials`` - the fp32 K-slice partial products of a GEMM that split K over
  workgroups, not yet summed: the norm kernel sums them while it reads the row.
* ``PendingReduce``  - this rank's partial sum of a row-parallel projection (a tensor or
  ``SplitKPartials``) whose all-reduce over the TP group is left to the consumer, which runs
  reduce + residual add + norm as one peer-to-peer kernel when it can.
* ``ResidualShard``  - this rank's rows of the residual stream after a two-shot fused
  reduce + norm: rank r keeps rows [r R, (r + 1) R) (R = rows / tp) and only normalised rows
  are gathered, so the stream stays row-sharded until the next fused norm consumes it.

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


class PendingReduce:
    """A row-parallel partial sum [rows, hidden] whose all-reduce is still to be done."""

    __slots__ = ("value",)

    def __init__(self, value: Union[torch.Tensor, SplitKPartials]):
        assert torch.is_tensor(value) or isinstance(value, SplitKPartials)
        self.value = value

    @property
    def shape(self):
        return self.value.shape

    def local_tensor(self) -> torch.Tensor:
        """This rank's partial sum as a plain tensor (K slices summed)."""
        v = self.value
        return v.materialize() if isinstance(v, SplitKPartials) else v


class ResidualShard:
    """``rows``: this rank's [total_rows / tp, hidden] slice of the residual stream."""

    __slots__ = ("rows", "total_rows")

    def __init__(self, rows: torch.Tensor, total_rows: int):
        self.rows, self.total_rows = rows, total_rows


def is_plain(x) -> bool:
    return torch.is_tensor(x)
