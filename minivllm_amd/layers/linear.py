"""Tensor-parallel linear layers (Megatron-style sharding).

Class names and the ``weight_loader`` protocol follow the reference (reference:
src/myvllm/layers/linear.py:5-221): every parameter carries a
``weight_loader(param, full_tensor[, shard_id])`` that narrows a full checkpoint
tensor to this rank's shard.  Corrected relative to the reference: the
column-parallel loader narrows by the shard *length* (the reference passes
``start + shard``), the row-parallel forward all-reduces in place and returns the
tensor (the reference returns the collective's return value), and bias is added
on rank 0 only before the reduction.

The GEMMs are library calls (hipBLASLt through ``F.linear``) except for decode-sized inputs
of the shapes where the weight-streaming kernel of ``csrc/skinny_gemm.hip`` is faster
(``ops.linear`` / ``ops.swiglu_linear`` decide).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from minivllm_amd import ops
from minivllm_amd.parallel import group as pg


def _divide(n: int, d: int) -> int:
    if n % d:
        raise ValueError(f"{n} is not divisible by {d}")
    return n // d


class LinearBase(nn.Module):
    def __init__(self, input_size: int, output_size: int, bias: bool = False,
                 tp_dim: Optional[int] = None, tp_rank: Optional[int] = None,
                 tp_size: Optional[int] = None):
        super().__init__()
        self.tp_dim = tp_dim
        self.tp_rank = pg.tp_rank() if tp_rank is None else tp_rank
        self.tp_size = pg.tp_size() if tp_size is None else tp_size
        self.weight = nn.Parameter(torch.empty(output_size, input_size), requires_grad=False)
        self.weight.weight_loader = self.weight_loader
        if bias:
            self.bias = nn.Parameter(torch.zeros(output_size), requires_grad=False)
            self.bias.weight_loader = self.weight_loader
        else:
            self.register_parameter("bias", None)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor, *args):
        raise NotImplementedError

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


class ReplicatedLinear(LinearBase):
    def __init__(self, input_size: int, output_size: int, bias: bool = False, **kw):
        super().__init__(input_size, output_size, bias, tp_dim=None, **kw)

    def weight_loader(self, param, loaded_weight, *args):
        param.data.copy_(loaded_weight)

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


class ColumnParallelLinear(LinearBase):
    """Output features sharded: weight [out/tp, in]."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False, **kw):
        tp_size = kw.get("tp_size") or pg.tp_size()
        super().__init__(input_size, _divide(output_size, tp_size), bias, tp_dim=0, **kw)
        self.full_output_size = output_size

    def weight_loader(self, param, loaded_weight, *args):
        shard = param.data.size(0)
        param.data.copy_(loaded_weight.narrow(0, self.tp_rank * shard, shard))

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Several column-parallel matrices fused along the output dim (gate|up).
    Each rank stores [shard(gate); shard(up)] so the fused GEMM output splits
    evenly into the per-matrix shards."""

    def __init__(self, input_size: int, output_sizes: List[int], bias: bool = False, **kw):
        self.output_sizes = list(output_sizes)
        super().__init__(input_size, sum(output_sizes), bias, **kw)

    def weight_loader(self, param, loaded_weight, loaded_weight_id: int = 0):
        off = sum(self.output_sizes[:loaded_weight_id]) // self.tp_size
        size = self.output_sizes[loaded_weight_id] // self.tp_size
        dst = param.data.narrow(0, off, size)
        dst.copy_(loaded_weight.narrow(0, self.tp_rank * size, size))

    def forward_swiglu(self, x):
        """``silu(gate(x)) * up(x)`` for a fused [gate; up] pair, without materialising the
        [T, 2I] GEMM output when the fused decode kernel applies."""
        assert len(self.output_sizes) == 2 and self.output_sizes[0] == self.output_sizes[1]
        return ops.swiglu_linear(x, self.weight, self.bias)


class QKVColumnParallelLinear(ColumnParallelLinear):
    """Fused q|k|v projection; heads are sharded across ranks."""

    def __init__(self, input_size: int, head_size: int, num_heads: int, num_kv_heads: int,
                 bias: bool = False, **kw):
        tp_size = kw.get("tp_size") or pg.tp_size()
        self.head_size = head_size
        self.total_num_heads = num_heads
        self.total_num_kv_heads = num_kv_heads
        self.num_heads = _divide(num_heads, tp_size)
        self.num_kv_heads = _divide(num_kv_heads, tp_size)
        super().__init__(input_size, (num_heads + 2 * num_kv_heads) * head_size, bias, **kw)

    def weight_loader(self, param, loaded_weight, loaded_weight_id: str = "q"):
        q = self.num_heads * self.head_size
        kv = self.num_kv_heads * self.head_size
        if loaded_weight_id == "q":
            off, size = 0, q
        elif loaded_weight_id == "k":
            off, size = q, kv
        elif loaded_weight_id == "v":
            off, size = q + kv, kv
        else:
            raise ValueError(f"shard id must be 'q', 'k' or 'v', got {loaded_weight_id!r}")
        dst = param.data.narrow(0, off, size)
        dst.copy_(loaded_weight.narrow(0, self.tp_rank * size, size))


QKVParallelLinear = QKVColumnParallelLinear


class RowParallelLinear(LinearBase):
    """Input features sharded: weight [out, in/tp]; partial sums are all-reduced."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False,
                 reduce_results: bool = True, **kw):
        tp_size = kw.get("tp_size") or pg.tp_size()
        super().__init__(_divide(input_size, tp_size), output_size, bias, tp_dim=1, **kw)
        self.reduce_results = reduce_results

    def weight_loader(self, param, loaded_weight, *args):
        if param.data.dim() == 1:                       # bias is replicated
            param.data.copy_(loaded_weight)
            return
        shard = param.data.size(1)
        param.data.copy_(loaded_weight.narrow(1, self.tp_rank * shard, shard))

    def forward(self, x, defer_reduce: bool = False, residual=None):
        """``defer_reduce``: the caller hands the result to ``ops.norm_linear`` / ``RMSNorm``
        with a residual and to nothing else.  Under the peer-to-peer all-reduce the partial sum
        of a decode-sized output is then returned as ``ops.PendingReduce`` and that consumer
        reduces, adds and normalises in one kernel; a prefill-sized output leaves as a
        reduce-scatter (chunk-pipelined against the GEMM) and the consumer norms this rank's rows
        only before gathering them (``pg.can_defer_reduce``).  The same flag lets a K-split decode GEMM
        hand its unsummed slices (``ops.SplitKPartials``) on: to the norm on a single rank, to
        the reducing kernel (which sums them while staging) otherwise.  ``residual`` (one rank):
        the stream that norm will add the result to, so that a prefill-sized projection can add it
        in its own epilogue (``ops.ResidualSum``)."""
        bias = self.bias if self.tp_rank == 0 else None
        if (self.tp_size > 1 or pg.collectives_active()) and self.reduce_results:
            from minivllm_amd.parallel.overlap import row_parallel_linear, row_parallel_linear_scatter
            how = pg.can_defer_reduce(x.shape[0], self.weight.shape[0], x.dtype, x.is_cuda) \
                if defer_reduce and x.dim() == 2 else ""
            if how == "p2p":
                return ops.PendingReduce(ops.linear(x, self.weight, bias, to_norm=bias is None))
            if how == "scatter":         # prefill-sized: the sum leaves as a reduce-scatter
                return ops.PendingReduce(scattered=row_parallel_linear_scatter(x, self.weight, bias))
            return row_parallel_linear(x, self.weight, bias)
        return ops.linear(x, self.weight, bias, to_norm=defer_reduce and x.dim() == 2,
                          residual=residual if torch.is_tensor(residual) else None)
