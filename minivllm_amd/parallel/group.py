"""Tensor-parallel process group: one process per GPU, RCCL over xGMI.

The reference creates an NCCL group on a fixed ``tcp://localhost:12345`` even for
a single GPU and layers query ``dist.get_rank()`` in their constructors
(reference: src/myvllm/engine/model_runner.py:24, layers/linear.py:20-21).  Here
the group is explicit state: ``world_size == 1`` never touches
``torch.distributed`` (which is also what lets the engine run on a CPU-only
machine), the rendezvous address is configurable, and the backend is ``nccl``
(= RCCL on ROCm) for GPU ranks and ``gloo`` for CPU ranks.
"""
from __future__ import annotations

import datetime
import os
from typing import List, Optional

import torch
import torch.distributed as dist

_TP_RANK = 0
_TP_SIZE = 1
_TP_GROUP: Optional[dist.ProcessGroup] = None
# Shard rehearsal: this process holds rank ``_TP_RANK``'s shard of a ``_TP_SIZE``-way sharded model
# but its process group has ONE member.  Every collective of the sharded code path is issued (RCCL
# nodes inside the decode graphs included) with nobody at the other end, so what a step costs is
# one rank's compute, launches and collective entry/exit: the floor under that rank's step time
# on a real node.  Outputs are not the model's (partial sums are never completed).
_REHEARSAL = False
# What a rehearsed collective does locally.  "compute" (default): only what the data flow needs
# (a reduce-scatter keeps this rank's rows, an all-gather fills this rank's slice): the step is
# then one rank's compute and launches, nothing else.  "traffic": an all-gather also writes the
# absent peers' slices (copies of this rank's), i.e. the local memory writes of the real
# collective are paid as well.  MINIVLLM_REHEARSAL_COLLECTIVES.
_REHEARSAL_TRAFFIC = os.environ.get("MINIVLLM_REHEARSAL_COLLECTIVES", "compute") == "traffic"
_OWNS_DEFAULT_GROUP = False
# gloo has no device transport: with GPU tensors (several ranks sharing one device, used only to
# rehearse the TP control flow on a single-GPU machine) every collective is staged through host
# memory HERE - copy to the host, CPU collective, copy back - instead of by gloo's own
# device-tensor path, which can deadlock when two processes share one GPU.  Never set with RCCL.
_HOST_STAGED = False
# optional peer-to-peer all-reduce for decode-sized tensors (parallel/custom_allreduce.py)
_CUSTOM_AR = None


def tp_rank() -> int:
    return _TP_RANK


def tp_size() -> int:
    return _TP_SIZE


def tp_group():
    return _TP_GROUP


def is_initialized() -> bool:
    return _TP_SIZE > 1 and _TP_GROUP is not None


def collectives_active() -> bool:
    """True when the sharded code path (collectives included) must run: more than one rank, or
    a forced single-rank group."""
    return _TP_GROUP is not None


def host_staged_collectives() -> bool:
    return _HOST_STAGED


def is_rehearsal() -> bool:
    """True when one process stands in for a rank of a larger tensor-parallel group."""
    return _REHEARSAL


def init_tensor_parallel(rank: int, world_size: int, backend: str = "auto",
                         master_addr: str = "127.0.0.1", master_port: int = 29512,
                         device: Optional[torch.device] = None,
                         timeout_s: float = 600.0, force_group: bool = False,
                         shard_of: int = 0, shard_rank: int = 0) -> None:
    """Join the TP group.  Reuses an already initialised default group (torchrun).
    ``force_group`` creates the group even for one rank, so that every collective of the
    sharded code path really runs (and is captured into the decode graphs) on a single GPU.
    ``shard_of`` = N (one process only): layers are built as rank ``shard_rank`` of an N-way
    sharded model while the process group has this one member (shard rehearsal)."""
    global _TP_RANK, _TP_SIZE, _TP_GROUP, _OWNS_DEFAULT_GROUP, _HOST_STAGED, _REHEARSAL
    _TP_RANK, _TP_SIZE = rank, world_size
    _REHEARSAL = False
    if shard_of > 1:
        if world_size != 1:
            raise ValueError("a shard rehearsal runs in one process (world_size 1)")
        force_group = True
    if world_size == 1 and not force_group:
        _TP_GROUP = None
        return
    on_gpu = device is not None and torch.device(device).type == "cuda"
    if backend == "auto":
        backend = "nccl" if on_gpu else "gloo"
    if not dist.is_initialized():
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = torch.device(device)
        # under torchrun the launcher already hosts the rendezvous store on MASTER_PORT:
        # join it (env://) instead of opening a second store on the same port
        under_launcher = ("TORCHELASTIC_RUN_ID" in os.environ and "MASTER_ADDR" in os.environ
                          and int(os.environ.get("WORLD_SIZE", "1")) == world_size)
        init_method = "env://" if under_launcher else f"tcp://{master_addr}:{master_port}"
        dist.init_process_group(
            backend=backend, init_method=init_method, world_size=world_size, rank=rank,
            timeout=datetime.timedelta(seconds=timeout_s), **kwargs)
        _OWNS_DEFAULT_GROUP = True
    else:
        assert dist.get_world_size() == world_size and dist.get_rank() == rank
    _TP_GROUP = dist.group.WORLD
    _HOST_STAGED = on_gpu and dist.get_backend(_TP_GROUP) == "gloo"
    if shard_of > 1:
        _TP_RANK, _TP_SIZE, _REHEARSAL = shard_rank, shard_of, True


def enable_custom_all_reduce(device, max_bytes: Optional[int] = None,
                             one_shot_max_bytes: Optional[int] = None,
                             timeout_ms: Optional[int] = None) -> bool:
    """Route all-reduces of tensors that fit through the peer-to-peer kernel.  Collective call:
    every rank of the group must make it.  Returns False (and keeps RCCL) when the group size is
    not 2/4/8 or the set-up fails on ANY rank."""
    global _CUSTOM_AR
    if _TP_GROUP is None or _TP_SIZE not in (2, 4, 8) or _CUSTOM_AR is not None:
        return _CUSTOM_AR is not None
    from .custom_allreduce import DEFAULT_MAX_BYTES, CustomAllReduce
    try:
        # the constructor fails on every rank or on none (it exchanges its own status)
        car = CustomAllReduce(_TP_GROUP, _TP_RANK, _TP_SIZE, torch.device(device),
                              max_bytes or DEFAULT_MAX_BYTES, one_shot_max_bytes, timeout_ms,
                              loopback=_REHEARSAL)
        verdicts = [None] * (1 if _REHEARSAL else _TP_SIZE)
        dist.all_gather_object(verdicts, car.self_test(), group=_TP_GROUP)
        if not all(verdicts):                  # wrong sums or a timed-out barrier on some rank
            car.close()
            raise RuntimeError(f"self-test failed (per-rank verdicts {verdicts})")
        _CUSTOM_AR = car
        return True
    except RuntimeError as e:                  # e.g. IPC not permitted between these devices
        if _TP_RANK == 0:
            print(f"[tp] peer-to-peer all-reduce unavailable ({e}); using the backend's",
                  flush=True)
        return False


def custom_all_reduce():
    return _CUSTOM_AR


def check_custom_all_reduce(synchronize: bool = False) -> None:
    """Raise if the peer-to-peer all-reduce ever timed out on a peer (no-op when it is off)."""
    if _CUSTOM_AR is not None:
        _CUSTOM_AR.check(synchronize)


def set_tp_for_testing(rank: int, world_size: int) -> None:
    """Shard-math tests build layers for an arbitrary (rank, size) without any
    communication backend."""
    global _TP_RANK, _TP_SIZE, _TP_GROUP
    _TP_RANK, _TP_SIZE, _TP_GROUP = rank, world_size, None


def destroy_tensor_parallel(peers_dead: bool = False) -> None:
    """Leave the TP group.  ``peers_dead``: another rank is known to be gone, so nothing that
    needs its cooperation is attempted (the peer-to-peer buffers are released without the
    closing barrier, the process group is left to die with the process)."""
    global _TP_RANK, _TP_SIZE, _TP_GROUP, _OWNS_DEFAULT_GROUP, _CUSTOM_AR
    if _CUSTOM_AR is not None:
        if peers_dead:
            _CUSTOM_AR._release()
        else:
            _CUSTOM_AR.close()
        _CUSTOM_AR = None
    keep = os.environ.get("MINIVLLM_KEEP_PG") == "1"      # another engine will reuse it
    if (_TP_GROUP is not None and _OWNS_DEFAULT_GROUP and dist.is_initialized() and not keep
            and not peers_dead):
        dist.destroy_process_group()
    _TP_RANK, _TP_SIZE, _TP_GROUP, _OWNS_DEFAULT_GROUP = 0, 1, None, False
    global _HOST_STAGED, _REHEARSAL
    _HOST_STAGED = False
    _REHEARSAL = False


# ------------------------------------------------------------- collectives
def tp_all_reduce(x: torch.Tensor) -> torch.Tensor:
    """In-place sum over the TP group; returns ``x`` (the reference returns the
    collective's return value instead of the tensor: layers/linear.py:218-219)."""
    if _TP_GROUP is not None:
        if _CUSTOM_AR is not None and _CUSTOM_AR.fits(x):
            return _CUSTOM_AR.all_reduce(x)
        if _HOST_STAGED and x.is_cuda:
            h = x.cpu()
            dist.all_reduce(h, group=_TP_GROUP)
            x.copy_(h)
        else:
            dist.all_reduce(x, group=_TP_GROUP)
    return x


# From this many rows on a row-parallel output that feeds (residual add +) RMSNorm is reduced as
# reduce-scatter -> add + norm on the rank's own rows -> all-gather of the normalised rows
# instead of all-reduce -> add + norm on every row on every rank: the same bytes on the links,
# 1 / tp of the norm's HBM passes (at 16384 x 4096 the replicated norms are 5.5 ms of a 30 ms
# Qwen3-8B TP=8 prefill step per rank), and the residual stream is never sent at all.
SP_MIN_ROWS = int(os.environ.get("MINIVLLM_SP_MIN_ROWS", 512))
SCATTER_NORM_CALLS = 0          # how often this process took that route (tests, bench JSON)


def configure_scatter_norm(min_rows: Optional[int]) -> None:
    global SP_MIN_ROWS
    if min_rows is not None:
        SP_MIN_ROWS = int(min_rows)


def scatter_norm_ok(rows: int) -> bool:
    return _TP_GROUP is not None and rows >= SP_MIN_ROWS and rows % _TP_SIZE == 0


def can_defer_reduce(rows: int, hidden: int, dtype: torch.dtype, on_gpu: bool) -> str:
    """How the all-reduce of a [rows, hidden] row-parallel partial sum may be left to the
    (residual add + RMSNorm) that consumes it: ``"p2p"`` (the peer-to-peer kernels are active
    and the message fits their buffers: the consumer runs reduce + add + norm as one kernel when
    the operands allow), ``"scatter"`` (prefill-sized: reduce-scatter, norm on own rows,
    all-gather) or ``""`` (all-reduce now).  Every case ends in the same values."""
    if _TP_GROUP is None:
        return ""
    if (_CUSTOM_AR is not None and on_gpu and dtype in (torch.bfloat16, torch.float16)
            and 0 < rows * hidden * 2 <= _CUSTOM_AR.max_bytes and hidden % 8 == 0):
        return "p2p"
    return "scatter" if scatter_norm_ok(rows) else ""


def reduce_scatter_rows(out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
    """``out`` [rows / tp, hidden] <- this rank's share (rank order) of the sum over ranks of
    ``inp`` [rows, hidden].  Returns the collective's work handle when ``async_op`` (else None)."""
    per = inp.shape[0] // _TP_SIZE
    mine = slice(_TP_RANK * per, (_TP_RANK + 1) * per)
    if _REHEARSAL:                                    # nobody to add: this rank's own rows
        out.copy_(inp[mine])
        return None
    if (_HOST_STAGED and inp.is_cuda) or dist.get_backend(_TP_GROUP) == "gloo":
        h = inp.cpu() if inp.is_cuda else inp.clone()          # gloo has no reduce-scatter
        dist.all_reduce(h, group=_TP_GROUP)
        out.copy_(h[mine])
        return None
    w = dist.reduce_scatter_tensor(out, inp, group=_TP_GROUP, async_op=async_op)
    return w if async_op else None


def all_gather_rows(out: torch.Tensor, inp: torch.Tensor) -> None:
    """``out`` [rows, hidden] <- every rank's ``inp`` [rows / tp, hidden], in rank order."""
    per = inp.shape[0]
    if _REHEARSAL:
        if _REHEARSAL_TRAFFIC:                        # the absent peers' rows: copies of this rank's
            out.view(_TP_SIZE, per, -1).copy_(inp.unsqueeze(0).expand(_TP_SIZE, per, inp.shape[1]))
        else:
            out[_TP_RANK * per:(_TP_RANK + 1) * per].copy_(inp)
        return
    if (_HOST_STAGED and inp.is_cuda) or dist.get_backend(_TP_GROUP) == "gloo":
        parts = _all_gather_parts(inp)
        for r, p in enumerate(parts):
            out[r * per:(r + 1) * per].copy_(p)
        return
    dist.all_gather_into_tensor(out, inp.contiguous(), group=_TP_GROUP)


def _finish_scattered(sc, residual, gamma: torch.Tensor, eps: float):
    """Second half of reduce-scatter -> norm -> all-gather: ``sc.own`` holds (or is about to
    hold) this rank's rows of the sum."""
    from .. import ops
    from ..ops.partials import ResidualShard
    global SCATTER_NORM_CALLS
    SCATTER_NORM_CALLS += 1
    sc.wait()
    own, bounds, T = sc.own, sc.bounds, sc.total_rows
    chunks = list(zip(bounds[:-1], bounds[1:]))

    def my_rows(lo, hi):
        per = (hi - lo) // _TP_SIZE
        return lo + _TP_RANK * per, lo + (_TP_RANK + 1) * per
    if residual is None:
        h_rows, res_rows = ops.rms_norm(own, gamma, eps), own
    else:
        if isinstance(residual, ResidualShard):
            if residual.total_rows != T or residual.bounds != bounds:
                raise RuntimeError("row-sharded residual and scattered projection disagree on who "
                                   "owns which rows")
            res_rows = residual.rows
        else:                                         # whole stream (first norm of a step): my rows
            res_rows = torch.cat([residual[slice(*my_rows(lo, hi))] for lo, hi in chunks]) \
                if len(chunks) > 1 else residual[slice(*my_rows(*chunks[0]))]
        h_rows, res_rows = ops.add_rms_norm(own, res_rows, gamma, eps)
    h = torch.empty((T, own.shape[1]), dtype=own.dtype, device=own.device)
    off = 0
    for lo, hi in chunks:
        per = (hi - lo) // _TP_SIZE
        all_gather_rows(h[lo:hi], h_rows[off:off + per])
        off += per
    return h, ResidualShard(res_rows, T, bounds)


def tp_reduce_add_rms_norm(x, residual, gamma: torch.Tensor, eps: float):
    """Consumer of a ``PendingReduce``: ``(rms_norm(all_reduce(x) + residual) * gamma,
    all_reduce(x) + residual)``.  Decode-sized, peer-to-peer kernels active: one kernel when the
    operands allow it.  Prefill-sized: the producer has reduce-scattered the rows; norm on this
    rank's rows, all-gather of the normalised rows.  Otherwise all-reduce + fused add/norm.  The
    second value is a ``ResidualShard`` in the reduce-scatter shaped cases (and ``residual`` may
    be one from an earlier such call)."""
    from .. import ops
    from ..ops.partials import PendingReduce, ResidualShard, SplitKPartials
    assert isinstance(x, PendingReduce)
    if x.scattered is not None:
        return _finish_scattered(x.scattered, residual, gamma, eps)
    val = x.value
    if residual is not None and _CUSTOM_AR is not None:
        mode = _CUSTOM_AR.norm_mode(val, residual, gamma)
        if mode is not None:
            return _CUSTOM_AR.all_reduce_add_rms_norm(val, residual, gamma, eps, two_shot=mode == "two")
    if isinstance(residual, ResidualShard):
        raise RuntimeError("the residual stream is row-sharded but this reduce cannot run as the "
                           "two-shot fused kernel (shape or alignment changed within a step)")
    t = val.materialize() if isinstance(val, SplitKPartials) else val
    t = tp_all_reduce(t)
    if residual is None:
        return ops.rms_norm(t, gamma, eps), t
    return ops.add_rms_norm(t, residual, gamma, eps)


def _all_gather_parts(x: torch.Tensor) -> List[torch.Tensor]:
    src = x.cpu() if (_HOST_STAGED and x.is_cuda) else x.contiguous()
    n = 1 if _REHEARSAL else _TP_SIZE
    parts: List[torch.Tensor] = [torch.empty_like(src) for _ in range(n)]
    dist.all_gather(parts, src, group=_TP_GROUP)
    if _REHEARSAL:                    # the absent peers' shards: copies of this rank's
        parts = parts + [parts[0].clone() for _ in range(_TP_SIZE - 1)]
    return [p.to(x.device) for p in parts] if src.device != x.device else parts


def tp_all_gather(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    if _TP_GROUP is None:
        return x
    return torch.cat(_all_gather_parts(x), dim=dim)


def tp_all_gather_stack(x: torch.Tensor) -> torch.Tensor:
    """[tp, *x.shape]: every rank's ``x`` stacked along a new leading dim."""
    if _TP_GROUP is None:
        return x.unsqueeze(0)
    if not _HOST_STAGED and x.is_cuda:
        # one output buffer, one collective, no per-rank copies (graph-capture friendly)
        out = torch.empty((_TP_SIZE,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        if _REHEARSAL:                # one member: its slice arrives, the others are copies
            dist.all_gather_into_tensor(out[:1], x.contiguous(), group=_TP_GROUP)
            out[1:].copy_(out[:1].expand_as(out[1:]))
        else:
            dist.all_gather_into_tensor(out, x.contiguous(), group=_TP_GROUP)
        return out
    return torch.stack(_all_gather_parts(x))


def tp_gather(x: torch.Tensor, dst: int = 0, dim: int = -1) -> Optional[torch.Tensor]:
    """Concatenate shards on ``dst``; other ranks get None (reference:
    layers/embedding_head.py:67-75)."""
    if _TP_GROUP is None:
        return x
    src = x.cpu() if (_HOST_STAGED and x.is_cuda) else x.contiguous()
    if _REHEARSAL:
        return torch.cat(_all_gather_parts(x), dim=dim) if _TP_RANK == dst else None
    parts = [torch.empty_like(src) for _ in range(_TP_SIZE)] if _TP_RANK == dst else None
    dist.gather(src, parts, dst=dst, group=_TP_GROUP)
    return torch.cat(parts, dim=dim).to(x.device) if parts is not None else None


def tp_barrier() -> None:
    if _TP_GROUP is not None:
        dist.barrier(group=_TP_GROUP)


def env_rank_world() -> tuple:
    """(rank, local_rank, world_size) from a torchrun-style environment, or (0,0,1)."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        return (int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"])),
                int(os.environ["WORLD_SIZE"]))
    return 0, 0, 1
