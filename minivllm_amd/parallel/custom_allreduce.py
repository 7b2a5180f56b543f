"""Peer-to-peer all-reduce for decode-sized tensors (``csrc/custom_allreduce.hip``).

The reference all-reduces every row-parallel output with NCCL on the compute stream
(reference: src/myvllm/layers/linear.py:216-221).  RCCL's ring needs 2 (P - 1) dependent hops,
which for the [B, hidden] messages of a decode step (tens of KB to a few MB, 2 per layer) is pure
latency.  The 8 GPUs of an MI355X node are fully connected by point-to-point xGMI links, so each
rank can instead read all peers' buffers at once: one kernel per rank, one- or two-shot.

Set-up (once per engine): every rank allocates a flag block, a staging buffer and a result buffer
outside the PyTorch caching allocator, publishes their IPC handles through the process group and
maps the peers' allocations.  ``all_reduce(x)`` then launches the kernel on the current stream;
the call is hipGraph-capturable (pointers are fixed, the call counter lives in device memory).
Messages that do not fit (too large, odd sizes, fp32) go to RCCL.

``all_reduce_add_rms_norm`` has the consumer folded in: the reference runs all-reduce, then a
fused residual-add + RMSNorm kernel after every row-parallel projection (reference:
src/myvllm/models/qwen3.py:191-195,216); the reducing kernel here owns whole rows, so it adds the
residual stream and normalises the row itself and the second kernel disappears (two launches fewer
per decoder layer).  Up to ``ONE_SHOT_MAX_BYTES`` every rank reduces every row (one-shot).  Above
(a 256-row batch of Qwen3-8B is 2 MB) rank r reduces and normalises rows [r R, (r + 1) R) only and
the ranks gather each other's normalised rows; the residual stream stays row-sharded
(``ops.ResidualShard``) from one fused norm to the next, so it never crosses the links.  Either form
also takes the producing GEMM's unsummed K-slice partials (``ops.SplitKPartials``) and sums them
while staging.

``loopback=True`` (shard rehearsal, one process): all "peers" are this rank's own buffers and every
barrier lane raises the flag its peer would raise; data paths, fences and launch shapes are the real
ones, the sums are not.

Opt-in: ``EngineConfig.custom_all_reduce`` / ``MINIVLLM_CUSTOM_AR=1``.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional

import torch
import torch.distributed as dist

# Above this the two-shot form (which moves fewer bytes per link but has one barrier more) is
# used.  The crossover has not been measured on a real xGMI node yet: override with
# MINIVLLM_CAR_ONE_SHOT_MAX_BYTES or EngineConfig.car_one_shot_max_bytes for a first run.
ONE_SHOT_MAX_BYTES = int(os.environ.get("MINIVLLM_CAR_ONE_SHOT_MAX_BYTES", 256 << 10))
DEFAULT_MAX_BYTES = 8 << 20             # [1024, 4096] bf16
# longest wait for a peer inside one kernel before the block gives up (status word set, output
# poisoned with NaN); MINIVLLM_CAR_TIMEOUT_MS
DEFAULT_TIMEOUT_MS = int(os.environ.get("MINIVLLM_CAR_TIMEOUT_MS", 2000))


class PeerAllReduceError(RuntimeError):
    """A peer-to-peer all-reduce kernel gave up waiting for a peer: activations of that step
    (and everything computed from them) are invalid."""


class CustomAllReduce:
    def __init__(self, group, rank: int, world_size: int, device: torch.device,
                 max_bytes: int = DEFAULT_MAX_BYTES, one_shot_max_bytes: Optional[int] = None,
                 timeout_ms: Optional[int] = None, loopback: bool = False):
        from minivllm_amd.ops import hip
        self.loopback = loopback
        self.one_shot_max_bytes = one_shot_max_bytes or ONE_SHOT_MAX_BYTES
        if world_size not in (2, 4, 8):
            raise ValueError("custom all-reduce supports 2, 4 or 8 ranks")
        self._lib = lib = hip._lib
        self.group, self.rank, self.world, self.device = group, rank, world_size, device
        self.max_bytes = max_bytes
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        lib.mvl_car_flags_bytes.restype = sz
        lib.mvl_car_alloc.argtypes = [ctypes.POINTER(vp), sz, ctypes.c_int]
        lib.mvl_car_free.argtypes = [vp]
        lib.mvl_car_ipc_get.argtypes = [vp, ctypes.c_char_p]
        lib.mvl_car_ipc_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
        lib.mvl_car_ipc_close.argtypes = [vp]
        lib.mvl_car_status.argtypes = [vp]
        lib.mvl_car_all_reduce.argtypes = [ctypes.POINTER(vp)] * 3 + [
            ctypes.c_int, ctypes.c_int, vp, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_long, vp]
        lib.mvl_car_all_reduce_norm.argtypes = [ctypes.POINTER(vp)] * 3 + [
            ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int, vp, vp, vp, ctypes.c_int,
            ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_long, vp]
        lib.mvl_car_set_loopback.argtypes = [ctypes.c_int]
        lib.mvl_car_set_loopback.restype = None
        lib.mvl_car_set_loopback(int(loopback))
        lib.mvl_car_set_timeout_ms.argtypes = [ctypes.c_long]
        lib.mvl_car_set_timeout_ms.restype = None
        lib.mvl_car_set_timeout_ms(int(timeout_ms or DEFAULT_TIMEOUT_MS))
        self._hbytes = lib.mvl_car_ipc_handle_bytes()
        torch.cuda.set_device(device)
        # flags, staging, result: fine-grained (uncached) allocations, coherent across the GPUs
        # (the staging buffer is two halves that consecutive calls alternate between: no call has
        # to wait for its peers to finish reading before it returns)
        assert max_bytes % 16 == 0
        sizes = [lib.mvl_car_flags_bytes(), 2 * max_bytes, max_bytes]
        self._own: List[int] = []
        self._mapped: List[int] = []
        handles, err = [], None
        try:
            for nbytes in sizes:
                p = vp()
                self._ck(lib.mvl_car_alloc(ctypes.byref(p), nbytes, 1), "alloc")
                self._own.append(p.value)
                buf = ctypes.create_string_buffer(self._hbytes)
                self._ck(lib.mvl_car_ipc_get(p, buf), "ipc_get")
                handles.append(buf.raw)
        except Exception as e:
            err = repr(e)
        # every rank takes part in both exchanges whatever happened locally, and all ranks give
        # up together: a rank that raised on its own would leave the others in a collective
        everyone = self._exchange((err, handles))
        self._raise_if_any_failed([e for e, _ in everyone])
        tables = []
        try:
            for k in range(3):
                arr = (vp * world_size)()
                for r in range(world_size):
                    if r == rank or loopback:
                        arr[r] = self._own[k]
                    else:
                        q = vp()
                        self._ck(lib.mvl_car_ipc_open(everyone[r][1][k], ctypes.byref(q)), "ipc_open")
                        self._mapped.append(q.value)
                        arr[r] = q.value
                tables.append(arr)
            torch.cuda.synchronize()
        except Exception as e:
            err = repr(e)
        self._raise_if_any_failed(self._exchange(err))   # also: every mapping exists from here on
        self._flags, self._stage, self._result = tables
        self.calls = 0
        self.fused_norm_calls = 0

    def _exchange(self, obj) -> list:
        out: List[Optional[object]] = [None] * (1 if self.loopback else self.world)
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def _raise_if_any_failed(self, errors: list) -> None:
        bad = [(r, e) for r, e in enumerate(errors) if e is not None]
        if bad:
            self._release()
            raise RuntimeError(f"peer-to-peer all-reduce set-up failed on rank(s) {bad}")

    def _release(self) -> None:
        for p in self._mapped:
            self._lib.mvl_car_ipc_close(p)
        for p in self._own:
            self._lib.mvl_car_free(p)
        self._mapped, self._own = [], []

    @staticmethod
    def _ck(rc: int, what: str) -> None:
        if rc != 0:
            raise RuntimeError(f"custom all-reduce: {what} failed with HIP error {rc}")

    def fits(self, x: torch.Tensor) -> bool:
        n = x.numel()
        return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.is_contiguous()
                and 0 < n * 2 <= self.max_bytes and n % (8 * self.world) == 0
                and x.data_ptr() % 16 == 0)

    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        """In-place sum over the group; every rank ends with bit-identical values."""
        two_shot = x.numel() * 2 > self.one_shot_max_bytes
        rc = self._lib.mvl_car_all_reduce(
            self._flags, self._stage, self._result, self.world, self.rank, x.data_ptr(),
            x.numel(), int(two_shot), 0 if x.dtype == torch.bfloat16 else 1, self.max_bytes,
            torch.cuda.current_stream().cuda_stream)
        self._ck(rc, "launch")
        self.calls += 1
        return x

    def norm_mode(self, x, residual, gamma: torch.Tensor) -> Optional[str]:
        """How ``all_reduce_add_rms_norm`` would run for these operands: ``"one"`` (one-shot, full
        residual), ``"two"`` (two-shot, row-sharded residual) or None (not as one kernel).  ``x``
        is this rank's [rows, hidden] partial sum or its ``SplitKPartials``; ``residual`` the full
        [rows, hidden] stream or the ``ResidualShard`` an earlier two-shot call returned."""
        from minivllm_amd.ops.partials import ResidualShard, SplitKPartials
        if isinstance(x, SplitKPartials):
            if x.bias is not None or not x.ws.is_contiguous() or x.ws.data_ptr() % 16:
                return None
            (rows, hidden), dtype = x.shape, x.dtype
        else:
            if x.dim() != 2 or not x.is_contiguous() or x.data_ptr() % 16:
                return None
            (rows, hidden), dtype = x.shape, x.dtype
        nbytes = rows * hidden * 2
        if (dtype not in (torch.bfloat16, torch.float16) or not 0 < nbytes <= self.max_bytes
                or hidden % 8 or hidden > 32768 or gamma.dtype != dtype or not gamma.is_contiguous()
                or gamma.numel() != hidden or gamma.data_ptr() % 16):
            return None
        if isinstance(residual, ResidualShard):
            r = residual.rows
            ok = (residual.total_rows == rows and rows % self.world == 0
                  and tuple(r.shape) == (rows // self.world, hidden))
            return "two" if ok and r.dtype == dtype and r.is_contiguous() and r.data_ptr() % 16 == 0 else None
        if (tuple(residual.shape) != (rows, hidden) or residual.dtype != dtype
                or not residual.is_contiguous() or residual.data_ptr() % 16):
            return None
        if nbytes <= self.one_shot_max_bytes:
            return "one"
        return "two" if rows % self.world == 0 else None

    def fits_norm(self, x: torch.Tensor, residual: torch.Tensor, gamma: torch.Tensor) -> bool:
        """Whether the ONE-SHOT fused form takes these operands."""
        return self.norm_mode(x, residual, gamma) == "one"

    def all_reduce_add_rms_norm(self, x, residual, gamma: torch.Tensor, eps: float,
                                two_shot: bool = False):
        """``r = sum over ranks of x + residual``; ``residual <- r``; returns ``(rms_norm(r) *
        gamma, residual)`` from one kernel, with the bits of ``all_reduce`` followed by the fused
        add + norm kernel.  ``x`` (tensor: overwritten with the result; ``SplitKPartials``: summed
        while staged) and ``residual`` must satisfy ``norm_mode``.  ``two_shot``: this rank only
        handles its rows of the residual stream; the second value is then an
        ``ops.ResidualShard`` (a view of this rank's rows when ``residual`` came in whole)."""
        from minivllm_amd.ops.partials import ResidualShard, SplitKPartials
        if isinstance(x, SplitKPartials):
            rows, hidden = x.shape
            out = torch.empty((rows, hidden), dtype=x.dtype, device=x.ws.device)
            xp, parts, nparts, dtype = None, x.ws.data_ptr(), x.ws.shape[0], x.dtype
        else:
            rows, hidden = x.shape
            out, xp, parts, nparts, dtype = x, x.data_ptr(), None, 0, x.dtype
        if two_shot:
            if not isinstance(residual, ResidualShard):
                per = rows // self.world
                residual = ResidualShard(residual[self.rank * per:(self.rank + 1) * per], rows)
            res_ptr = residual.rows.data_ptr()
        else:
            res_ptr = residual.data_ptr()
        rc = self._lib.mvl_car_all_reduce_norm(
            self._flags, self._stage, self._result, self.world, self.rank, xp, parts, nparts,
            out.data_ptr(), res_ptr, gamma.data_ptr(), rows, hidden, float(eps), int(two_shot),
            0 if dtype == torch.bfloat16 else 1, self.max_bytes,
            torch.cuda.current_stream().cuda_stream)
        self._ck(rc, "launch")
        self.calls += 1
        self.fused_norm_calls += 1
        return out, residual

    def self_test(self) -> bool:
        """All-reduce known integer patterns through both data paths and compare with the exact
        sums (collective call; the result is this rank's own verdict)."""
        P, ok = self.world, True
        if self.loopback:
            # every "peer" is this rank: a one-shot sum is P times the input; the two-shot and
            # fused forms only have to complete (their barriers are what is being rehearsed)
            x = (torch.arange(8 * P * 16, device=self.device) % 7).to(torch.bfloat16)
            want = x.float() * P
            ok = bool((self.all_reduce(x.clone()).float() == want).all())
            big = torch.ones(self.one_shot_max_bytes // 2 + 8 * P * 64, device=self.device,
                             dtype=torch.bfloat16)
            big = big[:big.numel() - big.numel() % (8 * P)]
            if big.numel() * 2 <= self.max_bytes:
                self.all_reduce(big)
            return ok and self.healthy()
        for n in (8 * P * 16, (self.one_shot_max_bytes // 2 + 8 * P * 64)):
            n -= n % (8 * P)
            if n * 2 > self.max_bytes:
                continue
            base = (torch.arange(n, device=self.device) % 7).to(torch.bfloat16)
            for _ in range(2):                                  # twice: epochs advance
                x = base + float(self.rank)
                self.all_reduce(x)
                want = base.float() * P + P * (P - 1) / 2
                ok = ok and bool((x.float() == want).all())
        # the fused form against the two-kernel route it replaces (same bits expected)
        from minivllm_amd.ops import hip
        g = torch.Generator(device=self.device).manual_seed(4321)
        base = torch.randn(5, 1024, device=self.device, dtype=torch.bfloat16, generator=g)
        res0 = torch.randn(5, 1024, device=self.device, dtype=torch.bfloat16, generator=g)
        gamma = torch.rand(1024, device=self.device, dtype=torch.bfloat16, generator=g) + 0.5
        x1, r1 = base * (self.rank + 1), res0.clone()
        x2, r2 = x1.clone(), res0.clone()
        self.all_reduce(x1)
        h1, r1 = hip.add_rms_norm(x1, r1, gamma, 1e-6)
        h2, r2 = self.all_reduce_add_rms_norm(x2, r2, gamma, 1e-6)
        ok = ok and bool(torch.equal(h1, h2)) and bool(torch.equal(r1, r2))
        # ... and the two-shot form (this rank's rows of the residual only) against the same route
        rows = 2 * P
        base = torch.randn(rows, 1024, device=self.device, dtype=torch.bfloat16, generator=g)
        res0 = torch.randn(rows, 1024, device=self.device, dtype=torch.bfloat16, generator=g)
        x1, r1 = base * (self.rank + 1), res0.clone()
        x2, r2 = x1.clone(), res0.clone()
        self.all_reduce(x1)
        h1, r1 = hip.add_rms_norm(x1, r1, gamma, 1e-6)
        h2, shard = self.all_reduce_add_rms_norm(x2, r2, gamma, 1e-6, two_shot=True)
        mine = slice(self.rank * 2, self.rank * 2 + 2)
        ok = ok and bool(torch.equal(h1, h2)) and bool(torch.equal(r1[mine], shard.rows))
        return ok and self.healthy()

    def healthy(self, synchronize: bool = True) -> bool:
        """False once any barrier of any call gave up waiting for a peer.  ``synchronize=False``
        reads the word as of the work that has completed so far (a caller that has just read
        the step's tokens back is already in sync)."""
        if synchronize:
            torch.cuda.synchronize()
        return self._lib.mvl_car_status(self._own[0]) == 0

    def check(self, synchronize: bool = False) -> None:
        """Raise :class:`PeerAllReduceError` if any call so far timed out on a peer."""
        if self._own and not self.healthy(synchronize):
            raise PeerAllReduceError(
                f"rank {self.rank}: a peer-to-peer all-reduce kernel gave up waiting for a peer "
                f"(time limit {DEFAULT_TIMEOUT_MS} ms); the outputs of that step were poisoned "
                f"with NaN.  Restart the engine, or run without custom_all_reduce (RCCL).")

    def close(self) -> None:
        """Collective: every rank of the group calls it (peers may still be reading this rank's
        buffers until all have arrived)."""
        if not self._own:
            return
        torch.cuda.synchronize()
        try:
            dist.barrier(group=self.group)
        except Exception:
            pass
        if self.loopback:
            self._lib.mvl_car_set_loopback(0)
        self._release()
