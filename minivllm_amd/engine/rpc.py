"""Rank-0 -> workers control channel over POSIX shared memory.

The reference ships ``pickle((method, args))`` through a 1 MiB ``SharedMemory`` block and
wakes each worker with an ``mp.Event`` (reference: src/myvllm/engine/model_runner.py:62-89,
106-123).  That protocol has no acknowledgement, so rank 0 can overwrite a message a slow
worker has not read yet, and it needs ``Event`` objects inherited from the parent, which
rules out independently launched ranks (torchrun).

This channel is a single-producer broadcast with sequence numbers and per-reader
acknowledgement words, addressed by *name* so that any process can attach:

    [0]  seq      (u64)  bumped by the writer after the payload is complete
    [1]  nbytes   (u64)
    [2]  pid      (u64)  of the writer: an idle reader checks now and then that it still exists
    [3..3+R) ack  (u64)  reader r stores the last seq it has fully consumed
    payload bytes

The writer waits until every reader has acknowledged message ``seq-1`` before it touches the
payload again.  Payloads are a small pickled header (method name, scalars, array specs)
followed by raw array bytes; batch tensors are never pickled element-wise.
"""
from __future__ import annotations

import os
import pickle
import struct
import time
from multiprocessing import shared_memory
from typing import Any, Dict, Tuple

import numpy as np

_HDR_WORDS = 3


class ChannelTimeout(RuntimeError):
    pass


def encode_message(method: str, scalars: Dict[str, Any], arrays: Dict[str, np.ndarray]) -> bytes:
    specs = []
    blobs = []
    for name, a in arrays.items():
        a = np.ascontiguousarray(a)
        specs.append((name, a.dtype.str, a.shape))
        blobs.append(a.tobytes())
    head = pickle.dumps((method, scalars, specs), protocol=pickle.HIGHEST_PROTOCOL)
    return struct.pack("<I", len(head)) + head + b"".join(blobs)


def decode_message(buf) -> Tuple[str, Dict[str, Any], Dict[str, np.ndarray]]:
    mv = memoryview(buf)
    (n,) = struct.unpack_from("<I", mv, 0)
    method, scalars, specs = pickle.loads(mv[4:4 + n])
    off = 4 + n
    arrays = {}
    for name, dt, shape in specs:
        dtype = np.dtype(dt)
        count = int(np.prod(shape)) if len(shape) else 1
        nbytes = count * dtype.itemsize
        arrays[name] = np.frombuffer(mv[off:off + nbytes], dtype=dtype).reshape(shape).copy()
        off += nbytes
    return method, scalars, arrays


def _attach_untracked(name: str) -> shared_memory.SharedMemory:
    """Attach to an existing segment without telling this process's resource tracker about it.
    Only the creator owns the segment's lifetime: a tracked reader would unlink it when it exits
    (own tracker, ranks started by a launcher) or make the shared tracker log KeyErrors for the
    second and later unregistrations (workers spawned by the front end).  Python >= 3.13 has
    ``track=False`` for this; here registration is switched off for the duration of the call."""
    from multiprocessing import resource_tracker
    orig = resource_tracker.register
    resource_tracker.register = lambda *a, **k: None
    try:
        return shared_memory.SharedMemory(name=name)
    finally:
        resource_tracker.register = orig


class ChannelClosed(RuntimeError):
    """The writer of a control channel has exited without saying goodbye."""


class ShmBroadcastChannel:
    def __init__(self, name: str, num_readers: int, create: bool, size: int = 8 << 20,
                 attach_timeout_s: float = 120.0):
        self.name = name
        self.num_readers = num_readers
        self.is_writer = create
        self._ctrl_bytes = (_HDR_WORDS + num_readers) * 8
        if create:
            try:   # a stale segment from a crashed run would carry old sequence numbers
                old = shared_memory.SharedMemory(name=name)
                old.close()
                old.unlink()
            except FileNotFoundError:
                pass
            self.shm = shared_memory.SharedMemory(name=name, create=True,
                                                  size=self._ctrl_bytes + size)
            self.shm.buf[:self._ctrl_bytes] = bytes(self._ctrl_bytes)
            np.ndarray((3,), dtype=np.uint64, buffer=self.shm.buf[:24])[2] = os.getpid()
        else:
            deadline = time.monotonic() + attach_timeout_s
            while True:
                try:
                    self.shm = _attach_untracked(name)
                    break
                except FileNotFoundError:
                    if time.monotonic() > deadline:
                        raise ChannelTimeout(f"control channel {name!r} never appeared")
                    time.sleep(0.01)
        self._ctrl = np.ndarray((_HDR_WORDS + num_readers,), dtype=np.uint64,
                                buffer=self.shm.buf[:self._ctrl_bytes])
        self._payload = self.shm.buf[self._ctrl_bytes:]
        self._seq = 0
        self.capacity = len(self._payload)
        # writer side, optional: callable that says whether every reader process is still alive
        self.liveness = None

    # ---------------------------------------------------------------- writer
    def _wait_acks(self, seq: int, timeout_s: float) -> None:
        deadline = time.monotonic() + timeout_s
        next_alive_check = time.monotonic() + 0.2
        spins = 0
        while True:
            if all(int(self._ctrl[_HDR_WORDS + r]) >= seq for r in range(self.num_readers)):
                return
            spins += 1
            if spins > 200:
                time.sleep(0.0002)
            now = time.monotonic()
            if now > deadline:
                raise ChannelTimeout(f"workers did not acknowledge message {seq}")
            # a reader that died will never acknowledge: do not sit out the whole timeout
            if self.liveness is not None and now > next_alive_check:
                next_alive_check = now + 0.2
                if not self.liveness():
                    raise ChannelTimeout(f"a worker died before acknowledging message {seq}")

    def send(self, method: str, scalars: Dict[str, Any] = None,
             arrays: Dict[str, np.ndarray] = None, timeout_s: float = 300.0) -> None:
        assert self.is_writer
        data = encode_message(method, scalars or {}, arrays or {})
        if len(data) > self.capacity:
            raise ValueError(f"control message of {len(data)} bytes exceeds the channel "
                             f"capacity of {self.capacity}")
        self._wait_acks(self._seq, timeout_s)            # previous message consumed by all
        self._payload[:len(data)] = data
        self._ctrl[1] = len(data)
        self._seq += 1
        self._ctrl[0] = self._seq                        # publish

    # ---------------------------------------------------------------- reader
    def recv(self, reader_index: int, timeout_s: float = None):
        deadline = None if timeout_s is None else time.monotonic() + timeout_s
        want = self._seq + 1
        spins = 0
        next_alive_check = time.monotonic() + 1.0
        while int(self._ctrl[0]) < want:
            spins += 1
            if spins > 2000:
                time.sleep(0.0001)
                now = time.monotonic()
                if deadline is not None and now > deadline:
                    raise ChannelTimeout("no control message")
                if now > next_alive_check:           # an idle reader: is the writer still there?
                    next_alive_check = now + 1.0
                    if not self.writer_alive():
                        raise ChannelClosed("the driving rank's process is gone")
        n = int(self._ctrl[1])
        msg = decode_message(self._payload[:n])
        self._seq = want
        self._ctrl[_HDR_WORDS + reader_index] = want     # acknowledge
        return msg

    def writer_alive(self) -> bool:
        pid = int(self._ctrl[2])
        if pid == 0:
            return True
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        except PermissionError:
            return True
        return True

    def close(self) -> None:
        self._ctrl = None
        self._payload = None
        try:
            self.shm.close()
        except BufferError:
            pass
        if self.is_writer:
            try:
                self.shm.unlink()
            except FileNotFoundError:
                pass
