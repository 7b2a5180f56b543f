"""Paged KV-cache block allocator with hash-chained prefix caching.

Behaviour follows the reference's ``BlockManager`` (reference:
src/myvllm/engine/block_manager.py:24-139): fixed-size blocks, a chain hash
(xxh64 over the previous block's hash and this block's tokens) identifies a full
block's content *and* its whole prefix, freed blocks keep their hash so a later
request with the same prefix can revive them, and partial blocks are never
shared.  Corrected relative to the reference: a freshly allocated block starts
with ``ref_count == 1`` (the reference leaves it at 0 and frees shared blocks
early), freeing/reviving is O(1) (ordered dict instead of ``deque.remove``), and
a stale hash entry is dropped when its block is reused for other content.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import xxhash

from .sequence import Sequence


class Block:
    __slots__ = ("block_id", "ref_count", "hash", "token_ids")

    def __init__(self, block_id: int):
        self.block_id = block_id
        self.ref_count = 0
        self.hash = -1
        self.token_ids: List[int] = []

    def update(self, h: int, token_ids: List[int]) -> None:
        self.hash = h
        self.token_ids = token_ids

    def reset(self) -> None:
        self.ref_count = 1
        self.hash = -1
        self.token_ids = []


class BlockManager:
    def __init__(self, num_blocks: int, block_size: int):
        if num_blocks <= 0:
            raise ValueError("num_blocks must be positive")
        self.block_size = block_size
        self.blocks: List[Block] = [Block(i) for i in range(num_blocks)]
        self.hash_to_block_id: Dict[int, int] = {}
        # insertion-ordered set: oldest freed block is reused first, so recently
        # freed (still hashed) blocks survive longest for prefix hits
        self.free_block_ids: Dict[int, None] = dict.fromkeys(range(num_blocks))
        self.used_block_ids: set = set()
        # statistics
        self.num_prefix_hits = 0
        self.num_prefix_queries = 0

    # ------------------------------------------------------------------ hash
    @staticmethod
    def compute_hash(token_ids: List[int], prefix_hash: int = -1) -> int:
        h = xxhash.xxh64()
        if prefix_hash != -1:
            h.update(int(prefix_hash).to_bytes(8, "little"))
        h.update(np.asarray(token_ids, dtype=np.int32).tobytes())
        return h.intdigest()

    # ------------------------------------------------------------ primitives
    @property
    def num_free_blocks(self) -> int:
        return len(self.free_block_ids)

    @property
    def num_blocks(self) -> int:
        return len(self.blocks)

    def _take_block(self, block_id: Optional[int] = None) -> Block:
        """Move a block from free to used.  With no id, take the oldest free one."""
        if block_id is None:
            block_id = next(iter(self.free_block_ids))
        del self.free_block_ids[block_id]
        block = self.blocks[block_id]
        assert block.ref_count == 0
        self.used_block_ids.add(block_id)
        return block

    def _fresh_block(self) -> Block:
        block = self._take_block()
        # reusing the block for other content invalidates its cached identity
        if block.hash != -1 and self.hash_to_block_id.get(block.hash) == block.block_id:
            del self.hash_to_block_id[block.hash]
        block.reset()
        return block

    def _release_block(self, block_id: int) -> None:
        assert self.blocks[block_id].ref_count == 0
        self.used_block_ids.remove(block_id)
        self.free_block_ids[block_id] = None      # keeps hash: revivable until reused

    # ---------------------------------------------------------------- prefill
    def can_allocate(self, seq: Sequence) -> bool:
        # conservative: ignores possible prefix hits (same as the reference)
        return len(self.free_block_ids) >= seq.num_blocks

    def allocate(self, seq: Sequence) -> None:
        assert not seq.block_table, "sequence already has blocks"
        bs = self.block_size
        h = -1
        miss = False
        seq.num_cached_tokens = 0
        for i in range(seq.num_blocks):
            token_ids = seq.block(i)
            full = len(token_ids) == bs
            h = self.compute_hash(token_ids, h) if full else -1
            block_id = self.hash_to_block_id.get(h, -1) if full else -1
            if full:
                self.num_prefix_queries += 1
            # a hit needs the same hash AND the same tokens (collision guard); once
            # one block misses every later block must be recomputed too
            if block_id == -1 or miss or self.blocks[block_id].token_ids != token_ids:
                miss = True
            if miss:
                block = self._fresh_block()
            else:
                self.num_prefix_hits += 1
                seq.num_cached_tokens += bs
                block = self.blocks[block_id]
                if block_id in self.used_block_ids:
                    block.ref_count += 1
                else:                                   # freed but content still there
                    block = self._take_block(block_id)
                    block.ref_count = 1
            if full:
                block.update(h, token_ids)
                self.hash_to_block_id[h] = block.block_id
            seq.block_table.append(block.block_id)
        # the last prompt token must be computed to produce logits
        if seq.num_cached_tokens >= seq.num_tokens:
            seq.num_cached_tokens = seq.num_tokens - 1

    def deallocate(self, seq: Sequence, num_valid_tokens: Optional[int] = None) -> None:
        """Give the sequence's blocks back.  ``num_valid_tokens``: only that many leading
        tokens have their KV in the cache (a request dropped in the middle of its prefill,
        or whose step failed): a block beyond them that nobody else holds loses its prefix-cache
        identity, so that a later request with the same prompt recomputes it instead of
        attending to KV that was never written."""
        first_invalid = (len(seq.block_table) if num_valid_tokens is None
                         else max(num_valid_tokens, 0) // self.block_size)
        for i in range(len(seq.block_table) - 1, -1, -1):
            block_id = seq.block_table[i]
            block = self.blocks[block_id]
            block.ref_count -= 1
            assert block.ref_count >= 0
            if block.ref_count == 0:
                if i >= first_invalid and block.hash != -1:
                    if self.hash_to_block_id.get(block.hash) == block_id:
                        del self.hash_to_block_id[block.hash]
                    block.hash = -1
                    block.token_ids = []
                self._release_block(block_id)
        seq.block_table = []
        seq.num_cached_tokens = 0

    # ----------------------------------------------------------------- decode
    def can_append(self, seq: Sequence) -> bool:
        """Is there room for the KV entry of ``seq``'s last token?"""
        need_new = len(seq.block_table) < seq.num_blocks
        return len(self.free_block_ids) >= int(need_new)

    def append(self, seq: Sequence) -> None:
        """Called before the decode step that writes the KV of ``seq.last_token``
        (position ``num_tokens - 1``)."""
        bs = self.block_size
        if len(seq.block_table) < seq.num_blocks:
            block = self._fresh_block()
            seq.block_table.append(block.block_id)
        assert len(seq.block_table) == seq.num_blocks
        if seq.num_tokens % bs == 0:
            # this step fills the last block: publish it for prefix sharing
            last = self.blocks[seq.block_table[-1]]
            if last.hash == -1:
                prefix = self.blocks[seq.block_table[-2]].hash if len(seq.block_table) > 1 else -1
                token_ids = seq.block(seq.num_blocks - 1)
                if prefix != -1 or len(seq.block_table) == 1:
                    h = self.compute_hash(token_ids, prefix)
                    last.update(h, token_ids)
                    self.hash_to_block_id[h] = last.block_id

    def append_ahead(self, seq: Sequence) -> None:
        """Give ``seq`` the block that the token AFTER its newest one starts, if it has none yet
        (for a decode step launched before the current step's token is known; the caller has
        checked ``num_free_blocks``)."""
        if len(seq.block_table) * self.block_size < seq.num_tokens + 1:
            seq.block_table.append(self._fresh_block().block_id)

    # ------------------------------------------------------------------ stats
    def stats(self) -> dict:
        return {
            "num_blocks": len(self.blocks),
            "free": len(self.free_block_ids),
            "used": len(self.used_block_ids),
            "cached_hashes": len(self.hash_to_block_id),
            "prefix_hit_rate": (self.num_prefix_hits / self.num_prefix_queries
                                if self.num_prefix_queries else 0.0),
        }


# ------------------------------------------------------------------ native implementation
_HOST_LIB = None


def _host_lib():
    """``_host_core.so`` (csrc/host/block_manager.cpp, built by minivllm_amd.build), or None."""
    global _HOST_LIB
    if _HOST_LIB is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                            "_host_core.so")
        if not os.path.exists(path):
            _HOST_LIB = False
            return None
        try:
            lib = ctypes.CDLL(path)
        except OSError as e:                  # built for another machine: use the Python allocator
            import warnings
            warnings.warn(f"cannot load {path} ({e}); falling back to the Python block manager")
            _HOST_LIB = False
            return None
        vp = ctypes.c_void_p
        lib.mvl_bm_create.restype = vp
        lib.mvl_bm_create.argtypes = [ctypes.c_int, ctypes.c_int]
        lib.mvl_bm_destroy.argtypes = [vp]
        lib.mvl_bm_num_free.argtypes = [vp]
        lib.mvl_bm_allocate.restype = ctypes.c_long
        lib.mvl_bm_allocate.argtypes = [vp, vp, ctypes.c_long, vp]
        lib.mvl_bm_deallocate.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int]
        lib.mvl_bm_append.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long, vp]
        lib.mvl_bm_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_long)]
        lib.mvl_bm_ref_count.argtypes = [vp, ctypes.c_int]
        lib.mvl_chain_hash.restype = ctypes.c_uint64
        lib.mvl_chain_hash.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_uint64]
        _HOST_LIB = lib
    return _HOST_LIB or None


class NativeBlockManager:
    """Same interface and the same allocation decisions as :class:`BlockManager`, with the block
    table bookkeeping, the free list and the xxh64 chain hashing in C++
    (``csrc/host/block_manager.cpp``).  Sequences stay Python objects; their token ids cross the
    boundary as int32 arrays only when a prompt is admitted or a block fills up."""

    def __init__(self, num_blocks: int, block_size: int):
        if num_blocks <= 0:
            raise ValueError("num_blocks must be positive")
        lib = _host_lib()
        if lib is None:
            raise RuntimeError("_host_core.so is missing: run `python -m minivllm_amd.build`")
        self._lib = lib
        self.block_size = block_size
        self._n = num_blocks
        self._h = lib.mvl_bm_create(num_blocks, block_size)
        self._stats = (__import__("ctypes").c_long * 6)()

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.mvl_bm_destroy(h)

    @staticmethod
    def compute_hash(token_ids: List[int], prefix_hash: int = -1) -> int:
        t = np.ascontiguousarray(token_ids, dtype=np.int32)
        return int(_host_lib().mvl_chain_hash(t.ctypes.data, len(t), int(prefix_hash != -1),
                                              0 if prefix_hash == -1 else prefix_hash))

    @property
    def num_free_blocks(self) -> int:
        return self._lib.mvl_bm_num_free(self._h)

    @property
    def num_blocks(self) -> int:
        return self._n

    def ref_count(self, block_id: int) -> int:
        return self._lib.mvl_bm_ref_count(self._h, block_id)

    @property
    def blocks(self) -> List[Block]:
        """Snapshot for inspection (ids and reference counts; hashes and tokens stay native)."""
        out = []
        for i in range(self._n):
            b = Block(i)
            b.ref_count = self.ref_count(i)
            out.append(b)
        return out

    def can_allocate(self, seq: Sequence) -> bool:
        return self.num_free_blocks >= seq.num_blocks

    def allocate(self, seq: Sequence) -> None:
        assert not seq.block_table, "sequence already has blocks"
        # the sequence's cached int32 image (no list -> array conversion per admission)
        toks = np.ascontiguousarray(seq.token_array(0, seq.num_tokens))
        table = np.empty(seq.num_blocks, dtype=np.int32)
        cached = self._lib.mvl_bm_allocate(self._h, toks.ctypes.data, len(toks), table.ctypes.data)
        if cached < 0:
            raise RuntimeError("allocate() without enough free blocks (check can_allocate first)")
        seq.block_table = table.tolist()
        seq.num_cached_tokens = int(cached)

    def deallocate(self, seq: Sequence, num_valid_tokens: Optional[int] = None) -> None:
        """See :meth:`BlockManager.deallocate`."""
        if seq.block_table:
            table = np.asarray(seq.block_table, dtype=np.int32)
            first_invalid = (len(table) if num_valid_tokens is None
                             else max(num_valid_tokens, 0) // self.block_size)
            self._lib.mvl_bm_deallocate(self._h, table.ctypes.data, len(table), first_invalid)
        seq.block_table = []
        seq.num_cached_tokens = 0

    def can_append(self, seq: Sequence) -> bool:
        # the common decode step neither starts nor completes a block: no call into the library
        if len(seq.block_table) >= seq.num_blocks:
            return True
        return self.num_free_blocks >= 1

    def append(self, seq: Sequence) -> None:
        n = seq.num_tokens
        bt = seq.block_table
        if n % self.block_size != 0 and len(bt) >= seq.num_blocks:
            return                                        # nothing to allocate, nothing to publish
        last_tokens = None
        if n % self.block_size == 0:                      # this step fills the last block
            last_tokens = np.ascontiguousarray(seq.block(seq.num_blocks - 1), dtype=np.int32)
        added = self._lib.mvl_bm_append(
            self._h, len(bt), bt[-1] if bt else -1, bt[-2] if len(bt) > 1 else -1, n,
            last_tokens.ctypes.data if last_tokens is not None else None)
        if added == -2:
            raise RuntimeError("append() without a free block (check can_append first)")
        if added >= 0:
            bt.append(added)
        assert len(bt) == seq.num_blocks

    def append_ahead(self, seq: Sequence) -> None:
        """See :meth:`BlockManager.append_ahead`."""
        bt = seq.block_table
        added = self._lib.mvl_bm_append(self._h, len(bt), bt[-1] if bt else -1,
                                        bt[-2] if len(bt) > 1 else -1, seq.num_tokens + 1, None)
        if added == -2:
            raise RuntimeError("append_ahead() without a free block (check num_free_blocks first)")
        if added >= 0:
            bt.append(added)

    @property
    def num_prefix_hits(self) -> int:
        return self.stats()["prefix_hits"]

    @property
    def num_prefix_queries(self) -> int:
        return self.stats()["prefix_queries"]

    def stats(self) -> dict:
        self._lib.mvl_bm_stats(self._h, self._stats)
        n, free, used, hashes, hits, queries = list(self._stats)
        return {"num_blocks": n, "free": free, "used": used, "cached_hashes": hashes,
                "prefix_hits": hits, "prefix_queries": queries,
                "prefix_hit_rate": hits / queries if queries else 0.0}


def make_block_manager(num_blocks: int, block_size: int, native: Optional[bool] = None):
    """The C++ allocator when its library is built (default), else the Python one.
    ``MINIVLLM_PY_HOST=1`` forces the Python implementation."""
    import os
    if native is None:
        native = os.environ.get("MINIVLLM_PY_HOST") != "1" and _host_lib() is not None
    return NativeBlockManager(num_blocks, block_size) if native else BlockManager(num_blocks, block_size)
