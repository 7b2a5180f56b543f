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

    def deallocate(self, seq: Sequence) -> None:
        for block_id in reversed(seq.block_table):
            block = self.blocks[block_id]
            block.ref_count -= 1
            assert block.ref_count >= 0
            if block.ref_count == 0:
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
