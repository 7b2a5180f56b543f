"""Continuous-batching scheduler: prefill first, otherwise one decode token per
running sequence, recompute-style preemption when KV blocks run out.

Policy follows the reference (reference: src/myvllm/engine/scheduler.py:6-82):
two queues, head-of-line prefill admission under a sequence budget and a token
budget, LIFO victim selection for preemption, preempted sequences go back to the
*front* of the waiting queue and are re-prefetched from scratch.  Corrected
relative to the reference: sampled tokens go through ``Sequence.append_token`` so
lengths advance and ``max_tokens`` triggers; a decode candidate that does not fit
the step's budget stays in the running queue instead of being dropped; prompts
that can never be scheduled are rejected when they are added.

Beyond the reference (which can never schedule a prompt longer than
``max_num_batched_tokens``, scheduler.py:31,38-39): chunked prefill.  The last sequence
admitted in a prefill step may receive only the tokens the step's budget still has room
for; it then waits in ``prefilling`` and later prefill steps continue it, attending to its
own earlier chunks through the paged cache, before anything new is admitted.  Only the chunk
that reaches the end of the sequence yields a sampled token.

Also beyond the reference, all optional: ``min_prefill_tokens`` / ``max_prefill_delay_s`` hold a
small prefill step back while sequences are decoding (until enough prompt tokens wait, or the
oldest waiter is that old), ``enable_mixed_batches`` lets every running sequence ride along in a
prefill step with one decode token (``last_num_decode`` trailing entries of the scheduled list),
and ``abort`` removes a request from whichever queue holds it.
"""
from __future__ import annotations

import time
from collections import deque
from typing import Deque, List, Optional, Tuple

from .block_manager import BlockManager, make_block_manager  # noqa: F401
from .sequence import Sequence, SequenceStatus


class Scheduler:
    def __init__(self, max_num_sequences: int, max_num_batched_tokens: int,
                 max_cached_blocks: int, block_size: int, eos: int,
                 max_model_length: Optional[int] = None, enable_chunked_prefill: bool = True,
                 min_prefill_tokens: int = 0, max_prefill_delay_s: float = 0.0,
                 enable_mixed_batches: bool = False, ahead_over_block_edges: bool = True):
        self.enable_chunked_prefill = enable_chunked_prefill
        # decode steps launched ahead may cross KV-block edges (blocks allocated ahead, block
        # hashes published when the step is collected) and go on while requests wait that cannot
        # be admitted; off: only steps that touch no edge, and only while nobody waits
        self.ahead_over_block_edges = ahead_over_block_edges
        self._edge_ahead = False
        # > 0: while sequences are decoding, newcomers are admitted only once this many prompt
        # tokens can go into one prefill step (or everything that waits fits).  The reference
        # interrupts the decode batch for every single newcomer (scheduler.py:29-41), which
        # turns a saturated server into an alternation of full decode steps and tiny prefills.
        self.min_prefill_tokens = min_prefill_tokens
        # online arrivals: even when every waiting request would fit, a step below
        # min_prefill_tokens is held back until the oldest waiting request is this old, so that
        # requests arriving within the window are prefetched together
        self.max_prefill_delay_s = max_prefill_delay_s
        # mixed steps: a prefill step also carries one decode token for every sequence that is
        # already running (they are appended behind the prefill sequences; their count is
        # ``last_num_decode``), so decoding does not stall while newcomers are prefetched
        self.enable_mixed_batches = enable_mixed_batches
        self.last_num_decode = 0
        self.num_mixed_steps = 0
        # admitted (blocks allocated) but not yet completely prefetched, oldest first
        self.prefilling: Deque[Sequence] = deque()
        self.block_manager = make_block_manager(max_cached_blocks, block_size)
        self.max_num_sequences = max_num_sequences
        self.max_num_batched_tokens = max_num_batched_tokens
        self.max_model_length = max_model_length
        self.block_size = block_size
        self.eos = eos
        self.waiting: Deque[Sequence] = deque()
        self.running: Deque[Sequence] = deque()
        self.num_preemptions = 0
        # steady-state decode shortcut: while no running sequence touches a block boundary
        # (no allocation, no hash to publish) and membership is unchanged, the previous decode
        # batch is simply scheduled again (same list object, which also tells the runner that
        # it may update its batch arrays incrementally)
        self._decode_batch: Optional[List[Sequence]] = None
        self._plain_decode_steps = 0
        # sequence ids whose partial prefill chunk of the running step has been accounted for
        # already (schedule_prefill_ahead): postprocess skips each of them once
        self._advanced_early: set = set()

    def is_finished(self) -> bool:
        return not self.waiting and not self.running and not self.prefilling

    def add_sequence(self, seq: Sequence) -> None:
        seq.block_size = self.block_size
        limit = self.max_num_batched_tokens
        # a preempted sequence is re-prefetched with its generated tokens, so the
        # whole request (prompt + max_tokens) must fit one prefill step
        worst = seq.num_prompt_tokens + seq.max_tokens
        if seq.num_prompt_tokens > limit and not self.enable_chunked_prefill:
            raise ValueError(
                f"prompt of {seq.num_prompt_tokens} tokens exceeds max_num_batched_tokens={limit}")
        if self.max_model_length is not None and worst > self.max_model_length:
            raise ValueError(
                f"prompt ({seq.num_prompt_tokens}) + max_tokens ({seq.max_tokens}) exceeds "
                f"max_model_length={self.max_model_length}")
        need = (worst + self.block_size - 1) // self.block_size
        if need > self.block_manager.num_blocks:
            raise ValueError(
                f"request needs {need} KV blocks but the cache has {self.block_manager.num_blocks}")
        seq.arrival_time = time.monotonic()
        self.waiting.append(seq)
        self._decode_batch = None

    # ------------------------------------------------------------- schedule
    def schedule(self) -> Tuple[List[Sequence], bool]:
        self.last_num_decode = 0
        scheduled, num_tokens = self._schedule_prefill()
        if scheduled:
            self._decode_batch = None
            if self.enable_mixed_batches:
                self._add_decode_riders(scheduled, num_tokens)
            return scheduled, True
        bm = self.block_manager

        # phase 2: one token for every running sequence, oldest first
        if (self._decode_batch is not None and self._plain_decode_steps > 0 and not self.waiting
                and self._decode_batch):
            self._plain_decode_steps -= 1
            return self._decode_batch, False
        candidates = self.running
        self.running = deque()
        bs = self.block_size
        deferred: List[Sequence] = []
        while candidates:
            seq = candidates.popleft()
            if len(scheduled) >= self.max_num_sequences or num_tokens >= self.max_num_batched_tokens:
                deferred.append(seq)
                continue
            n = seq.num_tokens
            if len(seq.block_table) * bs < n:           # this token starts a new KV block
                preempted_self = False
                while not bm.can_append(seq):
                    if candidates:
                        self.preempt(candidates.pop())  # newest sequence is the victim
                    elif deferred:
                        self.preempt(deferred.pop())
                    else:
                        self.preempt(seq)
                        preempted_self = True
                        break
                if preempted_self:
                    continue
                bm.append(seq)
            elif n % bs == 0:                           # this token completes a block: publish it
                bm.append(seq)
            scheduled.append(seq)
            num_tokens += 1
        self.running.extend(scheduled)
        self.running.extend(deferred)
        self._edge_ahead = False
        if scheduled and not deferred:
            # (somebody may be waiting: then admission was not possible just now, and the step
            # after this one can still be launched ahead while that stays so)
            self._decode_batch, self._plain_decode_steps = scheduled, self._count_plain_steps(scheduled)
        else:
            self._decode_batch = None
        return scheduled, False

    def _schedule_prefill(self) -> Tuple[List[Sequence], int]:
        """The sequences of a prefill step (and the prompt tokens it computes), [] when nothing
        is to be prefetched now."""
        scheduled: List[Sequence] = []
        num_tokens = 0
        bm = self.block_manager
        budget = self.max_num_batched_tokens
        if self.enable_mixed_batches:                   # leave room for the riders
            budget = max(budget - len(self.running), min(budget, self.block_size))
        # phase 0: continue sequences whose prefill was cut by an earlier step's budget
        while self.prefilling and len(scheduled) < self.max_num_sequences and num_tokens < budget:
            seq = self.prefilling[0]
            take = min(len(seq) - seq.num_cached_tokens, budget - num_tokens)
            seq.prefill_end = seq.num_cached_tokens + take
            scheduled.append(seq)
            num_tokens += take
            if seq.prefill_end < len(seq):
                break                                   # budget exhausted, still not complete
            self.prefilling.popleft()
            self.running.append(seq)

        # phase 1: admit waiting sequences (prefill) in arrival order.  Nothing new is admitted
        # while an older sequence is still incomplete after this step: a newcomer could hit its
        # prefix-cache blocks before their KV exists.
        while (self.waiting and not self.prefilling and len(scheduled) < self.max_num_sequences):
            if not scheduled and self._defer_prefill():
                break
            seq = self.waiting[0]
            fits = num_tokens + len(seq) <= budget
            # a cut sequence gets at least one KV block worth of tokens now, or waits
            room = budget - num_tokens >= min(len(seq), self.block_size)
            if not fits and scheduled and not (self.enable_chunked_prefill and room):
                break
            if not bm.can_allocate(seq):
                break
            if len(self.running) >= self.max_num_sequences:
                break
            self.waiting.popleft()
            bm.allocate(seq)
            seq.status = SequenceStatus.RUNNING
            todo = len(seq) - seq.num_cached_tokens
            take = todo
            if self.enable_chunked_prefill:
                take = min(todo, budget - num_tokens)
            # without chunking an over-long sequence (grew while running, then preempted) goes
            # through alone rather than stall forever
            seq.prefill_end = seq.num_cached_tokens + take
            scheduled.append(seq)
            num_tokens += take
            if seq.prefill_end < len(seq):
                self.prefilling.append(seq)
                break
            self.running.append(seq)
        return scheduled, num_tokens

    # ------------------------------------------------ prefill steps launched ahead
    def schedule_prefill_ahead(self) -> List[Sequence]:
        """While a prefill step is still on the device (``postprocess`` has not run for it): the
        sequences of the prefill step that follows, or [] when the next step is not a prefill
        step or cannot be decided yet.  Nothing a prefill step's make-up depends on needs the
        tokens the running step will sample: newcomers take blocks that are free now, a sequence
        of the running step that finishes only gives blocks and slots back, and a prompt the
        running step has cut at its token budget continues where that chunk ends - its
        bookkeeping for the running step is done here, early, and ``postprocess`` then skips it
        once.  Steps with decode riders are left to ``schedule``.  The caller must run the
        returned step next."""
        if self.enable_mixed_batches or not (self.waiting or self.prefilling):
            return []
        if self.prefilling:
            cut = self.prefilling[0]
            if cut.prefill_end is None or cut.seq_id in self._advanced_early:
                return []                           # not part of the step that is running
            cut.num_cached_tokens, cut.prefill_end = cut.prefill_end, None
            self._advanced_early.add(cut.seq_id)
        scheduled, _ = self._schedule_prefill()
        if scheduled:
            self._decode_batch = None
        return scheduled

    # ------------------------------------------------ decode steps launched ahead
    def _count_plain_steps(self, batch: List[Sequence]) -> int:
        """How many decode steps AFTER the one that processes the sequences' current lengths
        touch no block edge and still have somebody who wants a token."""
        bs = self.block_size
        # after that step a sequence holds n + 1 tokens; the step scheduled for length m is
        # plain unless m % bs is 0 (publishes a hash) or 1 (needs a new block)
        plain, wanted = bs, bs
        for seq in batch:
            r = (seq.num_tokens + 1) % bs
            plain = min(plain, 0 if r <= 1 else bs - r)
            # steps after that one in which the sequence still wants a token
            wanted = min(wanted, seq.max_tokens - seq.num_completion_tokens - 1)
        # no step is launched ahead of the point where a sequence reaches its max_tokens: the
        # batch that follows is a different one (its rows renumbered, which the sampler's noise
        # is keyed by), and a row launched for a sequence that is done would run for nothing
        return max(min(plain, wanted), 0) if bs > 2 else 0

    def next_decode_step_ahead(self) -> int:
        """After a decode step has been scheduled (or while one launched ahead is in flight):
        may the step after it be launched before its tokens are known?  0: no - someone waits
        who could be admitted (a free slot, enough blocks, no hold-back), the batch is not
        everybody, blocks are short, or a sequence reaches its ``max_tokens``.  1: yes, the same batch with the same block tables.  2: yes, and sequences whose
        next token starts a new KV block have just been given that block (their tables grew).
        A block that the step will complete has its hash published when the step is collected
        (``consume_decode_step_ahead``): that needs the token sampled in between."""
        if self._decode_batch is None or self.prefilling or self._admission_possible():
            return 0
        if self._plain_decode_steps > 0:
            return 1
        bs, bm, batch = self.block_size, self.block_manager, self._decode_batch
        if not self.ahead_over_block_edges or bs <= 2 or not batch:
            return 0
        if min(s.max_tokens - s.num_completion_tokens - 1 for s in batch) <= 0:
            return 0                                    # somebody finishes: the batch changes
        need = [s for s in batch if len(s.block_table) * bs < s.num_tokens + 1]
        if bm.num_free_blocks < len(need):
            return 0                                    # schedule() decides whom to preempt
        for s in need:
            bm.append_ahead(s)
        self._edge_ahead = True
        return 2 if need else 1

    def _admission_possible(self) -> bool:
        """Would ``schedule`` start a prefill step for the oldest waiting request right now?
        (While every slot is taken, the pool cannot hold the request or a small prefill step is
        being held back, a saturated engine keeps decoding the same batch.)"""
        if not self.ahead_over_block_edges:
            return bool(self.waiting)                   # the cautious rule: nobody may wait
        if not self.waiting or self._defer_prefill():
            return False
        return (self.block_manager.can_allocate(self.waiting[0])
                and len(self.running) < self.max_num_sequences)

    def next_decode_step_is_plain(self) -> bool:
        """The step after the scheduled one is the same batch with nothing to allocate or
        publish."""
        return (self._decode_batch is not None and self._plain_decode_steps > 0
                and not self.waiting and not self.prefilling)

    def consume_decode_step_ahead(self, seqs: List[Sequence]) -> None:
        """Account for a decode step that ran without ``schedule()`` having been asked (it had
        been launched ahead) and is now being collected for ``seqs``: publish the blocks it
        completes, use up one plain step or count the plain steps that follow an edge."""
        if self._edge_ahead:
            self._edge_ahead = False
            bm, bs = self.block_manager, self.block_size
            for s in seqs:
                if s.num_tokens % bs == 0:
                    bm.append(s)
            if self._decode_batch is not None:
                self._plain_decode_steps = self._count_plain_steps(self._decode_batch)
        elif self._decode_batch is not None and self._plain_decode_steps > 0:
            self._plain_decode_steps -= 1

    def _add_decode_riders(self, scheduled: List[Sequence], num_tokens: int) -> None:
        """Append every running sequence that can take its next token without evicting anyone
        (a sequence that needs a KV block the pool cannot give simply sits this step out)."""
        bm, bs = self.block_manager, self.block_size
        fresh = {id(s) for s in scheduled}
        riders: List[Sequence] = []
        for seq in self.running:                        # oldest first
            if id(seq) in fresh:
                continue
            if (len(scheduled) + len(riders) >= self.max_num_sequences
                    or num_tokens + len(riders) >= self.max_num_batched_tokens):
                break
            n = seq.num_tokens
            if len(seq.block_table) * bs < n:           # this token starts a new KV block
                if not bm.can_append(seq):
                    continue
                bm.append(seq)
            elif n % bs == 0:                           # this token completes a block: publish it
                bm.append(seq)
            riders.append(seq)
        scheduled.extend(riders)
        self.last_num_decode = len(riders)
        self.num_mixed_steps += bool(riders)

    def _defer_prefill(self) -> bool:
        """Keep decoding instead of starting a prefill step that would carry too few tokens."""
        if not self.min_prefill_tokens or not self.running:
            return False
        slots = self.max_num_sequences - len(self.running)
        tokens = admitted = 0
        for seq in self.waiting:
            if admitted >= slots or tokens >= self.min_prefill_tokens:
                break
            tokens += len(seq)
            admitted += 1
        if tokens >= self.min_prefill_tokens:
            return False
        if admitted < len(self.waiting):
            return True                 # slots are the limit: wait until more of them free up
        if self.max_prefill_delay_s > 0:
            age = time.monotonic() - getattr(self.waiting[0], "arrival_time", 0.0)
            return age < self.max_prefill_delay_s
        return False

    def preempt(self, seq: Sequence) -> None:
        self._decode_batch = None
        seq.prefill_end = None
        self.block_manager.deallocate(seq)
        seq.status = SequenceStatus.WAITING
        self.waiting.appendleft(seq)
        self.num_preemptions += 1

    def abort(self, seq: Sequence, step_failed: bool = False) -> bool:
        """Drop a request wherever it is queued (a client went away): its KV blocks return to
        the pool and the full blocks whose KV was computed stay in the prefix cache.  Blocks
        whose hash was published at admission but whose KV does not exist yet (the prompt was
        still being prefilled in chunks) are withdrawn from the prefix cache.  ``step_failed``:
        the step that last carried the sequence raised before completing, so nothing that step
        was to write counts as computed.  False when the request had already left."""
        found = in_prefill = False
        for q in (self.waiting, self.prefilling, self.running):
            for i, s in enumerate(q):
                if s is seq:
                    del q[i]
                    found = True
                    in_prefill = q is self.prefilling
                    break
        if not found:
            return False
        self._decode_batch = None
        self._advanced_early.discard(seq.seq_id)
        if seq.block_table:
            if in_prefill or seq.prefill_end is not None:
                # prefill incomplete, or its step never reached postprocess
                valid = 0 if step_failed else seq.num_cached_tokens
            elif step_failed:
                valid = seq.num_tokens - 1       # the newest token's KV slot may be unwritten
            else:
                valid = None
            self.block_manager.deallocate(seq, valid)
        seq.prefill_end = None
        seq.status = SequenceStatus.FINISHED
        return True

    # ---------------------------------------------------------- postprocess
    def postprocess(self, seqs: List[Sequence], token_ids: List[int]) -> List[Sequence]:
        finished = []
        for seq, token_id in zip(seqs, token_ids):
            if seq.seq_id in self._advanced_early:
                self._advanced_early.discard(seq.seq_id)     # a cut prompt, advanced ahead of time
                continue
            if seq.prefill_end is not None:
                end, seq.prefill_end = seq.prefill_end, None
                if end < seq.num_tokens:
                    # partial prefill chunk: its KV is in the cache now, nothing was generated
                    seq.num_cached_tokens = end
                    continue
            seq.append_token(int(token_id))
            hit_eos = ((not seq.ignore_eos) and token_id == self.eos) or token_id in seq.stop_token_ids
            hit_len = seq.num_completion_tokens >= seq.max_tokens
            hit_ctx = (self.max_model_length is not None
                       and seq.num_tokens >= self.max_model_length)
            if hit_eos or hit_len or hit_ctx:
                seq.status = SequenceStatus.FINISHED
                self.block_manager.deallocate(seq)
                finished.append(seq)
        if finished:
            self._decode_batch = None
            done = {id(s) for s in finished}
            self.running = deque(s for s in self.running if id(s) not in done)
        return finished
