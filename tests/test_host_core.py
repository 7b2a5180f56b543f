"""Host-only units: Sequence, BlockManager (ref-counting, prefix cache, revive, collision
guard), Scheduler (prefill priority, budgets, LIFO recompute preemption, finish rules)."""
import pickle

import pytest

from minivllm_amd.engine.block_manager import BlockManager
from minivllm_amd.engine.scheduler import Scheduler
from minivllm_amd.engine.sequence import Sequence, SequenceStatus
from minivllm_amd.sampling_parameters import SamplingParams

BS = 4


def seq(tokens, **kw):
    return Sequence(list(tokens), SamplingParams(**kw) if kw else None, block_size=BS)


# ------------------------------------------------------------------ Sequence
def test_sequence_properties_and_append():
    s = seq(range(10), max_tokens=5)
    assert len(s) == 10 and s.num_blocks == 3 and s.last_block_num_tokens == 2
    assert s.block(0) == [0, 1, 2, 3] and s.block(2) == [8, 9]
    s.append_token(99)
    assert s.last_token == 99 and s.num_tokens == 11 and s.num_completion_tokens == 1
    assert s.completion_token_ids == [99] and s.prompt_token_ids == list(range(10))
    s.append_token(100)
    assert s.num_blocks == 3 and s.last_block_num_tokens == 4
    with pytest.raises(IndexError):
        s.block(3)


def test_sequence_pickle_is_compact_and_complete():
    s = seq(range(9), temperature=0.5, max_tokens=7, ignore_eos=True)
    s.block_table = [3, 1, 2]
    r = pickle.loads(pickle.dumps(s))
    assert r.token_ids == list(range(9)) and r.block_table == [3, 1, 2]
    assert (r.temperature, r.max_tokens, r.ignore_eos, r.block_size) == (0.5, 7, True, BS)
    s.append_token(42)
    r = pickle.loads(pickle.dumps(s))                 # after the first completion token only
    assert r.last_token == 42 and r.num_tokens == 10  # the last token travels
    assert len(pickle.dumps(s)) < len(pickle.dumps(list(range(2000))))


def test_sequence_token_array_follows_the_token_list():
    """The int32 image used for batch preparation: prompt slices are views of one cached array,
    a range that reaches into generated tokens (re-prefill after preemption) rebuilds it, and an
    unpickled sequence (no image) builds it on demand."""
    import pickle
    import numpy as np
    s = Sequence(list(range(100, 110)), SamplingParams(max_tokens=8), block_size=4)
    a = s.token_array(0, 10)
    assert a.dtype == np.int32 and a.tolist() == list(range(100, 110))
    assert np.shares_memory(a, s.token_array(2, 5)) and s.token_array(2, 5).tolist() == [102, 103, 104]
    s.append_token(7)
    s.append_token(8)
    assert s.token_array(0, 12).tolist() == list(range(100, 110)) + [7, 8]       # rebuilt
    assert s.token_array(10, 12).tolist() == [7, 8]
    r = pickle.loads(pickle.dumps(Sequence([5, 6, 7], block_size=4)))
    assert r.token_array(1, 3).tolist() == [6, 7]


def test_sampling_params_validation():
    assert SamplingParams(temperature=0).greedy
    with pytest.raises(ValueError):
        SamplingParams(temperature=-1)
    with pytest.raises(ValueError):
        SamplingParams(max_tokens=0)


# -------------------------------------------------------------- BlockManager
def test_allocate_sets_refcount_and_free_on_deallocate():
    bm = BlockManager(8, BS)
    s = seq(range(10))
    assert bm.can_allocate(s)
    bm.allocate(s)
    assert len(s.block_table) == 3 and bm.num_free_blocks == 5
    assert all(bm.blocks[b].ref_count == 1 for b in s.block_table)
    bm.deallocate(s)
    assert bm.num_free_blocks == 8 and s.block_table == [] and s.num_cached_tokens == 0
    assert all(b.ref_count == 0 for b in bm.blocks)


def test_prefix_hit_shares_full_blocks_only():
    bm = BlockManager(16, BS)
    a = seq(list(range(8)) + [100, 101])
    b = seq(list(range(8)) + [200])
    bm.allocate(a)
    bm.allocate(b)
    assert b.num_cached_tokens == 8 and b.block_table[:2] == a.block_table[:2]
    assert b.block_table[2] != a.block_table[2]                 # partial blocks never shared
    assert bm.blocks[a.block_table[0]].ref_count == 2
    bm.deallocate(a)
    assert bm.blocks[b.block_table[0]].ref_count == 1            # still alive for b
    assert b.block_table[0] in bm.used_block_ids
    bm.deallocate(b)
    assert bm.num_free_blocks == 16


def test_hash_chain_needs_matching_prefix():
    bm = BlockManager(16, BS)
    a = seq([1, 2, 3, 4, 5, 6, 7, 8, 9])
    b = seq([9, 9, 9, 9, 5, 6, 7, 8, 9])          # second block equal, first differs
    bm.allocate(a)
    bm.allocate(b)
    assert b.num_cached_tokens == 0
    assert set(a.block_table).isdisjoint(b.block_table)


def test_freed_blocks_are_revived_until_reused():
    bm = BlockManager(4, BS)
    a = seq(range(8))
    bm.allocate(a)
    blocks = list(a.block_table)
    bm.deallocate(a)
    b = seq(list(range(8)) + [77])
    bm.allocate(b)
    assert b.block_table[:2] == blocks and b.num_cached_tokens == 8
    bm.deallocate(b)
    # burn through the pool with other content: the cached identity must disappear
    for base in (100, 200):
        c = seq(range(base, base + 16))
        bm.allocate(c)
        bm.deallocate(c)
    d = seq(list(range(8)) + [1])
    bm.allocate(d)
    assert d.num_cached_tokens == 0


def test_fully_cached_prompt_still_computes_last_token():
    bm = BlockManager(8, BS)
    a = seq(range(8))
    bm.allocate(a)
    b = seq(range(8))
    bm.allocate(b)
    assert b.block_table == a.block_table and b.num_cached_tokens == 7


def test_collision_guard_compares_tokens(monkeypatch):
    bm = BlockManager(8, BS)
    monkeypatch.setattr(BlockManager, "compute_hash", staticmethod(lambda t, p=-1: 1234))
    a, b = seq([1, 2, 3, 4, 5]), seq([9, 9, 9, 9, 5])
    bm.allocate(a)
    bm.allocate(b)
    assert b.num_cached_tokens == 0 and a.block_table[0] != b.block_table[0]


def test_append_allocates_at_block_boundary_and_publishes_full_blocks():
    bm = BlockManager(8, BS)
    s = seq(range(4))
    bm.allocate(s)                      # exactly one full block
    s.append_token(50)                  # token 5 needs a second block
    assert bm.can_append(s)
    bm.append(s)
    assert len(s.block_table) == 2
    for t in (51, 52):
        s.append_token(t)
        bm.append(s)
        assert len(s.block_table) == 2
    s.append_token(53)                  # 8 tokens: this step fills block 2
    bm.append(s)
    assert bm.blocks[s.block_table[1]].hash != -1
    other = seq(s.token_ids + [1])
    bm.allocate(other)
    assert other.num_cached_tokens == 8
    # no free block -> cannot start a new one
    bm2 = BlockManager(1, BS)
    t = seq(range(4))
    bm2.allocate(t)
    t.append_token(1)
    assert not bm2.can_append(t)


def test_compute_hash_is_chained_and_stable():
    h1 = BlockManager.compute_hash([1, 2, 3, 4])
    assert h1 == BlockManager.compute_hash([1, 2, 3, 4])
    assert BlockManager.compute_hash([1, 2, 3, 4], h1) != h1
    assert BlockManager.compute_hash([1, 2, 3, 5]) != h1


# ----------------------------------------------------------------- Scheduler
def make_sched(blocks=64, max_seqs=4, max_tokens=32, eos=999, max_len=None, chunked=False):
    """chunked=False is the reference policy (a prompt is prefetched whole or not at all)."""
    return Scheduler(max_seqs, max_tokens, blocks, BS, eos, max_model_length=max_len,
                     enable_chunked_prefill=chunked)


def test_prefill_has_priority_and_respects_budgets():
    sch = make_sched(max_seqs=3, max_tokens=20)
    seqs = [seq(range(i * 100, i * 100 + 8), max_tokens=4) for i in range(5)]
    for s in seqs:
        sch.add_sequence(s)
    batch, is_prefill = sch.schedule()
    assert is_prefill and batch == seqs[:2]                   # 8 + 8 fits, third would exceed 20
    batch, is_prefill = sch.schedule()
    assert is_prefill and batch == seqs[2:3]                  # running is capped at 3 sequences
    batch, is_prefill = sch.schedule()
    assert not is_prefill and batch == seqs[:3]               # decode, oldest first
    assert [s.status for s in seqs[:3]] == [SequenceStatus.RUNNING] * 3


def test_finish_on_max_tokens_and_eos():
    sch = make_sched()
    a, b = seq(range(5), max_tokens=2), seq(range(5), max_tokens=8)
    sch.add_sequence(a); sch.add_sequence(b)
    batch, _ = sch.schedule()
    assert sch.postprocess(batch, [1, 1]) == []
    batch, _ = sch.schedule()
    done = sch.postprocess(batch, [2, 999])                   # a hits max_tokens, b hits eos
    assert done == [a, b] and sch.is_finished()
    assert a.completion_token_ids == [1, 2] and b.completion_token_ids == [1, 999]
    assert sch.block_manager.num_free_blocks == 64
    c = seq(range(5), max_tokens=3, ignore_eos=True)
    sch.add_sequence(c)
    batch, _ = sch.schedule()
    assert sch.postprocess(batch, [999]) == []                # eos ignored
    d = seq(range(5), max_tokens=9, ignore_eos=True, stop_token_ids=[42, 43])
    sch.add_sequence(d)
    sch.postprocess(sch.schedule()[0], [5, 5])
    done = sch.postprocess(sch.schedule()[0], [6, 43])        # per-request stop id ends d, kept
    assert done == [d] and d.completion_token_ids == [5, 43]


def test_preemption_is_lifo_and_recomputes():
    sch = make_sched(blocks=4, max_seqs=4, max_tokens=64)
    a, b = seq(range(7), max_tokens=8), seq(range(100, 107), max_tokens=8)
    sch.add_sequence(a); sch.add_sequence(b)
    batch, is_prefill = sch.schedule()
    assert is_prefill and batch == [a, b] and sch.block_manager.num_free_blocks == 0
    sch.postprocess(batch, [1, 1])                            # both at 8 tokens: blocks full
    batch, is_prefill = sch.schedule()
    assert not is_prefill and batch == [a, b]                 # still fits (8 tokens = 2 blocks)
    sch.postprocess(batch, [2, 2])                            # 9 tokens: each needs a 3rd block
    batch, is_prefill = sch.schedule()
    assert not is_prefill and batch == [a]                    # b (newest) was preempted
    assert b.status == SequenceStatus.WAITING and b.block_table == [] and sch.waiting[0] is b
    assert sch.num_preemptions == 1
    sch.postprocess(batch, [3])
    # b comes back through prefill with its generated tokens once a finishes
    a.max_tokens = a.num_completion_tokens + 1
    batch, _ = sch.schedule()
    if batch == [a]:
        sch.postprocess(batch, [4])
        batch, is_prefill = sch.schedule()
    assert batch == [b] and b.num_tokens == 9


def test_decode_candidates_beyond_budget_stay_running():
    sch = make_sched(max_seqs=2, max_tokens=64)
    seqs = [seq(range(i * 10, i * 10 + 5), max_tokens=9) for i in range(2)]
    for s in seqs:
        sch.add_sequence(s)
    batch, _ = sch.schedule()
    sch.postprocess(batch, [1, 1])
    sch.max_num_sequences = 1                                  # shrink the step budget
    batch, _ = sch.schedule()
    assert batch == [seqs[0]] and len(sch.running) == 2        # nothing got lost


def test_unschedulable_requests_are_rejected_up_front():
    sch = make_sched(blocks=4, max_tokens=16, max_len=32)
    with pytest.raises(ValueError):
        sch.add_sequence(seq(range(17)))                       # longer than a prefill step
    with pytest.raises(ValueError):
        sch.add_sequence(seq(range(8), max_tokens=30))         # exceeds max_model_length
    big = make_sched(blocks=2, max_tokens=64)
    with pytest.raises(ValueError):
        big.add_sequence(seq(range(8), max_tokens=8))          # needs 4 blocks, cache has 2


# ------------------------------------------------------------ chunked prefill
def test_long_prompt_is_prefetched_in_chunks_and_samples_once():
    sch = make_sched(max_tokens=8, chunked=True)               # BS = 4, 8 tokens per step
    a = seq(range(21), max_tokens=3)
    sch.add_sequence(a)                                        # longer than a step: accepted
    spans = []
    for _ in range(3):
        batch, is_prefill = sch.schedule()
        assert is_prefill and batch == [a]
        spans.append((a.num_cached_tokens, a.prefill_end))
        done = sch.postprocess(batch, [777])
        assert done == []
    assert spans == [(0, 8), (8, 16), (16, 21)]
    assert a.completion_token_ids == [777] and a.num_tokens == 22   # only the last chunk sampled
    assert not sch.prefilling and list(sch.running) == [a]
    batch, is_prefill = sch.schedule()
    assert not is_prefill and batch == [a]


def test_chunk_fills_the_rest_of_the_budget_and_blocks_new_admissions():
    sch = make_sched(max_tokens=16, chunked=True)
    a, b, c = seq(range(10)), seq(range(100, 118)), seq(range(200, 203))
    for s in (a, b, c):
        sch.add_sequence(s)
    batch, is_prefill = sch.schedule()
    assert is_prefill and batch == [a, b]
    assert (b.num_cached_tokens, b.prefill_end) == (0, 6)      # 16 - 10 tokens left for b
    assert list(sch.prefilling) == [b] and list(sch.running) == [a]
    sch.postprocess(batch, [1, 999])
    assert a.num_tokens == 11 and b.num_tokens == 18 and b.num_cached_tokens == 6
    batch, is_prefill = sch.schedule()                          # b completes, then c is admitted
    assert is_prefill and batch == [b, c] and b.prefill_end == 18
    sch.postprocess(batch, [2, 3])
    assert b.completion_token_ids == [2] and c.completion_token_ids == [3]
    assert not sch.prefilling and list(sch.running) == [a, b, c]


def test_a_cut_sequence_needs_room_for_one_block():
    sch = make_sched(max_tokens=16, chunked=True)
    a, b = seq(range(14)), seq(range(100, 120))
    sch.add_sequence(a); sch.add_sequence(b)
    batch, _ = sch.schedule()
    assert batch == [a]                                        # 2 tokens of room < one block of 4


def test_preempting_drops_chunk_state():
    sch = make_sched(max_tokens=8, chunked=True)
    a = seq(range(12))
    sch.add_sequence(a)
    batch, _ = sch.schedule()
    sch.postprocess(batch, [0])
    assert a.num_cached_tokens == 8 and list(sch.prefilling) == [a]
    sch.prefilling.clear()
    sch.preempt(a)
    assert a.prefill_end is None and a.num_cached_tokens == 0 and a.block_table == []


def test_prefill_is_deferred_until_enough_tokens_can_be_admitted():
    sch = Scheduler(4, 64, 64, BS, 999, min_prefill_tokens=10)
    first = [seq(range(i * 10, i * 10 + 6), max_tokens=3) for i in range(4)]
    for s in first:
        sch.add_sequence(s)
    batch, is_prefill = sch.schedule()
    assert is_prefill and batch == first                       # nothing is running: no deferral
    sch.postprocess(batch, [1] * 4)
    late = [seq(range(100 + i * 10, 106 + i * 10), max_tokens=2) for i in range(3)]
    for s in late:
        sch.add_sequence(s)
    first[0].max_tokens = 2                                    # will finish after this step
    batch, is_prefill = sch.schedule()
    assert not is_prefill                                      # no free slot anyway
    sch.postprocess(batch, [2] * 4)                            # first[0] finishes -> 1 slot
    batch, is_prefill = sch.schedule()
    assert not is_prefill and len(batch) == 3                  # 6 tokens < 10: keep decoding
    sch.postprocess(batch, [3] * 3)                            # all three finish (max_tokens 3)
    batch, is_prefill = sch.schedule()
    assert is_prefill and batch == late                        # nothing running any more
    one = seq(range(7))
    sch.postprocess(batch, [4] * 3)
    sch.add_sequence(one)
    batch, is_prefill = sch.schedule()
    assert is_prefill and batch == [one]                       # everything that waits fits


def test_abort_removes_a_request_from_any_queue_and_returns_its_blocks():
    sch = make_sched(blocks=16, max_seqs=2, max_tokens=16, chunked=True)
    a, b, c = (seq(range(i * 50, i * 50 + 10), max_tokens=6) for i in range(3))
    for s in (a, b, c):
        sch.add_sequence(s)
    batch, is_prefill = sch.schedule()                        # a whole, b cut after 6 tokens
    assert is_prefill and batch == [a, b] and list(sch.prefilling) == [b]
    sch.postprocess(batch, [1, 0])
    free = sch.block_manager.num_free_blocks
    assert sch.abort(b) and not sch.prefilling                # mid-prefill
    assert sch.block_manager.num_free_blocks > free and b.block_table == []
    assert sch.abort(c) and not sch.waiting                   # never admitted
    assert not sch.abort(c)                                   # already gone
    batch, is_prefill = sch.schedule()
    assert not is_prefill and batch == [a]
    assert sch.abort(a) and sch.is_finished()                 # while decoding
    assert sch.block_manager.num_free_blocks == 16
    assert a.status == SequenceStatus.FINISHED


@pytest.mark.parametrize("native", [False, True])
def test_deallocate_withdraws_blocks_whose_kv_was_never_written(native):
    """Hashes are published at admission; a request dropped before its KV exists must not
    leave them behind (both allocator implementations)."""
    from minivllm_amd.engine.block_manager import make_block_manager, _host_lib
    if native and _host_lib() is None:
        pytest.skip("_host_core.so not built")
    bm = make_block_manager(16, BS, native=native)
    prompt = list(range(100, 100 + 3 * BS + 2))                 # 3 full blocks + a partial one
    a = seq(prompt)
    bm.allocate(a)
    assert a.num_cached_tokens == 0
    bm.deallocate(a, num_valid_tokens=BS + 1)                   # only the first block was computed
    assert bm.num_free_blocks == 16
    b = seq(prompt)
    bm.allocate(b)
    assert b.num_cached_tokens == BS                            # blocks 1 and 2 are recomputed
    bm.deallocate(b)                                            # everything computed this time
    c = seq(prompt)
    bm.allocate(c)
    assert c.num_cached_tokens == 3 * BS
    # a block somebody else still holds keeps its identity
    d = seq(prompt)
    bm.allocate(d)
    bm.deallocate(c, num_valid_tokens=0)
    e = seq(prompt)
    bm.allocate(e)
    assert e.num_cached_tokens == 3 * BS
    for s in (d, e):
        bm.deallocate(s)
    assert bm.num_free_blocks == 16


def test_abort_mid_prefill_leaves_no_uncomputed_blocks_in_the_prefix_cache():
    sch = make_sched(blocks=32, max_seqs=2, max_tokens=8, chunked=True)     # 2 blocks per step
    prompt = list(range(500, 500 + 5 * BS))
    a = seq(prompt, max_tokens=4)
    sch.add_sequence(a)
    batch, is_prefill = sch.schedule()
    assert is_prefill and a.prefill_end == 8 and list(sch.prefilling) == [a]
    sch.postprocess(batch, [0])                                 # first chunk (2 blocks) computed
    assert sch.abort(a)
    b = seq(prompt, max_tokens=4)
    sch.add_sequence(b)
    sch.schedule()
    assert b.num_cached_tokens == 8                             # not 16: blocks 2, 3 never existed
    # a step that raised: even the chunk it was writing does not count
    sch.abort(b, step_failed=True)
    c = seq(prompt, max_tokens=4)
    sch.add_sequence(c)
    sch.schedule()
    # conservative: blocks b held alone are all withdrawn (some of its cache hits may have
    # been blocks another request of the same failed step was only about to write)
    assert c.num_cached_tokens == 0


def test_small_prefill_steps_wait_for_company_but_not_forever():
    import time
    sch = Scheduler(8, 64, 64, BS, 999, min_prefill_tokens=16, max_prefill_delay_s=0.05)
    first = seq(range(8), max_tokens=50)
    sch.add_sequence(first)
    batch, is_prefill = sch.schedule()                        # nothing is decoding: no reason to wait
    assert is_prefill and batch == [first]
    sch.postprocess(batch, [1])
    late = seq(range(100, 106), max_tokens=4)
    sch.add_sequence(late)
    batch, is_prefill = sch.schedule()                        # 6 tokens < 16: keep decoding for now
    assert not is_prefill and batch == [first] and list(sch.waiting) == [late]
    sch.postprocess(batch, [2])
    time.sleep(0.06)
    batch, is_prefill = sch.schedule()                        # waited long enough
    assert is_prefill and batch == [late]
    sch.postprocess(batch, [1])
    a, b = seq(range(200, 208), max_tokens=4), seq(range(300, 308), max_tokens=4)
    sch.add_sequence(a); sch.add_sequence(b)
    batch, is_prefill = sch.schedule()                        # together they reach 16: no wait
    assert is_prefill and batch == [a, b]


def test_mixed_steps_carry_running_sequences_behind_the_prefill_ones():
    sch = Scheduler(4, 24, 64, BS, 999, enable_mixed_batches=True)
    a, b = seq(range(6), max_tokens=9), seq(range(50, 57), max_tokens=9)
    sch.add_sequence(a); sch.add_sequence(b)
    batch, is_prefill = sch.schedule()
    assert is_prefill and batch == [a, b] and sch.last_num_decode == 0   # nobody is running yet
    sch.postprocess(batch, [1, 1])
    c = seq(range(100, 110), max_tokens=9)
    sch.add_sequence(c)
    batch, is_prefill = sch.schedule()
    assert is_prefill and batch == [c, a, b] and sch.last_num_decode == 2
    assert len(a.block_table) * BS >= a.num_tokens                      # riders got their blocks
    sch.postprocess(batch, [5, 2, 2])
    got = (a.completion_token_ids, b.completion_token_ids, c.completion_token_ids)
    assert got == ([1, 2], [1, 2], [5])
    batch, is_prefill = sch.schedule()
    assert not is_prefill and batch == [a, b, c] and sch.last_num_decode == 0
    sch.postprocess(batch, [3, 3, 6])
    # the prompt budget shrinks by the number of riders: 24 - 3 = 21 < 22 prompt tokens, so the
    # newcomer is cut (chunked prefill) and the riders still fit behind its first chunk
    d = seq(range(200, 222), max_tokens=3)
    sch.add_sequence(d)
    batch, is_prefill = sch.schedule()
    assert is_prefill and batch[0] is d and d.prefill_end == 21 and sch.last_num_decode == 3
    assert sch.num_mixed_steps == 2


def test_decode_kernel_lds_pattern_is_conflict_free_in_the_bank_model():
    """The XOR pattern of the decode kernel's LDS tile image against a host model of the LDS
    banking (lane groups per instruction): no extra cycles for K fragment reads, transposed V
    reads and both kinds of staging stores; the pattern it replaced costs 64 per tile, which is
    what the hardware counter showed."""
    import importlib.util
    import pathlib
    path = pathlib.Path(__file__).resolve().parents[1] / "tools" / "lds_bank_model.py"
    spec = importlib.util.spec_from_file_location("lds_bank_model", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert sum(m.per_tile(m.x_decode).values()) == 0
    first = m.per_tile(m.x_first)
    assert first["k_reads"] == 32 and first["v_reads"] == 32
    # the model's pattern is the kernel's
    src = (path.parents[1] / "minivllm_amd" / "csrc" / "attn_decode.hip").read_text()
    assert "(((row & 7) << 1) | ((row >> 2) & 1)) & (CPR - 1)" in src
