"""Consecutive prefill steps: step n + 1 is admitted, prepared and enqueued before step n's tokens
are read back (``Scheduler.schedule_prefill_ahead``, ``ModelRunner.run(next_prefill=...)`` /
``run_prefill_ahead``, ``LLMEngine.step``).  Here the control flow runs on the CPU, where
"enqueue" is the eager step itself: what is checked is that the engine produces the plain
loop's tokens, that the step admitted ahead is always the next one to run, that sequences
aborted while their step is in flight lose their token and nothing else, and that every KV
block comes back."""
import numpy as np
import pytest
import torch

from minivllm_amd.config import EngineConfig, ModelConfig
from minivllm_amd.sampling_parameters import SamplingParams

MODEL = ModelConfig(vocab_size=300, hidden_size=64, num_heads=4, num_kv_heads=2, head_dim=32,
                    intermediate_size=96, num_layers=2, max_position=512, base=10000.0)


def _engine(**kw):
    from minivllm_amd.engine.llm_engine import LLMEngine
    base = dict(model=MODEL, max_num_sequences=8, max_num_batched_tokens=128, block_size=8,
                max_model_length=200, eos=-1, device="cpu", dtype=torch.float32, seed=2,
                decode_prelaunch=False)
    base.update(kw)
    return LLMEngine(EngineConfig(**base))


def _prompts(n=14, shared=0, seed=5):
    rng = np.random.default_rng(seed)
    head = rng.integers(0, 300, size=shared).tolist()
    return [head + rng.integers(0, 300, size=int(rng.integers(20, 60))).tolist() for _ in range(n)]


def _run(ahead, prompts, abort=None, max_tokens=(1, 3, 6), **kw):
    eng = _engine(prefill_prelaunch=ahead, **kw)
    seqs = [eng.add_prompt(p, SamplingParams(temperature=0, max_tokens=max_tokens[i % len(max_tokens)],
                                             ignore_eos=True)) for i, p in enumerate(prompts)]
    done, steps, in_flight, prefill_tokens = {}, 0, 0, 0
    while not eng.is_finished():
        finished, n, is_prefill = eng.step()
        done.update(finished)
        prefill_tokens += n if is_prefill else 0
        steps += 1
        r = eng.model_runner
        if r.has_prefill_ahead():
            in_flight += 1
            if abort is not None and in_flight == abort[0]:
                victim = r.prefill_ahead_info()[0][abort[1]]
                assert eng.scheduler.abort(victim)
        assert steps < 400
    bm = eng.scheduler.block_manager
    assert bm.num_free_blocks == bm.num_blocks
    eng.exit()
    return [done.get(s.seq_id) for s in seqs], in_flight, prefill_tokens


@pytest.mark.parametrize("decode_ahead", [False, True])
def test_outputs_equal_the_plain_loop(decode_ahead):
    prompts = _prompts()
    plain, n0, t0 = _run(False, prompts, decode_prelaunch=decode_ahead)
    ahead, n1, t1 = _run(True, prompts, decode_prelaunch=decode_ahead)
    assert n0 == 0 and n1 >= 3
    assert ahead == plain and all(o is not None for o in ahead)
    assert t0 == t1 == sum(len(p) for p in prompts)


def test_prompts_sharing_cached_blocks_with_the_step_in_flight():
    prompts = _prompts(shared=24)
    plain, _, t0 = _run(False, prompts)
    ahead, n, t1 = _run(True, prompts)
    assert n >= 3 and ahead == plain
    assert t0 == t1 < sum(len(p) for p in prompts)          # the shared blocks were computed once


def test_a_sequence_aborted_while_its_step_is_in_flight():
    prompts = _prompts()
    plain, _, _ = _run(False, prompts)
    ahead, n, _ = _run(True, prompts, abort=(2, 0))
    gone = [i for i, o in enumerate(ahead) if o is None]
    assert len(gone) == 1 and n >= 3
    assert [o for i, o in enumerate(ahead) if i not in gone] == \
        [o for i, o in enumerate(plain) if i not in gone]


def test_every_sequence_of_the_step_in_flight_aborted():
    prompts = _prompts(n=6)
    eng = _engine(prefill_prelaunch=True)
    sp = SamplingParams(temperature=0, max_tokens=2, ignore_eos=True)
    seqs = [eng.add_prompt(p, sp) for p in prompts]
    eng.step()
    r = eng.model_runner
    assert r.has_prefill_ahead()
    gone = list(r.prefill_ahead_info()[0])
    for s in gone:
        assert eng.scheduler.abort(s)
    done = {}
    while not eng.is_finished():
        done.update(eng.step()[0])
    assert not r.has_prefill_ahead()
    assert gone and {s.seq_id for s in seqs} - set(done) == {s.seq_id for s in gone}
    bm = eng.scheduler.block_manager
    assert bm.num_free_blocks == bm.num_blocks
    eng.exit()


def test_a_prompt_cut_at_the_budget_continues_in_the_step_admitted_ahead():
    from minivllm_amd.engine.scheduler import Scheduler
    from minivllm_amd.engine.sequence import Sequence
    sp = SamplingParams(temperature=0, max_tokens=4, ignore_eos=True)

    def sched(**kw):
        return Scheduler(max_num_sequences=8, max_num_batched_tokens=64, max_cached_blocks=64,
                         block_size=8, eos=-1, max_model_length=400, **kw)
    s = sched()
    a, b, c = (Sequence([7 * n + i for i in range(n)], sp, block_size=8) for n in (40, 100, 30))
    for q in (a, b, c):
        s.add_sequence(q)
    first, is_prefill = s.schedule()
    assert is_prefill and first == [a, b] and list(s.prefilling) == [b] and b.prefill_end == 24
    nxt = s.schedule_prefill_ahead()                       # b's chunk of the running step: done early
    assert nxt == [b] and b.num_cached_tokens == 24 and b.prefill_end == 88
    assert s.schedule_prefill_ahead() == []                # one step ahead, not two
    done = s.postprocess(first, [5, 6])                    # a gets its token, b is skipped once
    assert done == [] and a.num_tokens == 41 and b.num_tokens == 100 and b.num_cached_tokens == 24
    last = s.schedule_prefill_ahead()                      # during step 2: b's tail + the newcomer
    assert last == [b, c] and b.num_cached_tokens == 88 and b.prefill_end == 100 and not s.prefilling
    s.postprocess(nxt, [9])                                # step 2 sampled nothing for b
    assert b.num_tokens == 100
    s.postprocess(last, [1, 2])
    assert b.num_tokens == 101 and c.num_tokens == 31 and b.prefill_end is None
    assert s.schedule_prefill_ahead() == []                # nothing waits: the decode phase is next
    # the plain loop makes the same steps
    p = sched()
    a2, b2, c2 = (Sequence([7 * n + i for i in range(n)], sp, block_size=8) for n in (40, 100, 30))
    for q in (a2, b2, c2):
        p.add_sequence(q)
    steps = []
    for _ in range(3):
        seqs, is_prefill = p.schedule()
        assert is_prefill
        steps.append([(q.num_tokens, q.num_cached_tokens, q.prefill_end) for q in seqs])
        p.postprocess(seqs, [0] * len(seqs))
    assert steps == [[(40, 0, 40), (100, 0, 24)], [(100, 24, 88)], [(100, 88, 100), (30, 0, 30)]]
    # riders need the running step's tokens: mixed batches are never admitted ahead
    m = sched(enable_mixed_batches=True)
    for n in (40, 40):
        m.add_sequence(Sequence(list(range(n)), sp, block_size=8))
    m.schedule()
    assert m.schedule_prefill_ahead() == []
