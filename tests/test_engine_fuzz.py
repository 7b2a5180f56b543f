"""Randomised scheduling stress on the CPU path: requests arrive over time, some are aborted,
the KV cache is small enough to force preemption, the token budget small enough to force chunked
prefill, with and without mixed batches / deferred prefill.  Greedy outputs must not depend on any
of that: they are compared with one plain run, and the block pool must be whole again at the end."""
import numpy as np
import pytest
import torch

from minivllm_amd.config import EngineConfig, ModelConfig
from minivllm_amd.sampling_parameters import SamplingParams

MODEL = ModelConfig(vocab_size=300, hidden_size=64, num_heads=4, num_kv_heads=2, head_dim=32,
                    intermediate_size=96, num_layers=2, max_position=512, base=10000.0)


def _engine(**kw):
    from minivllm_amd.engine.llm_engine import LLMEngine
    base = dict(model=MODEL, max_num_sequences=5, max_num_batched_tokens=48, block_size=8,
                max_model_length=160, eos=-1, device="cpu", dtype=torch.float32, seed=2)
    base.update(kw)
    return LLMEngine(EngineConfig(**base))


def _workload(seed, n=14):
    rng = np.random.default_rng(seed)
    shared = [rng.integers(0, 300, size=int(rng.integers(8, 30))).tolist() for _ in range(2)]
    prompts, params = [], []
    for _ in range(n):
        head = shared[int(rng.integers(0, 2))] if rng.random() < 0.5 else []
        prompts.append(head + rng.integers(0, 300, size=int(rng.integers(1, 70))).tolist())
        params.append(SamplingParams(temperature=0, max_tokens=int(rng.integers(1, 30)), ignore_eos=True))
    return prompts, params


def _drive(eng, prompts, params, seed):
    """Random arrivals and aborts until everything has drained; returns (sequences by request
    index, aborted request indices, finished {seq_id: tokens}, steps)."""
    rng = np.random.default_rng(100 + seed)
    pending = list(range(len(prompts)))
    seqs, aborted, done = {}, set(), {}
    steps = 0
    while pending or not eng.is_finished():
        # arrivals: a few at a time, at random steps
        while pending and rng.random() < 0.4:
            i = pending.pop(0)
            seqs[i] = eng.add_prompt(prompts[i], params[i])
        # now and then a client goes away
        live = [i for i, s in seqs.items() if i not in aborted and s.seq_id not in done
                and not s.is_finished]
        if live and rng.random() < 0.05:
            victim = live[int(rng.integers(0, len(live)))]
            assert eng.scheduler.abort(seqs[victim])
            aborted.add(victim)
        if eng.is_finished():
            continue
        finished, _, _ = eng.step()
        done.update(finished)
        steps += 1
        assert steps < 5000, "no progress"
    return seqs, aborted, done, steps


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("policy", [
    dict(),
    dict(enable_mixed_batches=True),
    dict(enable_mixed_batches=True, min_prefill_tokens=24, max_prefill_delay_ms=1.0),
    dict(min_prefill_tokens=32),
])
def test_outputs_do_not_depend_on_scheduling(seed, policy):
    prompts, params = _workload(seed)
    plain = _engine(max_num_sequences=16, max_num_batched_tokens=2048, max_cached_blocks=512)
    want = plain.generate(prompts, params)["token_ids"]
    plain.exit()

    eng = _engine(max_cached_blocks=30, **policy)          # 240 tokens of KV for up to 5 sequences
    seqs, aborted, done, steps = _drive(eng, prompts, params, seed)
    bm = eng.scheduler.block_manager
    assert bm.num_free_blocks == bm.num_blocks
    for i, s in seqs.items():
        if i in aborted:
            got = s.completion_token_ids
            assert got == want[i][:len(got)]                 # whatever it produced was right
        else:
            assert done[s.seq_id] == want[i], (i, policy)
    if "enable_mixed_batches" in policy:
        assert eng.scheduler.num_mixed_steps > 0
    print(f"preemptions={eng.scheduler.num_preemptions} steps={steps} aborted={len(aborted)} "
          f"prefix_hits={bm.num_prefix_hits}")
    eng.exit()


@pytest.mark.parametrize("seed", [0, 1])
def test_fp8_cache_under_the_same_stress(seed):
    """fp8 KV cache: a sequence that is preempted or hits a cached prefix re-reads quantised K/V
    where an undisturbed one used the exact values, so token-for-token equality with a plain run
    is not a property of this mode; everything structural is: every request finishes with its
    length, tokens are in range, most tokens agree, no block is lost."""
    prompts, params = _workload(seed)
    plain = _engine(max_num_sequences=16, max_num_batched_tokens=2048, max_cached_blocks=512,
                    kv_cache_dtype="fp8")
    want = plain.generate(prompts, params)["token_ids"]
    plain.exit()
    eng = _engine(max_cached_blocks=30, kv_cache_dtype="fp8", enable_mixed_batches=True)
    assert eng.model_runner.kv_cache.dtype == torch.float8_e4m3fn
    seqs, aborted, done, steps = _drive(eng, prompts, params, seed)
    bm = eng.scheduler.block_manager
    assert bm.num_free_blocks == bm.num_blocks
    same = total = 0
    for i, s in seqs.items():
        if i in aborted:
            continue
        got = done[s.seq_id]
        assert len(got) == len(want[i]) and all(0 <= t < MODEL.vocab_size for t in got)
        same += sum(int(a == b) for a, b in zip(got, want[i]))
        total += len(got)
    assert same / total > 0.6, (same, total)
    eng.exit()


def test_retry_after_abort_mid_chunked_prefill_matches_a_plain_run():
    """A prompt longer than the token budget is dropped after its first chunk (a client
    time-out) and submitted again: the retry must not attend to blocks whose KV the aborted
    request never wrote."""
    rng = np.random.default_rng(7)
    prompt = rng.integers(0, 300, size=130).tolist()            # 3 chunks of 48 tokens
    sp = SamplingParams(temperature=0, max_tokens=12, ignore_eos=True)
    ref = _engine()
    want = ref.generate([prompt], sp)["token_ids"][0]
    ref.exit()
    eng = _engine()
    first = eng.add_prompt(prompt, sp)
    eng.step()                                                  # first chunk only
    assert list(eng.scheduler.prefilling) == [first]
    assert eng.scheduler.abort(first)
    got = eng.generate([prompt], sp)["token_ids"][0]
    assert got == want
    bm = eng.scheduler.block_manager
    assert bm.num_free_blocks == bm.num_blocks
    eng.exit()
