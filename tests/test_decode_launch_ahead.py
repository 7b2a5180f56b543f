"""Steady-state decode launches step n + 1 before step n's tokens are read back (model
runner, ``run`` / ``run_prelaunched``).  The control flow that matters on the GPU - batch-size
buckets with padding rows, static input / output buffers, two alternating staging blocks, rows
of sequences that finished in the meantime - is replayed here on the CPU with stand-in graphs,
and every staging block is validated against its batch before it is "uploaded" (a stale
block-table image or a slot outside the sequence's blocks is an out-of-bounds access on a
GPU)."""
import numpy as np
import pytest
import torch

from minivllm_amd.config import EngineConfig, ModelConfig
from minivllm_amd.sampling_parameters import SamplingParams

MODEL = ModelConfig(vocab_size=300, hidden_size=64, num_heads=4, num_kv_heads=2, head_dim=32,
                    intermediate_size=96, num_layers=2, max_position=512, base=10000.0)


class _StandInGraph:
    """What a captured decode graph does, eagerly: forward + sample on the first ``bs`` rows of
    the static input buffer, result into the static output tensor."""

    def __init__(self, runner, bs):
        self.r, self.bs = runner, bs

    def replay(self):
        from minivllm_amd.utils.context import reset_context, set_context
        r, bs = self.r, self.bs
        t = r._decode_tensors(bs)
        live = int((t["context_lens"] > 0).sum())            # padding rows sit at the end
        set_context(False, slot_mapping=t["slot_mapping"][:live], context_lens=t["context_lens"][:live],
                    block_tables=t["block_tables"][:live], positions=t["positions"][:live],
                    max_context_len=r.cfg.max_model_length, num_seqs=live)
        with torch.inference_mode():
            out = r._forward_and_sample({k: (v[:live] if k != "step" else v) for k, v in t.items()})
        reset_context()
        r._graph_out[bs][:live] = out.to(r._graph_out[bs].dtype)


def _engine(buckets=(1, 2, 4, 8), **kw):
    from minivllm_amd.engine import model_runner as mr
    from minivllm_amd.engine.llm_engine import LLMEngine
    base = dict(model=MODEL, max_num_sequences=8, max_num_batched_tokens=256, block_size=8,
                max_model_length=200, eos=-1, device="cpu", dtype=torch.float32, seed=2)
    base.update(kw)
    eng = LLMEngine(EngineConfig(**base))
    r = eng.model_runner
    if buckets:
        r._graph_sizes = sorted(buckets)
        r._graph_out = {b: torch.zeros(b, dtype=torch.int64) for b in buckets}
        r.graphs = {b: _StandInGraph(r, b) for b in buckets}
    return eng


@pytest.fixture(autouse=True)
def _validate_every_staging_block(monkeypatch):
    from minivllm_amd.engine import model_runner as mr
    monkeypatch.setattr(mr, "_CHECK_BATCHES", True)


def _run(ahead, stop=(), abort_at=None, lens=(5, 40, 17, 33, 40, 9, 28, 40), buckets=(1, 2, 4, 8)):
    rng = np.random.default_rng(9)
    prompts = [rng.integers(0, 300, size=int(rng.integers(5, 60))).tolist() for _ in lens]
    eng = _engine(buckets=buckets, decode_prelaunch=ahead)
    seqs = [eng.add_prompt(p, SamplingParams(temperature=0, max_tokens=n, ignore_eos=True,
                                             stop_token_ids=list(stop)))
            for p, n in zip(prompts, lens)]
    done, steps, launched = {}, 0, 0
    while not eng.is_finished():
        finished, _, _ = eng.step()
        done.update(finished)
        launched += eng.model_runner.has_prelaunched()
        steps += 1
        if steps == abort_at:
            assert eng.scheduler.abort(seqs[4])
        assert steps < 500
    bm = eng.scheduler.block_manager
    assert bm.num_free_blocks == bm.num_blocks
    eng.exit()
    return [done.get(s.seq_id) for s in seqs], launched


@pytest.mark.parametrize("buckets", [(1, 2, 4, 8), ()])
def test_outputs_equal_the_plain_loop(buckets):
    plain, n0 = _run(False, buckets=buckets)
    ahead, n1 = _run(True, buckets=buckets)
    assert n0 == 0 and n1 > 10
    assert ahead == plain


def test_unforeseen_finishes_and_an_abort_in_flight():
    plain, _ = _run(False)
    stop = [plain[1][20], plain[7][11]]
    plain_s, _ = _run(False, stop, abort_at=12)
    ahead_s, n = _run(True, stop, abort_at=12)
    assert n > 5 and ahead_s == plain_s
    assert len(plain_s[1]) <= 21 and len(plain_s[7]) <= 12 and plain_s[4] is None


def test_late_arrivals_wait_for_the_step_in_flight_and_then_join():
    rng = np.random.default_rng(3)
    prompts = [rng.integers(0, 300, size=20).tolist() for _ in range(5)]
    sp = SamplingParams(temperature=0, max_tokens=30, ignore_eos=True)

    def go(ahead):
        eng = _engine(decode_prelaunch=ahead)
        seqs = [eng.add_prompt(p, sp) for p in prompts[:3]]
        done, steps = {}, 0
        while not eng.is_finished():
            finished, _, _ = eng.step()
            done.update(finished)
            steps += 1
            if steps == 6:
                seqs += [eng.add_prompt(p, sp) for p in prompts[3:]]
        eng.exit()
        return [done[s.seq_id] for s in seqs]

    assert go(True) == go(False)
