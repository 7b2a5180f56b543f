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


def _run(ahead, stop=(), abort_at=None, lens=(5, 40, 17, 33, 40, 9, 28, 40), buckets=(1, 2, 4, 8),
         info=None, **kw):
    rng = np.random.default_rng(9)
    prompts = [rng.integers(0, 300, size=int(rng.integers(5, 60))).tolist() for _ in lens]
    eng = _engine(buckets=buckets, decode_prelaunch=ahead, **kw)
    seqs = [eng.add_prompt(p, SamplingParams(temperature=0, max_tokens=n, ignore_eos=True,
                                             stop_token_ids=list(stop)))
            for p, n in zip(prompts, lens)]
    done, steps, launched = {}, 0, 0
    while not eng.is_finished():
        finished, _, _ = eng.step()
        done.update(finished)
        launched += eng.model_runner.has_prelaunched()
        if info is not None and eng.model_runner.has_prelaunched():
            info["edges"] = info.get("edges", 0) + (not eng.model_runner._spec["batch"].tables_unchanged)
        steps += 1
        if steps == abort_at:
            assert eng.scheduler.abort(seqs[4])
        assert steps < 500
    bm = eng.scheduler.block_manager
    assert bm.num_free_blocks == bm.num_blocks
    if info is not None:
        info.update(decode_steps=steps, hashes=bm.stats()["cached_hashes"],
                    preemptions=eng.scheduler.num_preemptions)
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


def test_steps_across_block_edges_are_launched_ahead_too():
    """Block size 8, eight sequences of different lengths: nearly every decode step starts a
    new KV block or completes one for somebody.  Those steps are launched ahead as well (blocks
    given ahead, hashes published when the step is collected), the tokens and the set of
    published block hashes are those of the plain loop."""
    plain_i, edges_i, no_edges_i = {}, {}, {}
    plain, _ = _run(False, info=plain_i)
    ahead, n_ahead = _run(True, info=edges_i)
    old, n_old = _run(True, info=no_edges_i, decode_prelaunch_edges=False)
    assert ahead == plain == old
    assert edges_i["edges"] >= 6                        # steps whose tables grew, launched ahead
    assert no_edges_i.get("edges", 0) == 0
    # all steps but the prefill step and the one after each sequence's (foreseen) finish
    assert n_ahead > 2 * n_old and n_ahead >= edges_i["decode_steps"] - 10
    assert edges_i["hashes"] == plain_i["hashes"] == no_edges_i["hashes"] > 0


def test_launch_ahead_steps_aside_when_blocks_run_out():
    """A cache too small for everybody: a step that would need more blocks than are free is not
    launched ahead (the scheduler preempts as usual), and the outcome is the plain loop's."""
    lens = (40, 40, 40, 40, 40, 40)
    plain_i, ahead_i = {}, {}
    plain, _ = _run(False, lens=lens, info=plain_i, max_cached_blocks=40)
    ahead, n = _run(True, lens=lens, info=ahead_i, max_cached_blocks=40)
    assert plain_i["preemptions"] > 0 and ahead_i["preemptions"] > 0
    assert n > 5 and ahead == plain


def test_a_saturated_engine_keeps_launching_ahead_while_nobody_can_be_admitted():
    """More requests than sequence slots: while every slot is taken the waiting requests
    cannot be admitted, so the decode batch stays the same and its steps are launched ahead;
    the step after a sequence's last token is scheduled normally and admits the next request,
    exactly when the plain loop does."""
    lens = (9, 30, 14, 22, 30, 11, 26, 30, 7, 19, 30, 12, 25, 16, 30, 8)
    kw = dict(lens=lens, buckets=(1, 2, 4), max_num_sequences=4)
    plain_i, ahead_i = {}, {}
    plain, n0 = _run(False, info=plain_i, **kw)
    ahead, n1 = _run(True, info=ahead_i, **kw)
    assert n0 == 0 and ahead == plain and all(o is not None for o in ahead)
    assert ahead_i["decode_steps"] == plain_i["decode_steps"]          # same steps, same order
    # every step but the prefill steps and the first decode step after each admission
    assert n1 >= ahead_i["decode_steps"] - 3 * len(lens)
    assert n1 > 0.5 * ahead_i["decode_steps"]
