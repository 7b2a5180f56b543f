"""Logit-level checks of the GPU model runner at real geometry against the fp32 ``torch_ref``
model carrying the same weights: prefill over several KV blocks, decode (eager and hipGraph
replay), chunked prefill and a cached prefix.  Geometry: Qwen3-0.6B width with 16/8 heads and a
32/8-head variant (decode groups of 2 and 4 query heads per KV head), head_dim 128, KV blocks
of 256 tokens, contexts that span up to four blocks.  The reference has no tests at all
(its __main__ blocks only print shapes), so this is where the engine's numerics are pinned.

The error measure is max |gpu - oracle| relative to the largest oracle logit of the step; the
oracle runs in fp32 on weights rounded to bf16, so what is measured is the bf16 arithmetic of
the kernels (one rounding per op), a few parts in a thousand over these depths."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from minivllm_amd.config import EngineConfig, ModelConfig
from minivllm_amd.sampling_parameters import SamplingParams

BS = 256
VOCAB = 8192
GEOMETRIES = {
    "16q8kv": dict(hidden_size=1024, num_heads=16, num_kv_heads=8, intermediate_size=3072,
                   num_layers=3),
    "32q8kv": dict(hidden_size=1024, num_heads=32, num_kv_heads=8, intermediate_size=2048,
                   num_layers=2),
    # five query heads per KV head, biased QKV projection and no q/k norm (Qwen2.5-style): the
    # attention kernels' transforms run without norm weights
    # 64-wide heads, four query heads per KV head
    "16q4kv-d64": dict(hidden_size=1024, num_heads=16, num_kv_heads=4, intermediate_size=2048,
                       num_layers=2, head_dim=64),
    "40q8kv-bias": dict(hidden_size=1024, num_heads=40, num_kv_heads=8, intermediate_size=2048,
                        num_layers=2, qkv_bias=True),
}
REL_TOL = 0.03            # vs the fp32 oracle (observed: about 0.01)
REL_TOL_SAME = 0.015      # GPU against GPU through another path (graph / chunked / cached)


def _cfg(geom, **kw):
    model = ModelConfig(**{**dict(vocab_size=VOCAB, head_dim=128, max_position=4096, base=1e6,
                                  tie_word_embeddings=True), **GEOMETRIES[geom]})
    base = dict(model=model, max_num_sequences=8, max_num_batched_tokens=4096, block_size=BS,
                max_model_length=2048, eos=-1, max_cached_blocks=64, dtype=torch.bfloat16,
                device="cuda", enforce_eager=True, seed=3, tune_decode_gemms=False,
                prefill_graph_capture_after=1)
    base.update(kw)
    return EngineConfig(**base).validate()


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from minivllm_amd.ops import hip  # noqa: F401
    return True


def _runner(cfg, weights_from=None):
    from minivllm_amd.engine.model_runner import ModelRunner
    r = ModelRunner(cfg, 0)
    if weights_from is not None:
        sd = {k: v.to(dtype=next(r.model.parameters()).dtype, device=r.device)
              for k, v in weights_from.model.state_dict().items()}
        r.model.load_state_dict(sd)
    store = []
    orig = r._sample

    def spy(hidden, temps, step):
        store.append(r.model.lm_head.local_logits(hidden).float().cpu())
        return orig(hidden, temps, step)
    r._sample = spy
    r.logit_log = store
    return r


class _Driver:
    """Feeds one runner the steps a scheduler would, with fixed tokens (teacher forcing), and
    returns the logits of each step."""

    def __init__(self, runner):
        from minivllm_amd.engine.block_manager import BlockManager
        self.r = runner
        self.bm = BlockManager(runner.num_kv_blocks, BS)

    def seqs(self, prompts):
        from minivllm_amd.engine.sequence import Sequence
        out = [Sequence(list(p), SamplingParams(temperature=0, max_tokens=64), block_size=BS)
               for p in prompts]
        for s in out:
            self.bm.allocate(s)
        return out

    def logits(self, n_before):
        log = self.r.logit_log
        assert len(log) == n_before + 1 or self.r.graphs      # a replayed graph logs nothing
        return log[-1]

    def prefill(self, seqs, ends=None):
        """``ends``: per sequence the token index this step stops at (chunked prefill)."""
        for i, s in enumerate(seqs):
            s.prefill_end = None if ends is None else ends[i]
        n = len(self.r.logit_log)
        self.r.run(seqs, True)
        for i, s in enumerate(seqs):
            if s.prefill_end is not None and s.prefill_end < s.num_tokens:
                s.num_cached_tokens = s.prefill_end
            s.prefill_end = None
        return self.logits(n)

    def decode(self, seqs, tokens):
        for s, t in zip(seqs, tokens):
            s.append_token(int(t))
            self.bm.append(s)
        n = len(self.r.logit_log)
        # a new list object every step: handed the same list again, the runner takes its
        # steady-state shortcut and feeds the tokens IT sampled, not the forced ones
        self.r.run(list(seqs), False)
        if self.r.graphs:
            bucket = self.r._graph_bucket(len(seqs))
            return self.r._graph_logits[bucket][:len(seqs)].float().cpu()
        return self.logits(n)


def _rel_err(got, want):
    return float((got - want).abs().max() / want.abs().max())


def _prompts(lengths, seed):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, VOCAB, size=n).tolist() for n in lengths]


LENGTHS = (700, 300, 1000, 257, 64, 513)          # 1 to 4 KV blocks, on and off block edges


@pytest.mark.parametrize("geom", list(GEOMETRIES))
def test_prefill_and_decode_logits_match_the_fp32_oracle(gpu, geom):
    g = _runner(_cfg(geom))
    c = _runner(_cfg(geom, device="cpu", dtype=torch.float32), weights_from=g)
    prompts = _prompts(LENGTHS, 1)
    rng = np.random.default_rng(2)
    forced = [rng.integers(0, VOCAB, size=len(prompts)).tolist() for _ in range(3)]
    results = {}
    for name, r in (("gpu", g), ("cpu", c)):
        d = _Driver(r)
        seqs = d.seqs(prompts)
        steps = [d.prefill(seqs)]
        for toks in forced:
            steps.append(d.decode(seqs, toks))
        results[name] = steps
    errs = [_rel_err(a, b) for a, b in zip(results["gpu"], results["cpu"])]
    print(f"[{geom}] relative logit error per step (prefill, 3 x decode): {errs}")
    assert max(errs) < REL_TOL, errs
    # the greedy choice agrees wherever the oracle's margin is not within the error
    for a, b in zip(results["gpu"], results["cpu"]):
        top2 = b.topk(2, dim=-1).values
        clear = (top2[:, 0] - top2[:, 1]) > 2 * REL_TOL * b.abs().max()
        assert bool((a.argmax(-1) == b.argmax(-1))[clear].all())
    g.exit(); c.exit()


@pytest.mark.parametrize("geom", list(GEOMETRIES))
def test_graph_replay_logits_match_eager_and_oracle(gpu, geom):
    """The same decode steps through a captured hipGraph (batch padded to the bucket of 8)
    and eagerly: logits must agree to rounding, not merely most greedy tokens."""
    e = _runner(_cfg(geom))
    gr = _runner(_cfg(geom, enforce_eager=False, graph_batch_sizes=[4, 8]), weights_from=e)
    assert gr.graphs
    c = _runner(_cfg(geom, device="cpu", dtype=torch.float32), weights_from=e)
    prompts = _prompts(LENGTHS[:5], 3)                          # 5 sequences -> bucket 8
    rng = np.random.default_rng(4)
    forced = [rng.integers(0, VOCAB, size=len(prompts)).tolist() for _ in range(4)]
    out = {}
    for name, r in (("eager", e), ("graph", gr), ("cpu", c)):
        d = _Driver(r)
        seqs = d.seqs(prompts)
        d.prefill(seqs)
        out[name] = [d.decode(seqs, toks) for toks in forced]
    e_vs_g = [_rel_err(a, b) for a, b in zip(out["graph"], out["eager"])]
    g_vs_c = [_rel_err(a, b) for a, b in zip(out["graph"], out["cpu"])]
    print(f"[{geom}] graph vs eager {e_vs_g}; graph vs oracle {g_vs_c}")
    assert max(e_vs_g) < REL_TOL_SAME, e_vs_g
    assert max(g_vs_c) < REL_TOL, g_vs_c
    e.exit(); gr.exit(); c.exit()


@pytest.mark.parametrize("geom", list(GEOMETRIES))
def test_chunked_prefill_and_cached_prefix_logits(gpu, geom):
    """A 900-token prompt prefilled whole, in chunks of 512 + 388 (the second chunk attends to
    the first through the paged cache across a block edge), and with its first two blocks
    taken from the prefix cache: the last-token logits must agree with each other and with
    the oracle."""
    g = _runner(_cfg(geom))
    c = _runner(_cfg(geom, device="cpu", dtype=torch.float32), weights_from=g)
    prompt = _prompts((900,), 5)[0]
    other = _prompts((300,), 6)[0]
    dc = _Driver(c)
    want = dc.prefill(dc.seqs([prompt, other]))
    d = _Driver(g)
    whole = d.prefill(d.seqs([prompt, other]))
    # chunked: first 512 tokens of the long prompt + the whole short one, then the rest
    d2 = _Driver(_runner(_cfg(geom), weights_from=g))
    s_long, s_short = d2.seqs([prompt, other])
    first = d2.prefill([s_long, s_short], ends=[512, len(other)])
    second = d2.prefill([s_long], ends=[len(prompt)])
    chunked = torch.stack([second[0], first[-1]])
    # cached prefix: the allocator finds the two full blocks the first request left behind
    s_again = d.seqs([prompt])[0]
    assert s_again.num_cached_tokens == 512 + 256
    cached = d.prefill([s_again])
    errs = dict(whole_vs_oracle=_rel_err(whole, want), chunked_vs_whole=_rel_err(chunked, whole),
                cached_vs_whole=_rel_err(cached[0], whole[0]))
    print(f"[{geom}] {errs}")
    assert errs["whole_vs_oracle"] < REL_TOL
    assert errs["chunked_vs_whole"] < REL_TOL_SAME and errs["cached_vs_whole"] < REL_TOL_SAME
    g.exit(); c.exit(); d2.r.exit()


def test_decode_logits_through_the_hand_written_gemm_kernels(gpu):
    """Decode batches of 24 (mid-batch MFMA kernel, forced for every projection incl. the
    fused SwiGLU, a K split, and K-slice sums folded into the following norm) and of 6 rows
    (weight-streaming kernel) against the oracle."""
    from minivllm_amd.ops import hip
    geom = "16q8kv"
    g = _runner(_cfg(geom, max_num_sequences=32))
    c = _runner(_cfg(geom, max_num_sequences=32, device="cpu", dtype=torch.float32), weights_from=g)
    shapes = {(4096, 1024, False): (64, 2, True), (1024, 2048, False): (64, 4, True),
              (6144, 1024, True): (128, 1, True), (1024, 3072, False): (128, 3, False)}
    saved = dict(hip._MID_CHOICE)
    try:
        for (n, k, swi), choice in shapes.items():
            hip._MID_CHOICE[(32, n, k, swi)] = choice
        # o / down projections with their K-slice sum left to the norm kernel that follows
        hip._MID_CHOICE[(32, 1024, 2048, "norm")] = (64, 4, True)
        hip._MID_CHOICE[(32, 1024, 3072, "norm")] = (128, 3, False)
        for n_seqs in (24, 6):
            rng = np.random.default_rng(10 + n_seqs)
            prompts = _prompts([int(x) for x in rng.integers(20, 300, size=n_seqs)], 11 + n_seqs)
            forced = [rng.integers(0, VOCAB, size=n_seqs).tolist() for _ in range(2)]
            res = {}
            for name, r in (("gpu", g), ("cpu", c)):
                d = _Driver(r)
                seqs = d.seqs(prompts)
                d.prefill(seqs)
                res[name] = [d.decode(seqs, t) for t in forced]
            errs = [_rel_err(a, b) for a, b in zip(res["gpu"], res["cpu"])]
            print(f"[{n_seqs} rows] relative logit error per decode step: {errs}")
            assert max(errs) < REL_TOL, errs
    finally:
        hip._MID_CHOICE.clear()
        hip._MID_CHOICE.update(saved)
    g.exit(); c.exit()


@pytest.mark.parametrize("num_seqs", [16, 64])
def test_prefill_graphs_of_many_sequences_match_eager(gpu, num_seqs):
    """A prefill step of 16 / 64 sequences replays a hipGraph too: the first step of a (token
    bucket, sequence bucket) runs eagerly on the graph's static buffers and is recorded, later
    steps of that shape replay it (other prompts, other lengths, cached prefixes).  Logits of
    the recording step and of replays against the eager runner."""
    kw = dict(max_num_sequences=64, max_num_batched_tokens=8192, max_cached_blocks=160)
    # no logit spy here: it copies to the host inside ``_sample``, which a capture cannot record
    # (the runner would mark the shape as eager-only); ``_last_step_logits`` is read instead
    from minivllm_amd.engine.model_runner import ModelRunner
    e = ModelRunner(_cfg("16q8kv", **kw), 0)
    g = ModelRunner(_cfg("16q8kv", enforce_eager=False, graph_batch_sizes=[8], **kw), 0)
    g.model.load_state_dict(e.model.state_dict())
    rng = np.random.default_rng(40 + num_seqs)
    total = 4000 if num_seqs == 16 else 6000

    def lengths():                       # num_seqs uneven lengths that add up to ``total``
        while True:
            cuts = np.sort(rng.choice(np.arange(1, total), size=num_seqs - 1, replace=False))
            lens = np.diff(np.concatenate([[0], cuts, [total]])).tolist()
            if max(lens) <= 1500:
                return lens

    rounds = [lengths() for _ in range(3)]
    out = {}
    for name, r in (("eager", e), ("graph", g)):
        d = _Driver(r)
        logs = []
        for lens in rounds:
            prompts = [np.random.default_rng(sum(lens[:i]) + n).integers(0, VOCAB, size=n).tolist()
                       for i, n in enumerate(lens)]
            seqs = d.seqs(prompts)
            r.run(seqs, True)
            logs.append(r._last_step_logits[:len(seqs)].float().cpu())
            for s in seqs:
                d.bm.deallocate(s)
        out[name] = logs
    keys = list(g.prefill_graphs)
    assert any(k[1] >= num_seqs and k[0] >= total for k in keys), keys
    assert g.prefill_graph_replays >= 2, g.prefill_graph_replays
    errs = [_rel_err(a, b) for a, b in zip(out["graph"], out["eager"])]
    print(f"[{num_seqs} sequences] prefill graph vs eager, relative logit error per step: {errs}")
    assert max(errs) < REL_TOL_SAME, errs
    e.exit(); g.exit()


def test_prefill_with_norms_folded_into_the_gemm_epilogues(gpu):
    """Prefill with no (residual add + RMSNorm) pass in the layer loop: o / down add the residual
    stream and sum the squares of the new rows in their epilogue, qkv / gate_up (weights carrying
    the norms' gamma) scale their output rows.  Forced on for every prefill row bucket here (the
    engine decides by measurement); logits of two prefill steps (one behind a cached prefix)
    against the fp32 oracle carrying the same folded weights and against the unfused GPU path."""
    from minivllm_amd.engine.model_runner import ModelRunner
    from minivllm_amd.ops import hip
    kw = dict(max_num_batched_tokens=4096, prefill_graph_max_tokens=0)
    g = ModelRunner(_cfg("16q8kv", **kw), 0)
    assert g.norms_folded
    c = ModelRunner(_cfg("16q8kv", device="cpu", dtype=torch.float32, **kw), 0)
    c.model.load_state_dict({k: v.float().cpu() for k, v in g.model.state_dict().items()})
    prompts = _prompts((700, 300, 1000, 257, 64, 513), 11)
    second = [p[:512] + _prompts((90,), 12 + i)[0] for i, p in enumerate(prompts[:3])]   # 2 cached blocks each
    calls = {"n": 0}
    orig = hip.big_linear_resid

    def counted(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    saved = dict(hip._BIG_CHOICE)
    layer = g.model.model.layers[0]
    shapes = [tuple(layer.self_attn.o_proj.weight.shape), tuple(layer.mlp.down_proj.weight.shape)]
    out = {}
    try:
        hip.big_linear_resid = counted
        for name, r, force in (("unfused", g, False), ("fused", g, True), ("cpu", c, False)):
            for b in hip.BIG_ROW_BUCKETS:
                for n, k in shapes:
                    hip._BIG_CHOICE[(b, n, k, "resid")] = force
            d = _Driver(r)
            logs = []
            for batch in (prompts, second):
                seqs = d.seqs(batch)
                r.run(seqs, True)
                logs.append(r._last_step_logits[:len(seqs)].float().cpu())
            out[name] = logs
    finally:
        hip.big_linear_resid = orig
        hip._BIG_CHOICE.clear()
        hip._BIG_CHOICE.update(saved)
    n_layers = g.cfg.model.num_layers
    assert calls["n"] == 2 * 2 * n_layers, calls            # o and down of every layer, two steps
    vs_oracle = [_rel_err(a, b) for a, b in zip(out["fused"], out["cpu"])]
    vs_unfused = [_rel_err(a, b) for a, b in zip(out["fused"], out["unfused"])]
    print(f"folded-norm prefill: vs oracle {vs_oracle}, vs unfused kernels {vs_unfused}")
    assert max(vs_oracle) < REL_TOL and max(vs_unfused) < REL_TOL_SAME
    g.exit(); c.exit()
