"""Config parsing (the reference's flat dict), safetensors loading with TP sharding, and engine
behaviour on the CPU path: prefix-cache reuse and recompute preemption must not change greedy
outputs; API surface of the compat package."""
import os

import numpy as np
import pytest
import torch

from minivllm_amd.config import EngineConfig, ModelConfig, as_engine_config, model_preset
from minivllm_amd.sampling_parameters import SamplingParams

REFERENCE_DEMO_CONFIG = {       # the dict the reference's main.py passes to LLMEngine
    'max_num_sequences': 16, 'max_num_batched_tokens': 1024, 'max_cached_blocks': 1024,
    'block_size': 256, 'world_size': 1, 'model_name_or_path': 'gpt2', 'enforce_eager': True,
    'vocab_size': 151643, 'hidden_size': 1024, 'num_heads': 16, 'head_dim': 64, 'num_kv_heads': 8,
    'intermediate_size': 3072, 'num_layers': 28, 'tie_word_embeddings': True, 'base': 10000,
    'rms_norm_epsilon': 1e-6, 'qkv_bias': False, 'scale': 1, 'max_position': 128, 'ffn_bias': True,
    'max_num_batch_tokens': 4096, 'max_model_length': 128, 'gpu_memory_utilization': 0.9,
    'eos': 151642,
}


def test_reference_flat_dict_is_accepted():
    cfg = as_engine_config(dict(REFERENCE_DEMO_CONFIG, device="cpu"))
    assert cfg.model.head_dim == 64 and cfg.model.ffn_bias and cfg.model.vocab_size == 151643
    assert cfg.max_num_batched_tokens == 1024            # scheduler key wins over the warm-up key
    assert cfg.block_size == 256 and cfg.eos == 151642 and cfg.max_model_length == 128


def test_config_validation_and_presets():
    assert model_preset("qwen3-8b").num_layers == 36 and model_preset("qwen3-32b").num_heads == 64
    assert 7.5e9 < model_preset("qwen3-8b").num_parameters() < 8.5e9
    with pytest.raises(KeyError):
        EngineConfig.from_dict({"max_num_seqz": 3})
    with pytest.raises(ValueError):
        EngineConfig.from_dict({"block_size": 48, "device": "cpu"})
    with pytest.raises(ValueError):
        EngineConfig.from_dict({"world_size": 3, "device": "cpu"})
    c = EngineConfig.from_dict({"max_num_seqs": 7, "model_preset": "qwen3-tiny", "device": "cpu"})
    assert c.max_num_sequences == 7 and c.model.hidden_size == 128
    assert model_preset("qwen3-8b").kv_bytes_per_token(torch.bfloat16) == 144 * 1024


def test_safetensors_roundtrip_with_tp_shards(tmp_path):
    from safetensors.torch import save_file
    from minivllm_amd.models.qwen3 import Qwen3ForCausalLM, RandomCheckpoint
    from minivllm_amd.parallel import group as pg
    from minivllm_amd.utils.loader import load_weights_from_checkpoint, model_config_from_hf
    import json
    cfg = ModelConfig(vocab_size=96, hidden_size=32, num_heads=4, num_kv_heads=2, head_dim=16,
                      intermediate_size=48, num_layers=2, max_position=64, tie_word_embeddings=False)
    full = dict(RandomCheckpoint(cfg, seed=2, std=0.1).items())
    save_file({k: v.contiguous() for k, v in full.items()}, str(tmp_path / "model.safetensors"))
    (tmp_path / "config.json").write_text(json.dumps(dict(
        vocab_size=96, hidden_size=32, num_attention_heads=4, num_key_value_heads=2, head_dim=16,
        intermediate_size=48, num_hidden_layers=2, max_position_embeddings=64, rms_norm_eps=1e-6,
        rope_theta=1e6, tie_word_embeddings=False)))
    parsed = model_config_from_hf(str(tmp_path))
    assert (parsed.num_layers, parsed.head_dim, parsed.num_kv_heads) == (2, 16, 2)
    shards = []
    for r in range(2):
        pg.set_tp_for_testing(r, 2)
        m = Qwen3ForCausalLM(cfg)
        assert load_weights_from_checkpoint(m, str(tmp_path)) == len(full)
        shards.append(m)
    pg.set_tp_for_testing(0, 1)
    q = torch.cat([s.model.layers[1].self_attn.qkv_proj.weight[:2 * 16] for s in shards])
    torch.testing.assert_close(q, full["model.layers.1.self_attn.q_proj.weight"])
    v = torch.cat([s.model.layers[0].self_attn.qkv_proj.weight[-16:] for s in shards])
    torch.testing.assert_close(v, full["model.layers.0.self_attn.v_proj.weight"])
    up = torch.cat([s.model.layers[0].mlp.gate_up_proj.weight[24:] for s in shards])
    torch.testing.assert_close(up, full["model.layers.0.mlp.up_proj.weight"])
    down = torch.cat([s.model.layers[0].mlp.down_proj.weight for s in shards], dim=1)
    torch.testing.assert_close(down, full["model.layers.0.mlp.down_proj.weight"])
    emb = torch.cat([s.model.embed_tokens.weight for s in shards])[:96]
    torch.testing.assert_close(emb, full["model.embed_tokens.weight"])
    torch.testing.assert_close(shards[1].model.layers[0].self_attn.k_norm.weight,
                               full["model.layers.0.self_attn.k_norm.weight"])
    with pytest.raises(KeyError):
        from minivllm_amd.utils.loader import load_state_dict_into
        load_state_dict_into(shards[0], {"model.layers.0.bogus.weight": torch.zeros(1)})


def _engine(**kw):
    from minivllm_amd.engine.llm_engine import LLMEngine
    model = ModelConfig(vocab_size=400, hidden_size=64, num_heads=4, num_kv_heads=2, head_dim=32,
                        intermediate_size=96, num_layers=2, max_position=512, base=10000.0)
    base = dict(model=model, max_num_sequences=4, max_num_batched_tokens=256, block_size=8,
                max_model_length=256, eos=-1, device="cpu", dtype=torch.float32, seed=5)
    base.update(kw)
    return LLMEngine(EngineConfig(**base))


def test_prefix_cache_hits_do_not_change_outputs():
    rng = np.random.default_rng(1)
    prefix = rng.integers(0, 400, size=40).tolist()           # 5 full blocks
    prompts = [prefix + rng.integers(0, 400, size=n).tolist() for n in (3, 11, 6)]
    sp = SamplingParams(temperature=0, max_tokens=9, ignore_eos=True)
    cold = [(_engine(), p) for p in prompts]
    want = []
    for e, p in cold:
        want.append(e.generate([p], sp)["token_ids"][0])
        e.exit()
    eng = _engine()
    got = [eng.generate([p], sp)["token_ids"][0] for p in prompts]     # later prompts hit
    assert eng.scheduler.block_manager.num_prefix_hits >= 10
    assert eng.stats.prefill_tokens < sum(map(len, prompts))          # cached tokens not recomputed
    eng.exit()
    assert got == want


def test_preemption_recompute_keeps_outputs_and_frees_everything():
    rng = np.random.default_rng(2)
    prompts = [rng.integers(0, 400, size=n).tolist() for n in (20, 25, 18, 30)]
    sp = SamplingParams(temperature=0, max_tokens=30, ignore_eos=True)
    roomy = _engine()
    want = roomy.generate(prompts, sp)["token_ids"]
    assert roomy.scheduler.num_preemptions == 0
    roomy.exit()
    tight = _engine(max_cached_blocks=16)                     # 128 tokens of KV for ~200 needed
    got = tight.generate(prompts, sp)["token_ids"]
    assert tight.scheduler.num_preemptions > 0
    bm = tight.scheduler.block_manager
    assert bm.num_free_blocks == bm.num_blocks and all(b.ref_count == 0 for b in bm.blocks)
    tight.exit()
    assert got == want


def test_chunked_prefill_matches_whole_prompt_prefill():
    """Prompts longer than a step's token budget are prefetched over several steps (each chunk
    attends to the earlier ones through the paged cache); tokens equal the one-step prefill."""
    rng = np.random.default_rng(8)
    shared = rng.integers(0, 400, size=24).tolist()
    prompts = [rng.integers(0, 400, size=n).tolist() for n in (70, 9, 45)] + \
              [shared + rng.integers(0, 400, size=n).tolist() for n in (30, 5)]
    sp = SamplingParams(temperature=0, max_tokens=10, ignore_eos=True)
    whole = _engine()
    want = whole.generate(prompts, sp)["token_ids"]
    steps_whole = whole.stats.prefill_steps
    whole.exit()
    chunked = _engine(max_num_batched_tokens=32)
    got = chunked.generate(prompts, sp)["token_ids"]
    assert chunked.stats.prefill_steps > steps_whole
    assert chunked.stats.prefill_tokens <= sum(map(len, prompts))    # no token computed twice
    chunked.exit()
    assert got == want
    with pytest.raises(ValueError):                                  # reference policy: rejected
        _engine(max_num_batched_tokens=32, enable_chunked_prefill=False).add_prompt(prompts[0], sp)


def test_mixed_prefill_decode_steps_match_separate_phases():
    """With mixed batches the running sequences take a decode token inside every prefill step
    (prompt rows through the prefill path, decode rows through the decode path of the same
    forward); greedy tokens must equal the phase-separated engine's and fewer steps are needed."""
    rng = np.random.default_rng(21)
    shared = rng.integers(0, 400, size=16).tolist()
    prompts = [rng.integers(0, 400, size=n).tolist() for n in (30, 7, 52, 12, 90, 5, 33)] + \
              [shared + rng.integers(0, 400, size=n).tolist() for n in (9, 20)]
    sps = [SamplingParams(temperature=0, max_tokens=m, ignore_eos=True)
           for m in (12, 30, 6, 18, 9, 25, 14, 8, 16)]
    plain = _engine(max_num_batched_tokens=64)
    want = plain.generate(prompts, sps)["token_ids"]
    steps_plain = plain.stats.prefill_steps + plain.stats.decode_steps
    plain.exit()
    mixed = _engine(max_num_batched_tokens=64, enable_mixed_batches=True)
    got = mixed.generate(prompts, sps)["token_ids"]
    sch = mixed.scheduler
    assert sch.num_mixed_steps > 0
    assert mixed.stats.prefill_steps + mixed.stats.decode_steps < steps_plain
    assert sch.block_manager.num_free_blocks == sch.block_manager.num_blocks
    mixed.exit()
    assert got == want
    # a cache too small for everyone: riders that cannot get a block sit out, preemption in
    # the decode phase still works, outputs unchanged
    tight = _engine(max_num_batched_tokens=64, enable_mixed_batches=True, max_cached_blocks=26)
    got = tight.generate(prompts, sps)["token_ids"]
    tight.exit()
    assert got == want


def test_engine_api_surface_and_stop_conditions():
    eng = _engine(eos=7)
    s = eng.add_prompt([1, 2, 3], SamplingParams(temperature=0, max_tokens=5))
    finished, n, is_prefill = eng.step()
    assert is_prefill and n == 3
    while not eng.is_finished():
        finished, n, is_prefill = eng.step()
    assert finished and finished[0][0] == s.seq_id and 1 <= len(finished[0][1]) <= 5
    assert eng.step() == ([], 0, False)
    out = eng.generate(["hello", "abc"], SamplingParams(temperature=0.7, max_tokens=4))
    assert len(out["text"]) == 2 and all(isinstance(t, str) for t in out["text"])
    with pytest.raises(ValueError):
        eng.add_prompt([999999])
    with pytest.raises(ValueError):
        eng.add_prompt(list(range(300)))
    eng.exit()
    eng.exit()                                                # idempotent


def test_compat_package_import_paths():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "src"))
    from myvllm.engine.llm_engine import LLMEngine  # noqa: F401
    from myvllm.models.qwen3 import Qwen3ForCausalLM  # noqa: F401
    from myvllm.sampling_parameters import SamplingParams as SP
    from myvllm.layers import (Attention, ColumnParallelLinear, LayerNorm,  # noqa: F401
                               MergedColumnParallelLinear, ParallelLMHead, QKVColumnParallelLinear,
                               ReplicatedLinear, RotaryEmbedding, RowParallelLinear, SamplerLayer,
                               SiluAndMul, VocabParallelEmbedding)
    from myvllm.layers.attention import (flash_attention_prefill, paged_attention_decode,  # noqa: F401
                                         store_kvcache)
    from myvllm.utils.context import get_context, reset_context, set_context  # noqa: F401
    from myvllm.engine.block_manager import BlockManager  # noqa: F401
    from myvllm.engine.scheduler import Scheduler  # noqa: F401
    from myvllm.engine.sequence import Sequence  # noqa: F401
    from myvllm.engine.model_runner import ModelRunner  # noqa: F401
    assert SP is SamplingParams


def test_top_k_and_top_p_restrict_the_draw():
    """top_k=1 and a tiny top_p both reduce sampling to arg-max whatever the temperature; with
    top_k=3 every sampled token is among the three most likely of its step, and the batch may
    mix restricted and unrestricted requests."""
    rng = np.random.default_rng(31)
    prompts = [rng.integers(0, 400, size=n).tolist() for n in (12, 30, 7, 19)]
    greedy = _engine()
    want = greedy.generate(prompts, SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))
    greedy.exit()
    for sp in (SamplingParams(temperature=1.5, top_k=1, max_tokens=8, ignore_eos=True),
               SamplingParams(temperature=1.5, top_p=1e-6, max_tokens=8, ignore_eos=True)):
        eng = _engine()
        assert eng.generate(prompts, sp)["token_ids"] == want["token_ids"]
        eng.exit()
    eng = _engine()
    r = eng.model_runner
    seen = []
    orig = r._resample_filtered

    def spy(b, tokens, logits):
        out = orig(b, tokens, logits)
        seen.append((b.arrays["top_k"].copy(), logits[:b.num_seqs].detach().clone(), out.clone()))
        return out
    r._resample_filtered = spy
    sps = [SamplingParams(temperature=2.0, top_k=3, max_tokens=12, ignore_eos=True),
           SamplingParams(temperature=2.0, max_tokens=12, ignore_eos=True),          # unrestricted
           SamplingParams(temperature=2.0, top_k=3, top_p=0.9, max_tokens=12, ignore_eos=True),
           SamplingParams(temperature=0, top_k=3, max_tokens=12, ignore_eos=True)]   # greedy: ignored
    out = eng.generate(prompts, sps)["token_ids"]
    eng.exit()
    assert seen and all(len(t) == 12 for t in out)
    picked_other_than_best = 0
    for top_k, logits, toks in seen:
        for i in np.nonzero(top_k > 0)[0]:
            top3 = logits[i].topk(3).indices.tolist()
            assert int(toks[i]) in top3
            picked_other_than_best += int(toks[i]) != top3[0]
    assert picked_other_than_best > 0                      # it does sample, not just arg-max
    with pytest.raises(ValueError):
        SamplingParams(top_p=0.0)
    with pytest.raises(ValueError):
        SamplingParams(top_k=-1)


def test_fp8_kv_cache_on_the_cpu_path():
    """kv_cache_dtype="fp8": the cache tensor holds e4m3 bytes, prefix reuse and chunked prefill
    read them back through the block table, and greedy outputs stay (nearly) those of the 32-bit
    cache on this small model."""
    rng = np.random.default_rng(41)
    shared = rng.integers(0, 400, size=24).tolist()
    prompts = [shared + rng.integers(0, 400, size=n).tolist() for n in (5, 30, 11)] + \
              [rng.integers(0, 400, size=70).tolist()]
    sp = SamplingParams(temperature=0, max_tokens=10, ignore_eos=True)
    ref_eng = _engine(max_num_batched_tokens=48)
    want = ref_eng.generate(prompts, sp)["token_ids"]
    bytes32 = ref_eng.model_runner.kv_block_bytes()
    ref_eng.exit()
    eng = _engine(max_num_batched_tokens=48, kv_cache_dtype="fp8")
    assert eng.model_runner.kv_cache.dtype == torch.float8_e4m3fn
    assert eng.model_runner.kv_block_bytes() * 4 == bytes32        # one byte per value, not four
    got = eng.generate(prompts, sp)["token_ids"]
    assert eng.scheduler.block_manager.num_prefix_hits > 0
    eng.exit()
    same = sum(int(a == b) for x, y in zip(want, got) for a, b in zip(x, y))
    assert same / 40 >= 0.7, (want, got)
    with pytest.raises(ValueError):
        EngineConfig.from_dict({"kv_cache_dtype": "int4", "device": "cpu"})
