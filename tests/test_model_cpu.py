"""CPU eager path: layers and the whole Qwen3 model against independent oracles
(HF transformers' Qwen3 with the same weights; closed-form layer formulas)."""

import numpy as np
import pytest
import torch

from minivllm_amd.config import EngineConfig, ModelConfig
from minivllm_amd.models.qwen3 import Qwen3ForCausalLM, RandomCheckpoint
from minivllm_amd.ops import torch_ref as ref
from minivllm_amd.sampling_parameters import SamplingParams
from minivllm_amd.utils.context import reset_context, set_context
from minivllm_amd.utils.loader import load_state_dict_into


def tiny_cfg(**kw):
    base = dict(vocab_size=320, hidden_size=64, num_heads=4, num_kv_heads=2, head_dim=32,
                intermediate_size=96, num_layers=2, max_position=256, base=10000.0,
                tie_word_embeddings=False, rms_norm_epsilon=1e-6)
    base.update(kw)
    return ModelConfig(**base)


def hf_model(cfg: ModelConfig, state):
    tr = pytest.importorskip("transformers")
    hf_cfg = tr.Qwen3Config(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
        intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_layers,
        num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_kv_heads,
        head_dim=cfg.head_dim, max_position_embeddings=cfg.max_position,
        rms_norm_eps=cfg.rms_norm_epsilon, rope_theta=cfg.base,
        tie_word_embeddings=cfg.tie_word_embeddings, attention_bias=False,
        use_sliding_window=False, attn_implementation="eager")
    m = tr.Qwen3ForCausalLM(hf_cfg).eval().float()
    sd = dict(state.items())
    if cfg.tie_word_embeddings:
        sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not [k for k in missing if "rotary" not in k], missing
    assert not unexpected, unexpected
    return m


@pytest.mark.parametrize("tied", [False, True])
def test_prefill_logits_match_hf_qwen3(tied):
    cfg = tiny_cfg(tie_word_embeddings=tied)
    ckpt = RandomCheckpoint(cfg, seed=1, std=0.05)
    ours = Qwen3ForCausalLM(cfg).eval()
    assert load_state_dict_into(ours, ckpt) == len(ckpt.keys())
    hf = hf_model(cfg, ckpt)
    lens = [9, 17]
    ids = torch.randint(0, cfg.vocab_size, (sum(lens),))
    cu = torch.tensor([0, 9, 26], dtype=torch.int32)
    pos = torch.cat([torch.arange(n) for n in lens]).int()
    set_context(True, cu_seqlens_q=cu, cu_seqlens_k=cu, max_seqlen_q=17, max_seqlen_k=17,
                positions=pos, num_seqs=2)
    with torch.no_grad():
        hidden = ours(ids.int(), pos)
        mine = ours.lm_head.local_logits(hidden)          # last token of each sequence
    reset_context()
    with torch.no_grad():
        want = torch.stack([hf(ids[s:e][None]).logits[0, -1] for s, e in ((0, 9), (9, 26))])
    torch.testing.assert_close(mine, want, atol=2e-4, rtol=2e-4)


def test_engine_greedy_generation_matches_hf_generate():
    """Plumbing configuration: tiny Qwen3, CPU, greedy, through LLMEngine.generate with a
    small KV block so that decode crosses block boundaries."""
    from minivllm_amd.engine.llm_engine import LLMEngine
    cfg = tiny_cfg()
    eng = LLMEngine(EngineConfig(model=cfg, max_num_sequences=4, max_num_batched_tokens=128,
                                 block_size=8, max_model_length=128, eos=-1, device="cpu",
                                 dtype=torch.float32, seed=4))
    state = {k: v.clone() for k, v in eng.model_runner.model.state_dict().items()}
    hf_sd = {}
    for k, v in state.items():           # un-fuse our parameters back into HF names
        if "qkv_proj" in k:
            q, kk, vv = v.split([cfg.num_heads * cfg.head_dim, cfg.num_kv_heads * cfg.head_dim,
                                 cfg.num_kv_heads * cfg.head_dim])
            for n, t in (("q_proj", q), ("k_proj", kk), ("v_proj", vv)):
                hf_sd[k.replace("qkv_proj", n)] = t
        elif "gate_up_proj" in k:
            g, u = v.chunk(2)
            hf_sd[k.replace("gate_up_proj", "gate_proj")] = g
            hf_sd[k.replace("gate_up_proj", "up_proj")] = u
        else:
            hf_sd[k] = v
    hf = hf_model(cfg, hf_sd)
    rng = np.random.default_rng(0)
    prompts = [rng.integers(0, cfg.vocab_size, size=n).tolist() for n in (5, 13, 8)]
    out = eng.generate(prompts, SamplingParams(temperature=0, max_tokens=12, ignore_eos=True))
    eng.exit()
    for p, got in zip(prompts, out["token_ids"]):
        with torch.no_grad():
            ids = torch.tensor([p])          # explicit mask: token id 0 is a real token here
            want = hf.generate(ids, attention_mask=torch.ones_like(ids), max_new_tokens=12,
                               do_sample=False, pad_token_id=0)[0, len(p):].tolist()
        assert got == want


@pytest.mark.parametrize("family", ["llama", "qwen2"])
def test_other_decoder_families_match_hf(family, tmp_path):
    """The same decoder serves Llama 3 (no QKV bias, no q/k norm, llama3 RoPE scaling) and Qwen2.5
    (QKV bias, no q/k norm): logits against HF transformers with the same weights, and the HF
    config.json round trip."""
    import json
    tr = pytest.importorskip("transformers")
    from minivllm_amd.utils.loader import model_config_from_hf
    scaling = dict(rope_type="llama3", factor=4.0, low_freq_factor=1.0, high_freq_factor=4.0,
                   original_max_position_embeddings=64)
    common = dict(vocab_size=320, hidden_size=64, intermediate_size=96, num_hidden_layers=2,
                  num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=256,
                  rms_norm_eps=1e-5, rope_theta=10000.0, tie_word_embeddings=False,
                  attn_implementation="eager")
    if family == "llama":
        hf_cfg = tr.LlamaConfig(head_dim=16, rope_scaling=scaling, attention_bias=False, **common)
        hf = tr.LlamaForCausalLM(hf_cfg)
    else:
        hf_cfg = tr.Qwen2Config(use_sliding_window=False, **common)
        hf = tr.Qwen2ForCausalLM(hf_cfg)
    hf = hf.eval().float()
    (tmp_path / "config.json").write_text(json.dumps(hf_cfg.to_dict()))
    cfg = model_config_from_hf(str(tmp_path))
    assert not cfg.use_qk_norm and cfg.qkv_bias == (family == "qwen2") and cfg.head_dim == 16
    assert (cfg.rope_scaling is not None) == (family == "llama")
    torch.manual_seed(0)
    with torch.no_grad():
        for n_, p_ in hf.named_parameters():
            if "norm" in n_:
                p_.uniform_(0.5, 1.5)
            elif p_.dim() == 1:                          # biases
                p_.normal_(0.0, 0.05)
            else:
                p_.normal_(0.0, 0.08)
    ours = Qwen3ForCausalLM(cfg).eval()
    state = {k: v for k, v in hf.state_dict().items() if "rotary" not in k}
    assert load_state_dict_into(ours, state) == len(state)
    lens = [11, 150]                                     # the long one crosses the scaling band
    ids = torch.randint(0, cfg.vocab_size, (sum(lens),))
    cu = torch.tensor([0, 11, 161], dtype=torch.int32)
    pos = torch.cat([torch.arange(n) for n in lens]).int()
    set_context(True, cu_seqlens_q=cu, cu_seqlens_k=cu, max_seqlen_q=150, max_seqlen_k=150,
                positions=pos, num_seqs=2)
    with torch.no_grad():
        mine = ours.lm_head.local_logits(ours(ids.int(), pos))
    reset_context()
    with torch.no_grad():
        want = torch.stack([hf(ids[s:e][None]).logits[0, -1] for s, e in ((0, 11), (11, 161))])
    torch.testing.assert_close(mine, want, atol=3e-4, rtol=3e-4)


# ------------------------------------------------------------------ layers
def test_rms_norm_and_fused_add_formula():
    x, r, w = torch.randn(5, 48), torch.randn(5, 48), torch.rand(48) + 0.5
    want = x / torch.sqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * w
    torch.testing.assert_close(ref.rms_norm(x, w, 1e-5), want)
    o, s = ref.add_rms_norm(x, r, w, 1e-5)
    torch.testing.assert_close(s, x + r)
    torch.testing.assert_close(o, ref.rms_norm(x + r, w, 1e-5))


def test_rope_is_a_rotation_and_relative():
    d, T = 32, 11
    cs = ref.build_cos_sin_cache(d, 64, 10000.0)
    q, k = torch.randn(T, 2, d), torch.randn(T, 2, d)
    pos = torch.arange(T)
    c, s = cs[pos].chunk(2, -1)
    qr, kr = ref.apply_rope(q, c, s), ref.apply_rope(k, c, s)
    torch.testing.assert_close(qr.norm(dim=-1), q.norm(dim=-1))
    # scores depend on the position difference only
    c2, s2 = cs[pos + 7].chunk(2, -1)
    qr2, kr2 = ref.apply_rope(q, c2, s2), ref.apply_rope(k, c2, s2)
    torch.testing.assert_close((qr[5] * kr[2]).sum(-1), (qr2[5] * kr2[2]).sum(-1), atol=1e-4, rtol=1e-4)


def test_silu_and_mul_and_sampler():
    x = torch.randn(3, 10)
    torch.testing.assert_close(ref.silu_and_mul(x), torch.nn.functional.silu(x[:, :5]) * x[:, 5:])
    logits = torch.tensor([[0.0, 5.0, 1.0], [3.0, 0.0, 0.0]])
    assert ref.sample(logits, torch.zeros(2)).tolist() == [1, 0]
    g = torch.Generator().manual_seed(0)
    base = torch.tensor([1.0, 0.0, -1.0, 2.0])
    draws = ref.sample(base.repeat(8000, 1), torch.full((8000,), 0.9), g)
    freq = torch.bincount(draws, minlength=4).float() / 8000
    assert (freq - torch.softmax(base / 0.9, -1)).abs().max() < 0.02


def test_paged_attention_oracles_agree_with_dense():
    """The paged prefill/decode oracles equal dense causal attention over the same tokens."""
    torch.manual_seed(0)
    hq, hkv, d, bs, L = 4, 2, 16, 4, 13
    q, k, v = torch.randn(L, hq, d), torch.randn(L, hkv, d), torch.randn(L, hkv, d)
    cu = torch.tensor([0, L], dtype=torch.int32)
    dense = ref.flash_attention_prefill(q, k, v, cu, 0.25)
    kc, vc = torch.zeros(6, bs, hkv, d), torch.zeros(6, bs, hkv, d)
    table = torch.tensor([[5, 2, 0, 3]], dtype=torch.int32)
    slots = torch.tensor([table[0, i // bs] * bs + i % bs for i in range(L)])
    ref.store_kvcache(k, v, kc, vc, slots)
    paged = ref.paged_attention_prefill(q, kc, vc, cu, cu, table, 0.25)
    torch.testing.assert_close(paged, dense)
    tail = ref.paged_attention_prefill(q[9:], kc, vc, torch.tensor([0, 4], dtype=torch.int32), cu, table, 0.25)
    torch.testing.assert_close(tail, dense[9:])
    dec = ref.paged_attention_decode(q[-1:], kc, vc, table, torch.tensor([L]), 0.25)
    torch.testing.assert_close(dec[0], dense[-1])
