"""Qwen3 decoder-only transformer (dense), tensor-parallel aware.

Structure follows the reference model (reference: src/myvllm/models/qwen3.py:11-331):
fused QKV projection -> per-head q/k RMSNorm -> RoPE -> paged attention -> row-parallel
output projection; SwiGLU MLP with a fused gate|up projection; pre-norm residual stream
with fused add+RMSNorm; vocab-parallel embedding and LM head with optional tying.

What differs, on purpose:
* module/parameter names match the public HF Qwen3 checkpoint layout (``model.layers.N.
  self_attn.{qkv_proj,o_proj,q_norm,k_norm}``, ``mlp.{gate_up_proj,down_proj}``, ...) and
  ``packed_modules_mapping`` routes ``q_proj/k_proj/v_proj/gate_proj/up_proj`` into the fused
  parameters (the reference mapping is an identity rename and its loader is dead code);
* q and k have *separate* learnable norm weights and ``rms_norm_epsilon`` is honoured;
* rotary positions are an explicit input computed once per step by the runner; the
  reference rebuilds them inside every layer with a device->host sync (qwen3.py:199-206);
* per-head norm, RoPE and the KV-cache store run as ONE kernel working in place on the QKV
  GEMM output, and attention reads q straight from that buffer (row stride = qkv width).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from minivllm_amd import ops
from minivllm_amd.config import ModelConfig
from minivllm_amd.layers.activation import SiluAndMul
from minivllm_amd.layers.attention import Attention
from minivllm_amd.layers.embedding_head import ParallelLMHead, VocabParallelEmbedding
from minivllm_amd.layers.layernorm import RMSNorm
from minivllm_amd.layers.linear import (MergedColumnParallelLinear, QKVColumnParallelLinear,
                                        RowParallelLinear)
from minivllm_amd.layers.rotary_embedding import RotaryEmbedding
from minivllm_amd.parallel import group as pg
from minivllm_amd.utils.context import get_context


class Qwen3Attention(nn.Module):
    def __init__(self, hidden_size: int, num_heads: int, num_kv_heads: int, head_dim: int,
                 scale: float, rms_norm_epsilon: float, qkv_bias: bool, rope: RotaryEmbedding,
                 qk_norm: Optional[bool] = None):
        super().__init__()
        tp = pg.tp_size()
        self.total_num_heads = num_heads
        self.total_num_kv_heads = num_kv_heads
        self.num_heads = num_heads // tp
        self.num_kv_heads = num_kv_heads // tp           # per-rank (reference overwrites it: qwen3.py:31,35)
        self.head_dim = head_dim
        self.q_size = self.num_heads * head_dim
        self.kv_size = self.num_kv_heads * head_dim
        self.eps = rms_norm_epsilon
        self.qkv_proj = QKVColumnParallelLinear(hidden_size, head_dim, num_heads, num_kv_heads,
                                                bias=qkv_bias)
        self.o_proj = RowParallelLinear(num_heads * head_dim, hidden_size, bias=False)
        # Qwen3 applies q/k norms; Qwen2-style checkpoints (qkv_bias=True) and Llama do not
        self.use_qk_norm = (not qkv_bias) if qk_norm is None else qk_norm
        if self.use_qk_norm:
            self.q_norm = RMSNorm(head_dim, rms_norm_epsilon)
            self.k_norm = RMSNorm(head_dim, rms_norm_epsilon)
        self._rope = [rope]                               # shared table, not a submodule here
        self.attn = Attention(self.num_heads, head_dim, scale, self.num_kv_heads)

    def forward(self, positions: torch.Tensor, hidden: torch.Tensor) -> torch.Tensor:
        return self.attend(positions, self.qkv_proj(hidden))

    def attend(self, positions: torch.Tensor, qkv: torch.Tensor,
               defer_reduce: bool = False, residual=None) -> torch.Tensor:
        """Everything after the QKV projection (the decoder layer may have produced ``qkv``
        together with the preceding norm in one kernel).  ``defer_reduce``: the caller feeds
        the result into a norm with a residual, which may take over the TP all-reduce
        (``RowParallelLinear.forward``); ``residual``: the stream it will be added to."""
        ctx = get_context()
        cos_sin = self._rope[0].cos_sin_cache
        q_weight = self.q_norm.weight if self.use_qk_norm else None
        if self.attn.fuses_decode_store(ctx, qkv):
            # decode steps on the GPU: the attention kernel norms and rotates q and the new k
            # and writes the new k / v into the cache itself
            hq, hkv, d = self.num_heads, self.num_kv_heads, self.head_dim
            T = qkv.shape[0]
            q = qkv[:, :hq * d].view(T, hq, d)
            k = qkv[:, hq * d:(hq + hkv) * d].view(T, hkv, d)
            v = qkv[:, (hq + hkv) * d:].view(T, hkv, d)
            o = self.attn(q, k, v, new_kv=(cos_sin, q_weight,
                                           self.k_norm.weight if self.use_qk_norm else None, self.eps))
            return self.o_proj(o, defer_reduce=defer_reduce, residual=residual)
        # pure prefill steps on the GPU: the store kernel handles K and V only and the
        # attention kernel norms and rotates its Q rows while loading them
        fuse = self.attn.fuses_q_rope(ctx, qkv)
        # ... and when it reads K through the block table, the transformed K rows are needed in
        # the cache only
        k_in_place = not (fuse and self.attn.reads_k_from_cache(ctx))
        q, k, v = ops.qk_norm_rope_store(
            qkv, positions, cos_sin, q_weight,
            self.k_norm.weight if self.use_qk_norm else None,
            self.eps, self.num_heads, self.num_kv_heads, self.head_dim,
            self.attn.k_cache, self.attn.v_cache, ctx.slot_mapping, skip_q=fuse,
            k_in_place=k_in_place)
        o = self.attn(q, k, v, q_rope=(positions, cos_sin, q_weight, self.eps) if fuse else None)
        return self.o_proj(o, defer_reduce=defer_reduce, residual=residual)


class Qwen3MLP(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int, bias: bool):
        super().__init__()
        self.gate_up_proj = MergedColumnParallelLinear(hidden_size, [intermediate_size] * 2, bias=bias)
        self.down_proj = RowParallelLinear(intermediate_size, hidden_size, bias=bias)
        self.act_fn = SiluAndMul()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.down_proj(self.gate_up_proj.forward_swiglu(x))


class Qwen3DecoderLayer(nn.Module):
    def __init__(self, cfg: ModelConfig, rope: RotaryEmbedding):
        super().__init__()
        self.self_attn = Qwen3Attention(cfg.hidden_size, cfg.num_heads, cfg.num_kv_heads,
                                        cfg.head_dim, cfg.scale, cfg.rms_norm_epsilon,
                                        cfg.qkv_bias, rope, cfg.use_qk_norm)
        self.mlp = Qwen3MLP(cfg.hidden_size, cfg.intermediate_size, cfg.ffn_bias)
        self.input_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_epsilon)
        self.post_attention_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_epsilon)

    def forward(self, positions, hidden, residual):
        # Each norm is handed to ops together with the projection that consumes its output:
        # for decode batches of a few rows the pair runs as one weight-streaming kernel,
        # otherwise as the (fused add +) norm kernel followed by the GEMM.
        n1, n2 = self.input_layernorm, self.post_attention_layernorm
        qkv_proj, gate_up = self.self_attn.qkv_proj, self.mlp.gate_up_proj
        qkv, residual = ops.norm_linear(hidden, residual, n1.weight, n1.eps, qkv_proj.weight,
                                        qkv_proj.bias, gamma_folded=n1.folded)
        # o and down hand their result to the norm behind them and to nothing else, so they may
        # leave work to it (ops.PendingReduce / ops.SplitKPartials instead of a tensor)
        hidden = self.self_attn.attend(positions, qkv, defer_reduce=True, residual=residual)
        act, residual = ops.norm_linear(hidden, residual, n2.weight, n2.eps, gate_up.weight,
                                        gate_up.bias, swiglu=True, gamma_folded=n2.folded)
        return self.mlp.down_proj(act, defer_reduce=True, residual=residual), residual

    def fold_norm_weights(self) -> None:
        self.input_layernorm.fold_into(self.self_attn.qkv_proj.weight)
        self.post_attention_layernorm.fold_into(self.mlp.gate_up_proj.weight)


class Qwen3Model(nn.Module):
    def __init__(self, cfg: ModelConfig):
        super().__init__()
        self.embed_tokens = VocabParallelEmbedding(cfg.vocab_size, cfg.hidden_size)
        self.rotary_emb = RotaryEmbedding(cfg.head_dim, cfg.head_dim, cfg.max_position, cfg.base,
                                          cfg.rope_scaling)
        self.layers = nn.ModuleList(Qwen3DecoderLayer(cfg, self.rotary_emb)
                                    for _ in range(cfg.num_layers))
        self.norm = RMSNorm(cfg.hidden_size, cfg.rms_norm_epsilon)

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        hidden = self.embed_tokens(input_ids)
        residual = None
        for layer in self.layers:
            hidden, residual = layer(positions, hidden, residual)
        hidden, _ = self.norm(hidden, residual)
        # a projection's pending work (K slices to sum, an all-reduce to run: ops.partials) is
        # taken over by the norm behind it; none may survive the last one
        assert torch.is_tensor(hidden), f"pending {type(hidden).__name__} left the decoder stack"
        return hidden


class Qwen3ForCausalLM(nn.Module):
    # checkpoint name fragment -> (fused parameter name fragment, shard id)
    packed_modules_mapping = {
        "q_proj": ("qkv_proj", "q"),
        "k_proj": ("qkv_proj", "k"),
        "v_proj": ("qkv_proj", "v"),
        "gate_proj": ("gate_up_proj", 0),
        "up_proj": ("gate_up_proj", 1),
    }
    packed_module_mapping = packed_modules_mapping        # the reference's spelling

    def __init__(self, config: Optional[ModelConfig] = None, **hparams):
        """Either ``Qwen3ForCausalLM(ModelConfig(...))`` or the reference's keyword form
        ``Qwen3ForCausalLM(vocab_size=..., hidden_size=..., num_heads=..., ...)``."""
        super().__init__()
        if config is None:
            if hparams.get("head_dim") is None:
                hparams["head_dim"] = hparams["hidden_size"] // hparams["num_heads"]
            if hparams.get("num_kv_heads") is None:
                hparams["num_kv_heads"] = hparams["num_heads"]
            config = ModelConfig(**hparams)
        self.config = config
        self.model = Qwen3Model(config)
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if config.tie_word_embeddings:
            self.lm_head.weight = self.model.embed_tokens.weight

    @property
    def qwen3_model(self):                                 # the reference's attribute name
        return self.model

    def forward(self, input_ids: torch.Tensor, positions: Optional[torch.Tensor] = None
                ) -> torch.Tensor:
        if positions is None:
            positions = default_positions(input_ids)
        return self.model(input_ids, positions)

    def compute_logits(self, hidden: torch.Tensor) -> Optional[torch.Tensor]:
        return self.lm_head(hidden)

    def compute_local_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        return self.lm_head.local_logits(hidden)

    @torch.no_grad()
    def fold_norm_weights(self) -> None:
        """Multiply the scale vectors of the two per-layer norms into the QKV and gate_up
        weights (call once, after the weights are in place).  The math is unchanged; what it
        buys is that (residual add + norm + projection) can run as one kernel for decode-sized
        batches.  The final norm stays as it is: its consumer, the LM head, may be tied to the
        embedding table."""
        for layer in self.model.layers:
            layer.fold_norm_weights()

    def init_random_weights(self, seed: int = 0, std: float = 0.02) -> None:
        """Deterministic random init that is identical for every TP degree: each full
        tensor is drawn from its own seeded CPU generator and sharded through the same
        ``weight_loader`` path a checkpoint would take."""
        from minivllm_amd.utils.loader import load_state_dict_into
        p = next(self.parameters())
        load_state_dict_into(self, RandomCheckpoint(self.config, seed, std, p.device.type
                                                    if p.device.type != "cuda" else p.device,
                                                    p.dtype if p.is_cuda else torch.float32))


def default_positions(input_ids: torch.Tensor) -> torch.Tensor:
    """Positions from the current context (or 0..T-1 for a bare call)."""
    ctx = get_context()
    if ctx.positions is not None:
        return ctx.positions
    return torch.arange(input_ids.shape[0], device=input_ids.device, dtype=torch.int32)


class RandomCheckpoint:
    """Lazily generated HF-layout state dict with seeded N(0, std) weights and unit norms."""

    def __init__(self, cfg: ModelConfig, seed: int = 0, std: float = 0.02,
                 device: str = "cpu", dtype: torch.dtype = torch.float32):
        self.cfg, self.seed, self.std = cfg, seed, std
        # full tensors are drawn where the model lives (a GPU draws Qwen3-8B in seconds);
        # the stream depends on the device type, not on the TP degree
        self.device, self.dtype = torch.device(device), dtype

    def _shapes(self):
        c = self.cfg
        yield "model.embed_tokens.weight", (c.vocab_size, c.hidden_size)
        for i in range(c.num_layers):
            p = f"model.layers.{i}."
            yield p + "self_attn.q_proj.weight", (c.num_heads * c.head_dim, c.hidden_size)
            yield p + "self_attn.k_proj.weight", (c.num_kv_heads * c.head_dim, c.hidden_size)
            yield p + "self_attn.v_proj.weight", (c.num_kv_heads * c.head_dim, c.hidden_size)
            yield p + "self_attn.o_proj.weight", (c.hidden_size, c.num_heads * c.head_dim)
            if c.qkv_bias:
                yield p + "self_attn.q_proj.bias", (c.num_heads * c.head_dim,)
                yield p + "self_attn.k_proj.bias", (c.num_kv_heads * c.head_dim,)
                yield p + "self_attn.v_proj.bias", (c.num_kv_heads * c.head_dim,)
            if c.use_qk_norm:
                yield p + "self_attn.q_norm.weight", (c.head_dim,)
                yield p + "self_attn.k_norm.weight", (c.head_dim,)
            yield p + "mlp.gate_proj.weight", (c.intermediate_size, c.hidden_size)
            yield p + "mlp.up_proj.weight", (c.intermediate_size, c.hidden_size)
            yield p + "mlp.down_proj.weight", (c.hidden_size, c.intermediate_size)
            if c.ffn_bias:
                yield p + "mlp.gate_proj.bias", (c.intermediate_size,)
                yield p + "mlp.up_proj.bias", (c.intermediate_size,)
                yield p + "mlp.down_proj.bias", (c.hidden_size,)
            yield p + "input_layernorm.weight", (c.hidden_size,)
            yield p + "post_attention_layernorm.weight", (c.hidden_size,)
        yield "model.norm.weight", (c.hidden_size,)
        if not c.tie_word_embeddings:
            yield "lm_head.weight", (c.vocab_size, c.hidden_size)

    def keys(self):
        return [k for k, _ in self._shapes()]

    def items(self):
        for idx, (name, shape) in enumerate(self._shapes()):
            yield name, self._make(idx, name, shape)

    def _make(self, idx: int, name: str, shape) -> torch.Tensor:
        g = torch.Generator(device=self.device)
        g.manual_seed(self.seed * 1_000_003 + idx)
        kw = dict(generator=g, device=self.device, dtype=self.dtype)
        if name.endswith("norm.weight"):
            # around 1 so that norms are exercised but activations stay well scaled
            return 1.0 + 0.1 * torch.randn(shape, **kw)
        if name.endswith(".bias"):
            return 0.01 * torch.randn(shape, **kw)
        return torch.randn(shape, **kw) * self.std
