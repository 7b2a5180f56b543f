"""Plain-PyTorch implementations of every device op.

They serve two purposes: the CPU eager path of the engine (no GPU needed) and
the numerical oracle the HIP kernels are tested against.  Formulas follow the
reference's eager code: RMSNorm (reference: src/myvllm/layers/layernorm.py:13-22),
NeoX-style RoPE (rotary_embedding.py:4-45), SiluAndMul (activation.py:17-18),
exponential-race sampling (sampler.py:16-18), and the O(N^2) attention oracles of
benchmark_prefilling.py:9-52 and benchmark_decoding.py:136-189.  All math is done
in fp32 and cast back to the input dtype.

KV caches are addressed logically as ``[num_blocks, block_size, num_kv_heads,
head_dim]`` whatever their physical stride order is.
"""
from __future__ import annotations

from typing import Optional, Tuple

import math

import torch


# ----------------------------------------------------------------- norms
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    xf = x.float()
    var = xf.pow(2).mean(dim=-1, keepdim=True)
    out = xf * torch.rsqrt(var + eps) * weight.float()
    return out.to(x.dtype)


def add_rms_norm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                 eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (norm(x + residual), x + residual).  The sum is rounded to the
    activation dtype before normalising, as the fused kernel does."""
    s = (x.float() + residual.float()).to(x.dtype)
    return rms_norm(s, weight, eps), s


# ------------------------------------------------------------ activation
def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    a, b = x.float().chunk(2, dim=-1)
    return (torch.nn.functional.silu(a) * b).to(x.dtype)


# ------------------------------------------------------------------ rope
def scaled_inv_freq(inv_freq: torch.Tensor, rope_scaling) -> torch.Tensor:
    """Frequency rescaling for context extension.  ``linear``: every frequency divided by
    ``factor``.  ``llama3`` (Llama 3.1+): wavelengths longer than the original context divided
    by ``factor``, short ones kept, a smooth blend in between."""
    if not rope_scaling:
        return inv_freq
    kind = rope_scaling.get("rope_type", rope_scaling.get("type", "default"))
    if kind in ("default", None):
        return inv_freq
    factor = float(rope_scaling["factor"])
    if kind == "linear":
        return inv_freq / factor
    if kind == "llama3":
        low = float(rope_scaling.get("low_freq_factor", 1.0))
        high = float(rope_scaling.get("high_freq_factor", 4.0))
        old_ctx = float(rope_scaling.get("original_max_position_embeddings", 8192))
        wavelen = 2 * math.pi / inv_freq
        smooth = ((old_ctx / wavelen - low) / (high - low)).clamp(0.0, 1.0)
        blended = (1 - smooth) * inv_freq / factor + smooth * inv_freq
        out = torch.where(wavelen > old_ctx / low, inv_freq / factor, blended)
        return torch.where(wavelen < old_ctx / high, inv_freq, out)
    raise ValueError(f"unsupported rope_scaling type {kind!r} (have: linear, llama3)")


def build_cos_sin_cache(head_dim: int, max_position: int, base: float,
                        device=None, rope_scaling=None) -> torch.Tensor:
    """[max_position, head_dim] fp32 table: cos in the first half, sin in the second."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float64) / head_dim))
    inv_freq = scaled_inv_freq(inv_freq, rope_scaling)
    t = torch.arange(max_position, dtype=torch.float64)
    freqs = torch.outer(t, inv_freq)
    table = torch.cat([freqs.cos(), freqs.sin()], dim=-1).float()
    return table.to(device) if device is not None else table


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x: [T, H, D] fp32; cos/sin: [T, D/2].  Half-split (NeoX) rotation."""
    x1, x2 = x.chunk(2, dim=-1)
    c, s = cos[:, None, :], sin[:, None, :]
    return torch.cat([x1 * c - x2 * s, x1 * s + x2 * c], dim=-1)


def store_kvcache(key: torch.Tensor, value: torch.Tensor, k_cache: torch.Tensor,
                  v_cache: torch.Tensor, slot_mapping: torch.Tensor) -> None:
    """key/value [T, Hkv, D] -> cache[slot // bs, slot % bs]; slot -1 is skipped."""
    bs = k_cache.shape[1]
    slots = slot_mapping.long()
    keep = slots >= 0
    if not bool(keep.all()):
        slots, key, value = slots[keep], key[keep], value[keep]
    blk = torch.div(slots, bs, rounding_mode="floor")
    off = slots - blk * bs
    if k_cache.dtype == torch.float8_e4m3fn:      # indexed writes go through the byte view;
        # values saturate at the format's largest finite number (448), as in the HIP kernel
        k_cache.view(torch.uint8)[blk, off] = key.float().clamp(-448.0, 448.0).to(k_cache.dtype).view(torch.uint8)
        v_cache.view(torch.uint8)[blk, off] = value.float().clamp(-448.0, 448.0).to(v_cache.dtype).view(torch.uint8)
        return
    k_cache[blk, off] = key.to(k_cache.dtype)
    v_cache[blk, off] = value.to(v_cache.dtype)


def qk_norm_rope_store(qkv: torch.Tensor, positions: torch.Tensor, cos_sin: torch.Tensor,
                       q_weight: Optional[torch.Tensor], k_weight: Optional[torch.Tensor],
                       eps: float, num_heads: int, num_kv_heads: int, head_dim: int,
                       k_cache: Optional[torch.Tensor], v_cache: Optional[torch.Tensor],
                       slot_mapping: Optional[torch.Tensor], skip_q: bool = False):
    """Fused per-head RMSNorm(q,k) + RoPE(q,k) + KV-cache store.  ``skip_q``: q is returned as
    projected (``apply_q_rope`` is then the consumer's job).

    qkv: [T, (Hq + 2 Hkv) * D].  Returns (q [T,Hq,D], k [T,Hkv,D], v [T,Hkv,D]) in
    the input dtype; k/v are also scattered to the cache when one is given.
    """
    T = qkv.shape[0]
    qs, ks = num_heads * head_dim, num_kv_heads * head_dim
    q, k, v = qkv.split([qs, ks, ks], dim=-1)
    q_raw = q.reshape(T, num_heads, head_dim)
    q = q_raw.float()
    k = k.reshape(T, num_kv_heads, head_dim).float()
    v = v.reshape(T, num_kv_heads, head_dim)
    if q_weight is not None:
        q = q * torch.rsqrt(q.pow(2).mean(-1, keepdim=True) + eps) * q_weight.float()
        # round like the kernel does not: the kernel keeps fp32 between norm and rope
    if k_weight is not None:
        k = k * torch.rsqrt(k.pow(2).mean(-1, keepdim=True) + eps) * k_weight.float()
    cs = cos_sin[positions.long()].float()
    cos, sin = cs.chunk(2, dim=-1)
    q = apply_rope(q, cos, sin).to(qkv.dtype)
    k = apply_rope(k, cos, sin).to(qkv.dtype)
    if k_cache is not None and slot_mapping is not None:
        store_kvcache(k, v, k_cache, v_cache, slot_mapping)
    return (q_raw if skip_q else q), k, v.contiguous()


def apply_q_rope(q: torch.Tensor, q_rope) -> torch.Tensor:
    """The transform a prefill attention kernel applies to raw Q rows when it is handed
    ``q_rope`` = (positions, cos_sin, q_norm weight or None, eps): per-head RMSNorm, then RoPE."""
    if q_rope is None:
        return q
    positions, cos_sin, q_weight, eps = q_rope
    x = q.float()
    if q_weight is not None:
        x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * q_weight.float()
    cs = cos_sin[positions[:q.shape[0]].long()].float()
    cos, sin = cs.chunk(2, dim=-1)
    return apply_rope(x, cos, sin).to(q.dtype)


# ------------------------------------------------------------- attention
def _attend(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float,
            causal_offset: Optional[int]) -> torch.Tensor:
    """q [Lq,Hq,D], k/v [Lk,Hkv,D] -> [Lq,Hq,D] (fp32).  ``causal_offset`` = Lk - Lq
    for causal masking (query i sees keys <= i + offset); None = no mask."""
    Lq, Hq, D = q.shape
    Lk, Hkv, _ = k.shape
    g = Hq // Hkv
    qf = q.float().transpose(0, 1)                               # [Hq, Lq, D]
    kf = k.float().transpose(0, 1).repeat_interleave(g, dim=0)   # [Hq, Lk, D]
    vf = v.float().transpose(0, 1).repeat_interleave(g, dim=0)
    s = torch.matmul(qf, kf.transpose(1, 2)) * scale
    if causal_offset is not None:
        qi = torch.arange(Lq, device=q.device)[:, None] + causal_offset
        kj = torch.arange(Lk, device=q.device)[None, :]
        s = s.masked_fill((kj > qi)[None], float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, vf).transpose(0, 1)


def flash_attention_prefill(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                            cu_seqlens: torch.Tensor, scale: float) -> torch.Tensor:
    """Causal varlen self-attention over packed sequences (q_len == k_len)."""
    out = torch.empty_like(q)
    cu = cu_seqlens.tolist()
    for b in range(len(cu) - 1):
        s, e = cu[b], cu[b + 1]
        if e > s:
            out[s:e] = _attend(q[s:e], k[s:e], v[s:e], scale, 0).to(q.dtype)
    return out


def _gather_paged(cache: torch.Tensor, block_table: torch.Tensor, length: int) -> torch.Tensor:
    bs = cache.shape[1]
    nb = (length + bs - 1) // bs
    blocks = block_table[:nb].long()
    if cache.dtype == torch.float8_e4m3fn:
        got = cache.view(torch.uint8)[blocks].view(torch.float8_e4m3fn).float()
    else:
        got = cache[blocks]
    return got.reshape(nb * bs, cache.shape[2], cache.shape[3])[:length]


def paged_attention_prefill(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                            cu_seqlens_q: torch.Tensor, cu_seqlens_k: torch.Tensor,
                            block_tables: torch.Tensor, scale: float) -> torch.Tensor:
    """Causal attention of packed new tokens against the paged cache (which already
    holds prefix AND new tokens): q_len <= k_len per sequence."""
    out = torch.empty_like(q)
    cq, ck = cu_seqlens_q.tolist(), cu_seqlens_k.tolist()
    for b in range(len(cq) - 1):
        s, e = cq[b], cq[b + 1]
        lk = ck[b + 1] - ck[b]
        if e == s:
            continue
        k = _gather_paged(k_cache, block_tables[b], lk)
        v = _gather_paged(v_cache, block_tables[b], lk)
        out[s:e] = _attend(q[s:e], k, v, scale, lk - (e - s)).to(q.dtype)
    return out


def paged_attention_decode(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                           block_tables: torch.Tensor, context_lens: torch.Tensor,
                           scale: float) -> torch.Tensor:
    """q [B,Hq,D]; one query token per sequence over its paged KV."""
    out = torch.zeros_like(q)
    lens = context_lens.tolist()
    for b, L in enumerate(lens):
        if L <= 0:
            continue
        k = _gather_paged(k_cache, block_tables[b], L)
        v = _gather_paged(v_cache, block_tables[b], L)
        out[b:b + 1] = _attend(q[b:b + 1], k, v, scale, None).to(q.dtype)
    return out


# --------------------------------------------------------------- sampler
def sample(logits: torch.Tensor, temperatures: torch.Tensor,
           generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Rows with temperature 0 take the arg-max; the others draw from
    softmax(logits / T) through the exponential race argmax(p / E), E ~ Exp(1)."""
    lf = logits.float()
    greedy = lf.argmax(dim=-1)
    t = temperatures.float()
    if bool((t <= 0).all()):
        return greedy
    safe_t = torch.where(t > 0, t, torch.ones_like(t))
    probs = torch.softmax(lf / safe_t[:, None], dim=-1)
    noise = torch.empty_like(probs).exponential_(1.0, generator=generator).clamp_min_(1e-10)
    drawn = (probs / noise).argmax(dim=-1)
    return torch.where(t > 0, drawn, greedy)
