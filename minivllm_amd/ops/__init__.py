"""Device-op entry points.  Exactly two implementations exist: ``hip`` (hand-written
gfx950 kernels, used for every CUDA/HIP tensor) and ``torch_ref`` (plain PyTorch,
used on CPU and as the test oracle).  A GPU tensor never falls back to
``torch_ref``: if the compiled kernel library is missing the call raises.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import torch_ref
from .partials import PendingReduce, ResidualShard, ResidualSum, ScatteredRows, SplitKPartials


def _hip():
    from . import hip        # raises with a build hint when the .so is absent
    return hip


def _on_gpu(t: torch.Tensor) -> bool:
    return t.is_cuda


def rms_norm(x, weight, eps):
    if isinstance(x, ResidualSum):
        x = x.r
    if isinstance(x, PendingReduce):
        return _reduce_add_rms_norm(x, None, weight, eps)[0]
    if _is_partials(x):
        x = x.materialize()
    return _hip().rms_norm(x, weight, eps) if _on_gpu(x) else torch_ref.rms_norm(x, weight, eps)


def _is_partials(x) -> bool:
    return isinstance(x, SplitKPartials)


def _reduce_add_rms_norm(x: PendingReduce, residual, weight, eps):
    from ..parallel import group as pg
    return pg.tp_reduce_add_rms_norm(x, residual, weight, eps)


def add_rms_norm(x, residual, weight, eps):
    """``r = x + residual``; returns ``(rms_norm(r) * weight, r)``.  ``x`` may be a projection's
    pending work (``PendingReduce``: all-reduce still to do; ``SplitKPartials``: K slices still
    to sum), which the norm then takes over; ``r`` may come back as a ``ResidualShard`` (and be
    passed in as one) when the reduce ran as the two-shot peer-to-peer kernel."""
    if isinstance(x, ResidualSum):          # the residual is already in: norm the stream
        return rms_norm(x.r, weight, eps), x.r
    if isinstance(x, PendingReduce):
        return _reduce_add_rms_norm(x, residual, weight, eps)
    if isinstance(residual, ResidualShard):
        raise TypeError("a row-sharded residual stream can only be consumed together with a "
                        "pending row-parallel reduce")
    if _is_partials(x):
        return x.add_rms_norm(residual, weight, eps)
    if _on_gpu(x):
        return _hip().add_rms_norm(x, residual, weight, eps)
    return torch_ref.add_rms_norm(x, residual, weight, eps)


def silu_and_mul(x):
    return _hip().silu_and_mul(x) if _on_gpu(x) else torch_ref.silu_and_mul(x)


def qk_norm_rope_store(qkv, positions, cos_sin, q_weight, k_weight, eps, num_heads, num_kv_heads,
                       head_dim, k_cache, v_cache, slot_mapping, skip_q=False, k_in_place=True):
    """``skip_q``: leave the Q rows as projected; the prefill attention call that follows is
    then given ``q_rope`` and transforms them while loading (one pass over Q less).
    ``k_in_place=False``: the caller reads K from the cache only, so the GPU kernel does not
    write the transformed K rows back into ``qkv`` (the CPU path always does)."""
    if _on_gpu(qkv):
        return _hip().qk_norm_rope_store(qkv, positions, cos_sin, q_weight, k_weight, eps, num_heads,
                                         num_kv_heads, head_dim, k_cache, v_cache, slot_mapping,
                                         skip_q, k_in_place)
    return torch_ref.qk_norm_rope_store(qkv, positions, cos_sin, q_weight, k_weight, eps, num_heads,
                                        num_kv_heads, head_dim, k_cache, v_cache, slot_mapping, skip_q)


def store_kvcache(key, value, k_cache, v_cache, slot_mapping):
    f = _hip().store_kvcache if _on_gpu(key) else torch_ref.store_kvcache
    return f(key, value, k_cache, v_cache, slot_mapping)


def flash_attention_prefill(q, k, v, cu_seqlens, scale, max_seqlen=None, q_rope=None):
    """``q_rope`` = (positions, cos_sin, q_norm weight or None, eps): q holds raw projections and
    the kernel applies q-norm + RoPE to them (see ``qk_norm_rope_store(skip_q=True)``)."""
    if _on_gpu(q):
        return _hip().flash_attention_prefill(q, k, v, cu_seqlens, scale, max_seqlen, q_rope=q_rope)
    return torch_ref.flash_attention_prefill(torch_ref.apply_q_rope(q, q_rope), k, v, cu_seqlens, scale)


def _into(out, res):
    if out is None:
        return res
    out.copy_(res)
    return out


def paged_attention_prefill(q, k_cache, v_cache, cu_seqlens_q, cu_seqlens_k, block_tables, scale,
                            max_seqlen_q=None, out=None, max_seqlen_k=None, total_k=None,
                            q_rope=None, work_items=None, positions_in_context=False):
    """``out``: optional [T, Hq, D] destination (a view into a larger buffer is fine).
    ``positions_in_context``: the caller vouches that ``q_rope``'s positions are each query's
    place in its context (k_len - q_len + i, what a prefill step feeds); the GPU kernel then
    derives them instead of loading them (one dependent load less before its first tile).
    ``max_seqlen_k`` / ``total_k`` (longest context, sum of contexts) spare an fp8 cache's
    widening pass a device synchronisation and an oversized scratch buffer.  ``work_items``
    (GPU): the step's work list, which makes the launch shape independent of the lengths
    (``hip.prefill_work_items``)."""
    if _on_gpu(q):
        if positions_in_context and q_rope is not None:
            q_rope = (None,) + tuple(q_rope[1:])
        return _hip().paged_attention_prefill(q, k_cache, v_cache, cu_seqlens_q, cu_seqlens_k,
                                              block_tables, scale, max_seqlen_q, out=out,
                                              max_seqlen_k=max_seqlen_k, total_k=total_k,
                                              q_rope=q_rope, work_items=work_items)
    return _into(out, torch_ref.paged_attention_prefill(torch_ref.apply_q_rope(q, q_rope), k_cache,
                                                        v_cache, cu_seqlens_q, cu_seqlens_k,
                                                        block_tables, scale))


def paged_attention_decode(q, k_cache, v_cache, block_tables, context_lens, scale,
                           max_context_len=None, out=None, new_kv=None):
    """``new_kv`` (GPU, 16-bit caches; see ``hip.paged_attention_decode``): q and the new
    tokens' k / v are raw projections; the kernel norms, rotates and stores them itself."""
    if _on_gpu(q):
        return _hip().paged_attention_decode(q, k_cache, v_cache, block_tables, context_lens,
                                             scale, max_context_len, out=out, new_kv=new_kv)
    if new_kv is not None:
        raise ValueError("new_kv is a form of the GPU kernel")
    return _into(out, torch_ref.paged_attention_decode(q, k_cache, v_cache, block_tables,
                                                       context_lens, scale))


def linear(x, weight, bias=None, to_norm=False, residual=None):
    """``x @ weight.T (+ bias)``: the library GEMM, except for the GPU inputs (<= 16 rows: skinny
    kernel; 17..256 rows: mid-batch kernel; more: the 256 x 256-tile prefill kernel) of the
    shapes where the hand-written kernels were measured to be faster on this machine.  ``to_norm``: the
    caller passes the result straight to ``norm_linear`` / ``add_rms_norm`` (and to nothing
    else), which lets a K-split mid-batch GEMM hand over its unsummed slices
    (``SplitKPartials``) for the norm kernel to sum.  ``residual`` (with ``to_norm``): the
    stream that norm will add the result to; a prefill-sized GPU projection may then add it in
    its own epilogue and return ``ResidualSum`` (the stream, updated in place, not yet
    normalised), which ``norm_linear`` / ``add_rms_norm`` / ``rms_norm`` accept in place of a
    tensor."""
    if _on_gpu(x) and x.dim() == 2 and x.shape[0] <= 256:
        h = _hip()
        if x.shape[0] <= 16:
            sp = h.skinny_gemm_splits(x, weight) if to_norm else None
            if sp:
                return h.skinny_linear(x, weight, bias, splits=sp, defer=True)
            if sp is None and h.skinny_gemm_pays(x, weight):
                return h.skinny_linear(x, weight, bias)
        else:
            c = h.mid_gemm_choice(x, weight, to_norm=to_norm)
            if c is not None:
                return h.mid_linear(x, weight, bias, False, c[0], c[1], c[2], defer=to_norm)
    elif _on_gpu(x) and x.dim() == 2:
        h = _hip()
        if to_norm and residual is not None and h.big_gemm_resid_pays(x, weight, residual):
            return h.big_linear_resid(x, weight, bias, residual)
        if h.big_gemm_pays(x, weight):
            return h.big_linear(x, weight, bias)
    return F.linear(x, weight, bias)


def swiglu_linear(x, weight, bias=None):
    """``silu(x @ Wg.T) * (x @ Wu.T)`` for the fused ``[Wg; Wu]`` weight: one kernel where the
    start-up measurements preferred a hand-written one (decode-sized inputs: weight-streaming
    kernels; prefill-sized: the 256 x 256-tile kernel with the SwiGLU epilogue), otherwise
    GEMM + activation kernel."""
    if _on_gpu(x) and x.dim() == 2 and x.shape[0] <= 256:
        h = _hip()
        if x.shape[0] <= 16:
            if h.skinny_gemm_pays(x, weight, swiglu=True):
                return h.skinny_linear(x, weight, bias, swiglu=True)
        else:
            c = h.mid_gemm_choice(x, weight, swiglu=True)
            if c is not None:
                return h.mid_linear(x, weight, bias, True, c[0], c[1], c[2])
    elif _on_gpu(x) and x.dim() == 2 and _hip().big_gemm_pays(x, weight, swiglu=True):
        return _hip().big_linear(x, weight, bias, swiglu=True)     # prefill: GEMM + SwiGLU in one kernel
    return silu_and_mul(F.linear(x, weight, bias))


def fuses_norm_linear(x, weight, swiglu=False) -> bool:
    """True when ``norm_linear`` runs as one kernel for these operands (decode batches of one
    or two rows on the GPU), provided the weight carries the norm's scale vector."""
    return (_on_gpu(x) and x.dim() == 2 and x.shape[0] <= 2
            and _hip().skinny_norm_gemm_pays(x, weight, swiglu))


def norm_linear(x, residual, gamma, eps, weight, bias=None, swiglu=False, gamma_folded=False):
    """``r = x + residual`` (or ``x``); returns ``(proj(rms_norm(r) * gamma), r)`` where proj is
    ``linear`` or, with ``swiglu``, ``swiglu_linear``.  ``gamma_folded`` says that ``weight``
    was multiplied by ``gamma`` along its input dimension at load time (``gamma`` itself is all
    ones then); only such weights can take the one-kernel path, every other case runs the norm
    kernel followed by the projection."""
    if isinstance(x, ResidualSum):                     # the residual was added by the GEMM
        r = x.r
        if gamma_folded and _hip().big_gemm_ok(r, weight, swiglu):
            return _hip().big_linear(r, weight, bias, swiglu, row_scale=True, eps=eps), r
        h = rms_norm(r, gamma, eps)
        return (swiglu_linear(h, weight, bias) if swiglu else linear(h, weight, bias)), r
    if isinstance(x, PendingReduce):                   # row-parallel partial sum: reduce here
        h, residual = _reduce_add_rms_norm(x, residual, gamma, eps)
        return (swiglu_linear(h, weight, bias) if swiglu else linear(h, weight, bias)), residual
    if isinstance(residual, ResidualShard):
        raise TypeError("a row-sharded residual stream can only be consumed together with a "
                        "pending row-parallel reduce")
    if _is_partials(x):                                # K-split GEMM slices: summed by the norm
        h, residual = x.add_rms_norm(residual, gamma, eps)
        return (swiglu_linear(h, weight, bias) if swiglu else linear(h, weight, bias)), residual
    if gamma_folded and fuses_norm_linear(x, weight, swiglu):
        return _hip().skinny_norm_linear(x, residual, eps, weight, bias, swiglu)
    if residual is None:
        residual, h = x, rms_norm(x, gamma, eps)
    else:
        h, residual = add_rms_norm(x, residual, gamma, eps)
    return (swiglu_linear(h, weight, bias) if swiglu else linear(h, weight, bias)), residual


def sample(logits, temperatures, seed=0, step=0, generator=None):
    """Returns int64 token ids [B]."""
    if _on_gpu(logits):
        return _hip().sample(logits, temperatures, seed, step)
    return torch_ref.sample(logits, temperatures, generator)
