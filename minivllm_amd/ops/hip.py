"""ctypes bindings for the gfx950 kernel library (``_hip_kernels.so``).

Every launcher takes raw device pointers, element strides and the current HIP stream, so a
call made while a hipGraph is being captured on that stream becomes a graph node.  The
library is built in-tree by ``minivllm_amd.build``; importing this module on a machine with a
GPU but without the built library raises (there is no silent PyTorch fallback for GPU
tensors).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch

from .partials import ResidualSum, SplitKPartials      # the types live where CPU-only code can import them

_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                         "_hip_kernels.so")
_LIB_PATH = os.environ.get("MINIVLLM_HIP_LIB", _LIB_PATH)        # kernel experiments: another build
if not os.path.exists(_LIB_PATH):
    raise ImportError(
        f"{_LIB_PATH} is missing: build the HIP kernels first with "
        f"`python -m minivllm_amd.build` (needs hipcc; cross-compiles for gfx950 without a GPU)")
_lib = ctypes.CDLL(_LIB_PATH)

_c_long, _c_int, _c_float, _c_void, _c_u64 = (ctypes.c_long, ctypes.c_int, ctypes.c_float,
                                               ctypes.c_void_p, ctypes.c_uint64)


def _sig(name, *argtypes):
    f = getattr(_lib, name)
    f.argtypes = list(argtypes)
    f.restype = ctypes.c_int
    return f


_rms = _sig("mvl_rms_norm", _c_void, _c_void, _c_void, _c_long, _c_int, _c_long, _c_long,
            _c_float, _c_int, _c_void)
_add_rms = _sig("mvl_add_rms_norm", _c_void, _c_void, _c_void, _c_void, _c_long, _c_int, _c_long,
                _c_long, _c_long, _c_float, _c_int, _c_void)
_silu = _sig("mvl_silu_and_mul", _c_void, _c_void, _c_long, _c_int, _c_long, _c_long, _c_int,
             _c_void)
_rope = _sig("mvl_qk_norm_rope_store", _c_void, _c_long, _c_void, _c_void, _c_void, _c_void,
             _c_float, _c_long, _c_int, _c_int, _c_int, _c_void, _c_void, _c_long, _c_long,
             _c_long, _c_int, _c_void, _c_int, _c_int, _c_int, _c_int, _c_void)
_store = _sig("mvl_store_kvcache", _c_void, _c_void, _c_long, _c_long, _c_void, _c_void, _c_long,
              _c_long, _c_long, _c_int, _c_void, _c_long, _c_int, _c_int, _c_void)
_sample = _sig("mvl_sample", _c_void, _c_long, _c_int, _c_int, _c_int, _c_int, _c_void, _c_u64,
               _c_void, _c_void, _c_void, _c_int, _c_int, _c_void, _c_void, _c_void)
_sample_filtered = _sig("mvl_sample_filtered", _c_void, _c_long, _c_int, _c_int, _c_void, _c_void,
                        _c_void, _c_void, _c_void, _c_u64, _c_void, _c_void, _c_void, _c_int,
                        _c_void)
_decode = _sig("mvl_paged_attention_decode", _c_void, _c_long, _c_long, _c_void, _c_void, _c_long,
               _c_long, _c_long, _c_void, _c_int, _c_void, _c_void, _c_long, _c_long, _c_void,
               _c_void, _c_void, _c_float, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
               _c_int, _c_void, _c_void, _c_void, _c_void, _c_void, _c_float, _c_void)
_prefill = _sig("mvl_flash_attention_prefill", _c_void, _c_long, _c_long, _c_void, _c_void,
                _c_long, _c_long, _c_long, _c_void, _c_long, _c_long, _c_void, _c_void, _c_void,
                _c_int, _c_int, _c_float, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                _c_void, _c_void, _c_void, _c_float, _c_int, _c_void, _c_int, _c_void)

_naive = _sig("mvl_naive_attention", _c_void, _c_long, _c_long, _c_void, _c_void, _c_long, _c_long,
              _c_void, _c_long, _c_long, _c_void, _c_float, _c_int, _c_int, _c_int, _c_int, _c_int,
              _c_int, _c_void)

_skinny = _sig("mvl_skinny_gemm", _c_void, _c_long, _c_void, _c_void, _c_void, _c_long, _c_int,
               _c_int, _c_int, _c_int, _c_int, _c_void)

_skinny_splitk = _sig("mvl_skinny_gemm_splitk", _c_void, _c_long, _c_void, _c_void, _c_int, _c_int,
                      _c_int, _c_int, _c_int, _c_void)
_skinny_norm = _sig("mvl_skinny_norm_gemm", _c_void, _c_long, _c_void, _c_long, _c_void, _c_long,
                    _c_float, _c_void, _c_void, _c_void, _c_long, _c_int, _c_int, _c_int,
                    _c_int, _c_int, _c_void)

_mid = _sig("mvl_mid_gemm", _c_void, _c_long, _c_void, _c_void, _c_void, _c_long, _c_void, _c_int,
            _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void)

_big = _sig("mvl_big_gemm", _c_void, _c_long, _c_void, _c_void, _c_void, _c_long, _c_int, _c_int,
            _c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _c_void)

_mid_eff_splits = _lib.mvl_mid_effective_splits
_mid_eff_splits.argtypes = [_c_int, _c_int]
_mid_eff_splits.restype = _c_int
_mid_reduce = _sig("mvl_mid_reduce", _c_void, _c_void, _c_void, _c_long, _c_int, _c_int, _c_int,
                   _c_int, _c_void)
_splitk_norm = _sig("mvl_splitk_add_rms_norm", _c_void, _c_void, _c_void, _c_void, _c_void, _c_int,
                    _c_int, _c_int, _c_long, _c_long, _c_float, _c_int, _c_void)

_DT = {torch.bfloat16: 0, torch.float16: 1}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"HIP kernels take bf16/fp16 tensors, got {t.dtype}")


def _stream() -> int:
    # raw handle of the current stream: the Stream-object route costs ~9 us per call, which adds
    # up over the few hundred kernel launches of an eager prefill step
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _check(rc: int, name: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{name} failed (code {rc}): bad shape/alignment for the kernel or "
                           f"a HIP launch error")


def _rows2d(x: torch.Tensor) -> Tuple[torch.Tensor, int, int]:
    """View x as [rows, hidden] with unit inner stride; returns (x2d, rows, row_stride)."""
    h = x.shape[-1]
    x2 = x.reshape(-1, h)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    return x2, x2.shape[0], (x2.stride(0) if x2.shape[0] > 1 else h)


def _i32(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.int32 else t.to(torch.int32)


# --------------------------------------------------------------------- norms
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    x2, rows, xs = _rows2d(x)
    out = torch.empty((rows, x.shape[-1]), dtype=x.dtype, device=x.device)
    w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
    _check(_rms(x2.data_ptr(), w.data_ptr(), out.data_ptr(), rows, x.shape[-1], xs, x.shape[-1],
                eps, _dt(x), _stream()), "rms_norm")
    return out.view(x.shape)


def add_rms_norm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float):
    """In place: ``residual <- x + residual`` and ``x <- norm(residual)``; returns (x, residual)."""
    x2, rows, xs = _rows2d(x)
    r2, _, rs = _rows2d(residual)
    if x2.data_ptr() != x.data_ptr() or r2.data_ptr() != residual.data_ptr():
        raise ValueError("add_rms_norm needs row-contiguous tensors (it updates them in place)")
    w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
    _check(_add_rms(x2.data_ptr(), r2.data_ptr(), w.data_ptr(), x2.data_ptr(), rows, x.shape[-1],
                    xs, rs, xs, eps, _dt(x), _stream()), "add_rms_norm")
    return x, residual


def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    x2, rows, xs = _rows2d(x)
    inter = x.shape[-1] // 2
    out = torch.empty((rows, inter), dtype=x.dtype, device=x.device)
    code = 2 if x.dtype == torch.float32 else _dt(x)       # fp32 rows have their own kernel
    _check(_silu(x2.data_ptr(), out.data_ptr(), rows, inter, xs, inter, code, _stream()),
           "silu_and_mul")
    return out.view(*x.shape[:-1], inter)


# ---------------------------------------------------------------- rope / store
FP8_KV_DTYPE = torch.float8_e4m3fn        # gfx950 converts to / from the OCP e4m3 format


def _kv_fp8(cache: Optional[torch.Tensor]) -> int:
    return int(cache is not None and cache.dtype == FP8_KV_DTYPE)


def _cache_strides(cache: torch.Tensor):
    """cache is logically [block, token, kv_head, d] with arbitrary physical order."""
    if cache.stride(3) != 1:
        raise ValueError("KV cache head_dim must be contiguous")
    return cache.stride(0), cache.stride(1), cache.stride(2)


def qk_norm_rope_store(qkv: torch.Tensor, positions: torch.Tensor, cos_sin: torch.Tensor,
                       q_weight: Optional[torch.Tensor], k_weight: Optional[torch.Tensor],
                       eps: float, num_heads: int, num_kv_heads: int, head_dim: int,
                       k_cache: Optional[torch.Tensor], v_cache: Optional[torch.Tensor],
                       slot_mapping: Optional[torch.Tensor], skip_q: bool = False,
                       k_in_place: bool = True):
    """Works in place on ``qkv`` [T, (Hq+2Hkv)*D]; returns (q, k, v) views of it.  ``skip_q``:
    the Q rows are left as projected (the prefill attention kernel transforms them itself when
    it is handed ``q_rope``).  ``k_in_place=False`` (needs a cache): the transformed K rows go to
    the cache only and the K rows of ``qkv`` keep the raw projection (for steps whose attention
    reads K from the cache)."""
    assert qkv.dim() == 2 and qkv.stride(1) == 1
    T = qkv.shape[0]
    qs, ks = num_heads * head_dim, num_kv_heads * head_dim
    assert qkv.shape[1] == qs + 2 * ks
    # the kernel moves 16 bytes per lane on the qkv rows, the norm weights and the cache rows
    if qkv.data_ptr() % 16 or (T > 1 and qkv.stride(0) % 8) or head_dim % 16:
        raise ValueError("qk_norm_rope_store needs 16-byte aligned qkv rows (row stride a multiple "
                         "of 8 elements) and head_dim a multiple of 16")
    for wgt in (q_weight, k_weight):
        if wgt is not None and (wgt.data_ptr() % 16 or not wgt.is_contiguous()):
            raise ValueError("q/k norm weights must be contiguous and 16-byte aligned")
    if cos_sin.dtype != torch.float32:
        raise TypeError("cos/sin table must be fp32")
    positions = _i32(positions)
    have_cache = k_cache is not None and slot_mapping is not None
    if have_cache:
        sb, st, sh = _cache_strides(k_cache)
        assert _cache_strides(v_cache) == (sb, st, sh)
        slot_mapping = _i32(slot_mapping)
        bs = k_cache.shape[1]
        unit = 8 if _kv_fp8(k_cache) else 16          # bytes per access on a cache row
        esz = k_cache.element_size()
        if any((v * esz) % unit for v in (sb, st, sh)) or k_cache.data_ptr() % unit \
                or v_cache.data_ptr() % unit:
            raise ValueError(f"KV cache rows must be {unit}-byte aligned (strides {sb}, {st}, {sh})")
    else:
        sb = st = sh = 0
        bs = 1
    qw = q_weight.data_ptr() if q_weight is not None else None
    kw = k_weight.data_ptr() if k_weight is not None else None
    if q_weight is not None:
        assert q_weight.dtype == qkv.dtype and k_weight.dtype == qkv.dtype
    _check(_rope(qkv.data_ptr(), qkv.stride(0), positions.data_ptr(), cos_sin.data_ptr(), qw, kw,
                 eps, T, num_heads, num_kv_heads, head_dim,
                 k_cache.data_ptr() if have_cache else None,
                 v_cache.data_ptr() if have_cache else None, sb, st, sh, bs,
                 slot_mapping.data_ptr() if have_cache else None, _dt(qkv),
                 _kv_fp8(k_cache) if have_cache else 0, int(skip_q),
                 int(k_in_place or not have_cache), _stream()),
           "qk_norm_rope_store")
    q = qkv[:, :qs].view(T, num_heads, head_dim)
    k = qkv[:, qs:qs + ks].view(T, num_kv_heads, head_dim)
    v = qkv[:, qs + ks:].view(T, num_kv_heads, head_dim)
    return q, k, v


def store_kvcache(key: torch.Tensor, value: torch.Tensor, k_cache: torch.Tensor,
                  v_cache: torch.Tensor, slot_mapping: torch.Tensor) -> None:
    T, hkv, d = key.shape
    if _kv_fp8(k_cache):
        raise NotImplementedError("the stand-alone scatter copies 16-bit values; an fp8 cache is "
                                  "written by qk_norm_rope_store")
    assert key.stride(2) == 1 and key.stride(1) == d and value.stride(2) == 1 and value.stride(1) == d
    sb, st, sh = _cache_strides(k_cache)
    assert _cache_strides(v_cache) == (sb, st, sh)
    assert key.dtype == k_cache.dtype
    slot_mapping = _i32(slot_mapping)
    _check(_store(key.data_ptr(), value.data_ptr(), key.stride(0), value.stride(0),
                  k_cache.data_ptr(), v_cache.data_ptr(), sb, st, sh, k_cache.shape[1],
                  slot_mapping.data_ptr(), T, hkv, d, _stream()), "store_kvcache")


# ------------------------------------------------------------------ attention
def _q_strides(q: torch.Tensor):
    assert q.dim() == 3 and q.stride(2) == 1, "q must be [T, H, D] with contiguous D"
    return q.stride(0), q.stride(1)


def _q_rope_args(q: torch.Tensor, q_rope):
    """``q_rope`` = (positions [T] or None, cos_sin fp32 [max_pos, D], q_norm weight or None,
    eps): the raw-Q transform the prefill kernel applies while loading its Q rows -> kernel
    arguments.  Without positions the kernel takes a query's place in its context (the lengths
    must then stay below the table's rows)."""
    if q_rope is None:
        return None, None, None, 0.0
    positions, cos_sin, q_weight, eps = q_rope
    if cos_sin.dtype != torch.float32 or cos_sin.shape[-1] != q.shape[-1] or not cos_sin.is_contiguous():
        raise TypeError("cos/sin table must be contiguous fp32 [positions, head_dim]")
    if positions is not None:           # None: row i of a sequence sits at k_len - q_len + i
        positions = _i32(positions)
        assert positions.numel() >= q.shape[0] and positions.is_contiguous()
    if q_weight is not None:
        assert q_weight.dtype == q.dtype and q_weight.numel() == q.shape[-1] and q_weight.is_contiguous()
        assert q_weight.data_ptr() % 16 == 0
    # keep the converted positions alive until the launch has been issued
    return positions, cos_sin.data_ptr(), q_weight.data_ptr() if q_weight is not None else None, float(eps)


def flash_attention_prefill(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                            cu_seqlens: torch.Tensor, scale: float,
                            max_seqlen: Optional[int] = None, q_rope=None) -> torch.Tensor:
    T, hq, d = q.shape
    hkv = k.shape[1]
    cu = _i32(cu_seqlens)
    if max_seqlen is None:
        max_seqlen = int((cu[1:] - cu[:-1]).max().item())
    qs_t, qs_h = _q_strides(q)
    ks_t, ks_h = _q_strides(k)
    assert _q_strides(v) == (ks_t, ks_h), "k and v must share strides"
    out = torch.empty((T, hq, d), dtype=q.dtype, device=q.device)
    pos, cs, qw, eps = _q_rope_args(q, q_rope)
    _check(_prefill(q.data_ptr(), qs_t, qs_h, k.data_ptr(), v.data_ptr(), ks_t, ks_h, 0,
                    out.data_ptr(), out.stride(0), out.stride(1), cu.data_ptr(), cu.data_ptr(),
                    None, 0, 1, scale, cu.numel() - 1, hq, hkv, d, max_seqlen, 0, _dt(q),
                    pos.data_ptr() if pos is not None else None, cs, qw, eps, 0, None, 0,
                    _stream()), "flash_attention_prefill")
    return out


NAIVE_ATTENTION_MAX_LEN = 128


def naive_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cu_seqlens: torch.Tensor,
                    scale: float, max_seqlen: int) -> torch.Tensor:
    """Full-matrix causal attention for sequences of at most 128 tokens (benchmark baseline)."""
    if max_seqlen > NAIVE_ATTENTION_MAX_LEN:
        raise ValueError(f"naive attention holds the score matrix in LDS: seq_len <= "
                         f"{NAIVE_ATTENTION_MAX_LEN}, got {max_seqlen}")
    T, hq, d = q.shape
    cu = _i32(cu_seqlens)
    qs_t, qs_h = _q_strides(q)
    ks_t, ks_h = _q_strides(k)
    assert _q_strides(v) == (ks_t, ks_h)
    out = torch.empty((T, hq, d), dtype=q.dtype, device=q.device)
    _check(_naive(q.data_ptr(), qs_t, qs_h, k.data_ptr(), v.data_ptr(), ks_t, ks_h, out.data_ptr(),
                  out.stride(0), out.stride(1), cu.data_ptr(), scale, cu.numel() - 1, hq,
                  k.shape[1], d, max_seqlen, _dt(q), _stream()), "naive_attention")
    return out


def paged_attention_prefill(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                            cu_seqlens_q: torch.Tensor, cu_seqlens_k: torch.Tensor,
                            block_tables: torch.Tensor, scale: float,
                            max_seqlen_q: Optional[int] = None,
                            out: Optional[torch.Tensor] = None,
                            max_seqlen_k: Optional[int] = None,
                            total_k: Optional[int] = None, q_rope=None,
                            work_items: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Causal prefill through the block table (``q_len <= k_len`` per sequence).  An fp8 (e4m3)
    cache is read by the same kernel, which widens the tiles while staging them;
    ``max_seqlen_k`` / ``total_k`` are accepted for callers that know them and not needed.
    ``work_items`` (int32, device; see ``prefill_work_items``): the launch then has one workgroup
    per list slot and query head, independent of the lengths - the form a captured graph needs."""
    T, hq, d = q.shape
    hkv = k_cache.shape[2]
    cq, ck, bt = _i32(cu_seqlens_q), _i32(cu_seqlens_k), _i32(block_tables)
    if max_seqlen_q is None:
        max_seqlen_q = int((cq[1:] - cq[:-1]).max().item())
    qs_t, qs_h = _q_strides(q)
    sb, st, sh = _cache_strides(k_cache)
    assert _cache_strides(v_cache) == (sb, st, sh)
    assert bt.stride(1) == 1
    if _kv_fp8(k_cache):        # 16-byte chunks of 16 values
        assert d % 16 == 0 and st % 16 == 0 and sh % 16 == 0 and sb % 16 == 0 \
            and k_cache.data_ptr() % 16 == 0 and v_cache.data_ptr() % 16 == 0
    if out is None:
        out = torch.empty((T, hq, d), dtype=q.dtype, device=q.device)
    assert out.shape == (T, hq, d) and out.stride(2) == 1 and out.dtype == q.dtype
    pos, cs, qw, eps = _q_rope_args(q, q_rope)
    if work_items is not None:
        assert work_items.dtype == torch.int32 and work_items.is_contiguous() and work_items.is_cuda
        if work_items.numel() == 0:
            return out
        max_seqlen_q = max(max_seqlen_q, 1)
    _check(_prefill(q.data_ptr(), qs_t, qs_h, k_cache.data_ptr(), v_cache.data_ptr(), st, sh, sb,
                    out.data_ptr(), out.stride(0), out.stride(1), cq.data_ptr(), ck.data_ptr(),
                    bt.data_ptr(), bt.stride(0), k_cache.shape[1], scale, cq.numel() - 1, hq, hkv,
                    d, max_seqlen_q, 1, _dt(q), pos.data_ptr() if pos is not None else None, cs, qw,
                    eps, _kv_fp8(k_cache), work_items.data_ptr() if work_items is not None else None,
                    work_items.numel() if work_items is not None else 0, _stream()),
           "paged_attention_prefill")
    return out


PREFILL_Q_TILE = 128          # query rows per tile of the prefill kernel (csrc/attn_prefill.hip)


def prefill_work_items(q_lens, k_lens=None, slots: Optional[int] = None):
    """The prefill kernel's work list for sequences with ``q_lens`` new tokens (numpy int array):
    one entry ``sequence << 16 | pair`` per heavy / light pair of 128-row query tiles, long
    contexts first, padded with -1 to ``slots`` entries (default: the next multiple of 32, the
    granularity the kernel deals the list out in).  Never more than
    ``prefill_item_slots(sum(q_lens), len(q_lens))`` real entries."""
    import numpy as np
    q_lens = np.asarray(q_lens, dtype=np.int64)
    n_qt = (q_lens + PREFILL_Q_TILE - 1) // PREFILL_Q_TILE
    n_pairs = (n_qt + 1) // 2
    order = np.argsort(-(np.asarray(k_lens, dtype=np.int64) if k_lens is not None else q_lens),
                       kind="stable")
    reps = n_pairs[order]
    seq = np.repeat(order, reps)
    first = np.cumsum(reps) - reps
    pair = np.arange(int(reps.sum())) - np.repeat(first, reps)
    items = ((seq << 16) | pair).astype(np.int32)
    if slots is None:
        slots = (items.size + 31) // 32 * 32
    if items.size > slots or slots % 32:
        raise ValueError(f"{items.size} work items do not fit {slots} slots (a multiple of 32)")
    return np.concatenate([items, np.full(slots - items.size, -1, np.int32)])


def prefill_item_slots(num_tokens: int, num_seqs: int) -> int:
    """Upper bound of the work-list length of any step with at most that many new tokens in at
    most that many sequences (every sequence: ceil(ceil(n / 128) / 2) <= n / 256 + 1 pairs)."""
    n = num_tokens // (2 * PREFILL_Q_TILE) + num_seqs
    return (n + 31) // 32 * 32           # the kernel deals the list to the 8 XCDs in chunks of 4


def decode_num_splits(batch: int, num_kv_heads: int, max_context_len: Optional[int]) -> int:
    """Context splits so that ~2 workgroups per CU exist even at batch 1; never more splits
    than 128-token chunks of the longest context."""
    wgs = max(batch * num_kv_heads, 1)
    # one workgroup per CU or more: no split (batch 256 x 1 KV head at 1024 tokens: 24.5 us
    # unsplit, 27 us in two splits; what the second split adds is its merge, not bandwidth)
    n = 1 if wgs >= 256 else max(1, min(32, 512 // wgs))
    if max_context_len is not None:
        n = max(1, min(n, (max_context_len + 127) // 128))
    return n


# The decode kernel merges its context splits in the workgroup that finishes last; the tickets
# that decide which one that is are zero between launches (the kernel resets them), so one
# buffer per device serves every launch that cannot overlap another one on that device.
_DECODE_TICKETS: dict = {}
DECODE_TICKET_WORDS = 1 << 16


def decode_ticket_words(device: torch.device, need: int) -> torch.Tensor:
    """The device's ticket buffer (at least ``need`` ints).  It is allocated on first use, which
    must not happen inside a graph capture (the memory would belong to the graph's pool): the
    model runner calls this at start-up.  A captured graph has the buffer's address baked in, so
    a buffer is never freed or replaced: a caller that needs more words than exist gets a new,
    larger buffer for ITS launches while the old one stays alive for the graphs that use it."""
    idx = device.index if device.index is not None else torch._C._cuda_getDevice()
    bufs = _DECODE_TICKETS.setdefault(idx, [])
    for buf in bufs:
        if buf.numel() >= need:
            return buf
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("the decode kernel's ticket buffer must exist before graph capture: "
                           "call hip.decode_ticket_words(device, n) first")
    buf = torch.zeros(max(need, DECODE_TICKET_WORDS), dtype=torch.int32, device=device)
    bufs.append(buf)                 # earlier (smaller) buffers stay referenced here for good
    return buf


def paged_attention_decode(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                           block_tables: torch.Tensor, context_lens: torch.Tensor, scale: float,
                           max_context_len: Optional[int] = None,
                           num_splits: Optional[int] = None,
                           out: Optional[torch.Tensor] = None,
                           tickets: Optional[torch.Tensor] = None,
                           new_kv=None) -> torch.Tensor:
    """``tickets``: zeroed int32 scratch of B * Hkv * ceil(G / 16) words for callers that run
    decode launches concurrently on several streams of one device (each stream its own).
    ``new_kv`` = (k [B, Hkv, D], v [B, Hkv, D], cos_sin fp32 [positions, D], q_norm weight or
    None, k_norm weight or None, eps), 16-bit caches only: ``q``, ``k`` and ``v`` are the raw
    projections of the step's new tokens (views of one qkv tensor), whose positions are
    ``context_lens - 1``.  The kernel norms and rotates q and k itself and writes k / v into the
    caches at those positions: the stand-alone norm + rotate + store kernel is not needed."""
    B, hq, d = q.shape
    hkv = k_cache.shape[2]
    bt, ctx = _i32(block_tables), _i32(context_lens)
    qs_b, qs_h = _q_strides(q)
    sb, st, sh = _cache_strides(k_cache)
    assert _cache_strides(v_cache) == (sb, st, sh)
    assert bt.stride(1) == 1
    ns = num_splits or decode_num_splits(B, hkv, max_context_len)
    if out is None:
        out = torch.empty((B, hq, d), dtype=q.dtype, device=q.device)
    assert out.shape == (B, hq, d) and out.stride(2) == 1 and out.dtype == q.dtype
    if ns > 1:
        part_acc = torch.empty(B * hq * ns * d, dtype=torch.float32, device=q.device)
        part_ml = torch.empty(B * hq * ns * 2, dtype=torch.float32, device=q.device)
        need = B * hkv * ((hq // hkv + 15) // 16)
        if tickets is None:
            tickets = decode_ticket_words(q.device, need)
        assert tickets.dtype == torch.int32 and tickets.numel() >= need and tickets.is_contiguous()
        pa, pm, tk = part_acc.data_ptr(), part_ml.data_ptr(), tickets.data_ptr()
    else:
        pa = pm = tk = None
    fuse = (None, None, None, None, None, 0.0)
    if new_kv is not None:
        k_new, v_new, cos_sin, q_w, k_w, eps = new_kv
        if _kv_fp8(k_cache):
            raise ValueError("the fused decode form writes 16-bit caches only")
        if cos_sin.dtype != torch.float32 or cos_sin.shape[-1] != d or not cos_sin.is_contiguous():
            raise TypeError("cos/sin table must be contiguous fp32 [positions, head_dim]")
        for t in (k_new, v_new):
            assert t.shape == (B, hkv, d) and t.dtype == q.dtype and t.stride(2) == 1 \
                and t.stride(1) == d and (B == 1 or t.stride(0) == qs_b) and t.data_ptr() % 16 == 0
        assert q.data_ptr() % 16 == 0 and qs_b % 8 == 0 and qs_h % 8 == 0
        assert k_cache.data_ptr() % 16 == 0 and v_cache.data_ptr() % 16 == 0 \
            and sb % 8 == 0 and st % 8 == 0 and sh % 8 == 0
        for w_ in (q_w, k_w):
            assert w_ is None or (w_.dtype == q.dtype and w_.numel() == d and w_.is_contiguous()
                                  and w_.data_ptr() % 16 == 0)
        fuse = (k_new.data_ptr(), v_new.data_ptr(), cos_sin.data_ptr(),
                q_w.data_ptr() if q_w is not None else None,
                k_w.data_ptr() if k_w is not None else None, float(eps))
    _check(_decode(q.data_ptr(), qs_b, qs_h, k_cache.data_ptr(), v_cache.data_ptr(), sb, st, sh,
                   bt.data_ptr(), bt.stride(0), ctx.data_ptr(), out.data_ptr(), out.stride(0),
                   out.stride(1), pa, pm, tk, scale, B, hq, hkv, d, k_cache.shape[1], ns, _dt(q),
                   _kv_fp8(k_cache), *fuse, _stream()), "paged_attention_decode")
    return out


# ------------------------------------------------------- weight-streaming GEMM
SKINNY_MAX_ROWS = 16
SKINNY_PAYS_ROWS = 16                 # up to here the kernel beats the library (see skinny_gemm_pays)
SKINNY_MIN_WEIGHT_BYTES = 8 << 20     # below this a projection is launch-latency bound anyway


def skinny_gemm_ok(x: torch.Tensor, weight: torch.Tensor, swiglu: bool = False) -> bool:
    """Operands the weight-streaming kernel can take at all: <= 16 rows, K a multiple of 128,
    output columns a multiple of 16, contiguous weights of the activations' dtype."""
    if x.dim() != 2 or x.shape[0] == 0 or x.shape[0] > SKINNY_MAX_ROWS or x.dtype not in _DT:
        return False
    n, k = weight.shape
    n_out = n // 2 if swiglu else n
    return (weight.dtype == x.dtype and weight.is_contiguous() and x.stride(1) == 1
            and k % 128 == 0 and n_out % 16 == 0 and n % (32 if swiglu else 16) == 0)


# (row bucket, N, K, swiglu) -> use the kernel?  Filled by tune_skinny_gemm on the machine the
# engine runs on; shapes that were never measured fall back to the rule in skinny_gemm_pays.
_SKINNY_CHOICE: dict = {}
_SKINNY_ROW_BUCKETS = (1, 8, 16)


def _row_bucket(m: int) -> int:
    return 1 if m <= 2 else 8 if m <= 8 else 16


def skinny_gemm_pays(x: torch.Tensor, weight: torch.Tensor, swiglu: bool = False) -> bool:
    """Whether the weight-streaming kernel is the faster choice for this call.  Which side wins
    depends on the shape in ways no simple rule captures (MI355X, weights streamed cold, 1 row:
    Qwen3-8B o-projection 8.3 vs 15.2 us and gate_up+SwiGLU 35.9 vs 45.5 us for the kernel, but
    Qwen3-32B gate_up+SwiGLU 124 vs 91 us against it, and its down projection 81 vs 134 us for
    it again), so the engine measures both on its own weights at start-up (tune_skinny_gemm)
    and this function looks the answer up.  Unmeasured shapes: the kernel for fused SwiGLU and
    for N, K <= 4096 when the matrix is large enough to cover the chip, the library otherwise."""
    if x.dim() != 2 or x.shape[0] > SKINNY_PAYS_ROWS or not skinny_gemm_ok(x, weight, swiglu):
        return False
    n, k = weight.shape
    choice = _SKINNY_CHOICE.get((_row_bucket(x.shape[0]), n, k, swiglu))
    if choice is not None:
        return choice
    n_out = n // 2 if swiglu else n
    if n_out // 16 < 192 or n * k * 2 < SKINNY_MIN_WEIGHT_BYTES:
        return False                       # too few tiles to cover the chip / launch-bound anyway
    return swiglu or (n <= 4096 and k <= 4096)


def skinny_gemm_splits(x: torch.Tensor, weight: torch.Tensor) -> int:
    """For a projection whose output goes straight into residual add + RMSNorm: the K split
    (over workgroups) that was measured fastest together with that norm, 0 = library GEMM,
    None = never measured (the caller falls back to ``skinny_gemm_pays``)."""
    if x.dim() != 2 or x.shape[0] > SKINNY_PAYS_ROWS or not skinny_gemm_ok(x, weight):
        return None
    n, k = weight.shape
    return _SKINNY_CHOICE.get((_row_bucket(x.shape[0]), n, k, "norm"))


def tune_skinny_to_norm(weights, margin: float = 0.97) -> dict:
    """``tune_skinny_gemm`` for a projection followed by residual add + RMSNorm (o, down): the
    library GEMM + norm kernel against the weight-streaming kernel with K split 1..4 ways over
    workgroups, whose slices the norm kernel sums.  Records the best split (0 = library) under
    the ``"norm"`` key; returns {rows: (library us, best kernel us, split)}."""
    import torch.nn.functional as F
    weights = list(weights)
    n, k = weights[0].shape
    dtype, dev = weights[0].dtype, weights[0].device
    out = {}
    if not skinny_gemm_ok(torch.zeros(1, k, dtype=dtype, device=dev), weights[0]) \
            or n * k * 2 < (1 << 20) or n > 32768:
        return out
    reps = max(1, min(8, -(-(640 << 20) // (n * k * 2 * len(weights)))))
    gamma = torch.ones(n, dtype=dtype, device=dev)
    for m in _SKINNY_ROW_BUCKETS:
        x = torch.randn(m, k, dtype=dtype, device=dev)
        resid = torch.zeros(m, n, dtype=dtype, device=dev)
        t_lib = _graph_time_ms(lambda xx, w: add_rms_norm(F.linear(xx, w), resid, gamma, 1e-6),
                               x, weights, reps)
        best, t_best = 0, t_lib * margin
        for sp in (1, 2, 3, 4):
            if not skinny_splits_ok(k, sp):
                continue

            def ker(xx, w):
                y = skinny_linear(xx, w, splits=sp, defer=True)
                if isinstance(y, SplitKPartials):
                    return y.add_rms_norm(resid, gamma, 1e-6)
                return add_rms_norm(y, resid, gamma, 1e-6)
            t = _graph_time_ms(ker, x, weights, reps)
            if t < t_best:
                best, t_best = sp, t
        _SKINNY_CHOICE[(m, n, k, "norm")] = best
        out[m] = (round(t_lib * 1e3, 1), round(t_best * 1e3, 1) if best else None, best)
    return out


def tune_skinny_gemm(weights, swiglu: bool = False, margin: float = 0.97) -> dict:
    """Time the weight-streaming kernel against the library GEMM (+ activation kernel when
    ``swiglu``) for one projection and record the winner per row bucket.  ``weights`` are the
    same-shaped matrices of several layers: cycling through them keeps the reads cold, as in a
    real decode step.  Each candidate is captured into a hipGraph (no launch gaps in the
    measurement) and replayed three times; the kernel must win by ``1 - margin``."""
    import torch.nn.functional as F
    weights = list(weights)
    n, k = weights[0].shape
    dtype, dev = weights[0].dtype, weights[0].device
    out = {}
    probe = torch.zeros(1, k, dtype=dtype, device=dev)
    if not skinny_gemm_ok(probe, weights[0], swiglu) or n * k * 2 < (1 << 20):
        return out                             # below 1 MB either choice is launch latency
    # enough distinct matrices per pass that the 256 MB Infinity Cache cannot hold them (small
    # matrices stay cached in the real step as well: no point in more than a few passes)
    reps = max(1, min(8, -(-(640 << 20) // (n * k * 2 * len(weights)))))

    def lib(x, w):
        y = F.linear(x, w)
        return silu_and_mul(y) if swiglu else y

    def ker(x, w):
        return skinny_linear(x, w, swiglu=swiglu)

    def graph_ms(fn, x):
        for w in weights[:2]:
            fn(x, w)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(reps):
                for w in weights:
                    fn(x, w)
        g.replay()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(3):
            g.replay()
        t1.record()
        torch.cuda.synchronize()
        return t0.elapsed_time(t1) / (3 * reps * len(weights))

    for m in _SKINNY_ROW_BUCKETS:
        x = torch.randn(m, k, dtype=dtype, device=dev)
        t_lib, t_ker = graph_ms(lib, x), graph_ms(ker, x)
        use = t_ker < margin * t_lib
        _SKINNY_CHOICE[(m, n, k, swiglu)] = use
        out[m] = (round(t_lib * 1e3, 1), round(t_ker * 1e3, 1), use)
    return out


def skinny_splits_ok(k: int, splits: int) -> bool:
    return splits >= 1 and k % splits == 0 and (k // splits) % 128 == 0


def skinny_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                  swiglu: bool = False, splits: int = 1, defer: bool = False):
    """``x @ weight.T (+ bias)`` for <= 16 rows; with ``swiglu`` the weight is the fused
    [gate; up] matrix and the result is ``silu(gate) * up``.  ``splits`` > 1 (plain epilogue
    only) slices K over workgroups; the fp32 slices are summed by a second kernel or, with
    ``defer``, handed over as ``SplitKPartials`` to the norm that consumes the projection."""
    M, K = x.shape
    N = weight.shape[0]
    if splits > 1 and not swiglu and skinny_splits_ok(K, splits):
        ws = torch.empty((splits, M, N), dtype=torch.float32, device=x.device)
        _check(_skinny_splitk(x.data_ptr(), x.stride(0), weight.data_ptr(), ws.data_ptr(), M, N, K,
                              splits, _dt(x), _stream()), "skinny_gemm_splitk")
        part = SplitKPartials(ws, bias, x.dtype)
        return part if defer else part.materialize()
    out = torch.empty((M, N // 2 if swiglu else N), dtype=x.dtype, device=x.device)
    _check(_skinny(x.data_ptr(), x.stride(0), weight.data_ptr(),
                   bias.data_ptr() if bias is not None else None, out.data_ptr(), out.stride(0),
                   M, N, K, 1 if swiglu else 0, _dt(x), _stream()), "skinny_gemm")
    return out


SKINNY_NORM_MAX_ROWS = 2          # beyond this the two-kernel path is faster (measured)


def skinny_norm_gemm_pays(x: torch.Tensor, weight: torch.Tensor, swiglu: bool = False) -> bool:
    """Shapes for which (residual add + RMSNorm + projection) runs as ONE weight-streaming
    kernel: 1-2 rows and a projection the plain weight-streaming kernel takes (also QKV-sized
    outputs, where dropping the norm launch outweighs the library's small GEMM advantage)."""
    if x.dim() != 2 or x.shape[0] > SKINNY_NORM_MAX_ROWS or not skinny_gemm_ok(x, weight, swiglu):
        return False
    return swiglu or weight.shape[0] <= 8192


def skinny_norm_linear(x: torch.Tensor, residual: Optional[torch.Tensor], eps: float,
                       weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                       swiglu: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """``r = x + residual`` (``x`` when residual is None); returns
    ``(linear_or_swiglu(r * rsqrt(mean(r^2) + eps)), r)`` for a ``weight`` that already carries
    the norm's scale vector.  ``r`` is a new tensor: the inputs are left untouched (other
    workgroups of the kernel are still reading them)."""
    M, K = x.shape
    N = weight.shape[0]
    out = torch.empty((M, N // 2 if swiglu else N), dtype=x.dtype, device=x.device)
    if residual is not None:
        assert residual.shape == x.shape and residual.stride(1) == 1
        r = torch.empty((M, K), dtype=x.dtype, device=x.device)
        rin, rin_s, rout, rout_s = residual.data_ptr(), residual.stride(0), r.data_ptr(), K
    else:
        r, rin, rin_s, rout, rout_s = x, None, 0, None, 0
    _check(_skinny_norm(x.data_ptr(), x.stride(0), rin, rin_s, rout, rout_s, eps,
                        weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                        out.data_ptr(), out.stride(0), M, N, K, 1 if swiglu else 0, _dt(x),
                        _stream()), "skinny_norm_gemm")
    return out, r


# ------------------------------------------- mid-batch weight-streaming GEMM (17..256 rows)
MID_MAX_ROWS = 256
_NUM_CUS = 256


def mid_gemm_ok(x: torch.Tensor, weight: torch.Tensor, swiglu: bool = False) -> bool:
    """Operands the mid-batch kernel takes: <= 256 rows with 16-byte aligned rows, K a multiple
    of 64, N a multiple of 64 (SwiGLU: N/2 a multiple of 32), contiguous same-dtype weights."""
    if x.dim() != 2 or x.shape[0] == 0 or x.shape[0] > MID_MAX_ROWS or x.dtype not in _DT:
        return False
    n, k = weight.shape
    return (weight.dtype == x.dtype and weight.is_contiguous() and x.stride(1) == 1
            and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0 and k % 64 == 0 and n % 64 == 0
            and (not swiglu or (n // 2) % 32 == 0))


def mid_row_tile(m: int) -> int:
    return 64 if m <= 64 else 128 if m <= 128 else 256


def mid_gemm_candidates(m: int, n: int, k: int, swiglu: bool = False):
    """(bn, splits) pairs worth timing for this shape: both column-tile widths, with the K split
    that brings the grid closest to one or two workgroups per CU."""
    out = []
    ksteps = k // 64
    for bn in (128, 96, 64):
        if n % bn or (swiglu and (n // 2) % (bn // 2)) or (bn == 96 and m <= 128):
            continue
        tiles = n // bn
        seen = set()
        for target in (_NUM_CUS, _NUM_CUS // 2, 2 * _NUM_CUS):
            s = max(1, min(round(target / tiles), ksteps // 4 or 1, 16))
            if s not in seen:
                seen.add(s)
                out.append((bn, s))
        # short K (a tensor-parallel rank's o / down): one launch without a K split may beat two
        # (GEMM + slice sum) even though it leaves CUs idle
        if ksteps <= 32 and 1 not in seen:
            out.append((bn, 1))
    return out


# (row bucket, N, K, swiglu) -> (bn, splits, nt) or None (= library); filled by tune_mid_gemm
_MID_CHOICE: dict = {}
MID_ROW_BUCKETS = (32, 64, 128, 256)


def mid_row_bucket(m: int) -> int:
    return 32 if m <= 32 else 64 if m <= 64 else 128 if m <= 128 else 256


def mid_gemm_choice(x: torch.Tensor, weight: torch.Tensor, swiglu: bool = False,
                    to_norm: bool = False):
    """The tuned configuration for this call, or None when the library GEMM is to be used
    (unmeasured shapes included: the kernel only runs where it was seen to win).  ``to_norm``:
    the entry measured with the partial sums left to the norm kernel that follows (key
    ``"norm"`` in place of the SwiGLU flag), falling back to the plain entry."""
    if x.dim() != 2 or x.shape[0] <= SKINNY_MAX_ROWS or x.shape[0] > MID_MAX_ROWS:
        return None
    n, k = weight.shape
    b = mid_row_bucket(x.shape[0])
    key = (b, n, k, "norm") if to_norm and (b, n, k, "norm") in _MID_CHOICE else (b, n, k, swiglu)
    c = _MID_CHOICE.get(key)
    if c is None or not mid_gemm_ok(x, weight, swiglu):
        return None
    return c


def mid_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
               swiglu: bool = False, bn: int = 128, splits: int = 1, nt: bool = True,
               bm: Optional[int] = None, defer: bool = False):
    """``x @ weight.T (+ bias)`` (``swiglu``: ``silu(gate) * up`` of the fused weight) for up to
    256 rows with the weight-streaming MFMA kernel; ``splits`` > 1 slices K over workgroups
    (fp32 partial sums in a scratch buffer, summed by a second kernel).  ``defer`` (with a K
    split, no SwiGLU) returns the slices unsummed as ``SplitKPartials`` for a norm to consume."""
    M, K = x.shape
    N = weight.shape[0]
    if not mid_gemm_ok(x, weight, swiglu):
        raise ValueError(f"mid_linear cannot take x {tuple(x.shape)} / weight {tuple(weight.shape)}")
    bm = bm or mid_row_tile(M)
    if defer and not swiglu and splits > 1 and (eff := _mid_eff_splits(K, splits)) > 1:
        ws = torch.empty((eff, M, N), dtype=torch.float32, device=x.device)
        _check(_mid(x.data_ptr(), x.stride(0), weight.data_ptr(), None, None, 0, ws.data_ptr(),
                    M, N, K, 0, bm, bn, splits, int(nt), _dt(x), _stream()), "mid_gemm")
        return SplitKPartials(ws, bias, x.dtype)
    out = torch.empty((M, N // 2 if swiglu else N), dtype=x.dtype, device=x.device)
    ws = (torch.empty(splits * M * N, dtype=torch.float32, device=x.device)
          if splits > 1 else None)
    _check(_mid(x.data_ptr(), x.stride(0), weight.data_ptr(),
                bias.data_ptr() if bias is not None else None, out.data_ptr(), out.stride(0),
                ws.data_ptr() if ws is not None else None, M, N, K, int(swiglu), bm, bn, splits,
                int(nt), _dt(x), _stream()), "mid_gemm")
    return out


def splitk_materialize(part: SplitKPartials) -> torch.Tensor:
    """Sum of the K-slice partials (+ bias) as a plain [M, N] tensor."""
    S, M, N = part.ws.shape
    out = torch.empty((M, N), dtype=part.dtype, device=part.ws.device)
    _check(_mid_reduce(part.ws.data_ptr(), part.bias.data_ptr() if part.bias is not None
                       else None, out.data_ptr(), N, M, N, S, _DT[part.dtype], _stream()),
           "mid_reduce")
    return out


def splitk_add_rms_norm(part: SplitKPartials, residual: Optional[torch.Tensor],
                        weight: torch.Tensor, eps: float):
    """``(rms_norm(r) * weight, r)`` with ``r = sum of the slices (+ bias) + residual``;
    ``residual`` is updated in place and returned as ``r``."""
    if residual is None:
        r = splitk_materialize(part)
        return rms_norm(r, weight, eps), r
    S, M, N = part.ws.shape
    if (residual.shape != (M, N) or residual.dtype != part.dtype or residual.stride(1) != 1
            or residual.stride(0) % 8 or residual.data_ptr() % 16 or N % 8 or N > 32768):
        return add_rms_norm(splitk_materialize(part), residual, weight, eps)
    w = weight if weight.dtype == part.dtype else weight.to(part.dtype)
    out = torch.empty((M, N), dtype=part.dtype, device=part.ws.device)
    _check(_splitk_norm(part.ws.data_ptr(), part.bias.data_ptr() if part.bias is not None
                        else None, residual.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, S,
                        residual.stride(0), N, eps, _DT[part.dtype], _stream()),
           "splitk_add_rms_norm")
    return out, residual


def _graph_time_ms(fn, x, weights, reps: int) -> float:
    """Milliseconds per call of ``fn(x, w)`` cycling through ``weights`` (cold reads), measured
    on three replays of a captured graph (no launch gaps)."""
    for w in weights[:2]:
        fn(x, w)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(reps):
            for w in weights:
                fn(x, w)
    g.replay()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(3):
        g.replay()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / (3 * reps * len(weights))


def tune_mid_gemm(weights, swiglu: bool = False, rows=MID_ROW_BUCKETS,
                  margin: float = 0.97, to_norm: bool = False) -> dict:
    """For one projection (same-shaped matrices of several layers, cycled so that reads stay
    cold): time the library GEMM (+ activation kernel) against every candidate configuration
    of the mid-batch kernel at the top of each row bucket and record the winner.  ``to_norm``
    (projections whose output goes into residual add + RMSNorm): both sides are timed together
    with that norm, the kernel side with its K-slice sum folded into it, and the winner is
    recorded under the ``"norm"`` key.  Returns
    {rows: (library us, best kernel us or None, (bn, splits, nt) or None)}."""
    import torch.nn.functional as F
    weights = list(weights)
    n, k = weights[0].shape
    dtype, dev = weights[0].dtype, weights[0].device
    out = {}
    if not mid_gemm_ok(torch.zeros(64, k, dtype=dtype, device=dev), weights[0], swiglu) \
            or n * k * 2 < (1 << 20):
        return out
    assert not (to_norm and swiglu)
    reps = max(1, min(4, -(-(640 << 20) // (n * k * 2 * len(weights)))))
    gamma = torch.ones(n, dtype=dtype, device=dev)

    def lib(x, w):
        y = F.linear(x, w)
        if to_norm:
            return add_rms_norm(y, resid, gamma, 1e-6)
        return silu_and_mul(y) if swiglu else y

    for m in rows:
        x = torch.randn(m, k, dtype=dtype, device=dev)
        resid = torch.zeros(m, n, dtype=dtype, device=dev)
        t_lib = _graph_time_ms(lib, x, weights, reps)
        best, t_best = None, t_lib * margin
        for bn, sp in mid_gemm_candidates(m, n, k, swiglu):
            def ker(xx, w):
                y = mid_linear(xx, w, None, swiglu, bn, sp, True, defer=to_norm)
                if not to_norm:
                    return y
                if isinstance(y, SplitKPartials):
                    return y.add_rms_norm(resid, gamma, 1e-6)
                return add_rms_norm(y, resid, gamma, 1e-6)
            t = _graph_time_ms(ker, x, weights, reps)
            if t < t_best:
                best, t_best = (bn, sp, True), t
        _MID_CHOICE[(mid_row_bucket(m), n, k, "norm" if to_norm else swiglu)] = best
        out[m] = (round(t_lib * 1e3, 1), round(t_best * 1e3, 1) if best else None, best)
    return out


# -------------------------------------------------------------------- sampler
_LOGIT_DT = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}


def sample_filtered(logits: torch.Tensor, logit_rows: torch.Tensor, batch_rows: torch.Tensor,
                    temperatures: torch.Tensor, top_k: torch.Tensor, top_p: torch.Tensor,
                    seed: int, step, out: torch.Tensor, vocab: Optional[int] = None,
                    rounds: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Draw the listed rows again from their top-k / top-p restricted distributions (no sort:
    rejection sampling with a rising pivot, one workgroup per row, see csrc/sampler.hip) and
    write the tokens into ``out[batch_rows]`` (int32).  ``logit_rows[i]`` is the row of
    ``logits`` (full vocabulary, first ``vocab`` columns valid) that belongs to batch row
    ``batch_rows[i]``; ``temperatures`` / ``top_k`` (int32, <= 0 = off) / ``top_p`` (fp32,
    >= 1 = off) are indexed by batch row.  A row without a filter gets the token the plain
    sampler draws for (seed, step, batch row)."""
    assert logits.dim() == 2 and logits.stride(1) == 1 and out.dtype == torch.int32
    n = logit_rows.numel()
    assert batch_rows.numel() == n and logit_rows.dtype == batch_rows.dtype == torch.int32
    assert top_k.dtype == torch.int32 and top_p.dtype == torch.float32
    assert temperatures.dtype == torch.float32
    need = int(batch_rows.max().item()) + 1 if n and not torch.cuda.is_current_stream_capturing() else 0
    assert min(temperatures.numel(), top_k.numel(), top_p.numel(), out.numel()) >= need
    if not torch.is_tensor(step):
        step = torch.tensor([int(step) & 0x7fffffff], dtype=torch.int32, device=logits.device)
    _check(_sample_filtered(logits.data_ptr(), logits.stride(0), vocab or logits.shape[1], n,
                            logit_rows.data_ptr(), batch_rows.data_ptr(), temperatures.data_ptr(),
                            top_k.data_ptr(), top_p.data_ptr(), seed & 0xFFFFFFFFFFFFFFFF,
                            step.data_ptr(), out.data_ptr(),
                            rounds.data_ptr() if rounds is not None else None,
                            _LOGIT_DT[logits.dtype], _stream()), "sample_filtered")
    return out


def sample_local(logits: torch.Tensor, temperatures: torch.Tensor, seed: int, step,
                 vocab_offset: int = 0, valid_vocab: Optional[int] = None,
                 chunks: Optional[int] = None):
    """Best (score, global index) per row over this rank's vocabulary shard.
    ``step`` is an int or a 1-element int32 device tensor (graph-safe)."""
    assert logits.dim() == 2 and logits.stride(1) == 1
    B, V = logits.shape
    if valid_vocab is None:
        valid_vocab = vocab_offset + V
    if not torch.is_tensor(step):
        step = torch.tensor([int(step) & 0x7fffffff], dtype=torch.int32, device=logits.device)
    temps = temperatures if temperatures.dtype == torch.float32 else temperatures.float()
    vals = torch.empty(B, dtype=torch.float32, device=logits.device)
    idx = torch.empty(B, dtype=torch.int32, device=logits.device)
    # few rows: split each over several workgroups so that the whole chip takes part
    if chunks is None:
        chunks = max(1, min(16, 256 // max(B, 1))) if V >= 16384 else 1
    if chunks > 1:
        ws_val = torch.empty(B * chunks, dtype=torch.float32, device=logits.device)
        ws_idx = torch.empty(B * chunks, dtype=torch.int32, device=logits.device)
        wv, wi = ws_val.data_ptr(), ws_idx.data_ptr()
    else:
        wv = wi = None
    _check(_sample(logits.data_ptr(), logits.stride(0), B, V, vocab_offset, valid_vocab,
                   temps.data_ptr(), seed & 0xFFFFFFFFFFFFFFFF, step.data_ptr(), vals.data_ptr(),
                   idx.data_ptr(), _LOGIT_DT[logits.dtype], chunks, wv, wi, _stream()), "sample")
    return vals, idx


def sample(logits: torch.Tensor, temperatures: torch.Tensor, seed: int = 0, step=0) -> torch.Tensor:
    return sample_local(logits, temperatures, seed, step)[1].to(torch.int64)


# ------------------------------------------------------------ prefill-sized GEMM
BIG_MIN_ROWS = 257                    # below this the weight-streaming kernels / the library serve


def big_gemm_ok(x: torch.Tensor, weight: torch.Tensor, swiglu: bool = False) -> bool:
    """Operands the 256 x 256-tile kernel can take: 2-D 16-bit row-major x (row stride a multiple
    of 8 elements) and contiguous weights of the same dtype, K a multiple of 64 and >= 256, the
    output width (per SwiGLU half: N / 2) a multiple of 256 (128), every operand below 2 GiB (the
    kernel addresses them with 32-bit buffer offsets and refuses anything larger)."""
    if x.dim() != 2 or weight.dim() != 2 or x.dtype not in _DT or weight.dtype != x.dtype:
        return False
    M, K = x.shape
    N = weight.shape[0]
    if weight.shape[1] != K or not weight.is_contiguous() or x.stride(1) != 1 or x.stride(0) % 8:
        return False
    if K % 64 or K < 256 or M < 1:
        return False
    if swiglu:
        if N % 2 or (N // 2) % 128:
            return False
    elif N % 256:
        return False
    n_out = N // 2 if swiglu else N
    return (M * x.stride(0) * 2 < (1 << 31) and N * K * 2 < (1 << 31)
            and M * n_out * 2 < (1 << 31) and x.data_ptr() % 16 == 0)


# (row bucket, N, K, swiglu) -> True where the kernel was measured faster than the library path;
# filled by tune_big_gemm at start-up.  Unmeasured shapes go to the library.
_BIG_CHOICE: dict = {}
BIG_ROW_BUCKETS = (512, 1024, 2048, 4096, 8192, 16384)


def big_row_bucket(m: int) -> int:
    for b in BIG_ROW_BUCKETS:
        if m <= b:
            return b
    return BIG_ROW_BUCKETS[-1]


def big_gemm_pays(x: torch.Tensor, weight: torch.Tensor, swiglu: bool = False) -> bool:
    """True when this call should run on the 256 x 256-tile kernel: more than 256 rows, operands
    the kernel takes, and a start-up measurement that preferred it for this shape."""
    if x.dim() != 2 or x.shape[0] < BIG_MIN_ROWS:
        return False
    n, k = weight.shape
    return bool(_BIG_CHOICE.get((big_row_bucket(x.shape[0]), n, k, swiglu))) \
        and big_gemm_ok(x, weight, swiglu)


def tune_big_gemm(weights, swiglu: bool = False, rows=BIG_ROW_BUCKETS, margin: float = 0.98) -> dict:
    """For one projection (same-shaped matrices of a few layers): time the library GEMM
    (+ activation kernel with ``swiglu``) against the 256 x 256-tile kernel at each row count
    and record the winner under the row bucket.  Returns {rows: (library us, kernel us, use
    kernel)}."""
    import torch.nn.functional as F
    weights = list(weights)[:4]
    n, k = weights[0].shape
    dtype, dev = weights[0].dtype, weights[0].device
    out = {}
    for m in rows:
        x = torch.randn(m, k, dtype=dtype, device=dev)
        if not big_gemm_ok(x, weights[0], swiglu):
            return out
        lib = (lambda xx, w: silu_and_mul(F.linear(xx, w))) if swiglu else (lambda xx, w: F.linear(xx, w))
        ker = lambda xx, w: big_linear(xx, w, None, swiglu)
        reps = max(1, min(8, int(2e11 / (2.0 * m * n * k))))
        t_lib = _graph_time_ms(lib, x, weights, reps)
        try:
            t_ker = _graph_time_ms(ker, x, weights, reps)
        except (RuntimeError, ValueError):        # the kernel refuses the operands: library
            t_ker = float("inf")
        use = t_ker < t_lib * margin
        _BIG_CHOICE[(big_row_bucket(m), n, k, swiglu)] = use
        out[m] = (round(t_lib * 1e3, 1), round(t_ker * 1e3, 1), use)
        del x
    return out


def big_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
               swiglu: bool = False, out: Optional[torch.Tensor] = None,
               row_scale: bool = False, eps: float = 0.0) -> torch.Tensor:
    """``x @ weight.T (+ bias)``, or with ``swiglu`` ``silu(x @ Wg.T) * (x @ Wu.T)`` of the fused
    ``[Wg; Wu]`` weight, on the compute-bound 256 x 256-tile MFMA kernel (any row count; built
    for prefill steps).  ``row_scale``: the rows of the product are multiplied by
    ``rsqrt(mean(x[row]^2) + eps)`` before bias / SwiGLU, i.e. ``x`` is RMS-normalised behind the
    GEMM (for a ``weight`` that carries the norm's gamma); the kernel sums the squares from the
    x fragments it loads anyway."""
    if not big_gemm_ok(x, weight, swiglu):
        raise ValueError(f"big_linear cannot take x {tuple(x.shape)} / weight {tuple(weight.shape)}")
    M, K = x.shape
    N = weight.shape[0]
    n_out = N // 2 if swiglu else N
    if out is None:
        out = torch.empty((M, n_out), dtype=x.dtype, device=x.device)
    assert out.shape == (M, n_out) and out.stride(1) == 1 and out.stride(0) % 8 == 0 and out.dtype == x.dtype
    if bias is not None:
        assert bias.dtype == x.dtype and bias.numel() == N and bias.is_contiguous()
    _check(_big(x.data_ptr(), x.stride(0), weight.data_ptr(),
                bias.data_ptr() if bias is not None else None, out.data_ptr(), out.stride(0),
                M, N, K, int(swiglu), _dt(x), int(row_scale), float(eps), 0, _stream()), "big_gemm")
    return out


def big_resid_ok(x: torch.Tensor, weight: torch.Tensor, residual) -> bool:
    """Operands of ``big_linear_resid``: a plain ``big_gemm_ok`` projection onto a contiguous,
    16-byte aligned [M, N] residual stream of the same dtype."""
    return (torch.is_tensor(residual) and big_gemm_ok(x, weight) and residual.dim() == 2
            and tuple(residual.shape) == (x.shape[0], weight.shape[0]) and residual.dtype == x.dtype
            and residual.stride(1) == 1 and residual.stride(0) % 8 == 0
            and residual.data_ptr() % 16 == 0
            and residual.shape[0] * residual.stride(0) * 2 < (1 << 31))


def big_linear_resid(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                     residual: torch.Tensor) -> ResidualSum:
    """``residual <- round(round(x @ weight.T + bias) + residual)`` in place (the roundings of a
    GEMM followed by the fused add + norm kernel).  The RMSNorm that follows is left to the next
    projection (``big_linear(row_scale=True)``), or to a plain norm kernel."""
    if not big_resid_ok(x, weight, residual):
        raise ValueError("big_linear_resid cannot take these operands")
    M, K = x.shape
    N = weight.shape[0]
    _check(_big(x.data_ptr(), x.stride(0), weight.data_ptr(),
                bias.data_ptr() if bias is not None else None, residual.data_ptr(),
                residual.stride(0), M, N, K, 0, _dt(x), 0, 0.0, 1, _stream()),
           "big_gemm (residual epilogue)")
    return ResidualSum(residual)


def big_gemm_resid_pays(x: torch.Tensor, weight: torch.Tensor, residual) -> bool:
    """True when a projection that feeds (residual add + RMSNorm) should run with the residual
    epilogue: prefill-sized, operands the kernel takes, and a start-up measurement of the whole
    chain (projection, norm, next projection) that preferred the fused form."""
    if x.dim() != 2 or x.shape[0] < BIG_MIN_ROWS:
        return False
    n, k = weight.shape
    return bool(_BIG_CHOICE.get((big_row_bucket(x.shape[0]), n, k, "resid"))) \
        and big_resid_ok(x, weight, residual)


def tune_big_resid(prod_weights, cons_weights, cons_swiglu: bool, rows=BIG_ROW_BUCKETS,
                   eps: float = 1e-6, margin: float = 0.98) -> dict:
    """For a projection that feeds (residual add + RMSNorm) and the projection behind that norm
    (o -> gate_up, down -> qkv; ``cons_weights`` carry the norm's gamma): time the chain as it runs
    unfused (best plain projection, add + norm kernel, best next projection) against the fused
    chain (residual epilogue, row-scaled next projection) at each row count and record the winner
    under ``"resid"``.  Returns {rows: (unfused us, fused us, use fused)}."""
    import torch.nn.functional as F
    from . import linear as _linear, swiglu_linear as _swiglu_linear
    pairs = list(zip(list(prod_weights)[:4], list(cons_weights)[:4]))
    n, k = pairs[0][0].shape
    nc = pairs[0][1].shape[0]
    dtype, dev = pairs[0][0].dtype, pairs[0][0].device
    gamma = torch.ones(n, dtype=dtype, device=dev)
    out = {}
    for m in rows:
        x = torch.randn(m, k, dtype=dtype, device=dev)
        resid = torch.zeros(m, n, dtype=dtype, device=dev)
        if not big_resid_ok(x, pairs[0][0], resid) or not big_gemm_ok(resid, pairs[0][1], cons_swiglu):
            return out

        def unfused(xx, pair):
            h, r = add_rms_norm(_linear(xx, pair[0]), resid, gamma, eps)
            return _swiglu_linear(h, pair[1]) if cons_swiglu else _linear(h, pair[1])

        def fused(xx, pair):
            rs = big_linear_resid(xx, pair[0], None, resid)
            return big_linear(rs.r, pair[1], None, cons_swiglu, row_scale=True, eps=eps)
        reps = max(1, min(8, int(2e11 / (2.0 * m * (n * k + nc * n)))))
        t_a = _graph_time_ms(unfused, x, pairs, reps)
        resid.zero_()
        t_b = _graph_time_ms(fused, x, pairs, reps)
        use = t_b < t_a * margin
        _BIG_CHOICE[(big_row_bucket(m), n, k, "resid")] = use
        out[m] = (round(t_a * 1e3, 1), round(t_b * 1e3, 1), use)
        del x, resid
    return out


# ------------------------------------------------- kernel-choice tables: export / import
# Which kernel runs for a (row bucket, N, K, epilogue) is decided by a timing at start-up; two
# starts (or two TP ranks) timing for themselves can disagree inside the measurement noise and
# then round differently.  The tables are therefore values that can be saved, shipped to the
# other ranks and loaded instead of measuring again.
def export_gemm_choices() -> dict:
    """The three choice tables as JSON-serialisable lists of [rows, N, K, tag, value]."""
    def rows(table):
        return [[k[0], k[1], k[2], k[3], list(v) if isinstance(v, tuple) else v]
                for k, v in sorted(table.items(), key=lambda kv: tuple(map(str, kv[0])))]
    return {"skinny": rows(_SKINNY_CHOICE), "mid": rows(_MID_CHOICE), "big": rows(_BIG_CHOICE)}


def import_gemm_choices(tables: dict, replace: bool = True) -> None:
    """Install tables produced by ``export_gemm_choices`` (``replace``: drop what is there)."""
    for name, table in (("skinny", _SKINNY_CHOICE), ("mid", _MID_CHOICE), ("big", _BIG_CHOICE)):
        if replace:
            table.clear()
        for m, n, k, tag, v in tables.get(name, ()):
            table[(int(m), int(n), int(k), tag)] = tuple(v) if isinstance(v, list) else v


def kernel_library_stamp() -> str:
    """Identifies the built kernel library (a choice made for one build says nothing about
    another)."""
    st = os.stat(_LIB_PATH)
    return f"{st.st_size}-{int(st.st_mtime)}"
