"""Numerics of every HIP kernel against the plain-PyTorch fp32 oracle of the same op
(``minivllm_amd.ops.torch_ref``).  Needs a real MI355X."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from minivllm_amd.ops import hip as h      # raises loudly when the .so is missing
    return h


from minivllm_amd.ops import torch_ref as ref  # noqa: E402

DEV = "cuda"
TOL = {torch.bfloat16: dict(atol=2e-2, rtol=2e-2), torch.float16: dict(atol=4e-3, rtol=4e-3)}


def close(a, b, dtype, scale=1.0):
    t = TOL[dtype]
    torch.testing.assert_close(a.float(), b.float(), atol=t["atol"] * scale, rtol=t["rtol"])


# ------------------------------------------------------------------ elementwise
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,hidden", [(1, 1024), (7, 4096), (33, 5120), (5, 128), (3, 8192), (2, 64)])
def test_rms_norm(hip, dtype, rows, hidden):
    torch.manual_seed(0)
    x = torch.randn(rows, hidden, device=DEV, dtype=dtype)
    w = (1 + 0.1 * torch.randn(hidden, device=DEV)).to(dtype)
    close(hip.rms_norm(x, w, 1e-6), ref.rms_norm(x, w, 1e-6), dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,hidden", [(1, 1024), (19, 4096), (4, 5120)])
def test_add_rms_norm(hip, dtype, rows, hidden):
    torch.manual_seed(1)
    x = torch.randn(rows, hidden, device=DEV, dtype=dtype)
    r = torch.randn(rows, hidden, device=DEV, dtype=dtype)
    w = (1 + 0.1 * torch.randn(hidden, device=DEV)).to(dtype)
    want_o, want_r = ref.add_rms_norm(x, r, w, 1e-6)
    got_o, got_r = hip.add_rms_norm(x.clone(), r.clone(), w, 1e-6)
    torch.testing.assert_close(got_r.float(), want_r.float(), atol=0, rtol=0)
    close(got_o, want_o, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,inter", [(1, 3072), (17, 12288), (400, 800), (3, 8)])
def test_silu_and_mul(hip, dtype, rows, inter):
    torch.manual_seed(2)
    x = torch.randn(rows, 2 * inter, device=DEV, dtype=dtype)
    close(hip.silu_and_mul(x), ref.silu_and_mul(x), dtype)


@pytest.mark.parametrize("shape", [(400, 800), (3, 4000, 8000), (5, 8)])
def test_silu_and_mul_fp32(hip, shape):
    """fp32 rows: the precision of the reference's activation micro-benchmark."""
    torch.manual_seed(3)
    x = torch.randn(*shape, device=DEV, dtype=torch.float32)
    got = hip.silu_and_mul(x)
    assert got.dtype == torch.float32 and got.shape == (*shape[:-1], shape[-1] // 2)
    torch.testing.assert_close(got, ref.silu_and_mul(x), atol=2e-6, rtol=2e-6)


def _logical_cache(nb, bs, hkv, d, dtype, head_major=True):
    """[nb, bs, hkv, d] logical view; physical order head-major like the engine's."""
    if head_major:
        return torch.zeros(nb, hkv, bs, d, device=DEV, dtype=dtype).permute(0, 2, 1, 3)
    return torch.zeros(nb, bs, hkv, d, device=DEV, dtype=dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("with_norm", [True, False])
def test_qk_norm_rope_store(hip, dtype, d, with_norm):
    torch.manual_seed(3)
    T, hq, hkv, bs, nb = 37, 8, 2, 16, 8
    qkv = torch.randn(T, (hq + 2 * hkv) * d, device=DEV, dtype=dtype)
    pos = torch.randint(0, 500, (T,), device=DEV, dtype=torch.int32)
    cs = ref.build_cos_sin_cache(d, 512, 10000.0, DEV)
    qw = (1 + 0.1 * torch.randn(d, device=DEV)).to(dtype) if with_norm else None
    kw = (1 + 0.1 * torch.randn(d, device=DEV)).to(dtype) if with_norm else None
    slots = torch.randperm(nb * bs, device=DEV)[:T].to(torch.int32)
    slots[5] = -1
    kc_r, vc_r = _logical_cache(nb, bs, hkv, d, dtype), _logical_cache(nb, bs, hkv, d, dtype)
    kc_h, vc_h = _logical_cache(nb, bs, hkv, d, dtype), _logical_cache(nb, bs, hkv, d, dtype)
    q_r, k_r, v_r = ref.qk_norm_rope_store(qkv.clone(), pos, cs, qw, kw, 1e-6, hq, hkv, d, kc_r, vc_r, slots)
    q_h, k_h, v_h = hip.qk_norm_rope_store(qkv.clone(), pos, cs, qw, kw, 1e-6, hq, hkv, d, kc_h, vc_h, slots)
    close(q_h, q_r, dtype, 2)
    close(k_h, k_r, dtype, 2)
    close(v_h, v_r, dtype)
    close(kc_h, kc_r, dtype, 2)
    close(vc_h, vc_r, dtype)
    # prefill form: Q rows untouched, K transformed into the cache only (qkv keeps the raw K rows)
    kc_p, vc_p = _logical_cache(nb, bs, hkv, d, dtype), _logical_cache(nb, bs, hkv, d, dtype)
    raw = qkv.clone()
    hip.qk_norm_rope_store(raw, pos, cs, qw, kw, 1e-6, hq, hkv, d, kc_p, vc_p, slots, skip_q=True,
                           k_in_place=False)
    assert torch.equal(raw, qkv)
    assert torch.equal(kc_p, kc_h) and torch.equal(vc_p, vc_h)


@pytest.mark.parametrize("head_major", [True, False])
def test_store_kvcache(hip, head_major):
    torch.manual_seed(4)
    dtype = torch.bfloat16
    T, hkv, d, bs, nb = 50, 4, 128, 16, 6
    k = torch.randn(T, hkv, d, device=DEV, dtype=dtype)
    v = torch.randn(T, hkv, d, device=DEV, dtype=dtype)
    slots = torch.randperm(nb * bs, device=DEV)[:T]          # int64 like the reference passes
    slots[0] = -1
    kc_r, vc_r = _logical_cache(nb, bs, hkv, d, dtype, head_major), _logical_cache(nb, bs, hkv, d, dtype, head_major)
    kc_h, vc_h = _logical_cache(nb, bs, hkv, d, dtype, head_major), _logical_cache(nb, bs, hkv, d, dtype, head_major)
    ref.store_kvcache(k, v, kc_r, vc_r, slots)
    hip.store_kvcache(k, v, kc_h, vc_h, slots)
    torch.testing.assert_close(kc_h, kc_r, atol=0, rtol=0)
    torch.testing.assert_close(vc_h, vc_r, atol=0, rtol=0)


# --------------------------------------------------------------------- sampler
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_sampler_greedy(hip, dtype):
    torch.manual_seed(5)
    logits = torch.randn(9, 151936, device=DEV, dtype=dtype)
    t = torch.zeros(9, device=DEV)
    got = hip.sample(logits, t, seed=1, step=3)
    assert got.tolist() == logits.float().argmax(-1).tolist()


def test_sampler_distribution(hip):
    """Empirical frequencies of the exponential-race sampler match softmax(logits / T)."""
    torch.manual_seed(6)
    V, n, T = 16, 20000, 0.7
    base = torch.randn(V, device=DEV)
    logits = base.repeat(n, 1).to(torch.float32)
    temps = torch.full((n,), T, device=DEV)
    got = hip.sample(logits, temps, seed=123, step=7)
    freq = torch.bincount(got, minlength=V).float() / n
    want = torch.softmax(base / T, -1)
    assert (freq - want).abs().max().item() < 0.015
    # a different step gives a different draw, same step the same draw
    again = hip.sample(logits, temps, seed=123, step=7)
    other = hip.sample(logits, temps, seed=123, step=8)
    assert torch.equal(got, again) and not torch.equal(got, other)


def test_sampler_vocab_shards_agree(hip):
    """Sampling shard-wise and combining (value, index) pairs equals sampling the full row."""
    torch.manual_seed(7)
    B, V = 6, 4096
    logits = torch.randn(B, V, device=DEV, dtype=torch.bfloat16)
    temps = torch.tensor([0.0, 0.5, 1.0, 0.0, 2.0, 0.8], device=DEV)
    full_v, full_i = hip.sample_local(logits, temps, 9, 4)
    parts = [hip.sample_local(logits[:, s:s + 1024].contiguous(), temps, 9, 4, vocab_offset=s,
                              valid_vocab=V) for s in range(0, V, 1024)]
    vals = torch.stack([p[0] for p in parts])
    idx = torch.stack([p[1] for p in parts])
    best = vals.argmax(0)
    assert torch.equal(idx.gather(0, best[None]).squeeze(0), full_i)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("V", [151936, 16390, 40000])
def test_sampler_split_rows_equal_single_workgroup_rows(hip, dtype, V):
    """Small batches split every row over several workgroups and merge the partial winners:
    same scores, same tokens as one workgroup per row (noise is keyed by vocabulary index)."""
    torch.manual_seed(8)
    logits = torch.randn(3, V + 6, device=DEV, dtype=dtype)[:, :V]        # row stride != V
    temps = torch.tensor([0.0, 0.7, 1.3], device=DEV)
    one_v, one_i = hip.sample_local(logits, temps, 5, 11, valid_vocab=V - 3, chunks=1)
    for chunks in (2, 7, 16):
        v, i = hip.sample_local(logits, temps, 5, 11, valid_vocab=V - 3, chunks=chunks)
        assert torch.equal(i, one_i) and torch.equal(v, one_v)
    auto_v, auto_i = hip.sample_local(logits, temps, 5, 11, valid_vocab=V - 3)
    assert torch.equal(auto_i, one_i)
    assert one_i[0].item() == logits[0, :V - 3].float().argmax().item()


def _kept_mask(logits, temps, top_k, top_p):
    """The kept set of the filtered sampler, row by row, from a sort (fp64): token i stays iff
    fewer than k tokens are strictly more likely and those hold less than p of the mass."""
    sc = logits.double() / temps.double()[:, None]
    prob = torch.softmax(sc, -1)
    vals, order = sc.sort(-1, descending=True)
    ps = prob.gather(1, order)
    mass_before = ps.cumsum(-1) - ps
    # tokens tied with their predecessor share its count / mass of strictly greater tokens
    first = torch.ones_like(vals, dtype=torch.bool)
    first[:, 1:] = vals[:, 1:] != vals[:, :-1]
    pos = torch.arange(vals.shape[1], device=vals.device).expand_as(vals)
    start = torch.where(first, pos, torch.zeros_like(pos)).cummax(-1).values
    cnt_gt = start
    mass_gt = mass_before.gather(1, start)
    keep_sorted = ((top_k[:, None] <= 0) | (cnt_gt < top_k[:, None])) & (mass_gt < top_p[:, None].double())
    keep = torch.zeros_like(keep_sorted)
    keep.scatter_(1, order, keep_sorted)
    return keep, prob


@pytest.mark.parametrize("dtype,V", [(torch.bfloat16, 151936), (torch.float32, 40003), (torch.float16, 8192)])
def test_filtered_sampler_membership_and_plain_rows(hip, dtype, V):
    """Every drawn token lies in its row's kept set; rows without a filter and greedy rows get
    exactly the plain sampler's token; k = 1 and a tiny p give a maximal token; same call, same
    tokens.  Rows are listed in a shuffled order and read from a strided logits buffer."""
    torch.manual_seed(31)
    B = 24
    logits = (torch.randn(B, V + 8, device=DEV) * 3).to(dtype)[:, :V]
    temps = torch.full((B,), 0.9, device=DEV)
    temps[5] = 0.0
    top_k = torch.tensor([0, 1, 5, 50, 0, 3, 0, 1000] * 3, device=DEV, dtype=torch.int32)
    top_p = torch.tensor([1.0, 1.0, 1.0, 0.9, 0.5, 0.9, 1e-6, 0.999] * 3, device=DEV)
    plain = hip.sample(logits, temps, seed=77, step=5).to(torch.int32)
    rows = torch.randperm(B, device=DEV).to(torch.int32)
    out = torch.full((B,), -1, device=DEV, dtype=torch.int32)
    rounds = torch.zeros(B, device=DEV, dtype=torch.int32)
    hip.sample_filtered(logits, rows, rows, temps, top_k, top_p, 77, 5, out, rounds=rounds)
    keep, _ = _kept_mask(logits.float(), temps.clamp_min(1e-3), top_k, top_p)
    assert bool(keep.gather(1, out.long()[:, None]).all())
    top = logits.float().max(-1).values
    for r in range(B):
        if top_k[r] <= 0 and top_p[r] >= 1.0 or temps[r] == 0:
            assert out[r] == plain[r] and rounds[r] == 0
        if top_k[r] == 1 or top_p[r] < 1e-5:
            assert logits[r, out[r]].float() == top[r]        # (16-bit rows: the maximum may be tied)
    assert int(rounds.max()) < 96
    again = torch.full((B,), -1, device=DEV, dtype=torch.int32)
    hip.sample_filtered(logits, rows, rows, temps, top_k, top_p, 77, 5, again)
    assert torch.equal(out, again)
    # a subset of the rows, read from a gathered copy: the other outputs stay untouched
    sub = torch.tensor([7, 2, 19], device=DEV, dtype=torch.int32)
    part = torch.full((B,), -1, device=DEV, dtype=torch.int32)
    hip.sample_filtered(logits[sub.long()].contiguous(), torch.arange(3, device=DEV, dtype=torch.int32),
                        sub, temps, top_k, top_p, 77, 5, part)
    assert torch.equal(part[sub.long()], out[sub.long()]) and int((part >= 0).sum()) == 3


@pytest.mark.parametrize("k,p", [(4, 1.0), (0, 0.8), (6, 0.7), (0, 0.999)])
def test_filtered_sampler_distribution(hip, k, p):
    """Frequencies over 30000 independent rows with the same logits match the renormalised
    restricted distribution (flat-ish row: several rejection rounds are exercised)."""
    torch.manual_seed(32)
    V, n, T = 24, 30000, 1.1
    base = torch.randn(V, device=DEV) * 1.5
    logits = base.repeat(n, 1)
    temps = torch.full((n,), T, device=DEV)
    top_k = torch.full((n,), k, device=DEV, dtype=torch.int32)
    top_p = torch.full((n,), p, device=DEV)
    rows = torch.arange(n, device=DEV, dtype=torch.int32)
    out = torch.empty(n, device=DEV, dtype=torch.int32)
    hip.sample_filtered(logits, rows, rows, temps, top_k, top_p, 5, 9, out)
    keep, prob = _kept_mask(base[None], temps[:1], top_k[:1], top_p[:1])
    want = (prob * keep).float()[0]
    want = want / want.sum()
    freq = torch.bincount(out.long(), minlength=V).float() / n
    assert (freq - want).abs().max().item() < 0.012
    assert freq[~keep[0]].sum().item() == 0


# ------------------------------------------------------------------- attention
def _fill_paged(k, v, lens, bs, hkv, d, dtype, head_major=True, shuffle=True):
    """Scatter packed k/v [sum(lens), hkv, d] into a paged cache; returns caches + tables."""
    nblk = [(L + bs - 1) // bs for L in lens]
    total = sum(nblk) + 3
    perm = torch.randperm(total) if shuffle else torch.arange(total)
    kc, vc = _logical_cache(total, bs, hkv, d, dtype, head_major), _logical_cache(total, bs, hkv, d, dtype, head_major)
    tables = torch.full((len(lens), max(nblk)), -1, dtype=torch.int32)
    off = used = 0
    for i, L in enumerate(lens):
        ids = perm[used:used + nblk[i]]
        used += nblk[i]
        tables[i, :nblk[i]] = ids.to(torch.int32)
        for j, blk in enumerate(ids.tolist()):
            n = min(bs, L - j * bs)
            kc[blk, :n] = k[off + j * bs: off + j * bs + n]
            vc[blk, :n] = v[off + j * bs: off + j * bs + n]
        off += L
    return kc, vc, tables.to(DEV)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hq,hkv,d", [(32, 8, 128), (16, 8, 128), (8, 1, 128), (16, 8, 64), (4, 4, 128)])
@pytest.mark.parametrize("bs", [16, 256])
def test_paged_decode(hip, dtype, hq, hkv, d, bs):
    torch.manual_seed(8)
    lens = [1, 31, 32, 33, 260, 1000, 77]
    B = len(lens)
    k = torch.randn(sum(lens), hkv, d, device=DEV, dtype=dtype)
    v = torch.randn(sum(lens), hkv, d, device=DEV, dtype=dtype)
    kc, vc, tables = _fill_paged(k, v, lens, bs, hkv, d, dtype)
    q = torch.randn(B, hq, d, device=DEV, dtype=dtype)
    ctx = torch.tensor(lens, device=DEV, dtype=torch.int32)
    scale = 1.0 / math.sqrt(d)
    want = ref.paged_attention_decode(q, kc, vc, tables, ctx, scale)
    for ns in (1, 3, 8):
        got = hip.paged_attention_decode(q, kc, vc, tables, ctx, scale, num_splits=ns)
        close(got, want, dtype)
    got = hip.paged_attention_decode(q, kc, vc, tables, ctx, scale, max_context_len=max(lens))
    close(got, want, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hq,hkv,d", [(32, 8, 128), (16, 8, 128), (8, 1, 128), (16, 8, 64), (4, 4, 128),
                                      (40, 8, 128), (64, 4, 128), (40, 2, 128)])
@pytest.mark.parametrize("bs", [16, 256])
@pytest.mark.parametrize("with_norm", [True, False])
def test_paged_decode_with_fused_norm_rope_store(hip, dtype, hq, hkv, d, bs, with_norm):
    """Decode attention handed the raw q / k / v projections: against the stand-alone
    norm + rotate + store kernel followed by the plain decode kernel, with unused batch
    rows (length 0), contexts of one token and contexts that end on, before and behind a 32-token
    tile or a cache block; the slot of the new token holds NaN beforehand."""
    torch.manual_seed(80 + hq)
    lens = [1, 31, 32, 33, 0, 260, 1000, 77, 256, 257, 0, 64]
    B = len(lens)
    eps, scale = 1e-6, 1.0 / math.sqrt(d)
    # the cache holds the first L - 1 tokens; the last token's slot is poisoned
    old = [max(L - 1, 0) for L in lens]
    k = torch.randn(sum(old), hkv, d, device=DEV, dtype=dtype)
    v = torch.randn(sum(old), hkv, d, device=DEV, dtype=dtype)
    nblk = [(max(L, 1) + bs - 1) // bs for L in lens]
    perm = torch.randperm(sum(nblk) + 3)
    kc, vc = _logical_cache(sum(nblk) + 3, bs, hkv, d, dtype), _logical_cache(sum(nblk) + 3, bs, hkv, d, dtype)
    tables = torch.full((B, max(nblk)), -1, dtype=torch.int32)
    used = 0
    for i in range(B):
        tables[i, :nblk[i]] = perm[used:used + nblk[i]].to(torch.int32)
        used += nblk[i]
    tables = tables.to(DEV)
    off = 0
    slots = []
    for i, L in enumerate(lens):
        for t in range(max(L - 1, 0)):
            blk = int(tables[i, t // bs])
            kc[blk, t % bs] = k[off + t]
            vc[blk, t % bs] = v[off + t]
        off += max(L - 1, 0)
        if L > 0:
            blk = int(tables[i, (L - 1) // bs])
            kc[blk, (L - 1) % bs] = float("nan")
            vc[blk, (L - 1) % bs] = float("nan")
            slots.append(blk * bs + (L - 1) % bs)
        else:
            slots.append(-1)
    qkv = torch.randn(B, (hq + 2 * hkv) * d, device=DEV, dtype=dtype)
    qw = (1 + 0.1 * torch.randn(d, device=DEV)).to(dtype) if with_norm else None
    kw = (1 + 0.1 * torch.randn(d, device=DEV)).to(dtype) if with_norm else None
    cs = ref.build_cos_sin_cache(d, 2048, 10000.0, DEV)
    ctx = torch.tensor(lens, device=DEV, dtype=torch.int32)
    pos = (ctx - 1).clamp(min=0)
    slot_t = torch.tensor(slots, device=DEV, dtype=torch.int32)
    # unfused: store kernel (in place on a copy), then the plain decode kernel
    kc1, vc1, qkv1 = kc.clone(), vc.clone(), qkv.clone()
    # (clone() of the permuted view keeps the logical shape; make it head-major again)
    kc1 = kc1.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
    vc1 = vc1.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
    q1, _, _ = hip.qk_norm_rope_store(qkv1, pos, cs, qw, kw, eps, hq, hkv, d, kc1, vc1, slot_t)
    live = torch.tensor([L > 0 for L in lens], device=DEV)
    for ns in (1, 4):
        kc2 = kc.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
        vc2 = vc.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
        want = hip.paged_attention_decode(q1, kc1, vc1, tables, ctx, scale, num_splits=ns)
        q = qkv[:, :hq * d].view(B, hq, d)
        kn = qkv[:, hq * d:(hq + hkv) * d].view(B, hkv, d)
        vn = qkv[:, (hq + hkv) * d:].view(B, hkv, d)
        got = hip.paged_attention_decode(q, kc2, vc2, tables, ctx, scale, num_splits=ns,
                                         new_kv=(kn, vn, cs, qw, kw, eps))
        # every slot, touched or not; keys to one rounding step (the two kernels sum the squares
        # of a row in different orders)
        assert torch.equal(vc2, vc1) and not torch.isnan(kc2).any()
        ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
        assert float(((kc2.float() - kc1.float()).abs() / kc1.float().abs().clamp(min=0.25)).max()) <= ulp
        assert not torch.isnan(got[live]).any()
        close(got[live], want[live].float(), dtype)
    # and against the fp32 oracle on the transformed inputs
    oracle = ref.paged_attention_decode(q1[live], kc1, vc1, tables[live], ctx[live], scale)
    close(got[live], oracle, dtype)


def test_paged_decode_split_merge_tickets(hip):
    """The context splits are merged by the workgroup that finishes last (a ticket per
    (sequence, head group)).  Many back-to-back launches with different split counts and batch
    sizes, eager and as replays of a captured graph: same bits every time, tickets back at zero."""
    torch.manual_seed(81)
    dtype, hq, hkv, d, bs = torch.bfloat16, 32, 8, 128, 256
    lens = [1000, 31, 513, 2048, 77, 1, 300, 1500, 64]
    k = torch.randn(sum(lens), hkv, d, device=DEV, dtype=dtype)
    v = torch.randn(sum(lens), hkv, d, device=DEV, dtype=dtype)
    kc, vc, tables = _fill_paged(k, v, lens, bs, hkv, d, dtype)
    q = torch.randn(len(lens), hq, d, device=DEV, dtype=dtype)
    ctx = torch.tensor(lens, device=DEV, dtype=torch.int32)
    scale = 1.0 / math.sqrt(d)
    want = ref.paged_attention_decode(q, kc, vc, tables, ctx, scale)
    first = {}
    for rep in range(30):
        for ns in (2, 7, 16, 32, 64):
            for nb in (len(lens), 3, 1):
                got = hip.paged_attention_decode(q[:nb], kc, vc, tables[:nb], ctx[:nb], scale,
                                                 num_splits=ns)
                key = (ns, nb)
                if key not in first:
                    close(got, want[:nb], dtype)
                    first[key] = got
                else:
                    assert torch.equal(got, first[key]), (rep, key)
    torch.cuda.synchronize()
    tickets = hip.decode_ticket_words(q.device, 1)
    assert int(tickets.abs().sum()) == 0
    out = torch.empty_like(q)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4):                                   # several layers' worth per replay
            hip.paged_attention_decode(q, kc, vc, tables, ctx, scale, num_splits=16, out=out)
    for _ in range(20):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, first[(16, len(lens))])
    assert int(tickets.abs().sum()) == 0
    own = torch.zeros(len(lens) * hkv, dtype=torch.int32, device=DEV)      # caller-owned tickets
    got = hip.paged_attention_decode(q, kc, vc, tables, ctx, scale, num_splits=16, tickets=own)
    assert torch.equal(got, first[(16, len(lens))]) and int(own.abs().sum()) == 0


def test_paged_decode_reference_layout_and_padding(hip):
    """Token-major cache (the layout the reference benchmark builds), strided q, and rows with
    context 0 (graph padding) which must come out as zeros without touching the table."""
    torch.manual_seed(9)
    dtype, hq, hkv, d, bs = torch.float16, 32, 8, 128, 16
    lens = [60, 512, 0, 5]
    fill = [60, 512, 1, 5]
    k = torch.randn(sum(fill), hkv, d, device=DEV, dtype=dtype)
    v = torch.randn(sum(fill), hkv, d, device=DEV, dtype=dtype)
    kc, vc, tables = _fill_paged(k, v, fill, bs, hkv, d, dtype, head_major=False)
    tables[2] = -1
    qkv = torch.randn(len(lens), (hq + 2 * hkv) * d, device=DEV, dtype=dtype)
    q = qkv[:, :hq * d].view(len(lens), hq, d)
    ctx = torch.tensor(lens, device=DEV, dtype=torch.int32)
    want = ref.paged_attention_decode(q, kc, vc, tables, ctx, 0.09)
    got = hip.paged_attention_decode(q, kc, vc, tables, ctx, 0.09, num_splits=2)
    close(got, want, dtype)
    assert got[2].abs().max().item() == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hq,hkv,d", [(32, 8, 128), (4, 2, 128), (8, 1, 128), (16, 8, 64)])
def test_flash_prefill_packed(hip, dtype, hq, hkv, d):
    torch.manual_seed(10)
    lens = [1, 60, 64, 129, 300, 17]
    T = sum(lens)
    q = torch.randn(T, hq, d, device=DEV, dtype=dtype)
    k = torch.randn(T, hkv, d, device=DEV, dtype=dtype)
    v = torch.randn(T, hkv, d, device=DEV, dtype=dtype)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), device=DEV, dtype=torch.int32)
    scale = 1.0 / math.sqrt(d)
    want = ref.flash_attention_prefill(q, k, v, cu, scale)
    got = hip.flash_attention_prefill(q, k, v, cu, scale)
    close(got, want, dtype)


def test_flash_prefill_strided_qkv(hip):
    """q/k/v as column slices of one fused projection output (how the model calls it)."""
    torch.manual_seed(11)
    dtype, hq, hkv, d = torch.bfloat16, 16, 8, 128
    lens = [200, 45]
    T = sum(lens)
    qkv = torch.randn(T, (hq + 2 * hkv) * d, device=DEV, dtype=dtype)
    q = qkv[:, :hq * d].view(T, hq, d)
    k = qkv[:, hq * d:(hq + hkv) * d].view(T, hkv, d)
    v = qkv[:, (hq + hkv) * d:].view(T, hkv, d)
    cu = torch.tensor([0, 200, 245], device=DEV, dtype=torch.int32)
    want = ref.flash_attention_prefill(q, k, v, cu, 0.088)
    got = hip.flash_attention_prefill(q, k, v, cu, 0.088, max_seqlen=200)
    close(got, want, dtype)


def test_paged_prefill_output_views(hip):
    """The kernel stores 16 bytes per lane: an output that is a row-strided view of a wider
    buffer is fine as long as every row starts on a 16-byte boundary; one that does not is
    refused instead of written with misaligned stores."""
    torch.manual_seed(12)
    dtype, hq, hkv, d, bs = torch.bfloat16, 16, 8, 128, 64
    lens = [130, 70]
    T = sum(lens)
    q = torch.randn(T, hq, d, device=DEV, dtype=dtype)
    kk = torch.randn(T, hkv, d, device=DEV, dtype=dtype)
    vv = torch.randn(T, hkv, d, device=DEV, dtype=dtype)
    kc, vc, tables = _fill_paged(kk, vv, lens, bs, hkv, d, dtype)
    cu = torch.tensor([0, 130, 200], device=DEV, dtype=torch.int32)
    want = ref.paged_attention_prefill(q, kc, vc, cu, cu, tables, 0.088)
    wide = torch.zeros(T, hq * d + 64, device=DEV, dtype=dtype)
    out = wide[:, 32:32 + hq * d].view(T, hq, d)          # rows 64 bytes into a wider pitch
    hip.paged_attention_prefill(q, kc, vc, cu, cu, tables, 0.088, out=out)
    close(out, want, dtype)
    assert not wide[:, :32].any() and not wide[:, 32 + hq * d:].any()
    bad = wide[:, 4:4 + hq * d].view(T, hq, d)            # rows start 8 bytes off
    with pytest.raises(RuntimeError):
        hip.paged_attention_prefill(q, kc, vc, cu, cu, tables, 0.088, out=bad)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("bs", [16, 256])
@pytest.mark.parametrize("hq,hkv,d", [(32, 8, 128), (16, 8, 64)])
def test_paged_prefill_with_prefix(hip, dtype, bs, hq, hkv, d):
    """q_len <= k_len: the new tokens attend to a cached prefix through the block table."""
    torch.manual_seed(12)
    k_lens = [300, 64, 513, 40, 256]
    q_lens = [300, 1, 257, 40, 130]         # full, single token, chunk, full, partial
    k = torch.randn(sum(k_lens), hkv, d, device=DEV, dtype=dtype)
    v = torch.randn(sum(k_lens), hkv, d, device=DEV, dtype=dtype)
    kc, vc, tables = _fill_paged(k, v, k_lens, bs, hkv, d, dtype)
    q = torch.randn(sum(q_lens), hq, d, device=DEV, dtype=dtype)
    cq = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), device=DEV, dtype=torch.int32)
    ck = torch.tensor([0] + list(torch.tensor(k_lens).cumsum(0)), device=DEV, dtype=torch.int32)
    scale = 1.0 / math.sqrt(d)
    want = ref.paged_attention_prefill(q, kc, vc, cq, ck, tables, scale)
    got = hip.paged_attention_prefill(q, kc, vc, cq, ck, tables, scale)
    close(got, want, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hq,hkv,d", [(32, 8, 128), (16, 8, 64), (8, 1, 128)])
@pytest.mark.parametrize("with_norm", [True, False])
def test_prefill_transforms_raw_q_rows(hip, dtype, hq, hkv, d, with_norm):
    """skip_q + q_rope: the store kernel leaves the Q columns of the fused projection as they
    are and the prefill kernel applies q-norm + RoPE while loading them.  Checked against the
    unfused pair of launches (same arithmetic, so nearly the same bits) and the fp32 oracle;
    packed and paged with cached prefixes."""
    torch.manual_seed(50 + d)
    q_lens = [1, 60, 64, 129, 300, 17]
    prefix = [0, 0, 256, 0, 512, 256]
    k_lens = [a + b for a, b in zip(q_lens, prefix)]
    T, bs = sum(q_lens), 256
    eps, scale = 1e-6, 1.0 / math.sqrt(d)
    qkv = torch.randn(T, (hq + 2 * hkv) * d, device=DEV, dtype=dtype)
    pos = torch.cat([torch.arange(p, p + n) for p, n in zip(prefix, q_lens)]).to(DEV, torch.int32)
    cs = ref.build_cos_sin_cache(d, 1024, 10000.0, DEV)
    qw = (1 + 0.1 * torch.randn(d, device=DEV)).to(dtype) if with_norm else None
    kw = (1 + 0.1 * torch.randn(d, device=DEV)).to(dtype) if with_norm else None
    cu = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), device=DEV, dtype=torch.int32)
    rope = (pos, cs, qw, eps)

    a = qkv.clone()
    q_a, k_a, v_a = hip.qk_norm_rope_store(a, pos, cs, qw, kw, eps, hq, hkv, d, None, None, None)
    b = qkv.clone()
    q_b, k_b, v_b = hip.qk_norm_rope_store(b, pos, cs, qw, kw, eps, hq, hkv, d, None, None, None,
                                           skip_q=True)
    assert torch.equal(q_b, qkv[:, :hq * d].view(T, hq, d))        # untouched
    assert torch.equal(k_b, k_a) and torch.equal(v_b, v_a)
    close(ref.apply_q_rope(q_b, rope), q_a, dtype, 2)
    unfused = hip.flash_attention_prefill(q_a, k_a, v_a, cu, scale)
    fused = hip.flash_attention_prefill(q_b, k_b, v_b, cu, scale, q_rope=rope)
    close(fused, unfused, dtype)
    close(fused, ref.flash_attention_prefill(q_a, k_a, v_a, cu, scale), dtype)

    kk = torch.randn(sum(k_lens), hkv, d, device=DEV, dtype=dtype)
    vv = torch.randn(sum(k_lens), hkv, d, device=DEV, dtype=dtype)
    kc, vc, tables = _fill_paged(kk, vv, k_lens, bs, hkv, d, dtype)
    ck = torch.tensor([0] + list(torch.tensor(k_lens).cumsum(0)), device=DEV, dtype=torch.int32)
    unfused = hip.paged_attention_prefill(q_a, kc, vc, cu, ck, tables, scale)
    fused = hip.paged_attention_prefill(q_b, kc, vc, cu, ck, tables, scale, q_rope=rope)
    close(fused, unfused, dtype)
    close(fused, ref.paged_attention_prefill(q_a, kc, vc, cu, ck, tables, scale), dtype)


@pytest.mark.parametrize("hq,hkv,d", [(32, 8, 128), (16, 8, 128), (8, 1, 128), (16, 8, 64)])
def test_prefill_many_uneven_sequences(hip, hq, hkv, d):
    """Several thousand workgroups over 44 sequences of uneven length (Q tiles without rows,
    single-tile sequences and multi-tile ones side by side), packed and paged with cached
    prefixes; a second launch must give the same bits."""
    torch.manual_seed(40 + hq)
    dtype = torch.bfloat16
    rng = torch.Generator().manual_seed(7)
    B = 44
    q_lens = [int(x) for x in torch.randint(1, 700, (B,), generator=rng)]
    q_lens[3], q_lens[17], q_lens[30] = 1, 64, 129
    prefix = [int(x) for x in torch.randint(0, 3, (B,), generator=rng) * 150]
    k_lens = [a + b for a, b in zip(q_lens, prefix)]
    scale = 1.0 / math.sqrt(d)
    T = sum(q_lens)
    q = torch.randn(T, hq, d, device=DEV, dtype=dtype)
    k = torch.randn(T, hkv, d, device=DEV, dtype=dtype)
    v = torch.randn(T, hkv, d, device=DEV, dtype=dtype)
    cu = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), device=DEV, dtype=torch.int32)
    want = ref.flash_attention_prefill(q, k, v, cu, scale)
    got = hip.flash_attention_prefill(q, k, v, cu, scale)
    close(got, want, dtype)
    again = hip.flash_attention_prefill(q, k, v, cu, scale, max_seqlen=max(q_lens))
    assert torch.equal(got, again)
    kk = torch.randn(sum(k_lens), hkv, d, device=DEV, dtype=dtype)
    vv = torch.randn(sum(k_lens), hkv, d, device=DEV, dtype=dtype)
    kc, vc, tables = _fill_paged(kk, vv, k_lens, 256, hkv, d, dtype)
    ck = torch.tensor([0] + list(torch.tensor(k_lens).cumsum(0)), device=DEV, dtype=torch.int32)
    want = ref.paged_attention_prefill(q, kc, vc, cu, ck, tables, scale)
    got = hip.paged_attention_prefill(q, kc, vc, cu, ck, tables, scale)
    close(got, want, dtype)


def test_flash_prefill_long_and_softmax_rescale(hip):
    """4k tokens (many KV tiles) with a spiked key late in the sequence, which forces the
    running-max rescale branch of the online softmax on rows that already accumulated."""
    torch.manual_seed(13)
    dtype, hq, hkv, d, T = torch.bfloat16, 8, 2, 128, 4096
    q = torch.randn(T, hq, d, device=DEV, dtype=dtype)
    k = torch.randn(T, hkv, d, device=DEV, dtype=dtype)
    v = torch.randn(T, hkv, d, device=DEV, dtype=dtype)
    k[3000] = 6.0 * q[3500, 0]               # row 3500 of head 0 (kv head 0) sees a huge score
    cu = torch.tensor([0, T], device=DEV, dtype=torch.int32)
    scale = 1.0 / math.sqrt(d)
    want = ref.flash_attention_prefill(q, k, v, cu, scale)
    got = hip.flash_attention_prefill(q, k, v, cu, scale)
    close(got, want, dtype)


def test_decode_matches_prefill_last_row(hip):
    """Cross-check of the two attention kernels: decode of the last token == last row of prefill."""
    torch.manual_seed(14)
    dtype, hq, hkv, d, bs, L = torch.bfloat16, 32, 8, 128, 256, 777
    k = torch.randn(L, hkv, d, device=DEV, dtype=dtype)
    v = torch.randn(L, hkv, d, device=DEV, dtype=dtype)
    q = torch.randn(L, hq, d, device=DEV, dtype=dtype)
    kc, vc, tables = _fill_paged(k, v, [L], bs, hkv, d, dtype)
    cu = torch.tensor([0, L], device=DEV, dtype=torch.int32)
    pre = hip.paged_attention_prefill(q, kc, vc, cu, cu, tables, 0.0884)
    dec = hip.paged_attention_decode(q[-1:].contiguous(), kc, vc, tables,
                                     torch.tensor([L], device=DEV, dtype=torch.int32), 0.0884)
    close(dec[0], pre[-1], dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hq,hkv,d", [(32, 8, 128), (4, 4, 64)])
def test_naive_attention_baseline(hip, dtype, hq, hkv, d):
    torch.manual_seed(15)
    lens = [1, 60, 128, 17]
    T = sum(lens)
    q = torch.randn(T, hq, d, device=DEV, dtype=dtype)
    k = torch.randn(T, hkv, d, device=DEV, dtype=dtype)
    v = torch.randn(T, hkv, d, device=DEV, dtype=dtype)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), device=DEV, dtype=torch.int32)
    want = ref.flash_attention_prefill(q, k, v, cu, 1 / math.sqrt(d))
    got = hip.naive_attention(q, k, v, cu, 1 / math.sqrt(d), 128)
    close(got, want, dtype)
    with pytest.raises(ValueError):
        hip.naive_attention(q, k, v, cu, 0.1, 129)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 5, 16])
@pytest.mark.parametrize("N,K", [(4096, 4096), (6144, 4096), (4096, 12288), (3072, 2048), (4096, 1536)])
def test_skinny_gemm(hip, dtype, M, N, K):
    torch.manual_seed(16)
    x = torch.randn(M, K, device=DEV, dtype=dtype)
    w = (torch.randn(N, K, device=DEV) * 0.05).to(dtype)
    b = torch.randn(N, device=DEV, dtype=dtype)
    want = x.float() @ w.float().T
    close(hip.skinny_linear(x, w), want, dtype, 4)
    close(hip.skinny_linear(x, w, b), want + b.float(), dtype, 4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 7, 16])
@pytest.mark.parametrize("N,K,splits", [(4096, 4096, 2), (4096, 4096, 4), (4096, 12288, 3),
                                        (4096, 12288, 4), (1024, 3072, 2), (5120, 5120, 2),
                                        (6144, 4096, 2), (4096, 1536, 3)])
def test_skinny_gemm_split_k(hip, dtype, M, N, K, splits):
    """K split over workgroups: the fp32 slices summed by the reduce kernel and by the
    residual-add + RMSNorm kernel, against the fp32 product; x is a strided view."""
    torch.manual_seed(M + splits)
    assert hip.skinny_splits_ok(K, splits)
    x = torch.randn(M, K + 64, device=DEV, dtype=dtype)[:, :K]
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(dtype)
    bias = torch.randn(N, device=DEV, dtype=dtype) if M == 7 else None
    y = x.float() @ w.float().T + (bias.float() if bias is not None else 0)
    assert _rel(hip.skinny_linear(x, w, bias, splits=splits), y) < 0.01
    parts = hip.skinny_linear(x, w, bias, splits=splits, defer=True)
    assert isinstance(parts, hip.SplitKPartials) and tuple(parts.ws.shape) == (splits, M, N)
    # every slice is its own K range of the product
    ks = K // splits
    for i in range(splits):
        part = x[:, i * ks:(i + 1) * ks].float() @ w[:, i * ks:(i + 1) * ks].float().T
        assert _rel(parts.ws[i], part) < 1e-3
    gamma = (1 + 0.1 * torch.randn(N, device=DEV)).to(dtype)
    resid = torch.randn(M, N, device=DEV, dtype=dtype)
    want_r = y + resid.float()
    h, r = parts.add_rms_norm(resid, gamma, 1e-6)
    assert r.data_ptr() == resid.data_ptr() and _rel(r, want_r) < 0.01
    rf = r.float()
    assert _rel(h, rf * torch.rsqrt(rf.pow(2).mean(-1, keepdim=True) + 1e-6) * gamma.float()) < 0.01
    # a split that does not divide K into multiples of 128 falls back to the unsplit kernel
    t = hip.skinny_linear(x, w, bias, splits=7, defer=True)
    assert torch.is_tensor(t) and _rel(t, y) < 0.01


def test_ops_linear_to_norm_dispatch_few_rows(hip):
    """<= 16 rows: ops.linear(to_norm=True) follows the split recorded by the start-up tuner
    (0 = library, 1 = unsplit kernel, > 1 = partials), plain calls never see partials."""
    from minivllm_amd import ops
    M, N, K = 4, 4096, 4096
    x = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(torch.bfloat16)
    y = x.float() @ w.float().T
    saved = dict(hip._SKINNY_CHOICE)
    try:
        for choice, partials in ((None, False), (0, False), (1, False), (2, True), (4, True)):
            hip._SKINNY_CHOICE.clear()
            if choice is not None:
                hip._SKINNY_CHOICE[(8, N, K, "norm")] = choice
            out = ops.linear(x, w, to_norm=True)
            assert isinstance(out, hip.SplitKPartials) == partials
            assert torch.is_tensor(ops.linear(x, w))
            got = out.materialize() if partials else out
            assert _rel(got, y) < 0.01
    finally:
        hip._SKINNY_CHOICE.clear()
        hip._SKINNY_CHOICE.update(saved)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K,N,swiglu,with_res", [
    (1, 4096, 6144, False, True), (8, 4096, 6144, False, True), (3, 4096, 24576, True, True),
    (8, 4096, 24576, True, True), (2, 5120, 7168, False, False), (5, 2048, 8192, True, True)])
def test_skinny_norm_gemm(hip, dtype, M, K, N, swiglu, with_res):
    """residual add + RMSNorm + projection in one kernel (weights carry the norm's scale
    vector) vs the fp32 formula; the new residual must be the rounded sum, bit for bit."""
    torch.manual_seed(23)
    xx = torch.randn(M + 2, K + 32, device=DEV, dtype=dtype)
    x = xx[:M, :K]                                              # strided rows
    res = torch.randn(M, K, device=DEV, dtype=dtype) if with_res else None
    w = (torch.randn(N, K, device=DEV) * 0.03).to(dtype)
    x0, res0 = x.clone(), (res.clone() if with_res else None)
    assert hip.skinny_norm_gemm_pays(x, w, swiglu) == (M <= hip.SKINNY_NORM_MAX_ROWS)
    out, r = hip.skinny_norm_linear(x, res, 1e-6, w, swiglu=swiglu)
    r_want = (x0.float() + res0.float()).to(dtype) if with_res else x0
    assert torch.equal(r, r_want)
    assert torch.equal(x, x0) and (not with_res or torch.equal(res, res0))    # inputs untouched
    rf = r_want.float()
    y = (rf @ w.float().T) * torch.rsqrt(rf.pow(2).mean(-1, keepdim=True) + 1e-6)
    want = torch.nn.functional.silu(y[:, :N // 2]) * y[:, N // 2:] if swiglu else y
    close(out, want, dtype, 4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 7, 16])
def test_skinny_gemm_swiglu(hip, dtype, M):
    torch.manual_seed(17)
    K, I = 4096, 12288
    xx = torch.randn(M + 3, K + 64, device=DEV, dtype=dtype)
    x = xx[:M, :K]                                    # strided rows
    w = (torch.randn(2 * I, K, device=DEV) * 0.03).to(dtype)
    y = x.float() @ w.float().T
    want = torch.nn.functional.silu(y[:, :I]) * y[:, I:]
    close(hip.skinny_linear(x, w, swiglu=True), want, dtype, 4)
    assert hip.skinny_gemm_ok(x, w, swiglu=True)
    assert not hip.skinny_gemm_ok(torch.empty(17, K, device=DEV, dtype=dtype), w)


# ------------------------------------------------- mid-batch weight-streaming GEMM
# projection shapes (N, K): Qwen3-0.6B / 8B / 32B at TP=1 and the per-rank 8B / 32B shapes at TP=8
_MID_SHAPES = [(4096, 1024), (1024, 2048), (1024, 3072),                    # 0.6B qkv / o / down
               (6144, 4096), (4096, 4096), (4096, 12288),                   # 8B
               (10240, 5120), (5120, 8192), (5120, 25600),                  # 32B
               (768, 4096), (4096, 512), (4096, 1536), (1280, 5120), (5120, 3200)]   # TP=8 ranks


def _rel(got, want):
    return float((got.float() - want.float()).abs().max() / want.float().abs().max())


def _mid_check(hip, dtype, M, N, K, bn, splits, swiglu=False, bias=False, strided=False):
    torch.manual_seed(31)
    if strided:
        x = torch.randn(M + 3, K + 64, device=DEV, dtype=dtype)[:M, :K]
    else:
        x = torch.randn(M, K, device=DEV, dtype=dtype)
    w = (torch.randn(N, K, device=DEV) * 0.03).to(dtype)
    b = torch.randn(N, device=DEV, dtype=dtype) if bias else None
    y = x.float() @ w.float().T
    if bias:
        y = y + b.float()
    want = torch.nn.functional.silu(y[:, :N // 2]) * y[:, N // 2:] if swiglu else y
    for nt in (True, False):
        got = hip.mid_linear(x, w, b, swiglu=swiglu, bn=bn, splits=splits, nt=nt)
        assert got.shape == want.shape
        close(got, want, dtype, 4)


@pytest.mark.parametrize("M", [17, 32, 64, 128, 256])
@pytest.mark.parametrize("N,K", _MID_SHAPES)
def test_mid_gemm_shapes(hip, M, N, K):
    """Every projection shape x row count vs the fp32 product, with the first candidate
    configuration the tuner would try plus a deliberately uneven K split."""
    dtype = torch.bfloat16
    cands = hip.mid_gemm_candidates(M, N, K)
    assert cands
    bn, sp = cands[0]
    _mid_check(hip, dtype, M, N, K, bn, sp)
    _mid_check(hip, dtype, M, N, K, 64, 3)           # k-steps not divisible by the split count


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [19, 64, 100, 128, 200, 256])
@pytest.mark.parametrize("bn,splits", [(128, 1), (64, 1), (128, 4), (64, 7), (96, 1), (96, 3)])
def test_mid_gemm_tiles(hip, dtype, M, bn, splits):
    """Each row tile x column tile x split mode, with bias and strided activations."""
    if bn == 96 and M <= 128:
        pytest.skip("96-column tiles exist for the 256-row tile only")
    _mid_check(hip, dtype, M, 1536, 2048, bn, splits, bias=True, strided=True)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [17, 64, 128, 256])
@pytest.mark.parametrize("I,K", [(3072, 1024), (12288, 4096), (1536, 4096), (3200, 5120)])
@pytest.mark.parametrize("bn,splits", [(128, 1), (64, 2), (96, 1), (96, 2)])
def test_mid_gemm_swiglu(hip, dtype, M, I, K, bn, splits):
    """Fused [gate; up] weight -> silu(gate) * up, direct epilogue and through the K-split
    reduce kernel (3200 = 25 x 128: the Qwen3-32B intermediate size per TP=8 rank)."""
    if (2 * I) % bn or I % (bn // 2) or (bn == 96 and M <= 128):
        pytest.skip("tile does not divide the intermediate size / no such tile")
    _mid_check(hip, dtype, M, 2 * I, K, bn, splits, swiglu=True, bias=(M == 64))


def test_mid_gemm_k_structure(hip):
    """Exact small-integer data: every k position of every row must land in the right
    accumulator (a swapped chunk or a wrong XOR shows up as an integer mismatch)."""
    M, N, K = 256, 256, 512
    x = torch.randint(-2, 3, (M, K), device=DEV).to(torch.bfloat16)
    w = torch.randint(-2, 3, (N, K), device=DEV).to(torch.bfloat16)
    # asymmetric: scale w rows and x columns so that transposes / k-permutations cannot cancel
    w[::3] *= 2
    x[:, ::5] *= 2
    want = x.float() @ w.float().T
    for bn in (64, 128):
        for sp in (1, 2):
            got = hip.mid_linear(x, w, bn=bn, splits=sp)
            assert torch.equal(got.float(), want.to(torch.bfloat16).float())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,bn,splits", [(24, 1024, 2048, 64, 4), (256, 4096, 4096, 128, 4),
                                              (200, 4096, 12288, 128, 8), (64, 5120, 3200, 64, 3),
                                              (130, 2560, 1024, 128, 16), (33, 32768, 512, 64, 2)])
def test_mid_gemm_partials_into_add_rms_norm(hip, dtype, M, N, K, bn, splits):
    """K-slice partials handed to the norm: (sum of slices + bias + residual, its RMSNorm)
    against fp32, and against GEMM -> add_rms_norm through the separate kernels; the residual
    is a strided view (rows of a wider buffer)."""
    torch.manual_seed(M + N)
    if N % bn:
        pytest.skip("tile does not divide N")
    x = torch.randn(M, K, device=DEV, dtype=dtype)
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(dtype)
    bias = torch.randn(N, device=DEV, dtype=dtype) if M % 2 == 0 else None
    gamma = (1 + 0.1 * torch.randn(N, device=DEV)).to(dtype)
    wide = torch.randn(M, N + 64, device=DEV, dtype=dtype)
    resid = wide[:, :N]
    resid0, beside = resid.clone(), wide[:, N:].clone()
    parts = hip.mid_linear(x, w, bias, bn=bn, splits=splits, defer=True)
    assert isinstance(parts, hip.SplitKPartials) and tuple(parts.shape) == (M, N)
    assert parts.ws.shape[0] == min(splits, K // 64) or parts.ws.shape[0] < splits
    y = x.float() @ w.float().T + (bias.float() if bias is not None else 0)
    dense = parts.materialize()
    assert _rel(dense, y) < 0.01
    h, r = parts.add_rms_norm(resid, gamma, 1e-6)
    assert r.data_ptr() == resid.data_ptr()
    want_r = y + resid0.float()
    assert _rel(r, want_r) < 0.01
    rf = r.float()
    want_h = rf * torch.rsqrt(rf.pow(2).mean(-1, keepdim=True) + 1e-6) * gamma.float()
    assert _rel(h, want_h) < 0.01
    assert torch.equal(wide[:, N:], beside)                 # nothing written beside the view
    # no residual: plain norm of the summed slices
    h2, r2 = hip.mid_linear(x, w, bias, bn=bn, splits=splits, defer=True).add_rms_norm(
        None, gamma, 1e-6)
    assert _rel(r2, y) < 0.01 and r2.shape == (M, N)
    # without a real K split the call returns the finished tensor
    t = hip.mid_linear(x, w, bias, bn=bn, splits=1, defer=True)
    assert torch.is_tensor(t) and _rel(t, y) < 0.01


def test_ops_linear_to_norm_dispatch(hip):
    """ops.linear(to_norm=True) hands partials to ops.norm_linear / ops.add_rms_norm only for
    a shape whose tuned entry has a K split; everything else gets a tensor."""
    from minivllm_amd import ops
    M, N, K = 48, 1024, 2048
    x = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(torch.bfloat16)
    w2 = (torch.randn(512, N, device=DEV) / math.sqrt(N)).to(torch.bfloat16)
    gamma = torch.ones(N, device=DEV, dtype=torch.bfloat16)
    saved = dict(hip._MID_CHOICE)
    try:
        hip._MID_CHOICE.clear()
        assert torch.is_tensor(ops.linear(x, w, to_norm=True))
        hip._MID_CHOICE[(64, N, K, "norm")] = (64, 4, True)
        assert torch.is_tensor(ops.linear(x, w))                       # plain call: library
        p = ops.linear(x, w, to_norm=True)
        assert isinstance(p, hip.SplitKPartials)
        resid = torch.randn(M, N, device=DEV, dtype=torch.bfloat16)
        want_r = x.float() @ w.float().T + resid.float()
        out, r = ops.norm_linear(p, resid.clone(), gamma, 1e-6, w2)
        assert _rel(r, want_r) < 0.01
        nf = want_r * torch.rsqrt(want_r.pow(2).mean(-1, keepdim=True) + 1e-6)
        assert _rel(out, nf @ w2.float().T) < 0.02
        h, r = ops.add_rms_norm(ops.linear(x, w, to_norm=True), resid.clone(), gamma, 1e-6)
        assert _rel(h, nf) < 0.01
        assert _rel(ops.rms_norm(ops.linear(x, w, to_norm=True), gamma, 1e-6),
                    (lambda y: y * torch.rsqrt(y.pow(2).mean(-1, keepdim=True) + 1e-6))(
                        x.float() @ w.float().T)) < 0.01
    finally:
        hip._MID_CHOICE.clear()
        hip._MID_CHOICE.update(saved)


# ------------------------------------------------------- prefill-sized GEMM (256 x 256 tiles)
def _big_check(hip, dtype, M, N, K, swiglu=False, bias=False, strided=False, out_strided=False):
    torch.manual_seed(37)
    if strided:
        x = torch.randn(M + 3, K + 64, device=DEV, dtype=dtype)[:M, :K]
    else:
        x = torch.randn(M, K, device=DEV, dtype=dtype)
    w = (torch.randn(N, K, device=DEV) * 0.03).to(dtype)
    b = torch.randn(N, device=DEV, dtype=dtype) if bias else None
    y = x.float() @ w.float().T
    if bias:
        y = y + b.float()
    want = torch.nn.functional.silu(y[:, :N // 2]) * y[:, N // 2:] if swiglu else y
    n_out = want.shape[1]
    if out_strided:
        buf = torch.full((M + 2, n_out + 72), 7.0, device=DEV, dtype=dtype)
        got = hip.big_linear(x, w, b, swiglu, out=buf[:M, :n_out])
        assert got.data_ptr() == buf.data_ptr()
        assert bool((buf[M:] == 7).all()) and bool((buf[:, n_out:] == 7).all()), "wrote outside its view"
    else:
        guard = torch.full((M + 1, n_out), 7.0, device=DEV, dtype=dtype)
        got = hip.big_linear(x, w, b, swiglu, out=guard[:M])
        assert bool((guard[M] == 7).all()), "wrote a row beyond M"
    assert got.shape == want.shape
    close(got, want, dtype, 4)


# (N, K) of the projections of Qwen3-0.6B / 8B / 32B and of the per-rank shapes at TP = 8 that the
# kernel can take (N a multiple of 256)
_BIG_SHAPES = [(4096, 1024), (1024, 2048), (1024, 3072), (6144, 4096), (4096, 4096), (4096, 12288),
               (10240, 5120), (5120, 8192), (768, 4096), (4096, 512), (4096, 1536), (1280, 5120),
               (5120, 3200)]


@pytest.mark.parametrize("M", [257, 1000, 2048])
@pytest.mark.parametrize("N,K", _BIG_SHAPES)
def test_big_gemm_shapes(hip, M, N, K):
    """Projection shapes x row counts (one below a tile edge, one inside a tile, whole tiles) vs
    the fp32 product."""
    if K % 64:
        pytest.skip("K must be a multiple of 64")
    _big_check(hip, torch.bfloat16, M, N, K)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [300, 512, 1111])
@pytest.mark.parametrize("I,K", [(3072, 1024), (12288, 4096), (1536, 4096), (3200, 5120), (128, 256)])
def test_big_gemm_swiglu(hip, dtype, M, I, K):
    """Fused [gate; up] weight -> silu(gate) * up in the epilogue (3200 = 25 x 128: the Qwen3-32B
    intermediate size per TP=8 rank), with bias on one row count."""
    _big_check(hip, dtype, M, 2 * I, K, swiglu=True, bias=(M == 512))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("swiglu", [False, True])
def test_big_gemm_views(hip, dtype, swiglu):
    """Strided activations, an output view inside a wider buffer, bias."""
    _big_check(hip, dtype, 700, 1024, 512, swiglu=swiglu, bias=True, strided=True, out_strided=True)


def test_big_gemm_many_tiles_per_workgroup(hip):
    """More tiles than CUs (each workgroup walks several tiles: the half-tile pipeline crosses
    tile boundaries and the last round is partly empty), odd and even K-tile counts."""
    for M, N, K in ((4096 + 64, 8192, 256), (3000, 6144, 320), (8192, 2560, 448)):
        _big_check(hip, torch.bfloat16, M, N, K)
    _big_check(hip, torch.bfloat16, 5000, 2 * 4096, 384, swiglu=True)


def test_big_gemm_k_structure(hip):
    """Exact small-integer data: every k position of every row and column must land in the right
    accumulator, and every output element in the right place."""
    M, N, K = 777, 768, 512
    x = torch.randint(-2, 3, (M, K), device=DEV).to(torch.bfloat16)
    w = torch.randint(-2, 3, (N, K), device=DEV).to(torch.bfloat16)
    w[::3] *= 2
    x[:, ::5] *= 2
    want = x.float() @ w.float().T
    got = hip.big_linear(x, w)
    assert torch.equal(got.float(), want.to(torch.bfloat16).float())


def test_big_gemm_is_deterministic_under_memory_load(hip):
    """The kernel's accumulation order is fixed, so a run that differs from the first in a single
    bit has read a half-tile before it landed (or overwritten one still in use).  Repeats two
    shapes while a second stream keeps the memory system busy (tools/big_gemm_race_screen.py is
    the long form)."""
    torch.manual_seed(41)
    src = torch.empty(1 << 26, dtype=torch.int32, device=DEV)
    dst = torch.empty_like(src)
    side = torch.cuda.Stream()
    for M, N, K, swi in ((4096, 4096, 1024, False), (4096, 6144, 2048, True)):
        x = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
        w = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
        first = hip.big_linear(x, w, None, swi).clone()
        for _ in range(25):
            with torch.cuda.stream(side):
                dst.copy_(src, non_blocking=True)
            assert torch.equal(hip.big_linear(x, w, None, swi), first)
    torch.cuda.synchronize()


def test_big_gemm_rejects(hip):
    w = torch.zeros(1024, 1024, device=DEV, dtype=torch.bfloat16)
    x = torch.zeros(512, 1024, device=DEV, dtype=torch.bfloat16)
    assert hip.big_gemm_ok(x, w) and hip.big_gemm_ok(x, w, swiglu=True)
    assert not hip.big_gemm_ok(x, w[:1000])                         # N not a multiple of 256
    assert not hip.big_gemm_ok(x[:, :192].contiguous(), w[:, :192].contiguous())   # K < 256 ... or not a multiple of 64
    assert not hip.big_gemm_ok(x.float(), w.float())
    assert not hip.big_gemm_pays(x, w)                              # nothing measured: library
    with pytest.raises(ValueError):
        hip.big_linear(x, w[:1000])


def test_mid_gemm_rejects(hip):
    w = torch.zeros(1024, 1000, device=DEV, dtype=torch.bfloat16)
    assert not hip.mid_gemm_ok(torch.zeros(64, 1000, device=DEV, dtype=torch.bfloat16), w)
    w = torch.zeros(1024, 1024, device=DEV, dtype=torch.bfloat16)
    assert not hip.mid_gemm_ok(torch.zeros(257, 1024, device=DEV, dtype=torch.bfloat16), w)
    assert hip.mid_gemm_ok(torch.zeros(256, 1024, device=DEV, dtype=torch.bfloat16), w)
    with pytest.raises(ValueError):
        hip.mid_linear(torch.zeros(300, 1024, device=DEV, dtype=torch.bfloat16), w)


# ------------------------------------------------------------------ fp8 KV cache
FP8 = torch.float8_e4m3fn


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hq,hkv,d,bs", [(32, 8, 128, 256), (16, 8, 64, 16), (8, 1, 128, 16)])
def test_paged_decode_fp8_cache(hip, dtype, hq, hkv, d, bs):
    """The decode kernel over an fp8 (e4m3) cache against the oracle over the same cache widened
    to 16 bits: only the arithmetic may differ, the quantisation is shared."""
    torch.manual_seed(21)
    lens = [1, 31, 32, 33, 260, 1000, 77]
    k = torch.randn(sum(lens), hkv, d, device=DEV, dtype=dtype)
    v = torch.randn(sum(lens), hkv, d, device=DEV, dtype=dtype)
    kc, vc, tables = _fill_paged(k, v, lens, bs, hkv, d, dtype)
    kc8, vc8 = kc.to(FP8), vc.to(FP8)
    assert kc8.stride() == kc.stride()
    q = torch.randn(len(lens), hq, d, device=DEV, dtype=dtype)
    ctx = torch.tensor(lens, device=DEV, dtype=torch.int32)
    scale = 1.0 / math.sqrt(d)
    want = ref.paged_attention_decode(q, kc8.to(dtype), vc8.to(dtype), tables, ctx, scale)
    for ns in (1, 4):
        got = hip.paged_attention_decode(q, kc8, vc8, tables, ctx, scale, num_splits=ns)
        close(got, want, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d", [64, 128])
def test_qk_norm_rope_store_into_fp8_cache(hip, dtype, d):
    torch.manual_seed(22)
    T, hq, hkv, bs, nb = 41, 8, 2, 16, 8
    qkv = torch.randn(T, (hq + 2 * hkv) * d, device=DEV, dtype=dtype)
    pos = torch.randint(0, 500, (T,), device=DEV, dtype=torch.int32)
    cs = ref.build_cos_sin_cache(d, 512, 10000.0, DEV)
    qw = (1 + 0.1 * torch.randn(d, device=DEV)).to(dtype)
    kw = (1 + 0.1 * torch.randn(d, device=DEV)).to(dtype)
    slots = torch.randperm(nb * bs, device=DEV)[:T].to(torch.int32)
    slots[7] = -1
    kc_r, vc_r = _logical_cache(nb, bs, hkv, d, dtype), _logical_cache(nb, bs, hkv, d, dtype)
    kc_h, vc_h = _logical_cache(nb, bs, hkv, d, FP8), _logical_cache(nb, bs, hkv, d, FP8)
    ref.qk_norm_rope_store(qkv.clone(), pos, cs, qw, kw, 1e-6, hq, hkv, d, kc_r, vc_r, slots)
    q_h, k_h, v_h = hip.qk_norm_rope_store(qkv.clone(), pos, cs, qw, kw, 1e-6, hq, hkv, d, kc_h, vc_h, slots)
    # fp8 has 3 mantissa bits: a 16-bit value that differs by one ulp between the two
    # implementations may land in the neighbouring fp8 value
    for got, want in ((kc_h, kc_r), (vc_h, vc_r)):
        g, w = got.float(), want.to(FP8).float()
        assert (g - w).abs().max() <= 0.13 * w.abs().max().clamp(min=1.0)
        assert (g != w).float().mean() < 0.02
    assert torch.equal(vc_h.float(), vc_r.to(FP8).float())      # v is copied, never recomputed
    # out-of-range values saturate at the largest finite e4m3 number instead of becoming NaN
    big = qkv.clone()
    v0 = (hq + hkv) * d
    big[3, v0] = 3000.0
    big[3, v0 + 1] = -1.0e4
    kc_s, vc_s = _logical_cache(nb, bs, hkv, d, FP8), _logical_cache(nb, bs, hkv, d, FP8)
    hip.qk_norm_rope_store(big, pos, cs, qw, kw, 1e-6, hq, hkv, d, kc_s, vc_s, slots)
    blk, off = int(slots[3]) // bs, int(slots[3]) % bs
    row = vc_s[blk, off, 0].float()
    assert row[0] == 448.0 and row[1] == -448.0 and not torch.isnan(vc_s.float()).any()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("bs,hq,hkv,d", [(16, 16, 8, 128), (256, 32, 8, 128), (16, 8, 2, 64)])
def test_paged_prefill_over_fp8_cache(hip, dtype, bs, hq, hkv, d):
    torch.manual_seed(23)
    k_lens = [300, 64, 513, 40, 256]
    q_lens = [300, 1, 257, 40, 130]
    k = torch.randn(sum(k_lens), hkv, d, device=DEV, dtype=dtype)
    v = torch.randn(sum(k_lens), hkv, d, device=DEV, dtype=dtype)
    kc, vc, tables = _fill_paged(k, v, k_lens, bs, hkv, d, dtype)
    kc8, vc8 = kc.to(FP8), vc.to(FP8)
    q = torch.randn(sum(q_lens), hq, d, device=DEV, dtype=dtype)
    cq = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), device=DEV, dtype=torch.int32)
    ck = torch.tensor([0] + list(torch.tensor(k_lens).cumsum(0)), device=DEV, dtype=torch.int32)
    scale = 1.0 / math.sqrt(d)
    want = ref.paged_attention_prefill(q, kc8.to(dtype), vc8.to(dtype), cq, ck, tables, scale)
    close(hip.paged_attention_prefill(q, kc8, vc8, cq, ck, tables, scale), want, dtype)
    got = hip.paged_attention_prefill(q, kc8, vc8, cq, ck, tables, scale, max_seqlen_q=300,
                                      max_seqlen_k=513)            # the engine's call: no sync
    close(got, want, dtype)


# ------------------------------------- head geometries of the reference's model-shape sweep
# qwen3-30b 40 / 8 (group of 5), llama3-70b 64 / 8 (8), qwen3-235b-moe 64 / 4 (16) (reference:
# benchmark_models.py:10-43), plus what one rank of such a model sees under tensor parallelism
# (5 / 1, 16 / 1) and a group wider than one 16-head pass of the decode kernel (20 / 1).
_SWEEP_GEOMETRIES = [(40, 8), (64, 8), (64, 4), (5, 1), (16, 1), (20, 1)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hq,hkv", _SWEEP_GEOMETRIES)
@pytest.mark.parametrize("bs", [16, 256])
def test_paged_decode_sweep_geometries(hip, dtype, hq, hkv, bs):
    """Decode with every split count (incl. the in-kernel split merge and the automatic choice)
    over bf16/fp16 and fp8 caches."""
    torch.manual_seed(90 + hq)
    d = 128
    lens = [1, 33, 260, 1000, 77, 512]
    k = torch.randn(sum(lens), hkv, d, device=DEV, dtype=dtype)
    v = torch.randn(sum(lens), hkv, d, device=DEV, dtype=dtype)
    kc, vc, tables = _fill_paged(k, v, lens, bs, hkv, d, dtype)
    q = torch.randn(len(lens), hq, d, device=DEV, dtype=dtype)
    ctx = torch.tensor(lens, device=DEV, dtype=torch.int32)
    scale = 1.0 / math.sqrt(d)
    want = ref.paged_attention_decode(q, kc, vc, tables, ctx, scale)
    for ns in (1, 3, 8, 32):
        close(hip.paged_attention_decode(q, kc, vc, tables, ctx, scale, num_splits=ns), want, dtype)
    close(hip.paged_attention_decode(q, kc, vc, tables, ctx, scale, max_context_len=max(lens)), want, dtype)
    kc8, vc8 = kc.to(FP8), vc.to(FP8)
    want8 = ref.paged_attention_decode(q, kc8.to(dtype), vc8.to(dtype), tables, ctx, scale)
    for ns in (1, 4):
        close(hip.paged_attention_decode(q, kc8, vc8, tables, ctx, scale, num_splits=ns), want8, dtype)
    tickets = hip.decode_ticket_words(q.device, 1)
    torch.cuda.synchronize()
    assert int(tickets.abs().sum()) == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hq,hkv", _SWEEP_GEOMETRIES)
def test_prefill_sweep_geometries(hip, dtype, hq, hkv):
    """Packed prefill, paged prefill behind cached prefixes (16- and 256-token blocks), the
    fused raw-Q transform and the fp8-cache arm."""
    torch.manual_seed(190 + hq)
    d = 128
    q_lens = [1, 60, 129, 300, 64]
    prefix = [0, 256, 0, 512, 256]
    k_lens = [a + b for a, b in zip(q_lens, prefix)]
    T, scale, eps = sum(q_lens), 1.0 / math.sqrt(d), 1e-6
    cu = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), device=DEV, dtype=torch.int32)
    ck = torch.tensor([0] + list(torch.tensor(k_lens).cumsum(0)), device=DEV, dtype=torch.int32)
    qkv = torch.randn(T, (hq + 2 * hkv) * d, device=DEV, dtype=dtype)
    pos = torch.cat([torch.arange(p, p + n) for p, n in zip(prefix, q_lens)]).to(DEV, torch.int32)
    cs = ref.build_cos_sin_cache(d, 1024, 10000.0, DEV)
    qw = (1 + 0.1 * torch.randn(d, device=DEV)).to(dtype)
    kw = (1 + 0.1 * torch.randn(d, device=DEV)).to(dtype)
    rope = (pos, cs, qw, eps)
    a = qkv.clone()
    q_a, k_a, v_a = hip.qk_norm_rope_store(a, pos, cs, qw, kw, eps, hq, hkv, d, None, None, None)
    b = qkv.clone()
    q_b, k_b, v_b = hip.qk_norm_rope_store(b, pos, cs, qw, kw, eps, hq, hkv, d, None, None, None,
                                           skip_q=True)
    want = ref.flash_attention_prefill(q_a, k_a, v_a, cu, scale)
    close(hip.flash_attention_prefill(q_a, k_a, v_a, cu, scale), want, dtype)
    close(hip.flash_attention_prefill(q_b, k_b, v_b, cu, scale, q_rope=rope), want, dtype)
    kk = torch.randn(sum(k_lens), hkv, d, device=DEV, dtype=dtype)
    vv = torch.randn(sum(k_lens), hkv, d, device=DEV, dtype=dtype)
    for bs in (16, 256):
        kc, vc, tables = _fill_paged(kk, vv, k_lens, bs, hkv, d, dtype)
        want = ref.paged_attention_prefill(q_a, kc, vc, cu, ck, tables, scale)
        close(hip.paged_attention_prefill(q_a, kc, vc, cu, ck, tables, scale), want, dtype)
        close(hip.paged_attention_prefill(q_b, kc, vc, cu, ck, tables, scale, q_rope=rope), want, dtype)
        kc8, vc8 = kc.to(FP8), vc.to(FP8)
        want8 = ref.paged_attention_prefill(q_a, kc8.to(dtype), vc8.to(dtype), cu, ck, tables, scale)
        close(hip.paged_attention_prefill(q_a, kc8, vc8, cu, ck, tables, scale), want8, dtype)


@pytest.mark.parametrize("hq,hkv,d", [(32, 8, 128), (16, 8, 64), (5, 1, 128)])
@pytest.mark.parametrize("bs", [16, 256])
def test_paged_prefill_from_a_work_list(hip, hq, hkv, d, bs):
    """The launch form captured graphs use: one workgroup per (work-list slot, head), the list
    (sequence << 16 | pair of query tiles, -1 = unused slot) in device memory.  Uneven lengths,
    cached prefixes, zero-length sequence slots behind the real ones, far more slots than work,
    the fused raw-Q transform; against the oracle and the length-derived launch (same bits:
    the tiles and their arithmetic are the same, only who computes which pair differs)."""
    torch.manual_seed(300 + hq)
    dtype = torch.bfloat16
    rng = torch.Generator().manual_seed(11)
    B = 23
    q_lens = [int(x) for x in torch.randint(1, 600, (B,), generator=rng)]
    q_lens[2], q_lens[9], q_lens[20] = 1, 128, 257
    prefix = [int(x) for x in torch.randint(0, 3, (B,), generator=rng) * 256]
    k_lens = [a + b for a, b in zip(q_lens, prefix)]
    scale, eps = 1.0 / math.sqrt(d), 1e-6
    T = sum(q_lens)
    kk = torch.randn(sum(k_lens), hkv, d, device=DEV, dtype=dtype)
    vv = torch.randn(sum(k_lens), hkv, d, device=DEV, dtype=dtype)
    kc, vc, tables = _fill_paged(kk, vv, k_lens, bs, hkv, d, dtype)
    q = torch.randn(T, hq, d, device=DEV, dtype=dtype)
    # 9 unused sequence slots (zero length) as in a graph whose sequence bucket is 32
    S = 32
    cq = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)) + [T] * (S - B), device=DEV, dtype=torch.int32)
    ck = torch.tensor([0] + list(torch.tensor(k_lens).cumsum(0)) + [sum(k_lens)] * (S - B), device=DEV,
                      dtype=torch.int32)
    tab = torch.full((S, tables.shape[1]), -1, device=DEV, dtype=torch.int32)
    tab[:B] = tables
    slots = hip.prefill_item_slots(T + 100, S)
    items_np = hip.prefill_work_items(q_lens, k_lens, slots=slots)
    assert (items_np >= 0).sum() <= hip.prefill_item_slots(T, B)
    items = torch.from_numpy(items_np).to(DEV)
    want = ref.paged_attention_prefill(q, kc, vc, cq[:B + 1], ck[:B + 1], tables, scale)
    plain = hip.paged_attention_prefill(q, kc, vc, cq[:B + 1], ck[:B + 1], tables, scale)
    out = torch.full((T, hq, d), float("nan"), device=DEV, dtype=dtype)
    hip.paged_attention_prefill(q, kc, vc, cq, ck, tab, scale, max_seqlen_q=T, out=out, work_items=items)
    close(out, want, dtype)
    assert torch.equal(out, plain)
    # shuffled list: any order of the entries gives the same rows
    perm = torch.randperm(slots)
    out2 = torch.full_like(out, float("nan"))
    hip.paged_attention_prefill(q, kc, vc, cq, ck, tab, scale, max_seqlen_q=T, out=out2,
                                work_items=items[perm.to(DEV)].contiguous())
    assert torch.equal(out2, out)
    # raw Q rows transformed on load
    pos = torch.cat([torch.arange(p, p + n) for p, n in zip(prefix, q_lens)]).to(DEV, torch.int32)
    cs = ref.build_cos_sin_cache(d, 2048, 10000.0, DEV)
    qw = (1 + 0.1 * torch.randn(d, device=DEV)).to(dtype)
    rope = (pos, cs, qw, eps)
    a = hip.paged_attention_prefill(q, kc, vc, cq[:B + 1], ck[:B + 1], tables, scale, q_rope=rope)
    b = hip.paged_attention_prefill(q, kc, vc, cq, ck, tab, scale, max_seqlen_q=T, q_rope=rope,
                                    work_items=items)
    assert torch.equal(a, b)
    # positions derived from the lengths (k_len - q_len + i) instead of loaded
    c = hip.paged_attention_prefill(q, kc, vc, cq, ck, tab, scale, max_seqlen_q=T,
                                    q_rope=(None, cs, qw, eps), work_items=items)
    assert torch.equal(c, b)
    close(b, ref.paged_attention_prefill(ref.apply_q_rope(q, rope), kc, vc, cq[:B + 1], ck[:B + 1],
                                         tables, scale), dtype)
    # captured once, replayed on other work
    g = torch.cuda.CUDAGraph()
    out3 = torch.empty_like(out)
    with torch.cuda.graph(g):
        hip.paged_attention_prefill(q, kc, vc, cq, ck, tab, scale, max_seqlen_q=T, out=out3, work_items=items)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out3, out)
    half = hip.prefill_work_items(q_lens[:10], k_lens[:10], slots=slots)
    items.copy_(torch.from_numpy(half).to(DEV))
    T10 = sum(q_lens[:10])
    cq[11:] = T10
    out3.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out3[:T10], out[:T10]) and not out3[T10:].any()


# ------------------------------- prefill GEMM epilogues that absorb residual add + RMSNorm
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [257, 1000, 4096])
@pytest.mark.parametrize("N,K", [(1024, 2048), (1024, 3072), (4096, 4096), (4096, 12288), (5120, 3200)])
def test_big_gemm_residual_epilogue(hip, dtype, M, N, K):
    """o / down projection with the residual stream added in the epilogue (in place): against
    GEMM and add in fp32; rows beyond M untouched; with and without bias."""
    if K % 64:
        pytest.skip("K must be a multiple of 64")
    torch.manual_seed(61)
    x = torch.randn(M, K, device=DEV, dtype=dtype)
    w = (torch.randn(N, K, device=DEV) * 0.03).to(dtype)
    b = torch.randn(N, device=DEV, dtype=dtype) if M == 1000 else None
    buf = torch.randn(M + 1, N, device=DEV, dtype=dtype)
    res0 = buf[:M].clone()
    guard = buf[M].clone()
    y = x.float() @ w.float().T + (b.float() if b is not None else 0)
    want = y.to(dtype).float() + res0.float()
    rs = hip.big_linear_resid(x, w, b, buf[:M])
    assert rs.r.data_ptr() == buf.data_ptr() and torch.equal(buf[M], guard)
    close(rs.r, want, dtype, 4)
    # the same call as the unfused pair of kernels would leave it
    r2 = res0.clone()
    hip.add_rms_norm(hip.big_linear(x, w, b), r2, torch.ones(N, device=DEV, dtype=dtype), 1e-6)
    assert float((rs.r.float() - r2.float()).abs().max()) <= 2 * float(r2.float().abs().max()) * (2 ** -7)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("swiglu", [False, True])
@pytest.mark.parametrize("M", [300, 2048, 5000])
def test_big_gemm_row_scale_epilogue(hip, dtype, swiglu, M):
    """qkv / gate_up projection of an un-normalised x with the RMSNorm applied as a per-row factor
    behind the GEMM (weights carry gamma): against norm-then-GEMM in fp32 and against the
    unfused kernels."""
    torch.manual_seed(62)
    K, N, eps = 1024, 2048, 1e-6
    x = (torch.randn(M, K, device=DEV) * 3.0).to(dtype)
    gamma = (1 + 0.1 * torch.randn(K, device=DEV)).to(dtype)
    w = (torch.randn(N, K, device=DEV) * 0.03).to(dtype)
    wg = (w.float() * gamma.float()[None]).to(dtype)          # gamma folded into the weights
    ss = x.float().pow(2).sum(-1)
    got = hip.big_linear(x, wg, None, swiglu, row_scale=True, eps=eps)
    xn = x.float() * torch.rsqrt(ss / K + eps)[:, None]
    y = xn @ wg.float().T
    want = torch.nn.functional.silu(y[:, :N // 2]) * y[:, N // 2:] if swiglu else y
    close(got, want, dtype, 4)
    h = hip.rms_norm(x, gamma, eps)
    unfused = hip.silu_and_mul(hip.big_linear(h, w)) if swiglu else hip.big_linear(h, w)
    close(got, unfused.float(), dtype, 6)


def test_ops_residual_sum_chain(hip):
    """ops level: with the measured choice set, a projection handed its residual returns
    ``ResidualSum`` and ``norm_linear`` consumes it (row-scaled GEMM when the weights carry gamma,
    norm kernel + GEMM otherwise); both against the unfused chain."""
    from minivllm_amd import ops
    torch.manual_seed(63)
    dtype, M, H, Ko, I, eps = torch.bfloat16, 1024, 1024, 2048, 1536, 1e-6
    x = torch.randn(M, Ko, device=DEV, dtype=dtype)
    wo = (torch.randn(H, Ko, device=DEV) * 0.03).to(dtype)
    gamma = (1 + 0.1 * torch.randn(H, device=DEV)).to(dtype)
    wgu = (torch.randn(2 * I, H, device=DEV) * 0.03).to(dtype)
    wgu_folded = (wgu.float() * gamma.float()[None]).to(dtype)
    ones = torch.ones(H, device=DEV, dtype=dtype)
    res = torch.randn(M, H, device=DEV, dtype=dtype)
    h_ref, r_ref = hip.add_rms_norm(torch.nn.functional.linear(x, wo), res.clone(), gamma, eps)
    want = hip.silu_and_mul(torch.nn.functional.linear(h_ref, wgu)).float()
    key = (hip.big_row_bucket(M), H, Ko, "resid")
    saved = dict(hip._BIG_CHOICE)
    try:
        hip._BIG_CHOICE[key] = True
        r1 = res.clone()
        y = ops.linear(x, wo, None, to_norm=True, residual=r1)
        assert isinstance(y, ops.ResidualSum) and y.r.data_ptr() == r1.data_ptr()
        act, r_out = ops.norm_linear(y, r1, ones, eps, wgu_folded, None, swiglu=True, gamma_folded=True)
        assert r_out.data_ptr() == r1.data_ptr()
        close(act, want, dtype, 8)
        close(r_out, r_ref.float(), dtype, 4)
        r2 = res.clone()
        y = ops.linear(x, wo, None, to_norm=True, residual=r2)
        act2, _ = ops.norm_linear(y, r2, gamma, eps, wgu, None, swiglu=True, gamma_folded=False)
        close(act2, want, dtype, 8)
        hn, rr = ops.add_rms_norm(ops.linear(x, wo, None, to_norm=True, residual=res.clone()), None, gamma, eps)
        close(hn, h_ref.float(), dtype, 4)
        hip._BIG_CHOICE[key] = False
        assert torch.is_tensor(ops.linear(x, wo, None, to_norm=True, residual=res.clone()))
    finally:
        hip._BIG_CHOICE.clear()
        hip._BIG_CHOICE.update(saved)
