"""Does the row pitch of the QKV buffer matter?  Times the K/V pass of qk_norm_rope_store
(skip_q), the paged flash prefill reading Q from the same buffer, and the QKV GEMM writing
it, for a dense [T, 4096] buffer (8 KiB pitch: Qwen3-0.6B) and for views of wider buffers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivllm_amd.ops import hip

torch.manual_seed(0)
dev = "cuda"
T, S, Hq, Hkv, D, H = 16384, 1024, int(os.environ.get("HQ", 16)), 8, 128, int(os.environ.get("HID", 1024))
W = (Hq + 2 * Hkv) * D
B = T // S
bs = 256
nblk = T // bs
cos_sin = torch.randn(4096, D, device=dev, dtype=torch.float32)
qw = torch.ones(D, device=dev, dtype=torch.bfloat16)
kw = torch.ones(D, device=dev, dtype=torch.bfloat16)
positions = (torch.arange(T, device=dev, dtype=torch.int32) % S).contiguous()
slots = torch.arange(T, device=dev, dtype=torch.int32)
cache = torch.zeros(2, nblk, Hkv, bs, D, device=dev, dtype=torch.bfloat16)
kc = cache[0].permute(0, 2, 1, 3)
vc = cache[1].permute(0, 2, 1, 3)
cu = (torch.arange(B + 1, device=dev, dtype=torch.int32) * S).contiguous()
bt = torch.arange(nblk, device=dev, dtype=torch.int32).view(B, S // bs).contiguous()
x = torch.randn(T, H, device=dev, dtype=torch.bfloat16)
wqkv = torch.randn(W, H, device=dev, dtype=torch.bfloat16) * 0.02


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for pad in (0, 64, 128, 256, 2048):
    big = torch.randn(T, W + pad, device=dev, dtype=torch.bfloat16)
    qkv = big[:, :W]
    t_store = timed(lambda: hip.qk_norm_rope_store(qkv, positions, cos_sin, qw, kw, 1e-6, Hq, Hkv, D,
                                                    kc, vc, slots, skip_q=True))
    q = qkv[:, :Hq * D].view(T, Hq, D)
    t_attn = timed(lambda: hip.paged_attention_prefill(q, kc, vc, cu, cu, bt, D ** -0.5, S, max_seqlen_k=S,
                                                        total_k=T, q_rope=(positions, cos_sin, qw, 1e-6)))
    if pad == 0:
        t_gemm = timed(lambda: torch.nn.functional.linear(x, wqkv))
    else:
        t_gemm = timed(lambda: torch.mm(x, wqkv.t(), out=qkv))
        ref = torch.nn.functional.linear(x, wqkv)
        assert torch.equal(ref, qkv) or (ref.float() - qkv.float()).abs().max() < 0.05
        assert qkv.data_ptr() == big.data_ptr()
    print(f"Hq {Hq} pitch {W + pad:5d} el: rope+store {t_store:6.1f} us   flash prefill {t_attn:6.1f} us   "
          f"qkv gemm {t_gemm:6.1f} us", flush=True)

# ---- the same K/V pass into an engine-sized cache: [2, layers, blocks, Hkv, bs, D], blocks of the
# step scattered over one layer's 4.3 GB region
L, NB = 4, 8192
big_cache = torch.zeros(2, L, NB, Hkv, bs, D, device=dev, dtype=torch.bfloat16)
qkv = torch.randn(T, W, device=dev, dtype=torch.bfloat16)
for name, ids in (("first blocks", torch.arange(nblk)), ("random blocks", torch.randperm(NB)[:nblk]),
                  ("descending", torch.arange(NB - 1, NB - 1 - nblk, -1))):
    ids = ids.to(dev)
    sl = (ids.repeat_interleave(bs) * bs + torch.arange(bs, device=dev).repeat(nblk)).to(torch.int32)
    kc2 = big_cache[0, 2].permute(0, 2, 1, 3)
    vc2 = big_cache[1, 2].permute(0, 2, 1, 3)
    t = timed(lambda: hip.qk_norm_rope_store(qkv, positions, cos_sin, qw, kw, 1e-6, Hq, Hkv, D,
                                              kc2, vc2, sl, skip_q=True))
    print(f"engine-sized cache, {name:14s}: rope+store {t:6.1f} us", flush=True)
