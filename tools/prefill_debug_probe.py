"""Error of the flash prefill kernel against the fp32 oracle over a ladder of shapes, from one
single-tile work item to many items per workgroup: shows which part of the kernel (tile body,
multi-tile loop, hand-over between items) a regression sits in."""
import math
import sys

import torch

from minivllm_amd.ops import hip, torch_ref as ref


def run(lens, hq, hkv, d=128, dtype=torch.bfloat16, seed=0):
    torch.manual_seed(seed)
    T = sum(lens)
    q = torch.randn(T, hq, d, device="cuda", dtype=dtype)
    k = torch.randn(T, hkv, d, device="cuda", dtype=dtype)
    v = torch.randn(T, hkv, d, device="cuda", dtype=dtype)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), device="cuda", dtype=torch.int32)
    want = ref.flash_attention_prefill(q, k, v, cu, 1 / math.sqrt(d)).float()
    got = hip.flash_attention_prefill(q, k, v, cu, 1 / math.sqrt(d)).float()
    torch.cuda.synchronize()
    err = (got - want).abs()
    bad_rows = (err.amax(dim=(1, 2)) > 0.05).nonzero().flatten().tolist()
    bad_heads = (err.amax(dim=(0, 2)) > 0.05).nonzero().flatten().tolist()
    print(f"lens={lens if len(lens) < 8 else str(lens[:4]) + '...'} hq={hq} hkv={hkv} d={d}: "
          f"max err {err.max().item():.4f}  bad rows {len(bad_rows)}/{T} "
          f"{bad_rows[:6]}{'...' if len(bad_rows) > 6 else ''}  bad heads {bad_heads[:8]}", flush=True)


if __name__ == "__main__":
    for lens in ([32], [64], [65], [128], [192], [256], [300], [1024]):
        run(lens, 1, 1)
    run([256], 8, 8)
    run([256], 32, 8)
    run([300, 300], 32, 8)
    run([128] * 40, 32, 8)
    run([1, 60, 64, 129, 300, 17], 32, 8)
    run([64], 1, 1, d=64)
    run([300], 4, 2, d=64)
