#!/usr/bin/env python3
"""Host launch time vs GPU time of one eager prefill step (which of the two bounds small steps)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from minivllm_amd.config import EngineConfig, model_preset
from minivllm_amd.engine.block_manager import BlockManager
from minivllm_amd.engine.model_runner import ModelRunner
from minivllm_amd.engine.sequence import Sequence
from minivllm_amd.sampling_parameters import SamplingParams

name = os.environ.get("MODEL", "qwen3-0.6b")
model = model_preset(name)
cfg = EngineConfig(model=model, max_num_sequences=64, max_num_batched_tokens=16384, block_size=256,
                   eos=-1, enforce_eager=True, max_model_length=2048, dtype=torch.bfloat16,
                   device="cuda", max_cached_blocks=512).validate()
r = ModelRunner(cfg, 0)
rng = np.random.default_rng(0)
lin_time = [0.0, 0]
orig = F.linear


def timed_linear(*a, **k):
    t = time.perf_counter()
    out = orig(*a, **k)
    lin_time[0] += time.perf_counter() - t
    lin_time[1] += 1
    return out


for T in (512, 2048, 8192):
    bm = BlockManager(512, 256)
    seqs = [Sequence(rng.integers(0, model.vocab_size, size=512).tolist(),
                     SamplingParams(temperature=0.5, max_tokens=4), block_size=256)
            for _ in range(T // 512)]
    for s in seqs:
        bm.allocate(s)
    batch = r.prepare_prefill(seqs)
    batch.step = 1
    with torch.inference_mode():
        for _ in range(3):
            r.execute(batch)
        torch.cuda.synchronize()
        F.linear = timed_linear
        lin_time[:] = [0.0, 0]
        n = 10
        t0 = time.perf_counter()
        host = 0.0
        for _ in range(n):
            ta = time.perf_counter()
            t = r._upload_prefill(batch)
            from minivllm_amd.utils.context import reset_context, set_context
            set_context(True, cu_seqlens_q=t["cu_seqlens_q"], cu_seqlens_k=t["cu_seqlens_k"],
                        max_seqlen_q=batch.max_seqlen_q, max_seqlen_k=batch.max_seqlen_k,
                        slot_mapping=t["slot_mapping"], block_tables=t["block_tables"],
                        positions=t["positions"], num_seqs=batch.num_seqs, has_prefix=False)
            tok = r._forward_and_sample(t)
            reset_context()
            host += time.perf_counter() - ta
            tok.tolist()
        total = time.perf_counter() - t0
        F.linear = orig
    print(f"{name} T={T}: step {total / n * 1e3:.2f} ms, host launch part {host / n * 1e3:.2f} ms, "
          f"of which F.linear {lin_time[0] / n * 1e3:.2f} ms over {lin_time[1] // n} calls "
          f"({lin_time[0] / lin_time[1] * 1e6:.1f} us each)", flush=True)
