#!/usr/bin/env python3
"""Does splitting a decode batch into two half-batches on two HIP streams pay?  Attention is
HBM-bound and the M = 256 projections are bound by the on-chip load path, so one half's
attention could overlap the other half's GEMMs.  Compares one graph of the whole batch with one
graph that runs the two halves' decoder layers on two streams (LM head + sampling joint)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from minivllm_amd.config import EngineConfig, model_preset
from minivllm_amd.engine.model_runner import ModelRunner
from minivllm_amd.utils.context import reset_context, set_context

B, CTX, BS = int(os.environ.get("B", 256)), int(os.environ.get("CTX", 1024)), 256
model = model_preset(os.environ.get("MODEL", "qwen3-8b"))
cfg = EngineConfig(model=model, max_num_sequences=B, max_num_batched_tokens=16384, block_size=BS,
                   eos=-1, enforce_eager=True, max_model_length=2048, gpu_memory_utilization=0.85,
                   dtype=torch.bfloat16, device="cuda").validate()
r = ModelRunner(cfg, 0)
nblk = (CTX + BS - 1) // BS
host = r._dec_host.numpy()


def put(name, vals):
    off, _ = r._dec_off[name]
    host[off:off + len(vals)] = vals


rng = np.random.default_rng(0)
put("input_ids", rng.integers(0, model.vocab_size, B).astype(np.int32))
put("positions", np.full(B, CTX - 1, np.int32))
put("context_lens", np.full(B, CTX, np.int32))
put("temperatures", np.full(B, 0.6, np.float32).view(np.int32))
tables = np.full((B, r.max_blocks_per_seq), -1, np.int32)
for i in range(B):
    tables[i, :nblk] = np.arange(i * nblk, (i + 1) * nblk)
put("slot_mapping", (tables[:, nblk - 1] * BS + (CTX - 1) % BS).astype(np.int32))
off, _ = r._dec_off["block_tables"]
host[off:off + tables.size] = tables.reshape(-1)
r._dec_dev.copy_(r._dec_host)
full = r._decode_tensors(B)


def part(lo, hi):
    return {k: (v if k == "step" else v[lo:hi]) for k, v in full.items()}


def layers(t, n):
    set_context(False, slot_mapping=t["slot_mapping"], context_lens=t["context_lens"],
                block_tables=t["block_tables"], positions=t["positions"],
                max_context_len=cfg.max_model_length, num_seqs=n)
    h = r.model(t["input_ids"], t["positions"])
    reset_context()
    return h


def whole():
    return r._sample(layers(full, B), full["temperatures"], full["step"])


side = torch.cuda.Stream()


def halves():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    ha = layers(part(0, B // 2), B // 2)
    with torch.cuda.stream(side):
        hb = layers(part(B // 2, B), B - B // 2)
    cur.wait_stream(side)
    return r._sample(torch.cat([ha, hb]), full["temperatures"], full["step"])


def capture(fn):
    with torch.inference_mode():
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
    return g, out


def timeit(g, n=30):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


g1, o1 = capture(whole)
g2, o2 = capture(halves)
for _ in range(2):
    print(f"B={B} ctx={CTX}: one stream {timeit(g1):.3f} ms   two half-batches on two streams "
          f"{timeit(g2):.3f} ms", flush=True)
g1.replay(); g2.replay(); torch.cuda.synchronize()
print("same tokens:", float((o1 == o2).float().mean()))
