#!/usr/bin/env python3
"""Graph-replayed vs eager prefill step of a small model: wall time per step (with the token
read-back, as the engine does) for a few (sequences, tokens) shapes, and the host time spent
before the replay is issued."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from minivllm_amd.config import EngineConfig, model_preset
from minivllm_amd.engine.block_manager import BlockManager
from minivllm_amd.engine.model_runner import ModelRunner
from minivllm_amd.engine.sequence import Sequence
from minivllm_amd.sampling_parameters import SamplingParams

name = os.environ.get("MODEL", "qwen3-0.6b")
model = model_preset(name)
cfg = EngineConfig(model=model, max_num_sequences=64, max_num_batched_tokens=16384, block_size=256,
                   eos=-1, enforce_eager=False, graph_batch_sizes=[1], max_model_length=2048,
                   dtype=torch.bfloat16, device="cuda", max_cached_blocks=512).validate()
r = ModelRunner(cfg, 0)
print("prefill graphs:", sorted(r.prefill_graphs), "max seqs", r.pg_max_seqs, flush=True)
rng = np.random.default_rng(0)
for nseq, plen in ((1, 512), (3, 470), (8, 500), (24, 160)):
    bm = BlockManager(512, 256)
    seqs = [Sequence(rng.integers(0, model.vocab_size, size=plen).tolist(),
                     SamplingParams(temperature=0.5, max_tokens=4), block_size=256)
            for _ in range(nseq)]
    for s in seqs:
        bm.allocate(s)
    res = {}
    for mode in ("graph", "eager"):
        with torch.inference_mode():
            saved = r.prefill_graphs
            if mode == "eager":
                r.prefill_graphs = {}
            for _ in range(3):
                r.run(seqs, True)
            torch.cuda.synchronize()
            n = 20
            t0 = time.perf_counter()
            for _ in range(n):
                r.run(seqs, True)
            res[mode] = (time.perf_counter() - t0) / n
            # the host part alone: batch building (same for both modes)
            t0 = time.perf_counter()
            for _ in range(n):
                r.prepare_prefill(seqs)
            res["prepare"] = (time.perf_counter() - t0) / n
            r.prefill_graphs = saved
    T = nseq * plen
    print(f"{name} {nseq} x {plen} = {T} tokens: graph {res['graph']*1e3:.2f} ms, eager "
          f"{res['eager']*1e3:.2f} ms, prepare_prefill {res['prepare']*1e3:.2f} ms", flush=True)
