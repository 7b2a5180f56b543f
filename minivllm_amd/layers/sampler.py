"""Token sampler.

The reference draws from ``softmax(logits / T)`` with the exponential-race trick
``argmax(probs / Exp(1))`` inside a torch.compile'd function and forbids greedy
decoding (reference: src/myvllm/layers/sampler.py:5-19,
sampling_parameters.py:10-11).  ``argmax(p_i / E_i)`` equals
``argmax(logit_i / T - log E_i)``, so the HIP kernel makes ONE pass over the
logits (no softmax normalisation) with a counter-based random stream; rows with
temperature 0 take the plain arg-max.

With tensor parallelism every rank samples on its own vocabulary shard and only
``(best value, best index)`` pairs cross the interconnect.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from minivllm_amd import ops


class SamplerLayer(nn.Module):
    def __init__(self, seed: int = 0):
        super().__init__()
        self.seed = seed
        self.step = 0
        self._cpu_gen: Optional[torch.Generator] = None

    def forward(self, logits: torch.Tensor, temperatures: torch.Tensor) -> torch.Tensor:
        self.step += 1
        if logits.is_cuda:
            return ops.sample(logits, temperatures, seed=self.seed, step=self.step)
        if self._cpu_gen is None:
            self._cpu_gen = torch.Generator(device="cpu")
            self._cpu_gen.manual_seed(self.seed)
        return ops.sample(logits, temperatures, generator=self._cpu_gen)


Sampler = SamplerLayer
