"""RMSNorm with an optional fused residual add.

Same math as the reference's ``LayerNorm`` module (reference:
src/myvllm/layers/layernorm.py:3-28), which despite its name is an RMSNorm.  Here
the scale is a learnable parameter named ``weight`` (checkpoint compatible) and
``eps`` comes from the model configuration; the reference uses a fixed 1e-5 and a
non-learnable buffer of ones.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from minivllm_amd import ops


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size), requires_grad=False)
        # True once the scale vector has been multiplied into the weights of the projection
        # that consumes this norm's output (``weight`` is all ones from then on)
        self.folded = False

    @torch.no_grad()
    def fold_into(self, *linear_weights: torch.Tensor) -> None:
        """``norm(x) * g @ W.T == norm(x) @ (W * g).T``: move the scale vector into ``W`` (shape
        [out, in]) so that the normalisation left in front of the GEMM is a per-row scalar."""
        if self.folded:
            return
        g = self.weight.float()
        for w in linear_weights:
            w.copy_((w.float() * g[None, :]).to(w.dtype))
        self.weight.fill_(1.0)
        self.folded = True

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None
                ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        # ``x`` may be a projection's pending work (ops.partials): ops takes it over
        if residual is None:
            return ops.rms_norm(x, self.weight, self.eps)
        return ops.add_rms_norm(x, residual, self.weight, self.eps)


LayerNorm = RMSNorm      # the reference's class name
