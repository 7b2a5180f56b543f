"""Rotary position embedding table (NeoX / half-split convention).

The reference precomputes ``cos_sin_cache[max_position, head_dim]`` and applies
the rotation in a torch.compile'd function (reference:
src/myvllm/layers/rotary_embedding.py:48-83).  Here the module only owns the fp32
table; the rotation itself is fused with the per-head q/k RMSNorm and the KV-cache
store in one kernel (``ops.qk_norm_rope_store``).  ``forward`` keeps the
reference's standalone ``(positions, q, k) -> (q, k)`` behaviour.
"""
from __future__ import annotations

from functools import lru_cache

import torch
import torch.nn as nn

from minivllm_amd.ops import torch_ref


class RotaryEmbedding(nn.Module):
    def __init__(self, head_size: int, rotary_dim: int, max_position: int, base: float,
                 rope_scaling=None):
        """``rope_scaling``: HF-style dict (``rope_type`` linear / llama3) or None."""
        super().__init__()
        if rotary_dim != head_size:
            raise ValueError("partial rotary is not supported")
        self.head_size = head_size
        self.max_position = max_position
        self.base = base
        self.register_buffer("cos_sin_cache",
                             torch_ref.build_cos_sin_cache(head_size, max_position, base,
                                                           rope_scaling=rope_scaling),
                             persistent=False)

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor):
        """query [T,Hq,D] / key [T,Hkv,D] -> rotated copies."""
        cs = self.cos_sin_cache[positions.long()].float()
        cos, sin = cs.chunk(2, dim=-1)
        q = torch_ref.apply_rope(query.float(), cos, sin).to(query.dtype)
        k = torch_ref.apply_rope(key.float(), cos, sin).to(key.dtype)
        return q, k


@lru_cache(maxsize=4)
def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float) -> RotaryEmbedding:
    return RotaryEmbedding(head_size, rotary_dim, max_position, base)
