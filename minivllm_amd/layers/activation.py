"""SwiGLU gate: ``silu(x[..., :I]) * x[..., I:]`` (reference:
src/myvllm/layers/activation.py:6-18, there a torch.compile'd function; here one
vectorised HIP kernel on GPU)."""
import torch
import torch.nn as nn

from minivllm_amd import ops


class SiluAndMul(nn.Module):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.silu_and_mul(x)
