"""Vocab-parallel embedding and LM head.

Reference behaviour (src/myvllm/layers/embedding_head.py:12-77): vocabulary rows
are sharded across TP ranks; the embedding masks out-of-shard ids, looks up
locally and all-reduces; the head picks each sequence's last prefill token and
gathers the logit shards on rank 0.  Differences: the vocabulary is padded to a
multiple of the TP degree (the reference demo's 151643 is odd and cannot be
sharded), and besides the gather-to-rank-0 path the head offers
``local_logits`` so the sampler can reduce (max, argmax) pairs instead of moving
``[B, V]`` logits over xGMI.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from minivllm_amd.parallel import group as pg
from minivllm_amd.utils.context import get_context


def padded_vocab(vocab_size: int, tp_size: int, multiple: int = 64) -> int:
    m = tp_size * multiple
    return (vocab_size + m - 1) // m * m


class VocabParallelEmbedding(nn.Module):
    def __init__(self, num_embeddings: int, embedding_dim: int,
                 tp_rank: Optional[int] = None, tp_size: Optional[int] = None):
        super().__init__()
        self.tp_rank = pg.tp_rank() if tp_rank is None else tp_rank
        self.tp_size = pg.tp_size() if tp_size is None else tp_size
        self.num_embeddings = num_embeddings
        self.num_embeddings_padded = (padded_vocab(num_embeddings, self.tp_size)
                                      if self.tp_size > 1 else num_embeddings)
        self.num_embeddings_per_partition = self.num_embeddings_padded // self.tp_size
        self.vocab_start = self.tp_rank * self.num_embeddings_per_partition
        self.vocab_end = self.vocab_start + self.num_embeddings_per_partition
        self.embedding_dim = embedding_dim
        self.weight = nn.Parameter(
            torch.zeros(self.num_embeddings_per_partition, embedding_dim), requires_grad=False)
        self.weight.weight_loader = self.weight_loader

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor, *args):
        rows = min(self.vocab_end, loaded_weight.size(0)) - self.vocab_start
        param.data.zero_()
        if rows > 0:
            param.data[:rows].copy_(loaded_weight.narrow(0, self.vocab_start, rows))

    def forward(self, input_ids: torch.Tensor) -> torch.Tensor:
        if self.tp_size == 1 and not pg.collectives_active():
            return F.embedding(input_ids, self.weight)
        mask = (input_ids >= self.vocab_start) & (input_ids < self.vocab_end)
        local = (input_ids - self.vocab_start) * mask
        out = F.embedding(local, self.weight) * mask.unsqueeze(-1).to(self.weight.dtype)
        return pg.tp_all_reduce(out)


class ParallelLMHead(VocabParallelEmbedding):
    def select_last(self, hidden: torch.Tensor) -> torch.Tensor:
        ctx = get_context()
        if ctx.is_prefill and ctx.cu_seqlens_q is not None:
            last = (ctx.cu_seqlens_q[1:] - 1).long()
            hidden = hidden.index_select(0, last)
        return hidden

    def local_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        """[B, V/tp] logits of this rank's vocabulary shard."""
        return F.linear(self.select_last(hidden), self.weight)

    def forward(self, hidden: torch.Tensor) -> Optional[torch.Tensor]:
        logits = self.local_logits(hidden)
        if self.tp_size > 1 or pg.collectives_active():
            logits = pg.tp_gather(logits, dst=0, dim=-1)
            if logits is None:
                return None
        return logits[..., :self.num_embeddings]
