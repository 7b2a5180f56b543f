"""Per-request sampling parameters.

Superset of the reference's ``SamplingParams(temperature, max_tokens, ignore_eos)``
(reference: src/myvllm/sampling_parameters.py:4-11).  The reference forbids
greedy sampling; here ``temperature == 0`` selects arg-max decoding, which the
CPU plumbing configuration needs for token-exact comparisons.
"""
from dataclasses import dataclass, field
from typing import List


@dataclass
class SamplingParams:
    temperature: float = 1.0
    max_tokens: int = 64
    ignore_eos: bool = False
    # generation also ends (the token is kept) on any of these ids, e.g. a chat model's
    # end-of-turn token next to the engine-wide ``eos``
    stop_token_ids: List[int] = field(default_factory=list)
    # sample only among the top_k most likely tokens (0 = no limit) and within the smallest set
    # of tokens whose probabilities add up to top_p (1.0 = no limit); both are applied to the
    # temperature-scaled distribution.  The reference has neither (sampler.py:14-19).
    top_k: int = 0
    top_p: float = 1.0

    def __post_init__(self):
        if self.temperature < 0:
            raise ValueError("temperature must be >= 0 (0 selects greedy decoding)")
        if 0 < self.temperature < 1e-5:
            # tiny temperatures overflow logits / T in bf16; treat them as greedy
            self.temperature = 0.0
        if self.max_tokens < 1:
            raise ValueError("max_tokens must be >= 1")
        if self.top_k < 0:
            raise ValueError("top_k must be >= 0 (0 disables it)")
        if not 0.0 < self.top_p <= 1.0:
            raise ValueError("top_p must be in (0, 1]")

    @property
    def filtered(self) -> bool:
        return self.temperature > 0 and (self.top_k > 0 or self.top_p < 1.0)

    @property
    def greedy(self) -> bool:
        return self.temperature == 0.0
