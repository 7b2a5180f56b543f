from .activation import SiluAndMul
from .attention import Attention, store_kvcache, flash_attention_prefill, paged_attention_decode
from .embedding_head import ParallelLMHead, VocabParallelEmbedding
from .layernorm import LayerNorm, RMSNorm
from .linear import (LinearBase, ReplicatedLinear, ColumnParallelLinear,
                     MergedColumnParallelLinear, QKVColumnParallelLinear, QKVParallelLinear,
                     RowParallelLinear)
from .rotary_embedding import RotaryEmbedding, get_rope
from .sampler import SamplerLayer, Sampler
