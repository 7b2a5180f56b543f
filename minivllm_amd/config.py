"""Typed configuration for the engine and the model.

The reference passes one flat ``dict`` from ``main.py`` to ``LLMEngine`` and
``ModelRunner`` with no validation (reference: main.py:15-41,
src/myvllm/engine/llm_engine.py:14-22, src/myvllm/engine/model_runner.py:17-43).
Here the same keys (including the duplicate spellings the reference uses,
``max_num_batched_tokens``/``max_num_batch_tokens`` and
``max_num_sequences``/``max_num_seqs``) are parsed once into two dataclasses so
that every component reads ONE value for ``block_size`` and friends.
"""
from __future__ import annotations

from dataclasses import dataclass, field, fields, asdict
from typing import Any, Dict, Optional

import torch


_DTYPES = {
    "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
    "fp16": torch.float16, "float16": torch.float16, "half": torch.float16,
    "fp32": torch.float32, "float32": torch.float32, "float": torch.float32,
}


def parse_dtype(x) -> torch.dtype:
    if isinstance(x, torch.dtype):
        return x
    try:
        return _DTYPES[str(x).lower()]
    except KeyError:
        raise ValueError(f"unknown dtype {x!r}; expected one of {sorted(_DTYPES)}")


@dataclass
class ModelConfig:
    """Decoder hyper-parameters (reference keys: model_runner.py:28-43).  The reference builds
    Qwen3 only; the same pre-norm GQA + SwiGLU decoder covers Qwen2.5 (QKV bias, no q/k norm)
    and Llama 3 (neither), which differ in ``qkv_bias`` / ``qk_norm`` / ``rope_scaling``."""
    vocab_size: int = 151936
    hidden_size: int = 1024
    num_heads: int = 16
    num_kv_heads: int = 8
    head_dim: int = 128
    intermediate_size: int = 3072
    num_layers: int = 28
    rms_norm_epsilon: float = 1e-6
    qkv_bias: bool = False
    ffn_bias: bool = False
    base: float = 1_000_000.0          # RoPE theta
    max_position: int = 40960
    scale: float = 1.0                 # extra multiplier on 1/sqrt(head_dim)
    tie_word_embeddings: bool = True
    name: str = "qwen3-custom"
    # per-head RMSNorm on q and k before RoPE.  None = the reference's rule: on unless the QKV
    # projection has a bias (qwen3.py:48,104-106)
    qk_norm: Optional[bool] = None
    # HF-style RoPE frequency scaling, e.g. {"rope_type": "llama3", "factor": 8.0, ...}
    rope_scaling: Optional[Dict[str, Any]] = None

    @property
    def use_qk_norm(self) -> bool:
        return (not self.qkv_bias) if self.qk_norm is None else self.qk_norm

    def num_parameters(self) -> int:
        h, i, v = self.hidden_size, self.intermediate_size, self.vocab_size
        qkv = h * (self.num_heads + 2 * self.num_kv_heads) * self.head_dim
        o = self.num_heads * self.head_dim * h
        mlp = 3 * h * i
        per_layer = qkv + o + mlp + 2 * h + (2 * self.head_dim if self.use_qk_norm else 0)
        emb = v * h * (1 if self.tie_word_embeddings else 2)
        return per_layer * self.num_layers + emb + h

    def kv_bytes_per_token(self, dtype: torch.dtype, tp: int = 1) -> int:
        itemsize = torch.empty((), dtype=dtype).element_size()
        return 2 * self.num_layers * (self.num_kv_heads // tp) * self.head_dim * itemsize


# Public Qwen3 shapes.  head_dim is 128 for every size (not hidden/num_heads).
QWEN3_PRESETS: Dict[str, Dict[str, Any]] = {
    "qwen3-0.6b": dict(hidden_size=1024, num_layers=28, num_heads=16, num_kv_heads=8, head_dim=128,
                       intermediate_size=3072, vocab_size=151936, tie_word_embeddings=True),
    "qwen3-1.7b": dict(hidden_size=2048, num_layers=28, num_heads=16, num_kv_heads=8, head_dim=128,
                       intermediate_size=6144, vocab_size=151936, tie_word_embeddings=True),
    "qwen3-4b": dict(hidden_size=2560, num_layers=36, num_heads=32, num_kv_heads=8, head_dim=128,
                     intermediate_size=9728, vocab_size=151936, tie_word_embeddings=True),
    "qwen3-8b": dict(hidden_size=4096, num_layers=36, num_heads=32, num_kv_heads=8, head_dim=128,
                     intermediate_size=12288, vocab_size=151936, tie_word_embeddings=False),
    "qwen3-14b": dict(hidden_size=5120, num_layers=40, num_heads=40, num_kv_heads=8, head_dim=128,
                      intermediate_size=17408, vocab_size=151936, tie_word_embeddings=False),
    "qwen3-32b": dict(hidden_size=5120, num_layers=64, num_heads=64, num_kv_heads=8, head_dim=128,
                      intermediate_size=25600, vocab_size=151936, tie_word_embeddings=False),
    # same decoder, other families (public configurations)
    "llama-3.1-8b": dict(hidden_size=4096, num_layers=32, num_heads=32, num_kv_heads=8, head_dim=128,
                         intermediate_size=14336, vocab_size=128256, tie_word_embeddings=False,
                         qk_norm=False, base=500000.0, max_position=131072, rms_norm_epsilon=1e-5,
                         rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0,
                                           high_freq_factor=4.0,
                                           original_max_position_embeddings=8192)),
    "qwen2.5-7b": dict(hidden_size=3584, num_layers=28, num_heads=28, num_kv_heads=4, head_dim=128,
                       intermediate_size=18944, vocab_size=152064, tie_word_embeddings=False,
                       qkv_bias=True, qk_norm=False, base=1000000.0, max_position=32768),
    # tiny shapes for tests / smoke runs
    "qwen3-tiny": dict(hidden_size=128, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=64,
                       intermediate_size=256, vocab_size=512, tie_word_embeddings=True,
                       max_position=512),
}


def model_preset(name: str, **overrides) -> ModelConfig:
    key = name.lower()
    if key not in QWEN3_PRESETS:
        raise KeyError(f"unknown model preset {name!r}; have {sorted(QWEN3_PRESETS)}")
    kw = dict(QWEN3_PRESETS[key])
    kw.update(overrides)
    return ModelConfig(name=key, **kw)


_MODEL_KEYS = {f.name for f in fields(ModelConfig)}


@dataclass
class EngineConfig:
    """Everything the engine, scheduler and runner need (reference keys: main.py:15-41)."""
    model: ModelConfig = field(default_factory=ModelConfig)
    # scheduler budgets
    max_num_sequences: int = 256
    max_num_batched_tokens: int = 16384
    block_size: int = 256
    # None => derive the number of KV blocks from free device memory (the
    # reference keeps a separate, unrelated 'max_cached_blocks': llm_engine.py:18)
    max_cached_blocks: Optional[int] = None
    eos: int = 151645
    # runtime
    world_size: int = 1
    enforce_eager: bool = False
    max_model_length: int = 4096
    gpu_memory_utilization: float = 0.9
    model_name_or_path: Optional[str] = None
    dtype: torch.dtype = torch.bfloat16
    device: str = "auto"               # 'auto' | 'cuda' | 'cpu'
    seed: int = 0
    # distributed rendezvous (the reference hard-codes tcp://localhost:12345, model_runner.py:24)
    master_addr: str = "127.0.0.1"
    master_port: int = 29512
    dist_backend: str = "auto"         # 'auto' | 'nccl' | 'gloo'
    # decode graphs
    graph_batch_sizes: Optional[list] = None
    # Prefill steps replay hipGraphs too (an eager step costs ~6 ms of host launches however
    # small it is).  One graph per (token bucket, sequence-count bucket): the attention kernel
    # takes its work from a list in device memory, so a graph's launch shapes depend on the
    # step's budgets only.  Graphs for steps of at most prefill_graph_max_tokens new tokens and
    # prefill_graph_max_seqs sequences are captured at start-up; with prefill_graph_lazy every
    # other prefill step shape - mixed prefill + decode steps included, whose decode rows take
    # the first slots of the batch - is captured when it occurs for the
    # prefill_graph_capture_after-th time (that step runs eagerly on the graph's static buffers
    # and is then recorded; a capture costs about two steps of its shape, so shapes that never
    # come back are not worth one), up to prefill_graph_max_graphs graphs.
    # prefill_graph_max_tokens = 0: always launch eagerly.
    prefill_graph_max_tokens: int = 1024
    prefill_graph_max_seqs: int = 8
    prefill_graph_lazy: bool = True
    prefill_graph_max_graphs: int = 96
    prefill_graph_capture_after: int = 2

    def prefill_graph_limits(self) -> tuple:
        """(max tokens, max sequences) of the prefill graphs captured at start-up."""
        return min(self.prefill_graph_max_tokens, 4096), max(1, self.prefill_graph_max_seqs)
    log_stats: bool = False
    tuned_gemms: bool = True           # load the shipped TunableOp table for library GEMMs
    checkpoint_path: Optional[str] = None
    # first GPU of this engine's tensor-parallel group when LOCAL_RANK is not set by a launcher
    # (data-parallel replicas started from one process each take their own range of devices)
    device_base: int = 0
    # GPU only: at start-up, time the hand-written weight-streaming GEMM against the library on
    # this model's own projections (1-16 rows) and use whichever is faster per shape
    tune_decode_gemms: bool = True
    # GPU only: at start-up, time the hand-written 256 x 256-tile GEMM (csrc/big_gemm.hip; with
    # the SwiGLU epilogue for gate_up) against the library (+ activation kernel) on this model's
    # projections at prefill-sized row counts and use whichever is faster per shape and row bucket
    tune_prefill_gemms: bool = True
    # where the outcome of those start-up measurements is kept, so that a second start of the
    # same engine on the same GPU type loads the table instead of timing again (same kernels,
    # hence the same roundings, from one start to the next): a JSON file, "auto" =
    # ~/.cache/minivllm_amd/gemm_choices.json (env MINIVLLM_GEMM_CHOICES overrides), "" = off.
    # Under tensor parallelism rank 0's table is what every rank uses.
    gemm_choice_cache: str = "auto"
    # GPU: multiply each layer norm's scale vector into the weights of the projection behind it
    # at load time (same math).  The normalisation left in front of that projection is then a
    # per-row factor, which (a) the prefill GEMM applies in its epilogue from row sums of squares
    # that the preceding o / down GEMM accumulated while adding the residual stream
    # (csrc/big_gemm.hip: no norm passes over [T, hidden] in a prefill layer) and (b) lets 1-2 row
    # decode batches run (residual add + norm + projection) as one kernel
    # (csrc/skinny_norm_gemm.hip).  None = on for single-rank GPU engines.
    fold_norm_weights: Optional[bool] = None
    # let a prompt longer than the step's token budget be prefetched over several steps (the
    # reference can never schedule such a prompt)
    enable_chunked_prefill: bool = True
    # > 0: while sequences are decoding, start a prefill step only when at least this many prompt
    # tokens can be admitted at once (0 = the reference policy: every newcomer interrupts decode)
    min_prefill_tokens: int = 0
    # online serving: with min_prefill_tokens set, a newcomer that alone would make a small prefill
    # step is held for at most this long so that requests arriving close together share one step
    # (0 = only defer while sequence slots, not arrivals, are what limits the step)
    max_prefill_delay_ms: float = 0.0
    # let the running sequences ride along with every prefill step (one decode row each, behind
    # the prompt rows) instead of standing still while newcomers are prefetched; the reference
    # never mixes the two phases (scheduler.py:29-63)
    enable_mixed_batches: bool = False
    # "auto": the KV cache holds values of the model dtype.  "fp8": OCP e4m3, one byte per value
    # (gfx950 converts in hardware): twice the tokens per GB and half the bytes the decode
    # attention kernel streams, at 3 mantissa bits per cached value.  Off by default: it changes
    # the numerics of attention.
    kv_cache_dtype: str = "auto"
    # all-reduce decode-sized tensors with the peer-to-peer kernel (parallel/custom_allreduce.py)
    # instead of RCCL; larger tensors still go to RCCL.  Also switched on by MINIVLLM_CUSTOM_AR=1.
    custom_all_reduce: bool = False
    # steady-state decode: enqueue step n + 1 (inputs derived on the device from step n's
    # samples) before reading step n's tokens back, so the GPU does not idle while the host
    # post-processes; single-rank engines with graph-replayed decode steps only
    # (ModelRunner._can_prelaunch; MINIVLLM_NO_PRELAUNCH=1 switches it off)
    decode_prelaunch: bool = True
    # ... also across KV-block edges (a sequence whose next token starts a new block is given
    # that block when the step is launched ahead, and a block the step completes has its hash
    # published when the step is collected) and while requests wait that cannot be admitted
    # (every slot taken, blocks short, a small prefill step held back).  Off: only steps that
    # touch no block edge, and only while nobody waits - with mixed context lengths or a
    # saturated engine that is a minority of the steps (MINIVLLM_NO_EDGE_PRELAUNCH=1: off, for A/B runs)
    decode_prelaunch_edges: bool = True
    # consecutive pure prefill steps: admit, prepare and (when the step's graph exists) enqueue
    # step n + 1 before reading step n's tokens back (Scheduler.schedule_prefill_ahead,
    # ModelRunner.run(next_prefill=...)); MINIVLLM_NO_PRELAUNCH=1 switches it off as well
    prefill_prelaunch: bool = True
    # tuning knobs of the multi-GPU path that have no measurement on real xGMI links behind them
    # yet (None = module default, also settable through the environment, see the modules):
    # size above which the peer-to-peer all-reduce switches from one-shot to two-shot, how long
    # one of its kernels waits for a peer, and the chunking of the row-parallel GEMM whose
    # all-reduce is pipelined against the next chunk's GEMM (parallel/overlap.py)
    car_one_shot_max_bytes: Optional[int] = None
    car_timeout_ms: Optional[int] = None
    overlap_min_chunk_tokens: Optional[int] = None
    overlap_max_chunks: Optional[int] = None
    # tensor parallelism: from this many rows on, a row-parallel output that feeds a norm is
    # reduced as reduce-scatter -> norm on the rank's rows -> all-gather (parallel/group.py;
    # None = 512, a huge value switches the route off)
    scatter_norm_min_rows: Optional[int] = None
    # testing aid: with world_size 1 still create a (single-rank) process group and run every
    # tensor-parallel collective through it, graph capture included
    force_tp_group: bool = False
    # shard rehearsal (measurement aid, one process, one GPU): build rank ``tp_shard_rank``'s
    # shard of a ``tp_shard``-way tensor-parallel engine - per-rank weight shapes, per-rank KV
    # heads, every collective issued through a single-member process group (real RCCL nodes
    # inside the decode graphs, nobody at the other end).  A step then costs what ONE rank of the
    # sharded engine computes and launches: the floor under the step time of a real N-GPU run.
    # Generated tokens are meaningless (partial sums are never completed).  0 = off.
    tp_shard: int = 0
    tp_shard_rank: int = 0
    # number of engine replicas behind engine.dp_engine.DataParallelEngine (1 = none)
    data_parallel_size: int = 1

    # --- derived ---------------------------------------------------------
    def resolved_device(self) -> str:
        if self.device == "auto":
            return "cuda" if torch.cuda.is_available() else "cpu"
        return self.device

    @property
    def tp_degree(self) -> int:
        """Number of ways the model is sharded (= world_size, or ``tp_shard`` in a rehearsal)."""
        return self.tp_shard if self.tp_shard > 1 else self.world_size

    def validate(self) -> "EngineConfig":
        m = self.model
        if self.tp_shard > 1 and self.world_size != 1:
            raise ValueError("tp_shard rehearses ONE rank of a sharded engine: world_size must be 1")
        if self.tp_shard > 1 and not (0 <= self.tp_shard_rank < self.tp_shard):
            raise ValueError("tp_shard_rank must be in [0, tp_shard)")
        if self.block_size <= 0 or (self.block_size & (self.block_size - 1)):
            raise ValueError(f"block_size must be a power of two, got {self.block_size}")
        if m.num_heads % m.num_kv_heads:
            raise ValueError("num_heads must be a multiple of num_kv_heads")
        tp = self.tp_degree
        if m.num_heads % tp or m.num_kv_heads % tp:
            raise ValueError(
                f"tensor-parallel degree {tp} must divide num_heads={m.num_heads} and "
                f"num_kv_heads={m.num_kv_heads} (KV-head replication is not supported)")
        if m.intermediate_size % tp:
            raise ValueError("the tensor-parallel degree must divide intermediate_size")
        if m.head_dim not in (64, 128) and self.resolved_device() == "cuda":
            raise ValueError("HIP attention kernels support head_dim 64 and 128")
        if self.max_model_length > m.max_position:
            raise ValueError(
                f"max_model_length={self.max_model_length} exceeds the RoPE table "
                f"max_position={m.max_position}")
        if self.max_num_batched_tokens < 1 or self.max_num_sequences < 1:
            raise ValueError("scheduler budgets must be positive")
        if self.kv_cache_dtype not in ("auto", "fp8"):
            raise ValueError("kv_cache_dtype must be 'auto' or 'fp8'")
        if not (0.0 < self.gpu_memory_utilization <= 1.0):
            raise ValueError("gpu_memory_utilization must be in (0, 1]")
        return self

    @property
    def max_blocks_per_seq(self) -> int:
        return (self.max_model_length + self.block_size - 1) // self.block_size

    def to_dict(self) -> Dict[str, Any]:
        d = asdict(self)
        d["dtype"] = str(self.dtype).replace("torch.", "")
        return d

    # --- construction ----------------------------------------------------
    @classmethod
    def from_dict(cls, cfg: Dict[str, Any]) -> "EngineConfig":
        """Accept the reference's flat dict (main.py:15-41) or a nested one.

        Unknown keys raise, so typos do not silently fall back to defaults.
        """
        cfg = dict(cfg)
        if isinstance(cfg.get("model"), ModelConfig):
            model = cfg.pop("model")
        else:
            preset = cfg.pop("model_preset", None)
            mkw = {k: cfg.pop(k) for k in list(cfg) if k in _MODEL_KEYS}
            if isinstance(cfg.get("model"), dict):
                mkw.update(cfg.pop("model"))
            model = model_preset(preset, **mkw) if preset else ModelConfig(**mkw)
        # duplicate spellings used by the reference
        if "max_num_seqs" in cfg:
            v = cfg.pop("max_num_seqs")
            cfg.setdefault("max_num_sequences", v)
        if "max_num_batch_tokens" in cfg:
            v = cfg.pop("max_num_batch_tokens")
            # the reference uses this one only for the warm-up shape; the
            # scheduler budget key wins when both are present
            cfg.setdefault("max_num_batched_tokens", v)
        if "dtype" in cfg:
            cfg["dtype"] = parse_dtype(cfg["dtype"])
        known = {f.name for f in fields(cls)}
        unknown = set(cfg) - known
        if unknown:
            raise KeyError(f"unknown config keys: {sorted(unknown)}")
        out = cls(model=model, **cfg)
        # the reference demo sets max_position == max_model_length == 128; keep
        # whatever the user gave but make the pair consistent
        if out.max_model_length > out.model.max_position:
            out.max_model_length = out.model.max_position
        return out.validate()


def as_engine_config(cfg) -> EngineConfig:
    if isinstance(cfg, EngineConfig):
        return cfg.validate()
    if isinstance(cfg, dict):
        return EngineConfig.from_dict(cfg)
    raise TypeError(f"config must be a dict or EngineConfig, got {type(cfg)}")
