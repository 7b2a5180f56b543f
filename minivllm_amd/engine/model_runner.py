"""Per-rank model executor: owns the model shard, the paged KV cache, batch preparation,
hipGraph-captured decode steps and the worker side of the control channel.

Responsibilities mirror the reference's ``ModelRunner`` (reference:
src/myvllm/engine/model_runner.py:13-369): build the model, warm up, size and allocate the
KV cache from free device memory, turn scheduled sequences into flat tensors
(``prepare_prefill`` / ``prepare_decode``), run the model (eager or graph replay), sample
on rank 0, and serve ``call(method, *args)`` requests from rank 0 on worker ranks.

MI355X-first differences:
* the batch is prepared ONCE on rank 0 with vectorised numpy and shipped to workers as raw
  int32 arrays (no pickled ``Sequence`` objects, no per-token Python loops);
* every step uploads one pinned staging buffer with a single H2D copy; rotary positions are
  part of it (the reference recomputes them in every layer with a host sync);
* the KV cache is one allocation sized from ``hipMemGetInfo`` x utilisation and its block
  count is what the scheduler is built from; blocks are stored head-major so a (block,
  kv-head) pair is one contiguous slab;
* decode graphs contain the whole step including LM head and sampler, and the graph's
  inputs live at fixed offsets of one static device buffer;
* with tensor parallelism each rank samples on its vocabulary shard and only
  (value, index) pairs are exchanged.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass, field
from itertools import chain
from typing import Dict, List, Optional, Sequence as Seq

import numpy as np
import torch

from minivllm_amd.config import EngineConfig, as_engine_config
from minivllm_amd.engine.rpc import ShmBroadcastChannel
from minivllm_amd.engine.sequence import Sequence
from minivllm_amd.layers.sampler import SamplerLayer
from minivllm_amd.models.qwen3 import Qwen3ForCausalLM
from minivllm_amd.parallel import group as pg
from minivllm_amd.utils.context import reset_context, set_context
from minivllm_amd.utils.profiling import trace_range


@dataclass
class Batch:
    """Host-side description of one engine step (what crosses the control channel)."""
    is_prefill: bool
    num_seqs: int
    num_tokens: int
    max_seqlen_q: int = 0
    max_seqlen_k: int = 0
    max_context_len: int = 0
    has_prefix: bool = False
    step: int = 0
    tables_unchanged: bool = False      # decode: block tables equal the previous step's
    # mixed step: the FIRST num_decode_seqs sequences of a prefill-form batch contribute one
    # decode row each (rows 0 .. num_decode_seqs - 1), the num_prefill_tokens prompt rows follow
    num_decode_seqs: int = 0
    num_prefill_tokens: int = 0
    num_prefill_k_tokens: int = 0       # sum of the context lengths of the prefill sequences
    # decode step under tensor parallelism: rank 0 launches the step after this one ahead of
    # this one's read-back, and every worker has to enqueue that step too
    prelaunch: bool = False
    arrays: Dict[str, np.ndarray] = field(default_factory=dict)
    # set where the step executes: decode slots of the captured graph that ran a mixed step
    decode_slots: Optional[int] = None

    def scalars(self) -> dict:
        return dict(is_prefill=self.is_prefill, num_seqs=self.num_seqs, num_tokens=self.num_tokens,
                    max_seqlen_q=self.max_seqlen_q, max_seqlen_k=self.max_seqlen_k,
                    max_context_len=self.max_context_len, has_prefix=self.has_prefix,
                    step=self.step, tables_unchanged=self.tables_unchanged,
                    num_decode_seqs=self.num_decode_seqs,
                    num_prefill_tokens=self.num_prefill_tokens,
                    num_prefill_k_tokens=self.num_prefill_k_tokens, prelaunch=self.prelaunch)


_CHECK_BATCHES = os.environ.get("MINIVLLM_CHECK_BATCHES") == "1"


def _pad_block_tables(tables: List[List[int]]) -> np.ndarray:
    """[B, max blocks] int32, rows padded with -1, without a Python-level loop over rows."""
    n = np.fromiter(map(len, tables), np.int64, len(tables))
    flat = np.fromiter(chain.from_iterable(tables), np.int32, int(n.sum()))
    out = np.full((len(tables), int(n.max())), -1, np.int32)
    starts = np.cumsum(n) - n
    out[np.repeat(np.arange(len(tables)), n), np.arange(flat.size) - np.repeat(starts, n)] = flat
    return out


def prefill_token_bucket(num_tokens: int) -> int:
    """Token count a prefill step is padded to.  The GEMM library chooses (and PyTorch caches)
    a kernel per problem shape, and an unseen shape costs milliseconds of host time on its
    first use; with arbitrary token counts nearly every prefill step of a busy server brings
    four new shapes and small steps end up bound by the host.  Rounding up to a few dozen
    distinct sizes (at most ~6 % extra rows) makes the shapes repeat."""
    g = 64 if num_tokens <= 1024 else 256 if num_tokens <= 4096 else 512
    return (num_tokens + g - 1) // g * g


def default_graph_batch_sizes(max_bs: int) -> List[int]:
    """Same bucket policy as the reference: 1,2,4,8 then multiples of 16
    (reference: model_runner.py:334)."""
    sizes = [b for b in (1, 2, 4, 8) if b <= max_bs] + list(range(16, max_bs + 1, 16))
    if max_bs not in sizes:
        sizes.append(max_bs)
    return sorted(set(sizes))


class ModelRunner:
    P2P_CHECK_EVERY = 64          # worker ranks look at their all-reduce status this often

    def __init__(self, config, rank: int = 0, event=None, spawn_loop: bool = True):
        """``event`` is accepted for signature compatibility with the reference and unused:
        the control channel is addressed by name.  A worker rank (rank > 0) enters
        ``loop()`` at the end of construction when ``spawn_loop`` is set."""
        self.cfg: EngineConfig = as_engine_config(config)
        cfg = self.cfg
        if cfg.world_size > 1 and os.environ.get("MINIVLLM_TP_CONSERVATIVE") == "1":
            # Fallback for a multi-GPU run whose first attempt failed (bench.py retries a failed
            # tensor-parallel phase this way): the newer multi-rank paths off - no decode step
            # launched ahead on the workers, plain all-reduce in front of every norm, prefill
            # graphs only for the shapes captured at start-up
            import dataclasses
            self.cfg = cfg = dataclasses.replace(cfg, decode_prelaunch=False, prefill_prelaunch=False,
                                                 prefill_graph_lazy=False,
                                                 scatter_norm_min_rows=1 << 30)
        self.rank = rank
        self.world_size = cfg.world_size       # processes of this engine
        self.tp = cfg.tp_degree                # ways the model is sharded (a shard rehearsal: > processes)
        self.block_size = cfg.block_size
        self.enforce_eager = cfg.enforce_eager
        dev_type = cfg.resolved_device()
        if dev_type == "cuda":
            local = (int(os.environ["LOCAL_RANK"]) if "LOCAL_RANK" in os.environ
                     else cfg.device_base + rank)
            self.device = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
            torch.cuda.set_device(self.device)
        else:
            self.device = torch.device("cpu")
        self.on_gpu = self.device.type == "cuda"
        self.dtype = cfg.dtype if self.on_gpu else (cfg.dtype if cfg.dtype != torch.float16
                                                    else torch.float32)

        self.kv_dtype = torch.float8_e4m3fn if cfg.kv_cache_dtype == "fp8" else self.dtype

        pg.init_tensor_parallel(rank, cfg.world_size, cfg.dist_backend, cfg.master_addr,
                                cfg.master_port, self.device, force_group=cfg.force_tp_group,
                                shard_of=cfg.tp_shard, shard_rank=cfg.tp_shard_rank)

        if (self.on_gpu and self.tp > 1
                and (cfg.custom_all_reduce or os.environ.get("MINIVLLM_CUSTOM_AR") == "1")):
            pg.enable_custom_all_reduce(self.device, one_shot_max_bytes=cfg.car_one_shot_max_bytes,
                                        timeout_ms=cfg.car_timeout_ms)
        pg.configure_scatter_norm(cfg.scatter_norm_min_rows)
        if cfg.overlap_min_chunk_tokens or cfg.overlap_max_chunks:
            from minivllm_amd.parallel import overlap
            overlap.configure(cfg.overlap_min_chunk_tokens, cfg.overlap_max_chunks)
        if self.on_gpu and cfg.tuned_gemms:
            from minivllm_amd.utils.gemm_tuning import enable_tuned_gemms
            enable_tuned_gemms()

        # ---- model shard -------------------------------------------------
        prev = torch.get_default_dtype()
        torch.set_default_dtype(self.dtype)
        try:
            with torch.device(self.device):
                self.model = Qwen3ForCausalLM(cfg.model)
        finally:
            torch.set_default_dtype(prev)
        self.model.eval()
        if cfg.checkpoint_path:
            from minivllm_amd.utils.loader import load_weights_from_checkpoint
            load_weights_from_checkpoint(self.model, cfg.checkpoint_path)
        else:
            self.model.init_random_weights(cfg.seed)
        fold = cfg.fold_norm_weights
        if fold is None:
            fold = self.tp == 1 and not pg.collectives_active()
        self.norms_folded = bool(self.on_gpu and fold)
        if self.norms_folded:
            self.model.fold_norm_weights()
        self.sampler = SamplerLayer(seed=cfg.seed)
        self.step_counter = 0
        self._decode_prev = None          # (batch list object, arrays) of the last decode step
        self._last_tokens: Optional[List[int]] = None
        self._arange_cache = np.arange(256)
        self.graphs: Dict[int, "torch.cuda.CUDAGraph"] = {}
        self.graph_pool = None
        self.kv_cache: Optional[torch.Tensor] = None
        self.num_kv_blocks = 0

        # ---- buffers -----------------------------------------------------
        self.max_bs = cfg.max_num_sequences
        self.max_blocks_per_seq = cfg.max_blocks_per_seq
        self._init_buffers()

        with torch.inference_mode():
            self.warmup_model()
            if self.on_gpu and (cfg.tune_decode_gemms or cfg.tune_prefill_gemms):
                self._choose_gemm_kernels()
            self.allocate_kv_cache()
            if self.on_gpu and not self.enforce_eager:
                self._capture_graphs_or_fall_back()

        # ---- control channel --------------------------------------------
        self.channel: Optional[ShmBroadcastChannel] = None
        if self.world_size > 1:
            name = f"minivllm_amd_{cfg.master_port}"
            if rank == 0:
                self.channel = ShmBroadcastChannel(name, self.world_size - 1, create=True)
                pg.tp_barrier()
            else:
                pg.tp_barrier()
                self.channel = ShmBroadcastChannel(name, self.world_size - 1, create=False)
                if spawn_loop:
                    self.loop()

    # ------------------------------------------------------- GEMM kernel choices
    def _gemm_choice_key(self) -> str:
        """Everything the start-up measurements depend on."""
        from minivllm_amd.ops import hip
        m, cfg = self.cfg.model, self.cfg
        return "|".join(str(v) for v in (
            torch.cuda.get_device_name(self.device), hip.kernel_library_stamp(),
            m.hidden_size, m.num_heads, m.num_kv_heads, m.head_dim, m.intermediate_size,
            f"tp{self.tp}", self.dtype, f"seqs{self.max_bs}", f"tok{cfg.max_num_batched_tokens}",
            f"dec{int(cfg.tune_decode_gemms)}", f"pre{int(cfg.tune_prefill_gemms)}",
            f"fold{int(self.norms_folded)}",
            f"p2p{int(pg.custom_all_reduce() is not None)}", f"coll{int(pg.collectives_active())}"))

    def _gemm_choice_path(self) -> Optional[str]:
        path = os.environ.get("MINIVLLM_GEMM_CHOICES", self.cfg.gemm_choice_cache)
        if path == "auto":
            path = os.path.join(os.path.expanduser("~"), ".cache", "minivllm_amd", "gemm_choices.json")
        return path or None

    def _choose_gemm_kernels(self) -> None:
        """Hand-written kernel or library, per projection and row bucket: loaded from the choice
        cache when this (GPU type, kernel build, model shard, limits) has been measured before,
        measured now otherwise (and saved).  With several ranks only rank 0 does either and
        broadcasts its table, so that every rank runs the same kernels."""
        import json
        from minivllm_amd.ops import hip
        self.gemm_choice_source = "measured"
        tables = None
        if self.rank == 0:
            path, key = self._gemm_choice_path(), self._gemm_choice_key()
            store = {}
            if path and os.path.exists(path):
                try:
                    with open(path) as f:
                        store = json.load(f)
                except (OSError, ValueError):
                    store = {}
            if key in store:
                hip.import_gemm_choices(store[key])
                self.gemm_choice_source = f"cache:{path}"
            else:
                hip.import_gemm_choices({})              # start from empty tables
                if self.cfg.tune_decode_gemms:
                    self._tune_decode_gemms()
                if self.cfg.tune_prefill_gemms:
                    self._tune_prefill_gemms()
                if path:
                    store[key] = hip.export_gemm_choices()
                    try:
                        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
                        tmp = f"{path}.{os.getpid()}.tmp"
                        with open(tmp, "w") as f:
                            json.dump(store, f)
                        os.replace(tmp, path)
                    except OSError:
                        pass                             # a read-only home: measure every time
            tables = hip.export_gemm_choices()
        if self.world_size > 1:
            box = [tables]
            torch.distributed.broadcast_object_list(box, src=0, group=pg.tp_group())
            if self.rank != 0:
                hip.import_gemm_choices(box[0])
                self.gemm_choice_source = "rank 0"
        self.gemm_choices = hip.export_gemm_choices()

    def _tune_decode_gemms(self) -> None:
        """Per projection and row bucket: hand-written weight-streaming kernel or library GEMM,
        decided by timing both on this model's own weights (before the decode graphs are
        captured, so the choice is baked into them)."""
        from minivllm_amd.ops import hip
        layers = list(self.model.model.layers)[:16]
        roles = {"qkv": ([l.self_attn.qkv_proj.weight for l in layers], False),
                 "o": ([l.self_attn.o_proj.weight for l in layers], False),
                 "gate_up": ([l.mlp.gate_up_proj.weight for l in layers], True),
                 "down": ([l.mlp.down_proj.weight for l in layers], False)}
        self.decode_gemm_choices = {}
        self.mid_gemm_choices = {}
        # o / down feed a residual add + norm.  On one rank their K-slice sum is folded into
        # that norm; under the peer-to-peer all-reduce into the staging pass of the fused
        # reduce + norm kernel; with RCCL between GEMM and norm the slices are summed first.
        sharded = self.tp > 1 or pg.collectives_active()
        fold_slices = not sharded or pg.custom_all_reduce() is not None
        # decode batches are at most max_num_sequences rows: tune the buckets that can occur
        rows = [b for b in hip.MID_ROW_BUCKETS if b // 2 < self.max_bs]
        for name, (ws, swiglu) in roles.items():
            self.decode_gemm_choices[name] = hip.tune_skinny_gemm(ws, swiglu)
            if name in ("o", "down") and fold_slices:
                # few column tiles: K may also be split over workgroups, the norm behind the
                # projection sums the slices (rows: library us, kernel us, split; 0 = library)
                self.decode_gemm_choices[name + "+norm"] = hip.tune_skinny_to_norm(ws)
            if self.max_bs > hip.SKINNY_MAX_ROWS:
                to_norm = name in ("o", "down") and fold_slices
                self.mid_gemm_choices[name] = hip.tune_mid_gemm(ws, swiglu, rows=rows,
                                                                to_norm=to_norm)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        if self.rank == 0 and self.cfg.log_stats:
            print(f"[runner] decode GEMM choices (rows: library us, kernel us, use kernel): "
                  f"{self.decode_gemm_choices}", flush=True)
            print(f"[runner] mid-batch GEMM choices (rows: library us, kernel us, (bn, splits, "
                  f"nt)): {self.mid_gemm_choices}", flush=True)

    def _tune_prefill_gemms(self) -> None:
        """Per projection and row bucket of a prefill step: the 256 x 256-tile kernel (gate_up:
        with the SwiGLU epilogue, which saves the activation kernel's pass over the [T, 2I]
        intermediate) or the library GEMM, decided by timing both on this model's weights."""
        from minivllm_amd.ops import hip
        layers = list(self.model.model.layers)[:4]
        roles = {"qkv": ([l.self_attn.qkv_proj.weight for l in layers], False),
                 "o": ([l.self_attn.o_proj.weight for l in layers], False),
                 "gate_up": ([l.mlp.gate_up_proj.weight for l in layers], True),
                 "down": ([l.mlp.down_proj.weight for l in layers], False)}
        top = max(self.cfg.max_num_batched_tokens, 512)
        rows = [b for b in hip.BIG_ROW_BUCKETS if b // 2 < top]
        self.prefill_gemm_choices = {name: hip.tune_big_gemm(ws, swiglu, rows=rows)
                                     for name, (ws, swiglu) in roles.items()}
        if self.norms_folded and self.tp == 1 and not pg.collectives_active():
            # o -> norm -> gate_up and down -> norm -> next layer's qkv as chains: residual add and
            # row sums of squares in the first GEMM's epilogue, the norm as a row factor in the
            # second one's, against GEMM + add/norm kernel + GEMM
            nxt = list(self.model.model.layers)[1:5]
            self.prefill_gemm_choices["o+norm"] = hip.tune_big_resid(
                roles["o"][0], roles["gate_up"][0], True, rows=rows, eps=self.cfg.model.rms_norm_epsilon)
            if nxt:
                self.prefill_gemm_choices["down+norm"] = hip.tune_big_resid(
                    roles["down"][0], [l.self_attn.qkv_proj.weight for l in nxt], False, rows=rows,
                    eps=self.cfg.model.rms_norm_epsilon)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        if self.rank == 0 and self.cfg.log_stats:
            print(f"[runner] prefill GEMM choices (rows: library us, kernel us, use kernel): "
                  f"{self.prefill_gemm_choices}", flush=True)

    # ================================================================ buffers
    def _init_buffers(self) -> None:
        cfg = self.cfg
        mb, MB = self.max_bs, self.max_blocks_per_seq
        if self.on_gpu:
            # scratch the decode kernel needs to exist before any graph is captured
            from minivllm_amd.ops import hip
            groups = -(-(cfg.model.num_heads // cfg.model.num_kv_heads) // 16)
            hip.decode_ticket_words(self.device, mb * max(1, cfg.model.num_kv_heads // self.tp)
                                    * groups)
        # decode: fixed layout so that graphs can bake the addresses in
        self._dec_off = {}
        off = 0
        for name, n in (("input_ids", mb), ("positions", mb), ("slot_mapping", mb),
                        ("context_lens", mb), ("temperatures", mb), ("misc", 4),
                        ("block_tables", mb * MB)):
            self._dec_off[name] = (off, n)
            off += (n + 3) // 4 * 4
        self._dec_words = off
        # prefill: input_ids, positions, slot_mapping (T each) + cu_q, cu_k (B+1) + temps (B)
        #          + misc + block tables
        self._pre_words = 0          # grown on demand
        pin = self.on_gpu
        # two pinned staging blocks, used alternately: with a decode step launched ahead of the
        # previous one's read-back (see ``run``), the host fills the block of step n + 1 while
        # the upload of step n may still be pending
        self._dec_hosts = [torch.zeros(self._dec_words, dtype=torch.int32, pin_memory=pin)
                           for _ in range(2)]
        self._dec_host = self._dec_hosts[0]
        self._dec_flip = 0
        # per block: (padded batch, num_seqs, version of the block tables) of its table image
        self._dec_tables_keys = [None, None]
        self._tables_version = 0
        self._tok_hosts = [torch.zeros(max(mb, 1), dtype=torch.int32, pin_memory=pin)
                           for _ in range(2)]
        self._tok_flip = 0
        self._spec = None                        # decode step launched ahead, not yet collected
        self._pf_spec = None                     # prefill step admitted ahead, not yet collected
        self._dec_dev = torch.zeros(self._dec_words, dtype=torch.int32, device=self.device)
        self._pre_host = None
        self._pre_dev = None
        # prefill steps replay hipGraphs as decode steps do: fixed layout, one static buffer.
        # Key = (token bucket, sequence-count bucket); small ones are captured at start-up, the
        # others the first time they occur (``_prefill_graph_key`` / ``_run_prefill_graph``)
        self.prefill_graphs: Dict[tuple, "torch.cuda.CUDAGraph"] = {}
        self._pg_out: Dict[tuple, torch.Tensor] = {}
        self._pg_logits: Dict[tuple, torch.Tensor] = {}    # static logits of each prefill graph
        self._pg_failed: set = set()                       # shapes whose capture failed: eager
        self._graph_logits: Dict[int, torch.Tensor] = {}   # ... and of each decode graph
        self._last_logits: Optional[torch.Tensor] = None
        self._last_step_logits: Optional[torch.Tensor] = None
        self.prefill_graph_replays = 0
        pg_max, self.pg_max_seqs = cfg.prefill_graph_limits()
        if not self.on_gpu or cfg.enforce_eager:
            pg_max = 0
        pg_max = int(os.environ.get("MINIVLLM_PREFILL_GRAPH_MAX", pg_max))
        pg_max = min(pg_max, 4096, cfg.max_num_batched_tokens)
        self.pg_max_seqs = min(self.pg_max_seqs, cfg.max_num_sequences)
        # = the prefill token buckets up to pg_max (captured at start-up)
        self._pg_sizes = sorted({prefill_token_bucket(t) for t in range(64, pg_max + 1, 64)
                                 if prefill_token_bucket(t) <= pg_max})
        self._pg_lazy = bool(pg_max and cfg.prefill_graph_lazy
                             and os.environ.get("MINIVLLM_NO_LAZY_PREFILL_GRAPHS") != "1")
        # sequence-count buckets: a graph serves any step with at most that many sequences (its
        # attention grid and LM-head rows are sized for the bucket)
        S = cfg.max_num_sequences
        self._pg_seq_buckets = sorted({min(self.pg_max_seqs, S), min(32, S), min(128, S), S})
        # decode-slot buckets of mixed steps (the riders' rows at the front of the batch)
        self._pg_dec_buckets = sorted({min(1 << i, S) for i in range(0, 12)})
        self._pg_seen: Dict[tuple, int] = {}
        self._pg_off = {}
        if self._pg_sizes:
            from minivllm_amd.ops import hip
            tmax = prefill_token_bucket(cfg.max_num_batched_tokens) if self._pg_lazy else self._pg_sizes[-1]
            smax = S if self._pg_lazy else self.pg_max_seqs
            self._pg_tmax, self._pg_smax, off = tmax, smax, 0
            for name, n in (("input_ids", tmax), ("positions", tmax), ("slot_mapping", tmax),
                            ("cu_seqlens_q", smax + 1), ("cu_seqlens_k", smax + 1),
                            ("temperatures", smax), ("misc", 4), ("context_lens", smax),
                            ("prefill_items", hip.prefill_item_slots(tmax, smax)),
                            ("block_tables", smax * MB)):
                self._pg_off[name] = (off, n)
                off += (n + 3) // 4 * 4
            # two pinned staging blocks used alternately, as for decode: with a prefill step
            # launched ahead the host fills step n + 1's block while step n's upload may be pending
            self._pg_hosts = [torch.zeros(off, dtype=torch.int32, pin_memory=True) for _ in range(2)]
            self._pg_host, self._pg_flip = self._pg_hosts[0], 0
            self._pg_dev = torch.zeros(off, dtype=torch.int32, device=self.device)

    def _ensure_prefill_buffers(self, words: int) -> None:
        if self._pre_host is None or self._pre_host.numel() < words:
            words = max(words, 3 * self.cfg.max_num_batched_tokens + 4096)
            self._pre_host = torch.zeros(words, dtype=torch.int32, pin_memory=self.on_gpu)
            self._pre_dev = torch.zeros(words, dtype=torch.int32, device=self.device)

    def _dec_view(self, buf: torch.Tensor, name: str, n: Optional[int] = None) -> torch.Tensor:
        off, size = self._dec_off[name]
        return buf[off:off + (size if n is None else n)]

    # ============================================================== RPC plumbing
    def call(self, method_name: str, *args):
        """Rank 0: run ``method`` here and on every worker (reference: model_runner.py:117-123)."""
        if self.world_size > 1 and self.rank == 0 and method_name != "run":
            self.channel.send(method_name, {"args": args})
        return getattr(self, method_name)(*args)

    def loop(self) -> None:
        """Worker main loop: execute broadcast requests until ``exit``."""
        from minivllm_amd.engine.rpc import ChannelClosed
        while True:
            try:
                method, scalars, arrays = self.channel.recv(self.rank - 1)
            except ChannelClosed as e:       # rank 0 died: nobody will ever send 'exit'
                print(f"[runner] rank {self.rank}: {e}; shutting down", flush=True)
                self.exit(peers_dead=True)
                raise SystemExit(3)
            if method == "run_batch":
                self._spec = None
                b = Batch(arrays=arrays, **scalars)
                with torch.inference_mode():
                    if b.prelaunch:              # rank 0 runs this step and launches the next
                        self._prelaunch_after(None, b, self._launch_decode(b))
                    else:
                        self.execute(b)
            elif method == "prelaunch_next":     # rank 0 keeps launching ahead: one more step
                with torch.inference_mode():
                    self._prelaunch_after(None, self._spec["batch"], self._spec["tokens"],
                                          (arrays or {}).get("block_tables"))
                if self._spec["batch"].step % self.P2P_CHECK_EVERY == 0:
                    pg.check_custom_all_reduce(synchronize=True)
            elif method == "exit":
                self.exit()
                break
            else:
                getattr(self, method)(*scalars.get("args", ()))

    def exit(self, peers_dead: bool = False) -> None:
        """``peers_dead``: a worker process is gone; tear down locally without anything that
        would wait for it (no barrier, no collective destroy, no device drain behind a
        collective that can never complete)."""
        if getattr(self, "_exited", False):
            return
        self._exited = True
        self._spec = None
        self._pf_spec = None
        if self.on_gpu and not peers_dead:
            torch.cuda.synchronize()
            try:
                pg.check_custom_all_reduce(synchronize=False)
            except RuntimeError as e:
                print(f"[runner] {e}", flush=True)
        # give the device memory back: another engine may be built in this process
        self.graphs.clear()
        self.prefill_graphs.clear()
        self._pg_out = {}
        self._graph_out = {}
        self._pg_logits, self._graph_logits, self._last_logits = {}, {}, None
        self._last_step_logits = None
        self.graph_pool = None
        for layer in self.model.model.layers:
            layer.self_attn.attn.k_cache = layer.self_attn.attn.v_cache = None
        self.kv_cache = None
        self.model = None
        self._dec_dev = self._pre_dev = None
        if self.on_gpu:
            torch.cuda.empty_cache()
        if self.channel is not None:
            if not peers_dead:
                try:                # readers detach before the writer unlinks the segment
                    pg.tp_barrier()
                except Exception:
                    pass
            self.channel.close()
            self.channel = None
        pg.destroy_tensor_parallel(peers_dead)

    # ================================================================= timing
    def timer_start(self) -> None:
        """Every rank: drain the device, meet at a barrier, start a wall clock."""
        if self.on_gpu:
            torch.cuda.synchronize()
        pg.tp_barrier()
        self._t0 = time.perf_counter()

    def timer_stop(self) -> float:
        """Every rank: drain, barrier, and return the MAX elapsed seconds over ranks."""
        if self.on_gpu:
            torch.cuda.synchronize()
        pg.tp_barrier()
        dt = time.perf_counter() - self._t0
        if self.world_size > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=self.device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=pg.tp_group())
            dt = float(t.item())
        return dt

    # ================================================================ warm-up
    def warmup_model(self) -> None:
        """One prefill of the largest batch the scheduler can emit, without a KV cache, so
        that peak activation memory is known before the cache is sized
        (reference: model_runner.py:129-137)."""
        if not self.on_gpu:
            return
        cfg = self.cfg
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        seq_len = min(cfg.max_model_length, cfg.max_num_batched_tokens)
        num_seqs = max(1, min(cfg.max_num_batched_tokens // seq_len, cfg.max_num_sequences))
        lens = [seq_len] * num_seqs
        T = sum(lens)
        cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        batch = Batch(is_prefill=True, num_seqs=num_seqs, num_tokens=T, max_seqlen_q=seq_len,
                      max_seqlen_k=seq_len, arrays=dict(
                          input_ids=np.zeros(T, np.int32),
                          positions=np.concatenate([np.arange(n) for n in lens]).astype(np.int32),
                          slot_mapping=np.full(T, -1, np.int32),
                          cu_seqlens_q=cu, cu_seqlens_k=cu,
                          temperatures=np.zeros(num_seqs, np.float32),
                          block_tables=np.zeros((num_seqs, 1), np.int32)))
        self.execute(batch, warmup=True)
        torch.cuda.synchronize()

    # ================================================================ KV cache
    def kv_block_bytes(self) -> int:
        m = self.cfg.model
        itemsize = torch.empty((), dtype=self.kv_dtype).element_size()
        return (2 * m.num_layers * self.block_size * (m.num_kv_heads // self.tp)
                * m.head_dim * itemsize)

    def allocate_kv_cache(self) -> None:
        cfg, m = self.cfg, self.cfg.model
        block_bytes = self.kv_block_bytes()
        if self.on_gpu:
            free, total = torch.cuda.mem_get_info()
            stats = torch.cuda.memory_stats()
            peak = stats.get("allocated_bytes.all.peak", 0)
            current = stats.get("allocated_bytes.all.current", 0)
            used = total - free
            # same accounting as the reference (model_runner.py:140-157): budget = total *
            # utilisation - what is in use now - the activation head-room seen in warm-up
            budget = int(total * cfg.gpu_memory_utilization) - used - max(peak - current, 0)
            num_blocks = budget // block_bytes
            if self.world_size > 1:        # every rank must agree on the block count
                t = torch.tensor([num_blocks], dtype=torch.int64, device=self.device)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN,
                                             group=pg.tp_group())
                num_blocks = int(t.item())
        else:
            budget = 256 << 20
            num_blocks = max(budget // block_bytes, 8)
        if not self.on_gpu:
            # no sequence set can use more than this many blocks at once
            num_blocks = min(num_blocks, cfg.max_num_sequences * (cfg.max_blocks_per_seq + 1))
        if cfg.max_cached_blocks is not None:
            num_blocks = min(num_blocks, cfg.max_cached_blocks)
        if num_blocks <= 0:
            raise RuntimeError("not enough device memory for a single KV-cache block")
        self.num_kv_blocks = int(num_blocks)
        hkv = m.num_kv_heads // self.tp
        # physical order [2, layer, block, kv_head, token, d]; layers see the logical
        # [block, token, kv_head, d] view of it
        self.kv_cache = torch.zeros(2, m.num_layers, self.num_kv_blocks, hkv, self.block_size,
                                    m.head_dim, dtype=self.kv_dtype, device=self.device)
        for i, layer in enumerate(self.model.model.layers):
            attn = layer.self_attn.attn
            attn.k_cache = self.kv_cache[0, i].permute(0, 2, 1, 3)
            attn.v_cache = self.kv_cache[1, i].permute(0, 2, 1, 3)
            attn.block_size = self.block_size

    # ============================================================ batch building
    def prepare_prefill(self, seqs: Seq[Sequence], num_decode: int = 0) -> Batch:
        """``num_decode``: the last that many sequences are running sequences that ride along
        with one decode token each (mixed step); their row is the newest token, whose key is
        everything before it in the paged cache.  Inside the batch the riders come FIRST (rows
        and sequence slots 0 .. num_decode - 1), the prompts behind them: the decode rows of a
        step then sit at a fixed place whatever the prompts' lengths are, which is what lets a
        captured graph serve mixed steps.  ``execute`` hands the tokens back in the caller's
        order."""
        bs = self.block_size
        if num_decode:
            seqs = list(seqs[len(seqs) - num_decode:]) + list(seqs[:len(seqs) - num_decode])
        B = len(seqs)
        starts = np.empty(B, np.int64)
        ends = np.empty(B, np.int64)
        ids = []
        for i, seq in enumerate(seqs):
            if i < num_decode:
                starts[i], ends[i] = seq.num_tokens - 1, seq.num_tokens
            else:
                starts[i] = seq.num_cached_tokens
                ends[i] = seq.num_tokens if seq.prefill_end is None else seq.prefill_end
            ids.append(seq.token_array(int(starts[i]), int(ends[i])))
        lens = ends - starts
        max_q = int(lens[num_decode:].max()) if B > num_decode else 0
        max_ctx = int(ends[:num_decode].max()) if num_decode else 0
        max_k = int(ends.max())
        has_prefix = num_decode > 0 or bool(starts.any())
        cu_q = np.zeros(B + 1, np.int64)
        cu_k = np.zeros(B + 1, np.int64)
        np.cumsum(lens, out=cu_q[1:])
        np.cumsum(ends, out=cu_k[1:])
        total = int(cu_q[-1])
        # one pass over all rows, in int32: row i of sequence b is token starts[b] + (i - cu_q[b])
        row = np.repeat(np.arange(B, dtype=np.int32), lens)
        positions = np.arange(total, dtype=np.int32) - np.repeat((cu_q[:-1] - starts).astype(np.int32), lens)
        tables = _pad_block_tables([s.block_table for s in seqs])
        if bs & (bs - 1) == 0:
            sh = bs.bit_length() - 1
            blk, off = positions >> sh, positions & (bs - 1)
        else:
            blk, off = positions // bs, positions % bs
        slot_mapping = tables.reshape(-1).take(row * np.int32(tables.shape[1]) + blk)
        slot_mapping *= np.int32(bs)
        slot_mapping += off
        input_ids = np.concatenate(ids) if B > 1 else np.array(ids[0], dtype=np.int32)
        num_prefill_tokens = int(input_ids.size) - num_decode
        pad = prefill_token_bucket(input_ids.size) - input_ids.size if self.on_gpu else 0
        if pad:
            # rows that belong to no sequence: attention and the LM head go by cu_seqlens and
            # never see them, the KV store skips slot -1, every other op is row-wise
            input_ids = np.concatenate([input_ids, np.zeros(pad, np.int32)])
            positions = np.concatenate([positions, np.zeros(pad, np.int32)])
            slot_mapping = np.concatenate([slot_mapping, np.full(pad, -1, np.int32)])
        arrays = dict(input_ids=input_ids, positions=positions, slot_mapping=slot_mapping,
                      cu_seqlens_q=cu_q.astype(np.int32),
                      cu_seqlens_k=cu_k.astype(np.int32),
                      temperatures=np.asarray([s.temperature for s in seqs], np.float32),
                      block_tables=tables)
        if num_decode:
            arrays["context_lens"] = ends[:num_decode].astype(np.int32)
        if self.on_gpu:
            # the attention kernel's work list: one entry per pair of query tiles of a prompt
            # (sequence numbers count from the first prompt)
            from minivllm_amd.ops import hip
            arrays["prefill_items"] = hip.prefill_work_items(lens[num_decode:], ends[num_decode:])
        self._add_filter_arrays(arrays, seqs)
        return Batch(is_prefill=True, num_seqs=B, num_tokens=int(input_ids.size),
                     max_seqlen_q=max_q, max_seqlen_k=max_k, has_prefix=has_prefix,
                     num_prefill_k_tokens=int(cu_k[-1] - cu_k[num_decode]),
                     max_context_len=max_ctx, num_decode_seqs=num_decode,
                     num_prefill_tokens=num_prefill_tokens, arrays=arrays)

    def prepare_decode(self, seqs: Seq[Sequence]) -> Batch:
        bs = self.block_size
        B = len(seqs)
        prev = self._decode_prev
        if prev is not None and prev[0] is seqs and self._last_tokens is not None \
                and len(self._last_tokens) == B:
            # the scheduler handed back the very same batch (steady-state decode: every
            # sequence grew by the token sampled last step, no block table changed)
            a = prev[1]
            lens = a["context_lens"] + 1
            pos = lens - 1
            tables = a["block_tables"]
            blk = tables[self._arange(B), pos // bs]
            arrays = dict(input_ids=np.asarray(self._last_tokens, dtype=np.int32), positions=pos,
                          slot_mapping=blk * bs + pos % bs, context_lens=lens,
                          temperatures=a["temperatures"], block_tables=tables)
            if "top_k" in a:
                arrays["top_k"], arrays["top_p"] = a["top_k"], a["top_p"]
            self._decode_prev = (seqs, arrays)
            m = int(lens.max())
            return Batch(is_prefill=False, num_seqs=B, num_tokens=B, max_seqlen_q=1, max_seqlen_k=m,
                         max_context_len=m, arrays=arrays, tables_unchanged=True)
        lens = np.fromiter((s.num_tokens for s in seqs), np.int32, B)
        tables = _pad_block_tables([s.block_table for s in seqs])
        last_block = tables[self._arange(B), (lens - 1) // bs]
        pos = lens - 1
        arrays = dict(input_ids=np.fromiter((s.last_token for s in seqs), np.int32, B),
                      positions=pos, slot_mapping=last_block * bs + pos % bs, context_lens=lens,
                      temperatures=np.fromiter((s.temperature for s in seqs), np.float32, B),
                      block_tables=tables)
        self._add_filter_arrays(arrays, seqs)
        self._decode_prev = (seqs, arrays)
        return Batch(is_prefill=False, num_seqs=B, num_tokens=B, max_seqlen_q=1,
                     max_seqlen_k=int(lens.max()), max_context_len=int(lens.max()), arrays=arrays)

    @staticmethod
    def _add_filter_arrays(arrays: dict, seqs: Seq[Sequence]) -> None:
        """top-k / top-p per sequence, only when some request of the batch uses them."""
        if any(s.top_k > 0 or s.top_p < 1.0 for s in seqs):
            arrays["top_k"] = np.asarray([s.top_k for s in seqs], np.int32)
            arrays["top_p"] = np.asarray([s.top_p for s in seqs], np.float32)

    def _arange(self, n: int) -> np.ndarray:
        if self._arange_cache.size < n:
            self._arange_cache = np.arange(max(n, 256))
        return self._arange_cache[:n]

    def prepare_sample(self, seqs: Seq[Sequence]) -> torch.Tensor:
        return torch.tensor([s.temperature for s in seqs], dtype=torch.float32, device=self.device)

    # ================================================================== run
    def run(self, seqs: Seq[Sequence], is_prefill: bool, num_decode: int = 0,
            allow_prelaunch: int = 0, next_prefill=None) -> Optional[List[int]]:
        """Rank 0 entry point for one step; returns the sampled token per sequence.
        ``num_decode`` > 0 (prefill steps only): the last that many sequences decode.
        ``allow_prelaunch`` (decode steps; ``Scheduler.next_decode_step_ahead``): non-zero when the
        caller guarantees that the NEXT step is a decode step of the same sequences (nothing
        waits to be admitted; 2 = some of them have just been given a new KV block for it), so
        that step may be launched before this one's tokens are read back; the caller then
        collects it with ``run_prelaunched``.
        ``next_prefill`` (pure prefill steps): a callable that admits and returns the sequences
        of the prefill step after this one, or [] (``Scheduler.schedule_prefill_ahead``); that
        step is then prepared, and where possible launched, before this one's tokens are read
        back, and the caller must collect it with ``run_prefill_ahead``."""
        assert self._spec is None and self._pf_spec is None, "a step launched ahead must be collected first"
        if is_prefill:
            self._decode_prev = None
            batch = self.prepare_prefill(seqs, num_decode)
        else:
            batch = self.prepare_decode(seqs)
        self.step_counter += 1
        batch.step = self.step_counter
        batch.prelaunch = bool(allow_prelaunch) and self._can_prelaunch(batch)
        if not is_prefill and allow_prelaunch == 2:
            # the scheduler has just given some sequences the block their token after this one
            # starts: the tables every rank derives the next step from must hold it
            batch.arrays["block_tables"] = _pad_block_tables([s.block_table for s in seqs])
            batch.tables_unchanged = False
        if is_prefill and next_prefill is not None and num_decode == 0:
            return self._run_prefill_with_lookahead(batch, next_prefill, self._PG_KEY_UNSET)
        if self.world_size > 1 and self.rank == 0:
            self.channel.send("run_batch", batch.scalars(), batch.arrays)
        with torch.inference_mode(), trace_range("prefill_step" if is_prefill else "decode_step"):
            if batch.prelaunch:
                tokens_dev = self._launch_decode(batch)
                pending = self._read_tokens_async(tokens_dev, batch.num_seqs)
                self._prelaunch_after(seqs, batch, tokens_dev)
                out = self._finish_read(pending)
                self._last_tokens = out
                return out
            return self.execute(batch)

    # --------------------------------------------------- prefill launched one step ahead
    # Between two prefill steps the GPU otherwise idles while the host reads the samples back,
    # retires the sequences, admits the next ones, builds their batch arrays and replays the
    # graph (~0.4 ms: 2.5 % of a 16k-token Qwen3-0.6B step, more of the 1-4 ms steps a server
    # sees).  None of that needs the running step's samples when the next step is made of
    # sequences that are still waiting (or continues a prompt the running step has cut at its
    # token budget): they are admitted while the step runs
    # (``Scheduler.schedule_prefill_ahead``), their inputs go up behind it (second pinned
    # staging block; the device-side input buffer is only written in stream order, after the
    # running graph has read it) and their graph is replayed behind that; only then does the
    # host wait for the running step's tokens, whose copy to the host was enqueued before the
    # next graph could overwrite the graph's output buffer.  A step whose shape has no graph yet
    # (it has to be captured, or runs eagerly) is prepared ahead but executed by the next call.
    # Tensor-parallel workers need nothing new: they execute the batches in the order rank 0
    # sends them and never wait for tokens.
    _PG_KEY_UNSET = object()

    def _prefill_can_go_ahead(self, b: Batch, key) -> bool:
        if not self.cfg.prefill_prelaunch or "top_k" in b.arrays or b.num_decode_seqs:
            return False
        if os.environ.get("MINIVLLM_NO_PRELAUNCH") == "1":
            return False
        return (key is not None and key in self.prefill_graphs) if self.on_gpu else True

    def _enqueue_prefill_step(self, b: Batch, key) -> torch.Tensor:
        """Send the batch to the workers and enqueue the step on this rank without waiting for
        anything; returns the device tensor its tokens land in."""
        if self.world_size > 1 and self.rank == 0:
            self.channel.send("run_batch", b.scalars(), b.arrays)
        tokens, logits = self._enqueue_prefill(b, key if self.on_gpu else None)
        reset_context()
        self._last_step_logits = logits
        return tokens

    def _run_prefill_with_lookahead(self, b: Batch, next_prefill, key, launched=None) -> List[int]:
        """The step ``b`` (already on the device when ``launched`` = its pending token read),
        with the step after it admitted, prepared and - if its graph exists - launched before
        this one's tokens are waited for."""
        with torch.inference_mode(), trace_range("prefill_step"):
            if launched is None:
                if key is self._PG_KEY_UNSET:
                    key = self._prefill_graph_key(b)
                if not self._prefill_can_go_ahead(b, key):
                    # runs the ordinary way (eager, or captured now); nothing is launched behind a
                    # step that may synchronise, but the look-ahead stays available to the next one
                    if self.world_size > 1 and self.rank == 0:
                        self.channel.send("run_batch", b.scalars(), b.arrays)
                    return self.execute(b, pg_key=key)
                tokens_dev = self._enqueue_prefill_step(b, key)
                launched = self._read_tokens_async(tokens_dev, b.num_seqs)
            nxt = next_prefill() if next_prefill is not None else []
            if nxt:
                nb = self.prepare_prefill(nxt, 0)
                nkey = self._prefill_graph_key(nb)
                spec = dict(seqs=nxt, batch=nb, key=nkey, pending=None,
                            num_tokens=sum(s.num_scheduled_prefill_tokens for s in nxt))
                if self._prefill_can_go_ahead(nb, nkey):
                    self.step_counter += 1
                    nb.step = self.step_counter
                    spec["pending"] = self._read_tokens_async(self._enqueue_prefill_step(nb, nkey),
                                                              nb.num_seqs)
                self._pf_spec = spec
            out = self._finish_read(launched)
            self._last_tokens = None
            return out

    def has_prefill_ahead(self) -> bool:
        return self._pf_spec is not None

    def prefill_ahead_in_flight(self) -> bool:
        """The step admitted ahead is already on the device (its graph existed)."""
        return self._pf_spec is not None and self._pf_spec["pending"] is not None

    def prefill_ahead_info(self):
        """(sequences, prompt tokens computed) of the prefill step admitted ahead."""
        return self._pf_spec["seqs"], self._pf_spec["num_tokens"]

    def run_prefill_ahead(self, next_prefill=None) -> List[int]:
        """Collect (or, if it could not be launched ahead, run) the prefill step admitted ahead;
        returns one token per sequence of ``prefill_ahead_info()``."""
        spec, self._pf_spec = self._pf_spec, None
        b = spec["batch"]
        self._decode_prev = None
        if spec["pending"] is None:
            self.step_counter += 1
            b.step = self.step_counter
        return self._run_prefill_with_lookahead(b, next_prefill, spec["key"], launched=spec["pending"])

    def drop_prefill_ahead(self) -> None:
        """Forget the step admitted ahead (every sequence of it has left)."""
        if self._pf_spec is not None and self._pf_spec["pending"] is not None and self.on_gpu:
            torch.cuda.synchronize()
        self._pf_spec = None

    # ---------------------------------------------------- decode launched one step ahead
    # The GPU otherwise idles between two decode steps while the host reads the tokens back,
    # updates the sequences, schedules and fills the next step's inputs (~0.2-0.4 ms, 6 % of a
    # Qwen3-0.6B step at batch 256).  In steady-state decode none of that is needed to START
    # the next step: its positions / slots / lengths follow from this step's, and its input
    # tokens are this step's samples, which are already on the device.  So step n + 1 is
    # enqueued right behind step n (device-to-device copy of the samples into the input
    # buffer, upload of the other inputs from the second staging block, graph replay), and
    # only then does the host wait for step n's tokens.  A sequence that turns out to have
    # finished at step n (EOS) simply has its row of step n + 1 ignored: what that row wrote
    # to the KV cache lies beyond the sequence's tokens, in a block that is not shared.
    # Tensor-parallel engines do the same on every rank: the samples of step n are on every
    # rank's device (each rank reduces the gathered (score, index) pairs itself), so a worker
    # derives step n + 1 from step n exactly as rank 0 does.  Rank 0 says so in the batch it
    # broadcasts (``Batch.prelaunch``) and sends one small "prelaunch_next" message for every
    # further step launched ahead; the workers never read tokens back.
    def _can_prelaunch(self, b: Batch) -> bool:
        if b.is_prefill or not self.cfg.decode_prelaunch or "top_k" in b.arrays:
            return False
        if os.environ.get("MINIVLLM_NO_PRELAUNCH") == "1":
            return False
        if self.on_gpu and self._graph_bucket(b.num_seqs) is None:
            return False                 # eager GPU steps are host-bound anyway
        # the step after this one reads position = context length: inside the model's tables?
        return int(b.arrays["context_lens"].max()) + 1 < self.cfg.max_model_length

    def _launch_decode(self, b: Batch, input_ids_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Enqueue one decode step; returns the device tensor its sampled tokens land in.
        ``input_ids_dev``: take the input tokens from this device tensor (the previous step's
        samples) instead of the host arrays."""
        B = b.num_seqs
        bucket = self._graph_bucket(B)
        padded = bucket if bucket is not None else B
        self._fill_decode_host(b, padded)
        if _CHECK_BATCHES:
            self._check_decode_image(b, padded)
        self._dec_dev.copy_(self._dec_host, non_blocking=True)
        if input_ids_dev is not None:
            self._dec_view(self._dec_dev, "input_ids", B).copy_(input_ids_dev)
        if bucket is not None:
            self.graphs[bucket].replay()
            return self._graph_out[bucket][:B]
        t = self._decode_tensors(B)
        set_context(False, slot_mapping=t["slot_mapping"], context_lens=t["context_lens"],
                    block_tables=t["block_tables"], positions=t["positions"],
                    max_context_len=b.max_context_len, num_seqs=B, positions_in_context=True)
        tokens = self._forward_and_sample(t)
        reset_context()
        return tokens

    def _read_tokens_async(self, tokens_dev: torch.Tensor, n: int):
        """Start copying ``n`` sampled tokens to the host; returns what ``_finish_read`` needs."""
        if not self.on_gpu:
            return tokens_dev[:n].clone(), None
        self._tok_flip ^= 1
        host = self._tok_hosts[self._tok_flip][:n]
        host.copy_(tokens_dev[:n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return host, ev

    def _finish_read(self, pending) -> List[int]:
        host, ev = pending
        if ev is not None:
            ev.synchronize()
            pg.check_custom_all_reduce(synchronize=False)
        return host.tolist()

    def _prelaunch_after(self, seqs, b: Batch, tokens_dev: torch.Tensor, tables=None) -> None:
        """Enqueue the decode step that follows ``b`` for the same sequences.  ``tables``: the
        sequences' block tables if they have grown since ``b`` was made."""
        a, bs = b.arrays, self.block_size
        B = b.num_seqs
        lens = a["context_lens"] + 1
        pos = lens - 1
        grown = tables is not None
        if not grown:
            tables = a["block_tables"]
        blk = tables[self._arange(B), pos // bs]
        arrays = dict(input_ids=np.zeros(B, np.int32),         # replaced on the device
                      positions=pos, slot_mapping=blk * bs + pos % bs, context_lens=lens,
                      temperatures=a["temperatures"], block_tables=tables)
        m = int(lens.max())
        nxt = Batch(is_prefill=False, num_seqs=B, num_tokens=B, max_seqlen_q=1, max_seqlen_k=m,
                    max_context_len=m, arrays=arrays, tables_unchanged=not grown)
        nxt.step = self.step_counter = b.step + 1        # workers take the count from the batch
        out_dev = self._launch_decode(nxt, input_ids_dev=tokens_dev)
        self._spec = dict(seqs=seqs, batch=nxt, tokens=out_dev)

    def has_prelaunched(self) -> bool:
        return self._spec is not None

    def prelaunched_seqs(self):
        return self._spec["seqs"]

    def drop_prelaunched(self) -> None:
        """Forget the step launched ahead (every sequence of it has left)."""
        if self._spec is not None and self.on_gpu:
            torch.cuda.synchronize()
        self._spec = None
        self._decode_prev = None
        self._last_tokens = None

    def run_prelaunched(self, alive: Seq[int], allow_prelaunch: int = 0) -> List[int]:
        """Collect the step launched ahead: the tokens of the rows ``alive`` (indices into
        ``prelaunched_seqs()``; the other sequences finished in the meantime).  With every
        row alive and ``allow_prelaunch`` the step after it is launched ahead in turn."""
        spec, self._spec = self._spec, None
        b, seqs = spec["batch"], spec["seqs"]
        with torch.inference_mode(), trace_range("decode_step"):
            pending = self._read_tokens_async(spec["tokens"], b.num_seqs)
            if allow_prelaunch and len(alive) == b.num_seqs and self._can_prelaunch(b):
                tables = (_pad_block_tables([s.block_table for s in seqs])
                          if allow_prelaunch == 2 else None)
                if self.world_size > 1:
                    self.channel.send("prelaunch_next", {},
                                      {} if tables is None else {"block_tables": tables})
                self._prelaunch_after(seqs, b, spec["tokens"], tables)
            out = self._finish_read(pending)
        # a normal step for the same list continues from here (prepare_decode's shortcut)
        b.arrays["input_ids"] = np.asarray(out, dtype=np.int32)
        self._decode_prev = (seqs, b.arrays)
        self._last_tokens = out
        return [out[i] for i in alive]

    # ------------------------------------------------------------- uploads
    def _upload_prefill(self, b: Batch):
        a = b.arrays
        T, B = b.num_tokens, b.num_seqs
        MB = a["block_tables"].shape[1]
        al = lambda n: (n + 3) // 4 * 4
        layout, off = {}, 0
        for name, n in (("input_ids", T), ("positions", T), ("slot_mapping", T),
                        ("cu_seqlens_q", B + 1), ("cu_seqlens_k", B + 1), ("temperatures", B),
                        ("misc", 4), ("context_lens", b.num_decode_seqs),
                        ("prefill_items", a["prefill_items"].size if "prefill_items" in a else 0),
                        ("block_tables", B * MB)):
            layout[name] = (off, n)
            off += al(n)
        self._ensure_prefill_buffers(off)
        host = self._pre_host.numpy()
        for name, (o, n) in layout.items():
            if name == "misc":
                host[o] = b.step
            elif n == 0:
                continue
            elif name == "temperatures":
                host[o:o + n] = a[name].astype(np.float32).view(np.int32)
            else:
                host[o:o + n] = a[name].reshape(-1)
        self._pre_dev[:off].copy_(self._pre_host[:off], non_blocking=True)
        v = lambda name: self._pre_dev[layout[name][0]:layout[name][0] + layout[name][1]]
        return dict(input_ids=v("input_ids"), positions=v("positions"),
                    slot_mapping=v("slot_mapping"), cu_seqlens_q=v("cu_seqlens_q"),
                    cu_seqlens_k=v("cu_seqlens_k"),
                    temperatures=v("temperatures").view(torch.float32),
                    step=v("misc"), context_lens=v("context_lens"),
                    prefill_items=v("prefill_items") if "prefill_items" in a else None,
                    positions_in_context=True,
                    block_tables=v("block_tables").view(B, MB))

    def _fill_decode_host(self, b: Batch, padded_bs: int) -> None:
        a = b.arrays
        B = b.num_seqs
        MBmax = self.max_blocks_per_seq
        self._dec_flip ^= 1
        self._dec_host = self._dec_hosts[self._dec_flip]
        host = self._dec_host.numpy()

        def put(name, values, fill):
            off, _ = self._dec_off[name]
            host[off:off + B] = values
            if padded_bs > B:
                host[off + B:off + padded_bs] = fill
        put("input_ids", a["input_ids"], 0)
        put("positions", a["positions"], 0)
        put("slot_mapping", a["slot_mapping"], -1)
        put("context_lens", a["context_lens"], 0)
        put("temperatures", a["temperatures"].astype(np.float32).view(np.int32), 0)
        host[self._dec_off["misc"][0]] = b.step
        # Which table rows a staging block already holds is tracked by a counter, never by the
        # identity of the numpy array: a freed array's address is handed to the next one, and a
        # block taken for up to date on that ground ships another batch's tables (entries of
        # -1 under a live context length: the decode kernel then reads far outside the cache).
        if not b.tables_unchanged:
            self._tables_version += 1     # a new set of block tables (every rank counts alike)
        key = (padded_bs, B, self._tables_version)
        if self._dec_tables_keys[self._dec_flip] == key:
            return                        # this block's table image already holds these rows
        self._dec_tables_keys[self._dec_flip] = key
        off, _ = self._dec_off["block_tables"]
        tab = host[off:off + padded_bs * MBmax].reshape(padded_bs, MBmax)
        mb = a["block_tables"].shape[1]
        if mb > MBmax:
            raise RuntimeError("sequence uses more KV blocks than max_model_length allows")
        tab[:B, :mb] = a["block_tables"]
        tab[:B, mb:] = -1
        if padded_bs > B:
            tab[B:] = -1

    def _check_decode_image(self, b: Batch, padded_bs: int) -> None:
        """Debug aid (MINIVLLM_CHECK_BATCHES=1): the staging block about to be uploaded must
        describe exactly this batch - every row's slot inside a block its table names, no
        unset table entry below a row's context length, padding rows inert.  A kernel handed
        anything else reads or writes outside the KV cache."""
        host = self._dec_host.numpy()
        B, bs, MB = b.num_seqs, self.block_size, self.max_blocks_per_seq
        get = lambda name, n: host[self._dec_off[name][0]:self._dec_off[name][0] + n]
        lens, pos, slots = get("context_lens", padded_bs), get("positions", padded_bs), get("slot_mapping", padded_bs)
        tab = get("block_tables", padded_bs * MB).reshape(padded_bs, MB)
        a = b.arrays
        assert (lens[:B] == a["context_lens"]).all() and (pos[:B] == a["positions"]).all()
        assert (slots[:B] == a["slot_mapping"]).all()
        mb = a["block_tables"].shape[1]
        assert (tab[:B, :mb] == a["block_tables"]).all(), "stale block-table image"
        assert (tab[:B, mb:] == -1).all()
        for i in range(B):
            need = (int(lens[i]) + bs - 1) // bs
            assert (tab[i, :need] >= 0).all() and (tab[i, :need] < self.num_kv_blocks).all(), (i, tab[i], lens[i])
            assert slots[i] == tab[i, pos[i] // bs] * bs + pos[i] % bs and pos[i] == lens[i] - 1
            assert 0 <= pos[i] < self.cfg.max_model_length
        assert (lens[B:] == 0).all() and (slots[B:] == -1).all() and (tab[B:] == -1).all()

    def _decode_tensors(self, n: int):
        d = self._dec_dev
        MB = self.max_blocks_per_seq
        return dict(input_ids=self._dec_view(d, "input_ids", n),
                    positions=self._dec_view(d, "positions", n),
                    slot_mapping=self._dec_view(d, "slot_mapping", n),
                    context_lens=self._dec_view(d, "context_lens", n),
                    temperatures=self._dec_view(d, "temperatures", n).view(torch.float32),
                    step=self._dec_view(d, "misc"),
                    block_tables=self._dec_view(d, "block_tables", n * MB).view(n, MB))

    # -------------------------------------------------------------- execute
    def _enqueue_prefill(self, b: Batch, pg_key, warmup: bool = False):
        """Enqueue one prefill step (graph replay, first run + capture, or eager launches);
        returns (device tensor of the sampled tokens, the step's logits)."""
        if pg_key is not None:
            return self._run_prefill_graph(b, pg_key)
        t = self._upload_prefill(b)
        set_context(True, cu_seqlens_q=t["cu_seqlens_q"], cu_seqlens_k=t["cu_seqlens_k"],
                    max_seqlen_q=b.max_seqlen_q, max_seqlen_k=b.max_seqlen_k,
                    slot_mapping=None if warmup else t["slot_mapping"],
                    block_tables=None if warmup else t["block_tables"],
                    positions=t["positions"], num_seqs=b.num_seqs, has_prefix=b.has_prefix,
                    num_decode_seqs=b.num_decode_seqs, num_prefill_tokens=b.num_prefill_tokens,
                    context_lens=t["context_lens"] if b.num_decode_seqs else None,
                    max_context_len=b.max_context_len, total_k_tokens=b.num_prefill_k_tokens,
                    prefill_items=None if warmup else t["prefill_items"],
                    positions_in_context=True)
        tokens = self._forward_and_sample(t)
        return tokens, self._last_logits

    def execute(self, b: Batch, warmup: bool = False, pg_key=_PG_KEY_UNSET) -> Optional[List[int]]:
        if b.is_prefill:
            if pg_key is self._PG_KEY_UNSET:
                pg_key = self._prefill_graph_key(b) if not warmup else None
            tokens, logits = self._enqueue_prefill(b, pg_key, warmup)
        else:
            B = b.num_seqs
            bucket = self._graph_bucket(B)
            padded = bucket if bucket is not None else B
            self._fill_decode_host(b, padded)
            if _CHECK_BATCHES:
                self._check_decode_image(b, padded)
            self._dec_dev.copy_(self._dec_host, non_blocking=True)
            if bucket is not None:
                self.graphs[bucket].replay()
                tokens = self._graph_out[bucket][:B]
                logits = self._graph_logits.get(bucket)
            else:
                t = self._decode_tensors(B)
                set_context(False, slot_mapping=t["slot_mapping"], context_lens=t["context_lens"],
                            block_tables=t["block_tables"], positions=t["positions"],
                            max_context_len=b.max_context_len, num_seqs=B,
                            positions_in_context=True)
                tokens = self._forward_and_sample(t)
                logits = self._last_logits
        reset_context()
        self._last_step_logits = logits          # this step's (local) logits, whichever path ran
        if not warmup and "top_k" in b.arrays:
            tokens = self._resample_filtered(b, tokens, logits)
        if warmup or tokens is None or self.rank != 0:
            if not warmup and b.step % self.P2P_CHECK_EVERY == 0:
                pg.check_custom_all_reduce(synchronize=True)     # workers: now and then
            return None
        out = tokens.tolist()
        if b.is_prefill and b.num_decode_seqs:           # riders first inside, last for the caller
            Bd = b.num_decode_seqs
            db = b.decode_slots if b.decode_slots is not None else Bd    # a graph's decode slots
            out = out[db:db + b.num_seqs - Bd] + out[:Bd]
        # the read-back has drained the stream: looking at the peer-to-peer all-reduce's status
        # word costs one small copy, so every step of rank 0 can afford it
        pg.check_custom_all_reduce(synchronize=False)
        self._last_tokens = None if b.is_prefill else out
        return out

    def _graph_bucket(self, B: int) -> Optional[int]:
        if not self.graphs:
            return None
        for s in self._graph_sizes:
            if s >= B:
                return s
        return None

    def _forward_and_sample(self, t: dict) -> Optional[torch.Tensor]:
        hidden = self.model(t["input_ids"], t["positions"])
        return self._sample(hidden, t["temperatures"], t["step"])

    def _sample(self, hidden: torch.Tensor, temperatures: torch.Tensor,
                step: torch.Tensor) -> Optional[torch.Tensor]:
        if self.on_gpu:
            from minivllm_amd.ops import hip
            head = self.model.lm_head
            logits = head.local_logits(hidden)
            self._last_logits = logits
            vals, idx = hip.sample_local(logits, temperatures, self.cfg.seed, step,
                                         vocab_offset=head.vocab_start,
                                         valid_vocab=head.num_embeddings)
            if self.world_size == 1 and not pg.collectives_active():
                return idx
            pair = torch.stack([vals, idx.to(torch.float32)], dim=-1)          # [B, 2]
            allp = pg.tp_all_gather_stack(pair)                                 # [tp, B, 2]
            best = allp[..., 0].argmax(dim=0)
            return allp[..., 1].gather(0, best[None]).squeeze(0).to(torch.int32)
        if self.world_size > 1 or pg.is_rehearsal():
            # CPU ranks (gloo): every rank assembles the full logits and draws the same token from
            # its identically seeded generator, so that workers hold the step's samples as GPU
            # ranks do (a decode step launched ahead feeds them back without rank 0's help)
            head = self.model.lm_head
            logits = pg.tp_all_gather(head.local_logits(hidden), dim=-1)[..., :head.num_embeddings]
        else:
            logits = self.model.compute_logits(hidden)
        self._last_logits = logits
        if logits is None:
            return None
        return self.sampler(logits, temperatures)

    def _resample_filtered(self, b: Batch, tokens: Optional[torch.Tensor],
                           logits: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """Rows that ask for top-k / top-p are drawn again from the restricted distribution and
        overwrite the unrestricted draw (the step's logits are still there: inside a replayed
        graph they sit in a static buffer).  Runs outside the graphs, only for batches that
        carry such a request; with sharded logits (GPU tensor parallelism) every rank takes
        part in gathering the rows concerned."""
        a = b.arrays
        B = b.num_seqs
        need = (a["temperatures"][:B] > 0) & ((a["top_k"][:B] > 0) | (a["top_p"][:B] < 1.0))
        rows = np.nonzero(need)[0]
        if rows.size == 0 or logits is None:
            return tokens
        dev = logits.device
        ridx = torch.as_tensor(rows, device=dev)
        if self.on_gpu:
            # one kernel, no sort (rejection sampling with a rising pivot, csrc/sampler.hip)
            from minivllm_amd.ops import hip
            rows32 = ridx.to(torch.int32)
            if self.world_size > 1 or pg.collectives_active():
                lg = pg.tp_all_gather(logits[:B].index_select(0, ridx), dim=-1).contiguous()
                logit_rows = torch.arange(rows.size, dtype=torch.int32, device=dev)
            else:
                lg, logit_rows = logits, rows32
            if tokens is None:                   # workers only take part in the gather
                return None
            assert tokens.dtype == torch.int32
            hip.sample_filtered(
                lg, logit_rows, rows32,
                torch.as_tensor(a["temperatures"][:B].astype(np.float32), device=dev),
                torch.as_tensor(a["top_k"][:B].astype(np.int32), device=dev),
                torch.as_tensor(a["top_p"][:B].astype(np.float32), device=dev),
                self.cfg.seed, b.step, tokens, vocab=self.cfg.model.vocab_size)
            return tokens
        lg = logits[:B].index_select(0, ridx)
        lg = lg[:, :self.cfg.model.vocab_size].float()
        lg = lg / torch.as_tensor(a["temperatures"][rows], device=dev)[:, None]
        vals, order = lg.sort(dim=-1, descending=True)
        k = torch.as_tensor(a["top_k"][rows], device=dev).long()
        p = torch.as_tensor(a["top_p"][rows], device=dev)
        rank_of = torch.arange(vals.shape[1], device=dev)[None]
        drop = (k[:, None] > 0) & (rank_of >= k[:, None])
        probs = torch.softmax(vals, dim=-1)
        drop |= (probs.cumsum(-1) - probs) >= p[:, None]          # the first token always stays
        vals = vals.masked_fill(drop, float("-inf"))
        gen = torch.Generator(device=dev).manual_seed((self.cfg.seed * 1000003 + b.step) % (1 << 62))
        u = torch.rand(vals.shape, generator=gen, device=dev).clamp_(1e-10, 1.0 - 1e-7)
        choice = (vals - torch.log(-torch.log(u))).argmax(dim=-1)
        picked = order.gather(1, choice[:, None]).squeeze(1)
        if tokens is None:
            return None
        tokens[ridx] = picked.to(tokens.dtype)
        return tokens

    # ================================================================ graphs
    def _capture_graphs_or_fall_back(self) -> None:
        """hipGraph capture needs collectives that run as stream work (RCCL does).  A backend
        that stages through the host (gloo with GPU tensors, used only to rehearse the TP path
        on one device) synchronises inside every collective, which a capturing stream forbids
        and which leaves the stream unusable afterwards, so it is not even attempted there;
        every rank takes the same decision because the backend is a property of the group."""
        if pg.host_staged_collectives():
            if self.rank == 0:
                print("[runner] collectives are host-staged: decode runs eagerly (no hipGraph)",
                      flush=True)
            self.enforce_eager = True
            return
        try:
            self.capture_graphs()
        except Exception as e:                   # noqa: BLE001
            # One rank: a capture failure is a bug and stays loud.  Several ranks: a collective
            # library that cannot be captured on this machine must not cost the run - replayed
            # and eager steps issue the same collectives in the same order, so ranks that did
            # capture stay compatible with ranks that did not.
            if self.world_size == 1:
                raise
            print(f"[runner] rank {self.rank}: hipGraph capture failed ({e!r}); steps run eagerly",
                  flush=True)
            self.graphs.clear()
            self.prefill_graphs.clear()
            self.enforce_eager = True
            try:
                torch.cuda.synchronize()
            except Exception:                    # noqa: BLE001
                pass

    @torch.inference_mode()
    def capture_graphs(self) -> None:
        """Capture one hipGraph per decode batch-size bucket, largest first so that all
        graphs share one memory pool (same policy as reference model_runner.py:316-369)."""
        sizes = self.cfg.graph_batch_sizes or default_graph_batch_sizes(self.max_bs)
        self._graph_sizes = sorted(sizes)
        self._graph_out: Dict[int, torch.Tensor] = {}
        # a harmless decode batch: every row is padding (context 0, slot -1)
        host = self._dec_host.numpy()
        host[:] = 0
        o, n = self._dec_off["slot_mapping"]
        host[o:o + n] = -1
        o, n = self._dec_off["block_tables"]
        host[o:o + n] = -1
        self._dec_dev.copy_(self._dec_host)
        for bs in reversed(self._graph_sizes):
            t = self._decode_tensors(bs)

            def step():
                set_context(False, slot_mapping=t["slot_mapping"], context_lens=t["context_lens"],
                            block_tables=t["block_tables"], positions=t["positions"],
                            max_context_len=self.cfg.max_model_length, num_seqs=bs,
                            positions_in_context=True)
                out = self._forward_and_sample(t)
                reset_context()
                return out
            step()                                   # warm-up outside capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # thread-local capture: RCCL's watchdog thread may query events while this thread
            # captures, which the default (global) mode would turn into a capture error
            with torch.cuda.graph(g, pool=self.graph_pool, capture_error_mode="thread_local"):
                out = step()
            if self.graph_pool is None:
                self.graph_pool = g.pool()
            self.graphs[bs] = g
            self._graph_out[bs] = out
            self._graph_logits[bs] = self._last_logits
        torch.cuda.synchronize()
        self._capture_prefill_graphs()

    # ------------------------------------------------------------ prefill graphs
    def _pg_tensors(self, tb: int, sb: int, db: int = 0) -> dict:
        d = self._pg_dev
        MB = self.max_blocks_per_seq
        from minivllm_amd.ops import hip

        def v(name, n=None):
            off, size = self._pg_off[name]
            return d[off:off + (size if n is None else n)]
        return dict(input_ids=v("input_ids", tb), positions=v("positions", tb),
                    slot_mapping=v("slot_mapping", tb), cu_seqlens_q=v("cu_seqlens_q", sb + 1),
                    cu_seqlens_k=v("cu_seqlens_k", sb + 1),
                    temperatures=v("temperatures", sb).view(torch.float32), step=v("misc"),
                    context_lens=v("context_lens", db),
                    prefill_items=v("prefill_items", hip.prefill_item_slots(tb, sb)),
                    positions_in_context=True,
                    block_tables=v("block_tables", sb * MB).view(sb, MB))

    def _prefill_graph_key(self, b: Batch):
        """(token bucket, sequence-slot bucket, decode-slot bucket) of the graph that serves
        this prefill step, or None when the step launches eagerly: steps beyond the static
        buffers, shapes whose capture failed, mixed steps with restricted sampling, and - without
        lazy capture - anything not captured at start-up.  A mixed step's riders take the first
        ``db`` rows and sequence slots (unused ones have length 0), its prompts follow."""
        if not self._pg_off or self.enforce_eager:
            return None
        Bd = b.num_decode_seqs
        tb, slots, db = b.num_tokens, b.num_seqs, 0
        if Bd:
            if "top_k" in b.arrays:
                return None
            db = next((x for x in self._pg_dec_buckets if x >= Bd), None)
            if db is None:
                return None
            tb = prefill_token_bucket(db + b.num_prefill_tokens)
            slots = db + b.num_seqs - Bd
        if tb > self._pg_tmax or slots > self._pg_smax \
                or b.arrays["block_tables"].shape[1] > self.max_blocks_per_seq:
            return None
        sb = next(x for x in self._pg_seq_buckets if x >= slots)
        key = (tb, sb, db)
        if key in self.prefill_graphs:
            return key
        if (not self._pg_lazy or key in self._pg_failed
                or len(self.prefill_graphs) >= self.cfg.prefill_graph_max_graphs
                or "prefill_items" not in b.arrays):
            return None
        # a capture costs about two steps of this shape: only shapes that come back get one
        # (every rank counts alike, so all of them capture at the same step)
        seen = self._pg_seen.get(key, 0) + 1
        self._pg_seen[key] = seen
        return key if seen >= self.cfg.prefill_graph_capture_after else None

    def _pg_step(self, t: dict, tb: int, sb: int, db: int = 0):
        set_context(True, cu_seqlens_q=t["cu_seqlens_q"], cu_seqlens_k=t["cu_seqlens_k"],
                    max_seqlen_q=tb, max_seqlen_k=self.cfg.max_model_length,
                    slot_mapping=t["slot_mapping"], block_tables=t["block_tables"],
                    positions=t["positions"], num_seqs=sb, has_prefix=True,
                    num_decode_seqs=db, context_lens=t["context_lens"] if db else None,
                    max_context_len=self.cfg.max_model_length if db else 0,
                    prefill_items=t["prefill_items"], positions_in_context=True)
        out = self._forward_and_sample(t)
        reset_context()
        return out

    def _capture_prefill_graph(self, key: tuple, warm_up: bool) -> None:
        """Record one prefill step on the static buffers.  ``warm_up``: run it once eagerly
        first (start-up captures; a lazy capture has just run the real step there)."""
        tb, sb, db = key
        t = self._pg_tensors(tb, sb, db)
        if warm_up:
            self._pg_step(t, tb, sb, db)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.graph_pool, capture_error_mode="thread_local"):
            out = self._pg_step(t, tb, sb, db)
        if self.graph_pool is None:
            self.graph_pool = g.pool()
        self.prefill_graphs[key] = g
        self._pg_out[key] = out
        self._pg_logits[key] = self._last_logits

    def _capture_prefill_graphs(self) -> None:
        """Start-up captures: one graph per token bucket up to ``prefill_graph_max_tokens`` for
        steps of up to ``prefill_graph_max_seqs`` sequences (short prompts, a new chat turn
        behind a cached prefix, a chunk of a long prompt: the steps a server sees most).  Any
        cached prefix is fine: attention goes through the block table and reads the lengths
        from device memory; unused sequence slots have length 0."""
        if not self._pg_sizes:
            return
        host = self._pg_host.numpy()
        host[:] = 0
        for name in ("slot_mapping", "block_tables", "prefill_items"):
            o, n = self._pg_off[name]
            host[o:o + n] = -1
        # a harmless one-token sequence (reads block 0, writes no KV: slot -1) + empty slots
        for name in ("cu_seqlens_q", "cu_seqlens_k"):
            o, n = self._pg_off[name]
            host[o + 1:o + n] = 1
        host[self._pg_off["block_tables"][0]] = 0
        host[self._pg_off["prefill_items"][0]] = 0           # sequence 0, pair 0
        self._pg_dev.copy_(self._pg_host)
        for tb in reversed(self._pg_sizes):
            self._capture_prefill_graph((tb, self._pg_seq_buckets[0], 0), warm_up=True)
        torch.cuda.synchronize()

    def _fill_prefill_graph_inputs(self, b: Batch, key: tuple) -> None:
        a = b.arrays
        tb, sb, db = key
        B, MB = b.num_seqs, self.max_blocks_per_seq
        Bd = b.num_decode_seqs
        pad = db - Bd                      # unused decode slots: one row each, length 0
        T = Bd + b.num_prefill_tokens      # real rows of the batch (riders, then prompts)
        from minivllm_amd.ops import hip
        self._pg_flip ^= 1
        self._pg_host = self._pg_hosts[self._pg_flip]
        host = self._pg_host.numpy()
        for name, fill in (("input_ids", 0), ("positions", 0), ("slot_mapping", -1)):
            o, _ = self._pg_off[name]
            host[o:o + Bd] = a[name][:Bd]
            host[o + Bd:o + db] = fill
            host[o + db:o + pad + T] = a[name][Bd:T]
            host[o + pad + T:o + tb] = fill
        # sequence slots: riders, unused decode slots (a row each, no keys), prompts, unused
        # slots (zero-length sequences)
        o, _ = self._pg_off["cu_seqlens_q"]
        host[o:o + db + 1] = np.arange(db + 1, dtype=np.int32)
        host[o + db + 1:o + pad + B + 1] = a["cu_seqlens_q"][Bd + 1:] + pad
        host[o + pad + B + 1:o + sb + 1] = a["cu_seqlens_q"][-1] + pad
        o, _ = self._pg_off["cu_seqlens_k"]
        host[o:o + Bd + 1] = a["cu_seqlens_k"][:Bd + 1]
        host[o + Bd + 1:o + db + 1] = a["cu_seqlens_k"][Bd]
        host[o + db + 1:o + pad + B + 1] = a["cu_seqlens_k"][Bd + 1:]
        host[o + pad + B + 1:o + sb + 1] = a["cu_seqlens_k"][-1]
        o, _ = self._pg_off["temperatures"]
        temps = a["temperatures"].astype(np.float32).view(np.int32)
        host[o:o + Bd] = temps[:Bd]
        host[o + Bd:o + db] = 0
        host[o + db:o + pad + B] = temps[Bd:]
        host[o + pad + B:o + sb] = 0
        if db:
            o, _ = self._pg_off["context_lens"]
            host[o:o + Bd] = a["context_lens"]
            host[o + Bd:o + db] = 0
        host[self._pg_off["misc"][0]] = b.step
        o, _ = self._pg_off["prefill_items"]
        slots = hip.prefill_item_slots(tb, sb)
        items = a["prefill_items"]
        host[o:o + items.size] = items
        host[o + items.size:o + slots] = -1
        o, _ = self._pg_off["block_tables"]
        tab = host[o:o + sb * MB].reshape(sb, MB)
        mb = a["block_tables"].shape[1]
        tab[:] = -1
        tab[:Bd, :mb] = a["block_tables"][:Bd]
        tab[db:pad + B, :mb] = a["block_tables"][Bd:]
        self._pg_dev.copy_(self._pg_host, non_blocking=True)

    def _run_prefill_graph(self, b: Batch, key: tuple):
        """Replay the step's graph; returns (tokens, the step's logits).  A shape seen for the
        first time runs eagerly on the graph's static buffers (that IS this step) and is
        recorded afterwards for the next time.  A capture that fails leaves the step done and
        the shape marked as eager-only."""
        self._fill_prefill_graph_inputs(b, key)
        tb, sb, db = key
        B = b.num_seqs + db - b.num_decode_seqs      # slots in use, unused decode slots included
        b.decode_slots = db
        if key in self.prefill_graphs:
            self.prefill_graphs[key].replay()
            self.prefill_graph_replays += 1
            return self._pg_out[key][:B], self._pg_logits[key]
        tokens = self._pg_step(self._pg_tensors(tb, sb, db), tb, sb, db)
        tokens = tokens[:B].clone() if tokens is not None else None
        logits = self._last_logits
        try:
            self._capture_prefill_graph(key, warm_up=False)
        except Exception as e:                   # noqa: BLE001
            self._pg_failed.add(key)
            for d in (self.prefill_graphs, self._pg_out, self._pg_logits):
                d.pop(key, None)
            if self.rank == 0:
                print(f"[runner] prefill graph {key} could not be captured ({e!r}); such steps "
                      f"launch eagerly", flush=True)
                if os.environ.get("MINIVLLM_DEBUG_CAPTURE") == "1":
                    import traceback
                    traceback.print_exc()
            try:
                torch.cuda.synchronize()
            except Exception:                    # noqa: BLE001
                pass
        self._last_logits = logits
        return tokens, logits
