"""Tuned library-GEMM selection.

The projections are plain library GEMMs (hipBLASLt / rocBLAS through ``F.linear``).  The
library's default heuristic is good for large square problems but leaves a lot on the table
for serving shapes: on MI355X the decode GEMMs of Qwen3-8B at M = 128..256 run 1.2-1.6x faster
with a different solution (see ``profiles/r1_tunableop_probe.txt``).  PyTorch's TunableOp can
search the solutions once and replay the choice from a table; ``tools/tune_gemms.py`` builds
that table on an MI355X (with rotating buffers, so that weights stream from HBM as they do in
serving) and this module loads it at engine start-up.  Shapes that are not in the table, or a
table written by a different library version, fall back to the default heuristic.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

DEFAULT_TABLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                             "tuned", "tunableop_gfx950.csv")
_LOADED: Optional[str] = None


def enable_tuned_gemms(path: Optional[str] = None) -> bool:
    """Load a TunableOp result table (no tuning at run time).  Returns True when active."""
    global _LOADED
    import torch
    path = path or os.environ.get("MINIVLLM_GEMM_TABLE", DEFAULT_TABLE)
    if not torch.cuda.is_available() or not os.path.exists(path):
        return False
    if _LOADED == path:
        return True
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(False)
    tunable.record_untuned_enable(False)
    ok = bool(tunable.read_file(path))
    # whatever the library writes back at exit must not touch the shipped table
    tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"),
                                      f"minivllm_tunableop_{os.getpid()}.csv"))
    if ok:
        _LOADED = path
    else:
        tunable.enable(False)
    return ok


def write_table(path: str) -> int:
    """Dump the in-memory TunableOp results (validators + one line per tuned shape)."""
    import torch.cuda.tunable as tunable
    rows = tunable.get_results()
    with open(path, "w") as f:
        for key, val in tunable.get_validators():
            f.write(f"Validator,{key},{val}\n")
        for op, params, solution, t in rows:
            f.write(f"{op},{params},{solution},{t}\n")
    return len(rows)


def model_gemm_shapes(model, tp: int = 1) -> List[Tuple[str, int, int]]:
    """(name, N, K) of the weight matrices of one decoder layer + LM head, per TP rank."""
    from minivllm_amd.layers.embedding_head import padded_vocab
    h = model.hidden_size
    qkv = (model.num_heads + 2 * model.num_kv_heads) * model.head_dim // tp
    vocab = (padded_vocab(model.vocab_size, tp) if tp > 1 else model.vocab_size) // tp
    return [("qkv", qkv, h), ("o", h, model.num_heads * model.head_dim // tp),
            ("gate_up", 2 * model.intermediate_size // tp, h),
            ("down", h, model.intermediate_size // tp), ("lm_head", vocab, h)]
