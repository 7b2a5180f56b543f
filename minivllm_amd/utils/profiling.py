"""Tracing helpers.  The reference only prints wall-clock tokens/s per step
(reference: src/myvllm/engine/llm_engine.py:76-83).  Here:

* ``trace_range(name)`` pushes a roctx range (``torch.cuda.nvtx`` maps to roctx on ROCm), so
  prefill / decode / sampling show up as named spans in ``rocprofv3 --marker-trace`` output;
  it is a no-op unless ``MINIVLLM_TRACE=1`` (or ``enable_tracing()``), so the hot loop pays
  nothing by default;
* ``rocprof_command(...)`` builds the profiler command lines used for the files under
  ``profiles/`` (kernel trace, and PMC passes that respect the per-block counter limits).
"""
from __future__ import annotations

import contextlib
import os
from typing import Iterable, List

_ENABLED = os.environ.get("MINIVLLM_TRACE", "0") == "1"


def enable_tracing(on: bool = True) -> None:
    global _ENABLED
    _ENABLED = on


def tracing_enabled() -> bool:
    return _ENABLED


@contextlib.contextmanager
def trace_range(name: str):
    if not _ENABLED:
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    torch.cuda.nvtx.range_push(name)
    try:
        yield
    finally:
        torch.cuda.nvtx.range_pop()


# PMC passes that fit the gfx950 limits (<= 8 SQ, <= 4 TCC, <= 2 GRBM counters per pass)
PMC_PASSES = {
    "issue": ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
              "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT",
              "GRBM_GUI_ACTIVE"],
    "memory": ["SQ_LDS_IDX_ACTIVE", "SQ_LDS_UNALIGNED_STALL", "SQ_INSTS_VALU_MFMA_MOPS_BF16",
               "SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "TCC_EA0_RDREQ_sum", "TCC_HIT_sum",
               "TCC_MISS_sum"],
}


def rocprof_command(program: Iterable[str], out_dir: str, pmc_pass: str = "") -> List[str]:
    """Command line for one profiler run.  The program itself must follow ``--`` (no ``env``,
    shell or launcher in between) and counter collection is never combined with API tracing."""
    cmd = ["rocprofv3", "--output-format", "csv", "-d", out_dir]
    if pmc_pass:
        cmd += ["--pmc"] + PMC_PASSES[pmc_pass]
    else:
        cmd += ["--kernel-trace", "--stats"]
    return cmd + ["--"] + list(program)
