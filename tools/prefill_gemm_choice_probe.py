"""Prints the start-up choice (library vs csrc/big_gemm.hip) per projection and row bucket for a
model preset, as ModelRunner._tune_prefill_gemms measures it (4 layers are enough: the choice
depends on the projection shapes only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minivllm_amd.engine.model_runner import ModelRunner
from minivllm_amd.config import as_engine_config

for preset in sys.argv[1:] or ["qwen3-0.6b", "qwen3-8b"]:
    cfg = dict(model_preset=preset, num_layers=4, max_num_sequences=16, max_num_batched_tokens=16384,
               block_size=256, max_model_length=2048, eos=-1, max_cached_blocks=256, dtype="bf16",
               device="cuda", enforce_eager=True, tune_decode_gemms=False)
    r = ModelRunner(as_engine_config(cfg), 0)
    print(preset)
    for name, table in r.prefill_gemm_choices.items():
        print("  ", name, {m: v for m, v in table.items()})
    del r
