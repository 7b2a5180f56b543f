"""Prints, per model preset given on the command line, what the start-up measurement of
``ModelRunner._tune_decode_gemms`` chose per projection and row bucket (library GEMM us, weight-
streaming kernel us, kernel chosen)."""
import sys, time, torch
sys.path.insert(0, '.')
from minivllm_amd.config import EngineConfig, model_preset
from minivllm_amd.engine.model_runner import ModelRunner
for name in sys.argv[1:]:
    cfg = EngineConfig(model=model_preset(name), max_num_sequences=16, max_num_batched_tokens=2048, block_size=256, eos=-1,
                       enforce_eager=True, max_model_length=2048, dtype=torch.bfloat16, device="cuda", max_cached_blocks=64).validate()
    t0 = time.time(); r = ModelRunner(cfg, 0); dt = time.time() - t0
    print(name, f"init {dt:.1f}s")
    for k, v in r.decode_gemm_choices.items():
        print("   ", k, v)
    r.exit()
