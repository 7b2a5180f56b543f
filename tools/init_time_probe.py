"""Where engine start-up time goes (cProfile, top cumulative entries)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivllm_amd.config import EngineConfig, model_preset
from minivllm_amd.engine.llm_engine import LLMEngine

name = sys.argv[1] if len(sys.argv) > 1 else "qwen3-0.6b"
torch.zeros(1, device="cuda")                      # HIP context outside the measurement
cfg = EngineConfig(model=model_preset(name), max_num_sequences=256, max_num_batched_tokens=16384,
                   block_size=256, eos=-1, max_model_length=2048, dtype=torch.bfloat16,
                   device="cuda").validate()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
eng = LLMEngine(cfg)
pr.disable()
print(f"{name}: engine start-up {time.perf_counter() - t0:.2f} s", flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
eng.exit()
