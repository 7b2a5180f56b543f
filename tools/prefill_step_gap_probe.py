"""Where the host spends the gap between two large prefill steps: cProfile over the engine's
step() on 16 x 1024-token prompts per step (Qwen3-0.6B, eager prefill), plus wall time of the
phases before the first kernel of a step can be launched."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from minivllm_amd.config import EngineConfig, model_preset
from minivllm_amd.engine.llm_engine import LLMEngine
from minivllm_amd.sampling_parameters import SamplingParams

model = model_preset(os.environ.get("MODEL", "qwen3-0.6b"))
cfg = EngineConfig(model=model, max_num_sequences=256, max_num_batched_tokens=16384, block_size=256,
                   eos=-1, max_model_length=2048, gpu_memory_utilization=0.85, dtype=torch.bfloat16,
                   device="cuda", seed=0).validate()
eng = LLMEngine(cfg, spawn_workers=False)
rng = np.random.default_rng(0)
sp = SamplingParams(temperature=0.6, max_tokens=1, ignore_eos=True)
for p in rng.integers(0, model.vocab_size, size=(16 * 14, 1024)).tolist():
    eng.add_prompt(p, sp)
for _ in range(4):
    eng.step()
r = eng.model_runner
marks = {}
for name in ("prepare_prefill", "_upload_prefill", "_fill_prefill_graph_inputs", "_run_prefill_graph"):
    fn = getattr(r, name)
    def wrap(*a, _fn=fn, _n=name, **k):
        t = time.perf_counter(); out = _fn(*a, **k); marks.setdefault(_n, []).append(time.perf_counter() - t); return out
    setattr(r, name, wrap)
sched = eng.scheduler.schedule
def wsched(*a, **k):
    t = time.perf_counter(); out = sched(*a, **k); marks.setdefault("schedule", []).append(time.perf_counter() - t); return out
eng.scheduler.schedule = wsched
post = eng.scheduler.postprocess
def wpost(*a, **k):
    t = time.perf_counter(); out = post(*a, **k); marks.setdefault("postprocess", []).append(time.perf_counter() - t); return out
eng.scheduler.postprocess = wpost
torch.cuda.synchronize()
PROFILE = os.environ.get("CPROFILE") == "1"      # cProfile roughly doubles the host times
pr = cProfile.Profile()
t0 = time.perf_counter()
if PROFILE:
    pr.enable()
for _ in range(8):
    eng.step()
if PROFILE:
    pr.disable()
torch.cuda.synchronize()
print(f"step {1e3 * (time.perf_counter() - t0) / 8:.2f} ms")
for k, v in marks.items():
    print(f"  {k:28s} {1e6 * sum(v) / len(v):8.1f} us per call ({len(v)} calls)")
if PROFILE:
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
eng.exit()
