"""Where does the host spend its time in a decode step?  Wraps the phases of LLMEngine.step with
timers (GPU wait = time blocked in the token D2H copy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from minivllm_amd.config import EngineConfig, model_preset
from minivllm_amd.engine.llm_engine import LLMEngine
from minivllm_amd.sampling_parameters import SamplingParams

model = sys.argv[1] if len(sys.argv) > 1 else "qwen3-0.6b"
# "churn": every step goes the long way (full scheduling loop, batch arrays rebuilt from the
# sequences), as when the batch changes from step to step; default: the steady-state shortcuts
CHURN = len(sys.argv) > 2 and sys.argv[2] == "churn"
B, S, K = 256, 1024, 40
cfg = EngineConfig(model=model_preset(model), max_num_sequences=B, max_num_batched_tokens=16384,
                   block_size=256, eos=-1, max_model_length=2048, gpu_memory_utilization=0.85,
                   graph_batch_sizes=[B])
eng = LLMEngine(cfg)
rng = np.random.default_rng(0)
sp = SamplingParams(temperature=0.6, max_tokens=K + 8, ignore_eos=True)
for p in rng.integers(0, 150000, size=(B, S)).tolist():
    eng.add_prompt(p, sp)
while eng.scheduler.waiting or eng.model_runner.has_prefill_ahead():
    eng.step()
for _ in range(4):
    eng.step()
T = dict(schedule=0.0, prepare=0.0, fill_launch=0.0, gpu_wait=0.0, post=0.0)
r = eng.model_runner
torch.cuda.synchronize()
t_all = time.perf_counter()
for _ in range(K):
    if CHURN:
        eng.scheduler._decode_batch, r._decode_prev = None, None
    t0 = time.perf_counter()
    seqs, is_prefill = eng.scheduler.schedule()
    t1 = time.perf_counter()
    batch = r.prepare_decode(seqs)
    r.step_counter += 1; batch.step = r.step_counter
    t2 = time.perf_counter()
    with torch.inference_mode():
        bucket = r._graph_bucket(batch.num_seqs)
        r._fill_decode_host(batch, bucket)
        r._dec_dev.copy_(r._dec_host, non_blocking=True)
        r.graphs[bucket].replay()
        toks = r._graph_out[bucket][:batch.num_seqs]
        t3 = time.perf_counter()
        toks = toks.tolist()
    t4 = time.perf_counter()
    eng.scheduler.postprocess(seqs, toks)
    t5 = time.perf_counter()
    for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        T[k] += v
tot = time.perf_counter() - t_all
print(f"{model}{' (churn)' if CHURN else ''}: {tot / K * 1e3:.3f} ms/step;", {k: f"{v / K * 1e6:.0f} us" for k, v in T.items()})
eng.exit()
