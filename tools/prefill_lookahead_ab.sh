#!/bin/bash
# Same-box A/B of the prefill step launched one ahead (bench.py --mode prefill, Qwen3-8B and
# Qwen3-0.6B, 16 x 1024-token prompts per step): MINIVLLM_NO_PRELAUNCH=1 against the default, twice.
O=gpurun_out/pla; mkdir -p $O
for r in 1 2; do
  for v in off on; do
    if [ $v = off ]; then export MINIVLLM_NO_PRELAUNCH=1; else unset MINIVLLM_NO_PRELAUNCH; fi
    timeout -k 10 200 python bench.py --mode prefill --steps 20 --warmup 5 > $O/$v$r.json 2> $O/$v$r.err || { tail -3 $O/$v$r.err; exit 1; }
    python - $O/$v$r.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"look-ahead {sys.argv[2]:3s}: qwen3-8b {d['prefill_ms_per_step']:.2f} ms/step   qwen3-0.6b {d['secondary']['prefill_ms_per_step']:.3f} ms/step", flush=True)
PY
  done
done
