#!/bin/bash
# Long generations with varied lengths (benchmark_throughput.py: 256 requests, prompts 256-768
# tokens, 384-1152 generated tokens, 256 sequence slots: one decode batch whose sequences cross
# KV-block edges at different steps and finish one by one).  Decode steps launched ahead only
# where no block edge is touched and nobody waits (MINIVLLM_NO_EDGE_PRELAUNCH=1) against the
# default.  Same box.  MODELS="qwen3-0.6b qwen3-8b" by default.
O=gpurun_out/edge; mkdir -p $O
for model in ${MODELS:-qwen3-0.6b qwen3-8b}; do
  for v in off on off on; do
    if [ $v = off ]; then export MINIVLLM_NO_EDGE_PRELAUNCH=1; else unset MINIVLLM_NO_EDGE_PRELAUNCH; fi
    timeout -k 10 280 python benchmark_throughput.py --model $model --requests 256 --output-len 768 --json $O/${model}_$v.json > $O/${model}_$v.log 2>&1 || { tail -5 $O/${model}_$v.log; exit 1; }
    python - $O/${model}_$v.json $model $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "beyond plain steps:", sys.argv[3], {k: d[k] for k in d if k in ("wall_s", "requests_per_s", "generated_tok_s", "prefill_steps", "decode_steps")}, flush=True)
PY
  done
done
