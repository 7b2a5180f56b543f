#!/bin/bash
# The whole N > 1 flow of bench.py (launcher -> orchestrating ranks -> one child per phase and
# rank -> one JSON line) with the ranks sharing ONE GPU over gloo: checks the plumbing (votes,
# time budget, assembly), not speed.  Usage: tools/bench_orchestration_rehearsal.sh [ranks] [extra bench args]
N=${1:-2}; shift
export MINIVLLM_DIST_BACKEND=gloo
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 \
  --master-port 29610 bench.py --gpus "$N" --steps 4 --warmup 2 --model qwen3-0.6b --second-model '' \
  --batch 32 --prefill-tokens 4096 --long-context off --kv-blocks 400 "$@"
