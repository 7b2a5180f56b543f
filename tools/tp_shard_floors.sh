#!/bin/bash
# Step times of one rank of a Qwen3-8B engine at TP = 1, 2, 4, 8 (shard rehearsal, no profiler): prefill 16 x 1024 and decode
# 256 x 1024, collectives through a single-member RCCL group and through the peer-to-peer kernels in loop-back form.
for tp in 1 2 4 8; do
  for mode in "" "--p2p"; do
    [ $tp = 1 ] && [ -n "$mode" ] && continue
    a="--tp-shard $tp"; [ $tp = 1 ] && a=""
    timeout -k 10 300 python bench.py --model qwen3-8b --second-model "" $a $mode 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tp $tp $mode', 'prefill', d.get('prefill_ms_per_step'), 'decode', d.get('decode_ms_per_step'))" || exit 1
  done
done
