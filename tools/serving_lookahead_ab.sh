#!/bin/bash
# Online serving with / without the prefill step launched one ahead: Qwen3-0.6B, Poisson arrivals
# far above what one step absorbs (rate 0 = all at once, and 400 req/s), phases separated (no mixed
# batches: steps with riders are never launched ahead).
O=gpurun_out/sla; mkdir -p $O
for rate in 0 400; do
  for v in off on; do
    if [ $v = off ]; then export MINIVLLM_NO_PRELAUNCH=1; else unset MINIVLLM_NO_PRELAUNCH; fi
    timeout -k 10 250 python benchmark_serving.py --model qwen3-0.6b --requests 768 --rate $rate --input-len 512 --output-len 32 --max-batched-tokens 8192 --warmup-requests 64 --json $O/$v$rate.json > $O/$v$rate.log 2>&1 || { tail -5 $O/$v$rate.log; exit 1; }
    echo "rate $rate look-ahead $v"; tail -6 $O/$v$rate.log
  done
done
