#!/usr/bin/env bash
# 1/2/4/8-GPU sweep of the headline benchmark plus the attention micro-benchmarks on one
# MI355X node (the counterpart of the reference's Slurm launcher run_gb200_benchmark.sh, which
# runs main.py and the two attention benchmarks on one node and logs to logs/).
#
#   tools/run_mi355x_benchmark.sh [OUT_DIR] [GPU_COUNTS...]      e.g.  ... logs 1 2 4 8
set -euo pipefail
cd "$(dirname "$0")/.."
OUT=${1:-logs/$(date +%Y%m%d_%H%M%S)}
shift || true
COUNTS=("$@")
[ ${#COUNTS[@]} -eq 0 ] && COUNTS=(1 2 4 8)
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0            # dmabuf IPC for RCCL / cross-process tensors
export NCCL_DEBUG=${NCCL_DEBUG:-WARN}
export TORCH_NCCL_HEARTBEAT_TIMEOUT_SEC=${TORCH_NCCL_HEARTBEAT_TIMEOUT_SEC:-1800}
python -m minivllm_amd.build
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
echo "node has $NGPU GPU(s); results in $OUT"
for N in "${COUNTS[@]}"; do
  if [ "$N" -gt "$NGPU" ]; then echo "skip N=$N (only $NGPU GPUs)"; continue; fi
  echo "== bench.py on $N GPU(s)"
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 | tee "$OUT/bench_n1.json"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 \
      --master-port $((29500 + N)) bench.py --gpus "$N" | tee "$OUT/bench_n$N.json"
  fi
done
echo "== attention micro-benchmarks (1 GPU)"
python benchmark_prefilling.py --json "$OUT/benchmark_prefilling.json" > "$OUT/benchmark_prefilling.log" 2>&1
python benchmark_decoding.py   --json "$OUT/benchmark_decoding.json"   > "$OUT/benchmark_decoding.log" 2>&1
python benchmark_activation.py --json "$OUT/benchmark_activation.json" > "$OUT/benchmark_activation.log" 2>&1
python benchmark_throughput.py --json "$OUT/benchmark_throughput.json" > "$OUT/benchmark_throughput.log" 2>&1
python benchmark_serving.py --rate 32 --mixed --json "$OUT/benchmark_serving.json" > "$OUT/benchmark_serving.log" 2>&1
python benchmark_models.py --config qwen3-8b --mode all --quick --json "$OUT/benchmark_models.json" \
  > "$OUT/benchmark_models.log" 2>&1
echo "== end-to-end demo"
python main.py > "$OUT/main.log" 2>&1
tail -3 "$OUT/main.log"
echo "done: $OUT"
