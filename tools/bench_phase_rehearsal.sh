#!/bin/bash
# Two ranks of ONE bench.py phase sharing the one GPU over gloo (what a launcher's ranks do on a
# node, minus the orchestration): each rank's output goes to gpurun_out/ph_rank{r}.log.
#   tools/bench_phase_rehearsal.sh [phase=tp] [world=2] [extra bench args...]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
PHASE=${1:-tp}; WORLD=${2:-2}; shift; shift
pids=""
for r in $(seq 0 $((WORLD-1))); do
  MINIVLLM_HANG_DUMP=100 MINIVLLM_DIST_BACKEND=gloo MINIVLLM_BENCH_PHASE=$PHASE RANK=$r LOCAL_RANK=0 WORLD_SIZE=$WORLD \
    MASTER_ADDR=127.0.0.1 MASTER_PORT=29777 timeout -k 10 170 python -u bench.py --gpus $WORLD --steps 6 --warmup 2 \
    --layers 4 --batch 64 --kv-blocks 700 --second-model "" "$@" > gpurun_out/ph_rank$r.log 2>&1 &
  pids="$pids $!"
done
for i in $(seq 1 34); do sleep 5; echo "t=$((5*i))s $(tail -c 200 gpurun_out/ph_rank0.log | tr '\n' ' ' | cut -c1-150)"; alive=0; for p in $pids; do kill -0 $p 2>/dev/null && alive=1; done; [ $alive = 0 ] && break; done
wait
for r in $(seq 0 $((WORLD-1))); do echo "=== rank $r"; grep -v "Gloo\|socket.cpp\|amdgpu.ids" gpurun_out/ph_rank$r.log | tail -40 | cut -c1-400; done
