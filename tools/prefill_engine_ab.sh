#!/bin/bash
# Engine-level same-box A/B of two builds of the kernel library on the prefill step
# (bench.py --mode prefill, Qwen3-0.6B and Qwen3-8B): "tree" = the library in the tree,
# any other name = tools/native/_hip_kernels_<name>.so.   tools/prefill_engine_ab.sh base tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for rep in 1 2; do
for lib in "$@"; do
  if [ $lib = tree ]; then unset MINIVLLM_HIP_LIB; else export MINIVLLM_HIP_LIB=$GRAFT_REPO_ROOT/tools/native/_hip_kernels_$lib.so; fi
  for model in qwen3-0.6b qwen3-8b; do
    timeout -k 10 300 python bench.py --model $model --second-model "" --mode prefill --steps 10 --warmup 3 > gpurun_out/ab/pe_${lib}_${model}_$rep.json 2> gpurun_out/ab/pe_${lib}_${model}_$rep.err || { tail -5 gpurun_out/ab/pe_${lib}_${model}_$rep.err; exit 1; }
    python -c "
import json,sys
d=json.load(open('gpurun_out/ab/pe_${lib}_${model}_$rep.json'))
print('$lib rep $rep $model prefill ms/step', d.get('prefill_ms_per_step'))"
  done
done
done
