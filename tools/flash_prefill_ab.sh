#!/bin/bash
# Same-box A/B of the flash prefill kernel: the library in the tree against a baseline library
# (tools/native/_hip_kernels_base.so, built from another revision of csrc/attn_prefill.hip),
# after the kernel's numerics tests.  The first case of a process runs before the GPU's clocks
# have ramped up: every process measures its first shape twice and the first line is dropped.
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
timeout -k 10 500 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_hip_kernels.py tests/test_gpu_engine_logits.py -k "prefill or flash or logit or attention" > gpurun_out/ab/tests.log 2>&1
rc=$?; tail -3 gpurun_out/ab/tests.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; exit $rc; fi
for rep in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export MINIVLLM_HIP_LIB=$GRAFT_REPO_ROOT/tools/native/_hip_kernels_base.so; else unset MINIVLLM_HIP_LIB; fi
  echo "== $lib rep $rep hq32"; timeout -k 10 120 python tools/attn_kernel_bench.py --mode prefill --iters 30 --prefill 16x1024,16x1024,8x2048,4x4096,1x16384 2>&1 | grep prefill | tail -n +2 || exit 1
  echo "== $lib rep $rep hq16"; timeout -k 10 120 python tools/attn_kernel_bench.py --mode prefill --iters 30 --hq 16 --prefill 16x1024,16x1024,32x512,4x4096 2>&1 | grep prefill | tail -n +2 || exit 1
done
done
