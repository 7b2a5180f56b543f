#!/bin/bash
# Same-box comparison of builds of the kernel library that differ in csrc/attn_prefill.hip:
#   tools/flash_prefill_variants.sh <name> ...   (name "tree" = the library in the tree, any other
#   name = tools/native/_hip_kernels_<name>.so), two rounds, 32 and 16 query heads.
#   EXTRA="--work-list --q-rope": further arguments of tools/attn_kernel_bench.py.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "$@"; do
  if [ $lib = tree ]; then unset MINIVLLM_HIP_LIB; else export MINIVLLM_HIP_LIB=$GRAFT_REPO_ROOT/tools/native/_hip_kernels_$lib.so; fi
  echo "== $lib rep $rep hq32"; timeout -k 10 120 python tools/attn_kernel_bench.py --mode prefill --iters 30 --prefill 16x1024,16x1024,8x2048,4x4096,1x16384 $EXTRA 2>&1 | grep prefill | tail -n +2 || exit 1
  echo "== $lib rep $rep hq16"; timeout -k 10 120 python tools/attn_kernel_bench.py --mode prefill --iters 30 --hq 16 --prefill 16x1024,16x1024,32x512,4x4096 $EXTRA 2>&1 | grep prefill | tail -n +2 || exit 1
done
done
