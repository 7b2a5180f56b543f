#!/bin/bash
# Kernel traces of the three headline steps (rocprofv3 --kernel-trace) and their summaries:
# Qwen3-0.6B and Qwen3-8B prefill (16 x 1024), Qwen3-8B decode (256 x 1024).  Output: gpurun_out/tr/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tr; mkdir -p $O
run() {  # name model mode marker
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/$1 -- python3 $R/bench.py --model $2 --second-model "" --mode $3 --steps 8 --warmup 3 > $O/$1.json 2> $O/$1.err || { tail -5 $O/$1.err; return 1; }
  f=$(ls $O/$1/*/*_kernel_trace.csv | head -1)
  python3 $R/tools/prof_summary.py $f --marker sample_kernel --last-steps 6 > $O/$1.txt 2>&1
  head -16 $O/$1.txt; rm -rf $O/$1
}
run prefill_06b qwen3-0.6b prefill && run prefill_8b qwen3-8b prefill && run decode_8b qwen3-8b decode
