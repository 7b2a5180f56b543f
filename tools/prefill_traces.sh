# Kernel traces of the prefill steps (16 x 1024-token prompts per step, graphs replayed, steps
# launched one ahead): bench.py --mode prefill for Qwen3-0.6B and Qwen3-8B, last 6 steps.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tr; mkdir -p $O
run() {  # name extra-args
  n=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/$n -- python3 $R/bench.py --second-model "" --mode prefill --steps 8 --warmup 3 "$@" > $O/$n.json 2> $O/$n.err || { tail -5 $O/$n.err; return 1; }
  f=$(ls $O/$n/*/*_kernel_trace.csv | head -1)
  python3 $R/tools/prof_summary.py $f --marker sample_kernel --last-steps 6 > $O/$n.txt 2>&1
  head -14 $O/$n.txt; rm -rf $O/$n
}
run pre06b --model qwen3-0.6b && run pre8b --model qwen3-8b
