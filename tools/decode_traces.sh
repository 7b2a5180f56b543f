cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tr; mkdir -p $O
run() {  # name extra-args
  n=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/$n -- python3 $R/bench.py --second-model "" --mode decode --steps 8 --warmup 3 "$@" > $O/$n.json 2> $O/$n.err || { tail -5 $O/$n.err; return 1; }
  f=$(ls $O/$n/*/*_kernel_trace.csv | head -1)
  python3 $R/tools/prof_summary.py $f --marker sample_kernel --last-steps 6 > $O/$n.txt 2>&1
  head -12 $O/$n.txt; rm -rf $O/$n
}
[ "$1" = p2p ] || { run dec8b --model qwen3-8b && run dec8b_tp8 --model qwen3-8b --tp-shard 8 && run dec8b_b1 --model qwen3-8b --batch 1; }
# the same rank with the peer-to-peer all-reduce kernels in loop-back form
[ "$1" = p2p ] && run dec8b_tp8_p2p --model qwen3-8b --tp-shard 8 --p2p
