cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tr; mkdir -p $O
n=pre8b_64
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/$n -- python3 $R/bench.py --second-model "" --mode prefill --model qwen3-8b --seq-len 64 --prefill-tokens 64 --steps 8 --warmup 3 > $O/$n.json 2> $O/$n.err || { tail -5 $O/$n.err; exit 1; }
f=$(ls $O/$n/*/*_kernel_trace.csv | head -1)
python3 $R/tools/prof_summary.py $f --marker sample_kernel --last-steps 6 > $O/$n.txt 2>&1
head -24 $O/$n.txt; rm -rf $O/$n; tail -c 300 $O/$n.json
