#!/bin/bash
# Per-rank step of a tensor-parallel engine, measured on ONE GPU (bench.py --tp-shard N): rank 0's
# shard of the full-size model, every collective issued with nobody at the other end.  For each
# TP degree: kernel trace of the prefill steps (16384 tokens) and of the decode steps (256 x 1024)
# with RCCL (single-member group) and, decode only, with the peer-to-peer kernels in loop-back form.
# Output: gpurun_out/shard/*.txt (trace summaries) and *.json (bench lines).
#   tools/tp_shard_rehearsal.sh [model] [tp degrees...]      default: qwen3-8b 2 4 8
# Extra bench.py arguments for every run: env BENCH_ARGS (e.g. "--seq-len 32768 --prefill-tokens 32768 --batch 8")
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/shard; mkdir -p $O
MODEL=${1:-qwen3-8b}; shift; TPS=${@:-2 4 8}
W=${W:-3}; K=${K:-8}; LAST=${LAST:-6}
run() {  # name, extra bench args...
  local name=$1; shift
  timeout -k 10 480 rocprofv3 --kernel-trace --output-format csv -d $O/$name -- python3 $R/bench.py \
      --model $MODEL --second-model "" --steps $K --warmup $W $BENCH_ARGS "$@" > $O/$name.json 2> $O/$name.err \
      || { echo "$name FAILED"; tail -8 $O/$name.err; return 1; }
  f=$(ls $O/$name/*/*_kernel_trace.csv | head -1)
  echo "== $name: $(python3 -c "import json,sys; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('prefill_ms_per_step','decode_ms_per_step','p2p_allreduce','gemm_choice_source','kv_blocks')})")"
  python3 $R/tools/prof_summary.py $f --marker sample_kernel --last-steps $LAST --top 20 > $O/$name.txt 2>&1
  head -24 $O/$name.txt
  rm -rf $O/$name
}
for tp in $TPS; do
  run ${MODEL}_tp${tp}_prefill --tp-shard $tp --mode prefill || exit 1
  run ${MODEL}_tp${tp}_decode --tp-shard $tp --mode decode || exit 1
  if [ -z "$NO_P2P" ]; then run ${MODEL}_tp${tp}_decode_p2p --tp-shard $tp --p2p --mode decode || exit 1; fi
done
