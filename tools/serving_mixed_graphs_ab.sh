#!/bin/bash
# Online serving with mixed batches (decode rows ride along in every prefill step), Poisson arrivals:
# prefill / mixed steps launched eagerly (--no-lazy-prefill-graphs) against captured on first use.
#   tools/serving_mixed_graphs_ab.sh [model=qwen3-0.6b] [rate=64] [requests=512]
cd "$(dirname "$0")/.."
M=${1:-qwen3-0.6b}; R=${2:-64}; N=${3:-512}
for extra in "--no-lazy-prefill-graphs" ""; do
  echo "== $M rate $R ${extra:-graphs}"
  timeout -k 10 400 python -u benchmark_serving.py --model $M --requests $N --rate $R --input-len 512 --output-len 128 \
      --mixed --warmup-requests ${WARM:-64} $extra 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print({k: d[k] for k in ('achieved_req_s', 'generated_tok_s', 'ttft_ms', 'tpot_ms', 'prefill_steps', 'decode_steps', 'prefill_graphs', 'prefill_graph_replays')})" || exit 1
done
