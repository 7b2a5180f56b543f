#!/bin/bash
# Counter passes over the flash prefill kernel alone (1 x 16384 tokens, 32/8 heads, paged KV), one
# rocprofv3 --pmc run per pass (the hardware holds at most 8 SQ counters per run), then the
# per-kernel means (tools/pmc_summary.py).  Output: gpurun_out/pmc_flash/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_flash; mkdir -p $O
pass() {  # name counters...
  n=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $O/$n -- python3 $R/tools/attn_kernel_bench.py --mode prefill --iters 3 --prefill 1x16384 > $O/$n.log 2>&1 || { tail -3 $O/$n.log; return 1; }
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE &&
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE &&
python3 $R/tools/pmc_summary.py $(ls $O/*/*/*counter_collection.csv) flash_prefill > $O/summary.txt 2>&1
cat $O/summary.txt; rm -rf $O/a $O/b
