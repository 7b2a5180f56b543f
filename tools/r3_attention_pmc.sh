#!/bin/bash
# Counter passes over the two attention kernels in the form the engine's steps use them:
# flash prefill from a work list with q-norm + rotation on load (16 x 1024 tokens) and paged decode
# that norms / rotates / stores the new token itself (256 x 1024).  One rocprofv3 --pmc run per
# pass; per-kernel means by tools/pmc_summary.py.  Output: gpurun_out/pmc_attn/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_attn; mkdir -p $O
pass() {  # name counters...
  n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $O/$n -- python3 $R/tools/attn_kernel_bench.py --iters 3 --prefill 16x1024 --decode 256x1024 --work-list --q-rope --implicit-positions --fused-store > $O/$n.log 2>&1 || { tail -3 $O/$n.log; return 1; }
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE &&
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE &&
pass c TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE &&
python3 $R/tools/pmc_summary.py $(ls $O/*/*/*counter_collection.csv) flash_prefill paged_decode > $O/summary.txt 2>&1
cat $O/summary.txt; rm -rf $O/a $O/b $O/c
