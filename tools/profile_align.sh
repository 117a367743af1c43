#!/bin/bash
# SQ counters and kernel statistics of wfa_align_kernel on the GPU box (2 000 queries of the config[4] workload); results in gpurun_out/align_prof/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/align_prof; mkdir -p $O
CMD="python bench.py --align-only --align-queries 2000 --align-cpu-queries 0 --steps 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o a --output-format csv -- $CMD > $O/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM -d $O/sqa -o a --output-format csv -- $CMD > $O/sqa.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS -d $O/sqb -o b --output-format csv -- $CMD > $O/sqb.log 2>&1 || exit 1
python tools/pmc_summary.py sq_a=$(ls $O/sqa/*counter_collection.csv) sq_b=$(ls $O/sqb/*counter_collection.csv) > $O/summary.json || exit 1
echo done
