#!/bin/bash
# Round 4: what scan3_kernel waits for -- SQ wait/active split, LDS, L2 hits and misses -- at config[1] and in config[2]'s regime.
# bash tools/r04_pmc.sh  (one box; separate --pmc passes, kernel trace only: the pool refuses other combinations)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04/pmc; mkdir -p $O
C1="python bench.py --steps 3 --warmup 1 --no-sweep --cpu-refs 0 --no-parity --align-queries 0"
C2="python bench.py --queries 10000 --refs 131072 --mode acgt --pool 65536 --steps 2 --warmup 1 --no-sweep --cpu-refs 0 --no-parity --align-queries 0"
pass() { name=$1; cfg=$2; shift 2
  echo "== $name"; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o p --output-format csv -- $cfg > $O/$name.log 2>&1 || { echo "$name failed"; tail -5 $O/$name.log; return 1; }
}
pass c1_sq_a "$C1" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU || exit 1
pass c1_sq_b "$C1" SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT || exit 1
pass c1_sq_c "$C1" SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_LDS_ATOMIC SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM || exit 1
pass c1_tcc "$C1" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum || exit 1
pass c2_sq_a "$C2" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU || exit 1
pass c2_tcc "$C2" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum || exit 1
pass c2_fetch "$C2" FETCH_SIZE WRITE_SIZE || exit 1
python tools/pmc_summary.py $(for d in c1_sq_a c1_sq_b c1_sq_c c1_tcc c2_sq_a c2_tcc c2_fetch; do echo $d=$(ls $O/$d/*counter_collection.csv | head -1); done) > $O/summary.json || exit 1
cat $O/summary.json | head -c 6000
