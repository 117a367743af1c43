#!/bin/bash
# One measurement pass on the GPU box; everything lands in gpurun_out/measure/.  Usage: bash tools/measure_round.sh [a|b]
# (two halves so that each fits one gpurun call: a = config[1] bench, kernel stats, PMC passes; b = sweeps, config[2], emulated shards)
set -o pipefail
PART=${1:-ab}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/measure; mkdir -p $O
step() { echo "== $*"; }
if [[ $PART == *a* ]]; then
[ -x tools/hbm_read ] || hipcc --offload-arch=gfx950 -O3 tools/hbm_read.hip -o tools/hbm_read || exit 1
step hbm ceiling;    timeout -k 10 120 ./tools/hbm_read 4 > $O/hbm_read.txt 2>&1 || exit 1
step bench config1;  timeout -k 10 400 python bench.py --steps 5 --warmup 1 > $O/bench_c2.json 2> $O/bench_c2.err || exit 1
step kernel stats;   timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o c2 --output-format csv -- python bench.py --steps 5 --warmup 1 --cpu-refs 0 > $O/stats.log 2>&1 || exit 1
step pmc fetch;      timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-refs 0 > $O/pmc_fetch.log 2>&1 || exit 1
step pmc write;      timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-refs 0 > $O/pmc_write.log 2>&1 || exit 1
step pmc fetch q4;   UVAIA_GPU_SCAN=compressed timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_q4 -o f --output-format csv -- python bench.py --queries 4 --refs 1000000 --pool 1000000 --steps 2 --warmup 1 --cpu-refs 0 > $O/pmc_fetch_q4.log 2>&1 || exit 1
step pmc sq a;       timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM -d $O/pmc_sqa -o a --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-refs 0 > $O/pmc_sqa.log 2>&1 || exit 1
step pmc sq b;       timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $O/pmc_sqb -o b --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-refs 0 > $O/pmc_sqb.log 2>&1 || exit 1
step acgt config1;   timeout -k 10 300 python bench.py --mode acgt --steps 5 --warmup 1 --cpu-refs 512 > $O/bench_acgt_c2.json 2> $O/bench_acgt_c2.err || exit 1
fi
if [[ $PART == *b* ]]; then
for q in 1 4 16 64; do step sweep q=$q; timeout -k 10 300 python bench.py --queries $q --refs 1000000 --pool 1000000 --steps 5 --warmup 1 --cpu-refs 0 > $O/sweep_q$q.json 2> $O/sweep_q$q.err || exit 1; done
step c3;             timeout -k 10 500 python bench.py --mode acgt --queries 10000 --refs 1000000 --steps 1 --warmup 1 --cpu-refs 0 > $O/bench_c3.json 2> $O/bench_c3.err || exit 1
for n in 2 4 8; do step emulated shard of $n; timeout -k 10 300 python bench.py --emulate-shard-of $n --steps 5 --warmup 1 --cpu-refs 0 > $O/emu_$n.json 2> $O/emu_$n.err || exit 1; done
fi
[[ $PART == *a* ]] && python tools/pmc_summary.py config1_fetch=$(ls $O/pmc_fetch/*counter_collection.csv) config1_write=$(ls $O/pmc_write/*counter_collection.csv) q4_1Mrefs_fetch=$(ls $O/pmc_fetch_q4/*counter_collection.csv) config1_sq_a=$(ls $O/pmc_sqa/*counter_collection.csv) config1_sq_b=$(ls $O/pmc_sqb/*counter_collection.csv) > $O/pmc_summary.json
echo done
