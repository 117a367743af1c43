#!/bin/bash
# One measurement pass on the GPU box; everything lands in gpurun_out/measure/.  Usage: bash tools/measure_round.sh [a|b|c]
#   a = the default bench.py run (config[1] + sweep + ball), kernel statistics and PMC passes of the config[1] step
#   b = --acgt config[1], the Q = 4 kernel timeline, push-path and ingest timings
#   c = reference-shard emulation (2, 4, 8 contexts on one GPU)
#   d = uvaialign: SQ counters of the aligner, the command line end to end; query preparation timings
# (parts so that each fits one gpurun call; tools/collect_profiles.py turns the results into profiles/r03_*)
set -o pipefail
PART=${1:-abcd}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/measure; mkdir -p $O
step() { echo "== $*"; }
C1="python bench.py --steps 5 --warmup 1 --no-sweep --cpu-refs 0 --no-parity"
if [[ $PART == *a* ]]; then
[ -x tools/hbm_read ] || hipcc --offload-arch=gfx950 -O3 tools/hbm_read.hip -o tools/hbm_read || exit 1
step hbm ceiling;    timeout -k 10 120 ./tools/hbm_read 4 > $O/hbm_read.txt 2>&1 || exit 1
step bench default;  timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || exit 1
step kernel stats;   timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o c1 --output-format csv -- $C1 > $O/stats.log 2>&1 || exit 1
step pmc fetch;      timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- $C1 > $O/pmc_fetch.log 2>&1 || exit 1
step pmc write;      timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- $C1 > $O/pmc_write.log 2>&1 || exit 1
step pmc sq a;       timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM -d $O/pmc_sqa -o a --output-format csv -- $C1 > $O/pmc_sqa.log 2>&1 || exit 1
step pmc sq b;       timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $O/pmc_sqb -o b --output-format csv -- $C1 > $O/pmc_sqb.log 2>&1 || exit 1
step ball profile;   timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/ball -o ball --output-format csv -- python bench.py --ball-only --steps 3 > $O/ball.json 2> $O/ball.err || exit 1
step c1 timeline;    timeout -k 10 300 rocprofv3 --kernel-trace -d $O/c1trace -o c1 --output-format csv -- python bench.py --no-sweep --cpu-refs 0 --no-parity --steps 3 --warmup 1 > $O/c1trace.log 2>&1 || exit 1
step pmc fetch q4;   timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_q4 -o f --output-format csv -- python bench.py --queries 4 --refs 1000000 --pool 1000000 --steps 2 --warmup 1 --no-sweep --cpu-refs 0 --no-parity > $O/pmc_fetch_q4.log 2>&1 || exit 1
python tools/pmc_summary.py config1_fetch=$(ls $O/pmc_fetch/*counter_collection.csv) config1_write=$(ls $O/pmc_write/*counter_collection.csv) q4_1Mrefs_fetch=$(ls $O/pmc_fetch_q4/*counter_collection.csv) config1_sq_a=$(ls $O/pmc_sqa/*counter_collection.csv) config1_sq_b=$(ls $O/pmc_sqb/*counter_collection.csv) > $O/pmc_summary.json || exit 1
fi
if [[ $PART == *b* ]]; then
step acgt config1;   timeout -k 10 300 python bench.py --mode acgt --steps 10 --warmup 2 --no-sweep --cpu-refs 2048 > $O/bench_acgt_c1.json 2> $O/bench_acgt_c1.err || exit 1
step q4 timeline;    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/q4 -o q4 --output-format csv -- python bench.py --queries 4 --refs 1000000 --pool 1000000 --no-sweep --cpu-refs 0 --no-parity --steps 5 --warmup 1 > $O/q4.log 2>&1 || exit 1
step push rate;      timeout -k 10 300 python tools/push_rate.py > $O/push_rate.json 2> $O/push_rate.err || exit 1
step ingest;         timeout -k 10 600 python tools/ingest_bench.py --refs 100000 --queries 100 > $O/ingest.json 2> $O/ingest.err || exit 1
fi
if [[ $PART == *c* ]]; then
for n in 2 4 8; do step emulated reference shards $n; timeout -k 10 500 python bench.py --emulate-refshard $n --steps 5 --warmup 1 > $O/emu_refshard_$n.json 2> $O/emu_refshard_$n.err || exit 1; done
fi
if [[ $PART == *d* ]]; then
step aligner counters;  bash tools/profile_align.sh > $O/align_prof.log 2>&1 || exit 1
step uvaialign cli;     timeout -k 10 600 python tools/align_cli_bench.py > $O/align_cli.json 2> $O/align_cli.err || exit 1
step query preparation; timeout -k 10 300 python tools/prune_timing.py 3000 10000 > $O/prune_timing.txt 2>&1 || exit 1
fi
echo done
