#!/bin/bash
# One measurement pass on the GPU box; everything lands in gpurun_out/measure/.  Usage: bash tools/measure_round.sh [a|b|c|e]
#   a = the default bench.py run (config[1] + sweep + ball + align), kernel statistics and PMC passes of the config[1] step and of a config[2]-shaped step
#   b = --acgt config[1], the Q = 4 and config[1] kernel timelines, push-path and ingest timings
#   c = reference-shard emulation (2, 4, 8 contexts on one GPU; 8 contexts with config[3]'s 10 000 --acgt queries)
#   e = probes: dependent-load latency next to a streaming kernel, tiles the small-query replay opens, the replay's time split (timing build)
# (parts so that each fits one gpurun call; tools/collect_profiles.py turns the results into profiles/r04_*)
set -o pipefail
PART=${1:-abce}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/measure; mkdir -p $O
step() { echo "== $*"; }
C1="python bench.py --steps 5 --warmup 1 --no-sweep --cpu-refs 0 --no-parity --align-queries 0"
C2="python bench.py --queries 10000 --refs 125000 --mode acgt --pool 65536 --steps 2 --warmup 1 --no-sweep --cpu-refs 0 --no-parity --align-queries 0"
pmc() { name=$1; cfg=$2; shift 2
  step pmc $name; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$name -o p --output-format csv -- $cfg > $O/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 $O/pmc_$name.log; exit 1; }
}
if [[ $PART == *a* ]]; then
[ -x tools/hbm_read ] || hipcc --offload-arch=gfx950 -O3 tools/hbm_read.hip -o tools/hbm_read || exit 1
step hbm ceiling;    timeout -k 10 120 ./tools/hbm_read 4 > $O/hbm_read.txt 2>&1 || exit 1
step bench default;  timeout -k 10 700 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
step kernel stats;   timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o c1 --output-format csv -- $C1 > $O/stats.log 2>&1 || exit 1
step kernel stats config2; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats2 -o c2 --output-format csv -- $C2 > $O/stats2.log 2>&1 || exit 1
pmc c1_fetch "$C1" FETCH_SIZE
pmc c1_write "$C1" WRITE_SIZE
pmc c1_insts_a "$C1" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
pmc c1_waits "$C1" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pmc c1_units "$C1" SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_IFETCH SQ_INST_LEVEL_SMEM
pmc c1_tcc "$C1" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc c2_fetch "$C2" FETCH_SIZE
pmc c2_write "$C2" WRITE_SIZE
pmc c2_insts_a "$C2" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
pmc c2_waits "$C2" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pmc c2_tcc "$C2" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc q4_fetch "python bench.py --queries 4 --refs 1000000 --pool 1000000 --steps 2 --warmup 1 --no-sweep --cpu-refs 0 --no-parity --align-queries 0" FETCH_SIZE
Q16="python bench.py --queries 16 --refs 1000000 --pool 1000000 --steps 2 --warmup 1 --no-sweep --cpu-refs 0 --no-parity --align-queries 0"
pmc q16_insts "$Q16" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
pmc q16_waits "$Q16" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
step ball profile;   timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/ball -o ball --output-format csv -- python bench.py --ball-only --steps 3 > $O/ball.json 2> $O/ball.err || exit 1
python tools/pmc_summary.py $(for d in c1_fetch c1_write c1_insts_a c1_waits c1_units c1_tcc c2_fetch c2_write c2_insts_a c2_waits c2_tcc q4_fetch q16_insts q16_waits; do echo $d=$(ls $O/pmc_$d/*counter_collection.csv | head -1); done) > $O/pmc_summary.json || exit 1
fi
if [[ $PART == *b* ]]; then
step acgt config1;   timeout -k 10 300 python bench.py --mode acgt --steps 10 --warmup 2 --no-sweep --cpu-refs 2048 --align-queries 0 > $O/bench_acgt_c1.json 2> $O/bench_acgt_c1.err || exit 1
step c1 timeline;    timeout -k 10 300 rocprofv3 --kernel-trace -d $O/c1trace -o c1 --output-format csv -- python bench.py --no-sweep --cpu-refs 0 --no-parity --align-queries 0 --steps 3 --warmup 1 > $O/c1trace.log 2>&1 || exit 1
step q4 timeline;    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/q4 -o q4 --output-format csv -- python bench.py --queries 4 --refs 1000000 --pool 1000000 --no-sweep --cpu-refs 0 --no-parity --align-queries 0 --steps 5 --warmup 1 > $O/q4.log 2>&1 || exit 1
step q16 timeline;   timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/q16 -o q16 --output-format csv -- python bench.py --queries 16 --refs 1000000 --pool 1000000 --no-sweep --cpu-refs 0 --no-parity --align-queries 0 --steps 5 --warmup 1 > $O/q16.log 2>&1 || exit 1
step push rate;      timeout -k 10 300 python tools/push_rate.py > $O/push_rate.json 2> $O/push_rate.err || exit 1
step ingest;         timeout -k 10 600 python tools/ingest_bench.py --refs 100000 --queries 100 > $O/ingest.json 2> $O/ingest.err || exit 1
fi
if [[ $PART == *c* ]]; then
for n in 2 4 8; do step emulated reference shards $n; timeout -k 10 500 python bench.py --emulate-refshard $n --steps 5 --warmup 1 > $O/emu_refshard_$n.json 2> $O/emu_refshard_$n.err || exit 1; done
step emulated config3 regime; timeout -k 10 600 python bench.py --emulate-refshard 8 --queries 10000 --mode acgt --refs 32768 --steps 2 --warmup 1 > $O/emu_refshard_8_config3.json 2> $O/emu_refshard_8_config3.err || { tail -5 $O/emu_refshard_8_config3.err; exit 1; }
fi
if [[ $PART == *e* ]]; then
[ -x tools/load_latency ] || hipcc --offload-arch=gfx950 -O3 tools/load_latency.hip -o tools/load_latency || exit 1
step load latency;   timeout -k 10 200 ./tools/load_latency 4 > $O/load_latency.txt 2>&1 || exit 1
step probe q4;       timeout -k 10 250 python tools/r04_probe.py 4 1000000 > $O/probe_q4.txt 2>&1 || exit 1
if [ -f uvaia_amd/lib/variants/libuvaia_gpu_timing.so ]; then
  cp uvaia_amd/lib/libuvaia_gpu.so /tmp/libuvaia_gpu_product.so && cp uvaia_amd/lib/variants/libuvaia_gpu_timing.so uvaia_amd/lib/libuvaia_gpu.so || exit 1
  step replay timing q4;  timeout -k 10 250 python tools/r04_probe.py 4 1000000 > $O/probe_q4_timing.txt 2>&1
  step replay timing c1;  timeout -k 10 250 python tools/r04_probe_c1.py > $O/probe_c1_timing.txt 2>&1
  cp /tmp/libuvaia_gpu_product.so uvaia_amd/lib/libuvaia_gpu.so
fi
fi
echo done
