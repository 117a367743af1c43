#!/bin/bash
# Round 4: what the 16-query packed-plane scan (scan2_iupac_kernel<16>) waits for.  bash tools/r04_pmc_q16.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04/pmc_q16; mkdir -p $O
CFG="python bench.py --queries 16 --refs 1000000 --pool 1000000 --steps 2 --warmup 1 --no-sweep --cpu-refs 0 --no-parity --align-queries 0"
pass() { name=$1; shift
  echo "== $name"; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o p --output-format csv -- $CFG > $O/$name.log 2>&1 || { echo "$name failed"; tail -5 $O/$name.log; return 1; }
}
pass sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU || exit 1
pass sq_b SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_LEVEL_SMEM || exit 1
python tools/pmc_summary.py sq_a=$(ls $O/sq_a/*counter_collection.csv | head -1) sq_b=$(ls $O/sq_b/*counter_collection.csv | head -1) > $O/summary.json || exit 1
python - $O/summary.json <<'P'
import json, sys
s = json.load(open(sys.argv[1]))
for blk, ks in s.items():
    for k, v in ks.items():
        if "scan2" in k:
            print(blk, k[:40], {c: (round(x["sum"]), x["n"]) for c, x in v.items()})
P
