#!/bin/bash
# Round 4, config[1] experiments on one box: launch tails and overlapping scans.  bash tools/r04_c1_exp.sh [tag]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-c1exp}; O=gpurun_out/r04/$TAG; mkdir -p $O
run() { name=$1; shift
  timeout -k 10 240 python bench.py --steps 10 --warmup 3 --no-sweep --cpu-refs 0 --no-parity --align-queries 0 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; return 1; }
  python - $O/$name.json "$name" <<'P'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-34s ms/step %7.3f  value %12.1f  scan launch ms %7.4f x %d" % (sys.argv[2], b["ms_per_step"], b["value"], b["roofline"].get("avg_launch_ms", 0), b["roofline"].get("launches", 0)), flush=True)
P
}
for rep in 1 2; do
run base_$rep || exit 1
run streams3_$rep --tuning scan_streams=3 || exit 1
run streams2_$rep --tuning scan_streams=2 || exit 1
run refs98304_sub32768_$rep --refs 98304 --subslice 32768 || exit 1
run refs98304_sub32768_streams3_$rep --refs 98304 --subslice 32768 --tuning scan_streams=3 || exit 1
run sub25088_$rep --subslice 25088 || exit 1
run sub25088_streams3_$rep --subslice 25088 --tuning scan_streams=3 || exit 1
done
