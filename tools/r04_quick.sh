#!/bin/bash
# quick look at the headline and the handful-of-queries entries on one box: bash tools/r04_quick.sh [tag]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-quick}; O=gpurun_out/r04/$TAG; mkdir -p $O
run() { name=$1; shift
  timeout -k 10 240 python bench.py --steps 10 --warmup 3 --no-sweep --cpu-refs 0 --align-queries 0 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; return 1; }
  python - $O/$name.json "$name" <<'P'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-22s ms/step %7.3f  value %12.1f  scan launch ms %7.4f x %d  derive %.3f search %.3f  parity %s" % (sys.argv[2], b["ms_per_step"], b["value"], b["roofline"].get("avg_launch_ms", 0), b["roofline"].get("launches", 0),
      b["step_parts"]["derived_planes_ms"], b["step_parts"]["scan_and_replay_ms"], b.get("parity_check_on_timed_path")), flush=True)
P
}
run c1_a || exit 1
run q1 --queries 1 --refs 1000000 --pool 1000000 || exit 1
run q4 --queries 4 --refs 1000000 --pool 1000000 || exit 1
run q16 --queries 16 --refs 1000000 --pool 1000000 || exit 1
run q64 --queries 64 --refs 1000000 --pool 1000000 --steps 5 || exit 1
run c1_b || exit 1
run c1_acgt --mode acgt || exit 1
if [ "$2" = more ]; then
run q8 --queries 8 --refs 1000000 --pool 1000000 || exit 1
run q32 --queries 32 --refs 1000000 --pool 1000000 || exit 1
run c2 --queries 10000 --refs 100000 --mode acgt --pool 65536 --steps 3 --warmup 1 || exit 1
fi
