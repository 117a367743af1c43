#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04/first; mkdir -p $O
run() { name=$1; shift
  timeout -k 10 240 python bench.py --steps 10 --warmup 3 --no-sweep --cpu-refs 0 --no-parity --align-queries 0 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; return 1; }
  python - $O/$name.json "$name" <<'P'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-30s ms/step %7.3f  scan launch ms %7.4f x %d  derive %.3f search %.3f" % (sys.argv[2], b["ms_per_step"], b["roofline"].get("avg_launch_ms", 0), b["roofline"].get("launches", 0), b["step_parts"]["derived_planes_ms"], b["step_parts"]["scan_and_replay_ms"]), flush=True)
P
}
for rep in 1 2; do run equal_$rep; for p in 50 65 80; do run first${p}_$rep --tuning scan_streams=$((100 + p)); done; done
