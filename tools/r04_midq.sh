#!/bin/bash
# Round 4, last: the replay with scan-side extras at 16-128 queries (staging depth by query count; the column-compressed scan followed by
# extras_bounds_kernel from 33).  bash tools/r04_midq.sh [tag] [query counts...]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-midq}; shift; O=gpurun_out/r04/$TAG; mkdir -p $O
QS=${@:-16 32 64 128}
for nq in $QS; do
  timeout -k 10 240 python bench.py --queries $nq --refs 1000000 --pool 1000000 --steps 10 --warmup 2 --no-sweep --cpu-refs 0 --align-queries 0 > $O/q$nq.json 2> $O/q$nq.err || { echo "q$nq failed"; tail -5 $O/q$nq.err; exit 1; }
  python - $O/q$nq.json <<'P'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(b["config"].get("queries"), "ms/step", b["ms_per_step"], "value", b["value"], "scan ms", b["roofline"].get("avg_launch_ms"), "frac", b["roofline"]["frac"], "whole", b["roofline"].get("whole_step_frac"), "parity", b.get("parity_check_on_timed_path"), b.get("replay"), flush=True)
P
done
