#!/bin/bash
# Round 4: the handful-of-queries regime after the scan-side extras.  bash tools/r04_smallq.sh [tag]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-smallq}; O=gpurun_out/r04/$TAG; mkdir -p $O
for nq in 1 4 16 32; do
  timeout -k 10 240 python bench.py --queries $nq --refs 1000000 --pool 1000000 --steps 10 --warmup 2 --no-sweep --cpu-refs 0 --align-queries 0 > $O/q$nq.json 2> $O/q$nq.err || { echo "q$nq failed"; tail -5 $O/q$nq.err; exit 1; }
  python - $O/q$nq.json <<'P'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(b["config"].get("queries"), "ms/step", b["ms_per_step"], "value", b["value"], "scan ms", b["roofline"].get("avg_launch_ms"), "frac", b["roofline"]["frac"], "parity", b.get("parity_check_on_timed_path"), b.get("replay"), flush=True)
P
done
