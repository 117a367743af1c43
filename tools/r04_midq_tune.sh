#!/bin/bash
# bash tools/r04_midq_tune.sh <tag> <queries> <tuning strings, '+'-joined KEY=VALUE or 'none'>...
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; NQ=$2; shift; shift; O=gpurun_out/r04/$TAG; mkdir -p $O
for tn in "$@"; do
  FL=""; if [ "$tn" != none ]; then for kv in ${tn//+/ }; do FL="$FL --tuning $kv"; done; fi
  timeout -k 10 240 python bench.py --queries $NQ --refs 1000000 --pool 1000000 --steps 10 --warmup 2 --no-sweep --cpu-refs 0 --align-queries 0 $FL > $O/q${NQ}_$tn.json 2> $O/q${NQ}_$tn.err || { echo "$tn failed"; tail -5 $O/q${NQ}_$tn.err; exit 1; }
  python - $O/q${NQ}_$tn.json "$tn" <<'P'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(b["config"].get("queries"), sys.argv[2], "ms/step", b["ms_per_step"], "scan+replay", b["step_parts"]["scan_and_replay_ms"], "scan ms", b["roofline"].get("avg_launch_ms"), "parity", b.get("parity_check_on_timed_path"), flush=True)
P
done
