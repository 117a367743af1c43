#!/bin/bash
# bash tools/r04_smallq2.sh <tag> <query counts...>: packed-plane scan regime, parity on the timed path included
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; shift; O=gpurun_out/r04/$TAG; mkdir -p $O
for nq in "$@"; do
  timeout -k 10 240 python bench.py --queries $nq --refs 1000000 --pool 1000000 --steps 10 --warmup 2 --no-sweep --cpu-refs 0 --align-queries 0 > $O/q$nq.json 2> $O/q$nq.err || { echo "q$nq failed"; tail -5 $O/q$nq.err; exit 1; }
  python - $O/q$nq.json <<'P'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(b["config"].get("queries"), "ms/step", b["ms_per_step"], "whole-step frac of HBM", round(b["config"]["refs_per_gpu"] * 14976 / (b["ms_per_step"] * 1e-3) / 8e12, 3), "scan ms", b["roofline"].get("avg_launch_ms"), "frac", b["roofline"]["frac"], "parity", b.get("parity_check_on_timed_path"), flush=True)
P
done
