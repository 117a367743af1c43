#!/bin/bash
# A/B on one box: chunks of the rebuild of the derived planes on one stream or alternating over three (bench.py --rederive-streams)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab_rederive; mkdir -p $O
for rep in 1 2; do for n in 1 3 2; do
  timeout -k 10 300 python bench.py --rederive-streams $n --steps 30 --warmup 5 --no-sweep --no-parity --cpu-refs 0 > $O/s${n}_$rep.json 2> $O/s${n}_$rep.err || exit 1
done; done
echo done
