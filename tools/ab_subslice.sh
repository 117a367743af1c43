#!/bin/bash
# sweep of the sub-slice length of the resident search at config[1] on one box (bench.py --subslice)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab_subslice; mkdir -p $O
for n in 0 16704 20032 25024 50048 0; do
  timeout -k 10 300 python bench.py --subslice $n --steps 30 --warmup 5 --no-sweep --no-parity --cpu-refs 0 > $O/s${n}_$RANDOM.json 2> $O/err_$n.txt || exit 1
done
echo done
