#!/bin/bash
# time split of the replay waves at 16-128 queries (timing build): bash tools/r04_probe_mid.sh <query counts...>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04/probe_mid; mkdir -p $O
cp uvaia_amd/lib/libuvaia_gpu.so /tmp/libuvaia_gpu_product.so && cp uvaia_amd/lib/variants/libuvaia_gpu_timing.so uvaia_amd/lib/libuvaia_gpu.so || exit 1
trap 'cp /tmp/libuvaia_gpu_product.so uvaia_amd/lib/libuvaia_gpu.so' EXIT
for nq in "$@"; do
  timeout -k 10 250 python tools/r04_probe.py $nq 1000000 > $O/q$nq.txt 2>&1 || { tail -5 $O/q$nq.txt; exit 1; }
  tail -1 $O/q$nq.txt
done
