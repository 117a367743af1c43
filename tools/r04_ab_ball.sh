#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04/abball; mkdir -p $O
SAVE=$(mktemp /tmp/libuvaia_gpu_default.XXXXXX.so) || exit 1
cp uvaia_amd/lib/libuvaia_gpu.so "$SAVE" || exit 1
restore() { cp "$SAVE" uvaia_amd/lib/libuvaia_gpu.so; rm -f "$SAVE"; }
trap restore EXIT
trap 'exit 130' INT TERM
for rep in 1 2; do
for v in head ballpf; do
  if [ $v = head ]; then cp "$SAVE" uvaia_amd/lib/libuvaia_gpu.so; else cp uvaia_amd/lib/variants/libuvaia_gpu_$v.so uvaia_amd/lib/libuvaia_gpu.so; fi
  timeout -k 10 200 python bench.py --ball-only --steps 5 > $O/${v}_$rep.json 2> $O/${v}_$rep.err || { echo "$v failed"; tail -3 $O/${v}_$rep.err; continue; }
  python - "$O/${v}_$rep.json" "$v" <<'P'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["ball"]
print(sys.argv[2], b["ms_per_search"], b["parity"], [(k["kernel"], k["ms"], k.get("frac")) for k in b["kernels"]], flush=True)
P
done
done
