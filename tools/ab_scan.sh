#!/bin/bash
# Engine variants (uvaia_amd/lib/variants/libuvaia_gpu_<name>.so) on the headline workload, one box, interleaved twice:
#   bash tools/ab_scan.sh [extra bench.py flags]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab_scan; mkdir -p $O
SAVE=$(mktemp /tmp/libuvaia_gpu_default.XXXXXX.so) || exit 1
cp uvaia_amd/lib/libuvaia_gpu.so "$SAVE" || exit 1
# the engine is put back on every way out; a signal ends the script (it does not go on to the next variant)
restore() { cp "$SAVE" uvaia_amd/lib/libuvaia_gpu.so; rm -f "$SAVE" uvaia_amd/lib/variants/libuvaia_gpu_head.so; }
trap restore EXIT
trap 'exit 130' INT TERM
cp "$SAVE" uvaia_amd/lib/variants/libuvaia_gpu_head.so
for rep in 1 2; do
for lib in uvaia_amd/lib/variants/libuvaia_gpu_*.so; do
  v=$(basename $lib .so); v=${v#libuvaia_gpu_}
  cp $lib uvaia_amd/lib/libuvaia_gpu.so || exit 1
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-sweep --cpu-refs 0 --no-parity "$@" > $O/${v}_$rep.json 2> $O/${v}_$rep.err || { echo "variant $v failed"; tail -5 $O/${v}_$rep.err; exit 1; }
  python - "$O/${v}_$rep.json" "$v" <<'P'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", b["value"], "ms/step", b["ms_per_step"], "scan ms", b["roofline"]["avg_launch_ms"], "derive", b["step_parts"]["derived_planes_ms"], "search", b["step_parts"]["scan_and_replay_ms"], flush=True)
P
done
done
rm -f uvaia_amd/lib/variants/libuvaia_gpu_head.so
echo done
