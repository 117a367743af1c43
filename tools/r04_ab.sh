#!/bin/bash
# engine variants under uvaia_amd/lib/variants on the headline workload, interleaved twice: bash tools/r04_ab.sh [bench flags]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04/ab; mkdir -p $O
SAVE=$(mktemp /tmp/libuvaia_gpu_default.XXXXXX.so) || exit 1
cp uvaia_amd/lib/libuvaia_gpu.so "$SAVE" || exit 1
restore() { cp "$SAVE" uvaia_amd/lib/libuvaia_gpu.so; rm -f "$SAVE" uvaia_amd/lib/variants/libuvaia_gpu_head.so; }
trap restore EXIT
trap 'exit 130' INT TERM
cp "$SAVE" uvaia_amd/lib/variants/libuvaia_gpu_head.so
for rep in 1 2; do
for lib in uvaia_amd/lib/variants/libuvaia_gpu_*.so; do
  v=$(basename $lib .so); v=${v#libuvaia_gpu_}
  [ "$v" = timing ] && continue
  cp $lib uvaia_amd/lib/libuvaia_gpu.so || exit 1
  timeout -k 10 240 python bench.py --steps 10 --warmup 3 --no-sweep --cpu-refs 0 --no-parity --align-queries 0 "$@" > $O/${v}_$rep.json 2> $O/${v}_$rep.err || { echo "variant $v failed"; tail -3 $O/${v}_$rep.err; continue; }
  python - "$O/${v}_$rep.json" "$v" <<'P'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-16s ms/step %7.3f  scan launch ms %7.4f x %d  derive %.3f search %.3f" % (sys.argv[2], b["ms_per_step"], b["roofline"].get("avg_launch_ms", 0), b["roofline"].get("launches", 0), b["step_parts"]["derived_planes_ms"], b["step_parts"]["scan_and_replay_ms"]), flush=True)
P
done
done
