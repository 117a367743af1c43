#!/bin/bash
# prebuilt library variants (uvaia_amd/lib/variants/libuvaia_gpu_<name>.so) on the config[4] aligner workload, one box: bash tools/ab_libs.sh N_QUERIES
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab_libs; mkdir -p $O
SAVE=$(mktemp /tmp/libuvaia_gpu_default.XXXXXX.so) || exit 1
cp uvaia_amd/lib/libuvaia_gpu.so "$SAVE" || exit 1
# whatever ends the script (a failed variant, a timeout from outside, a signal), the tree gets its default library back
# the engine is put back on every way out; a signal ends the script (it does not go on to the next variant)
restore() { cp "$SAVE" uvaia_amd/lib/libuvaia_gpu.so; rm -f "$SAVE" uvaia_amd/lib/variants/libuvaia_gpu_head.so; }
trap restore EXIT
trap 'exit 130' INT TERM
for rep in 1 2; do
for lib in uvaia_amd/lib/variants/libuvaia_gpu_*.so; do
  v=$(basename $lib .so); v=${v#libuvaia_gpu_}
  cp $lib uvaia_amd/lib/libuvaia_gpu.so || exit 1
  timeout -k 10 300 python bench.py --align-only --align-queries ${1:-2000} --align-cpu-queries 0 --steps 2 > $O/${v}_$rep.json 2> $O/${v}_$rep.err || { echo "variant $v failed"; tail -5 $O/${v}_$rep.err; cp "$SAVE" uvaia_amd/lib/libuvaia_gpu.so; exit 1; }
  python - "$O/${v}_$rep.json" "$v" <<'P'
import json, sys
a = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["align"]
print(sys.argv[2], a["value"], a["unit"], "kernel_ms", a["kernel_ms_per_pool"], "passes", a["passes"], flush=True)
P
done
done
cp "$SAVE" uvaia_amd/lib/libuvaia_gpu.so
echo done
