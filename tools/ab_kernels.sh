#!/bin/bash
# per-kernel statistics of engine variants (uvaia_amd/lib/variants/) on the headline workload: bash tools/ab_kernels.sh [bench flags]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab_kernels; mkdir -p $O
SAVE=$(mktemp /tmp/libuvaia_gpu_default.XXXXXX.so) || exit 1
cp uvaia_amd/lib/libuvaia_gpu.so "$SAVE" || exit 1
# the engine is put back on every way out; a signal ends the script (it does not go on to the next variant)
restore() { cp "$SAVE" uvaia_amd/lib/libuvaia_gpu.so; rm -f "$SAVE" uvaia_amd/lib/variants/libuvaia_gpu_head.so; }
trap restore EXIT
trap 'exit 130' INT TERM
cp "$SAVE" uvaia_amd/lib/variants/libuvaia_gpu_head.so
for lib in uvaia_amd/lib/variants/libuvaia_gpu_*.so; do
  v=$(basename $lib .so); v=${v#libuvaia_gpu_}
  cp $lib uvaia_amd/lib/libuvaia_gpu.so || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/$v -o k --output-format csv -- python bench.py --steps 10 --warmup 2 --no-sweep --cpu-refs 0 --no-parity "$@" > $O/$v.json 2> $O/$v.err || { echo "variant $v failed"; tail -5 $O/$v.err; exit 1; }
  echo "== $v"; python - "$O/$v/k_kernel_stats.csv" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print("%-40s calls %5s avg %9.1f us  total %8.2f ms  max %9.1f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["MaxNs"]) / 1e3), flush=True)
P
done
rm -f uvaia_amd/lib/variants/libuvaia_gpu_head.so
echo done
