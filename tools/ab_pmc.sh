#!/bin/bash
# SQ counters of engine variants (uvaia_amd/lib/variants/) on the headline workload: bash tools/ab_pmc.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab_pmc; mkdir -p $O
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
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $O/$v -o k --output-format csv -- python bench.py --steps 5 --warmup 1 --no-sweep --cpu-refs 0 --no-parity > $O/$v.json 2> $O/$v.err || { echo "variant $v failed"; tail -5 $O/$v.err; exit 1; }
  echo "== $v"; python tools/pmc_summary.py x=$O/$v/k_counter_collection.csv > $O/$v.pmc.json; python - "$O/$v.pmc.json" <<'P'
import json, sys
s = json.load(open(sys.argv[1]))["x"]
for k, v in s.items():
    if "scan3" in k or "replay2" in k:
        print(k[:28], {c.replace("SQ_", ""): round(x["mean"] / 1e6, 3) for c, x in v.items()}, flush=True)
P
done
rm -f uvaia_amd/lib/variants/libuvaia_gpu_head.so
echo done
