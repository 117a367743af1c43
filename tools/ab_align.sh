#!/bin/bash
# Aligner shapes on one box: bash tools/ab_align.sh N_QUERIES NW:GROUP [NW:GROUP ...] builds uvaia_align.hip with -DWFA_NW / -DWFA_GROUP
# (waves per block, groups of 64 diagonals a wave has in flight) and runs the config[4] workload on every build; results in
# gpurun_out/ab_align/.  The library in the tree is put back at the end.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab_align; mkdir -p $O
N=${1:-2000}; shift
C=uvaia_amd/csrc; FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value"
[ -f $C/build/uvaia_gpu.o ] || make -C $C -s || exit 1
SAVE=$(mktemp /tmp/libuvaia_gpu_default.XXXXXX.so) || exit 1
cp uvaia_amd/lib/libuvaia_gpu.so "$SAVE" || exit 1
# whatever ends the script (a failed variant, a timeout from outside, a signal), the tree gets its default library back
# the engine is put back on every way out; a signal ends the script (it does not go on to the next variant)
restore() { cp "$SAVE" uvaia_amd/lib/libuvaia_gpu.so; rm -f "$SAVE" uvaia_amd/lib/variants/libuvaia_gpu_head.so; }
trap restore EXIT
trap 'exit 130' INT TERM
for v in "$@"; do
  nw=${v%%:*}; g=${v##*:}
  hipcc $FL -DWFA_NW=$nw -DWFA_GROUP=$g -c $C/uvaia_align.hip -o /tmp/v_align.o 2> /tmp/v_align.err && hipcc $FL -shared -o uvaia_amd/lib/libuvaia_gpu.so $C/build/uvaia_gpu.o /tmp/v_align.o || { cat /tmp/v_align.err; exit 1; }
  timeout -k 10 300 python bench.py --align-only --align-queries $N --align-cpu-queries 0 --steps 2 > $O/nw${nw}_g$g.json 2> $O/nw${nw}_g$g.err || { echo "variant $v failed"; tail -5 $O/nw${nw}_g$g.err; cp "$SAVE" uvaia_amd/lib/libuvaia_gpu.so; exit 1; }
  python - "$O/nw${nw}_g$g.json" "nw=$nw group=$g" <<'P'
import json, sys
a = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["align"]
print(sys.argv[2], a["value"], a["unit"], "kernel_ms", a["kernel_ms_per_pool"], "passes", a["passes"], flush=True)
P
done
cp "$SAVE" uvaia_amd/lib/libuvaia_gpu.so
echo done
