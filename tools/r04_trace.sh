#!/bin/bash
# kernel timeline of one configuration: bash tools/r04_trace.sh <tag> <bench.py flags...>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; shift; O=gpurun_out/r04/trace_$TAG; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o t --output-format csv -- python bench.py "$@" --no-sweep --cpu-refs 0 --no-parity --align-queries 0 --steps 4 --warmup 1 > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
python - $O <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ks = [r for r in rows if any(x in r["Kernel_Name"] for x in ("scan2", "scan3", "replay", "derive_all", "pair_extras", "tile_bounds"))]
t0 = int(ks[0]["Start_Timestamp"])
for r in ks[-40:]:
    print("%-28s start %9.1f us  dur %8.1f us  stream %s" % (r["Kernel_Name"][5:33], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Stream_Id"]))
P
