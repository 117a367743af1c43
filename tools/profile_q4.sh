set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/q4 -o q4 --output-format csv -- python bench.py --queries 4 --refs 1000000 --pool 1000000 --no-sweep --cpu-refs 0 --no-parity --steps 5 --warmup 1 > $O/q4.log 2>&1 || exit 1
python - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/r2j/q4/**/*kernel_stats.csv", recursive=True) + glob.glob("gpurun_out/r2j/q4/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
    break
PY
tail -2 $O/q4.log | cut -c1-600
