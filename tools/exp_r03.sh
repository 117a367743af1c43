#!/bin/bash
# Round-3 experiments on one box; results in gpurun_out/exp/.  Usage: bash tools/exp_r03.sh [parts]
set -o pipefail
PART=${1:-ab}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/exp; mkdir -p $O
C1="python bench.py --steps 10 --warmup 2 --no-sweep --cpu-refs 0"
show() { python - "$1" <<'P'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = b["roofline"]
print(sys.argv[1], "value", b["value"], "ms/step", b["ms_per_step"], "scan ms", r["avg_launch_ms"], "parts", b["step_parts"]["derived_planes_ms"], b["step_parts"]["scan_and_replay_ms"], "parity", b["parity_check_on_timed_path"], flush=True)
P
}
if [[ $PART == *a* ]]; then
timeout -k 10 300 $C1 > $O/c1_r2.json 2> $O/c1_r2.err || { tail -5 $O/c1_r2.err; exit 1; }; show $O/c1_r2.json
timeout -k 10 300 $C1 --tuning scan_tiles_per_wave=4 > $O/c1_r4.json 2> $O/c1_r4.err || { tail -5 $O/c1_r4.err; exit 1; }; show $O/c1_r4.json
timeout -k 10 300 $C1 --mode acgt > $O/c1_acgt_r2.json 2> $O/c1_acgt_r2.err || { tail -5 $O/c1_acgt_r2.err; exit 1; }; show $O/c1_acgt_r2.json
fi
if [[ $PART == *b* ]]; then
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/ball -o ball --output-format csv -- python bench.py --ball-only --steps 3 > $O/ball.json 2> $O/ball.err || { tail -5 $O/ball.err; exit 1; }
tail -1 $O/ball.json | cut -c1-600
head -12 $O/ball/ball_kernel_stats.csv
fi
if [[ $PART == *c* ]]; then
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python bench.py --steps 5 --warmup 1 --no-sweep --cpu-refs 0 --no-parity > $O/pmc_write.log 2>&1 || { tail -5 $O/pmc_write.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python bench.py --steps 5 --warmup 1 --no-sweep --cpu-refs 0 --no-parity > $O/pmc_fetch.log 2>&1 || { tail -5 $O/pmc_fetch.log; exit 1; }
python tools/pmc_summary.py w=$(ls $O/pmc_write/*counter_collection.csv) f=$(ls $O/pmc_fetch/*counter_collection.csv) > $O/pmc_summary.json
python - <<'P'
import json
s = json.load(open("gpurun_out/exp/pmc_summary.json"))
for lab in s:
    for k, v in s[lab].items():
        if "scan3" in k or "replay2" in k or "derive" in k:
            print(lab, k[:40], {c: round(x["mean"], 1) for c, x in v.items()}, flush=True)
P
fi
if [[ $PART == *d* ]]; then
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
python - <<'P'
import json
b = json.loads(open("gpurun_out/exp/bench_default.json").read().strip().splitlines()[-1])
print("headline", b["value"], b["ms_per_step"], b["parity_check_on_timed_path"], b["cpu_baseline"])
for e in b["sweep"]: print(e["workload"][:60], e["value"], e["ms_per_step"], e["whole_step_frac_of_hbm_peak"], e["roofline"]["frac"], e.get("parity"))
print({k: v for k, v in b["ball"].items() if k != "workload"})
print({k: v for k, v in b["align"].items() if k not in ("note", "workload")})
P
fi

if [[ $PART == *e* ]]; then
for nq in 1 4 16 32; do for ph in 1; do
  timeout -k 10 300 python bench.py --queries $nq --refs 1000000 --pool 1000000 --steps 5 --warmup 1 --no-sweep --cpu-refs 0 --parity-refs 4096  > $O/q${nq}_p$ph.json 2> $O/q${nq}_p$ph.err || { tail -5 $O/q${nq}_p$ph.err; exit 1; }
  echo -n "queries $nq phases $ph: "; show $O/q${nq}_p$ph.json
done; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/q4trace -o q4 --output-format csv -- python bench.py --queries 4 --refs 1000000 --pool 1000000 --no-sweep --cpu-refs 0 --no-parity --steps 3 --warmup 1 > $O/q4trace.log 2>&1 || { tail -5 $O/q4trace.log; exit 1; }
python - <<'P'
import csv
rows = list(csv.DictReader(open("gpurun_out/exp/q4trace/q4_kernel_trace.csv")))
ks = [(r["Kernel_Name"].split("(")[0][:44], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
t0 = None
for k, s, e in ks[-22:]:
    if t0 is None: t0 = s
    print("%-46s start %8.1f end %8.1f dur %7.1f us" % (k, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3), flush=True)
P
fi

if [[ $PART == *f* ]]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/c1trace -o c1 --output-format csv -- python bench.py --no-sweep --cpu-refs 0 --no-parity --steps 4 --warmup 1 > $O/c1trace.log 2>&1 || { tail -5 $O/c1trace.log; exit 1; }
python - <<'P'
import csv
rows = list(csv.DictReader(open("gpurun_out/exp/c1trace/c1_kernel_trace.csv")))
ks = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]) for r in rows])
# the last complete step = from the last init_state_kernel but one... print the last 20 kernels
t0 = None
for s, e, k in ks[-19:]:
    if t0 is None: t0 = s
    print("%-42s start %8.1f end %8.1f dur %7.1f us" % (k, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3), flush=True)
P
fi
echo done
