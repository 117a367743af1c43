set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2e; mkdir -p $O
B="python bench.py --steps 6 --warmup 1 --no-sweep --cpu-refs 0 --no-parity"
for a in 0 1 2 3; do UVAIA_GPU_SCAN_ABLATE=$a timeout -k 10 200 $B > $O/abl_$a.json 2> $O/abl_$a.err || exit 1; done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM -d $O/pmc_a -o a --output-format csv -- $B > $O/pmc_a.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT -d $O/pmc_b -o b --output-format csv -- $B > $O/pmc_b.log 2>&1 || exit 1
rocprofv3 -L > $O/counters.txt 2>&1
python tools/pmc_summary.py a=$(ls $O/pmc_a/*/*counter_collection.csv $O/pmc_a/*counter_collection.csv 2>/dev/null | head -1) b=$(ls $O/pmc_b/*/*counter_collection.csv $O/pmc_b/*counter_collection.csv 2>/dev/null | head -1) > $O/pmc_summary.json
python - <<'PY'
import json
for a in range(4):
    d=json.load(open("gpurun_out/r2e/abl_%d.json"%a)); print("ablate",a, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["step_parts"]["scan_and_replay_ms"])
s=json.load(open("gpurun_out/r2e/pmc_summary.json"))
for lab in s:
    for k in s[lab]:
        if "scan3" in k: print(lab, {c:(v["n"], round(v["mean"])) for c,v in s[lab][k].items()})
PY
