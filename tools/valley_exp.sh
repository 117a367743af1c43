cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
run() { python bench.py --refs 1000000 --steps 5 --warmup 1 --no-sweep --cpu-refs 0 --no-parity "$@" > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; exit 1; }
  python - "$*" <<'P'
import json, sys
b = json.loads(open("/tmp/o.json").read().strip().splitlines()[-1])
print(sys.argv[1], "| ms/step", b["ms_per_step"], "scan ms", b["roofline"]["avg_launch_ms"], "launches", b["roofline"]["launches"], "derive", b["step_parts"]["derived_planes_ms"], "search", b["step_parts"]["scan_and_replay_ms"], flush=True)
P
}
for rs in 3 1 2; do
run --queries 64 --pool 1000000 --rederive-streams $rs
run --queries 256 --pool 65536 --rederive-streams $rs
run --queries 10000 --mode acgt --pool 65536 --steps 2 --rederive-streams $rs
done
