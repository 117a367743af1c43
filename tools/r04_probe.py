"""Round 4 probe: tolerances and the (query, tile) pairs the replay opens, a handful of queries over prefixes of the benchmark stream.
python tools/r04_probe.py <queries> <refs>"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from uvaia_amd import capi, hostlib
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4
total = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
gen = hostlib.Synth(29903, seed=20241008, preset=0)
qseqs, _ = gen.generate_bytes(1 << 40, nq)
pq = hostlib.PreparedQuery(qseqs, ["query_%d" % i for i in range(nq)])
for n_ref in (65536, total):
    eng = pq.open_engine(nbest=100, max_pool=n_ref, device=0)
    eng.db_reserve(n_ref)
    for a in range(0, n_ref, 8192):
        n = min(8192, n_ref - a)
        rows, non_n = gen.generate(a, n)
        eng.db_append_block(rows, non_n)
    eng.reset()
    eng.replay_stats(reset=True); eng.replay_tiles_opened(reset=True); eng.replay_timing(reset=True)
    eng.search_resident(n_ref, ordinal0=0, want_entered=False)
    eng.sync()
    n, T, sc, od = eng.drain()
    print(json.dumps({"queries": nq, "refs": n_ref, "tiles": (n_ref + 63) // 64, "T": [int(t) for t in T], "admitted_demanded_dense": eng.replay_stats(), "tiles_opened": eng.replay_tiles_opened(),
                      "timing_us_sum_over_waves": dict(zip(["wait_staged", "request", "in_tiles", "in_admissions", "late_fetch", "n_late", "whole_wave", "prologue", "n_waves_x100"], [x / 100.0 for x in eng.replay_timing()[:9]])),
                      "worst_first_key": [int(sc[q][1][0]) if n[q] else None for q in range(nq)], "n_idx_c": int(pq.n_idx_c) if hasattr(pq, "n_idx_c") else None}), flush=True)
    eng.close()
