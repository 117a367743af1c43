"""Round 4 probe (engine built with -DREPLAY_TIMING): where a replay wave's time goes at config[1] -- per admission the round trip for the
exact comparison's words, the comparison, the heap update; per group of tiles the wait for its counters; tolerance rises.
python tools/r04_probe_c1.py [queries] [refs]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvaia_amd import capi, hostlib
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_ref = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
gen = hostlib.Synth(29903, seed=20241008, preset=0)
qseqs, _ = gen.generate_bytes(1 << 40, nq)
pq = hostlib.PreparedQuery(qseqs, ["query_%d" % i for i in range(nq)])
eng = pq.open_engine(nbest=100, max_pool=65536, device=0)
eng.db_reserve(n_ref)
for a in range(0, n_ref, 8192):
    n = min(8192, n_ref - a)
    rows, non_n = gen.generate(a, n)
    eng.db_append_block(rows, non_n)
for rep in range(2):
    eng.reset(); eng.db_rederive()
    eng.replay_stats(reset=True); eng.replay_timing(reset=True)
    eng.search_resident(65536, ordinal0=0, want_entered=False)
    eng.sync()
adm, dem, dense = eng.replay_stats()
t = eng.replay_timing()
us = lambda x: x / 100.0
print(json.dumps({"queries": nq, "refs": n_ref, "admissions": adm, "exact_comparisons": dem,
                  "per_query_wave_us": {"whole": us(t[6]) / nq, "slowest_wave_of_any_slice": us(t[7]), "fetch_wait": us(t[0]) / nq, "comparison": us(t[1]) / nq, "heap_update": us(t[2]) / nq, "group_counter_wait": us(t[3]) / nq},
                  "per_admission_us": {"fetch_wait": us(t[0]) / max(adm, 1), "comparison": us(t[1]) / max(dem, 1), "heap_update": us(t[2]) / max(adm, 1)},
                  "tile_groups_per_query": t[5] / nq, "group_wait_us_each": us(t[3]) / max(t[5], 1), "tolerance_rises_per_query": t[4] / nq,
                  "note": "timing build: the waits are made explicit (vmcnt(0) before the use), which also serialises what the product overlaps"}), flush=True)
