#!/usr/bin/env python3
"""Where does the exposed replay of the last slice spend its time?  Scans the three slices of config[1], replays the first two,
then times the replay of the last slice alone for ranges of the (valid-site-sorted) queries.  GPU box: python tools/replay_tail_probe.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvaia_amd import hostlib  # noqa: E402

NQ, NREF = 1000, 100000
gen = hostlib.Synth(29903, seed=20241008, preset=0)
qseqs, _ = gen.generate_bytes(1 << 40, NQ)
pq = hostlib.PreparedQuery(qseqs, ["q%d" % i for i in range(NQ)])
eng = pq.open_engine(nbest=100, max_pool=65536, device=0)
eng.db_reserve(NREF)
for a in range(0, NREF, 8192):
    rows, non_n = gen.generate(a, min(8192, NREF - a))
    eng.db_append_block(rows, non_n)
cuts = [0, 33344, 66688, NREF]
eng.reset()
for b in range(3):
    eng.slice_scan(cuts[b], cuts[b + 1] - cuts[b], b)
eng.slice_replay(0, 0, True)
eng.slice_replay(1, cuts[1], False)
eng.sync()
state = np.zeros(eng.state_bytes(), dtype=np.uint8)
eng.state_export(state.ctypes.data)
eng.sync()


def timed(q0, q1, reps=5):
    best = 1e9
    for _ in range(reps):
        eng.state_import(state.ctypes.data)
        eng.sync()
        t0 = time.perf_counter()
        eng.slice_replay_range(2, cuts[2], q0, q1, False)
        eng.sync()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


print("all queries            %.3f ms" % timed(0, NQ))
for q0 in range(0, NQ, 125):
    q1 = min(NQ, q0 + 125)
    if q0 % 16:
        q0 -= q0 % 16
    print("queries %4d..%4d     %.3f ms" % (q0, q1, timed(q0, q1)))
for q0, q1 in ((0, 16), (0, 1), (496, 512), (992, 1000)):
    print("queries %4d..%4d     %.3f ms" % (q0, q1, timed(q0, q1)))
eng.close()
