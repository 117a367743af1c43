"""Experiment (GPU box): sharpness of the bound T <= 1 + max mismatches among the (A+1) lexicographically worst kept
entries (valid until A admissions have happened), and the number of admissions per query and step."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uvaia_amd import hostlib

Q, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
preset = int(sys.argv[4]) if len(sys.argv) > 4 else 0
gen = hostlib.Synth(29903, seed=20241008, preset=preset)
qs, _ = gen.generate_bytes(1 << 40, Q)
pq = hostlib.PreparedQuery(qs, ["q%d" % i for i in range(Q)])
steps = [256, 1024, 4096, 16384]
while sum(steps) < N: steps.append(min(65536, N - sum(steps)))
eng = pq.open_engine(nbest=K, max_pool=max(steps))
pos = 0
prev_ord = None
for st in steps:
    st = min(st, N - pos)
    if st <= 0: break
    n, T, sc, od = eng.drain()
    full = n == eng.slots
    bounds = {}
    if full.all():
        keys = [tuple(sc[q, s]) for q in range(Q) for s in range(1, eng.slots + 1)]
        for A in (0, 4, 16, 64):
            tb = np.zeros(Q, dtype=np.int64)
            for q in range(Q):
                ent = sorted((tuple(sc[q, s]) for s in range(1, eng.slots + 1)))[: A + 1]     # lex-worst first
                tb[q] = 1 + max(e[3] - e[0] for e in ent)
            bounds[A] = tb
    W0 = np.where(full, sc[:, 1, 0], -1)
    rows, non_n = gen.generate(pos, st)
    eng.push([rows[i].tobytes() for i in range(st)], non_n=non_n)
    S = eng.last_batch_scores(st)
    m = S[:, :, 3] - S[:, :, 0]
    n2, T2, sc2, od2 = eng.drain()
    newin = np.array([len(set(od2[q, 1:n2[q] + 1]) - set(od[q, 1:n[q] + 1])) for q in range(Q)])
    line = "seen %7d step %6d T(mean %.1f max %d) survivors-of-step/query mean %.1f max %d |" % (pos, st, T.mean(), T.max(), newin.mean(), newin.max())
    for A, tb in bounds.items():
        c = (m < tb[None, :]) & (S[:, :, 0] >= W0[None, :])
        line += " A=%d: Tub %.1f pass %.5f |" % (A, tb.mean(), c.mean())
    print(line, flush=True)
    pos += st
