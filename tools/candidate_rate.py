"""Experiment (GPU box): how many (query, ref) pairs pass a conservative filter that only knows r0 (ACGT matches) and
r3 (valid pairs), as a function of how much of the database has been seen.  Decides the two-phase scan design."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uvaia_amd import hostlib, capi

Q, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
preset = int(sys.argv[4]) if len(sys.argv) > 4 else 0
gen = hostlib.Synth(29903, seed=20241008, preset=preset)
qs, _ = gen.generate_bytes(1 << 40, Q)
pq = hostlib.PreparedQuery(qs, ["q%d" % i for i in range(Q)])
steps = [256, 1024, 4096, 16384]
while sum(steps) < N: steps.append(min(65536, N - sum(steps)))
eng = pq.open_engine(nbest=K, max_pool=max(steps))
pos = 0
for st in steps:
    st = min(st, N - pos)
    if st <= 0: break
    n, T, sc, od = eng.drain()
    full = n == eng.slots
    # bounds valid for the whole coming step
    m_heap = np.where(np.arange(eng.slots + 1)[None, :] <= n[:, None], sc[:, :, 3] - sc[:, :, 0], -1)
    m_heap[:, 0] = -1
    Tub = np.where(full, m_heap.max(axis=1) + 1, 1 << 30)
    W0 = np.where(full, sc[:, 1, 0], -1)
    rows, non_n = gen.generate(pos, st)
    refs = [rows[i].tobytes() for i in range(st)]
    eng.push(refs, non_n=non_n)
    S = eng.last_batch_scores(st)            # [st, Q, 6]
    m = S[:, :, 3] - S[:, :, 0]
    c_m = (m < Tub[None, :])
    c_ms = c_m & (S[:, :, 0] >= W0[None, :])
    lex = np.zeros_like(c_m)
    print("seen %7d  step %6d  heaps full %4d/%d  pass(m<Tub) %.4f  pass(m<Tub & S0>=W0) %.5f  refs with any candidate %.4f  max cand/query %d"
          % (pos, st, full.sum(), Q, c_m.mean(), c_ms.mean(), c_ms.any(axis=1).mean(), c_ms.sum(axis=0).max()), flush=True)
    pos += st
