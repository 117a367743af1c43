#!/usr/bin/env python3
"""How much of the pair counters a scan-side filter could skip.  CPU only: the oracle's all-pairs
scores of a sample of the benchmark's queries against its first references, the reference's state machine replayed in Python, and at
every slice start the bounds a scan running AHEAD of the ordered machine may use:

  monotone    tolerance <= T_ub = 1 + largest mismatch count held; first key >= W0 = first key of the worst kept entry
  j-bound     after at most j admissions the tolerance is at most 1 + the largest mismatch count among the j + 1 lowest-key entries
              (an old entry becomes the root only once everything below it is gone); evaluated with j = the slice's TRUE number of
              admissions (an oracle: no implementable filter can do better) and with fixed caps (hand-over to the exact path after j)
  live        what the replay itself opens: the tolerance and worst key at the moment the tile is reached

Reported per slice: the fraction of (query, tile of 64 references) pairs that hold at least one reference passing the filter, i.e. the
fraction of the scan's counters that would still have to be written.    python tools/admission_bound.py [n_queries_sample] [n_refs]"""
import heapq
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from uvaia_amd import hostlib

NQ_ALL, K = 1000, 100
n_sample = int(sys.argv[1]) if len(sys.argv) > 1 else 48
n_ref = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
slice_len = (n_ref + 2) // 3
gen = hostlib.Synth(29903, seed=20241008, preset=0)
qall, _ = gen.generate_bytes(1 << 40, NQ_ALL)
qs = [qall[i] for i in range(0, NQ_ALL, NQ_ALL // n_sample)][:n_sample]
q = O.Query(qs, ["q%d" % i for i in range(len(qs))])
S = np.zeros((n_ref, q.ntax, 6), np.int32)
for a in range(0, n_ref, 4096):
    refs, _ = gen.generate_bytes(a, min(4096, n_ref - a))
    S[a:a + len(refs)] = q.allpairs(refs)
M = S[:, :, 3] - S[:, :, 0]                      # mismatches; first key = S[:, :, 0]
n_tiles = (n_ref + 63) // 64
caps = [0, 4, 16, 64]
acc = {}


def tiles_with(flags, lo, hi):
    t = np.zeros(n_tiles, bool)
    idx = np.nonzero(flags)[0] + lo
    t[idx // 64] = True
    return int(t[lo // 64:(hi + 63) // 64].sum()), (hi + 63) // 64 - lo // 64


for iq in range(q.ntax):
    m, k0 = M[:, iq], S[:, iq, 0]
    keys = [tuple(int(x) for x in S[r, iq]) for r in range(0)]  # built lazily below
    heap = []                                   # (key tuple, mismatches): root = worst kept
    T = 29903
    live = np.zeros(n_ref, bool)                # passes the replay's own conservative test when it is reached
    starts = {}
    n_adm = np.zeros(3, int)
    for r in range(n_ref):
        if r % slice_len == 0 and r:
            ent = sorted(heap)
            ms = np.array([e[1] for e in ent])
            starts[r // slice_len] = (np.maximum.accumulate(ms) + 1, int(ms.max()) + 1, ent[0][0][0], T)
        mr = int(m[r])
        if mr >= T:
            continue
        full = len(heap) == K
        if full and int(k0[r]) < heap[0][0][0]:
            continue
        live[r] = True
        key = tuple(int(x) for x in S[r, iq])
        if full and not key > heap[0][0]:
            continue
        if full:
            heapq.heapreplace(heap, (key, mr))
        else:
            heapq.heappush(heap, (key, mr))
        n_adm[min(r // slice_len, 2)] += 1
        if len(heap) == K:
            T = heap[0][1] + 1
    for sl in (1, 2):
        lo, hi = sl * slice_len, min(n_ref, (sl + 1) * slice_len)
        tb, tub, w0, t_start = starts[sl]
        mm, kk = m[lo:hi], k0[lo:hi]
        first = kk >= w0
        filt = {"monotone (T_ub, W0)": (mm < tub) & first, "tolerance at slice start, W0 (NOT exact: T can rise)": (mm < t_start) & first,
                "j-bound, j = true admissions of the slice": (mm < tb[min(int(n_adm[sl]), K - 1)]) & first, "live (what the replay opens)": live[lo:hi]}
        for c in caps:
            filt["j-bound, j = %d" % c] = (mm < tb[min(c, K - 1)]) & first
        for name, f in filt.items():
            kept, tot = tiles_with(f, lo, hi)
            a = acc.setdefault((sl, name), [0, 0, 0, 0])
            a[0] += kept; a[1] += tot; a[2] += int(f.sum()); a[3] += hi - lo
        acc.setdefault((sl, "admissions"), [0, 0, 0, 0])[0] += int(n_adm[sl])
out = {"workload": "%d of the benchmark's 1000 queries x its first %d references, k = %d, three equal slices (slice 0 starts with empty heaps: every tile is needed)" % (q.ntax, n_ref, K), "slices": {}}
for (sl, name), a in sorted(acc.items()):
    d = out["slices"].setdefault("slice %d" % sl, {})
    if name == "admissions":
        d["admissions per query"] = a[0] / q.ntax
    else:
        d[name] = {"tiles kept": round(a[0] / a[1], 4), "pairs kept": round(a[2] / a[3], 5)}
print(json.dumps(out, indent=1))
