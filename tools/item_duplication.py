#!/usr/bin/env python3
"""How many of scan3_kernel's per-query items of a super-tile are copies of one another.  CPU only.
An item is what a query that is not ACGT in part of a 32-column word costs a (query, tile) pair: {word, mask of its non-ACGT sites}.
Two queries of the same super-tile of 64 with the same word and mask could share one count and pay an LDS add each.  The benchmark's
generator places its N runs uniformly at random (uvaia_amd/csrc/host/synth.c: fill_n at below(L)), so identical masks are chance."""
import collections, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uvaia_amd import hostlib
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
gen = hostlib.Synth(29903, seed=20241008, preset=0)
qs, _ = gen.generate_bytes(1 << 40, nq)
pq = hostlib.PreparedQuery(qs, ["query_%d" % i for i in range(nq)])
rows = np.frombuffer(b"".join(pq.seqs), dtype=np.uint8).reshape(pq.ntax, -1)           # in the engine's order (by valid sites)
L = rows.shape[1]
acgt = np.isin(rows, np.frombuffer(b"ACGT", dtype=np.uint8))
pad = (-L) % 32
bad = np.pad(~acgt, ((0, 0), (0, pad)))                                             # non-ACGT sites
words = np.packbits(bad.reshape(pq.ntax, -1, 32), axis=2, bitorder="little").view(np.uint32)[:, :, 0]
partial = (words != 0) & (words != 0xFFFFFFFF)
items = distinct = 0
for st in range(0, pq.ntax, 64):
    w = words[st:st + 64]; p = partial[st:st + 64]
    for col in np.nonzero(p.any(axis=0))[0]:
        masks = w[p[:, col], col]
        items += len(masks); distinct += len(set(masks.tolist()))
print(json.dumps({"queries": pq.ntax, "partially dirty (query, word) items": int(items), "distinct {word, mask} per super-tile of 64": int(distinct),
                  "duplication": round(items / max(distinct, 1), 4), "items per query": round(items / pq.ntax, 1),
                  "note": "shared items would pay off from about 1.5; the generator's runs start at uniformly random columns"}))
