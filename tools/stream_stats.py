#!/usr/bin/env python3
"""What the item streams of scan3_kernel hold for the benchmark's query set (uvaia_gpu_export_query_table 7 and 10): records and
items by kind, per query and word group.  python tools/stream_stats.py [queries] [mode]"""
import json
import sys
import os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from uvaia_amd import hostlib

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
mode = sys.argv[2] if len(sys.argv) > 2 else "iupac"
gen = hostlib.Synth(29903, seed=20241008, preset=0)
qs, _ = gen.generate_bytes(bench.QUERY_INDEX0, nq)
pq = hostlib.PreparedQuery(qs, ["q%d" % i for i in range(nq)], acgt=(mode == "acgt"))
eng = pq.open_engine(nbest=100, max_pool=4096)
s = eng.query_table(7).view(np.uint32)
cnt = eng.query_table(10).view(np.int32)
eng.close()
p, n = 0, len(s) - 64
out = {"group_records": 0, "full_entries": 0, "general_items": 0, "word_items": 0, "rare_records": 0, "rare_items": 0, "dwords": int(n)}
while p < n:
    h = s[p:p + 4]
    if h[3] == 4 and h[0] == 0 and h[1] == 0 and h[2] == 0:
        p += 4
        continue
    if h[3] >= 8:
        out["group_records"] += 1
        out["full_entries"] += int(h[1] & 0xFFFF) * 4
        out["general_items"] += int(h[1] >> 16)
        nw = int(sum((h[2] >> (8 * j)) & 255 for j in range(4)))
        out["word_items"] += nw
        w0 = p + 4 + int(h[1] & 0xFFFF) * 4 + int(h[1] >> 16) * 12
        out["run_items"] = out.get("run_items", 0) + int(np.count_nonzero(s[w0 + 3:w0 + 4 * nw:4]))      # word items that also carry full words
        p += int(h[3])
        continue
    k = int(sum((h[1] >> (8 * j)) & 255 for j in range(4)))
    assert k > 0, (p, h)
    out["rare_records"] += 1
    out["rare_items"] += k
    p += 4 + 4 * k
names = ["polymorphic_dense", "rare", "NP4", "NR4", "rare_max", "scan", "need_e", "need_v", "need_g", "need_r", "replay_lq"]
out.update({k: int(v) for k, v in zip(names, cnt)})
out["queries"] = pq.ntax
out["per_query"] = {k: round(out[k] / pq.ntax, 2) for k in ("full_entries", "general_items", "word_items", "run_items", "rare_items")}
print(json.dumps(out))
