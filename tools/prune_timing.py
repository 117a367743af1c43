#!/usr/bin/env python3
"""Times the redundancy pruning of the query preparation (exclude_redundant_query_sequences) on the host loop and with the pair
test on the device, on benchmark-shaped queries.  Usage (GPU box): python tools/prune_timing.py [n_queries ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uvaia_amd import hostlib  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1000, 3000, 10000]
gen = hostlib.Synth(29903, seed=20241008, preset=0)
for nq in sizes:
    qseqs, _ = gen.generate_bytes(1 << 40, nq)
    names = ["q%d" % i for i in range(nq)]
    t0 = time.time()
    base = hostlib.PreparedQuery(qseqs, names)
    t_plain = time.time() - t0
    row = {"queries": nq, "prepare_without_pruning_s": round(t_plain, 2)}
    for mode in ("host", "device"):           # the column walk of create_query_indices alone decides these two
        hostlib.set_prune_mode(mode)
        t0 = time.time()
        hostlib.PreparedQuery(qseqs, names)
        row["prepare_without_pruning_columns_on_%s_s" % mode] = round(time.time() - t0, 3)
    hostlib.set_prune_mode("auto")
    for mode in ("device", "host"):
        if mode == "host" and nq > 3000:
            continue                      # minutes
        hostlib.set_prune_mode(mode)
        t0 = time.time()
        pq = hostlib.PreparedQuery(qseqs, names, keep_resolved=True)
        row["prepare_with_pruning_%s_s" % mode] = round(time.time() - t0, 2)
        row["kept_%s" % mode] = pq.ntax
    print(row, flush=True)
