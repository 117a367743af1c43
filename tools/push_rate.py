#!/usr/bin/env python3
"""Rate of the reference-shaped boundary call (uvaia_gpu_push: one pool of raw sequences from host memory per call) on the benchmark
data: references per second including the hand-over of the characters, next to the resident rate bench.py reports.  One JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uvaia_amd import hostlib  # noqa: E402


def main():
    gen = hostlib.Synth(29903, seed=20241008, preset=0)
    out = {}
    for nq in (100, 1000):
        qs, _ = gen.generate_bytes(1 << 40, nq)
        pq = hostlib.PreparedQuery(qs, ["query_%d" % i for i in range(nq)])
        pool, n_pools = 32768, 3
        rows, non_n = gen.generate(0, pool)
        refs = [rows[i].tobytes() for i in range(pool)]
        with pq.open_engine(nbest=100, max_pool=pool) as eng:
            eng.push(refs, non_n=non_n, ordinal0=0)                     # warm-up
            eng.reset()
            t0 = time.perf_counter()
            for k in range(n_pools):
                eng.push(refs, non_n=non_n, ordinal0=k * pool)
            dt = time.perf_counter() - t0
        out["%d_queries" % nq] = {"refs_per_s": round(n_pools * pool / dt), "pool": pool, "pools": n_pools, "bytes_per_ref_over_pcie": 29903,
                                  "host_to_device_GBps": round(n_pools * pool * 29952 / dt / 1e9, 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
