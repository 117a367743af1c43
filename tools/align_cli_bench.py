#!/usr/bin/env python3
"""End-to-end timing of bin/uvaialign on the GPU box: the BASELINE config[4] workload of bench.py written as FASTA (reference + unaligned
queries), aligned to stdout and to an xz file.  Prints one JSON line.  Usage: python tools/align_cli_bench.py [--queries 10000] [--dir /tmp/align_cli]"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from uvaia_amd import hostlib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=10000)
    ap.add_argument("--dir", default="/tmp/align_cli")
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    gen = hostlib.Synth(29903, seed=20241008, preset=1)
    ref = np.array(gen.generate(7, 1)[0][0], dtype=np.uint8)
    ref[~np.isin(ref, np.frombuffer(b"ACGT", dtype=np.uint8))] = ord("A")
    gen = hostlib.Synth(29903, seed=20241008, preset=0)
    rng = np.random.default_rng(20241008)
    ref_fa, q_fa = os.path.join(a.dir, "ref.fa"), os.path.join(a.dir, "queries.fa")
    with open(ref_fa, "wb") as fh:
        fh.write(b">reference\n" + ref.tobytes() + b"\n")
    n = 0
    with open(q_fa, "wb") as fh:
        for first in range(0, a.queries, 2048):
            rows, _ = gen.generate(bench.QUERY_INDEX0 + first, min(2048, a.queries - first))
            for s in bench.unaligned_from_rows(np.asarray(rows, dtype=np.uint8), rng):
                fh.write(b">q_%d\n" % n + s + b"\n")
                n += 1
    exe = os.path.join(ROOT, "bin", "uvaialign")

    def run(extra, stdout):
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-r", ref_fa, q_fa, "-a", "1.0"] + extra, stdout=stdout, stderr=subprocess.PIPE)
        if r.returncode:
            sys.stderr.write(r.stderr.decode(errors="replace")[-2000:])
            sys.exit(1)
        return time.perf_counter() - t0, [l for l in r.stderr.decode(errors="replace").splitlines() if "secs" in l or "Output" in l][-2:]

    with open(os.path.join(a.dir, "out.fa"), "wb") as fh:
        t_plain, log_plain = run(["--stdout"], fh)
    t_xz, log_xz = run(["-o", os.path.join(a.dir, "out")], subprocess.DEVNULL)
    print(json.dumps({"queries": n, "query_fasta_bytes": os.path.getsize(q_fa), "uvaialign_stdout_s": round(t_plain, 2), "uvaialign_xz_s": round(t_xz, 2),
                      "queries_per_s_stdout": round(n / t_plain), "queries_per_s_xz": round(n / t_xz), "aligned_bytes": os.path.getsize(os.path.join(a.dir, "out.fa")),
                      "progress_lines": log_plain + log_xz, "note": "wall clock of the whole command: FASTA parsing, filters, GPU alignment in pools of 256 x host threads, output"}))


if __name__ == "__main__":
    main()
