#!/usr/bin/env python3
"""End-to-end ingest comparison on the GPU box: the same synthetic references as (a) raw FASTA through the
text path of bin/uvaia, (b) packed once by bin/uvaiapack and searched with --packed.  Prints one JSON line.
Usage: python tools/ingest_bench.py [--refs 20000] [--queries 100] [--dir /tmp/ingest]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uvaia_amd import hostlib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--refs", type=int, default=20000)
    ap.add_argument("--queries", type=int, default=100)
    ap.add_argument("--dir", default="/tmp/ingest")
    ap.add_argument("--pool", type=int, default=8192)
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    gen = hostlib.Synth()
    ref_fa, q_fa, db = (os.path.join(a.dir, x) for x in ("refs.fa", "query.fa", "refs.uvdb"))
    with open(ref_fa, "wb") as fh:
        for first in range(0, a.refs, 2048):
            rows, _ = gen.generate(first, min(2048, a.refs - first))
            for i in range(rows.shape[0]):
                fh.write(b">ref_%d\n" % (first + i) + rows[i].tobytes() + b"\n")
    with open(q_fa, "wb") as fh:
        rows, _ = gen.generate(10_000_000, a.queries)
        for i in range(rows.shape[0]):
            fh.write(b">q_%d\n" % i + rows[i].tobytes() + b"\n")

    logs = {}

    def run(cmd, tag=None):
        t0 = time.perf_counter()
        r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        if r.returncode:
            sys.stderr.write("FAILED (%d): %s\n%s\n" % (r.returncode, " ".join(cmd), r.stderr.decode(errors="replace")[-3000:]))
            sys.exit(1)
        if tag:
            logs[tag] = [l for l in r.stderr.decode(errors="replace").splitlines() if "secs" in l]
        return time.perf_counter() - t0

    uv, pk = os.path.join(ROOT, "bin", "uvaia"), os.path.join(ROOT, "bin", "uvaiapack")
    common = [q_fa, "-p", str(a.pool), "-n", "100"]
    t_text = run([uv, "-r", ref_fa, "-o", os.path.join(a.dir, "out_text")] + common, "text")
    t_pack = run([pk, "-o", db, ref_fa], "pack")
    t_packed = run([uv, "--packed", db, "-o", os.path.join(a.dir, "out_packed")] + common)
    t_packed2 = run([uv, "--packed", db, "-o", os.path.join(a.dir, "out_packed")] + common, "packed")      # page cache warm
    same = all(open(os.path.join(a.dir, "out_text" + s), "rb").read() == open(os.path.join(a.dir, "out_packed" + s), "rb").read() for s in (".csv.xz",))
    print(json.dumps({"refs": a.refs, "queries": a.queries, "pool": a.pool, "fasta_bytes": os.path.getsize(ref_fa), "uvdb_bytes": os.path.getsize(db),
                      "uvaia_text_s": round(t_text, 2), "uvaiapack_s": round(t_pack, 2), "uvaia_packed_s": round(t_packed, 2), "uvaia_packed_warm_s": round(t_packed2, 2),
                      "refs_per_s_text": round(a.refs / t_text), "refs_per_s_packed": round(a.refs / t_packed2), "same_table": same,
                      "progress_lines": logs, "note": "wall clock of the whole command (query preparation, engine start-up, xz output included)"}))


if __name__ == "__main__":
    main()
