#!/usr/bin/env python3
"""Writes tests/golden/config1_oracle_snapshot.json: the table the oracle (oracle/uvaia_oracle.c) produces for BASELINE config[0]
shaped runs on the bundled alignment -- the first 10 names of 04.sample_1_1k.names as queries against the whole 9 185-sequence
database, --nbest 5, pools 64 and 9185, default and --acgt, trim 0 and 230.

This is a regression snapshot of the ORACLE (the reference itself cannot be built here: DESIGN.md section 6), committed so that a
change of the oracle, of the host preparation or of the engine shows up against fixed numbers and so that the GPU parity test has
expected rows on a box where only the repository exists.  Usage: python tools/make_golden.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures as F        # noqa: E402
import oracle_lib as O      # noqa: E402


def main():
    names, seqs = F.load_bundled()
    by = dict(zip(names, seqs))
    qn = F.sample_names_1k()[:10]
    out = {"what": "oracle output, config[0]-shaped runs on tests/golden/03.unique_acgt.aln.xz; made by tools/make_golden.py", "runs": []}
    for acgt in (False, True):
        for trim in (0, 230):
            for pool in (64, 9185):
                q = O.Query([by[n] for n in qn], qn, acgt=acgt, trim=trim)
                g = O.search(q, seqs, names, pool=pool, nbest=5)
                out["runs"].append({"acgt": acgt, "trim": trim, "pool": pool, "nbest": 5, "queries": list(q.names),
                                    "n_lowqual": g.n_lowqual, "final_T": list(g.final_T), "n_saved": len(g.saved),
                                    "rows": [[[name] + list(score) for _, name, score in rows] for rows in g.rows]})
    path = os.path.join(ROOT, "tests", "golden", "config1_oracle_snapshot.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
