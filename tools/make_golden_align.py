#!/usr/bin/env python3
"""Writes tests/golden/uvaialign_oracle_snapshot.json: what the oracle (oracle/wfa_oracle.c) produces for `uvaialign` on sequences of the
reference's bundled alignment -- the sequence with the fewest non-ACGT characters among the first 2 000 as the reference, 24 others as
the unaligned queries, all with their gap characters removed: alignment score, computed cells and a SHA-1 of the aligned row of each.

A regression snapshot of the ORACLE (the WFA library is an absent submodule and the reference holds no output of uvaialign: parity
unpinned), committed so that a change of the oracle or of the GPU aligner shows up against fixed numbers and so
that the command-line test has expected rows without the oracle in the loop.  Usage: python tools/make_golden_align.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures as F        # noqa: E402
import oracle_lib as O      # noqa: E402


def pick():
    """(name, sequence) of the reference and of the queries, as the tests take them from the bundled alignment"""
    names, seqs = F.load_bundled()
    ref_i = min(range(2000), key=lambda i: (sum(1 for c in seqs[i] if c not in b"ACGT"), i))
    ref = seqs[ref_i].replace(b"-", b"")
    qs = [(names[i], seqs[i].replace(b"-", b"")) for i in range(0, len(seqs), len(seqs) // 24) if i != ref_i][:24]
    return names[ref_i], ref, qs


def main():
    ref_name, ref, qs = pick()
    out = {"what": "oracle output for uvaialign on sequences of tests/golden/03.unique_acgt.aln.xz (gap characters removed); made by tools/make_golden_align.py",
           "reference": ref_name, "reference_length": len(ref), "queries": []}
    for name, s in qs:
        score, row, cells = O.uvaialign_query(ref, s)
        out["queries"].append({"name": name, "length": len(s), "accepted": O.uvaialign_accepts(s, len(ref)), "score": score, "cells": cells,
                               "row_sha1": hashlib.sha1(row).hexdigest()})
    path = os.path.join(ROOT, "tests", "golden", "uvaialign_oracle_snapshot.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes;", "scores", [q["score"] for q in out["queries"]])


if __name__ == "__main__":
    main()
