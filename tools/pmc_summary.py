#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection CSVs per kernel:  python tools/pmc_summary.py label=path.csv ... > out.json
Values are reported as rocprofv3 prints them (FETCH_SIZE / WRITE_SIZE: KB per dispatch; apply the gfx950 corrections of
MI355X_MICROARCH.md when turning FETCH_SIZE into bytes)."""
import collections
import csv
import json
import sys


def summarise(path):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        per[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    out = {}
    for k, counters in per.items():
        out[k] = {}
        for c, disp in counters.items():
            v = list(disp.values())
            out[k][c] = {"n": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v), "sum": sum(v)}
    return out


if __name__ == "__main__":
    res = {}
    for a in sys.argv[1:]:
        label, path = a.split("=", 1)
        res[label] = summarise(path)
    json.dump(res, sys.stdout, indent=1)
