#!/usr/bin/env python3
"""Writes tests/golden/timed_path_digests.json: the SHA-256 digests of the CPU oracle's answers to the generator workloads of
tests/test_timed_path_gpu.py (tests/timed_path_workloads.py).  Runs the oracle (about a minute); no GPU.
Usage: python tests/golden/make_timed_path_digests.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import timed_path_workloads as W  # noqa: E402

out = {key: W.Workload(key).oracle_digest() for key in W.SPECS}
json.dump(out, open(W.DIGESTS, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
