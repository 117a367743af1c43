"""bench.py on a box without a GPU: it must refuse to run (the engine has no CPU path) rather than time something else, and its
command line must keep the driver's contract (--gpus/--steps/--warmup, defaults that finish in minutes)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flags_and_defaults():
    sys.path.insert(0, ROOT)
    import bench
    argv = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = bench.parse()
        assert (a.gpus, a.queries, a.refs, a.nbest, a.mode) == (1, 1000, 100000, 100, "iupac") and a.steps <= 10 and a.warmup >= 1
        sys.argv = ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "2"]
        a = bench.parse()
        assert (a.gpus, a.steps, a.warmup) == (8, 7, 2)
    finally:
        sys.argv = argv


def test_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "no CPU path" in (p.stderr + p.stdout)
    assert not p.stdout.strip().startswith("{")          # no result line
