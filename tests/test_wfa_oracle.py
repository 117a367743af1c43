"""The CPU restatement of the `uvaialign` path (oracle/wfa_oracle.c) checked on the CPU.

The WFA library the reference calls (src/align.c:304-309,357-364) is an absent submodule and the reference holds no test or
documented output of uvaialign: PARITY IS UNPINNED for the aligner.  What can be pinned is what the published algorithm fixes:
with complete wavefronts the score is the optimal gap-affine score (checked against an independent Gotoh recurrence), every
CIGAR spells its pair of sequences at exactly the reported score, the adaptive reduction never reports less than the optimum,
and the projection follows src/align.c:366-390."""
import numpy as np
import pytest

import fixtures as F
import oracle_lib as O


def _pairs(seed, n, max_len=220):
    rng = np.random.default_rng(seed)
    for i in range(n):
        L = int(rng.integers(1, max_len))
        p = F.random_acgt(L, seed * 1000 + i)
        t = F.unaligned_queries(p, 1, seed * 7919 + i, p_snp=float(rng.random()) * 0.2, p_indel=float(rng.random()) * 0.05, max_indel=6,
                                n_runs=(5, 5, 20), run_prob=0.3, ambiguity=0.01)[0]
        yield p, t


def test_complete_wavefronts_give_the_optimal_gap_affine_score():
    for p, t in _pairs(1, 400):
        score, cigar, cells, width = O.wfa_align(p, t, min_wavefront_length=0)
        assert score == O.gotoh_score(p, t)
        assert O.cigar_score(cigar, p, t) == score
        assert cells >= 1 and width >= 1


@pytest.mark.parametrize("pen", [(0, 4, 6, 2), (0, 1, 0, 1), (0, 3, 5, 1), (0, 2, 12, 3), (0, 7, 1, 5)])
def test_other_penalties_are_optimal_too(pen):
    for p, t in _pairs(2, 60, max_len=120):
        score, cigar, _, _ = O.wfa_align(p, t, penalties=pen, min_wavefront_length=0)
        assert score == O.gotoh_score(p, t, pen) == O.cigar_score(cigar, p, t, pen)


def test_reduced_wavefronts_are_consistent_and_never_below_the_optimum():
    ref = F.random_acgt(6000, 11)
    seqs = F.unaligned_queries(ref, 6, 12, n_runs=(90, 75, 350), run_prob=1.0)
    trimmed = 0
    for t in seqs:
        full, _, cells_full, _ = O.wfa_align(ref, t, min_wavefront_length=0)
        red, cigar, cells_red, width = O.wfa_align(ref, t)                     # src/align.c:308: 128, 512
        assert red >= full and O.cigar_score(cigar, ref, t) == red
        assert cells_red <= cells_full
        trimmed += cells_red < cells_full
    assert trimmed > 0            # the N runs make the wavefronts long enough for the reduction to act


def test_projection_rule_of_update_query_aligned():
    """src/align.c:366-390: one output character per reference position; M/X copy, D gives '-', inserted characters vanish."""
    ref = F.random_acgt(3000, 21)
    for t in F.unaligned_queries(ref, 8, 22, p_indel=0.002, n_runs=(40, 30, 100)):
        score, row, _ = O.uvaialign_query(ref, t)
        assert score >= 0 and len(row) == len(ref)
        kept = bytes(c for c in row if c != ord("-"))
        it = iter(t)
        assert all(c in it for c in kept)                                      # a subsequence of the query, in order
        _, cigar, _, _ = O.wfa_align(ref, t)
        assert row.count(b"-") == cigar.count(b"D") and len(t) - len(kept) == cigar.count(b"I")


def test_identical_and_degenerate_inputs():
    ref = F.random_acgt(500, 31)
    assert O.uvaialign_query(ref, ref)[:2] == (0, ref)
    score, row, _ = O.uvaialign_query(ref, ref[:100])                           # a long deletion: 6 + 2 * 400
    assert score == 6 + 2 * 400 and row.count(b"-") == 400
    score, row, _ = O.uvaialign_query(ref[:100], ref)                           # a long insertion leaves no trace in the row
    assert score == 6 + 2 * 400 and len(row) == 100 and b"-" not in row
    score, row, _ = O.uvaialign_query(b"A", b"C")
    assert (score, row) == (4, b"C")


def test_query_filter_of_the_read_loop():
    """src/align.c:199-213: length within [2/3, 3/2] of the reference, N fraction <= a, ACGT fraction >= 1 - 1.1 a"""
    ref_len = 3000
    ok = F.random_acgt(3000, 41)
    assert O.uvaialign_accepts(ok, ref_len)
    assert not O.uvaialign_accepts(ok[:1999], ref_len) and O.uvaialign_accepts(ok[:2000], ref_len)
    assert not O.uvaialign_accepts(ok + ok[:1501], ref_len) and O.uvaialign_accepts(ok + ok[:1500], ref_len)
    assert not O.uvaialign_accepts(b"N" * 1600 + ok[:1400], ref_len) and O.uvaialign_accepts(b"N" * 1400 + ok[:1600], ref_len)
    assert not O.uvaialign_accepts(b"Y" * 1500 + ok[:1500], ref_len, 0.3)       # too few ACGT although nothing is N


def test_oracle_reproduces_its_committed_snapshot():
    """tests/golden/uvaialign_oracle_snapshot.json (tools/make_golden_align.py): bundled sequences with their gaps removed against the
    cleanest one.  A regression pin of the oracle's own numbers, not reference output."""
    import hashlib
    import json
    import os
    sys_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_align", os.path.join(sys_path, "make_golden_align.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    snap = json.load(open(os.path.join(F.GOLDEN, "uvaialign_oracle_snapshot.json")))
    ref_name, ref, qs = mod.pick()
    assert ref_name == snap["reference"] and len(ref) == snap["reference_length"] and len(qs) == len(snap["queries"])
    for (name, s), want in list(zip(qs, snap["queries"]))[:8]:              # the first eight: the whole list runs on the GPU test
        score, row, cells = O.uvaialign_query(ref, s)
        assert (name, len(s), score, cells, hashlib.sha1(row).hexdigest()) == (want["name"], want["length"], want["score"], want["cells"], want["row_sha1"])
