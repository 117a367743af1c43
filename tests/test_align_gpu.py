"""The wavefront aligner of `uvaialign` on the MI355X against the CPU restatement (`pytest -m gpu`): scores and aligned rows bit
for bit (oracle/wfa_oracle.c; the WFA library itself is an absent submodule, so what is compared is the published algorithm as
the oracle states it: see tests/test_wfa_oracle.py for what pins the oracle)."""
import numpy as np
import pytest

import fixtures as F
import oracle_lib as O
from uvaia_amd import align

pytestmark = pytest.mark.gpu


def _check(ref, seqs, al, **oracle_kw):
    score, rows = al.align(seqs)
    for i, t in enumerate(seqs):
        if oracle_kw:
            want, cigar, _, _ = O.wfa_align(ref, t, **oracle_kw)
            row = bytearray()
            pos = 0
            for op in cigar:
                if op in b"MX":
                    row.append(t[pos]); pos += 1
                elif op == ord("I"):
                    pos += 1
                else:
                    row.append(ord("-"))
            want_row = bytes(row)
        else:
            want, want_row, _ = O.uvaialign_query(ref, t)
        assert score[i] == want, (i, score[i], want)
        assert rows[i].tobytes() == want_row, i
    return score


def test_small_random_pairs_equal_oracle():
    """lengths 1..400, heavy divergence, indels, N runs: every branch of the recurrences and of the backtrace"""
    rng = np.random.default_rng(5)
    for rep in range(8):
        L = int(rng.integers(1, 400))
        ref = F.random_acgt(L, 100 + rep)
        seqs = F.unaligned_queries(ref, 30, 200 + rep, p_snp=0.05, p_indel=0.02, max_indel=8, n_runs=(10, 10, 40), run_prob=0.4, ambiguity=0.01)
        seqs += [ref, ref[: max(1, L // 3)], ref + ref[: L // 2], b"A", b"N" * max(1, L // 2)]
        with align.Aligner(ref) as al:
            _check(ref, seqs, al)


def test_sars_cov_2_shaped_queries_equal_oracle():
    """29 903-column reference, queries with SNPs, short indels, leading/trailing N runs and an amplicon dropout: long wavefronts,
    the adaptive reduction at work (widths above 128), match runs of thousands of characters"""
    ref = F.random_acgt(29903, 7)
    seqs = F.unaligned_queries(ref, 96, 8)
    with align.Aligner(ref) as al:
        score = _check(ref, seqs, al)
        st = al.stats()
    assert st["passes"] == 1 and st["cells"] > 0
    assert score.max() > 600                                   # N runs cost 4 per site: the reduction is in play


def test_results_do_not_depend_on_the_order_of_the_pool():
    """the aligner starts the queries of a pool by expected cost (the dearer half first, the cheaper half in descending order):
    every query's row and score are its own, whatever came before it and wherever it stands in the pool"""
    ref = F.random_acgt(4000, 21)
    seqs = F.unaligned_queries(ref, 120, 22, n_runs=(40, 30, 300)) + [b"N" * 3000, ref, ref[:100], b"N" * 10 + ref[10:]]
    with align.Aligner(ref) as al:
        s1, r1 = al.align(seqs)
        s2, r2 = al.align(seqs[::-1])
        s3, r3 = al.align(seqs[:7])
    assert np.array_equal(s1, s2[::-1]) and all(a.tobytes() == b.tobytes() for a, b in zip(r1, r2[::-1]))
    assert np.array_equal(s1[:7], s3) and all(a.tobytes() == b.tobytes() for a, b in zip(r1[:7], r3))
    assert len(set(int(x) for x in s1)) > 20                    # scores differ widely: the start order is not the pool's


def test_resident_pool_and_repeat_runs_agree():
    ref = F.random_acgt(5000, 9)
    seqs = F.unaligned_queries(ref, 300, 10, n_runs=(50, 40, 200))
    gold_score, gold_rows = O.uvaialign_batch(ref, seqs)
    with align.Aligner(ref) as al:
        al.load(seqs)
        for _ in range(2):
            al.run()
            score, rows = al.fetch()
            assert np.array_equal(score, gold_score) and np.array_equal(rows, gold_rows)


@pytest.mark.parametrize("opts", [dict(min_wavefront_length=0), dict(mismatch=3, gap_opening=5, gap_extension=1), dict(mismatch=2, gap_opening=12, gap_extension=3),
                                  dict(min_wavefront_length=16, max_distance_threshold=20), dict(mismatch=1, gap_opening=0, gap_extension=1)])
def test_options_follow_the_oracle(opts):
    ref = F.random_acgt(1500, 13)
    seqs = F.unaligned_queries(ref, 24, 14, p_snp=0.01, p_indel=0.003, n_runs=(30, 30, 120))
    d = align.default_options()
    pen = (0, opts.get("mismatch", d.mismatch), opts.get("gap_opening", d.gap_opening), opts.get("gap_extension", d.gap_extension))
    kw = dict(penalties=pen, min_wavefront_length=opts.get("min_wavefront_length", d.min_wavefront_length),
              max_distance_threshold=opts.get("max_distance_threshold", d.max_distance_threshold))
    with align.Aligner(ref, **opts) as al:
        _check(ref, seqs, al, **kw)


def test_queries_that_find_the_pool_empty_are_run_again():
    """a small workspace (13 chunks of 2 MB; the widest queries need 10): four queries in flight exhaust the pool, the ones that find
    it empty give their chunks back and finish in a later pass with fewer blocks"""
    ref = F.random_acgt(6000, 17)
    seqs = F.unaligned_queries(ref, 24, 18, n_runs=(90, 75, 1200), run_prob=0.9)
    with align.Aligner(ref, workspace_bytes=26 << 20, max_blocks=8) as al:
        _check(ref, seqs, al)
        assert al.stats()["passes"] >= 2


def test_a_query_beyond_the_whole_workspace_is_an_error_not_a_wrong_row():
    ref = F.random_acgt(20000, 19)
    with align.Aligner(ref, workspace_bytes=64 << 20, max_blocks=4) as al:
        with pytest.raises(align.AlignError) as ei:
            al.align([b"N" * 15000])
        assert ei.value.code == -3


def test_bad_arguments():
    with pytest.raises(align.AlignError):
        align.Aligner(b"ACGT", mismatch=0)
    with pytest.raises(align.AlignError):
        align.Aligner(b"ACGT", mismatch=64)
    with align.Aligner(b"ACGTACGT") as al:
        with pytest.raises(align.AlignError) as ei:
            al.fetch()                                   # nothing has run yet
        assert ei.value.code == -5
        score, rows = al.align([])
        assert len(score) == 0 and rows.shape == (0, 8)
        score, rows = al.align([b"ACGTTACGT"])
        assert score[0] == 8 and rows[0].tobytes() == b"ACGTACGT"


# ---- the drop-in command line (src/align.c main) ----
import lzma
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UVAIALIGN = os.path.join(ROOT, "bin", "uvaialign")


def _write_fasta(path, names, seqs, opener=open, width=None):
    with opener(path, "wb") as fh:
        for n, s in zip(names, seqs):
            fh.write(b">" + n.encode() + b"\n")
            if width:
                for a in range(0, len(s), width):
                    fh.write(s[a:a + width] + b"\n")
            else:
                fh.write(s + b"\n")


def test_uvaialign_cli_matches_oracle(tmp_path):
    """reference + two query files (one xz, one multi-line with lower case), small pools so that several batches run; rejected
    sequences (size, N fraction) leave the stream as in src/align.c:199-221; --stdout and the compressed file hold the same rows"""
    ref = F.random_acgt(4000, 51)
    good = F.unaligned_queries(ref, 70, 52, n_runs=(60, 50, 200))
    bad = [ref[:2000], b"N" * 2500 + ref[:1500], ref + ref]                        # too short, too many N, too long
    seqs = good[:30] + bad[:1] + good[30:55] + bad[1:] + good[55:]
    names = ["s%d" % i for i in range(len(seqs))]
    _write_fasta(tmp_path / "ref.fa", ["the_ref", "second_record_is_ignored"], [ref, b"ACGT"], width=60)
    half = 40
    _write_fasta(tmp_path / "q1.fa.xz", names[:half], seqs[:half], opener=lzma.open)
    _write_fasta(tmp_path / "q2.fa", names[half:], [s.lower() for s in seqs[half:]], width=70)
    want = []
    for n, s in zip(names, seqs):
        if O.uvaialign_accepts(s, len(ref), 0.5):
            want.append((n, O.uvaialign_query(ref, s)[1]))
    assert len(want) == len(good)
    cmd = [UVAIALIGN, "-r", str(tmp_path / "ref.fa"), str(tmp_path / "q1.fa.xz"), str(tmp_path / "q2.fa"), "-p", "16"]
    r = subprocess.run(cmd + ["--stdout"], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    got_names, got_rows = F.read_fasta_bytes(r.stdout)
    assert [(n, s) for n, s in zip(got_names, got_rows)] == want
    err = r.stderr.decode()
    assert "has size too different from reference" in err and "proportion of N etc." in err
    assert "Output %d aligned sequences." % len(want) in err
    subprocess.run(cmd + ["-o", str(tmp_path / "out")], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    with lzma.open(tmp_path / "out.aln.xz", "rb") as fh:
        assert fh.read() == r.stdout
    r2 = subprocess.run(cmd + ["--stdout", "--devices", "0,0,0"], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)   # three aligners on one GPU
    assert r2.stdout == r.stdout and b"aligned on 3 GPUs" in r2.stderr


def test_uvaialign_cli_against_the_committed_snapshot(tmp_path):
    """bin/uvaialign on real sequences (the bundled alignment with its gaps removed) against tests/golden/uvaialign_oracle_snapshot.json,
    without the oracle in the loop: scores through the library, rows through the command line"""
    import hashlib
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("make_golden_align", os.path.join(ROOT, "tools", "make_golden_align.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    snap = json.load(open(os.path.join(F.GOLDEN, "uvaialign_oracle_snapshot.json")))
    ref_name, ref, qs = mod.pick()
    assert ref_name == snap["reference"] and [n for n, _ in qs] == [q["name"] for q in snap["queries"]]
    with align.Aligner(ref) as al:
        score, rows = al.align([s for _, s in qs])
    assert list(score) == [q["score"] for q in snap["queries"]]
    assert [hashlib.sha1(r.tobytes()).hexdigest() for r in rows] == [q["row_sha1"] for q in snap["queries"]]
    _write_fasta(tmp_path / "ref.fa", [ref_name], [ref])
    _write_fasta(tmp_path / "q.fa", [n for n, _ in qs], [s for _, s in qs], width=80)
    r = subprocess.run([UVAIALIGN, "-r", str(tmp_path / "ref.fa"), str(tmp_path / "q.fa"), "--stdout", "-a", "1.0"], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    got_names, got_rows = F.read_fasta_bytes(r.stdout)
    assert got_names == [q["name"] for q in snap["queries"]]
    assert [hashlib.sha1(x).hexdigest() for x in got_rows] == [q["row_sha1"] for q in snap["queries"]]


def test_a_query_that_is_half_N_equals_oracle():
    """the most ambiguous query the filter of src/align.c:204-212 lets through (49 % N in three long runs): a score near 60 000, wavefronts
    wider than the 2 560 diagonals that stay in LDS (those steps read their sources from memory), 57 M cells = 270 MB of history"""
    ref = F.random_acgt(29903, 7)
    q = bytearray(ref)
    for a, n in ((100, 6000), (9000, 5000), (20000, 3600)):
        q[a:a + n] = b"N" * n
    q = bytes(q)
    assert O.uvaialign_accepts(q, len(ref))
    with align.Aligner(ref) as al:
        score = _check(ref, [q, ref], al)
        assert score[0] > 55000 and score[1] == 0
        assert al.stats()["cells"] > 50_000_000


@pytest.mark.parametrize("opts", [dict(min_wavefront_length=0), dict(min_wavefront_length=3000, max_distance_threshold=2000)])
def test_wavefronts_wider_than_the_lds_ring_equal_oracle(opts):
    """complete (or barely reduced) wavefronts over a run of 2 000 N: widths far above the 2 560 diagonals kept in LDS, so steps read
    sources that never were in LDS, sources that were (the narrow steps before) and write their own I/D wavefronts to memory"""
    ref = F.random_acgt(8000, 71)
    q = bytearray(ref)
    q[3000:5000] = b"N" * 2000
    q = bytes(q[:6500] + q[6510:])
    d = align.default_options()
    kw = dict(penalties=(0, d.mismatch, d.gap_opening, d.gap_extension), min_wavefront_length=opts["min_wavefront_length"],
              max_distance_threshold=opts.get("max_distance_threshold", d.max_distance_threshold))
    with align.Aligner(ref, **opts) as al:
        _check(ref, [q, ref[:7000]], al, **kw)
