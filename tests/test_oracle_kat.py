"""Pins the oracle's scoring kernels on the reference's own known-answer rows.

KAT 1: /root/reference/README.md:227-233 -- query England/NORW-3078E97/2021 against six references; all seven
       sequences are in data/03.unique_acgt.aln.xz (copied to tests/golden/).  Columns 4-7 and 9 are pairwise
       quantities (ACGT_matches, text_matches, partial_matches, valid_pair_comparisons, valid_ref_sites);
       column 8 depends on the whole (undocumented) query set of that run and is not a KAT.
KAT 2: /root/reference/README.md:307-316 -- three toy sequences.
"""
import fixtures as F
import oracle_lib as O

README_QUERY = "England/NORW-3078E97/2021"
README_ROWS = [  # reference, ACGT, text, partial, valid_pairs, valid_ref_sites   (README.md:228-233)
    ("England/NORW-3061C36/2021", 14985, 14985, 14987, 14988, 29843),
    ("England/NORW-302EA07/2021", 14984, 14984, 14986, 14988, 29875),
    ("England/NORW-3034A7D/2021", 14984, 14984, 14986, 14988, 29869),
    ("England/NORW-3034B26/2021", 14984, 14984, 14986, 14988, 29851),
    ("England/NORW-306AB26/2021", 14983, 14983, 14985, 14988, 29813),
    ("England/NORW-31425AC/2022", 14982, 14982, 14985, 14988, 29169),
]


def test_readme_table_rows(bundled_db):
    names, seqs = bundled_db
    by = dict(zip(names, seqs))
    q = by[README_QUERY]
    for ref, acgt, text, partial, valid, ref_sites in README_ROWS:
        r = by[ref]
        assert O.score4(r, q) == [acgt, text, partial, valid], ref
        assert O.lib().orc_count_non_N(r, len(r)) == ref_sites, ref


def test_readme_table_order_is_the_heap_order():
    # rows are printed best-first by the 6-int lexicographic key (README.md:249-262, src/min_heap.c:41-47)
    L = O.lib()
    import ctypes as C
    keys = [(a, t, p, v, a, s) for _, a, t, p, v, s in README_ROWS]
    for k0, k1 in zip(keys, keys[1:]):
        a = (C.c_int * 6)(*k0)
        b = (C.c_int * 6)(*k1)
        assert L.orc_compare_score(a, b) < 0


def test_readme_toy_example():
    s1, s2, s3 = b"AACGTTA--", b"AACG-TAM-", b"MNCGTTMC-"
    r12, r13, r23 = O.score4(s1, s2), O.score4(s1, s3), O.score4(s2, s3)
    assert (r12[0], r12[2], r12[3]) == (6, 6, 6)
    assert (r13[0], r13[2], r13[3]) == (4, 6, 6)
    assert (r23[0], r23[2], r23[3]) == (3, 6, 6)


def test_truncation_stops_at_maxdist():
    a = b"ACGTACGTAC"
    b = b"TGCATGCATG"   # every site mismatches
    assert O.score4(a, b, maxdist=3) == [0, 0, 0, 3]
    assert O.score_acgt(a, b, maxdist=4) == [4, 4]
    assert O.score_acgt(a, b) == [10, 10]


def test_oracle_reproduces_the_committed_config1_table(bundled_db):
    """tests/golden/config1_oracle_snapshot.json (tools/make_golden.py) pins the oracle's own output on the bundled alignment:
    a change in oracle/uvaia_oracle.c that alters a score, an order or a tolerance shows up here, on the CPU."""
    import json
    import os
    snap = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config1_oracle_snapshot.json")))
    names, seqs = bundled_db
    by = dict(zip(names, seqs))
    qn = F.sample_names_1k()[:10]
    for run in snap["runs"]:
        if run["pool"] != 64 and run["trim"] != 230:
            continue                                 # four of the eight runs keep this test within seconds
        q = O.Query([by[n] for n in qn], qn, acgt=run["acgt"], trim=run["trim"])
        g = O.search(q, seqs, names, pool=run["pool"], nbest=run["nbest"])
        assert list(q.names) == run["queries"] and list(g.final_T) == run["final_T"]
        assert (g.n_lowqual, len(g.saved)) == (run["n_lowqual"], run["n_saved"])
        assert [[[name] + list(score) for _, name, score in rows] for rows in g.rows] == run["rows"]
