"""CPU tests of the product's host C code (uvaia_amd/csrc/host) against the oracle's literal restatement."""
import ctypes as C
import gzip
import lzma
import os

import numpy as np
import pytest

import fixtures as F
import oracle_lib as O
from uvaia_amd import hostlib as H


@pytest.fixture(scope="module", autouse=True)
def _built():
    H.build_library()


def _same_query(a, b):
    assert a.ntax == b.ntax and a.nchar == b.nchar and a.trim == b.trim and a.dist == b.dist
    assert a.names == b.names
    assert a.seqs == b.seqs
    if a.ntax:
        assert a.consensus == b.consensus
        for f in ("idx_c", "idx_m", "idx"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f


@pytest.mark.parametrize("acgt", [False, True])
@pytest.mark.parametrize("trim", [0, 230, 20000])
def test_query_preparation_bundled(bundled_db, acgt, trim):
    names, seqs = bundled_db
    by = dict(zip(names, seqs))
    qn = F.sample_names_1k()[:40]
    qs = [by[n] for n in qn]
    _same_query(H.PreparedQuery(qs, qn, trim=trim, acgt=acgt), O.Query(qs, qn, trim=trim, acgt=acgt))


@pytest.mark.parametrize("acgt", [False, True])
@pytest.mark.parametrize("keep", [False, True])
@pytest.mark.parametrize("ball", [False, True])
def test_query_preparation_with_redundancy(acgt, keep, ball):
    # near-duplicates with different N patterns: exercises exclude_redundant_query_sequences
    base, root, cols = F.synth_alignment(12, 1500, seed=3, p_snp=0.004)
    rng = np.random.default_rng(9)
    qs = []
    for s in base:
        qs.append(s)
        t = bytearray(s)
        a, ln = int(rng.integers(0, 1400)), int(rng.integers(1, 90))
        t[a:a + ln] = b"N" * ln
        qs.append(bytes(t))
        qs.append(s)                       # exact duplicate
    names = ["q%d" % i for i in range(len(qs))]
    a = H.PreparedQuery(qs, names, acgt=acgt, keep_resolved=keep, is_ball=ball, dist=3)
    b = O.Query(qs, names, acgt=acgt, keep_resolved=keep, is_ball=ball, dist=3)
    _same_query(a, b)
    if ball or keep:
        assert a.ntax < len(qs)


def test_host_pruning_loop_stays_available_for_large_sets(request):
    """From 512 queries on the pair test of the pruning runs on the device (tests/test_query_prep_gpu.py); uvaia_set_prune_mode(host)
    keeps the serial host loop, which is what this CPU test can run: same survivors as the oracle."""
    H.set_prune_mode("host")
    request.addfinalizer(lambda: H.set_prune_mode("auto"))
    base, root, cols = F.synth_alignment(270, 400, seed=33, p_snp=0.01)
    qs = []
    for i, s in enumerate(base):
        qs.append(s)
        t = bytearray(s)
        t[(i * 7) % 300:(i * 7) % 300 + 20] = b"N" * 20
        qs.append(bytes(t))
    names = ["q%d" % i for i in range(len(qs))]
    assert len(qs) >= 512
    a = H.PreparedQuery(qs, names, keep_resolved=True)
    b = O.Query(qs, names, keep_resolved=True)
    _same_query(a, b)
    assert a.ntax < len(qs)


def test_site_counts_agree_with_the_character_classes_for_every_byte():
    """quick_count_sequence_non_N / quick_count_sequence_acgt are written as vectorisable byte comparisons (they sit in the serial
    read loop); they must count exactly what is_site_valid / is_site_acgt (src/utils.c:255-295) say, for all 256 byte values."""
    L = H.load_library()
    L.quick_count_sequence_non_N.restype = C.c_int
    L.quick_count_sequence_non_N.argtypes = [C.c_char_p, C.c_size_t]
    L.quick_count_sequence_acgt.restype = C.c_int
    L.quick_count_sequence_acgt.argtypes = [C.c_char_p, C.c_size_t]
    L.is_site_valid.restype = C.c_int
    L.is_site_valid.argtypes = [C.c_char]
    L.is_site_acgt.restype = C.c_int
    L.is_site_acgt.argtypes = [C.c_char]
    valid = [L.is_site_valid(bytes([b])) for b in range(256)]
    acgt = [L.is_site_acgt(bytes([b])) for b in range(256)]
    assert sum(acgt) == 8 and 256 - sum(valid) == 9
    for b in range(1, 256):                                  # single characters (0 would end the C string view, counted below)
        one = bytes([b])
        assert L.quick_count_sequence_non_N(one, 1) == valid[b] and L.quick_count_sequence_acgt(one, 1) == acgt[b], b
    rng = np.random.default_rng(4)
    for n in (0, 1, 15, 16, 17, 63, 1000, 29903):
        buf = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        assert L.quick_count_sequence_non_N(buf, n) == sum(valid[b] for b in buf)
        assert L.quick_count_sequence_acgt(buf, n) == sum(acgt[b] for b in buf)


def test_low_quality_queries_are_dropped():
    good, _, _ = F.synth_alignment(3, 600, seed=1)
    bad = b"N" * 400 + good[0][400:]
    qs = [good[0], bad, good[1], b"ACGT" + b"-" * 596, good[2]]
    names = list("abcde")
    a, b = H.PreparedQuery(qs, names), O.Query(qs, names)
    _same_query(a, b)
    assert sorted(a.names) == ["a", "c", "e"]


def test_host_heap_matches_oracle_heap():
    L, OL = H.load_library(), O.lib()
    OL.orc_heap_new.restype = C.POINTER(O.OrcItem)   # placeholder to keep ctypes happy; real signature below
    class OrcHeap(C.Structure):
        _fields_ = [("seq", C.POINTER(O.OrcItem)), ("heap_size", C.c_int), ("n", C.c_int), ("max_incompatible", C.c_int)]
    OL.orc_heap_new.restype = C.POINTER(OrcHeap)
    OL.orc_heap_new.argtypes = [C.c_int]
    OL.orc_heap_insert.restype = C.c_int
    OL.orc_heap_insert.argtypes = [C.POINTER(OrcHeap), C.POINTER(O.OrcItem)]
    OL.orc_heap_finalise.argtypes = [C.POINTER(OrcHeap)]
    OL.orc_heap_del.argtypes = [C.POINTER(OrcHeap)]
    rng = np.random.default_rng(2)
    for size in (1, 2, 3, 7, 50):
        for n_items in (0, 1, size - 1, size, size + 1, 5 * size + 3):
            n_items = max(0, n_items)
            h, o = L.new_heap_t(size), OL.orc_heap_new(size)
            for i in range(n_items):
                sc = [int(x) for x in rng.integers(0, 3, size=6)]       # tiny range: plenty of full ties
                nm = ("s%d" % i).encode()
                it = H.QItem((C.c_int * 6)(*sc), nm)
                ot = O.OrcItem((C.c_int * 6)(*sc), nm, i)
                assert bool(L.heap_insert(h, it)) == bool(OL.orc_heap_insert(o, C.byref(ot)))
                assert h.contents.n == o.contents.n
                for s in range(1, h.contents.n + 1):                    # same layout after every operation
                    assert list(h.contents.seq[s].score) == list(o.contents.seq[s].score)
                    assert h.contents.seq[s].name == o.contents.seq[s].name
            if n_items != size - 1 or size == 1:                          # n == size-1 reads an unused slot in the reference
                L.heap_finalise_heap_qsort(h); OL.orc_heap_finalise(o)
                assert h.contents.heap_size == o.contents.heap_size
                for s in range(h.contents.n):
                    assert list(h.contents.seq[s].score) == list(o.contents.seq[s].score)
                    assert h.contents.seq[s].name == o.contents.seq[s].name
            L.del_heap_t(h); OL.orc_heap_del(o)


def test_readfasta_line_shapes(tmp_path):
    """Clean upper-case lines take a copy-only path in the parser; CR LF endings, tabs, spaces, lower case and a last line without
    a line end must come out exactly as the character-by-character path gives them."""
    L = H.load_library()
    body = (b">a\r\nACGTACGT\r\nNNNN--RY\r\n"            # CR LF, clean lines
            b">b\nACGT\tAC gt\nacgtn\n"                      # tab, space, lower case
            b">c\n" + b"ACGTNRYKM-" * 3000 + b"\n"            # one long clean line
            b">d\nAC\nGT")                                     # no line end at the end of the file
    p = tmp_path / "shapes.fa"
    p.write_bytes(body)
    want = [("a", b"ACGTACGTNNNN--RY"), ("b", b"ACGTACGTACGTN"), ("c", b"ACGTNRYKM-" * 3000), ("d", b"ACGT")]
    r = L.new_readfasta(str(p).encode())
    got = []
    while L.readfasta_next(r) >= 0:
        got.append((r.contents.name.decode(), C.string_at(r.contents.seq, r.contents.seqlength)))
    L.del_readfasta(r)
    assert got == want


def test_readfasta_streams_plain_gz_xz(tmp_path):
    L = H.load_library()
    recs = [("seq one", b"acgtnn--ACGT"), ("s2", b"AC GT\nAC"), ("third/3", b"NNNN")]
    text = b"".join(b">" + n.encode() + b"\n" + s + b"\n\n" for n, s in recs)
    paths = {"plain": tmp_path / "a.fa", "gz": tmp_path / "a.fa.gz", "xz": tmp_path / "a.fa.xz"}
    paths["plain"].write_bytes(text)
    with gzip.open(paths["gz"], "wb") as fh: fh.write(text)
    with lzma.open(paths["xz"], "wb") as fh: fh.write(text)
    want = [(n, s.replace(b" ", b"").replace(b"\n", b"").upper()) for n, s in recs]
    for kind, p in paths.items():
        r = L.new_readfasta(str(p).encode())
        got = []
        while True:
            n = L.readfasta_next(r)
            if n < 0:
                break
            got.append((r.contents.name.decode(), C.string_at(r.contents.seq, n)))
        L.del_readfasta(r)
        assert got == want, kind


def test_truncated_compressed_reference_is_an_error_not_a_short_database(tmp_path):
    """A decompressor that dies in mid-stream looks like end of data to getline(): the reader must end the run with a message
    (biomcmc_error), while a reader that is closed before its end of data (the tool then gets SIGPIPE) must not."""
    import subprocess
    import sys
    rng = np.random.default_rng(3)
    text = b"".join(b">r%d\n" % i + bytes(rng.choice(list(b"ACGT"), 3000).astype(np.uint8)) + b"\n" for i in range(400))
    good, cut = tmp_path / "db.fa.xz", tmp_path / "cut.fa.xz"
    with lzma.open(good, "wb") as fh: fh.write(text)
    cut.write_bytes(good.read_bytes()[:-2000])
    prog = ("import sys; sys.path.insert(0, %r); from uvaia_amd import hostlib as H; L = H.load_library()\n"
            "r = L.new_readfasta(sys.argv[1].encode()); n = 0\n"
            "while L.readfasta_next(r) >= 0:\n"
            "    n += 1\n"
            "    if len(sys.argv) > 2 and n == 3: break\n"
            "L.del_readfasta(r); print('read', n)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ok = subprocess.run([sys.executable, "-c", prog, str(good)], capture_output=True, text=True)
    assert ok.returncode == 0 and "read 400" in ok.stdout
    early = subprocess.run([sys.executable, "-c", prog, str(good), "stop-early"], capture_output=True, text=True)
    assert early.returncode == 0 and "read 3" in early.stdout
    bad = subprocess.run([sys.executable, "-c", prog, str(cut)], capture_output=True, text=True)
    assert bad.returncode != 0 and "decompressor reported an error" in bad.stderr and "read" not in bad.stdout


def test_query_alignment_length_follows_the_kept_sequences(tmp_path):
    """The first query record is dropped (too short) and has another length than the kept ones: nchar must be theirs."""
    seqs = [b"ACGT", b"ACGTACGTACGTACGTACGT", b"ACGTACGTACGTACGTACGA", b"ACGTACGTACGTACGTACCA"]
    p = tmp_path / "q.fa"
    p.write_bytes(b"".join(b">q%d\n%s\n" % (i, s) for i, s in enumerate(seqs)))
    L = H.load_library()
    L.read_fasta_alignment_from_file.restype = C.POINTER(H.Alignment)
    L.read_fasta_alignment_from_file.argtypes = [C.c_char_p, C.c_int]
    L.uvaia_keep_only_valid_sequences.argtypes = [C.POINTER(H.Alignment), C.c_double, C.c_bool]
    L.del_alignment.argtypes = [C.POINTER(H.Alignment)]
    aln = L.read_fasta_alignment_from_file(str(p).encode(), 0)
    assert aln.contents.ntax == 4 and aln.contents.nchar == 4
    L.uvaia_keep_only_valid_sequences(aln, 0.5, True)
    assert aln.contents.ntax == 3 and aln.contents.nchar == 20
    L.del_alignment(aln)


def test_generator_is_deterministic_and_shaped():
    g = H.Synth(29903, seed=20241008, preset=0)
    a, na = g.generate(1000, 64)
    b, nb = g.generate(1032, 32)
    assert np.array_equal(a[32:], b) and np.array_equal(na[32:], nb)      # sequence i depends only on (seed, i)
    rows, non_n = g.generate(0, 2000)
    frac_invalid = 1.0 - non_n / 29903.0
    assert 0.10 < np.median(frac_invalid) < 0.25 and frac_invalid.max() < 0.5
    assert set(np.unique(rows)) <= set(b"ACGTN-YRKMSWDHVB")
    want = np.array([O.lib().orc_count_non_N(rows[i].tobytes(), 29903) for i in range(50)])
    assert np.array_equal(want, non_n[:50])
    clean = H.Synth(29903, seed=20241008, preset=1).generate(0, 500)[1]
    assert np.median(1.0 - clean / 29903.0) < 0.03


def test_quick_pairwise_score_acgt_and_valid_equals_oracle():
    """the one-pair entry point of the reference's header (src/fastaseq.h:74, src/fastaseq.c:585-596) against the oracle's
    restatement: every truncation point, every character of the alphabet on either side"""
    rng = np.random.default_rng(3)
    alpha = np.frombuffer(b"ACGTNMRWSYKVHDB-X?O.acgtn", dtype=np.uint8)
    for n in (1, 7, 64, 333):
        a = alpha[rng.integers(0, len(alpha), size=n)].tobytes()
        b = alpha[rng.integers(0, len(alpha), size=n)].tobytes()
        idx = sorted(rng.choice(n, size=max(1, n * 2 // 3), replace=False).tolist())
        for maxdist in (0, 1, 2, 5, 2 ** 31 - 1):
            assert H.score_acgt_and_valid(a, b, idx, maxdist) == O.score_acgt(a, b, maxdist, idx), (n, maxdist)
