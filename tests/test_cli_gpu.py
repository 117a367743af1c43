"""End-to-end tests of the drop-in command lines (bin/uvaia, bin/uvaiaball) and of the radius search, on a GPU box.
Outputs (CSV rows, dumped sequences and their order) must equal what the oracle's restatement of the reference's main loops
gives for the same files and options."""
import csv
import io
import lzma
import os
import subprocess

import numpy as np
import pytest

import fixtures as F
import oracle_lib as O
from uvaia_amd import capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UVAIA = os.path.join(ROOT, "bin", "uvaia")
UVAIABALL = os.path.join(ROOT, "bin", "uvaiaball")


def _write_fasta(path, names, seqs, opener=open, width=None):
    with opener(path, "wb") as fh:
        for n, s in zip(names, seqs):
            fh.write(b">" + n.encode() + b"\n")
            if width:
                for a in range(0, len(s), width):
                    fh.write(s[a:a + width] + b"\n")
            else:
                fh.write(s + b"\n")


def _read_xz_text(path):
    with lzma.open(path, "rt") as fh:
        return fh.read()


@pytest.fixture(scope="module")
def files(tmp_path_factory, bundled_db):
    d = tmp_path_factory.mktemp("cli")
    names, seqs = bundled_db
    by = dict(zip(names, seqs))
    qn = F.sample_names_1k()[:10]
    _write_fasta(d / "query.fa", qn, [by[n] for n in qn], width=70)          # multi-line FASTA
    _write_fasta(d / "ref1.aln.xz", names[:1200], seqs[:1200], opener=lzma.open)
    _write_fasta(d / "ref2.fa", names[1200:1500], seqs[1200:1500])
    return d, qn, [by[n] for n in qn], names[:1500], seqs[:1500]


@pytest.mark.parametrize("extra,kw", [
    ([], {}),
    (["--trim", "230", "-x"], {"trim": 230, "exclude_self": True}),
    (["-k", "-n", "1"], {"keep_resolved": True, "nbest": 1}),
    (["--devices", "0,0"], {}),                                                        # two contexts on one GPU: reference shards from the C host
    (["--acgt", "--trim", "230", "--devices", "0,0,0"], {"acgt": True, "trim": 230}),
])
def test_uvaia_cli_matches_oracle(files, extra, kw):
    d, qn, qs, rnames, rseqs = files
    nbest = kw.get("nbest", 5)
    out = str(d / ("out_" + "_".join(x.strip("-") for x in extra) if extra else "out_default"))
    cmd = [UVAIA, "-r", str(d / "ref1.aln.xz"), "-r", str(d / "ref2.fa"), str(d / "query.fa"), "-p", "64", "-o", out]
    if "nbest" not in kw:
        cmd += ["-n", str(nbest)]
    subprocess.run(cmd + extra, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    q = O.Query(qs, qn, trim=kw.get("trim", 0), acgt=kw.get("acgt", False), keep_resolved=kw.get("keep_resolved", False))
    gold = O.search(q, rseqs, rnames, pool=64, nbest=nbest, exclude_self=kw.get("exclude_self", False), file_breaks=(1200,))
    # table
    rows = list(csv.reader(io.StringIO(_read_xz_text(out + ".csv.xz"))))
    hdr = rows[0]
    assert hdr[:4] == ["query", "reference", "rank", "ACGT_matches"]
    assert hdr[4] == ("valid_ACGT_comparisons" if kw.get("acgt") else "text_matches")
    want = []
    for iq in range(q.ntax):
        for rank, (o, name, s) in enumerate(gold.rows[iq], 1):
            want.append([q.names[iq], name, str(rank)] + [str(v) for v in s])
    assert rows[1:] == want
    # dump: every reference that ever entered a heap, in stream order
    dump_names, dump_seqs = F.read_fasta_bytes(lzma.open(out + ".aln.xz", "rb").read())
    assert dump_names == [rnames[i] for i in gold.saved]
    assert dump_seqs == [rseqs[i] for i in gold.saved]


@pytest.mark.parametrize("acgt,trim,pool", [(False, 0, 64), (True, 230, 9185), (False, 230, 9185), (True, 0, 64)])
def test_uvaia_cli_on_the_whole_bundled_database_matches_the_committed_table(tmp_path, acgt, trim, pool):
    """BASELINE config[0]: `uvaia` on the reference's bundled alignment (9 185 sequences, xz) with the first 10 sample names as
    queries, --nbest 5: the table must equal tests/golden/config1_oracle_snapshot.json row for row (no oracle in this test)."""
    import json
    snap = json.load(open(os.path.join(ROOT, "tests", "golden", "config1_oracle_snapshot.json")))
    run = [r for r in snap["runs"] if (r["acgt"], r["trim"], r["pool"]) == (acgt, trim, pool)][0]
    names, seqs = F.load_bundled()
    by = dict(zip(names, seqs))
    qn = F.sample_names_1k()[:10]
    _write_fasta(tmp_path / "query.fa", qn, [by[n] for n in qn])
    out = str(tmp_path / "out")
    cmd = [UVAIA, "-r", os.path.join(ROOT, "tests", "golden", "03.unique_acgt.aln.xz"), str(tmp_path / "query.fa"),
           "-p", str(pool), "-n", "5", "--trim", str(trim), "-o", out] + (["--acgt"] if acgt else [])
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    rows = list(csv.reader(io.StringIO(_read_xz_text(out + ".csv.xz"))))[1:]
    want = [[qname, r[0], str(rank)] + [str(v) for v in r[1:]]
            for qname, qrows in zip(run["queries"], run["rows"]) for rank, r in enumerate(qrows, 1)]
    assert rows == want
    dump_names, _ = F.read_fasta_bytes(lzma.open(out + ".aln.xz", "rb").read())
    assert len(dump_names) == run["n_saved"]


@pytest.mark.parametrize("acgt", [False, True])
@pytest.mark.parametrize("dist", [0, 1, 5])
def test_ball_api_matches_oracle(bundled_db, acgt, dist):
    names, seqs = bundled_db
    by = dict(zip(names, seqs))
    qn = F.sample_names_1k()[:12]
    q = O.Query([by[n] for n in qn], qn, dist=dist, acgt=acgt, is_ball=True)
    refs = seqs[:900]
    md_want, _ = q.ball(refs, ambig_r=0.001)                 # ambig_r ~ 0: no reference is filtered before scoring
    with capi.Engine.from_query(q, nbest=2, max_pool=512) as eng:
        got = np.concatenate([eng.ball(refs[a:a + 512], q.dist + 1) for a in range(0, len(refs), 512)])
    assert np.array_equal(got, md_want)


def test_ball_synthetic_many_radii():
    refs, root, cols = F.synth_alignment(400, 2000, seed=61, p_snp=0.003)
    qs, _, _ = F.synth_alignment(15, 2000, seed=62, root=root, poly_cols=cols, p_snp=0.003)
    for acgt in (False, True):
        for dist in (0, 2, 7, 30):
            q = O.Query(qs, ["q%d" % i for i in range(len(qs))], dist=dist, acgt=acgt, is_ball=True)
            md_want, _ = q.ball(refs, ambig_r=0.001)
            with capi.Engine.from_query(q, nbest=2, max_pool=512) as eng:
                assert np.array_equal(eng.ball(refs, q.dist + 1), md_want), (acgt, dist)


@pytest.mark.parametrize("gather", [1, 2])
@pytest.mark.parametrize("acgt", [False, True])
def test_ball_over_the_resident_database_matches_oracle(acgt, gather):
    """uvaia_gpu_ball_resident on generator data at full genome length: radii from "nearly everything stops at the consensus" to
    "every reference goes on to the queries" (more of them than one scan batch holds), ranges that start inside a tile; the columns
    of query->idx gathered by a pass of its own (1) or by the consensus pass (2, the default)."""
    from uvaia_amd import hostlib
    gen = hostlib.Synth(29903, seed=20241008, preset=1)
    qs, _ = gen.generate_bytes(1 << 40, 24)
    refs, _ = gen.generate_bytes(0, 1500)
    refs = refs + qs[:6]                                   # some references identical to queries: distance 0
    for dist in (0, 3, 40, 4000):
        q = O.Query(qs, ["q%d" % i for i in range(len(qs))], dist=dist, acgt=acgt, is_ball=True)
        md, keep = q.ball(refs, ambig_r=0.001)                 # (uvaiaball keeps references with at least nchar * A valid sites: nothing is dropped)
        with capi.Engine.from_query(q, nbest=2, max_pool=512, tuning={"ball_gather": gather}) as eng:
            eng.db_reserve(len(refs))
            eng.db_append(refs)
            got = eng.ball_resident(q.dist + 1)
            assert np.array_equal(got, md), dist
            assert np.array_equal(eng.ball_resident(q.dist + 1, first=70, n=1000), md[70:1070])
            assert eng.ball_asked(reset=True) <= 2 * len(refs)          # two searches: at most every reference goes on to the queries
        assert keep.sum() == (md <= q.dist).sum()


@pytest.mark.parametrize("gather", [1, 2])
@pytest.mark.parametrize("acgt", [False, True])
def test_ball_with_the_most_diverse_columns_first(acgt, gather):
    """Enough polymorphic query columns (>= 512) that the gathered words start with the 256 most diverse ones and the scan's queries
    leave early: same cq->mindist as the oracle, for radii where few, many and all references go on to the queries."""
    from uvaia_amd import hostlib
    gen = hostlib.Synth(29903, seed=20241008, preset=0)
    qs, _ = gen.generate_bytes(1 << 40, 400)
    refs, _ = gen.generate_bytes(0, 1200)
    refs = refs + qs[:5]
    for dist in (2, 11, 600):
        q = O.Query(qs, ["q%d" % i for i in range(len(qs))], dist=dist, acgt=acgt, is_ball=True)
        assert len(q.idx) >= 512
        md, _ = q.ball(refs, ambig_r=0.001)
        with capi.Engine.from_query(q, nbest=2, max_pool=512, tuning={"ball_gather": gather}) as eng:
            eng.db_reserve(len(refs))
            eng.db_append(refs)
            assert np.array_equal(eng.ball_resident(q.dist + 1), md), dist
            assert np.array_equal(np.concatenate([eng.ball(refs[a:a + 500], q.dist + 1) for a in range(0, len(refs), 500)]), md), dist


@pytest.mark.parametrize("acgt", [False, True])
def test_seq_ball_against_query_structure_one_sequence_api(acgt):
    """the reference header's one-sequence entry point of the radius search (src/fastaseq.h:78, src/fastaseq.c:660-696), served by
    the GPU engine the host library keeps for the query set; a second query set replaces the engine"""
    from uvaia_amd import hostlib
    refs, root, cols = F.synth_alignment(60, 2000, seed=71, p_snp=0.003)
    for seed, dist in ((72, 2), (73, 9)):
        qs, _, _ = F.synth_alignment(9, 2000, seed=seed, root=root, poly_cols=cols, p_snp=0.003)
        names = ["q%d" % i for i in range(len(qs))]
        q = O.Query(qs, names, dist=dist, acgt=acgt, is_ball=True)
        md_want, _ = q.ball(refs, ambig_r=0.001)
        pq = hostlib.PreparedQuery(qs, names, dist=dist, acgt=acgt, is_ball=True)
        assert pq.ntax == q.ntax
        got = [pq.seq_ball(r, dist + 1) for r in refs]
        assert got == list(md_want), (acgt, dist)


def test_uvaiaball_cli_matches_oracle(files):
    d, qn, qs, rnames, rseqs = files
    out = str(d / "ball_out")
    subprocess.run([UVAIABALL, "-r", str(d / "ref1.aln.xz"), str(d / "query.fa"), "-d", "3", "-p", "100", "-o", out],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    q = O.Query(qs, qn, dist=3, is_ball=True)
    md, keep = q.ball(rseqs[:1200], ambig_r=0.5)
    dump_names, dump_seqs = F.read_fasta_bytes(lzma.open(out + ".aln.xz", "rb").read())
    assert dump_names == [rnames[i] for i in np.nonzero(keep)[0]]


UVAIAPACK = os.path.join(ROOT, "bin", "uvaiapack")


@pytest.mark.parametrize("extra", [[], ["--acgt"], ["--trim", "230", "-k"], ["-x"], ["-x", "--acgt", "-n", "3"], ["--devices", "0,0"], ["--acgt", "--devices", "0,0,0"]])
def test_packed_database_gives_the_same_files_as_the_text_path(files, extra):
    """`uvaiapack` + `uvaia --packed` against `uvaia -r` on the same references (two files, one xz, with gaps,
    ambiguity codes and sequences the -A filter drops): identical table and identical dump, byte for byte."""
    d, qn, qs, rnames, rseqs = files
    db = str(d / "refs.uvdb")
    if not os.path.exists(db):
        subprocess.run([UVAIAPACK, "-o", db, str(d / "ref1.aln.xz"), str(d / "ref2.fa")], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=600)
    tag = "_".join(x.strip("-") for x in extra) or "default"
    out_t, out_p = str(d / ("t_" + tag)), str(d / ("p_" + tag))
    common = [str(d / "query.fa"), "-p", "200", "-n", "5"] + extra
    subprocess.run([UVAIA, "-r", str(d / "ref1.aln.xz"), "-r", str(d / "ref2.fa"), "-o", out_t] + common, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    subprocess.run([UVAIA, "--packed", db, "-o", out_p] + common, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    assert _read_xz_text(out_p + ".csv.xz") == _read_xz_text(out_t + ".csv.xz")
    assert lzma.open(out_p + ".aln.xz", "rb").read() == lzma.open(out_t + ".aln.xz", "rb").read()
    assert len(_read_xz_text(out_p + ".csv.xz").splitlines()) > 10


def test_packed_database_refuses_what_it_cannot_honour(files):
    d = files[0]
    db = str(d / "refs.uvdb")
    if not os.path.exists(db):
        subprocess.run([UVAIAPACK, "-o", db, str(d / "ref1.aln.xz"), str(d / "ref2.fa")], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=600)
    for bad in (["-A", "0.3"],):
        r = subprocess.run([UVAIA, "--packed", db, str(d / "query.fa"), "-o", str(d / "refused")] + bad, stdout=subprocess.DEVNULL,
                           stderr=subprocess.PIPE, timeout=600)
        assert r.returncode != 0 and b"packed" in r.stderr
