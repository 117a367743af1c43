"""Query preprocessing on the device: the O(Q^2) pair test of exclude_redundant_query_sequences
(src/fastaseq.c:797-841) as uvaia_gpu_agree_on_polymorphic, and the host walk over its matrix, against the host-only loop and
the oracle."""
import numpy as np
import pytest

import fixtures as F
import oracle_lib as O
from uvaia_amd import capi, hostlib as H

pytestmark = pytest.mark.gpu

INVALID = np.frombuffer(b"NX-?O.", dtype=np.uint8)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _redundant_set(n_base, nchar, seed):
    base, root, cols = F.synth_alignment(n_base, nchar, seed=seed, p_snp=0.004)
    rng = np.random.default_rng(seed + 1)
    qs = []
    for s in base:
        qs.append(s)
        t = bytearray(s)
        a, ln = int(rng.integers(0, nchar - 100)), int(rng.integers(1, 90))
        t[a:a + ln] = b"N" * ln
        qs.append(bytes(t))                # less resolved copy
        if rng.random() < 0.5:
            qs.append(s)                   # exact duplicate
    return qs, ["q%d" % i for i in range(len(qs))]


def _agree_reference(q, seqs, acgt):
    """numpy restatement of quick_pairwise_score_truncated_idx_indelcheck / quick_pairwise_score_acgt == 0 (maxdist 1)."""
    a = np.frombuffer(b"".join(seqs), dtype=np.uint8).reshape(len(seqs), -1)[:, q.idx]
    b = np.frombuffer(b"".join(q.seqs), dtype=np.uint8).reshape(q.ntax, -1)[:, q.idx]
    ok_a = np.isin(a, ACGT) if acgt else ~np.isin(a, INVALID)
    ok_b = np.isin(b, ACGT) if acgt else ~np.isin(b, INVALID)
    out = np.zeros((len(seqs), q.ntax), dtype=np.uint8)
    for i in range(len(seqs)):
        out[i] = ~np.any(ok_a[i][None, :] & ok_b & (a[i][None, :] != b), axis=1)
    return out


@pytest.mark.parametrize("acgt", [False, True])
@pytest.mark.parametrize("trim", [0, 150])
def test_agreement_matrix(acgt, trim):
    qs, names = _redundant_set(30, 2300, seed=5)
    q = O.Query(qs, names, acgt=acgt, trim=trim)
    other, _, _ = F.synth_alignment(70, 2300, seed=77)
    with capi.Engine.from_query(q, nbest=1, max_pool=256) as eng:
        for batch in (q.seqs, other, q.seqs[:65]):
            got = eng.agree_on_polymorphic(batch)
            assert np.array_equal(got, _agree_reference(q, batch, acgt))
    assert got[:, :65].diagonal().all()            # a sequence agrees with itself


@pytest.mark.parametrize("acgt", [False, True])
@pytest.mark.parametrize("keep,ball", [(True, False), (False, True), (True, True)])
def test_pruning_on_device_equals_host_and_oracle(acgt, keep, ball):
    qs, names = _redundant_set(40, 1800, seed=8)
    try:
        H.set_prune_mode("host")
        host = H.PreparedQuery(qs, names, acgt=acgt, keep_resolved=keep, is_ball=ball, dist=3)
        H.set_prune_mode("device")
        dev = H.PreparedQuery(qs, names, acgt=acgt, keep_resolved=keep, is_ball=ball, dist=3)
    finally:
        H.set_prune_mode("auto")
    gold = O.Query(qs, names, acgt=acgt, keep_resolved=keep, is_ball=ball, dist=3)
    assert dev.names == host.names == gold.names and dev.seqs == gold.seqs
    assert dev.ntax < len(qs)
    for f in ("idx_c", "idx_m", "idx"):
        assert np.array_equal(getattr(dev, f), getattr(gold, f)), f


def test_large_query_sets_prune_on_the_device_by_default():
    """From 512 queries on the pair matrix comes from the device without any switch; batches of 2 048 sequences."""
    qs, names = _redundant_set(260, 900, seed=21)
    assert len(qs) >= 512
    dev = H.PreparedQuery(qs, names, keep_resolved=True)
    gold = O.Query(qs, names, keep_resolved=True)
    assert dev.names == gold.names and dev.ntax < len(qs)


@pytest.mark.parametrize("acgt", [False, True])
@pytest.mark.parametrize("trim", [0, 150])
def test_column_classes_on_the_device_equal_oracle(acgt, trim):
    """uvaia_gpu_query_columns = the walk of create_query_indices (src/fastaseq.c:732-777): consensus characters and the three index
    classes; several batches of rows (the batch is capped by the staging buffer) and a last partial group of 64"""
    qs, _, _ = F.synth_alignment(333, 2300, seed=31, p_amb=0.003)
    q = O.Query(qs, ["q%d" % i for i in range(len(qs))], acgt=acgt, trim=trim)
    cons, miss = capi.query_columns(q.seqs, trim=q.trim, acgt=acgt)
    assert cons == q.consensus
    cols = np.arange(q.nchar)
    inside = (cols >= q.trim) & (cols < q.nchar - q.trim)
    c = np.frombuffer(cons, dtype=np.uint8)
    assert np.array_equal(cols[inside & (c == ord("#"))], q.idx)
    assert np.array_equal(cols[inside & (c != ord("#")) & (c != ord("N")) & (miss == 1)], q.idx_m)
    assert np.array_equal(cols[inside & (c != ord("#")) & (c != ord("N")) & (miss == 0)], q.idx_c)


def test_many_queries_take_the_device_walk_by_default():
    """from 2 048 queries on create_query_indices runs its column walk on the device without any switch (30 kb rows: several staging
    batches); same query structure as the oracle's"""
    gen = H.Synth(29903, seed=20241008, preset=0)
    qs, _ = gen.generate_bytes(1 << 40, 2500)
    names = ["q%d" % i for i in range(len(qs))]
    dev = H.PreparedQuery(qs, names)
    gold = O.Query(qs, names)
    assert dev.names == gold.names and dev.consensus == gold.consensus
    for f in ("idx_c", "idx_m", "idx"):
        assert np.array_equal(getattr(dev, f), getattr(gold, f)), f


TABLES = ["query plane words", "recoded planes", "ambiguity-word lists", "column classes", "rare-column mask", "compressed polymorphic planes",
          "planes on the rare columns", "item streams", "stream directory", "column split", "counts"]


def _tables(q, **tuning):
    with capi.Engine.from_query(q, nbest=3, max_pool=256, tuning=tuning) as eng:
        return [eng.query_table(w) for w in range(len(TABLES))]


@pytest.mark.parametrize("acgt", [False, True])
@pytest.mark.parametrize("n_query,nchar,trim,more", [
    (3, 333, 0, {}), (70, 2300, 0, {}), (70, 2300, 150, {"scan": "compressed"}), (130, 4097, 7, {"rare_max": 2}), (200, 1500, 0, {"rare_max": -1}),
    (257, 3000, 0, {"scan_tiles_per_wave": 1, "scan_waves_per_block": 4}), (600, 2500, 30, {"scan_tiles_per_wave": 4})])
def test_query_tables_built_on_the_device_equal_those_of_the_host_threads(acgt, n_query, nchar, trim, more):
    """SURVEY 8f rank 2, second half: everything the scans read about the query set -- plane words, recoded planes and ambiguity-word
    lists, column classes, rare columns, compressed planes, the item stream of every super-tile and its directory -- built by kernels
    from the raw rows (uvaia_gpu_tuning.query_tables = 2, the default) is byte for byte what the host threads build (= 1)."""
    qs, _, _ = F.synth_alignment(n_query, nchar, seed=1000 + n_query, p_snp=0.004, p_amb=0.002, n_run_frac=0.3)
    q = O.Query(qs, ["q%d" % i for i in range(len(qs))], acgt=acgt, trim=trim, ambig_q=1.0, keep_resolved=True)
    host = _tables(q, query_tables=1, **more)
    dev = _tables(q, query_tables=2, **more)
    for name, a, b in zip(TABLES, host, dev):
        assert a.shape == b.shape, name
        assert np.array_equal(a, b), (name, int(np.flatnonzero(a != b)[0]))
    assert len(host[0]) > 0 and len(host[7]) > 0


def test_query_tables_on_the_device_at_genome_length_and_a_bad_byte():
    from uvaia_amd import hostlib
    gen = hostlib.Synth(29903, seed=20241008, preset=0)
    qs, _ = gen.generate_bytes(1 << 40, 700)
    for acgt in (False, True):
        q = O.Query(qs, ["q%d" % i for i in range(len(qs))], acgt=acgt, ambig_q=1.0, keep_resolved=True)
        host, dev = _tables(q, query_tables=1), _tables(q, query_tables=2)
        for name, a, b in zip(TABLES, host, dev):
            assert np.array_equal(a, b), name
    # a byte outside the alphabet: the same refusal, naming the first such query and byte
    bad = list(q.seqs)
    bad[5] = bad[5][:100] + b"!" + bad[5][101:]
    bad[3] = bad[3][:200] + b"#" + bad[3][201:]
    msgs = []
    for how in (1, 2):
        with pytest.raises(capi.GpuError) as e:
            capi.Engine(bad, q.consensus, q.idx_c, q.idx_m, q.idx, trim=q.trim, acgt=q.acgt, nbest=3, max_pool=256, tuning={"query_tables": how})
        msgs.append(str(e.value))
    assert msgs[0] == msgs[1] and "query 3" in msgs[0] and "0x23" in msgs[0]
