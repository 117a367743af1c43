"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle, bit for bit.

Everything here needs a real MI355X: `pytest -m gpu`.
"""
import numpy as np
import pytest

import fixtures as F
import oracle_lib as O
from uvaia_amd import capi

pytestmark = pytest.mark.gpu


def _names(n, p="r"):
    return ["%s%d" % (p, i) for i in range(n)]


def _gpu_search(q, refs, pool, nbest, non_n=None, tuning=None):
    with capi.Engine.from_query(q, nbest=nbest, max_pool=pool, tuning=tuning) as eng:
        entered = []
        for a in range(0, len(refs), pool):
            nn = None if non_n is None else non_n[a:a + pool]
            entered.append(eng.push(refs[a:a + pool], non_n=nn))
        n, T, sc, od = eng.drain()
    return capi.finalise_heaps(n, sc, od), list(T), np.concatenate(entered) if entered else np.zeros(0, np.uint8)


def _assert_same_search(q, refs, pool, nbest, **kw):
    gold = O.search(q, refs, _names(len(refs)), pool=pool, nbest=nbest, ambig_r=1.0)   # ambig_r=1: no reference is filtered
    rows, T, entered = _gpu_search(q, refs, pool, nbest, **kw)
    for iq in range(q.ntax):
        want = [(tuple(s), o) for o, _, s in gold.rows[iq]]
        assert rows[iq] == want, "query %d (%s)" % (iq, q.names[iq])
    assert T == gold.final_T
    assert list(np.nonzero(entered)[0]) == list(gold.saved)


@pytest.fixture(scope="module")
def synth():
    refs, root, cols = F.synth_alignment(700, 2500, seed=11)
    qs, _, _ = F.synth_alignment(40, 2500, seed=12, root=root, poly_cols=cols)
    return refs, qs


@pytest.mark.parametrize("acgt", [False, True])
@pytest.mark.parametrize("trim", [0, 230])
def test_allpairs_counts_synthetic(synth, acgt, trim):
    refs, qs = synth
    q = O.Query(qs, _names(len(qs), "q"), trim=trim, acgt=acgt)
    want = q.allpairs(refs[:300])
    with capi.Engine.from_query(q, nbest=5, max_pool=512) as eng:
        eng.push(refs[:300])
        got = eng.last_batch_scores(300)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("acgt", [False, True])
def test_allpairs_counts_bundled(bundled_db, acgt):
    """16 queries x 512 references of the reference's own alignment, untruncated counts, trim 0 and 230."""
    names, seqs = bundled_db
    by = dict(zip(names, seqs))
    qn = F.sample_names_1k()[:16]
    for trim in (0, 230):
        q = O.Query([by[n] for n in qn], qn, trim=trim, acgt=acgt)
        want = q.allpairs(seqs[:512])
        with capi.Engine.from_query(q, nbest=5, max_pool=512) as eng:
            eng.push(seqs[:512])
            got = eng.last_batch_scores(512)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("acgt", [False, True])
@pytest.mark.parametrize("pool,nbest", [(64, 5), (100, 1), (700, 20), (33, 100)])
def test_search_matches_oracle_synthetic(synth, acgt, pool, nbest):
    refs, qs = synth
    q = O.Query(qs, _names(len(qs), "q"), acgt=acgt)
    _assert_same_search(q, refs, pool, nbest)


def test_search_host_supplied_non_n(synth):
    refs, qs = synth
    q = O.Query(qs, _names(len(qs), "q"))
    non_n = np.array([O.lib().orc_count_non_N(r, len(r)) for r in refs], dtype=np.int32)
    _assert_same_search(q, refs, 128, 10, non_n=non_n)


@pytest.mark.parametrize("acgt", [False, True])
def test_config1_bundled_db(bundled_db, acgt):
    """BASELINE config 1: bundled alignment as the database, the first 10 sample names as queries, --nbest 5, pool 64."""
    names, seqs = bundled_db
    by = dict(zip(names, seqs))
    qn = F.sample_names_1k()[:10]
    q = O.Query([by[n] for n in qn], qn, acgt=acgt)
    refs = seqs[:3000]
    gold = O.search(q, refs, names[:3000], pool=64, nbest=5)
    # the oracle drops low-quality references before they occupy a pool slot (src/nearest.c:263-270): feed the GPU the survivors
    keep = [i for i, r in enumerate(refs) if O.lib().orc_count_non_N(r, len(r)) >= int(len(r) * 0.5)]
    assert len(keep) == len(refs) - gold.n_lowqual
    kept = [refs[i] for i in keep]
    with capi.Engine.from_query(q, nbest=5, max_pool=64) as eng:
        ent = []
        for a in range(0, len(kept), 64):
            ent.append(eng.push(kept[a:a + 64], ordinal0=a))
        n, T, sc, od = eng.drain()
    rows = capi.finalise_heaps(n, sc, od)
    for iq in range(q.ntax):
        want = [(tuple(s), keep.index(o)) for o, _, s in gold.rows[iq]]
        assert rows[iq] == want
    assert list(T) == gold.final_T
    ent = np.concatenate(ent)
    assert [keep[i] for i in np.nonzero(ent)[0]] == list(gold.saved)


def test_resident_equals_streaming(synth):
    refs, qs = synth
    q = O.Query(qs, _names(len(qs), "q"))
    for pool in (64, 100, 333):
        rows_s, T_s, ent_s = _gpu_search(q, refs, pool, 7)
        with capi.Engine.from_query(q, nbest=7, max_pool=512) as eng:
            eng.db_reserve(len(refs))
            eng.db_append(refs[:130])          # appends need not be tile aligned
            eng.db_append(refs[130:])
            ent = eng.search_resident(pool)
            n, T, sc, od = eng.drain()
            assert capi.finalise_heaps(n, sc, od) == rows_s
            assert list(T) == T_s
            assert np.array_equal(ent, ent_s)
            eng.reset()                         # a second pass over the resident database gives the same answer
            eng.search_resident(pool, want_entered=False)
            n2, T2, sc2, od2 = eng.drain()
            assert capi.finalise_heaps(n2, sc2, od2) == rows_s


def test_packed_interchange_form_round_trip(synth):
    """a database exported in packed form (from a default-mode context) and imported again -- into a default
    and into an --acgt context, in two tile-aligned pieces -- gives the same heaps as the text path."""
    refs, qs = synth
    q4 = O.Query(qs, _names(len(qs), "q"))
    with capi.Engine.from_query(q4, nbest=7, max_pool=512) as eng:
        eng.db_append(refs)
        planes, non_n, side = eng.db_export()
    assert planes.shape[0] == (len(refs) + 63) // 64
    for acgt in (False, True):
        q = O.Query(qs, _names(len(qs), "q"), acgt=acgt)
        rows_s, T_s, ent_s = _gpu_search(q, refs, 100, 7)
        with capi.Engine.from_query(q, nbest=7, max_pool=512) as eng:
            eng.db_reserve(len(refs))
            eng.db_append_packed(planes[:2], non_n[:128], side[:128], 128)
            eng.db_append_packed(planes[2:], non_n[128:], side[128:], len(refs) - 128)
            assert eng.db_size() == len(refs)
            ent = eng.search_resident(100)
            n, T, sc, od = eng.drain()
            assert capi.finalise_heaps(n, sc, od) == rows_s
            assert list(T) == T_s
            assert np.array_equal(ent, ent_s)
            with pytest.raises(capi.GpuError):          # a partial tile cannot be followed by packed tiles
                eng.db_append_packed(planes[:1], non_n[:64], side[:64], 64)
    with capi.Engine.from_query(O.Query(qs, _names(len(qs), "q"), acgt=True), nbest=7, max_pool=512) as eng:
        eng.db_append(refs[:64])
        with pytest.raises(capi.GpuError):              # the interchange form is exported from a 4-plane context
            eng.db_export()


@pytest.mark.parametrize("acgt,gappy", [(False, False), (True, False), (False, True)])
def test_query_shards_equal_one_context(synth, acgt, gappy):
    """Query shards (uvaia_amd/shards.py): contexts opened with the WHOLE query set, each active on a range of it and holding the
    whole database, give the heaps of one context over all queries -- with the snapshot exchanged pool by pool when the query set
    has constant-and-complete columns.  Three shards on one GPU, driven in lockstep."""
    from uvaia_amd import shards
    refs, _ = synth
    _r, root, cols = F.synth_alignment(4, 2500, seed=11)
    qs, _, _ = F.synth_alignment(150, 2500, seed=12, root=root, poly_cols=cols)      # three shards of whole super-tiles (64 queries)
    qs = list(qs)
    if gappy:
        qs = [bytearray(s) for s in qs]
        for i, s in enumerate(qs[:10]):
            a = i * len(s) // 10
            s[a:a + len(s) // 10 + 1] = b"N" * len(s[a:a + len(s) // 10 + 1])
        qs = [bytes(s) for s in qs]
    q = O.Query(qs, _names(len(qs), "q"), acgt=acgt, ambig_q=1.0)
    cons = len(q.idx_c) > 0
    assert cons == (not gappy)
    pool, nbest, world = 100, 7, 3
    rows_1, T_1, ent_1 = _gpu_search(q, refs, pool, nbest)
    engines = [capi.Engine.from_query(q, nbest=nbest, max_pool=512) for _ in range(world)]
    try:
        cuts = [shards.query_shard(q.ntax, r, world) for r in range(world)]
        assert cuts == [(0, 64), (64, 128), (128, 150)]
        for e in engines:
            e.db_append(refs)
        if not cons:
            for e, (q0, q1) in zip(engines, cuts):
                shards.run_query_shard(e, q0, q1, len(refs), pool, cons)
        else:       # lockstep emulation of the all-reduce: what every rank would receive is the maximum over the ranks
            for e, (q0, q1) in zip(engines, cuts):
                e.set_active_queries(q0, q1)
                e.entered_flags(clear=True)
            for a in range(0, len(refs), pool):
                snap = max(e.max_tolerance() for e in engines)
                for e in engines:
                    e.search_resident_pool(a, min(pool, len(refs) - a), a, snap)
        ent = np.zeros(len(refs), dtype=np.uint8)
        for e, (q0, q1) in zip(engines, cuts):
            n, T, sc, od = e.drain()
            rows = capi.finalise_heaps(n, sc, od)
            assert rows[q0:q1] == rows_1[q0:q1]
            assert list(T)[q0:q1] == T_1[q0:q1]
            ent |= e.entered_flags()
            with pytest.raises(capi.GpuError):          # streamed batches act on the whole query set
                e.push(refs[:10])
        assert np.array_equal(ent, ent_1)               # a reference is dumped if it entered a heap on any rank
        with pytest.raises(capi.GpuError):
            engines[0].set_active_queries(16, 80)       # ranges start on a super-tile of 64 queries
    finally:
        for e in engines:
            e.close()


@pytest.mark.parametrize("rare_max", [1, 3, 40])
@pytest.mark.parametrize("acgt,trim", [(False, 0), (True, 0), (False, 230), (True, 230)])
def test_rare_column_path_matches_oracle(synth, acgt, trim, rare_max):
    """Columns where all but a few queries carry the same base are scanned as constant columns plus sparse items (and, with
    --acgt, dist_unique takes its rare-column part on demand).  The engine only does that for >= 64 queries; forced here on the
    40-query set, from 'singletons only' to 'every polymorphic column is rare'.  Streaming and resident paths, bundled data too."""
    tuning = {"rare_max": rare_max}
    refs, qs = synth
    q = O.Query(qs, _names(len(qs), "q"), acgt=acgt, trim=trim)
    _assert_same_search(q, refs, 100, 7, tuning=tuning)
    rows_s, T_s, ent_s = _gpu_search(q, refs, 100, 7, tuning=tuning)
    with capi.Engine.from_query(q, nbest=7, max_pool=512, tuning=tuning) as eng:
        eng.db_append(refs)
        ent = eng.search_resident(100)
        n, T, sc, od = eng.drain()
        assert capi.finalise_heaps(n, sc, od) == rows_s and list(T) == T_s and np.array_equal(ent, ent_s)


@pytest.mark.parametrize("acgt", [False, True])
def test_rare_column_path_on_bundled_alignment(bundled_db, acgt):
    names, seqs = bundled_db
    by = dict(zip(names, seqs))
    qn = F.sample_names_1k()[:24]
    q = O.Query([by[n] for n in qn], qn, acgt=acgt)
    _assert_same_search(q, seqs[:1500], 512, 5, tuning={"rare_max": 2, "scan": "compressed"})


@pytest.mark.parametrize("nchar,nq,nref,seed", [(777, 80, 500, 5), (129, 70, 300, 6), (2047, 130, 400, 7)])
@pytest.mark.parametrize("acgt", [False, True])
def test_odd_shapes_with_enough_queries_for_every_path(acgt, nchar, nq, nref, seed):
    """Alignment lengths that are not multiples of 32 or 128, query counts that are not multiples of 16 and large enough (>= 64)
    for the rare-column path to switch itself on: streaming with two pool sizes against the oracle."""
    refs, root, cols = F.synth_alignment(nref, nchar, seed=seed)
    qs, _, _ = F.synth_alignment(nq, nchar, seed=seed + 100, root=root, poly_cols=cols)
    q = O.Query(qs, _names(len(qs), "q"), acgt=acgt)
    _assert_same_search(q, refs, 97, 6)
    _assert_same_search(q, refs, nref, 3)


@pytest.mark.parametrize("acgt", [False, True])
@pytest.mark.parametrize("tuning", [None, {"scan_tiles_per_wave": 1}, {"scan_waves_per_block": 4}, {"scan_tiles_per_wave": 1, "scan_waves_per_block": 4}])
def test_rederive_is_idempotent_and_scan_shapes_agree(acgt, tuning):
    """Rebuilding the planes the scan reads on a resident database changes nothing; the block shapes of the column-compressed scan
    (one or two tiles of references per wave, four or eight waves per super-tile of queries) give the same heaps."""
    refs, root, cols = F.synth_alignment(700, 3001, seed=41)
    qs, _, _ = F.synth_alignment(90, 3001, seed=141, root=root, poly_cols=cols)
    q = O.Query(qs, _names(len(qs), "q"), acgt=acgt)
    with capi.Engine.from_query(q, nbest=6, max_pool=256, tuning=tuning) as eng:
        eng.db_reserve(len(refs))
        eng.db_append(refs[:333])
        eng.db_append(refs[333:])
        ent = eng.search_resident(200)
        n, T, sc, od = eng.drain()
        out = (capi.finalise_heaps(n, sc, od), list(T), ent.copy())
        eng.reset()
        eng.db_rederive()
        ent2 = eng.search_resident(200)
        n, T, sc, od = eng.drain()
        assert (capi.finalise_heaps(n, sc, od), list(T)) == out[:2] and np.array_equal(ent2, out[2])
    gold = O.search(q, refs, _names(len(refs)), pool=200, nbest=6, ambig_r=1.0)
    assert out[0] == [[(tuple(s_), o) for o, _, s_ in rows] for rows in gold.rows] and out[1] == gold.final_T


@pytest.mark.parametrize("acgt,nq", [(False, 5), (True, 16), (False, 32), (True, 32)])
def test_small_query_sets_on_either_scan(acgt, nq):
    """Up to 32 queries the engine scans the packed planes directly (two-counter kernels with tile bounds); the column-compressed
    scan can be forced and must give the same heaps.  Both against the oracle, streamed and resident."""
    refs, root, cols = F.synth_alignment(900, 2100, seed=51)
    qs, _, _ = F.synth_alignment(nq, 2100, seed=151, root=root, poly_cols=cols)
    q = O.Query(qs, _names(len(qs), "q"), acgt=acgt)
    with capi.Engine.from_query(q, nbest=4, max_pool=128) as eng:
        assert eng.scan_variant() == 0
    _assert_same_search(q, refs, 128, 4)
    with capi.Engine.from_query(q, nbest=4, max_pool=128, tuning={"scan": "compressed"}) as eng:
        assert eng.scan_variant() == 2
        eng.db_append(refs)
        ent = eng.search_resident(128)
        n, T, sc, od = eng.drain()
    gold = O.search(q, refs, _names(len(refs)), pool=128, nbest=4, ambig_r=1.0)
    assert capi.finalise_heaps(n, sc, od) == [[(tuple(s_), o) for o, _, s_ in rows] for rows in gold.rows]
    assert list(T) == gold.final_T and list(np.nonzero(ent)[0]) == list(gold.saved)
    _assert_same_search(q, refs, 128, 4, tuning={"scan": "compressed"})


def test_long_alignments_take_the_wide_counter_scan():
    """More than ~49 000 columns do not fit the 16-bit counter halves of the default scan: the engine switches to the
    four-counter scan by itself; results as the oracle's."""
    nchar = 50017
    refs, root, cols = F.synth_alignment(150, nchar, seed=31, p_snp=0.002)
    qs, _, _ = F.synth_alignment(5, nchar, seed=32, root=root, poly_cols=cols, p_snp=0.002)
    for acgt in (False, True):
        _assert_same_search(O.Query(qs, _names(len(qs), "q"), acgt=acgt), refs, 64, 4)


@pytest.mark.parametrize("acgt", [False, True])
def test_packed_plane_scan_with_many_query_tiles(synth, acgt):
    """tuning.scan = packed: the two-counter scan over the packed planes (what up to 32 queries get) on 40 queries = three query tiles"""
    refs, qs = synth
    _assert_same_search(O.Query(qs, _names(len(qs), "q"), acgt=acgt), refs[:400], 100, 6, tuning={"scan": "packed"})


def test_query_tile_sizes_agree(synth):
    refs, qs = synth
    q = O.Query(qs, _names(len(qs), "q"))
    want = q.allpairs(refs[:200])
    for qt in (8, 16, 32):
        with capi.Engine.from_query(q, nbest=5, max_pool=256) as eng:
            eng.set_query_tile(qt)
            eng.push(refs[:200])
            assert np.array_equal(eng.last_batch_scores(200), want)


def test_refuses_unknown_bytes(synth):
    refs, qs = synth
    q = O.Query(qs, _names(len(qs), "q"))
    bad = bytearray(refs[0]); bad[100] = ord("U")
    with capi.Engine.from_query(q, nbest=5, max_pool=64) as eng:
        with pytest.raises(capi.GpuError) as ei:
            eng.push([bytes(bad)])
        assert ei.value.code == -5
        eng.push(refs[:10])          # the context stays usable


def test_truncated_consensus_prescore_is_reproduced():
    """Cross-query coupling through cq->max_incompatible (src/nearest.c:290-291,431-432): a reference whose
    consensus pre-score was cut at the batch-start snapshot can still pass the gate of a query whose tolerance rose
    above the snapshot inside the batch; the truncated counters then enter the heap.  Built so that it happens."""
    L = 400
    A, C_, G, T = b"A", b"C", b"G", b"T"
    base = bytearray(A * L)
    qseq = bytes(base)                       # one clean query -> every column is idx_c
    q = O.Query([qseq], ["q0"])
    assert len(q.idx_c) == L

    def ref_with(mism_sites, n_sites=()):
        s = bytearray(base)
        for i in mism_sites: s[i] = ord("C")
        for i in n_sites: s[i] = ord("N")
        return bytes(s)

    # heap size 2, pool 2.  Batch 1 fills the heap with
    #   A: 10 mismatches, full length  -> 390 matches, m = 10
    #   B: 1 mismatch, 200 N's         -> 199 matches, m = 1     (root = worst = B, so T = 2)
    b1 = [ref_with(range(1, 11)), ref_with([20], n_sites=range(200, 400))]
    # Batch 2 starts with snapshot = max T = 2.
    #   r2: perfect full-length sequence: enters, evicts B; the new root is A (m = 10) -> T = 11 > snapshot.
    #   r3: 3 mismatches at sites 390, 395, 399: its pre-score was cut at the 2nd mismatch (site 395), so cq->res holds
    #       matches = 394, valid = 396; the gate (2 < 11) lets it through and the TRUNCATED scores 394/396 enter the heap
    #       (the untruncated ones would be 397/400).
    b2 = [ref_with([]), ref_with([390, 395, 399])]
    refs = b1 + b2
    gold = O.search(q, refs, _names(4), pool=2, nbest=2, ambig_r=1.0)
    got_scores = sorted(tuple(s) for _, _, s in gold.rows[0])
    assert (394, 394, 394, 396, 0, 400) in got_scores, got_scores      # the oracle shows the reference's quirk
    rows, Tq, ent = _gpu_search(q, refs, 2, 2)
    assert rows[0] == [(tuple(s), o) for o, _, s in gold.rows[0]]
    assert Tq == gold.final_T
    # with one big batch nothing is truncated in a way that matters and the true scores enter
    gold1 = O.search(q, refs, _names(4), pool=4, nbest=2, ambig_r=1.0)
    rows1, T1, _ = _gpu_search(q, refs, 4, 2)
    assert rows1[0] == [(tuple(s), o) for o, _, s in gold1.rows[0]]


@pytest.mark.parametrize("acgt", [False, True])
def test_random_small_many_states(acgt):
    """Many tiny problems with heavy ties and tiny heaps: exercises heap layout / tie handling / T dynamics."""
    rng = np.random.default_rng(5)
    for it in range(5):
        L = int(rng.integers(40, 200))
        refs, root, cols = F.synth_alignment(int(rng.integers(5, 150)), L, seed=100 + it, p_snp=0.02, p_amb=0.01)
        qs, _, _ = F.synth_alignment(int(rng.integers(1, 9)), L, seed=200 + it, root=root, poly_cols=cols, p_snp=0.02, p_amb=0.01)
        q = O.Query(qs, _names(len(qs), "q"), acgt=acgt, ambig_q=1.0)
        if q.ntax < 1:
            continue
        _assert_same_search(q, refs, int(rng.integers(1, 40)), int(rng.integers(1, 12)))


def test_heavily_ambiguous_sequences_use_the_dense_rescan():
    """Sequences with more partially ambiguous words than the per-sequence list holds: the replay must fall back to a dense
    rescan for the on-demand counters and still agree with the oracle (text/partial matches decide the order here)."""
    rng = np.random.default_rng(77)
    L = 1600
    refs, root, cols = F.synth_alignment(260, L, seed=31, p_snp=0.004, p_amb=0.03)      # ~48 ambiguous sites per sequence
    qs, _, _ = F.synth_alignment(9, L, seed=32, root=root, poly_cols=cols, p_snp=0.004, p_amb=0.03)
    qs += F.synth_alignment(3, L, seed=33, root=root, poly_cols=cols, p_snp=0.004, p_amb=0.0005)[0]   # and a few clean ones
    q = O.Query(qs, _names(len(qs), "q"), ambig_q=1.0)
    gold = O.search(q, refs, _names(len(refs)), pool=50, nbest=8, ambig_r=1.0)
    with capi.Engine.from_query(q, nbest=8, max_pool=50) as eng:
        for a in range(0, len(refs), 50):
            eng.push(refs[a:a + 50])
        n, T, sc, od = eng.drain()
        admitted, demanded, dense = eng.replay_stats()
    rows = capi.finalise_heaps(n, sc, od)
    for iq in range(q.ntax):
        assert rows[iq] == [(tuple(s), o) for o, _, s in gold.rows[iq]]
    assert list(T) == gold.final_T
    assert dense > 0 and demanded > 0 and admitted > 0


@pytest.mark.parametrize("acgt", [False, True])
def test_four_counter_path_still_agrees(synth, acgt):
    """tuning.scan = wide selects the four-counter scan + its replay (what alignments above 49 000 columns get)."""
    refs, qs = synth
    q = O.Query(qs, _names(len(qs), "q"), acgt=acgt)
    _assert_same_search(q, refs, 96, 9, tuning={"scan": "wide"})


@pytest.mark.parametrize("acgt", [False, True])
@pytest.mark.parametrize("world,slice_size,per_rank", [(2, 64, 256), (3, 50, 170)])
def test_ring_mode_with_several_contexts_on_one_gpu(synth, acgt, world, slice_size, per_rank):
    """Multi-GPU protocol (block-cyclic slices, state blob handed rank to rank) with one context per 'rank' on the same card;
    must equal a single process running the same stream with pool = world x slice."""
    from uvaia_amd import ring
    refs, qs = synth
    refs = refs[:world * per_rank]
    q = O.Query(qs, _names(len(qs), "q"), acgt=acgt)
    gold = O.search(q, refs, _names(len(refs)), pool=world * slice_size, nbest=9, ambig_r=1.0)
    layouts = [ring.block_cyclic_layout(per_rank, slice_size, r, world) for r in range(world)]
    engines = []
    try:
        for r in range(world):
            e = capi.Engine.from_query(q, nbest=9, max_pool=slice_size)
            local = []
            for sl in layouts[r]:
                local += refs[sl.ordinal0:sl.ordinal0 + sl.n]
            e.db_reserve(len(local)); e.db_append(local)
            engines.append(e)

        class HostBlob:
            def __init__(self, nbytes):
                self.arr = np.zeros(nbytes, dtype=np.uint8); self.ptr = self.arr.ctypes.data
        last = ring.run_ring_in_one_process(engines, layouts, lambda: HostBlob(engines[0].state_bytes()))
        n, T, sc, od = last.drain()
        first_result = (n.copy(), T.copy(), sc.copy(), od.copy())
        # and the query-group pipelined variant of the protocol gives the same final state
        for e in engines:
            e.reset()
        last = ring.run_ring_grouped_in_one_process(engines, layouts, q.ntax, len(q.idx_c) > 0, lambda nbytes: HostBlob(nbytes))
        n, T, sc, od = last.drain()
        for x, y in zip(first_result, (n, T, sc, od)):
            assert np.array_equal(x, y)
        rows = capi.finalise_heaps(n, sc, od)
        for iq in range(q.ntax):
            assert rows[iq] == [(tuple(s), o) for o, _, s in gold.rows[iq]]
        assert list(T) == gold.final_T
        entered = set()
        for r in range(world):
            flags = engines[r].entered_flags()
            pos = 0
            for sl in layouts[r]:
                entered |= {sl.ordinal0 + i for i in np.nonzero(flags[pos:pos + sl.n])[0]}
                pos += sl.n
        assert sorted(entered) == list(gold.saved)
    finally:
        for e in engines:
            e.close()


@pytest.mark.parametrize("nbest,n_refs", [(2500, 700), (600, 700)])
def test_large_heaps(synth, nbest, n_refs):
    """k = 2500 needs more than 64 KB of LDS per query heap (dynamic LDS attribute); k = 600 < n_refs fills and churns a big heap."""
    refs, qs = synth
    q = O.Query(qs[:6], _names(6, "q"))
    _assert_same_search(q, refs[:n_refs], 256, nbest)


def test_host_non_n_does_not_leak_into_counts(synth):
    """cq->non_n only feeds the last score column (src/nearest.c:501); handing the engine a different number must change
    that column and nothing else (the valid-pair counter uses the engine's own per-reference total)."""
    refs, qs = synth
    q = O.Query(qs, _names(len(qs), "q"))
    refs = refs[:200]

    def kept(non_n):
        with capi.Engine.from_query(q, nbest=300, max_pool=256) as eng:      # heap larger than the stream: everything is kept
            eng.push(refs, non_n=non_n)
            n, T, sc, od = eng.drain()
        return [{int(od[iq, s]): tuple(sc[iq, s]) for s in range(1, n[iq] + 1)} for iq in range(q.ntax)]

    base, alt = kept(None), kept(np.full(len(refs), 7, dtype=np.int32))
    for iq in range(q.ntax):
        assert base[iq].keys() == alt[iq].keys() and len(base[iq]) == len(refs)
        for o in base[iq]:
            assert base[iq][o][:5] == alt[iq][o][:5] and alt[iq][o][5] == 7


@pytest.mark.parametrize("acgt", [False, True])
def test_benchmark_shaped_data_push_and_resident(acgt):
    """Generator data at full genome length with k = 100: dozens of tiles per query, tolerances that rise and fall, and a
    resident search cut into sub-slices (scan several counter buffers ahead of the replay).  Both paths must equal the oracle."""
    from uvaia_amd import hostlib
    gen = hostlib.Synth(29903, seed=20241008, preset=0)
    qs, _ = gen.generate_bytes(1 << 40, 40)
    names = _names(40, "q")
    refs, non_n = gen.generate_bytes(0, 2600)
    oq = O.Query(qs, names, acgt=acgt)
    pq = hostlib.PreparedQuery(qs, names, acgt=acgt)
    gold = O.search(oq, refs, _names(len(refs)), pool=2600, nbest=100, ambig_r=0.5)
    want = [[(tuple(s), o) for o, _, s in gold.rows[iq]] for iq in range(oq.ntax)]
    with pq.open_engine(nbest=100, max_pool=2600) as e:
        ent = e.push(refs)
        n, T, sc, od = e.drain()
        assert capi.finalise_heaps(n, sc, od) == want and list(T) == gold.final_T
        assert list(np.nonzero(ent)[0]) == list(gold.saved)
    # 6 sub-slices, not tile aligned, ring of counter buffers wraps
    with pq.open_engine(nbest=100, max_pool=2600, tuning={"subslice_refs": 448}) as e:
        e.db_reserve(len(refs))
        e.db_append(refs[:1000]); e.db_append(refs[1000:])
        for _ in range(2):
            e.reset()
            ent = e.search_resident(2600)
            n, T, sc, od = e.drain()
            assert capi.finalise_heaps(n, sc, od) == want and list(T) == gold.final_T
            assert list(np.nonzero(ent)[0]) == list(gold.saved)


def _resident_search(q, refs, pool, nbest, tuning=None):
    with capi.Engine.from_query(q, nbest=nbest, max_pool=pool, tuning=tuning) as eng:
        eng.db_reserve(len(refs))
        eng.db_append(refs)
        out = []
        for _ in range(2):                                   # a second search over the same resident database: same answer
            eng.reset()
            ent = eng.search_resident(pool)
            n, T, sc, od = eng.drain()
            out.append((capi.finalise_heaps(n, sc, od), list(T), ent))
        assert out[0][0] == out[1][0] and out[0][1] == out[1][1] and np.array_equal(out[0][2], out[1][2])
    return out[0]


@pytest.mark.parametrize("acgt", [False, True])
def test_resident_search_of_few_queries_many_states(acgt):
    """A handful of queries over a resident database (the packed-plane scan, sub-slices that wrap the ring of counter buffers) must
    reproduce the oracle on many tiny problems with heavy ties, tiny heaps and tiny pools (snapshots taken often: the cut pre-scores
    of src/nearest.c:431-432)."""
    rng = np.random.default_rng(17)
    for it in range(4):
        L = int(rng.integers(40, 260))
        refs, root, cols = F.synth_alignment(int(rng.integers(5, 400)), L, seed=300 + it, p_snp=0.02, p_amb=0.01)
        qs, _, _ = F.synth_alignment(int(rng.integers(1, 9)), L, seed=400 + it, root=root, poly_cols=cols, p_snp=0.02, p_amb=0.01)
        q = O.Query(qs, _names(len(qs), "q"), acgt=acgt, ambig_q=1.0)
        if q.ntax < 1:
            continue
        pool, nbest = int(rng.integers(1, 70)), int(rng.integers(1, 12))
        gold = O.search(q, refs, _names(len(refs)), pool=pool, nbest=nbest, ambig_r=1.0)
        rows, T, ent = _resident_search(q, refs, pool, nbest, tuning={"subslice_refs": int(rng.integers(64, 200))})
        assert rows == [[(tuple(s_), o) for o, _, s_ in r] for r in gold.rows], (it, pool, nbest)
        assert T == gold.final_T and list(np.nonzero(ent)[0]) == list(gold.saved), (it, pool, nbest)


def test_resident_search_reproduces_the_truncated_prescore():
    """the quirk of test_truncated_consensus_prescore_is_reproduced through the resident path: the pre-score is cut at the batch
    snapshot exactly as cq->res would be, and a tolerance that rises above the snapshot inside the batch lets the cut scores in"""
    L = 400
    base = bytearray(b"A" * L)
    q = O.Query([bytes(base)], ["q0"])

    def ref_with(mism_sites, n_sites=()):
        s = bytearray(base)
        for i in mism_sites: s[i] = ord("C")
        for i in n_sites: s[i] = ord("N")
        return bytes(s)

    refs = [ref_with(range(1, 11)), ref_with([20], n_sites=range(200, 400)), ref_with([]), ref_with([390, 395, 399])]
    for pool in (2, 4):
        gold = O.search(q, refs, _names(4), pool=pool, nbest=2, ambig_r=1.0)
        rows, T, _ = _resident_search(q, refs, pool, 2)
        assert rows[0] == [(tuple(s), o) for o, _, s in gold.rows[0]], pool
        assert T == gold.final_T
    gold = O.search(q, refs, _names(4), pool=2, nbest=2, ambig_r=1.0)
    assert (394, 394, 394, 396, 0, 400) in [tuple(s) for _, _, s in gold.rows[0]]


@pytest.mark.parametrize("acgt", [False, True])
@pytest.mark.parametrize("nq", [3, 32, 70])
def test_resident_search_of_few_queries_at_genome_length(acgt, nq):
    """generator data at full length (N runs, ambiguity codes: the on-demand counters of the default mode), several sub-slices, pools
    that matter when the query set has constant-and-complete columns (3 queries) and when it has none; 70 queries forced onto the
    packed-plane scan"""
    from uvaia_amd import hostlib
    gen = hostlib.Synth(29903, seed=20241008, preset=0)
    qs, _ = gen.generate_bytes(1 << 40, nq)
    refs, _ = gen.generate_bytes(0, 3000)
    refs = refs + qs[:2]
    q = O.Query(qs, _names(nq, "q"), acgt=acgt)
    gold = O.search(q, refs, _names(len(refs)), pool=700, nbest=25, ambig_r=0.5)
    rows, T, ent = _resident_search(q, refs, 700, 25, tuning={"scan": "packed", "subslice_refs": 500})
    assert rows == [[(tuple(s_), o) for o, _, s_ in r] for r in gold.rows]
    assert T == gold.final_T and list(np.nonzero(ent)[0]) == list(gold.saved)


@pytest.mark.parametrize("acgt", [False, True])
@pytest.mark.parametrize("tiles_per_wave", [1, 2, 4])
def test_n_runs_at_every_word_offset_run_items(acgt, tiles_per_wave):
    """Run items of the column-compressed scan (a word item that also carries the byte mask of the group's words without any valid
    query character): 90 queries whose N runs start at every offset of a 128-column word group and cover none, one, two or three
    whole words beside a partial one, with gaps and ambiguity codes at the edges; references with their own N runs.  Heaps,
    tolerances and dump flags as the oracle's, streamed and resident."""
    nchar = 3000
    refs, root, cols = F.synth_alignment(500, nchar, seed=71, p_snp=0.004, n_run_frac=0.3)
    qs, _, _ = F.synth_alignment(90, nchar, seed=72, root=root, poly_cols=cols, p_snp=0.004, p_amb=0.001, n_run_frac=0.0)
    rng = np.random.default_rng(73)
    out = []
    for i, s in enumerate(qs):
        t = bytearray(s)
        for k in range(4):                                         # four runs per query, each in a group of its own
            g0 = 128 * (1 + 4 * k + (i % 4)) + 256 * (i // 30)
            a = g0 + (i * 7 + k * 13) % 128                        # every start offset over the queries
            ln = [1, 31, 32, 33, 64, 65, 96, 97, 127, 128, 129, 200][(i + k) % 12]
            t[a:a + ln] = b"N" * len(t[a:a + ln])
            if a > 0 and rng.random() < 0.5:
                t[a - 1] = ord("-") if rng.random() < 0.5 else ord("R")      # a gap or an ambiguity code right at the edge
        out.append(bytes(t))
    q = O.Query(out, _names(len(out), "q"), acgt=acgt, ambig_q=1.0, keep_resolved=True)
    assert q.ntax > 64
    _assert_same_search(q, refs, 256, 7, tuning={"scan": "compressed", "scan_tiles_per_wave": tiles_per_wave})
    rows, T, entered = _resident_search(q, refs, 256, 7, tuning={"scan": "compressed", "scan_tiles_per_wave": tiles_per_wave})
    gold = O.search(q, refs, _names(len(refs)), pool=256, nbest=7, ambig_r=1.0)
    assert rows == [[(tuple(s_), o) for o, _, s_ in r_] for r_ in gold.rows] and T == gold.final_T and list(np.nonzero(entered)[0]) == list(gold.saved)


@pytest.mark.parametrize("nq,trim", [(1, 0), (3, 70), (13, 70), (32, 0), (40, 0), (100, 70)])
def test_scan_side_extras_equal_the_on_demand_counters(nq, trim):
    """Up to 32 queries in default mode the packed-plane scan also leaves text - ACGT matches and partial - text matches of EVERY pair
    (scan2_extras) and the replay admits from registers (replay3_kernel); from 33 to 128 queries pair_extras_kernel and
    tile_bounds_kernel make the same arrays next to the column-compressed scan.  A mix of sequences: most list a few ambiguous words, some
    references and one query list more than the side row holds (their pairs stay 'unknown' and are counted on demand), some none.
    Streamed and resident, both settings of tuning.replay_extras, against the oracle; the keys after the first decide the order here."""
    L = 1900
    refs, root, cols = F.synth_alignment(420, L, seed=61, p_snp=0.003, p_amb=0.002)            # ~4 ambiguous sites per sequence
    refs += F.synth_alignment(60, L, seed=62, root=root, poly_cols=cols, p_snp=0.003, p_amb=0.03)[0]    # lists overflow
    refs += F.synth_alignment(60, L, seed=63, root=root, poly_cols=cols, p_snp=0.003, p_amb=0.0)[0]     # no ambiguity at all
    order = np.random.default_rng(9).permutation(len(refs))
    refs = [refs[i] for i in order]
    qs = F.synth_alignment(nq, L, seed=64, root=root, poly_cols=cols, p_snp=0.003, p_amb=0.002)[0]
    if nq >= 3:
        qs[1] = F.synth_alignment(1, L, seed=65, root=root, poly_cols=cols, p_snp=0.003, p_amb=0.03)[0][0]  # a query whose list overflows
        qs[2] = F.synth_alignment(1, L, seed=66, root=root, poly_cols=cols, p_snp=0.003, p_amb=0.0)[0][0]
    q = O.Query(qs, _names(len(qs), "q"), ambig_q=1.0, trim=trim)
    gold = O.search(q, refs, _names(len(refs)), pool=130, nbest=7, ambig_r=1.0)
    want = [[(tuple(s_), o) for o, _, s_ in r] for r in gold.rows]
    for extras in (2, 1):
        tuning = {"replay_extras": extras, "subslice_refs": 192}
        with capi.Engine.from_query(q, nbest=7, max_pool=130, tuning=tuning) as eng:
            assert eng.scan_variant() == (0 if nq <= 32 else 2)
            for a in range(0, len(refs), 130):
                eng.push(refs[a:a + 130])
            n, T, sc, od = eng.drain()
            admitted, demanded, dense = eng.replay_stats(reset=True)
            assert capi.finalise_heaps(n, sc, od) == want and list(T) == gold.final_T, extras
            if extras == 2:
                assert 0 < demanded < admitted        # only the pairs with an overflowed list went to memory
            else:
                assert demanded >= admitted > 0
        rows, T, ent = _resident_search(q, refs, 130, 7, tuning=tuning)
        assert rows == want and T == gold.final_T and list(np.nonzero(ent)[0]) == list(gold.saved), extras


@pytest.mark.parametrize("nbest,n_ref", [(3000, 3400), (4500, 4700)])
def test_large_heaps_with_a_handful_of_queries(nbest, n_ref):
    """Three queries in default mode with heaps of thousands of entries: the replay over the scan-side extras shrinks its staging buffers to
    make room for the heap in LDS (3 000 entries) or gives way to the on-demand replay (4 500 entries: nothing fits next to 144 KB of
    heap); either way the heaps fill, turn over, and equal the oracle's (heaps above 127 entries also take the level-by-level update)."""
    L = 300
    refs, root, cols = F.synth_alignment(n_ref, L, seed=71, p_snp=0.01, p_amb=0.002)
    qs = F.synth_alignment(3, L, seed=72, root=root, poly_cols=cols, p_snp=0.01, p_amb=0.002)[0]
    q = O.Query(qs, _names(3, "q"), ambig_q=1.0)
    gold = O.search(q, refs, _names(len(refs)), pool=1500, nbest=nbest, ambig_r=1.0)
    rows, T, ent = _resident_search(q, refs, 1500, nbest)
    assert rows == [[(tuple(s_), o) for o, _, s_ in r] for r in gold.rows] and T == gold.final_T and list(np.nonzero(ent)[0]) == list(gold.saved)
