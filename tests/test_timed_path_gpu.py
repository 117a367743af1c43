"""The path bench.py times, checked in the regime it is timed in (`pytest -m gpu`).

bench.py's step is uvaia_gpu_db_rederive (own stream) + uvaia_gpu_search_resident over an HBM-resident database with the default
switches: at 1 000 queries that means the column-compressed scan with rare columns, pools cut into sub-slices that run several
counter buffers ahead of the replay, and a replay that does not cache the query row in LDS.  The tests below send generator data
of BASELINE config[1] / config[2] shape through exactly that sequence of calls and compare with the CPU oracle (heaps, tolerances,
dump flags: src/nearest.c:288-306,479-510), then check at the full config[1] size what does not need the oracle to finish:
the three ways of running the same search agree, and the scores the heaps hold are the oracle's pair scores.
"""
import numpy as np
import pytest

import oracle_lib as O
import timed_path_workloads as W
from uvaia_amd import capi, hostlib

pytestmark = pytest.mark.gpu

QUERY_INDEX0 = 1 << 40          # as bench.py: queries and references come from disjoint sequence numbers of one generator


def _names(n, p="r"):
    return ["%s%d" % (p, i) for i in range(n)]


def _want(gold, ntax):
    return [[(tuple(s), o) for o, _, s in gold.rows[iq]] for iq in range(ntax)]


def _load(eng, gen, first, n, chunk=8192):
    eng.db_reserve(n)
    for a in range(0, n, chunk):
        m = min(chunk, n - a)
        rows, non_n = gen.generate(first + a, m)
        eng.db_append_block(rows, non_n)


def _timed_step(eng, pool):
    """one bench.py step, with the dump flags read back"""
    eng.reset()
    eng.db_rederive()
    ent = eng.search_resident(pool)
    eng.sync()
    n, T, sc, od = eng.drain()
    return capi.finalise_heaps(n, sc, od), list(T), ent


@pytest.fixture(scope="module")
def config1_sample():
    """1 000 generator queries x 8 000 generator references x 29 903 columns, k = 100, one pool (pool 65 536 > 8 000); the oracle's
    answer is compared through its committed digest (tests/timed_path_workloads.py: the oracle itself runs only to explain a difference)."""
    w = W.Workload("config1_sample")
    return w.gen, w.qs, w.qn, w.refs, w, w


def test_config1_regime_default_switches_equal_oracle(config1_sample):
    """(a) db_append -> db_rederive -> search_resident with nothing forced: scan3_kernel with rare columns, replay without the LDS
    query row (>= 256 queries), the rebuild of the derived planes overlapping the scan."""
    gen, qs, qn, refs, oq, gold = config1_sample
    pq = hostlib.PreparedQuery(qs, qn)
    assert pq.ntax == oq.ntax == 1000
    with pq.open_engine(nbest=100, max_pool=8000) as eng:
        assert eng.scan_variant() == 2
        _load(eng, gen, 0, len(refs))
        for _ in range(2):                                   # a second step over the same resident database gives the same answer
            rows, T, ent = _timed_step(eng, 8000)
            gold.check(rows, T, np.nonzero(ent)[0])
        admitted, demanded, dense = eng.replay_stats(reset=True)
        assert admitted > 100 * 1000                         # heaps fill and keep turning over: the regime of the benchmark


def test_config1_regime_sub_slices_equal_oracle(config1_sample):
    """(a) the same search cut into five sub-slices that are not tile aligned (the 100 000-reference benchmark run cuts its pools
    into three): scans run ahead of the replay in the ring of counter buffers, which wraps."""
    gen, qs, qn, refs, oq, gold = config1_sample
    pq = hostlib.PreparedQuery(qs, qn)
    with pq.open_engine(nbest=100, max_pool=8000, tuning={"subslice_refs": 1700}) as eng:
        _load(eng, gen, 0, len(refs))
        rows, T, ent = _timed_step(eng, 8000)
        gold.check(rows, T, np.nonzero(ent)[0])


def test_config1_regime_pipelined_replay_equals_oracle(config1_sample):
    """tuning.pipeline = 2: one long scan launch per merged slice, its replay launched with it and following the scan's progress counters
    stripe by stripe (write-through counters, agent-scope loads); tuning.head_scan = 2: the first 128 references through the four-counter
    scan and its replay.  Same heaps, tolerances and dump flags."""
    gen, qs, qn, refs, oq, gold = config1_sample
    pq = hostlib.PreparedQuery(qs, qn)
    with pq.open_engine(nbest=100, max_pool=65536, tuning={"pipeline": 2, "head_scan": 2, "subslice_refs": 1600}) as eng:
        _load(eng, gen, 0, len(refs))
        for _ in range(2):
            rows, T, ent = _timed_step(eng, 65536)
            gold.check(rows, T, np.nonzero(ent)[0])


def test_config1_regime_streaming_push_equals_oracle(config1_sample):
    """the reference-shaped boundary call (uvaia_gpu_push, one pool) on the same data"""
    gen, qs, qn, refs, oq, gold = config1_sample
    pq = hostlib.PreparedQuery(qs, qn)
    with pq.open_engine(nbest=100, max_pool=8000) as eng:
        ent = eng.push(refs)
        n, T, sc, od = eng.drain()
        gold.check(capi.finalise_heaps(n, sc, od), T, np.nonzero(ent)[0])


def test_group_of_four_contexts_at_benchmark_shape(config1_sample):
    """reference shards inside one process (uvaia_gpu_group_*), four contexts on one card, on the same data and against the same oracle run:
    the regime of the 8-GPU run (63 query tiles per scan, rare columns, 250 queries per replaying context)"""
    gen, qs, qn, refs, oq, gold = config1_sample
    pq = hostlib.PreparedQuery(qs, qn)
    with capi.Group(pq, [0, 0, 0, 0], nbest=100, max_pool=4096, piece_refs=1024) as g:
        g.db_reserve(len(refs))
        for a in range(0, len(refs), 4000):
            g.db_append(refs[a:a + 4000])
        g.reset()
        g.db_rederive()
        ent = g.search_resident(8000)
        n, T, sc, od = g.drain()
    gold.check(capi.finalise_heaps(n, sc, od), T, np.nonzero(ent)[0])


def test_config2_regime_acgt_many_query_tiles_equal_oracle():
    """(b) --acgt with 2 048 queries (128 query tiles, the many-tile regime of BASELINE config[2]) x 3 000 references, k = 100,
    through db_append -> db_rederive -> search_resident with the default switches."""
    gen = hostlib.Synth(29903, seed=20241008, preset=0)
    qs, _ = gen.generate_bytes(QUERY_INDEX0, 2048)
    qn = _names(2048, "query_")
    refs, _ = gen.generate_bytes(0, 3000)
    oq = O.Query(qs, qn, acgt=True)
    gold = O.search(oq, refs, _names(len(refs)), pool=3000, nbest=100, ambig_r=0.5)
    pq = hostlib.PreparedQuery(qs, qn, acgt=True)
    assert pq.ntax == oq.ntax
    with pq.open_engine(nbest=100, max_pool=3000) as eng:
        assert eng.scan_variant() == 2
        _load(eng, gen, 0, len(refs))
        rows, T, ent = _timed_step(eng, 3000)
        assert rows == _want(gold, oq.ntax) and T == gold.final_T
        assert list(np.nonzero(ent)[0]) == list(gold.saved)


@pytest.fixture(scope="module")
def config2_sample():
    """10 000 generator queries x 1 536 generator references x 29 903 columns, --acgt, k = 100, one pool; the oracle's answer as above."""
    w = W.Workload("config2_sample")
    return w.gen, w.qs, w.qn, w.refs, w, w


def test_group_of_eight_contexts_at_config3_shape(config2_sample):
    """BASELINE config[3]'s regime on one card: 10 000 generator queries, --acgt, k = 100, EIGHT contexts (1 250 queries per replaying
    member: not a multiple of a super-tile of 64, nor of a query tile of 16 -> shards of 1 264 and a last one of 1 152), pieces of 128
    references so that the 1 536 references make two stripes (the second one short: four pieces), 157 super-tiles per scan, the rare-column
    cap clamped at 64.  Heaps, tolerances and dump flags against the oracle's single loop."""
    gen, qs, qn, refs, oq, gold = config2_sample
    nq, n_ref = len(qs), len(refs)
    pq = hostlib.PreparedQuery(qs, qn, acgt=True)
    with capi.Group(pq, [0] * 8, nbest=100, max_pool=512, piece_refs=128) as g:
        shards = [g.query_shard(i) for i in range(8)]
        assert shards[0] == (0, 1264) and shards[-1][1] == nq and all(a1 == b0 for (_, a1), (b0, _) in zip(shards, shards[1:]))
        g.db_reserve(n_ref)
        for a in range(0, n_ref, 1000):
            g.db_append(refs[a:a + 1000])
        g.reset()
        g.db_rederive()
        ent = g.search_resident(n_ref)
        n, T, sc, od = g.drain()
    gold.check(capi.finalise_heaps(n, sc, od), T, np.nonzero(ent)[0])


def test_config2_regime_10000_queries_acgt_equal_oracle(config2_sample):
    """(b') BASELINE config[2]'s own query count: 10 000 generator queries (157 super-tiles of 64, rare-column cap clamped at 64,
    several hundred dense polymorphic columns) x 1 536 references, --acgt, k = 100, through db_append -> db_rederive ->
    search_resident with the default switches; heaps, tolerances and dump flags against the oracle (src/nearest.c:442-477)."""
    gen, qs, qn, refs, oq, gold = config2_sample
    pq = hostlib.PreparedQuery(qs, qn, acgt=True)
    assert pq.ntax == oq.ntax == 10000
    with pq.open_engine(nbest=100, max_pool=1536) as eng:
        assert eng.scan_variant() == 2
        _load(eng, gen, 0, len(refs))
        for _ in range(2):
            rows, T, ent = _timed_step(eng, 1536)
            gold.check(rows, T, np.nonzero(ent)[0])
    # the same queries, sub-slices that are not tile aligned and wrap the ring of counter buffers
    with pq.open_engine(nbest=100, max_pool=1536, tuning={"subslice_refs": 500}) as eng:
        _load(eng, gen, 0, len(refs))
        rows, T, ent = _timed_step(eng, 1536)
        gold.check(rows, T, np.nonzero(ent)[0])


def _heap_pairs(rows):
    """{ordinal: [(query, scores)]} of everything the heaps hold"""
    by_ref = {}
    for iq, r in enumerate(rows):
        for s, o in r:
            by_ref.setdefault(o, []).append((iq, s))
    return by_ref


@pytest.mark.parametrize("acgt,n_ref", [(False, 100000)])
def test_full_size_config1_three_ways_agree_and_scores_are_the_oracles(acgt, n_ref):
    """(c) 1 000 queries x 100 000 references (BASELINE config[1]; --acgt at full size is config[2]'s test below: the suite has to stay
    well inside eight minutes), pool 65 536: the timed step (rederive on its own streams overlapping the sub-slice scans), the same step with
    every launch serialised, and the streaming push path must leave identical heaps, tolerances and dump flags; and the six scores
    of heap entries are the oracle's untruncated pair scores (checked for every entry that refers to one of 192 sampled references)."""
    pool = 65536 if n_ref > 65536 else 32768
    gen = hostlib.Synth(29903, seed=20241008, preset=0)
    qs, _ = gen.generate_bytes(QUERY_INDEX0, 1000)
    qn = _names(1000, "query_")
    pq = hostlib.PreparedQuery(qs, qn, acgt=acgt)
    with pq.open_engine(nbest=100, max_pool=pool) as eng:
        _load(eng, gen, 0, n_ref)
        rows, T, ent = _timed_step(eng, pool)
        rows2, T2, ent2 = _timed_step(eng, pool)
        assert rows2 == rows and T2 == T and np.array_equal(ent, ent2)
    with pq.open_engine(nbest=100, max_pool=pool, tuning={"serial": 1}) as eng:
        _load(eng, gen, 0, n_ref)
        rows_s, T_s, ent_s = _timed_step(eng, pool)
    assert rows_s == rows and T_s == T and np.array_equal(ent_s, ent)
    with pq.open_engine(nbest=100, max_pool=pool) as eng:              # streaming: two pools of raw characters
        ent_p = []
        for a in range(0, n_ref, pool):
            m = min(pool, n_ref - a)
            rows_b, non_n = gen.generate(a, m)
            ent_p.append(eng.push([rows_b[i].tobytes() for i in range(m)], non_n=non_n, ordinal0=a))
            del rows_b
        n, Tp, sc, od = eng.drain()
        assert capi.finalise_heaps(n, sc, od) == rows and list(Tp) == T
        assert np.array_equal(np.concatenate(ent_p), ent)
    # every heap is full, tolerances are those of the worst kept entries (src/nearest.c:506-508)
    assert all(len(r) == 100 for r in rows)
    # the heaps hold the oracle's scores
    by_ref = _heap_pairs(rows)
    assert set(by_ref) <= set(np.nonzero(ent)[0].tolist())             # whatever is kept at the end was dumped
    rng = np.random.default_rng(5)
    sample = sorted(rng.choice(sorted(by_ref), size=min(192, len(by_ref)), replace=False).tolist())
    seqs = [gen.generate(o, 1)[0][0].tobytes() for o in sample]
    oq = O.Query(qs, qn, acgt=acgt)
    want = oq.allpairs(seqs)
    checked = 0
    for j, o in enumerate(sample):
        for iq, s in by_ref[o]:
            assert tuple(int(x) for x in want[j, iq]) == s, "reference %d, query %d" % (o, iq)
            checked += 1
    assert checked >= len(sample)


def test_full_size_config2_three_ways_agree_and_scores_are_the_oracles():
    """(c') BASELINE config[2] at its full size: 10 000 queries x 1 000 000 references, --acgt, k = 100, pool 65 536 (what
    bench.py's sweep times).  The oracle cannot finish that, so at the full size: the timed step (rederive on its own streams
    overlapping the sub-slice scans) leaves every heap full, what is kept was dumped, and the six scores of heap entries are the
    oracle's untruncated pair scores (every entry that refers to one of 160 sampled references).  The three ways of running a
    search -- the timed step, the same step with every launch serialised, the streaming push path -- are compared on the first
    132 000 references of the same stream (heaps, tolerances, dump flags identical): three pools, sub-slices that wrap the ring of
    counter buffers, a seventh of the generating and loading (the suite's budget)."""
    n_ref, n_three, pool, nq = 1000000, 132000, 65536, 10000
    gen = hostlib.Synth(29903, seed=20241008, preset=0)
    qs, _ = gen.generate_bytes(QUERY_INDEX0, nq)
    qn = _names(nq, "query_")
    pq = hostlib.PreparedQuery(qs, qn, acgt=True)
    with pq.open_engine(nbest=100, max_pool=pool) as eng:
        _load(eng, gen, 0, n_ref)
        rows, T, ent = _timed_step(eng, pool)
    with pq.open_engine(nbest=100, max_pool=pool) as eng:
        _load(eng, gen, 0, n_three)
        rows_t, T_t, ent_t = _timed_step(eng, pool)
    with pq.open_engine(nbest=100, max_pool=pool, tuning={"serial": 1}) as eng:
        _load(eng, gen, 0, n_three)
        rows_s, T_s, ent_s = _timed_step(eng, pool)
    assert rows_s == rows_t and T_s == T_t and np.array_equal(ent_s, ent_t)
    del rows_s, ent_s
    with pq.open_engine(nbest=100, max_pool=pool) as eng:              # streaming: three pools of raw characters
        ent_p = []
        for a in range(0, n_three, pool):
            m = min(pool, n_three - a)
            parts = []
            for b in range(a, a + m, 8192):
                rows_b, non_n = gen.generate(b, min(8192, a + m - b))
                parts.append(([rows_b[i].tobytes() for i in range(len(rows_b))], non_n))
                del rows_b
            ent_p.append(eng.push([s for p_ in parts for s in p_[0]], non_n=np.concatenate([p_[1] for p_ in parts]), ordinal0=a))
            del parts
        n, Tp, sc, od = eng.drain()
        assert capi.finalise_heaps(n, sc, od) == rows_t and list(Tp) == T_t
        assert np.array_equal(np.concatenate(ent_p), ent_t)
    del rows_t, ent_t
    assert all(len(r) == 100 for r in rows)
    by_ref = _heap_pairs(rows)
    assert set(by_ref) <= set(np.nonzero(ent)[0].tolist())
    rng = np.random.default_rng(6)
    sample = sorted(rng.choice(sorted(by_ref), size=min(160, len(by_ref)), replace=False).tolist())
    seqs = [gen.generate(o, 1)[0][0].tobytes() for o in sample]
    oq = O.Query(qs, qn, acgt=True)
    want = oq.allpairs(seqs)
    checked = 0
    for j, o in enumerate(sample):
        for iq, s in by_ref[o]:
            assert tuple(int(x) for x in want[j, iq]) == s, "reference %d, query %d" % (o, iq)
            checked += 1
    assert checked >= len(sample)


@pytest.mark.parametrize("nq", [100, 40])
def test_slice_longer_than_the_pool_with_a_partial_super_tile(nq):
    """`uvaia --packed -p 8192` on 100 queries x 100 000 references: without constant-and-complete query columns pools have no effect,
    the slices are laid over the whole stream and the counter buffers grow past nq_pad x max_pool.  Their row count has to cover the
    whole last super-tile of 64 queries the scan writes (100 queries: rows 0..127), not the last tile of 16.  (The 40-query set
    has constant-and-complete columns: there the pools stay and the pre-score counters are in play.)"""
    gen = hostlib.Synth(29903, seed=20241008, preset=0)
    qs, _ = gen.generate_bytes(QUERY_INDEX0, nq)
    qn = _names(nq, "query_")
    refs, _ = gen.generate_bytes(0, 2500)
    oq = O.Query(qs, qn)
    gold = O.search(oq, refs, _names(len(refs)), pool=128, nbest=20, ambig_r=0.5)
    pq = hostlib.PreparedQuery(qs, qn)
    with pq.open_engine(nbest=20, max_pool=128) as eng:
        _load(eng, gen, 0, len(refs))
        for _ in range(2):
            rows, T, ent = _timed_step(eng, 128)
            assert rows == _want(gold, oq.ntax) and T == gold.final_T
            assert list(np.nonzero(ent)[0]) == list(gold.saved)
