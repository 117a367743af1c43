"""Reference shards on the GPU (`pytest -m gpu`): several contexts on ONE card run the protocol the driver's 8-GPU run uses --
each context derives and scans only its pieces of the stream against all queries, the rows of each query shard are copied to the
context that replays those queries (peer copies inside uvaia_gpu_group_*; all_to_all_single between processes), and the union of
the contexts' heaps must equal the oracle's single loop (src/nearest.c:245-330): heaps, tolerances and dump flags."""
import numpy as np
import pytest

import fixtures as F
import oracle_lib as O
from uvaia_amd import capi

pytestmark = pytest.mark.gpu


def _names(n, p="r"):
    return ["%s%d" % (p, i) for i in range(n)]


def _want(gold, ntax):
    return [[(tuple(s), o) for o, _, s in gold.rows[iq]] for iq in range(ntax)]


@pytest.fixture(scope="module")
def data():
    refs, root, cols = F.synth_alignment(1500, 2500, seed=21)
    qs, _, _ = F.synth_alignment(70, 2500, seed=22, root=root, poly_cols=cols)
    gappy = [bytearray(s) for s in qs]
    for i, s in enumerate(gappy[:10]):           # every column invalid in some query: no constant-and-complete column
        a = i * 250
        s[a:a + 251] = b"N" * len(s[a:a + 251])
    return refs, qs, [bytes(s) for s in gappy]


@pytest.mark.parametrize("world,piece,acgt,gappy,pool", [(w, p_, a, g_, pl) for w, p_ in ((2, 128), (3, 64), (4, 192)) for a, g_, pl in ((False, False, 512), (True, False, 300), (False, True, 256), (True, True, 1500))]
                         + [(8, 64, False, False, 512), (8, 64, True, True, 1500)])      # (8 members, 70 queries: three of them replay nothing)
def test_group_of_contexts_on_one_gpu_equals_oracle(data, world, piece, acgt, gappy, pool):
    refs, qs, qs_gappy = data
    q = O.Query(qs_gappy if gappy else qs, _names(70, "q"), acgt=acgt)
    assert (len(q.idx_c) > 0) == (not gappy)
    gold = O.search(q, refs, _names(len(refs)), pool=pool, nbest=12, ambig_r=1.0)
    with capi.Group(q, [0] * world, nbest=12, max_pool=max(pool, piece), piece_refs=piece) as g:
        assert [g.query_shard(i) for i in range(world)][0][0] == 0
        g.db_reserve(len(refs))
        g.db_append(refs[:700]); g.db_append(refs[700:])
        for _ in range(2):                        # a second search over the same resident database gives the same answer
            g.reset()
            g.db_rederive()
            ent = g.search_resident(pool)
            n, T, sc, od = g.drain()
            assert capi.finalise_heaps(n, sc, od) == _want(gold, q.ntax)
            assert list(T) == gold.final_T
            assert list(np.nonzero(ent)[0]) == list(gold.saved)


@pytest.mark.parametrize("acgt", [False, True])
def test_group_push_streams_batches_like_one_context(data, acgt):
    """the reference-shaped call: one pool of raw sequences per push"""
    refs, qs, _ = data
    q = O.Query(qs, _names(70, "q"), acgt=acgt)
    pool = 400
    gold = O.search(q, refs, _names(len(refs)), pool=pool, nbest=7, ambig_r=1.0)
    with capi.Group(q, [0, 0, 0], nbest=7, max_pool=pool, piece_refs=64) as g:
        ent = [g.push(refs[a:a + pool], ordinal0=a) for a in range(0, len(refs), pool)]
        n, T, sc, od = g.drain()
    assert capi.finalise_heaps(n, sc, od) == _want(gold, q.ntax) and list(T) == gold.final_T
    assert list(np.nonzero(np.concatenate(ent))[0]) == list(gold.saved)


# (the groups at the benchmark's shapes -- 1 000 queries x 4 contexts, 10 000 queries --acgt x 8 contexts: BASELINE config[3]'s regime --
# live in tests/test_timed_path_gpu.py, next to the single-context tests whose oracle results they share)


def test_a_group_of_one_is_a_plain_context(data):
    refs, qs, _ = data
    q = O.Query(qs, _names(70, "q"))
    gold = O.search(q, refs, _names(len(refs)), pool=500, nbest=5, ambig_r=1.0)
    with capi.Group(q, [0], nbest=5, max_pool=500) as g:
        g.db_reserve(len(refs)); g.db_append(refs)
        ent = g.search_resident(500)
        n, T, sc, od = g.drain()
    assert capi.finalise_heaps(n, sc, od) == _want(gold, q.ntax) and list(T) == gold.final_T
    assert list(np.nonzero(ent)[0]) == list(gold.saved)


def test_tolerance_range_may_start_anywhere_under_reference_shards(data):
    """uvaia_amd/refshard.py asks every rank for the largest tolerance of ITS queries (the batch snapshot, src/nearest.c:290-291) by
    narrowing the active range: 24 queries on two ranks give the range [16, 24).  A range that is scanned must start at a super-tile
    of 64 queries; under reference shards every scan covers all queries and the range only selects tolerances."""
    refs, qs, _ = data
    q = O.Query(qs[:24], _names(24, "q"))
    with capi.Engine.from_query(q, nbest=5, max_pool=256) as eng:
        with pytest.raises(capi.GpuError):
            eng.set_active_queries(16, 24)
    with capi.Engine.from_query(q, nbest=5, max_pool=256) as eng:
        eng.db_set_shard(1, 2, 128)
        eng.set_active_queries(16, 24)
        assert isinstance(eng.max_tolerance(), int)
        eng.set_active_queries(0, 24)


@pytest.mark.parametrize("flags", [["--queries", "200", "--refs", "8192", "--nbest", "20"],
                                   ["--queries", "2500", "--refs", "1536", "--mode", "acgt"]])       # config[3]'s regime per rank: 1 250 queries replayed by each of the two ranks (not a multiple of 64), --acgt, k = 100
def test_bench_two_ranks_over_gloo_on_one_card_end_to_end(flags):
    """`bench.py --gpus 2` as the driver launches it for N > 1 (it starts its own ranks when no launcher did), rehearsed on one card
    with the exchange over gloo (UVAIA_BENCH_BACKEND): the JSON line must be a two-rank line and the sample that went through the
    reference-shard protocol of the timed step must equal the oracle on every rank."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UVAIA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"] + flags,
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["parity_check_on_timed_path"] is True
    assert out["value"] > 0 and "reference shards" in out["multi_gpu"]
