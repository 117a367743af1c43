"""world_size-2 (and 3) CPU test of the query-shard driver (uvaia_amd/shards.py) over gloo: every rank holds the whole reference
stream and the heaps of a contiguous range of the queries; the union of the ranks' heaps == one process with all queries.  With
constant-and-complete columns in the query set the batch snapshot is all-reduced pool by pool."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

import fixtures as F
import oracle_lib as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _dataset(n_refs, gappy_queries):
    refs, root, cols = F.synth_alignment(n_refs, 1200, seed=51, p_snp=0.006)
    qs, _, _ = F.synth_alignment(150, 1200, seed=52, root=root, poly_cols=cols, p_snp=0.006)
    if gappy_queries:      # every column is invalid in some query: idx_c empty, the snapshot cannot influence anything
        qs = [bytearray(s) for s in qs]
        for i, s in enumerate(qs[:8]):
            a = i * 1200 // 8
            s[a:a + 1200 // 8 + 1] = b"N" * len(s[a:a + 1200 // 8 + 1])
        qs = [bytes(s) for s in qs]
    return refs, qs


class ShardOracleEngine:
    """Oracle-backed stand-in for the engine calls the shard driver makes (tests only)."""

    def __init__(self, query, refs, nbest, max_pool):
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from ring_oracle_engine import OracleRingEngine
        self.inner = OracleRingEngine(query, refs, nbest, max_pool)      # sets up the ctypes signatures and the search object
        self.L, self.s, self.q, self.refs = self.inner.L, self.inner.s, query, refs
        self.q0, self.q1 = 0, query.ntax

    def set_active_queries(self, q0, q1):
        assert q0 % 64 == 0 and 0 <= q0 < q1 <= self.q.ntax
        self.q0, self.q1 = q0, q1

    def max_tolerance(self):
        return max(self.L.orc_search_final_T(self.s, iq) for iq in range(self.q0, self.q1))

    def entered_flags(self, clear=False):
        return None

    def search_resident_pool(self, first, n, ordinal0, snapshot=-1):
        snap = self.max_tolerance() if snapshot < 0 else snapshot
        seqs = self.refs[first:first + n]
        ords = (C.c_int64 * n)(*range(ordinal0, ordinal0 + n))
        names = O._cstr_array(["r%d" % o for o in range(ordinal0, ordinal0 + n)])
        assert self.L.orc_search_process_slice_range(self.s, n, O._cstr_array(seqs), names, ords, snap, self.q0, self.q1) == 0

    def search_resident(self, pool, want_entered=False):
        for a in range(0, len(self.refs), pool):
            self.search_resident_pool(a, min(pool, len(self.refs) - a), a)

    def result(self):
        rows, T = self.inner.result()
        return rows[self.q0:self.q1], T[self.q0:self.q1]


def _worker(rank, world, port, n_refs, pool, nbest, acgt, gappy, out_dir):
    import torch
    import torch.distributed as dist
    from uvaia_amd import shards
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    refs, qs = _dataset(n_refs, gappy)
    q = O.Query(qs, ["q%d" % i for i in range(len(qs))], acgt=acgt, ambig_q=1.0)
    q0, q1 = shards.query_shard(q.ntax, rank, world)
    eng = ShardOracleEngine(q, refs, nbest, pool)
    shards.run_query_shard(eng, q0, q1, len(refs), pool, len(q.idx_c) > 0, shards.TorchMax(dist, torch.device("cpu")))
    if q1 > q0:
        rows, T = eng.result()
        np.save(os.path.join(out_dir, "rows_%d.npy" % rank), np.array([[list(s) + [o] for s, o in r] for r in rows], dtype=object), allow_pickle=True)
        np.save(os.path.join(out_dir, "T_%d.npy" % rank), np.array(T))
    dist.barrier()
    dist.destroy_process_group()


def test_query_shard_boundaries():
    from uvaia_amd import shards
    for nq in (1, 15, 16, 17, 40, 1000, 10000):
        for world in (1, 2, 3, 8):
            cuts = [shards.query_shard(nq, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and max(c[1] for c in cuts) == nq
            for (a0, a1), (b0, b1) in zip(cuts, cuts[1:]):
                assert a1 == b0 or (b0 == b1 == nq)              # contiguous; trailing ranks may be empty
            assert all(c[0] % 64 == 0 or c[0] == nq for c in cuts)
    assert shards.query_shard(1000, 3, 8) == (384, 512)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("acgt,gappy", [(False, False), (True, False), (False, True)])
def test_query_shards_over_gloo_equal_single_process(tmp_path, world, acgt, gappy):
    import torch.multiprocessing as mp
    from uvaia_amd import shards
    nbest, n_refs, pool = 6, 150, 32
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_refs, pool, nbest, acgt, gappy, str(tmp_path)), nprocs=world, join=True)
    refs, qs = _dataset(n_refs, gappy)
    q = O.Query(qs, ["q%d" % i for i in range(len(qs))], acgt=acgt, ambig_q=1.0)
    assert (len(q.idx_c) > 0) == (not gappy)
    gold = O.search(q, refs, ["r%d" % i for i in range(len(refs))], pool=pool, nbest=nbest, ambig_r=1.0)
    for rank in range(world):
        q0, q1 = shards.query_shard(q.ntax, rank, world)
        if q1 <= q0:
            continue
        rows = np.load(tmp_path / ("rows_%d.npy" % rank), allow_pickle=True)
        T = np.load(tmp_path / ("T_%d.npy" % rank))
        for k, iq in enumerate(range(q0, q1)):
            assert [list(r) for r in rows[k]] == [list(s) + [o] for o, _, s in gold.rows[iq]]
        assert list(T) == gold.final_T[q0:q1]
