"""world_size-2/3/4 CPU test of the reference-shard driver (uvaia_amd/refshard.py) over gloo.

Every rank scans its pieces of the stream against ALL queries, the rows of each query shard move by all_to_all_single, and every
rank replays its queries over all pieces in stream order; the union of the ranks' heaps must equal one process (the oracle's
src/nearest.c loop) with all queries.  The engine stand-in is the oracle: its "scan" fills the counter buffers with a pattern that
names (query, reference), its "replay" checks that the rows it was handed are exactly those of its queries and of the piece it is
asked to replay -- so the exchange (splits, offsets, double buffering) is checked byte for byte -- and then lets the oracle process
the piece for that query range with the exchanged snapshot."""
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


def _dataset(n_refs, gappy_queries, n_query=40):
    refs, root, cols = F.synth_alignment(n_refs, 1200, seed=61, p_snp=0.006)
    qs, _, _ = F.synth_alignment(n_query, 1200, seed=62, root=root, poly_cols=cols, p_snp=0.006)
    if gappy_queries:      # every column is invalid in some query: idx_c empty, batches cannot influence anything
        qs = [bytearray(s) for s in qs]
        for i, s in enumerate(qs[:8]):
            a = i * 1200 // 8
            s[a:a + 1200 // 8 + 1] = b"N" * len(s[a:a + 1200 // 8 + 1])
        qs = [bytes(s) for s in qs]
    return refs, qs


def _pattern(q, pos):
    return (q * 1000003 + pos * 7 + 11) & 0x7FFFFFFF


class RefShardOracleEngine:
    """Oracle-backed stand-in for the engine's reference-shard calls (tests only)."""

    def __init__(self, query, refs, nbest, max_pool, rank, world, piece):
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from ring_oracle_engine import OracleRingEngine
        self.inner = OracleRingEngine(query, refs, nbest, max_pool)
        self.L, self.s, self.q, self.refs = self.inner.L, self.inner.s, query, refs
        self.rank, self.world, self.piece = rank, world, piece
        self.a0, self.a1 = 0, query.ntax
        self.snap = query.nchar                                  # cq->max_incompatible before the first batch (src/nearest.c:375)
        self.rows = (query.ntax + 31) // 32 * 32
        self.scanned, self.replayed = [], []

    def shard_rows(self):
        return self.rows

    @staticmethod
    def _view(ptr, n_ints):
        return np.ctypeslib.as_array((C.c_int32 * n_ints).from_address(ptr))

    def shard_aux_bytes(self, n_tiles):
        return n_tiles * 64 * 4                                  # one int per reference (the GPU engine: valid sites, + 16 bytes of pre-score)

    def shard_scan(self, first, n, cnt_ptr, tmin_ptr, aux_ptr):
        assert (first // self.piece) % self.world == self.rank, "scans only its own pieces"
        assert first // self.piece == (first + n - 1) // self.piece
        t0 = first // 64
        tiles = (first + n + 63) // 64 - t0
        cnt = self._view(cnt_ptr, self.rows * tiles * 64).reshape(self.rows, tiles * 64)          # one dword per pair
        tmin = self._view(tmin_ptr, self.rows * tiles * 2).reshape(self.rows, tiles, 2)
        pos = t0 * 64 + np.arange(tiles * 64, dtype=np.int64)
        for q in range(self.rows):
            cnt[q, :] = (q * 1000003 + pos * 7 + 11) & 0x7FFFFFFF
            tmin[q, :, 0] = q
            tmin[q, :, 1] = t0 + np.arange(tiles)
        self._view(aux_ptr, tiles * 64)[:] = (pos * 13 + 5 + 1000003 * self.rank) & 0x7FFFFFFF   # names (reference, scanning rank)
        self.scanned.append((first, n))

    def scan_wait(self):
        pass

    def replay_wait(self):
        pass

    def set_active_queries(self, q0, q1):
        self.a0, self.a1 = q0, q1

    def max_tolerance(self):
        return max(self.L.orc_search_final_T(self.s, iq) for iq in range(self.a0, self.a1))

    def set_snapshot(self, v):
        self.snap = v

    def shard_replay(self, cnt_ptr, tmin_ptr, aux_ptr, owner, first, n, ordinal0, q0, q1):
        t0 = first // 64
        tiles = (first + n + 63) // 64 - t0
        nq = q1 - q0
        assert owner == (first // self.piece) % self.world, "the piece's scanning rank"
        pos_ = t0 * 64 + np.arange(tiles * 64, dtype=np.int64)
        assert np.array_equal(self._view(aux_ptr, tiles * 64), ((pos_ * 13 + 5 + 1000003 * owner) & 0x7FFFFFFF).astype(np.int32)), "aux block of another piece or rank"
        cnt = self._view(cnt_ptr, nq * tiles * 64).reshape(nq, tiles * 64)
        tmin = self._view(tmin_ptr, nq * tiles * 2).reshape(nq, tiles, 2)
        pos = t0 * 64 + np.arange(tiles * 64, dtype=np.int64)
        for k in range(nq):
            assert np.array_equal(cnt[k, :], (((q0 + k) * 1000003 + pos * 7 + 11) & 0x7FFFFFFF).astype(np.int32)), "rows of another query or piece"
            assert (tmin[k, :, 0] == q0 + k).all()
            assert np.array_equal(tmin[k, :, 1], t0 + np.arange(tiles))
        if self.replayed:
            assert first == self.replayed[-1][0] + self.replayed[-1][1], "pieces are replayed in stream order, none skipped"
        else:
            assert first == 0
        self.replayed.append((first, n))
        seqs = self.refs[first:first + n]
        ords = (C.c_int64 * n)(*range(ordinal0, ordinal0 + n))
        names = O._cstr_array(["r%d" % o for o in range(ordinal0, ordinal0 + n)])
        assert self.L.orc_search_process_slice_range(self.s, n, O._cstr_array(seqs), names, ords, self.snap, q0, q1) == 0

    def result(self):
        return self.inner.result()


def _worker(rank, world, port, n_refs, pool, piece, nbest, acgt, gappy, n_query, out_dir):
    import torch.distributed as dist
    from uvaia_amd import refshard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    refs, qs = _dataset(n_refs, gappy, n_query)
    q = O.Query(qs, ["q%d" % i for i in range(len(qs))], acgt=acgt, ambig_q=1.0)
    plan = refshard.Plan(world, rank, -(-n_refs // world), q.ntax, pool=pool, piece=piece)
    plan.total = n_refs                                           # the test's stream is not a multiple of the world size
    eng = RefShardOracleEngine(q, refs, nbest, max(pool, piece) + 64, rank, world, piece)
    xchg = refshard.TorchExchange(dist, plan, eng, "cpu")
    refshard.run(eng, plan, xchg, len(q.idx_c) > 0)
    mine = [p for a, b in plan.pools(len(q.idx_c) > 0) for p in plan.pieces_of_pool(a, b) if p.owner == rank]
    assert eng.scanned == [(p.first, p.n) for p in mine]          # every own piece scanned exactly once, nothing else
    if plan.q1 > plan.q0:
        assert sum(n for _, n in eng.replayed) == n_refs
        rows, T = eng.result()
        rows, T = rows[plan.q0:plan.q1], T[plan.q0:plan.q1]
        np.save(os.path.join(out_dir, "rows_%d.npy" % rank), np.array([[list(s) + [o] for s, o in r] for r in rows], dtype=object), allow_pickle=True)
        np.save(os.path.join(out_dir, "T_%d.npy" % rank), np.array(T))
    dist.barrier()
    dist.destroy_process_group()


def test_plan_covers_the_stream_once():
    from uvaia_amd import refshard
    for world in (1, 2, 3, 8):
        for refs_per_rank, pool in ((100000, None), (1000, 192), (77, 64)):
            plan = refshard.Plan(world, 0, refs_per_rank, 1000, pool=pool, piece=(64 if refs_per_rank < 5000 else None))
            for cons in (False, True):
                at = 0
                for a, b in plan.pools(cons):
                    for p in plan.pieces_of_pool(a, b):
                        assert p.first == at and p.n >= 1 and p.owner == (p.first // plan.piece) % world
                        assert p.first // plan.piece == (p.first + p.n - 1) // plan.piece       # inside one piece of the map
                        at += p.n
                assert at == plan.total
    assert refshard.query_shard(1000, 3, 8) == (384, 512)
    assert refshard.Plan(8, 0, 100000, 1000).piece == 33344


@pytest.mark.parametrize("world,n_query", [(2, 40), (3, 40), (4, 70)])
@pytest.mark.parametrize("acgt,gappy,pool,piece", [(False, False, 128, 64), (True, False, 100, 64), (False, True, 64, 64), (False, False, 300, 128)])
def test_reference_shards_over_gloo_equal_single_process(tmp_path, world, n_query, acgt, gappy, pool, piece):
    """pool 100 with pieces of 64: batch boundaries cut pieces in the middle of a tile (only matters when the query set has complete
    constant columns, which it has unless `gappy`)"""
    import torch.multiprocessing as mp
    from uvaia_amd import refshard
    nbest, n_refs = 6, 470
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_refs, pool, piece, nbest, acgt, gappy, n_query, str(tmp_path)), nprocs=world, join=True)
    refs, qs = _dataset(n_refs, gappy, n_query)
    q = O.Query(qs, ["q%d" % i for i in range(len(qs))], acgt=acgt, ambig_q=1.0)
    assert (len(q.idx_c) > 0) == (not gappy)
    gold = O.search(q, refs, ["r%d" % i for i in range(len(refs))], pool=pool, nbest=nbest, ambig_r=1.0)
    seen = 0
    for rank in range(world):
        q0, q1 = refshard.query_shard(q.ntax, rank, world)
        if q1 <= q0:
            continue
        rows = np.load(tmp_path / ("rows_%d.npy" % rank), allow_pickle=True)
        T = np.load(tmp_path / ("T_%d.npy" % rank))
        for k, iq in enumerate(range(q0, q1)):
            assert [list(r) for r in rows[k]] == [list(s) + [o] for o, _, s in gold.rows[iq]]
        assert list(T) == gold.final_T[q0:q1]
        seen += q1 - q0
    assert seen == q.ntax
