"""world_size-2 (and 3) CPU test of the multi-rank ring protocol (uvaia_amd/ring.py) over gloo: block-cyclic slices,
state handed rank to rank, final heaps on the last rank == one process scanning the same stream with pool = world x slice."""
import os
import socket
import sys

import numpy as np
import pytest

import fixtures as F
import oracle_lib as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, slice_size, per_rank, nbest, acgt, out_dir, grouped=False, gappy_queries=False):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from ring_oracle_engine import NumpyStateBuffer, OracleRingEngine
    from uvaia_amd import ring
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    refs, qs = _dataset(world * per_rank, gappy_queries)
    q = O.Query(qs, ["q%d" % i for i in range(len(qs))], acgt=acgt, ambig_q=1.0)
    slices = ring.block_cyclic_layout(per_rank, slice_size, rank, world)
    local = []
    for sl in slices:                                    # this rank's resident shard, in local order
        local += refs[sl.ordinal0:sl.ordinal0 + sl.n]
    eng = OracleRingEngine(q, local, nbest, slice_size)
    if grouped:
        final = ring.run_ring_grouped(eng, ring.TorchRingComm(dist, rank, world, cuda=False), rank, world, slices, q.ntax,
                                      len(q.idx_c) > 0, lambda nbytes: NumpyStateBuffer(nbytes))
    else:
        final = ring.run_ring(eng, ring.TorchComm(dist, cuda=False), rank, world, slices, lambda: NumpyStateBuffer(eng.state_bytes()))
    if final:
        rows, T = eng.result()
        np.save(os.path.join(out_dir, "rows.npy"), np.array([[list(s) + [o] for s, o in r] for r in rows], dtype=object), allow_pickle=True)
        np.save(os.path.join(out_dir, "T.npy"), np.array(T))
    dist.barrier()
    dist.destroy_process_group()


def _dataset(n_refs, gappy_queries=False):
    refs, root, cols = F.synth_alignment(n_refs, 1200, seed=41, p_snp=0.006)
    qs, _, _ = F.synth_alignment(7, 1200, seed=42, root=root, poly_cols=cols, p_snp=0.006)
    if gappy_queries:      # every column is invalid in some query: no constant-and-complete column (idx_c empty), snapshots are moot
        qs = [bytearray(s) for s in qs]
        for i, s in enumerate(qs):
            a = i * 1200 // 7
            s[a:a + 1200 // 7 + 1] = b"N" * len(s[a:a + 1200 // 7 + 1])
        qs = [bytes(s) for s in qs]
    return refs, qs


@pytest.mark.parametrize("world,slice_size,per_rank", [(2, 16, 80), (2, 25, 60), (3, 10, 45)])
@pytest.mark.parametrize("acgt", [False, True])
def test_ring_over_gloo_equals_single_process(tmp_path, world, slice_size, per_rank, acgt):
    import torch.multiprocessing as mp
    nbest = 6
    port = _free_port()
    mp.spawn(_worker, args=(world, port, slice_size, per_rank, nbest, acgt, str(tmp_path)), nprocs=world, join=True)
    refs, qs = _dataset(world * per_rank)
    q = O.Query(qs, ["q%d" % i for i in range(len(qs))], acgt=acgt)
    # the stream the ring represents: stripe s = slice s of rank 0, 1, ...; with equal slices that is simply refs[] in order,
    # except that the last (partial) stripe is made of the partial slices
    gold = O.search(q, refs, ["r%d" % i for i in range(len(refs))], pool=world * slice_size, nbest=nbest, ambig_r=1.0)
    rows = np.load(tmp_path / "rows.npy", allow_pickle=True)
    T = np.load(tmp_path / "T.npy")
    for iq in range(q.ntax):
        want = [list(s) + [o] for o, _, s in gold.rows[iq]]
        assert [list(r) for r in rows[iq]] == want
    assert list(T) == gold.final_T


@pytest.mark.parametrize("world,slice_size,per_rank", [(2, 16, 80), (3, 10, 45), (4, 8, 24)])
@pytest.mark.parametrize("acgt,clean", [(False, False), (True, False), (False, True)])
def test_query_group_pipelined_ring_over_gloo(tmp_path, world, slice_size, per_rank, acgt, clean):
    clean = not clean      # third case = gappy queries (idx_c empty: rank 0 does not gather before opening a stripe)
    """The pipelined variant: the state travels as `world` per-query-group blobs, all receives pre-posted, wrap-around link on its
    own process group; with clean queries (idx_c not empty) rank 0 gathers every group before it opens a stripe."""
    import torch.multiprocessing as mp
    nbest = 6
    port = _free_port()
    mp.spawn(_worker, args=(world, port, slice_size, per_rank, nbest, acgt, str(tmp_path), True, not clean), nprocs=world, join=True)
    refs, qs = _dataset(world * per_rank, not clean)
    q = O.Query(qs, ["q%d" % i for i in range(len(qs))], acgt=acgt, ambig_q=1.0)
    assert (len(q.idx_c) > 0) == clean
    gold = O.search(q, refs, ["r%d" % i for i in range(len(refs))], pool=world * slice_size, nbest=nbest, ambig_r=1.0)
    rows = np.load(tmp_path / "rows.npy", allow_pickle=True)
    T = np.load(tmp_path / "T.npy")
    for iq in range(q.ntax):
        assert [list(r) for r in rows[iq]] == [list(s) + [o] for o, _, s in gold.rows[iq]]
    assert list(T) == gold.final_T
