"""Multi-GPU driver by QUERY shards: every rank keeps the whole reference database and the heaps of a contiguous range of
the queries.

uvaia's per-query result is an order-dependent state machine over the reference stream (src/nearest.c:435-510), but the
machines of different queries never read each other's state -- except for one number per batch, the snapshot
cq->max_incompatible = max over all heaps (src/nearest.c:290-291), which only matters when the query set has
constant-and-complete columns (n_idx_c > 0: the consensus pre-score is cut at that value).  So:

  * n_idx_c == 0 (the usual case: N-rich queries leave no complete column): no exchange at all; every rank runs
    `search_resident` on its range.  Work per rank = (queries / world) x (all references): pairs per GPU are constant when the
    database grows with the number of GPUs (weak scaling), and the result equals a single process over the whole stream.
  * n_idx_c > 0: pool by pool, the ranks all-reduce (max) their local maximum tolerance and pass it as the snapshot.

The column classes come from the WHOLE query set on every rank (the context is opened with all queries): scores such as
ACGT_matches_unique depend on them.  Compared with the reference-sharded ring of ring.py there is no state hand-over on the
critical path; the ring remains for databases that do not fit one GPU."""


def query_shard(n_query, rank, world):
    """[q0, q1) of this rank: contiguous, q0 a multiple of 64 (the scan's super-tile of queries); may be empty on high ranks."""
    per = -(-n_query // world)
    per = -(-per // 64) * 64
    q0 = min(n_query, rank * per)
    q1 = min(n_query, q0 + per)
    return q0, q1


def run_query_shard(engine, q0, q1, n_refs, pool, cons, allreduce_max=None):
    """One search over the resident database for queries [q0, q1).  allreduce_max(int) -> int over all ranks (only called when
    cons is true; every rank must call run_query_shard, also with an empty range)."""
    active = q1 > q0
    if active:
        engine.set_active_queries(q0, q1)
    if not cons or allreduce_max is None:
        if active:
            engine.search_resident(pool, want_entered=False)
        return
    if active:
        engine.entered_flags(clear=True)
    for a in range(0, n_refs, pool):
        local = engine.max_tolerance() if active else -(2 ** 31) + 1
        snap = allreduce_max(local)
        if active:
            engine.search_resident_pool(a, min(pool, n_refs - a), a, snap)


class TorchMax:
    """all-reduce(max) of one int over torch.distributed (nccl = RCCL on the GPU box, gloo on CPU)."""

    def __init__(self, dist, device):
        import torch
        self.dist, self.buf = dist, torch.zeros(1, dtype=torch.int32, device=device)

    def __call__(self, v):
        self.buf[0] = int(v)
        self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.MAX)
        return int(self.buf.item())
