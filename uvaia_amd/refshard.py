"""Multi-GPU driver by REFERENCE shards (one process per GPU; the default of bench.py --gpus N).

What shards how (include/uvaia_gpu.h "reference shards", DESIGN.md "Multi-GPU"):

  * the pair scan -- more than nine tenths of a search -- depends on nothing but the pair: every rank derives and scans only its own
    pieces of the reference stream, against ALL queries (the regime the scan kernel is built for: every reference plane reused by
    every query tile);
  * the gate + heap machine of a query (src/nearest.c:488,504-508) is sequential over the stream but independent of the other
    queries: every rank replays a contiguous range of the queries over ALL references, in stream order;
  * in between, the pair counters of a stripe (one piece per rank) move once: rank r sends to rank d the rows of d's queries --
    ONE all-to-all per stripe (RCCL over xGMI on the GPU box: torch.distributed all_to_all_single; gloo on CPU for the tests) for
    the counters and one for the per-tile bounds the replay skips tiles by;
  * the one coupling between queries, the batch snapshot (src/nearest.c:290-291), is a maximum over all ranks per batch, and only
    when the query set has constant-and-complete columns.

Layout: the stream is dealt in pieces of `piece` references (whole tiles of 64); piece p belongs to rank p % world.  A rank keeps
only its own pieces -- packed planes, side rows, the planes derived for the query set: 25 KB x its own references -- and ingests
only those.  What a replay needs of a piece scanned elsewhere travels with the counters (valid sites, consensus pre-score: the `aux`
block, all-gathered per stripe); the few words of packed planes and side rows it reads for the pairs that reach the exact comparison
are read in place from the rank that keeps them, through inter-process mappings set up once (TorchExchange.connect_peers).

Any object with the engine's shard calls works as `engine` (uvaia_amd.capi.Engine on a GPU; the CPU tests drive the same protocol over
gloo with an oracle-backed stand-in).
"""


class Piece:
    def __init__(self, first, n, owner):
        self.first, self.n, self.owner = int(first), int(n), int(owner)
        self.tiles = (self.first + self.n + 63) // 64 - self.first // 64

    def __repr__(self):
        return "Piece(%d,+%d,rank %d)" % (self.first, self.n, self.owner)


def query_shard(n_query, rank, world):
    """[q0, q1) of this rank: contiguous, whole query tiles of 16 (as uvaia_gpu_group_open cuts them); may be empty on high ranks."""
    per = -(-n_query // world)
    per = -(-per // 16) * 16
    q0 = min(n_query, rank * per)
    return q0, min(n_query, q0 + per)


class Plan:
    """Weak-scaling layout of bench.py: `world` x `refs_per_rank` references in all; pieces of about one scan launch each."""

    def __init__(self, world, rank, refs_per_rank, n_query, pool=None, piece=None):
        self.world, self.rank, self.n_query = int(world), int(rank), int(n_query)
        self.total = int(world) * int(refs_per_rank)
        if piece is None:        # as many near-equal pieces per rank as it takes to keep them near the scan's sub-slice length
            k = max(1, round(refs_per_rank / 32768.0))
            piece = -(-(-(-refs_per_rank // k)) // 64) * 64
        assert piece % 64 == 0 and piece >= 64
        self.piece = int(piece)
        self.slice_refs = self.piece                      # what the engine is opened with (max_pool): a rank never handles more at once
        self.pool = int(pool) if pool else self.total     # batches of the reference (src/nearest.c:249-251); whole stream when they cannot matter
        self.q0, self.q1 = query_shard(n_query, rank, world)
        self.shards = [query_shard(n_query, r, world) for r in range(world)]

    def pieces_of_pool(self, a, b):
        """the parts of the shard map's pieces inside the batch [a, b), in stream order"""
        out, x = [], a
        while x < b:
            pe = min(b, (x // self.piece + 1) * self.piece)
            out.append(Piece(x, pe - x, (x // self.piece) % self.world))
            x = pe
        return out

    def pools(self, cons):
        pool = self.pool if cons else self.total
        return [(a, min(self.total, a + pool)) for a in range(0, self.total, pool)]

    def stream_pieces(self):
        """the whole stream as pieces of the shard map, in order: what a rank walks while loading (its own pieces are appended, the
        others skipped: uvaia_gpu_db_skip)"""
        return self.pieces_of_pool(0, self.total)

    def describe(self):
        return ("reference shards: every GPU derives and scans its pieces of %d references (1/%d of the %d-reference stream) against all %d queries, "
                "one RCCL all-to-all per stripe moves the pair counters to the GPU that replays the query (%d queries per GPU, stream order); "
                "every GPU keeps and ingests its own pieces only (packed and derived planes; remote words read in place over hipIpc mappings); exact" % (self.piece, self.world, self.total, self.n_query, self.q1 - self.q0))


class TorchExchange:
    """Buffers and collectives of one rank.  send[b]: what this rank's scan writes (all query rows of one piece); recv[b]: the rows of
    this rank's queries for the `world` pieces of a stripe, source-major.  Two of each: the scan of stripe s+1 runs while stripe s is
    exchanged and replayed."""

    def __init__(self, dist, plan, engine, device, pinned=False):
        """device "cuda": RCCL moves device buffers.  device "cpu" with pinned=True: a GPU engine writes its counters into pinned host
        memory and gloo moves them (rehearsal of several ranks on a box with fewer GPUs); plain "cpu": the CPU stand-in of the tests."""
        import torch
        self.torch, self.dist, self.plan, self.device = torch, dist, plan, device
        rows = engine.shard_rows()
        cols = plan.piece + 64                                    # a piece cut by a batch boundary may start inside a tile
        my = max(1, plan.q1 - plan.q0)
        mk = lambda n: torch.empty(n, dtype=torch.int32, device=device, pin_memory=(pinned and device == "cpu"))
        self.send_cnt = [mk(rows * cols) for _ in range(2)]            # one dword per pair: first counter | second << 16
        self.send_tmin = [mk(rows * (cols // 64) * 2) for _ in range(2)]
        self.recv_cnt = [mk(plan.world * my * cols) for _ in range(2)]
        self.recv_tmin = [mk(plan.world * my * (cols // 64) * 2) for _ in range(2)]
        self.aux_ints = (engine.shard_aux_bytes(cols // 64) + 3) // 4         # per piece: valid sites (+ consensus pre-score) of its references
        self.send_aux = [mk(self.aux_ints) for _ in range(2)]
        self.recv_aux = [mk(plan.world * self.aux_ints) for _ in range(2)]
        self.maxbuf = torch.zeros(1, dtype=torch.int32, device=device)
        if device != "cpu":
            torch.cuda.current_stream().synchronize()

    def connect_peers(self, engine):
        """once, after the database is reserved on every rank: every rank maps the packed planes and side rows of the others (hipIpc
        handles, all-gathered as bytes) so that its replay can read the few words it needs of a reference kept elsewhere"""
        if self.plan.world == 1 or not hasattr(engine, "shard_ipc_handles"):
            return
        torch = self.torch
        mine = engine.shard_ipc_handles()
        t = torch.tensor(list(mine), dtype=torch.uint8, device=self.device)
        out = [torch.empty_like(t) for _ in range(self.plan.world)]
        self.dist.all_gather(out, t)
        for r in range(self.plan.world):
            if r != self.plan.rank:
                engine.shard_ipc_open(r, bytes(out[r].cpu().tolist()))

    def disconnect_peers(self, engine):
        """before the engines are closed: every rank unmaps the others' arrays, then all meet (an owner must not free what is still mapped)"""
        if self.plan.world > 1 and hasattr(engine, "shard_ipc_close"):
            engine.shard_ipc_close()
            self.dist.barrier()

    def all_max(self, v):
        self.maxbuf[0] = int(v)
        self.dist.all_reduce(self.maxbuf, op=self.dist.ReduceOp.MAX)
        return int(self.maxbuf.item())

    def exchange(self, b, stripe):
        """rows of every rank's queries out of send[b] (this rank's piece of the stripe, if it has one), into recv[b].  Returns, per piece
        of the stripe, (pointer to its counters, pointer to its bounds) in recv[b]."""
        plan, torch = self.plan, self.torch
        mine = [p for p in stripe if p.owner == plan.rank]
        my = plan.q1 - plan.q0
        where = []
        for what, ints_per_col, send, recv in (("cnt", 64, self.send_cnt[b], self.recv_cnt[b]), ("tmin", 2, self.send_tmin[b], self.recv_tmin[b])):
            # input: for destination d the rows [q0_d, q1_d) of this rank's piece -- contiguous in the scan's output, in rank order
            t_me = mine[0].tiles if mine else 0
            in_split = [(q1 - q0) * t_me * ints_per_col for q0, q1 in plan.shards]
            by_owner = {p.owner: p for p in stripe}
            out_split = [my * by_owner[r].tiles * ints_per_col if r in by_owner else 0 for r in range(plan.world)]
            inp = send[:sum(in_split)]
            out = recv[:sum(out_split)]
            self.dist.all_to_all_single(out, inp, out_split, in_split)
            offs, at = {}, 0
            for r in range(plan.world):
                offs[r] = at
                at += out_split[r]
            where.append({r: recv.data_ptr() + 4 * offs[r] for r in by_owner})
        # the aux block of every piece of the stripe goes to every rank (fixed size; a rank without a piece in a short last stripe sends filler)
        self.dist.all_gather_into_tensor(self.recv_aux[b], self.send_aux[b])
        return [(where[0][p.owner], where[1][p.owner], self.recv_aux[b].data_ptr() + 4 * self.aux_ints * p.owner) for p in stripe]

    def stream(self):
        """the stream the collectives run on (RCCL: torch's current stream), as the engine's ordering calls take it"""
        return self.torch.cuda.current_stream().cuda_stream


def run(engine, plan, xchg, cons, ordinal0=0):
    """One search of the resident database (the whole while-loop of src/nearest.c:249-330), sharded.  Every rank calls it."""
    active = plan.q1 > plan.q0
    stripe_no = 0
    for a, b in plan.pools(cons):
        if cons:         # the batch snapshot: maximum of the tolerances over all queries, i.e. over all ranks
            engine.replay_wait()
            if active:
                engine.set_active_queries(plan.q0, plan.q1)
                local = engine.max_tolerance()
                engine.set_active_queries(0, plan.n_query)
            else:
                local = -(2 ** 31) + 1
            engine.set_snapshot(xchg.all_max(local))
        pieces = plan.pieces_of_pool(a, b)
        stripes = [pieces[i:i + plan.world] for i in range(0, len(pieces), plan.world)]

        def scan(k, buf):
            for p in stripes[k]:
                if p.owner == plan.rank:
                    engine.shard_scan(p.first, p.n, xchg.send_cnt[buf].data_ptr(), xchg.send_tmin[buf].data_ptr(), xchg.send_aux[buf].data_ptr())

        # On device buffers (RCCL) nothing below blocks the host: the scan stream, the stream of the collectives and the replay stream
        # are ordered by events (uvaia_gpu_mark / stream_wait_mark / wait_stream).  Marks 0, 1 = the scan into send[0], send[1];
        # 2, 3 = the replays that read recv[0], recv[1].  Over gloo the buffers are pinned host memory and the collectives run on
        # the host: there the host waits.
        on_device = xchg.device != "cpu" and hasattr(engine, "mark")
        SCANS, REPLAYS = 0, 1
        scan(0, stripe_no & 1)
        if on_device:
            engine.mark(SCANS, stripe_no & 1)
        for k, stripe in enumerate(stripes):
            buf = stripe_no & 1
            if on_device:
                engine.stream_wait_mark(xchg.stream(), buf)      # the exchange's input: this stripe's scan
            else:
                engine.scan_wait()                               # this stripe's counters are complete
            if k + 1 < len(stripes):
                if on_device:
                    engine.wait_stream(SCANS, xchg.stream())     # send[buf ^ 1] is still being read by the exchange of the stripe before
                scan(k + 1, buf ^ 1)                             # the next scan runs while this stripe is exchanged and replayed
                if on_device:
                    engine.mark(SCANS, buf ^ 1)
            if stripe_no >= 2:                                   # recv[buf] was read by the replays of two stripes ago
                if on_device:
                    engine.stream_wait_mark(xchg.stream(), 2 + buf)
                else:
                    engine.replay_wait()
            got = xchg.exchange(buf, stripe)
            if on_device:
                engine.wait_stream(REPLAYS, xchg.stream())       # the replays below read what the exchange delivers
            if active:
                for p, (cnt_ptr, tmin_ptr, aux_ptr) in zip(stripe, got):  # stream order
                    engine.shard_replay(cnt_ptr, tmin_ptr, aux_ptr, p.owner, p.first, p.n, ordinal0 + p.first, plan.q0, plan.q1)
            if on_device:
                engine.mark(REPLAYS, 2 + buf)
            stripe_no += 1
