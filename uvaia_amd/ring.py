"""Multi-GPU driver: block-cyclic database, concurrent scans, heap state rippling rank to rank.

Why a ring and not an all-gather of per-shard top-k heaps: uvaia's per-query result is an order-dependent state machine
(ranked by matches, gated by a mismatch tolerance that is re-derived from the current worst kept entry, reference
src/nearest.c:488,504-508), so per-shard heaps cannot be merged afterwards and a candidate list that is valid for a
whole batch passes 25-99 % of all pairs (profiles/r01_candidate_rates*.txt).  What is small is the state itself
(n_query x (k+1) x 32 B), and what is expensive -- the pair scan -- does not depend on it.  So every rank scans its
slice of a stripe concurrently, and the state visits the ranks in stream order, point to point (xGMI is point to point).

    stripe s = one batch ("pool") of the reference = slice(s, rank 0) ++ slice(s, rank 1) ++ ... in stream order

Any object with slice_scan / slice_replay / state_import / state_export / state_bytes works as the engine (uvaia_amd.capi.
Engine on a GPU; the CPU tests drive the same protocol over gloo with an oracle-backed stand-in).
"""


class Slice:
    """One rank's part of one stripe: `n` references starting at local database position `first`, whose first reference
    has stream ordinal `ordinal0`."""

    def __init__(self, first, n, ordinal0):
        self.first, self.n, self.ordinal0 = int(first), int(n), int(ordinal0)


def block_cyclic_layout(n_total_per_rank, slice_size, rank, world):
    """Weak-scaling layout: every rank holds n_total_per_rank references cut into slices of slice_size; stripe s is made of
    slice s of ranks 0..world-1 in that order.  Returns this rank's slices with their stream ordinals."""
    out, first, s = [], 0, 0
    while first < n_total_per_rank:
        n = min(slice_size, n_total_per_rank - first)
        # stream position: all earlier stripes (world full slices each) + the slices of earlier ranks in this stripe
        out.append(Slice(first, n, s * world * slice_size + rank * n))
        first += n
        s += 1
    return out


class TorchComm:
    """State blobs over torch.distributed point-to-point ops (backend nccl = RCCL over xGMI on the GPU box, gloo on CPU).
    Buffers are objects with `.tensor` (what is sent) and `.ptr` (what the engine reads/writes)."""

    def __init__(self, dist, cuda):
        self.dist, self.cuda = dist, cuda

    def _host_wait(self):
        if self.cuda:
            import torch
            torch.cuda.current_stream().synchronize()      # the engine uses its own streams: order through the host

    def send(self, buf, dst):
        self.dist.send(buf.tensor, dst)

    def recv(self, buf, src):
        self.dist.recv(buf.tensor, src)
        self._host_wait()

    def flush(self):
        self._host_wait()                                   # earlier sends have read their buffer


class TorchStateBuffer:
    """Raw bytes the engine exports a state range into / imports it from.  Uninitialised on purpose: the engine writes or the
    transport fills the whole range, and a fill kernel on torch's stream would not be ordered against the engine's own streams."""

    def __init__(self, nbytes, device):
        import torch
        self.tensor = torch.empty(nbytes, dtype=torch.uint8, device=device)
        if device != "cpu":
            torch.cuda.current_stream().synchronize()      # the allocation itself may queue work on torch's stream
        self.ptr = self.tensor.data_ptr()


def run_ring(engine, comm, rank, world, slices, alloc_state):
    """Runs all stripes on one rank.  comm.send(buf, dst) / comm.recv(buf, src) move a state blob (blocking, in order);
    alloc_state() returns a buffer object with a `.ptr` the engine can export to / import from.
    After the last stripe the final heaps live on rank world-1; returns True there."""
    n = len(slices)
    if n == 0:
        return None
    state, outbox = alloc_state(), alloc_state()
    engine.slice_scan(slices[0].first, slices[0].n, 0)
    for s in range(n):
        if s + 1 < n:                                   # keep the GPU busy with the next slice while the state is away
            engine.slice_scan(slices[s + 1].first, slices[s + 1].n, (s + 1) & 1)
        first_hop = (rank == 0 and s == 0)
        if world > 1 and not first_hop:
            comm.recv(state, (rank - 1) % world)
            engine.state_import(state.ptr)
        engine.slice_replay(s & 1, slices[s].ordinal0, stripe_start=(rank == 0))
        last_hop = (rank == world - 1 and s == n - 1)
        if world > 1 and not last_hop:
            comm.flush()
            engine.state_export(outbox.ptr)
            comm.send(outbox, (rank + 1) % world)
    if world > 1:
        comm.flush()
    return rank == world - 1


def run_ring_in_one_process(engines, slices_per_rank, alloc_state):
    """The same protocol with every rank's engine driven from one process, hop by hop in stream order (used to test the
    GPU slice/state entry points with several contexts on one card)."""
    world = len(engines)
    n = len(slices_per_rank[0])
    state = alloc_state()
    for r in range(world):
        if n:
            engines[r].slice_scan(slices_per_rank[r][0].first, slices_per_rank[r][0].n, 0)
    have_state = False
    for s in range(n):
        for r in range(world):
            sl = slices_per_rank[r]
            if s + 1 < n:
                engines[r].slice_scan(sl[s + 1].first, sl[s + 1].n, (s + 1) & 1)
            if have_state and world > 1:
                engines[r].state_import(state.ptr)
            engines[r].slice_replay(s & 1, sl[s].ordinal0, stripe_start=(r == 0))
            if world > 1:
                engines[r].state_export(state.ptr)
                have_state = True
    return engines[-1]


# ----------------------------------------------------------------------------------------------------------------------
# Query-group pipelined ring.  The per-query machines are independent, so the state travels as `world` blobs (one per
# contiguous group of queries).  While rank r replays group j of its slice, rank r+1 replays group j-1 of its own slice:
# every hop carries 1/world of the replay work and, after a fill of world-1 hops, all ranks replay concurrently.
# The only coupling between queries is the batch snapshot (max tolerance over ALL queries at stripe start, reference
# src/nearest.c:290-291), which matters only if the query set has constant-and-complete columns (`cons`): then rank 0
# waits for every group of the previous stripe before it opens the next one.
# ----------------------------------------------------------------------------------------------------------------------
def query_groups(n_query, world):
    b = [n_query * j // world for j in range(world + 1)]
    return [(b[j], b[j + 1]) for j in range(world)]


class TorchRingComm:
    """Non-blocking point-to-point transport.  The wrap-around link (last rank -> rank 0) uses its own process group so that,
    whatever the backend serialises per communicator, the last rank's sends and rank 0's receives never queue behind the
    forward traffic of the same pair of ranks (world = 2)."""

    def __init__(self, dist, rank, world, cuda):
        self.dist, self.rank, self.world, self.cuda = dist, rank, world, cuda
        self.fwd = dist.new_group(list(range(world)))
        self.wrap = dist.new_group(list(range(world)))

    def _group(self, src, dst):
        return self.wrap if (src == self.world - 1 and dst == 0) else self.fwd

    def irecv(self, buf, src):
        return self.dist.irecv(buf.tensor, src, group=self._group(src, self.rank))

    def isend(self, buf, dst):
        return self.dist.isend(buf.tensor, dst, group=self._group(self.rank, dst))

    def wait(self, work):
        work.wait()
        if self.cuda:
            import torch
            torch.cuda.current_stream().synchronize()       # the engine's streams are not torch's: order through the host


def run_ring_grouped(engine, comm, rank, world, slices, n_query, cons, make_buffer):
    """All stripes on one rank, state in `world` per-query-group blobs.  make_buffer(nbytes) -> object with .tensor/.ptr.
    Returns True on the rank that ends up holding the final heaps (world-1)."""
    groups = query_groups(n_query, world)
    n = len(slices)
    if n == 0:
        return rank == world - 1
    src, dst = (rank - 1) % world, (rank + 1) % world
    # Receives are posted in program order, right before they are needed (never ahead of this rank's own sends): if the
    # backend serialises all point-to-point operations of a process group on one stream, a receive posted early would
    # hold up the sends queued behind it.  Sends are non-blocking; the wrap-around link has its own process group.
    sends = []                     # (buffer, work) of sends that may still be reading their buffer
    free_out = {}                  # size -> buffers whose send has completed (recycled: memory stays bounded by the sends in flight)
    recv_buf = {}                  # one receive buffer per group size: state_import_range returns once the engine has read it

    def out_buffer(nbytes):
        for i in range(len(sends) - 1, -1, -1):
            if sends[i][1].is_completed():
                b, _ = sends.pop(i)
                free_out.setdefault(b.tensor.numel(), []).append(b)
        pool = free_out.get(nbytes)
        return pool.pop() if pool else make_buffer(nbytes)

    def receive(s, j, q0, q1):
        nb = engine.state_range_bytes(q0, q1)
        buf = recv_buf.get(nb)
        if buf is None:
            buf = recv_buf[nb] = make_buffer(nb)
        comm.wait(comm.irecv(buf, src))
        engine.state_import_range(buf.ptr, q0, q1)

    engine.slice_scan(slices[0].first, slices[0].n, 0)
    for s in range(n):
        if s + 1 < n:
            engine.slice_scan(slices[s + 1].first, slices[s + 1].n, (s + 1) & 1)
        expects = not (rank == 0 and s == 0)
        gather_first = (rank == 0 and s > 0 and cons)      # the snapshot needs the tolerance of every query
        if gather_first:
            for j, (q0, q1) in enumerate(groups):
                receive(s, j, q0, q1)
        for j, (q0, q1) in enumerate(groups):
            if expects and not gather_first:
                receive(s, j, q0, q1)
            take_snapshot = (rank == 0 and j == 0 and (cons or s == 0))
            engine.slice_replay_range(s & 1, slices[s].ordinal0, q0, q1, take_snapshot)
            if not (rank == world - 1 and s == n - 1):
                out = out_buffer(engine.state_range_bytes(q0, q1))
                engine.state_export_range(out.ptr, q0, q1)
                sends.append((out, comm.isend(out, dst)))
    for out, work in sends:
        work.wait()
    return rank == world - 1


def run_ring_grouped_in_one_process(engines, slices_per_rank, n_query, cons, make_buffer):
    """The grouped protocol with every rank's engine driven from one process in a valid serial order (tests)."""
    world = len(engines)
    groups = query_groups(n_query, world)
    n = len(slices_per_rank[0])
    blobs = {}                                             # (stripe, receiving rank, group) -> buffer
    for r in range(world):
        if n:
            engines[r].slice_scan(slices_per_rank[r][0].first, slices_per_rank[r][0].n, 0)
    for s in range(n):
        for r in range(world):
            sl = slices_per_rank[r]
            if s + 1 < n:
                engines[r].slice_scan(sl[s + 1].first, sl[s + 1].n, (s + 1) & 1)
            gather_first = (r == 0 and s > 0 and cons)
            if gather_first:
                for j, (q0, q1) in enumerate(groups):
                    blob = blobs.pop((s, r, j))                 # keep a reference while the engine reads it
                    engines[r].state_import_range(blob.ptr, q0, q1)
            for j, (q0, q1) in enumerate(groups):
                if (s, r, j) in blobs:
                    blob = blobs.pop((s, r, j))
                    engines[r].state_import_range(blob.ptr, q0, q1)
                engines[r].slice_replay_range(s & 1, sl[s].ordinal0, q0, q1, r == 0 and j == 0 and (cons or s == 0))
                if not (r == world - 1 and s == n - 1):
                    out = make_buffer(engines[r].state_range_bytes(q0, q1))
                    engines[r].state_export_range(out.ptr, q0, q1)
                    blobs[(s, r + 1, j) if r + 1 < world else (s + 1, 0, j)] = out
    return engines[-1]
