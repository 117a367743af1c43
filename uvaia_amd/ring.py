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
    def __init__(self, nbytes, device):
        import torch
        self.tensor = torch.zeros(nbytes, dtype=torch.uint8, device=device)
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
