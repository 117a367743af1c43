"""ctypes binding of include/uvaia_gpu.h (the only way Python reaches the engine)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
_LIB = os.path.join(_HERE, "lib", "libuvaia_gpu.so")
NSCORE = 6

# every symbol include/uvaia_gpu.h declares (tests check the library exports all of them)
SYMBOLS = [
    "uvaia_gpu_open", "uvaia_gpu_open_tuned", "uvaia_gpu_close", "uvaia_gpu_last_error", "uvaia_gpu_push", "uvaia_gpu_drain",
    "uvaia_gpu_heap_slots", "uvaia_gpu_n_query", "uvaia_gpu_reset", "uvaia_gpu_db_reserve", "uvaia_gpu_db_append",
    "uvaia_gpu_db_append_block", "uvaia_gpu_db_size", "uvaia_gpu_search_resident", "uvaia_gpu_sync", "uvaia_gpu_ball", "uvaia_gpu_ball_resident", "uvaia_gpu_ball_asked", "uvaia_gpu_ball_kernel_ms", "uvaia_gpu_export_query_table", "uvaia_gpu_agree_on_polymorphic", "uvaia_gpu_query_columns",
    "uvaia_gpu_last_batch_scores", "uvaia_gpu_scan_stats", "uvaia_gpu_replay_stats", "uvaia_gpu_replay_tiles_opened", "uvaia_gpu_replay_timing",
    "uvaia_gpu_state_bytes", "uvaia_gpu_state_export", "uvaia_gpu_state_import", "uvaia_gpu_slice_scan", "uvaia_gpu_slice_replay",
    "uvaia_gpu_entered_flags", "uvaia_gpu_state_range_bytes", "uvaia_gpu_state_export_range", "uvaia_gpu_state_import_range",
    "uvaia_gpu_slice_replay_range", "uvaia_gpu_slice_buffers", "uvaia_gpu_scan_bytes_per_ref", "uvaia_gpu_derived_bytes_per_ref", "uvaia_gpu_scan_variant", "uvaia_gpu_set_query_tile", "uvaia_gpu_packed_bytes_per_ref",
    "uvaia_gpu_db_tile_bytes", "uvaia_gpu_db_side_row_ints", "uvaia_gpu_db_export", "uvaia_gpu_db_append_packed", "uvaia_gpu_db_clear", "uvaia_gpu_db_rederive",
    "uvaia_gpu_set_active_queries", "uvaia_gpu_max_tolerance", "uvaia_gpu_search_resident_pool",
    "uvaia_gpu_db_set_shard", "uvaia_gpu_shard_rows", "uvaia_gpu_shard_scan", "uvaia_gpu_scan_wait", "uvaia_gpu_replay_wait", "uvaia_gpu_set_snapshot", "uvaia_gpu_shard_replay",
    "uvaia_gpu_mark", "uvaia_gpu_stream_wait_mark", "uvaia_gpu_wait_stream",
    "uvaia_gpu_shard_aux_bytes", "uvaia_gpu_db_skip", "uvaia_gpu_shard_set_peer", "uvaia_gpu_shard_planes", "uvaia_gpu_shard_side_rows",
    "uvaia_gpu_shard_ipc_handle_bytes", "uvaia_gpu_shard_ipc_handles", "uvaia_gpu_shard_ipc_open", "uvaia_gpu_shard_ipc_close",
    "uvaia_gpu_group_open", "uvaia_gpu_group_close", "uvaia_gpu_group_last_error", "uvaia_gpu_group_size", "uvaia_gpu_group_member", "uvaia_gpu_group_query_shard",
    "uvaia_gpu_group_db_reserve", "uvaia_gpu_group_db_append", "uvaia_gpu_group_db_append_packed", "uvaia_gpu_group_db_clear", "uvaia_gpu_group_db_rederive",
    "uvaia_gpu_group_db_size", "uvaia_gpu_group_reset", "uvaia_gpu_group_search_resident", "uvaia_gpu_group_push", "uvaia_gpu_group_drain", "uvaia_gpu_group_sync",
]


class GpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("uvaia_gpu error %d: %s" % (code, msg))
        self.code = code


class Tuning(C.Structure):
    """uvaia_gpu_tuning: optional knobs of uvaia_gpu_open_tuned (0 = the library's choice); they change speed, never results."""
    _fields_ = [("subslice_refs", C.c_size_t), ("rare_max", C.c_int), ("scan", C.c_int), ("serial", C.c_int),
                ("scan_tiles_per_wave", C.c_int), ("scan_waves_per_block", C.c_int), ("rederive_streams", C.c_int), ("ball_gather", C.c_int), ("query_tables", C.c_int), ("replay_extras", C.c_int), ("replay_cus", C.c_int), ("scan_streams", C.c_int), ("pipeline", C.c_int), ("head_scan", C.c_int)]

    SCAN = {"auto": 0, "packed": 1, "compressed": 2, "wide": 3}

    @classmethod
    def make(cls, tuning):
        """None, a Tuning, or a dict such as {"scan": "compressed", "subslice_refs": 448, "serial": 1}"""
        if tuning is None or isinstance(tuning, cls):
            return tuning
        t = cls()
        for k, v in tuning.items():
            setattr(t, k, cls.SCAN[v] if k == "scan" and isinstance(v, str) else int(v))
        return t


class _Query(C.Structure):
    _fields_ = [
        ("n_query", C.c_int), ("nchar", C.c_int),
        ("seq", C.POINTER(C.c_char_p)), ("consensus", C.c_char_p),
        ("idx_c", C.POINTER(C.c_size_t)), ("idx_m", C.POINTER(C.c_size_t)), ("idx", C.POINTER(C.c_size_t)),
        ("n_idx_c", C.c_int), ("n_idx_m", C.c_int), ("n_idx", C.c_int),
        ("trim", C.c_size_t), ("acgt", C.c_int),
    ]


def library_path():
    return _LIB


def build_library(force=False):
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".inc"))] + [os.path.join(ROOT, "include", "uvaia_gpu.h")]
    if not force and os.path.exists(_LIB) and os.path.getmtime(_LIB) >= max(os.path.getmtime(f) for f in srcs):
        return _LIB
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s"])
    return _LIB


_lib = None


def load_library():
    """Load libuvaia_gpu.so; raises if it has not been built (there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise GpuError(-2, "HIP library %s is missing: run __graft_entry__.build() (no CPU fallback exists)" % _LIB)
    L = C.CDLL(_LIB)
    vp = C.c_void_p
    pp = C.POINTER(C.c_char_p)
    pi = C.POINTER(C.c_int)
    sig = {
        "uvaia_gpu_open": (C.c_int, [C.POINTER(vp), C.POINTER(_Query), C.c_int, C.c_int, C.c_size_t]),
        "uvaia_gpu_open_tuned": (C.c_int, [C.POINTER(vp), C.POINTER(_Query), C.c_int, C.c_int, C.c_size_t, C.POINTER(Tuning)]),
        "uvaia_gpu_close": (None, [vp]),
        "uvaia_gpu_last_error": (C.c_char_p, [vp]),
        "uvaia_gpu_push": (C.c_int, [vp, pp, pi, C.c_int, C.c_int64, C.POINTER(C.c_uint8)]),
        "uvaia_gpu_drain": (C.c_int, [vp, pi, pi, pi, C.POINTER(C.c_int64)]),
        "uvaia_gpu_heap_slots": (C.c_int, [vp]),
        "uvaia_gpu_n_query": (C.c_int, [vp]),
        "uvaia_gpu_reset": (C.c_int, [vp]),
        "uvaia_gpu_db_reserve": (C.c_int, [vp, C.c_size_t]),
        "uvaia_gpu_db_append": (C.c_int, [vp, pp, pi, C.c_int]),
        "uvaia_gpu_db_append_block": (C.c_int, [vp, C.c_void_p, C.c_size_t, pi, C.c_int]),
        "uvaia_gpu_db_size": (C.c_size_t, [vp]),
        "uvaia_gpu_search_resident": (C.c_int, [vp, C.c_size_t, C.c_int64, C.POINTER(C.c_uint8)]),
        "uvaia_gpu_sync": (C.c_int, [vp]),
        "uvaia_gpu_ball": (C.c_int, [vp, pp, C.c_int, C.c_int, pi]),
        "uvaia_gpu_ball_resident": (C.c_int, [vp, C.c_size_t, C.c_size_t, C.c_int, pi]),
        "uvaia_gpu_ball_asked": (C.c_ulonglong, [vp, C.c_int]),
        "uvaia_gpu_ball_kernel_ms": (None, [vp, C.POINTER(C.c_double), C.c_int]),
        "uvaia_gpu_export_query_table": (C.c_int, [vp, C.c_int, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
        "uvaia_gpu_agree_on_polymorphic": (C.c_int, [vp, pp, C.c_int, C.POINTER(C.c_uint8)]),
        "uvaia_gpu_query_columns": (C.c_int, [pp, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_uint8)]),
        "uvaia_gpu_last_batch_scores": (C.c_int, [vp, pi, C.c_int]),
        "uvaia_gpu_scan_stats": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.c_int]),
        "uvaia_gpu_replay_stats": (C.c_int, [vp, C.POINTER(C.c_ulonglong), C.c_int]),
        "uvaia_gpu_replay_tiles_opened": (C.c_int, [vp, C.POINTER(C.c_ulonglong), C.c_int]),
        "uvaia_gpu_replay_timing": (C.c_int, [vp, C.POINTER(C.c_ulonglong), C.c_int]),
        "uvaia_gpu_state_bytes": (C.c_size_t, [vp]),
        "uvaia_gpu_state_export": (C.c_int, [vp, C.c_void_p]),
        "uvaia_gpu_state_import": (C.c_int, [vp, C.c_void_p]),
        "uvaia_gpu_slice_scan": (C.c_int, [vp, C.c_size_t, C.c_size_t, C.c_int]),
        "uvaia_gpu_slice_replay": (C.c_int, [vp, C.c_int, C.c_int64, C.c_int]),
        "uvaia_gpu_entered_flags": (C.c_int, [vp, C.POINTER(C.c_uint8), C.c_int]),
        "uvaia_gpu_state_range_bytes": (C.c_size_t, [vp, C.c_int, C.c_int]),
        "uvaia_gpu_state_export_range": (C.c_int, [vp, C.c_void_p, C.c_int, C.c_int]),
        "uvaia_gpu_state_import_range": (C.c_int, [vp, C.c_void_p, C.c_int, C.c_int]),
        "uvaia_gpu_slice_replay_range": (C.c_int, [vp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int]),
        "uvaia_gpu_slice_buffers": (C.c_int, []),
        "uvaia_gpu_scan_bytes_per_ref": (C.c_size_t, [vp]),
        "uvaia_gpu_derived_bytes_per_ref": (C.c_size_t, [vp]),
        "uvaia_gpu_scan_variant": (C.c_int, [vp]),
        "uvaia_gpu_set_query_tile": (C.c_int, [vp, C.c_int]),
        "uvaia_gpu_packed_bytes_per_ref": (C.c_size_t, [vp]),
        "uvaia_gpu_set_active_queries": (C.c_int, [vp, C.c_int, C.c_int]),
        "uvaia_gpu_max_tolerance": (C.c_int, [vp, pi]),
        "uvaia_gpu_search_resident_pool": (C.c_int, [vp, C.c_size_t, C.c_size_t, C.c_int64, C.c_int]),
        "uvaia_gpu_db_tile_bytes": (C.c_size_t, [vp]),
        "uvaia_gpu_db_clear": (C.c_int, [vp]),
        "uvaia_gpu_db_rederive": (C.c_int, [vp]),
        "uvaia_gpu_db_side_row_ints": (C.c_int, []),
        "uvaia_gpu_db_export": (C.c_int, [vp, C.c_size_t, C.c_size_t, C.c_void_p, pi, pi]),
        "uvaia_gpu_db_append_packed": (C.c_int, [vp, C.c_void_p, pi, pi, C.c_int]),
        "uvaia_gpu_db_set_shard": (C.c_int, [vp, C.c_int, C.c_int, C.c_size_t]),
        "uvaia_gpu_shard_rows": (C.c_int, [vp]),
        "uvaia_gpu_shard_scan": (C.c_int, [vp, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
        "uvaia_gpu_shard_aux_bytes": (C.c_size_t, [vp, C.c_size_t]), "uvaia_gpu_db_skip": (C.c_int, [vp, C.c_size_t]),
        "uvaia_gpu_shard_set_peer": (C.c_int, [vp, C.c_int, C.c_void_p, C.c_void_p]),
        "uvaia_gpu_shard_planes": (C.c_void_p, [vp]), "uvaia_gpu_shard_side_rows": (C.c_void_p, [vp]),
        "uvaia_gpu_shard_ipc_handle_bytes": (C.c_int, []), "uvaia_gpu_shard_ipc_handles": (C.c_int, [vp, C.c_void_p]),
        "uvaia_gpu_shard_ipc_open": (C.c_int, [vp, C.c_int, C.c_void_p]), "uvaia_gpu_shard_ipc_close": (C.c_int, [vp]),
        "uvaia_gpu_scan_wait": (C.c_int, [vp]),
        "uvaia_gpu_replay_wait": (C.c_int, [vp]),
        "uvaia_gpu_set_snapshot": (C.c_int, [vp, C.c_int]),
        "uvaia_gpu_shard_replay": (C.c_int, [vp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_int64, C.c_int, C.c_int]),
        "uvaia_gpu_mark": (C.c_int, [vp, C.c_int, C.c_int]), "uvaia_gpu_stream_wait_mark": (C.c_int, [vp, C.c_void_p, C.c_int]),
        "uvaia_gpu_wait_stream": (C.c_int, [vp, C.c_int, C.c_void_p]),
        "uvaia_gpu_group_open": (C.c_int, [C.POINTER(vp), C.POINTER(_Query), C.c_int, pi, C.c_int, C.c_size_t, C.c_size_t]),
        "uvaia_gpu_group_close": (None, [vp]),
        "uvaia_gpu_group_last_error": (C.c_char_p, [vp]),
        "uvaia_gpu_group_size": (C.c_int, [vp]),
        "uvaia_gpu_group_member": (vp, [vp, C.c_int]),
        "uvaia_gpu_group_query_shard": (C.c_int, [vp, C.c_int, pi, pi]),
        "uvaia_gpu_group_db_reserve": (C.c_int, [vp, C.c_size_t]),
        "uvaia_gpu_group_db_append": (C.c_int, [vp, pp, pi, C.c_int]),
        "uvaia_gpu_group_db_append_packed": (C.c_int, [vp, C.c_void_p, pi, pi, C.c_int]),
        "uvaia_gpu_group_db_clear": (C.c_int, [vp]),
        "uvaia_gpu_group_db_rederive": (C.c_int, [vp]),
        "uvaia_gpu_group_db_size": (C.c_size_t, [vp]),
        "uvaia_gpu_group_reset": (C.c_int, [vp]),
        "uvaia_gpu_group_search_resident": (C.c_int, [vp, C.c_size_t, C.c_int64, C.POINTER(C.c_uint8)]),
        "uvaia_gpu_group_push": (C.c_int, [vp, pp, pi, C.c_int, C.c_int64, C.POINTER(C.c_uint8)]),
        "uvaia_gpu_group_drain": (C.c_int, [vp, pi, pi, pi, C.POINTER(C.c_int64)]),
        "uvaia_gpu_group_sync": (C.c_int, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def _cstrs(strs):
    arr = (C.c_char_p * len(strs))()
    for i, s in enumerate(strs):
        arr[i] = s
    return arr


def _int_ptr(a):
    if a is None:
        return None, None
    arr = np.ascontiguousarray(a, dtype=np.int32)
    return arr, arr.ctypes.data_as(C.POINTER(C.c_int))


class Engine:
    """One GPU context over a prepared query set (fields of struct query_struct, src/fastaseq.h:41-48)."""

    def __init__(self, seqs, consensus, idx_c, idx_m, idx, trim=0, acgt=False, nbest=100, max_pool=4096, device=-1, tuning=None):
        self.L = load_library()
        self.nq, self.nchar = len(seqs), len(consensus)
        self._keep = [_cstrs(seqs), consensus,
                      (C.c_size_t * len(idx_c))(*[int(x) for x in idx_c]),
                      (C.c_size_t * len(idx_m))(*[int(x) for x in idx_m]),
                      (C.c_size_t * len(idx))(*[int(x) for x in idx])]
        q = _Query(self.nq, self.nchar, self._keep[0], consensus, self._keep[2], self._keep[3], self._keep[4],
                   len(idx_c), len(idx_m), len(idx), int(trim), int(bool(acgt)))
        self.ctx = C.c_void_p()
        tn = Tuning.make(tuning)
        rc = self.L.uvaia_gpu_open_tuned(C.byref(self.ctx), C.byref(q), int(nbest), int(device), int(max_pool), C.byref(tn) if tn is not None else None)
        if rc != 0:
            msg = self.L.uvaia_gpu_last_error(None)
            self.ctx = None
            raise GpuError(rc, msg.decode() if msg else "?")
        self.slots = self.L.uvaia_gpu_heap_slots(self.ctx)
        self.ordinal = 0

    @classmethod
    def from_query(cls, q, **kw):
        """q: any object with seqs, consensus, idx_c, idx_m, idx, trim, acgt."""
        return cls(q.seqs, q.consensus, q.idx_c, q.idx_m, q.idx, trim=q.trim, acgt=q.acgt, **kw)

    def _chk(self, rc):
        if rc != 0:
            msg = self.L.uvaia_gpu_last_error(self.ctx)
            raise GpuError(rc, msg.decode() if msg else "?")

    def close(self):
        if getattr(self, "ctx", None):
            self.L.uvaia_gpu_close(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def push(self, refs, non_n=None, ordinal0=None):
        """One batch (= one pool of the reference) in stream order; returns the 'entered any heap' flags."""
        n = len(refs)
        if ordinal0 is None:
            ordinal0 = self.ordinal
        self.ordinal = ordinal0 + n
        entered = np.zeros(n, dtype=np.uint8)
        if n == 0:
            return entered
        _k, nn = _int_ptr(non_n)
        self._chk(self.L.uvaia_gpu_push(self.ctx, _cstrs(refs), nn, n, ordinal0, entered.ctypes.data_as(C.POINTER(C.c_uint8))))
        return entered

    def last_batch_scores(self, n):
        out = np.zeros((n, self.nq, NSCORE), dtype=np.int32)
        self._chk(self.L.uvaia_gpu_last_batch_scores(self.ctx, out.ctypes.data_as(C.POINTER(C.c_int)), n))
        return out

    def drain(self):
        """(n_items[q], max_incompatible[q], scores[q, slot, 6], ordinals[q, slot]) in heap layout (slot 0 unused)."""
        n = np.zeros(self.nq, dtype=np.int32)
        T = np.zeros(self.nq, dtype=np.int32)
        sc = np.zeros((self.nq, self.slots + 1, NSCORE), dtype=np.int32)
        od = np.zeros((self.nq, self.slots + 1), dtype=np.int64)
        pi = C.POINTER(C.c_int)
        self._chk(self.L.uvaia_gpu_drain(self.ctx, n.ctypes.data_as(pi), T.ctypes.data_as(pi), sc.ctypes.data_as(pi),
                                         od.ctypes.data_as(C.POINTER(C.c_int64))))
        return n, T, sc, od

    def reset(self):
        self.ordinal = 0
        self._chk(self.L.uvaia_gpu_reset(self.ctx))

    def db_reserve(self, n):
        self._chk(self.L.uvaia_gpu_db_reserve(self.ctx, int(n)))

    def db_append(self, refs, non_n=None):
        _k, nn = _int_ptr(non_n)
        self._chk(self.L.uvaia_gpu_db_append(self.ctx, _cstrs(refs), nn, len(refs)))

    def db_append_block(self, block, non_n=None):
        """block: uint8 array [n, pitch] (pitch >= nchar)."""
        block = np.ascontiguousarray(block, dtype=np.uint8)
        _k, nn = _int_ptr(non_n)
        self._chk(self.L.uvaia_gpu_db_append_block(self.ctx, block.ctypes.data, block.shape[1], nn, block.shape[0]))

    def db_size(self):
        return self.L.uvaia_gpu_db_size(self.ctx)

    def set_active_queries(self, q0, q1):
        self._chk(self.L.uvaia_gpu_set_active_queries(self.ctx, int(q0), int(q1)))

    def max_tolerance(self):
        v = C.c_int(0)
        self._chk(self.L.uvaia_gpu_max_tolerance(self.ctx, C.byref(v)))
        return v.value

    def search_resident_pool(self, first, n, ordinal0, snapshot=-1):
        self._chk(self.L.uvaia_gpu_search_resident_pool(self.ctx, int(first), int(n), int(ordinal0), int(snapshot)))

    def db_clear(self):
        self._chk(self.L.uvaia_gpu_db_clear(self.ctx))

    # ---- reference shards (several GPUs): see include/uvaia_gpu.h
    def db_set_shard(self, rank, world, piece_refs):
        self._chk(self.L.uvaia_gpu_db_set_shard(self.ctx, int(rank), int(world), int(piece_refs)))

    def shard_rows(self):
        return self.L.uvaia_gpu_shard_rows(self.ctx)

    def shard_scan(self, first, n, cnt_ptr, tmin_ptr, aux_ptr):
        self._chk(self.L.uvaia_gpu_shard_scan(self.ctx, int(first), int(n), C.c_void_p(cnt_ptr), C.c_void_p(tmin_ptr), C.c_void_p(aux_ptr)))

    def shard_aux_bytes(self, n_tiles):
        return int(self.L.uvaia_gpu_shard_aux_bytes(self.ctx, int(n_tiles)))

    def db_skip(self, n_ref):
        """the next n_ref references of the stream belong to other ranks' pieces"""
        self._chk(self.L.uvaia_gpu_db_skip(self.ctx, int(n_ref)))

    def shard_ipc_handles(self):
        """bytes that let another process map this context's packed planes and side rows (uvaia_gpu_shard_ipc_open there)"""
        buf = C.create_string_buffer(self.L.uvaia_gpu_shard_ipc_handle_bytes())
        self._chk(self.L.uvaia_gpu_shard_ipc_handles(self.ctx, buf))
        return buf.raw

    def shard_ipc_close(self):
        self._chk(self.L.uvaia_gpu_shard_ipc_close(self.ctx))

    def shard_ipc_open(self, rank, handles):
        self._chk(self.L.uvaia_gpu_shard_ipc_open(self.ctx, int(rank), C.c_char_p(bytes(handles))))

    def scan_wait(self):
        self._chk(self.L.uvaia_gpu_scan_wait(self.ctx))

    def replay_wait(self):
        self._chk(self.L.uvaia_gpu_replay_wait(self.ctx))

    def set_snapshot(self, v):
        self._chk(self.L.uvaia_gpu_set_snapshot(self.ctx, int(v)))

    SCANS, REPLAYS = 0, 1

    def mark(self, what, slot):
        self._chk(self.L.uvaia_gpu_mark(self.ctx, int(what), int(slot)))

    def stream_wait_mark(self, stream, slot):
        self._chk(self.L.uvaia_gpu_stream_wait_mark(self.ctx, C.c_void_p(stream), int(slot)))

    def wait_stream(self, what, stream):
        self._chk(self.L.uvaia_gpu_wait_stream(self.ctx, int(what), C.c_void_p(stream)))

    def shard_replay(self, cnt_ptr, tmin_ptr, aux_ptr, owner, first, n, ordinal0, q0, q1):
        self._chk(self.L.uvaia_gpu_shard_replay(self.ctx, C.c_void_p(cnt_ptr), C.c_void_p(tmin_ptr), C.c_void_p(aux_ptr), int(owner), int(first), int(n), int(ordinal0), int(q0), int(q1)))

    def db_rederive(self):
        """Rebuild the query-set-dependent planes of the whole resident database (asynchronous)."""
        self._chk(self.L.uvaia_gpu_db_rederive(self.ctx))

    def db_export(self, first_tile=0, n_tiles=None):
        """Packed interchange form of the resident database: (planes uint8 [n_tiles, tile_bytes], non_n int32 [n_tiles*64],
        side_rows int32 [n_tiles*64, row_ints])."""
        if n_tiles is None:
            n_tiles = (self.db_size() + 63) // 64 - first_tile
        tb, ri = self.L.uvaia_gpu_db_tile_bytes(self.ctx), self.L.uvaia_gpu_db_side_row_ints()
        planes = np.zeros((n_tiles, tb), dtype=np.uint8)
        non_n = np.zeros(n_tiles * 64, dtype=np.int32)
        side = np.zeros((n_tiles * 64, ri), dtype=np.int32)
        self._chk(self.L.uvaia_gpu_db_export(self.ctx, first_tile, n_tiles, planes.ctypes.data, non_n.ctypes.data_as(C.POINTER(C.c_int)),
                                             side.ctypes.data_as(C.POINTER(C.c_int))))
        return planes, non_n, side

    def db_append_packed(self, planes, non_n, side_rows, n_ref):
        planes = np.ascontiguousarray(planes, dtype=np.uint8)
        non_n = np.ascontiguousarray(non_n, dtype=np.int32)
        side_rows = np.ascontiguousarray(side_rows, dtype=np.int32)
        self._chk(self.L.uvaia_gpu_db_append_packed(self.ctx, planes.ctypes.data, non_n.ctypes.data_as(C.POINTER(C.c_int)),
                                                    side_rows.ctypes.data_as(C.POINTER(C.c_int)), int(n_ref)))

    def search_resident(self, pool, ordinal0=0, want_entered=True):
        ent, p = None, None
        if want_entered:
            ent = np.zeros(self.db_size(), dtype=np.uint8)
            p = ent.ctypes.data_as(C.POINTER(C.c_uint8))
        self._chk(self.L.uvaia_gpu_search_resident(self.ctx, int(pool), int(ordinal0), p))
        return ent

    def sync(self):
        self._chk(self.L.uvaia_gpu_sync(self.ctx))

    def scan_stats(self, reset=False):
        ms, n, b = C.c_double(), C.c_longlong(), C.c_double()
        self._chk(self.L.uvaia_gpu_scan_stats(self.ctx, C.byref(ms), C.byref(n), C.byref(b), int(reset)))
        return ms.value, n.value, b.value

    # ---- ring mode (several GPUs): see include/uvaia_gpu.h
    def state_bytes(self):
        return self.L.uvaia_gpu_state_bytes(self.ctx)

    def state_export(self, ptr):
        self._chk(self.L.uvaia_gpu_state_export(self.ctx, C.c_void_p(ptr)))

    def state_import(self, ptr):
        self._chk(self.L.uvaia_gpu_state_import(self.ctx, C.c_void_p(ptr)))

    def slice_scan(self, first, n, buf):
        self._chk(self.L.uvaia_gpu_slice_scan(self.ctx, int(first), int(n), int(buf)))

    def slice_replay(self, buf, ordinal0, stripe_start):
        self._chk(self.L.uvaia_gpu_slice_replay(self.ctx, int(buf), int(ordinal0), int(bool(stripe_start))))

    def state_range_bytes(self, q0, q1):
        return self.L.uvaia_gpu_state_range_bytes(self.ctx, int(q0), int(q1))

    def state_export_range(self, ptr, q0, q1):
        self._chk(self.L.uvaia_gpu_state_export_range(self.ctx, C.c_void_p(ptr), int(q0), int(q1)))

    def state_import_range(self, ptr, q0, q1):
        self._chk(self.L.uvaia_gpu_state_import_range(self.ctx, C.c_void_p(ptr), int(q0), int(q1)))

    def slice_replay_range(self, buf, ordinal0, q0, q1, take_snapshot):
        self._chk(self.L.uvaia_gpu_slice_replay_range(self.ctx, int(buf), int(ordinal0), int(q0), int(q1), int(bool(take_snapshot))))

    def entered_flags(self, clear=False):
        ent = np.zeros(self.db_size(), dtype=np.uint8)
        self._chk(self.L.uvaia_gpu_entered_flags(self.ctx, ent.ctypes.data_as(C.POINTER(C.c_uint8)), int(clear)))
        return ent

    def replay_stats(self, reset=False):
        """(admissions, on-demand evaluations, dense rescans) since the last reset."""
        out = (C.c_ulonglong * 3)()
        self._chk(self.L.uvaia_gpu_replay_stats(self.ctx, out, int(reset)))
        return int(out[0]), int(out[1]), int(out[2])

    def replay_timing(self, reset=False):
        """ticks of an engine built with -DREPLAY_TIMING (see include/uvaia_gpu.h); zeros otherwise"""
        out = (C.c_ulonglong * 12)()
        self._chk(self.L.uvaia_gpu_replay_timing(self.ctx, out, int(reset)))
        return [int(x) for x in out]

    def replay_tiles_opened(self, reset=False):
        """(query, tile) pairs the replay of the packed-plane scan opened since the last reset."""
        out = C.c_ulonglong(0)
        self._chk(self.L.uvaia_gpu_replay_tiles_opened(self.ctx, C.byref(out), int(reset)))
        return int(out.value)

    def set_query_tile(self, qt):
        self._chk(self.L.uvaia_gpu_set_query_tile(self.ctx, int(qt)))

    def scan_bytes_per_ref(self):
        return self.L.uvaia_gpu_scan_bytes_per_ref(self.ctx)

    def derived_bytes_per_ref(self):
        return self.L.uvaia_gpu_derived_bytes_per_ref(self.ctx)

    def scan_variant(self):
        return self.L.uvaia_gpu_scan_variant(self.ctx)

    def packed_bytes_per_ref(self):
        return self.L.uvaia_gpu_packed_bytes_per_ref(self.ctx)

    def agree_on_polymorphic(self, seqs):
        """uint8 [len(seqs), n_query]: 1 where a sequence and a query differ at no polymorphic column (redundancy test)."""
        out = np.zeros((len(seqs), self.nq), dtype=np.uint8)
        self._chk(self.L.uvaia_gpu_agree_on_polymorphic(self.ctx, _cstrs(seqs), len(seqs), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def ball(self, refs, radius):
        md = np.zeros(len(refs), dtype=np.int32)
        self._chk(self.L.uvaia_gpu_ball(self.ctx, _cstrs(refs), len(refs), int(radius), md.ctypes.data_as(C.POINTER(C.c_int))))
        return md

    def ball_resident(self, radius, first=0, n=None, want=True):
        n = self.db_size() - first if n is None else n
        md = np.zeros(n, dtype=np.int32) if want else None
        self._chk(self.L.uvaia_gpu_ball_resident(self.ctx, int(first), int(n), int(radius), md.ctypes.data_as(C.POINTER(C.c_int)) if want else None))
        return md

    def ball_asked(self, reset=False):
        return int(self.L.uvaia_gpu_ball_asked(self.ctx, int(reset)))

    def query_table(self, which):
        """uvaia_gpu_export_query_table: the table as bytes (numpy uint8)"""
        n = C.c_size_t(0)
        self._chk(self.L.uvaia_gpu_export_query_table(self.ctx, int(which), None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint8)
        if n.value:
            self._chk(self.L.uvaia_gpu_export_query_table(self.ctx, int(which), out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out

    def ball_kernel_ms(self, reset=False):
        """Device ms since the last reset of (consensus pass, gather of the asked references, pair scan)."""
        out = (C.c_double * 3)()
        self.L.uvaia_gpu_ball_kernel_ms(self.ctx, out, int(reset))
        return tuple(out)


def finalise_heaps(n, scores, ordinals):
    """heap_finalise_heap_qsort (src/min_heap.c:149-158) on drained heaps: per query the rows best-first.

    The reference moves slots [1..n] to [0..n-1] by swapping slot 0 with slot n and then calls glibc qsort, which is
    a stable merge sort; a stable sort over that exact starting order gives the same table on any libc."""
    out = []
    for q in range(len(n)):
        k = int(n[q])
        order = ([k] + list(range(1, k))) if k else []          # slot n moved to position 0
        rows = [(tuple(int(x) for x in scores[q, s]), int(ordinals[q, s])) for s in order]
        rows.sort(key=lambda r: tuple(-v for v in r[0]))        # list.sort is stable
        out.append(rows)
    return out


class Group:
    """Several contexts driven from one host thread (uvaia_gpu_group_*): the reference-shard search with peer copies as the exchange."""

    def __init__(self, q, devices, nbest=100, max_pool=4096, piece_refs=0):
        self.L = load_library()
        self.nq, self.nchar = len(q.seqs), len(q.consensus)
        self._keep = [_cstrs(q.seqs), q.consensus,
                      (C.c_size_t * len(q.idx_c))(*[int(x) for x in q.idx_c]),
                      (C.c_size_t * len(q.idx_m))(*[int(x) for x in q.idx_m]),
                      (C.c_size_t * len(q.idx))(*[int(x) for x in q.idx])]
        qq = _Query(self.nq, self.nchar, self._keep[0], q.consensus, self._keep[2], self._keep[3], self._keep[4],
                    len(q.idx_c), len(q.idx_m), len(q.idx), int(q.trim), int(bool(q.acgt)))
        self.g = C.c_void_p()
        dev = (C.c_int * len(devices))(*[int(d) for d in devices])
        rc = self.L.uvaia_gpu_group_open(C.byref(self.g), C.byref(qq), int(nbest), dev, len(devices), int(max_pool), int(piece_refs))
        if rc != 0:
            msg = self.L.uvaia_gpu_group_last_error(None)
            self.g = None
            raise GpuError(rc, msg.decode() if msg else "?")
        self.slots = self.L.uvaia_gpu_heap_slots(self.L.uvaia_gpu_group_member(self.g, 0))

    def _chk(self, rc):
        if rc != 0:
            msg = self.L.uvaia_gpu_group_last_error(self.g)
            raise GpuError(rc, msg.decode() if msg else "?")

    def close(self):
        if getattr(self, "g", None):
            self.L.uvaia_gpu_group_close(self.g)
            self.g = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def query_shard(self, i):
        a, b = C.c_int(), C.c_int()
        self._chk(self.L.uvaia_gpu_group_query_shard(self.g, i, C.byref(a), C.byref(b)))
        return a.value, b.value

    def member_bytes_per_ref(self):
        """(packed planes, planes derived for the query set without V) bytes per reference a member keeps for its own pieces"""
        self.L.uvaia_gpu_packed_bytes_per_ref.restype = C.c_size_t
        self.L.uvaia_gpu_derived_bytes_per_ref.restype = C.c_size_t
        m0 = C.c_void_p(self.L.uvaia_gpu_group_member(self.g, 0))
        return int(self.L.uvaia_gpu_packed_bytes_per_ref(m0)), int(self.L.uvaia_gpu_derived_bytes_per_ref(m0))

    def db_reserve(self, n):
        self._chk(self.L.uvaia_gpu_group_db_reserve(self.g, int(n)))

    def db_append(self, refs, non_n=None):
        _k, nn = _int_ptr(non_n)
        self._chk(self.L.uvaia_gpu_group_db_append(self.g, _cstrs(refs), nn, len(refs)))

    def db_rederive(self):
        self._chk(self.L.uvaia_gpu_group_db_rederive(self.g))

    def reset(self):
        self._chk(self.L.uvaia_gpu_group_reset(self.g))

    def sync(self):
        self._chk(self.L.uvaia_gpu_group_sync(self.g))

    def search_resident(self, pool, ordinal0=0, want_entered=True):
        ent, p = None, None
        if want_entered:
            ent = np.zeros(self.L.uvaia_gpu_group_db_size(self.g), dtype=np.uint8)
            p = ent.ctypes.data_as(C.POINTER(C.c_uint8))
        self._chk(self.L.uvaia_gpu_group_search_resident(self.g, int(pool), int(ordinal0), p))
        return ent

    def push(self, refs, non_n=None, ordinal0=0):
        entered = np.zeros(len(refs), dtype=np.uint8)
        _k, nn = _int_ptr(non_n)
        self._chk(self.L.uvaia_gpu_group_push(self.g, _cstrs(refs), nn, len(refs), int(ordinal0), entered.ctypes.data_as(C.POINTER(C.c_uint8))))
        return entered

    def drain(self):
        n = np.zeros(self.nq, dtype=np.int32)
        T = np.zeros(self.nq, dtype=np.int32)
        sc = np.zeros((self.nq, self.slots + 1, NSCORE), dtype=np.int32)
        od = np.zeros((self.nq, self.slots + 1), dtype=np.int64)
        pi = C.POINTER(C.c_int)
        self._chk(self.L.uvaia_gpu_group_drain(self.g, n.ctypes.data_as(pi), T.ctypes.data_as(pi), sc.ctypes.data_as(pi), od.ctypes.data_as(C.POINTER(C.c_int64))))
        return n, T, sc, od


def query_columns(seqs, trim=0, acgt=False, device=-1):
    """create_query_indices' column walk on the device (uvaia_gpu_query_columns): (consensus bytes, some_missing uint8 array)."""
    L = load_library()
    n, nchar = len(seqs), len(seqs[0])
    arr = (C.c_char_p * n)(*seqs)
    cons = C.create_string_buffer(nchar)
    miss = np.zeros(nchar, dtype=np.uint8)
    rc = L.uvaia_gpu_query_columns(arr, n, nchar, int(trim), int(bool(acgt)), int(device), cons, miss.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc:
        raise GpuError(rc, (L.uvaia_gpu_last_error(None) or b"").decode())
    return cons.raw[:nchar], miss
