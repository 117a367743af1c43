"""Oracle-backed stand-in for the GPU engine's ring-mode entry points (tests only): lets the multi-rank protocol of
uvaia_amd/ring.py run on CPU ranks over gloo."""
import ctypes as C

import numpy as np

import oracle_lib as O


class NumpyStateBuffer:
    def __init__(self, nbytes):
        import torch
        self.arr = np.zeros((nbytes + 3) // 4, dtype=np.int32)
        self.tensor = torch.from_numpy(self.arr)
        self.ptr = self.arr.ctypes.data


class OracleRingEngine:
    def __init__(self, query, local_refs, nbest, max_slice):
        L = O.lib()
        L.orc_search_process_slice.restype = C.c_int
        L.orc_search_process_slice.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_int]
        L.orc_search_last_snapshot.restype = C.c_int
        L.orc_search_last_snapshot.argtypes = [C.c_void_p]
        L.orc_search_state_ints.restype = C.c_size_t
        L.orc_search_state_ints.argtypes = [C.c_void_p]
        L.orc_search_get_state.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_search_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
        L.orc_search_process_slice_range.restype = C.c_int
        L.orc_search_process_slice_range.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_int]
        L.orc_search_max_T.restype = C.c_int
        L.orc_search_max_T.argtypes = [C.c_void_p]
        L.orc_search_get_state_range.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_search_set_state_range.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        self.L, self.q, self.refs = L, query, local_refs
        self.s = L.orc_search_new(query.ptr, max_slice, nbest, 1.0, 0)
        self.pending = {}
        self.snap = -1

    def state_bytes(self):
        return 4 * self.L.orc_search_state_ints(self.s)

    def slice_scan(self, first, n, buf):
        self.pending[buf] = (first, n)

    def slice_replay(self, buf, ordinal0, stripe_start):
        first, n = self.pending[buf]
        seqs = self.refs[first:first + n]
        ords = (C.c_int64 * n)(*range(ordinal0, ordinal0 + n))
        names = O._cstr_array(["r%d" % o for o in range(ordinal0, ordinal0 + n)])
        rc = self.L.orc_search_process_slice(self.s, n, O._cstr_array(seqs), names, ords, -1 if stripe_start else self.snap)
        assert rc == 0
        self.snap = self.L.orc_search_last_snapshot(self.s)

    # ---- query-group variants
    def state_range_bytes(self, q0, q1):
        per_q = (self.L.orc_search_state_ints(self.s) - 1) // self.q.ntax
        return 4 * (1 + per_q * (q1 - q0))

    def slice_replay_range(self, buf, ordinal0, q0, q1, take_snapshot):
        first, n = self.pending[buf]
        if take_snapshot:
            self.snap = self.L.orc_search_max_T(self.s)
        seqs = self.refs[first:first + n]
        ords = (C.c_int64 * n)(*range(ordinal0, ordinal0 + n))
        names = O._cstr_array(["r%d" % o for o in range(ordinal0, ordinal0 + n)])
        rc = self.L.orc_search_process_slice_range(self.s, n, O._cstr_array(seqs), names, ords, self.snap, q0, q1)
        assert rc == 0

    def state_export_range(self, ptr, q0, q1):
        self.L.orc_search_get_state_range(self.s, C.c_void_p(ptr), q0, q1)
        C.cast(ptr, C.POINTER(C.c_int))[0] = self.snap

    def state_import_range(self, ptr, q0, q1):
        self.L.orc_search_set_state_range(self.s, C.c_void_p(ptr), b"r", q0, q1)
        self.snap = C.cast(ptr, C.POINTER(C.c_int))[0]

    def state_export(self, ptr):
        self.L.orc_search_get_state(self.s, C.c_void_p(ptr))

    def state_import(self, ptr):
        self.L.orc_search_set_state(self.s, C.c_void_p(ptr), b"r")
        self.snap = C.cast(ptr, C.POINTER(C.c_int))[0]

    def result(self):
        L = self.L
        L.orc_search_finish(self.s)
        rows, T = [], []
        for iq in range(self.q.ntax):
            hn = L.orc_search_heap_n(self.s, iq)
            rows.append([(tuple(L.orc_search_row(self.s, iq, r).contents.score), L.orc_search_row(self.s, iq, r).contents.ordinal) for r in range(hn)])
            T.append(L.orc_search_final_T(self.s, iq))
        return rows, T
