"""ctypes binding of include/uvaia_align.h (tests and bench.py reach the wavefront aligner of `uvaialign` through it)."""
import ctypes as C

import numpy as np

from . import capi

# every symbol include/uvaia_align.h declares (tests check the library exports all of them)
SYMBOLS = [
    "uvaia_align_default_options", "uvaia_align_open", "uvaia_align_close", "uvaia_align_last_error", "uvaia_align_batch",
    "uvaia_align_load", "uvaia_align_load_block", "uvaia_align_run", "uvaia_align_sync", "uvaia_align_fetch", "uvaia_align_stats",
]


class AlignError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("uvaia_align error %d: %s" % (code, msg))
        self.code = code


class Options(C.Structure):
    _fields_ = [("mismatch", C.c_int), ("gap_opening", C.c_int), ("gap_extension", C.c_int), ("min_wavefront_length", C.c_int),
                ("max_distance_threshold", C.c_int), ("workspace_bytes", C.c_size_t), ("max_blocks", C.c_int)]


_ready = False


def _lib():
    global _ready
    L = capi.load_library()
    if not _ready:
        vp, pi = C.c_void_p, C.POINTER(C.c_int)
        L.uvaia_align_default_options.argtypes = [C.POINTER(Options)]
        L.uvaia_align_default_options.restype = None
        L.uvaia_align_open.argtypes = [C.POINTER(vp), C.c_char_p, C.c_int, C.c_int, C.POINTER(Options)]
        L.uvaia_align_close.argtypes = [vp]
        L.uvaia_align_close.restype = None
        L.uvaia_align_last_error.argtypes = [vp]
        L.uvaia_align_last_error.restype = C.c_char_p
        L.uvaia_align_batch.argtypes = [vp, C.POINTER(C.c_char_p), pi, C.c_int, C.c_void_p, pi]
        L.uvaia_align_load.argtypes = [vp, C.POINTER(C.c_char_p), pi, C.c_int]
        L.uvaia_align_load_block.argtypes = [vp, C.c_void_p, C.POINTER(C.c_int64), C.c_int]
        L.uvaia_align_run.argtypes = [vp]
        L.uvaia_align_sync.argtypes = [vp]
        L.uvaia_align_fetch.argtypes = [vp, C.c_void_p, pi]
        L.uvaia_align_stats.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_double), pi, C.POINTER(C.c_double)]
        _ready = True
    return L


def default_options():
    o = Options()
    _lib().uvaia_align_default_options(C.byref(o))
    return o


class Aligner:
    """One reference sequence on one GPU (new_queue, src/align.c:286-313)."""

    def __init__(self, ref, device=0, **options):
        self.L = _lib()
        self.ref_len = len(ref)
        opt = default_options()
        for k, v in options.items():
            setattr(opt, k, v)
        self.ptr = C.c_void_p()
        rc = self.L.uvaia_align_open(C.byref(self.ptr), ref, len(ref), device, C.byref(opt))
        if rc:
            raise AlignError(rc, (self.L.uvaia_align_last_error(None) or b"").decode())
        self.n = 0

    def _chk(self, rc):
        if rc:
            raise AlignError(rc, (self.L.uvaia_align_last_error(self.ptr) or b"").decode())

    def close(self):
        if self.ptr:
            self.L.uvaia_align_close(self.ptr)
            self.ptr = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def load(self, seqs):
        """seqs: list of bytes; they stay resident until the next load"""
        n = len(seqs)
        off = np.zeros(n + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s) for s in seqs])
        blob = b"".join(seqs)
        self._chk(self.L.uvaia_align_load_block(self.ptr, blob, off.ctypes.data_as(C.POINTER(C.c_int64)), n))
        self.n = n

    def run(self):
        self._chk(self.L.uvaia_align_run(self.ptr))

    def fetch(self):
        rows = np.zeros((self.n, self.ref_len + 1), dtype=np.uint8)
        score = np.zeros(self.n, dtype=np.int32)
        self._chk(self.L.uvaia_align_fetch(self.ptr, rows.ctypes.data, score.ctypes.data_as(C.POINTER(C.c_int))))
        return score, rows[:, :self.ref_len]

    def align(self, seqs):
        """(scores [n], aligned rows [n, ref_len]) through uvaia_align_batch, the reference-shaped call"""
        n = len(seqs)
        arr = (C.c_char_p * n)(*seqs)
        lens = (C.c_int * n)(*[len(s) for s in seqs])
        rows = np.zeros((n, self.ref_len + 1), dtype=np.uint8)
        score = np.zeros(n, dtype=np.int32)
        self._chk(self.L.uvaia_align_batch(self.ptr, arr, lens, n, rows.ctypes.data, score.ctypes.data_as(C.POINTER(C.c_int))))
        self.n = n
        return score, rows[:, :self.ref_len]

    def stats(self):
        cells, nbytes, passes, ms = C.c_ulonglong(0), C.c_double(0), C.c_int(0), C.c_double(0)
        self._chk(self.L.uvaia_align_stats(self.ptr, C.byref(cells), C.byref(nbytes), C.byref(passes), C.byref(ms)))
        return {"cells": cells.value, "wavefront_bytes": nbytes.value, "passes": passes.value, "kernel_ms": ms.value}
