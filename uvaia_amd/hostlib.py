"""ctypes binding of the host C library (uvaia_amd/csrc/host): query preparation, heaps, FASTA reader, generator.

Like capi.py this is plumbing for tests and bench.py: all logic lives in the C sources.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "lib", "libuvaia_host.so")


class CharVector(C.Structure):
    _fields_ = [("string", C.POINTER(C.c_char_p)), ("nchars", C.POINTER(C.c_size_t)), ("nstrings", C.c_int)]


class Alignment(C.Structure):
    _fields_ = [("ntax", C.c_int), ("nchar", C.c_int), ("character", C.POINTER(CharVector)), ("taxlabel", C.POINTER(CharVector)),
                ("taxlabel_hash", C.c_void_p), ("filename", C.c_char_p)]


class QueryStruct(C.Structure):
    _fields_ = [("aln", C.POINTER(Alignment)), ("consensus", C.POINTER(C.c_char)),
                ("idx_c", C.POINTER(C.c_size_t)), ("idx_m", C.POINTER(C.c_size_t)), ("idx", C.POINTER(C.c_size_t)),
                ("trim", C.c_size_t), ("n_idx_c", C.c_int), ("n_idx_m", C.c_int), ("n_idx", C.c_int), ("dist", C.c_int),
                ("acgt", C.c_bool)]


class QItem(C.Structure):
    _fields_ = [("score", C.c_int * 6), ("name", C.c_char_p)]


class HeapStruct(C.Structure):
    _fields_ = [("seq", C.POINTER(QItem)), ("heap_size", C.c_int), ("n", C.c_int), ("max_incompatible", C.c_int)]


class ReadFasta(C.Structure):
    _fields_ = [("seqfile", C.c_void_p), ("line_read", C.c_void_p), ("next_name", C.c_void_p),
                ("name", C.c_char_p), ("seq", C.c_void_p), ("linelength", C.c_size_t), ("seqlength", C.c_size_t), ("newseq", C.c_bool)]


_lib = None


def build_library():
    capi.build_library()
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc", "host"), "-s"])
    return _LIB


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise RuntimeError("host library %s is missing: run __graft_entry__.build()" % _LIB)
    capi.load_library()          # libuvaia_host.so links libuvaia_gpu.so
    L = C.CDLL(_LIB)
    pp = C.POINTER(C.c_char_p)
    qp = C.POINTER(QueryStruct)
    hp = C.POINTER(HeapStruct)
    L.uvaia_prepare_query_from_arrays.restype = qp
    L.uvaia_prepare_query_from_arrays.argtypes = [C.c_int, C.c_int, pp, pp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
    L.del_query_structure.argtypes = [qp]
    L.uvaia_gpu_open_query.restype = C.c_int
    L.uvaia_gpu_open_query.argtypes = [C.POINTER(C.c_void_p), qp, C.c_int, C.c_int, C.c_size_t]
    L.uvaia_gpu_open_query_tuned.restype = C.c_int
    L.uvaia_gpu_open_query_tuned.argtypes = [C.POINTER(C.c_void_p), qp, C.c_int, C.c_int, C.c_size_t, C.POINTER(capi.Tuning)]
    L.uvaia_set_prune_mode.argtypes = [C.c_int]
    L.new_heap_t.restype = hp
    L.new_heap_t.argtypes = [C.c_int]
    L.del_heap_t.argtypes = [hp]
    L.heap_insert.restype = C.c_bool
    L.heap_insert.argtypes = [hp, QItem]
    L.heap_finalise_heap_qsort.argtypes = [hp]
    L.new_readfasta.restype = C.POINTER(ReadFasta)
    L.new_readfasta.argtypes = [C.c_char_p]
    L.readfasta_next.restype = C.c_int
    L.readfasta_next.argtypes = [C.POINTER(ReadFasta)]
    L.del_readfasta.argtypes = [C.POINTER(ReadFasta)]
    L.quick_count_sequence_non_N.restype = C.c_int
    L.quick_count_sequence_non_N.argtypes = [C.c_char_p, C.c_size_t]
    L.quick_pairwise_score_acgt_and_valid.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.seq_ball_against_query_structure.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, qp]
    L.uvaia_set_prepare_device.argtypes = [C.c_int]
    L.uvaia_synth_new.restype = C.c_void_p
    L.uvaia_synth_new.argtypes = [C.c_int, C.c_uint64, C.c_int]
    L.uvaia_synth_free.argtypes = [C.c_void_p]
    L.uvaia_synth_generate.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
    _lib = L
    return L


def _cstrs(strs):
    arr = (C.c_char_p * len(strs))()
    for i, s in enumerate(strs):
        arr[i] = s if isinstance(s, bytes) else s.encode()
    return arr


class PreparedQuery:
    """query_t as built by the host C code (src/nearest.c:203-224 order of operations)."""

    def __init__(self, seqs, names, trim=0, dist=1, acgt=False, ambig_q=0.5, keep_resolved=False, is_ball=False):
        L = load_library()
        nchar = len(seqs[0])
        assert all(len(s) == nchar for s in seqs), "query sequences must be aligned"
        self._L = L
        self.ptr = L.uvaia_prepare_query_from_arrays(len(seqs), nchar, _cstrs(seqs), _cstrs(names), trim, dist, int(acgt),
                                                     ambig_q, int(keep_resolved), int(is_ball))
        q = self.ptr.contents
        a = q.aln.contents
        self.ntax, self.nchar, self.acgt, self.trim, self.dist = a.ntax, a.nchar, bool(q.acgt), q.trim, q.dist
        ch, tl = a.character.contents, a.taxlabel.contents
        self.seqs = [C.string_at(ch.string[i], a.nchar) for i in range(a.ntax)]
        self.names = [tl.string[i].decode() for i in range(a.ntax)]
        if a.ntax:
            self.consensus = C.string_at(q.consensus, a.nchar)
            self.idx_c = np.array([q.idx_c[i] for i in range(q.n_idx_c)], dtype=np.int64)
            self.idx_m = np.array([q.idx_m[i] for i in range(q.n_idx_m)], dtype=np.int64)
            self.idx = np.array([q.idx[i] for i in range(q.n_idx)], dtype=np.int64)

    def open_engine(self, nbest=100, max_pool=4096, device=-1, tuning=None):
        """uvaia_gpu_open_query(): an Engine over this query set."""
        eng = capi.Engine.__new__(capi.Engine)
        eng.L = capi.load_library()
        eng.nq, eng.nchar = self.ntax, self.nchar
        eng.ctx = C.c_void_p()
        tn = capi.Tuning.make(tuning)
        rc = self._L.uvaia_gpu_open_query_tuned(C.byref(eng.ctx), self.ptr, int(nbest), int(device), int(max_pool), C.byref(tn) if tn is not None else None)
        if rc != 0:
            msg = eng.L.uvaia_gpu_last_error(None)
            eng.ctx = None
            raise capi.GpuError(rc, msg.decode() if msg else "?")
        eng.slots = eng.L.uvaia_gpu_heap_slots(eng.ctx)
        eng.ordinal = 0
        eng._keep = self
        return eng

    def seq_ball(self, seq, radius):
        """seq_ball_against_query_structure (src/fastaseq.h:78) for one sequence: what the reference leaves in *min_dist"""
        one = (C.c_char_p * 1)(seq)
        md = C.c_int(0)
        self._L.seq_ball_against_query_structure(one, C.byref(md), int(radius), self.ptr)
        return md.value

    def __del__(self):
        try:
            self._L.del_query_structure(self.ptr)
        except Exception:
            pass


def score_acgt_and_valid(s1, s2, idx, maxdist=2 ** 31 - 1):
    """quick_pairwise_score_acgt_and_valid (src/fastaseq.h:74) over the sites idx"""
    L = load_library()
    arr = (C.c_size_t * len(idx))(*[int(i) for i in idx])
    out = (C.c_int * 2)()
    L.quick_pairwise_score_acgt_and_valid(s1, s2, len(idx), int(maxdist), out, arr)
    return list(out)


def set_prune_mode(mode):
    """where exclude_redundant_query_sequences' pair test runs: "auto" (device from 512 queries), "host", "device" """
    load_library().uvaia_set_prune_mode({"auto": 0, "host": 1, "device": 2}[mode])


class Synth:
    """Deterministic SARS-CoV-2-shaped sequence generator (csrc/host/synth.c)."""

    def __init__(self, nchar=29903, seed=20241008, preset=0):
        self.L = load_library()
        self.nchar = nchar
        self.h = self.L.uvaia_synth_new(nchar, seed, preset)
        if not self.h:
            raise ValueError("bad generator parameters")

    def generate(self, first_index, n, pitch=None):
        """uint8 array [n, pitch] (pitch defaults to nchar) and the valid-site counts."""
        pitch = pitch or self.nchar
        rows = np.empty((n, pitch), dtype=np.uint8)
        if pitch > self.nchar:
            rows[:, self.nchar:] = ord("N")
        non_n = np.empty(n, dtype=np.int32)
        self.L.uvaia_synth_generate(self.h, int(first_index), int(n), rows.ctypes.data, pitch, non_n.ctypes.data_as(C.POINTER(C.c_int)))
        return rows, non_n

    def generate_bytes(self, first_index, n):
        rows, non_n = self.generate(first_index, n)
        return [rows[i].tobytes() for i in range(n)], non_n

    def __del__(self):
        try:
            self.L.uvaia_synth_free(self.h)
        except Exception:
            pass
