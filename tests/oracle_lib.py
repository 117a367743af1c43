"""ctypes wrapper around oracle/build/liboracle.so (the CPU restatement; test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "build", "liboracle.so")

NSCORE = 6


class OrcItem(C.Structure):
    _fields_ = [("score", C.c_int * NSCORE), ("name", C.c_char_p), ("ordinal", C.c_int64)]


class OrcQuery(C.Structure):
    _fields_ = [
        ("ntax", C.c_int), ("nchar", C.c_int),
        ("seq", C.POINTER(C.c_char_p)), ("name", C.POINTER(C.c_char_p)),
        ("consensus", C.POINTER(C.c_char)),
        ("idx_c", C.POINTER(C.c_size_t)), ("idx_m", C.POINTER(C.c_size_t)), ("idx", C.POINTER(C.c_size_t)),
        ("trim", C.c_size_t),
        ("n_idx_c", C.c_int), ("n_idx_m", C.c_int), ("n_idx", C.c_int), ("dist", C.c_int), ("acgt", C.c_int),
    ]


class WfaPenalties(C.Structure):
    _fields_ = [("match", C.c_int), ("mismatch", C.c_int), ("gap_opening", C.c_int), ("gap_extension", C.c_int)]


UVAIALIGN_PENALTIES = (0, 4, 6, 2)          # src/align.c:305

_lib = None


def build():
    """(Re)build liboracle.so if missing or stale."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("uvaia_oracle.c", "uvaia_oracle.h", "wfa_oracle.c", "wfa_oracle.h", "Makefile")]
    if os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    pp = C.POINTER(C.c_char_p)
    L.orc_query_prepare.restype = C.POINTER(OrcQuery)
    L.orc_query_prepare.argtypes = [C.c_int, C.c_int, pp, pp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
    L.orc_query_del.argtypes = [C.POINTER(OrcQuery)]
    L.orc_search_new.restype = C.c_void_p
    L.orc_search_new.argtypes = [C.POINTER(OrcQuery), C.c_int, C.c_int, C.c_double, C.c_int]
    L.orc_search_del.argtypes = [C.c_void_p]
    L.orc_search_feed.restype = C.c_int
    L.orc_search_feed.argtypes = [C.c_void_p, C.c_int, pp, pp, C.POINTER(C.c_int)]
    L.orc_search_end_of_file.argtypes = [C.c_void_p]
    L.orc_search_finish.argtypes = [C.c_void_p]
    L.orc_search_nrows.restype = C.c_int
    L.orc_search_nrows.argtypes = [C.c_void_p, C.c_int]
    L.orc_search_heap_n.restype = C.c_int
    L.orc_search_heap_n.argtypes = [C.c_void_p, C.c_int]
    L.orc_search_row.restype = C.POINTER(OrcItem)
    L.orc_search_row.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_search_n_saved.restype = C.c_int64
    L.orc_search_n_saved.argtypes = [C.c_void_p]
    L.orc_search_saved_ordinals.restype = C.POINTER(C.c_int64)
    L.orc_search_saved_ordinals.argtypes = [C.c_void_p]
    for f in ("orc_search_n_seen", "orc_search_n_lowqual", "orc_search_n_samename"):
        getattr(L, f).restype = C.c_int64
        getattr(L, f).argtypes = [C.c_void_p]
    L.orc_search_final_T.restype = C.c_int
    L.orc_search_final_T.argtypes = [C.c_void_p, C.c_int]
    L.orc_allpairs_scores.argtypes = [C.POINTER(OrcQuery), C.c_int, pp, C.POINTER(C.c_int)]
    L.orc_ball.argtypes = [C.POINTER(OrcQuery), C.c_double, C.c_int, pp, C.POINTER(C.c_int), C.POINTER(C.c_ubyte)]
    L.orc_score_matches_truncated_idx.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.orc_score_acgt_and_valid.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.orc_count_non_N.restype = C.c_int
    L.orc_count_non_N.argtypes = [C.c_char_p, C.c_size_t]
    L.orc_set_threads.argtypes = [C.c_int]
    L.orc_max_threads.restype = C.c_int
    # uvaialign (wfa_oracle.h)
    L.orc_wfa_new.restype = C.c_void_p
    L.orc_wfa_new.argtypes = [WfaPenalties, C.c_int, C.c_int]
    L.orc_wfa_del.argtypes = [C.c_void_p]
    L.orc_wfa_align.restype = C.c_int
    L.orc_wfa_align.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int]
    L.orc_wfa_cigar.restype = C.POINTER(C.c_char)
    L.orc_wfa_cigar.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.orc_wfa_cells.restype = C.c_int64
    L.orc_wfa_cells.argtypes = [C.c_void_p]
    L.orc_wfa_max_width.restype = C.c_int
    L.orc_wfa_max_width.argtypes = [C.c_void_p]
    L.orc_uvaialign_query.restype = C.c_int
    L.orc_uvaialign_query.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.POINTER(C.c_int64)]
    L.orc_uvaialign_accepts.restype = C.c_int
    L.orc_uvaialign_accepts.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_double]
    L.orc_uvaialign_batch.argtypes = [C.c_char_p, C.c_int, C.c_int, pp, C.POINTER(C.c_int), C.c_char_p, C.POINTER(C.c_int)]
    L.orc_gotoh_score.restype = C.c_int
    L.orc_gotoh_score.argtypes = [WfaPenalties, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    L.orc_cigar_score.restype = C.c_int
    L.orc_cigar_score.argtypes = [WfaPenalties, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    _lib = L
    return L


def _cstr_array(strs):
    arr = (C.c_char_p * len(strs))()
    for i, s in enumerate(strs):
        arr[i] = s if isinstance(s, bytes) else s.encode()
    return arr


def score4(s1, s2, maxdist=2 ** 31 - 1, idx=None):
    """biomcmc 4-count kernel restatement over idx (default: all sites)."""
    L = lib()
    n = len(s1) if idx is None else len(idx)
    idx_arr = (C.c_size_t * n)(*(range(n) if idx is None else idx))
    out = (C.c_int * 4)()
    L.orc_score_matches_truncated_idx(s1, s2, n, maxdist, out, idx_arr)
    return list(out)


def score_acgt(s1, s2, maxdist=2 ** 31 - 1, idx=None):
    L = lib()
    n = len(s1) if idx is None else len(idx)
    idx_arr = (C.c_size_t * n)(*(range(n) if idx is None else idx))
    out = (C.c_int * 2)()
    L.orc_score_acgt_and_valid(s1, s2, n, maxdist, out, idx_arr)
    return list(out)


class Query:
    """Prepared query structure (src/nearest.c:203-224 / src/ball.c:174-194)."""

    def __init__(self, seqs, names, trim=0, dist=1, acgt=False, ambig_q=0.5, keep_resolved=False, is_ball=False):
        L = lib()
        nchar = len(seqs[0])
        assert all(len(s) == nchar for s in seqs)
        self._seqs = _cstr_array(seqs)
        self._names = _cstr_array(names)
        self.ptr = L.orc_query_prepare(len(seqs), nchar, self._seqs, self._names, trim, dist, int(acgt),
                                       ambig_q, int(keep_resolved), int(is_ball))
        q = self.ptr.contents
        self.ntax, self.nchar, self.acgt, self.trim, self.dist = q.ntax, q.nchar, bool(q.acgt), q.trim, q.dist
        self.names = [q.name[i].decode() for i in range(q.ntax)]
        self.seqs = [C.string_at(q.seq[i], q.nchar) for i in range(q.ntax)]
        if q.ntax:
            self.consensus = C.string_at(q.consensus, q.nchar)
            self.idx_c = np.array([q.idx_c[i] for i in range(q.n_idx_c)], dtype=np.int64)
            self.idx_m = np.array([q.idx_m[i] for i in range(q.n_idx_m)], dtype=np.int64)
            self.idx = np.array([q.idx[i] for i in range(q.n_idx)], dtype=np.int64)

    def __del__(self):
        try:
            lib().orc_query_del(self.ptr)
        except Exception:
            pass

    def allpairs(self, refs):
        """Untruncated S[6] for every (ref, query): int32 [n_ref, ntax, 6]."""
        out = np.zeros((len(refs), self.ntax, NSCORE), dtype=np.int32)
        lib().orc_allpairs_scores(self.ptr, len(refs), _cstr_array(refs), out.ctypes.data_as(C.POINTER(C.c_int)))
        return out

    def ball(self, refs, ambig_r=0.5):
        md = np.zeros(len(refs), dtype=np.int32)
        keep = np.zeros(len(refs), dtype=np.uint8)
        lib().orc_ball(self.ptr, ambig_r, len(refs), _cstr_array(refs), md.ctypes.data_as(C.POINTER(C.c_int)),
                       keep.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return md, keep


class SearchResult:
    def __init__(self):
        self.rows = []        # per query: list of (ordinal, name, [6 scores]) best-first, as the CSV prints them
        self.heap_n = []
        self.saved = None     # ordinals dumped to the .aln, in stream order
        self.final_T = []
        self.n_seen = self.n_lowqual = self.n_samename = 0


def search(query, refs, names, pool=64, nbest=100, ambig_r=0.5, exclude_self=False, file_breaks=(), threads=None,
           chunk=4096):
    """Run the restated src/nearest.c main loop over refs (bytes, upper-case) in order."""
    L = lib()
    if threads:
        L.orc_set_threads(threads)
    s = L.orc_search_new(query.ptr, pool, nbest, ambig_r, int(exclude_self))
    breaks = set(file_breaks)
    try:
        i = 0
        n = len(refs)
        cuts = sorted(b for b in breaks if 0 < b < n) + [n]
        for cut in cuts:
            while i < cut:
                j = min(cut, i + chunk)
                lens = (C.c_int * (j - i))(*[len(r) for r in refs[i:j]])
                rc = L.orc_search_feed(s, j - i, _cstr_array(refs[i:j]), _cstr_array(names[i:j]), lens)
                if rc != 0:
                    raise ValueError("all sequences must be aligned")
                i = j
            if cut != n:
                L.orc_search_end_of_file(s)
        L.orc_search_finish(s)
        res = SearchResult()
        for iq in range(query.ntax):
            hn = L.orc_search_heap_n(s, iq)
            rows = []
            for r in range(hn):
                it = L.orc_search_row(s, iq, r).contents
                rows.append((it.ordinal, it.name.decode() if it.name else None, list(it.score)))
            res.rows.append(rows)
            res.heap_n.append(hn)
            res.final_T.append(L.orc_search_final_T(s, iq))
        ns = L.orc_search_n_saved(s)
        so = L.orc_search_saved_ordinals(s)
        res.saved = np.array([so[k] for k in range(ns)], dtype=np.int64)
        res.n_seen = L.orc_search_n_seen(s)
        res.n_lowqual = L.orc_search_n_lowqual(s)
        res.n_samename = L.orc_search_n_samename(s)
        return res
    finally:
        L.orc_search_del(s)


# ---- uvaialign (oracle/wfa_oracle.h) ----
def wfa_align(pattern, text, penalties=UVAIALIGN_PENALTIES, min_wavefront_length=128, max_distance_threshold=512, max_score=1 << 30):
    """(score, cigar bytes, M-wavefront cells, widest wavefront); min_wavefront_length <= 0: complete wavefronts."""
    L = lib()
    w = L.orc_wfa_new(WfaPenalties(*penalties), min_wavefront_length, max_distance_threshold)
    try:
        score = L.orc_wfa_align(w, pattern, len(pattern), text, len(text), max_score)
        n = C.c_int(0)
        ops = L.orc_wfa_cigar(w, C.byref(n))
        cigar = C.string_at(ops, n.value) if score >= 0 else b""
        return score, cigar, L.orc_wfa_cells(w), L.orc_wfa_max_width(w)
    finally:
        L.orc_wfa_del(w)


def gotoh_score(pattern, text, penalties=UVAIALIGN_PENALTIES):
    return lib().orc_gotoh_score(WfaPenalties(*penalties), pattern, len(pattern), text, len(text))


def cigar_score(cigar, pattern, text, penalties=UVAIALIGN_PENALTIES):
    return lib().orc_cigar_score(WfaPenalties(*penalties), cigar, len(cigar), pattern, len(pattern), text, len(text))


def uvaialign_query(ref, seq):
    """(score, aligned row of len(ref) bytes, cells) as src/align.c:357-390 produces it."""
    out = C.create_string_buffer(len(ref) + 1)
    cells = C.c_int64(0)
    score = lib().orc_uvaialign_query(ref, len(ref), seq, len(seq), out, C.byref(cells))
    return score, out.raw[:len(ref)], cells.value


def uvaialign_accepts(seq, ref_len, ambiguity=0.5):
    return bool(lib().orc_uvaialign_accepts(seq, len(seq), ref_len, ambiguity))


def uvaialign_batch(ref, seqs, threads=None):
    """scores [n] and aligned rows [n, len(ref)] over OpenMP threads (src/align.c:224-233)."""
    L = lib()
    if threads:
        L.orc_set_threads(threads)
    n = len(seqs)
    rows = np.zeros((n, len(ref) + 1), dtype=np.uint8)
    score = np.zeros(n, dtype=np.int32)
    lens = np.array([len(s) for s in seqs], dtype=np.int32)
    L.orc_uvaialign_batch(ref, len(ref), n, _cstr_array(seqs), lens.ctypes.data_as(C.POINTER(C.c_int)),
                          rows.ctypes.data_as(C.c_char_p), score.ctypes.data_as(C.POINTER(C.c_int)))
    return score, rows[:, :len(ref)]
