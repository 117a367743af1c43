"""Packed database file format (uvaia_amd/csrc/host/uvdb.[ch]) without a GPU: tiles built here in numpy with the layout the
engine exports, written with the C writer, read back with the C reader; the exact text (gaps and other invalid characters, IUPAC
codes) must come back, as the .aln.xz dump of `uvaia --packed` needs."""
import ctypes as C
import os

import numpy as np
import pytest

from uvaia_amd import hostlib

CODE = {"A": 1, "C": 2, "G": 4, "T": 8, "M": 3, "R": 5, "W": 9, "S": 6, "Y": 10, "K": 12, "V": 7, "H": 11, "D": 13, "B": 14}
SIDE_INTS = 64


class Header(C.Structure):
    _fields_ = [("magic", C.c_char * 8), ("version", C.c_uint32), ("nchar", C.c_uint32), ("W4", C.c_uint32), ("side_row_ints", C.c_uint32),
                ("n_ref", C.c_uint64), ("n_tiles", C.c_uint64), ("tile_bytes", C.c_uint64), ("ref_ambiguity", C.c_double),
                ("off", C.c_uint64 * 8), ("reserved", C.c_uint64 * 2)]


class Reader(C.Structure):
    _fields_ = [("h", Header), ("map", C.c_void_p), ("map_len", C.c_size_t), ("name_idx", C.c_void_p), ("names", C.c_void_p),
                ("exc_idx", C.c_void_p), ("exc", C.c_void_p), ("non_n", C.POINTER(C.c_int32))]


def _lib():
    L = hostlib.load_library()
    L.uvdb_create.restype = C.c_void_p
    L.uvdb_create.argtypes = [C.c_char_p, C.c_int, C.c_size_t, C.c_int, C.c_double]
    L.uvdb_add_reference.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.uvdb_add_tiles.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.uvdb_close.argtypes = [C.c_void_p]
    L.uvdb_open.restype = C.POINTER(Reader)
    L.uvdb_open.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.uvdb_name.restype = C.c_char_p
    L.uvdb_name.argtypes = [C.POINTER(Reader), C.c_uint64]
    L.uvdb_unpack_reference.argtypes = [C.POINTER(Reader), C.c_uint64, C.c_char_p]
    L.uvdb_close_reader.argtypes = [C.POINTER(Reader)]
    return L


def _tiles(seqs, nchar):
    """[tile][w4][plane][lane][4 words] uint32, the engine's interchange form."""
    W4 = ((nchar + 31) // 32 + 3) // 4
    nt = (len(seqs) + 63) // 64
    t = np.zeros((nt, W4, 4, 64, 4), dtype=np.uint32)
    for i, s in enumerate(seqs):
        for pos, ch in enumerate(s.decode()):
            code = CODE.get(ch, 0)
            w = pos >> 5
            for p in range(4):
                if (code >> p) & 1:
                    t[i // 64, w >> 2, p, i % 64, w & 3] |= np.uint32(1 << (pos & 31))
    return t


def _random_alignment(n, nchar, seed):
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"ACGT" * 12 + b"NNNN--" + b"MRWSYKVHDB" + b"?XO.", dtype=np.uint8)
    seqs = []
    for _ in range(n):
        a = alphabet[rng.integers(0, len(alphabet), nchar)].copy()
        run = min(37, nchar - 2)
        lo = int(rng.integers(0, nchar - run))
        a[lo:lo + run] = ord("-")                     # a gap run that crosses a word boundary
        seqs.append(a.tobytes())
    return seqs


@pytest.mark.parametrize("n,nchar", [(1, 33), (64, 128), (131, 301)])
def test_round_trip(tmp_path, n, nchar):
    L = _lib()
    seqs = _random_alignment(n, nchar, seed=n)
    names = ["ref/%d|x y" % i for i in range(n)]
    tiles = _tiles(seqs, nchar)
    nt = tiles.shape[0]
    non_n = np.zeros(nt * 64, dtype=np.int32)
    non_n[:n] = [sum(ch not in b"N-?XO." for ch in s) for s in seqs]
    n_words = ((nchar + 31) // 32 + 3) // 4 * 4
    side = np.arange(nt * 64 * SIDE_INTS, dtype=np.int32).reshape(nt * 64, SIDE_INTS)      # plane words: any bits
    side[:, 0] = np.arange(nt * 64) % 14                                                    # listed-word count (may exceed the 11 slots)
    side[:, 1:12] = (np.arange(nt * 64)[:, None] + np.arange(11)[None, :]) % n_words        # word indices inside the alignment
    path = str(tmp_path / "db.uvdb").encode()
    w = L.uvdb_create(path, nchar, tiles[0].nbytes, SIDE_INTS, 0.5)
    assert w
    half = nt // 2
    for i in range(n):
        assert L.uvdb_add_reference(w, names[i].encode(), seqs[i]) == 0
    for a, b in ((0, half), (half, nt)):             # tiles arrive in pieces, after or between the names
        if b > a:
            assert L.uvdb_add_tiles(w, b - a, tiles[a:b].ctypes.data, non_n[a * 64:].ctypes.data, side[a * 64:].ctypes.data) == 0
    assert L.uvdb_close(w) == 0
    err = C.create_string_buffer(256)
    r = L.uvdb_open(path, err, 256)
    assert r, err.value
    h = r.contents.h
    assert (h.n_ref, h.n_tiles, h.nchar, h.tile_bytes, h.ref_ambiguity) == (n, nt, nchar, tiles[0].nbytes, 0.5)
    assert os.path.getsize(path) == h.off[7]
    out = C.create_string_buffer(nchar + 1)
    for i in range(n):
        assert L.uvdb_name(r, i).decode() == names[i]
        L.uvdb_unpack_reference(r, i, out)
        assert out.value == seqs[i]
        assert r.contents.non_n[i] == non_n[i]
    # sections come back as they were written
    raw = open(path, "rb").read()
    assert raw[h.off[0]:h.off[0] + tiles.nbytes] == tiles.tobytes()
    assert raw[h.off[2]:h.off[2] + side.nbytes] == side.tobytes()
    L.uvdb_close_reader(r)


def test_rejects_damaged_files(tmp_path):
    L = _lib()
    seqs = _random_alignment(3, 70, seed=9)
    tiles = _tiles(seqs, 70)
    path = str(tmp_path / "db.uvdb")
    w = L.uvdb_create(path.encode(), 70, tiles[0].nbytes, SIDE_INTS, 0.5)
    for i, s in enumerate(seqs):
        L.uvdb_add_reference(w, b"r%d" % i, s)
    L.uvdb_add_tiles(w, 1, tiles.ctypes.data, np.zeros(64, np.int32).ctypes.data, np.zeros((64, SIDE_INTS), np.int32).ctypes.data)
    assert L.uvdb_close(w) == 0
    raw = open(path, "rb").read()
    err = C.create_string_buffer(256)
    for name, data in (("short", raw[:-8]), ("magic", b"X" + raw[1:]), ("tiny", raw[:40])):
        p = str(tmp_path / (name + ".uvdb"))
        open(p, "wb").write(data)
        assert not L.uvdb_open(p.encode(), err, 256)
        assert err.value
    # damage that keeps the size and the header: interior index entries, an unterminated name table, side rows and valid-site
    # counts that point outside the alignment (these go to the device: the engine indexes planes with them)
    h = Header.from_buffer_copy(raw[:C.sizeof(Header)])
    off_nonn, off_side, off_name_idx, off_names, off_exc_idx = h.off[1], h.off[2], h.off[3], h.off[4], h.off[5]

    def patched(at, value, fmt="<q"):
        import struct
        b = bytearray(raw)
        b[at:at + struct.calcsize(fmt)] = struct.pack(fmt, value)
        return bytes(b)

    n_words = ((70 + 31) // 32 + 3) // 4 * 4
    cases = {
        "name_idx_interior_back": patched(off_name_idx + 8 * 2, 0),                 # name_idx[2] < name_idx[1]
        "name_idx_interior_far": patched(off_name_idx + 8 * 1, 1 << 40),            # beyond the names section
        "name_idx_first": patched(off_name_idx, 1),
        "exc_idx_interior": patched(off_exc_idx + 8 * 1, 1 << 50),
        "names_unterminated": patched(off_names + 2, ord("x"), "<B"),              # the NUL that ends "r0"
        "side_word_outside": patched(off_side + 4 * 1, n_words, "<i"),              # first listed word of reference 0 ...
        "side_count_negative": patched(off_side, -1, "<i"),
        "non_n_above_nchar": patched(off_nonn, 71, "<i"),
    }
    cases["side_word_outside"] = patched(off_side, 1, "<i")[:off_side + 4] + patched(off_side + 4, n_words, "<i")[off_side + 4:]
    for name, data in cases.items():
        p = str(tmp_path / (name + ".uvdb"))
        open(p, "wb").write(data)
        err.value = b""
        assert not L.uvdb_open(p.encode(), err, 256), name
        assert err.value, name
    # section sizes that would wrap 64 bits: n_ref, n_tiles huge with everything else in place
    hb = bytearray(raw)
    import struct
    struct.pack_into("<QQ", hb, Header.n_ref.offset, (1 << 61) + 3, ((1 << 61) + 3 + 63) // 64)
    p = str(tmp_path / "wrap.uvdb")
    open(p, "wb").write(bytes(hb))
    assert not L.uvdb_open(p.encode(), err, 256)
    # a writer that was given names but not their tiles does not produce a file that claims to be complete
    w = L.uvdb_create(str(tmp_path / "incomplete.uvdb").encode(), 70, tiles[0].nbytes, SIDE_INTS, 0.5)
    L.uvdb_add_reference(w, b"r0", seqs[0])
    assert L.uvdb_close(w) != 0
