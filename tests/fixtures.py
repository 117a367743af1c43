"""Data fixtures shared by the tests: the reference's bundled alignment and small synthetic sets."""
import functools
import lzma
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_fasta_bytes(data):
    """Minimal FASTA parser: returns (names, seqs) with sequences upper-cased and stripped (bytes)."""
    names, seqs, cur = [], [], []
    for line in data.split(b"\n"):
        line = line.strip()
        if not line:
            continue
        if line.startswith(b">"):
            if names:
                seqs.append(b"".join(cur).upper())
            names.append(line[1:].decode())
            cur = []
        else:
            cur.append(line.replace(b" ", b""))
    if names:
        seqs.append(b"".join(cur).upper())
    return names, seqs


@functools.lru_cache(maxsize=1)
def load_bundled():
    with lzma.open(os.path.join(GOLDEN, "03.unique_acgt.aln.xz"), "rb") as fh:
        names, seqs = read_fasta_bytes(fh.read())
    return names, seqs


def sample_names_1k():
    with open(os.path.join(GOLDEN, "04.sample_1_1k.names")) as fh:
        return [l.strip() for l in fh if l.strip()]


def synth_alignment(n, L, seed, p_snp=0.002, p_amb=0.0005, n_run_frac=0.1, root=None, poly_cols=None):
    """Small synthetic aligned set shaped like SARS-CoV-2 data: a root, SNPs on a shared set of polymorphic
    columns, leading/trailing/internal N runs, sparse IUPAC codes and gaps."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    if root is None:
        root = acgt[rng.choice(4, size=L, p=[0.299, 0.184, 0.196, 0.321])]
    if poly_cols is None:
        poly_cols = rng.choice(L, size=max(1, L // 4), replace=False)
    amb = np.frombuffer(b"YRKMSWDHVB", dtype=np.uint8)
    out = []
    for _ in range(n):
        s = root.copy()
        k = rng.binomial(len(poly_cols), min(1.0, p_snp * L / len(poly_cols)))
        cols = rng.choice(poly_cols, size=k, replace=False)
        s[cols] = acgt[rng.integers(0, 4, size=k)]
        k = rng.binomial(L, p_amb)
        s[rng.integers(0, L, size=k)] = amb[rng.integers(0, len(amb), size=k)]
        k = rng.binomial(L, 0.0008)
        s[rng.integers(0, L, size=k)] = ord("-")
        lead, trail = rng.integers(0, max(2, L // 200)), rng.integers(0, max(2, L // 250))
        s[:lead] = ord("N")
        if trail:
            s[L - trail:] = ord("N")
        for _ in range(rng.poisson(n_run_frac * 6)):
            a = rng.integers(0, L)
            s[a:a + rng.integers(1, max(2, L // 60))] = ord("N")
        out.append(s.tobytes())
    return out, root, poly_cols


def unaligned_queries(ref, n, seed, p_snp=0.001, p_indel=0.0002, max_indel=12, n_runs=(90, 75, 350), run_prob=0.5, ambiguity=0.0003):
    """Unaligned sequences shaped like the input of uvaialign: the reference with SNPs, short insertions and deletions, sparse
    IUPAC codes, and N runs (leading, trailing, one amplicon-sized dropout), as bytes of differing lengths."""
    rng = np.random.default_rng(seed)
    acgt = b"ACGT"
    amb = b"YRKMSW"
    out = []
    for _ in range(n):
        s = bytearray()
        i, L = 0, len(ref)
        while i < L:
            r = rng.random()
            if r < p_snp:
                s.append(acgt[rng.integers(0, 4)]); i += 1
            elif r < p_snp + p_indel:
                i += int(rng.integers(1, max_indel + 1))                                  # deletion
            elif r < p_snp + 2 * p_indel:
                s.extend(acgt[j] for j in rng.integers(0, 4, size=int(rng.integers(1, max_indel + 1))))   # insertion
            elif r < p_snp + 2 * p_indel + ambiguity:
                s.append(amb[rng.integers(0, len(amb))]); i += 1
            else:
                s.append(ref[i]); i += 1
        if not s:
            s.append(acgt[0])
        lead, trail, drop = n_runs
        if lead and rng.random() < run_prob:
            k = min(len(s), int(rng.integers(1, lead + 1))); s[:k] = b"N" * k
        if trail and rng.random() < run_prob:
            k = min(len(s), int(rng.integers(1, trail + 1))); s[len(s) - k:] = b"N" * k
        if drop and rng.random() < run_prob and len(s) > 2:
            k = int(rng.integers(1, drop + 1)); a = int(rng.integers(0, max(1, len(s) - k))); k = min(k, len(s) - a); s[a:a + k] = b"N" * k
        out.append(bytes(s))
    return out


def random_acgt(n, seed):
    rng = np.random.default_rng(seed)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, size=n, p=[0.299, 0.184, 0.196, 0.321])].tobytes()
