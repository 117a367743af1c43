"""The two generator workloads tests/test_timed_path_gpu.py sends through the timed path, and the digest of the oracle's answer to each.

The oracle needs 25-35 s per workload on the GPU box's cores -- a ninth of the GPU suite -- so its answers (heaps with their six scores and
ordinals, final tolerances, dump flags) are committed as SHA-256 digests in tests/golden/timed_path_digests.json, written by
tests/golden/make_timed_path_digests.py (which runs the oracle) and re-derived from the oracle by the CPU suite
(tests/test_timed_path_digests.py).  A GPU result whose digest equals the committed one IS the oracle's answer; only when it differs is the
oracle run in the GPU test, to say where.  Test infrastructure, like everything under oracle/."""
import hashlib
import json
import os

import numpy as np

import oracle_lib as O
from uvaia_amd import hostlib

QUERY_INDEX0 = 1 << 40          # as bench.py: queries and references come from disjoint sequence numbers of one generator
DIGESTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "timed_path_digests.json")
SPECS = {   # key: (queries, references, acgt, pool)
    "config1_sample": (1000, 8000, False, 8000),      # 1 000 generator queries x 8 000 references x 29 903 columns, k = 100, one pool
    "config2_sample": (10000, 1536, True, 1536),      # 10 000 generator queries x 1 536 references, --acgt, k = 100, one pool
}


def names(n, p="r"):
    return ["%s%d" % (p, i) for i in range(n)]


def digest(rows, T, saved):
    """rows: per query the kept entries, best first, as ((six scores), ordinal); T: final tolerances; saved: ordinals that entered any heap"""
    h = hashlib.sha256()
    h.update(np.asarray([len(r) for r in rows], dtype=np.int64).tobytes())
    flat = [x for r in rows for (s, o) in r for x in (*s, o)]
    h.update(np.asarray(flat, dtype=np.int64).tobytes())
    h.update(np.asarray(list(T), dtype=np.int64).tobytes())
    h.update(np.asarray(list(saved), dtype=np.int64).tobytes())
    return h.hexdigest()


def want_rows(gold, ntax):
    return [[(tuple(s), o) for o, _, s in gold.rows[iq]] for iq in range(ntax)]


class Workload:
    def __init__(self, key):
        nq, n_ref, self.acgt, self.pool = SPECS[key]
        self.key = key
        self.gen = hostlib.Synth(29903, seed=20241008, preset=0)
        self.qs, _ = self.gen.generate_bytes(QUERY_INDEX0, nq)
        self.qn = names(nq, "query_")
        self.refs, _ = self.gen.generate_bytes(0, n_ref)
        self._gold = None
        committed = json.load(open(DIGESTS)) if os.path.exists(DIGESTS) else {}
        self.committed = committed.get(key, {})
        self.ntax = self.committed.get("ntax", nq)        # queries the oracle keeps (all of them for these generator sets)

    def oracle(self):
        """(query set, result) of the CPU oracle -- src/nearest.c:245-330 restated in oracle/"""
        if self._gold is None:
            oq = O.Query(self.qs, self.qn, acgt=self.acgt)
            self._gold = (oq, O.search(oq, self.refs, names(len(self.refs)), pool=self.pool, nbest=100, ambig_r=0.5))
        return self._gold

    def oracle_digest(self):
        oq, gold = self.oracle()
        return {"ntax": int(oq.ntax), "digest": digest(want_rows(gold, oq.ntax), gold.final_T, list(gold.saved))}

    def check(self, rows, T, saved, what=""):
        """rows / tolerances / dump flags of a GPU search of this workload equal the oracle's"""
        if self.committed and digest(rows, T, saved) == self.committed["digest"]:
            return
        oq, gold = self.oracle()
        assert rows == want_rows(gold, oq.ntax), what
        assert list(T) == list(gold.final_T), what
        assert list(saved) == list(gold.saved), what
        raise AssertionError("the result equals the oracle's but not tests/golden/timed_path_digests.json[%s]: rerun tests/golden/make_timed_path_digests.py" % self.key)
