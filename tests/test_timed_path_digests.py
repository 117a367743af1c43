"""CPU: the committed digests of the timed-path workloads (tests/golden/timed_path_digests.json) are the oracle's answers."""
import json

import pytest

import timed_path_workloads as W


@pytest.mark.parametrize("key", sorted(W.SPECS))
def test_committed_digest_is_the_oracles(key):
    w = W.Workload(key)
    assert w.oracle_digest() == json.load(open(W.DIGESTS))[key]
    oq, gold = w.oracle()
    w.check(W.want_rows(gold, oq.ntax), gold.final_T, gold.saved)                      # what a GPU test calls with its own result
    with pytest.raises(AssertionError):
        w.check(W.want_rows(gold, oq.ntax), [t + 1 for t in gold.final_T], gold.saved)


def test_digest_tells_results_apart():
    rows = [[((5, 5, 5, 9, 1, 30), 7), ((4, 5, 5, 9, 1, 30), 2)], []]
    d = W.digest(rows, [3, 1], [2, 7])
    assert d != W.digest([[rows[0][1], rows[0][0]], []], [3, 1], [2, 7])            # order of a heap
    assert d != W.digest([rows[0][:1], rows[0][1:]], [3, 1], [2, 7])                # which query holds what
    assert d != W.digest(rows, [3, 2], [2, 7]) and d != W.digest(rows, [3, 1], [2])
    assert d == W.digest([[(tuple(int(x) for x in s), int(o)) for s, o in r] for r in rows], (3, 1), (2, 7))
