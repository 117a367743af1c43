#!/usr/bin/env python3
"""Randomised GPU-vs-oracle campaign (run on the GPU box): shapes, modes and tuning values drawn at random; every search must
give the oracle's heaps, tolerances and dump flags, streamed and resident.  python tools/stress_parity.py [--n 40] [--seed 1]"""
import argparse
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures as F  # noqa: E402
import oracle_lib as O  # noqa: E402
from uvaia_amd import capi  # noqa: E402


def one(rng, k):
    nchar = rng.choice([61, 128, 333, 777, 1024, 2500, 4097])
    nq = rng.choice([1, 2, 3, 4, 5, 8, 9, 16, 17, 40, 64, 65, 100, 150, 257])
    nref = rng.choice([70, 300, 1000, 2500])
    acgt = rng.random() < 0.5
    trim = rng.choice([0, 0, 7, min(230, nchar // 4)])
    nbest = rng.choice([1, 2, 5, 13, 40])
    pool = rng.choice([64, 97, 333, 1000, nref])
    tuning = {"rare_max": rng.choice([0, -1, 1, 3, 50]), "subslice_refs": rng.choice([0, 64, 256]),
              "scan_tiles_per_wave": rng.choice([0, 1, 2, 4]), "scan_waves_per_block": rng.choice([0, 4, 8]),
              "rederive_streams": rng.choice([0, 1, 2]), "scan": rng.choice(["auto", "auto", "compressed"])}          # <= 16 queries scan the packed planes unless told otherwise
    if tuning["scan_tiles_per_wave"] == 4 and tuning["scan_waves_per_block"] == 4:
        tuning["scan_waves_per_block"] = 8                                        # four tiles per wave go with eight waves per block
    p_snp = rng.choice([0.002, 0.006, 0.02])
    refs, root, cols = F.synth_alignment(nref, nchar, seed=1000 + k, p_snp=p_snp)
    qs, _, _ = F.synth_alignment(nq, nchar, seed=5000 + k, root=root, poly_cols=cols, p_snp=p_snp)
    if rng.random() < 0.4:          # N runs across the queries: no constant-and-complete column, pools stop mattering
        qs = [bytearray(s) for s in qs]
        for i, s in enumerate(qs[:max(1, nq // 2)]):
            a = (i * 37) % max(1, nchar - 20)
            s[a:a + 20] = b"N" * len(s[a:a + 20])
        qs = [bytes(s) for s in qs]
    desc = dict(nchar=nchar, nq=nq, nref=nref, acgt=acgt, trim=trim, nbest=nbest, pool=pool, p_snp=p_snp, **{k_: v for k_, v in tuning.items() if v not in (0, "auto")})
    q = O.Query(qs, ["q%d" % i for i in range(nq)], acgt=acgt, trim=trim, ambig_q=1.0)
    if q.ntax < 1:
        return desc, True
    gold = O.search(q, refs, ["r%d" % i for i in range(nref)], pool=pool, nbest=nbest, ambig_r=1.0)
    want = [[(tuple(s), o) for o, _, s in gold.rows[iq]] for iq in range(q.ntax)]
    ok = True
    failed = []

    def check(what, good):
        nonlocal ok
        if not good:
            failed.append(what)
            desc["failed"] = failed
        ok &= bool(good)
    with capi.Engine.from_query(q, nbest=nbest, max_pool=pool, tuning=tuning) as eng:           # streamed
        ent = [eng.push(refs[a:a + pool]) for a in range(0, nref, pool)]
        n, T, sc, od = eng.drain()
        check("streamed", capi.finalise_heaps(n, sc, od) == want and list(T) == gold.final_T and list(np.nonzero(np.concatenate(ent))[0]) == list(gold.saved))
    with capi.Engine.from_query(q, nbest=nbest, max_pool=pool, tuning=tuning) as eng:           # resident
        eng.db_append(refs)
        e2 = eng.search_resident(pool)
        n, T, sc, od = eng.drain()
        check("resident", capi.finalise_heaps(n, sc, od) == want and list(T) == gold.final_T and list(np.nonzero(e2)[0]) == list(gold.saved))
        eng.reset()                                                              # planes rebuilt in place, searched again
        eng.db_rederive()
        e3 = eng.search_resident(pool)
        n, T, sc, od = eng.drain()
        check("resident after a rebuild", capi.finalise_heaps(n, sc, od) == want and list(T) == gold.final_T and list(np.nonzero(e3)[0]) == list(gold.saved))
    if rng.random() < 0.35 and pool >= 64:       # reference shards: several contexts on this card, each keeping its own pieces only
        world, piece = rng.choice([2, 3, 4]), rng.choice([64, 128])
        if piece <= pool:
            with capi.Group(q, [0] * world, nbest=nbest, max_pool=pool, piece_refs=piece) as g:
                g.db_append(refs)
                eg = g.search_resident(pool)
                n, T, sc, od = g.drain()
                check("group", capi.finalise_heaps(n, sc, od) == want and list(T) == gold.final_T and list(np.nonzero(eg)[0]) == list(gold.saved))
            desc["group"] = "%d x %d" % (world, piece)
    if rng.random() < 0.3:                       # the radius search on the same data
        dist = rng.choice([0, 1, 3, 9, 40])
        qb = O.Query(qs, ["q%d" % i for i in range(nq)], acgt=acgt, trim=trim, ambig_q=1.0, dist=dist, is_ball=True)
        if qb.ntax >= 1:
            md, _ = qb.ball(refs, ambig_r=0.001)
            gather = int(rng.choice([1, 2]))      # the columns of query->idx gathered by a pass of its own, or by the consensus pass
            with capi.Engine.from_query(qb, nbest=2, max_pool=pool, tuning={"ball_gather": gather}) as eng:
                got = np.concatenate([eng.ball(refs[a:a + pool], qb.dist + 1) for a in range(0, nref, pool)])   # qb.dist: the radius as the query structure corrects it (src/fastaseq.c:713-715)
                eng.db_append(refs)
                check("ball streamed", np.array_equal(got, md))
                check("ball resident", np.array_equal(eng.ball_resident(qb.dist + 1), md))
            desc["ball"] = "%d/g%d" % (dist, gather)
    desc["cons"] = len(q.idx_c) > 0
    return desc, bool(ok)


def one_alignment_set(rng, k):
    """the aligner of uvaialign: random reference length, divergence, penalties and reduction settings against the oracle"""
    from uvaia_amd import align
    L = rng.choice([1, 7, 64, 300, 1500, 6000])
    ref = F.random_acgt(L, 9000 + k)
    seqs = F.unaligned_queries(ref, rng.choice([1, 5, 40]), 9500 + k, p_snp=rng.choice([0.001, 0.02, 0.2]), p_indel=rng.choice([0.0002, 0.01]), max_indel=rng.choice([3, 12, 60]),
                               n_runs=(rng.choice([0, 30]), rng.choice([0, 30]), rng.choice([0, 40, 400])), run_prob=0.6, ambiguity=0.002)
    opts = rng.choice([{}, {}, dict(min_wavefront_length=0), dict(mismatch=3, gap_opening=5, gap_extension=1), dict(mismatch=5, gap_opening=2, gap_extension=3),
                       dict(min_wavefront_length=10, max_distance_threshold=rng.choice([5, 50]))])
    d = align.default_options()
    pen = (0, opts.get("mismatch", d.mismatch), opts.get("gap_opening", d.gap_opening), opts.get("gap_extension", d.gap_extension))
    ok = True
    table = L * pen[1] + pen[2] + 2 * L * pen[3]          # scores the reference's aligner has wavefronts for (src/align.c:306-309)
    try:
        with align.Aligner(ref, workspace_bytes=rng.choice([0, 64 << 20, 1 << 30]), **opts) as al:
            score, rows = al.align(seqs)
    except align.AlignError as e:                          # refused: right only if some sequence does score above the table
        i = int(str(e).split("sequence ")[1].split(":")[0]) if "alignment score above" in str(e) else -1
        over = i >= 0 and O.wfa_align(ref, seqs[i], penalties=pen, min_wavefront_length=opts.get("min_wavefront_length", d.min_wavefront_length),
                                      max_distance_threshold=opts.get("max_distance_threshold", d.max_distance_threshold))[0] > table
        return dict(aligner=True, L=L, n=len(seqs), refused="score above the table", **opts), bool(over)
    for i, t in enumerate(seqs):
        want, cigar, _, _ = O.wfa_align(ref, t, penalties=pen, min_wavefront_length=opts.get("min_wavefront_length", d.min_wavefront_length),
                                        max_distance_threshold=opts.get("max_distance_threshold", d.max_distance_threshold))
        row, pos = bytearray(), 0
        for op in cigar:
            if op in b"MX":
                row.append(t[pos]); pos += 1
            elif op == ord("I"):
                pos += 1
            else:
                row.append(ord("-"))
        ok &= int(score[i]) == want and rows[i].tobytes() == bytes(row)
    return dict(aligner=True, L=L, n=len(seqs), **opts), bool(ok)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--aligner-only", action="store_true", help="every configuration is an aligner set")
    a = ap.parse_args()
    rng = random.Random(a.seed)
    bad = 0
    for k in range(a.n):
        desc, ok = one_alignment_set(rng, k + 100 * a.seed) if (k % 5 == 4 or a.aligner_only) else one(rng, k + 100 * a.seed)
        print(("ok   " if ok else "FAIL ") + str(desc), flush=True)
        bad += not ok
    print("%d of %d configurations failed" % (bad, a.n))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
