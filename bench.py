#!/usr/bin/env python3
"""bench.py -- uvaia nearest-neighbour hot path on MI355X: reference sequences scored per second.

One "step" = one complete nearest-neighbour search: every reference of the HBM-resident packed database is scored
against the whole resident query set and passed through the ordered gate + top-k heaps (what `uvaia` does for one
reference file, src/nearest.c:249-330 of the reference), heaps reset between steps.  Inputs are synthetic
SARS-CoV-2-shaped alignments (uvaia_amd/csrc/host/synth.c, seed 20241008) and are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its own N rank processes)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N=1 runs BASELINE.json config[1]: 1 000 queries x 100 000 references x 29 903 columns, 4-bit IUPAC planes, top-k 100, and adds
to the same JSON line a `sweep` of other resident-query counts (1, 4, 16 queries x 1 M references: the HBM-bound regime) and
BASELINE config[2] (10 000 x 1 000 000, --acgt), each with its own roofline.  With N>1 every rank holds its own 100 000-reference
shard of the database (weak scaling).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
QUERY_INDEX0 = 1 << 40           # queries come from the same generator, disjoint sequence numbers
KERNEL_SOURCES = sorted(f for f in os.listdir(os.path.join(ROOT, "uvaia_amd", "csrc")) if f == "uvaia_gpu.hip" or (f.startswith("kernels_") and f.endswith(".inc")))   # the engine's device code (host_*.inc launch it)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--refs", type=int, default=100000, help="references per GPU")
    ap.add_argument("--nbest", type=int, default=100)
    ap.add_argument("--pool", type=int, default=65536, help="batch size (the reference's --pool)")
    ap.add_argument("--mode", choices=["iupac", "acgt"], default="iupac")
    ap.add_argument("--preset", type=int, default=0, help="0 = bundled-like N content, 1 = clean")
    ap.add_argument("--nchar", type=int, default=29903)
    ap.add_argument("--seed", type=int, default=20241008)
    ap.add_argument("--qt", type=int, default=0, help="query tile of the scan kernel (8/16/32, 0 = default)")
    ap.add_argument("--cpu-refs", type=int, default=8192, help="references in the timed CPU-baseline sample (0 = skip), after --cpu-warm untimed ones")
    ap.add_argument("--cpu-warm", type=int, default=2048, help="references fed to the CPU baseline before its timed sample (heaps full, tolerances settled)")
    ap.add_argument("--cpu-refs-1thread", type=int, default=512, help="timed sample of the single-thread CPU baseline (same warm state)")
    ap.add_argument("--parity-refs", type=int, default=1536, help="references of the in-run GPU-vs-oracle check")
    ap.add_argument("--multi", choices=["shards", "ring", "refshard"], default="refshard",
                    help="N > 1: 'refshard' = every GPU derives and scans 1/N of the references against all queries, the pair counters "
                         "move by one RCCL all-to-all per slice, the ordered replay is sharded by query (uvaia_amd/refshard.py); "
                         "'shards' = every GPU holds the whole database and a range of the queries (no data-path exchange); "
                         "'ring' = every GPU holds 1/N of the database, heap state handed rank to rank (uvaia_amd/ring.py)")
    ap.add_argument("--emulate-shard-of", type=int, default=0, metavar="N",
                    help="single process: do the work of rank 0 of N query shards (whole stream of N x --refs references, 1/N of the queries) "
                         "to measure the per-rank time of an N-GPU run on one GPU; the line is marked as emulated")
    ap.add_argument("--emulate-refshard", type=int, default=0, metavar="N",
                    help="single process, one GPU: N contexts run the reference-shard protocol of an N-GPU run (N x --refs references in all; "
                         "peer copies instead of RCCL); the contexts share the card, so the step time / N estimates one rank's step")
    ap.add_argument("--search-only", action="store_true",
                    help="leave the per-query-set planes of the resident database as the load built them (the timed step then "
                         "holds scan + replay only); by default every step rebuilds them first")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run GPU-vs-oracle check on the sample")
    ap.add_argument("--no-sweep", action="store_true", help="headline workload only (profiling runs)")
    ap.add_argument("--sweep-refs", type=int, default=1000000, help="references of the sweep entries")
    ap.add_argument("--scan", choices=["auto", "packed", "compressed"], default="auto", help="tuning: which pair scan the engine runs (auto: packed planes up to 32 queries)")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=VALUE", help="any other field of uvaia_gpu_tuning (repeatable), e.g. scan_waves_per_block=4")
    ap.add_argument("--subslice", type=int, default=0, help="tuning: sub-slice length of the resident search in references (0 = the library's choice)")
    ap.add_argument("--rederive-streams", type=int, default=0, help="tuning: streams the chunks of uvaia_gpu_db_rederive alternate over (0 = the library's choice)")
    ap.add_argument("--align-queries", type=int, default=10000, help="queries of the uvaialign record (BASELINE config[4]; 0 = skip)")
    ap.add_argument("--align-only", action="store_true", help="only the uvaialign record (profiling runs); prints {\"align\": ...}")
    ap.add_argument("--ball-only", action="store_true", help="only the uvaiaball record (profiling runs); prints {\"ball\": ...}")
    ap.add_argument("--align-cpu-queries", type=int, default=1024, help="queries of the uvaialign CPU-baseline sample (0 = skip)")
    return ap.parse_args(argv)


def kernel_source_hash():
    """sha256 over the engine's kernel sources: PMC figures in profiles/ are only quoted for the build they were measured on"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES + ["Makefile"]:                       # (the Makefile: its flags are part of what the kernels compile to)
        with open(os.path.join(ROOT, "uvaia_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def launch_ranks(args):
    """--gpus N without a launcher: start N rank processes (fresh children, before anything in this process touches the GPU),
    pass their output through and exit with their code.  Never falls back to one rank."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    sys.exit(rc)


def usable_cores():
    """Host threads this process may actually run on: the affinity mask and the cgroup CPU quota both bound it (a GPU box hands a
    rank a share of its host; omp_get_max_threads() reports every core of the machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q_ > 0 and p_ > 0:
                n = min(n, max(1, int(q_ / p_ + 0.5)))
        except Exception:
            pass
    return max(1, n)


def oracle_module():
    """the CPU restatement (oracle/): the checker and the cpu_baseline leg, never part of a timed GPU step"""
    if os.path.join(ROOT, "tests") not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    return O


def survey_bytes_per_ref(nchar, mode):
    """SURVEY 8d's implementation-independent figure: ceil(L*b/8) bytes per reference (14 952 at 4 bits; 2 bits + validity plane: 11 214)"""
    return (nchar * 4 + 7) // 8 if mode == "iupac" else (nchar * 2 + 7) // 8 + (nchar + 7) // 8


def roofline_of(eng, scan_ms, scan_launches, refs_scanned, nchar, mode, n_query):
    launches = max(1, scan_launches)
    avg_ms = scan_ms / launches
    variant = eng.scan_variant()        # 2 column-compressed, 0 / 1 two counters over the packed planes, -1 four counters
    bytes_per_ref = eng.packed_bytes_per_ref()
    kernel_bytes_per_ref = bytes_per_ref if variant != 2 else eng.scan_bytes_per_ref()
    sb = survey_bytes_per_ref(nchar, mode)
    refs_per_launch = float(refs_scanned) / launches
    achieved = refs_per_launch * sb / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    on_kernel = refs_per_launch * kernel_bytes_per_ref / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    r = {
        # the contract's figure is HBM: algorithmic bytes over the launch time.  What actually binds depends on the resident queries (every
        # reference byte is reused by every query tile): the packed-plane scan of up to 32 queries is bound by HBM, the column-compressed
        # scan above that by instruction issue -- its own `issue` block (wave-instructions per second against the measured issue peak)
        # comes from the SQ passes in profiles/ (attach_pmc_traffic), quoted only for the build they were measured on
        "bound": "hbm" if variant != 2 else "issue", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
        "algorithmic_bytes_per_ref": sb, "kernel_bytes_per_ref": kernel_bytes_per_ref,
        "achieved_on_kernel_bytes": round(on_kernel, 2), "frac_on_kernel_bytes": round(on_kernel / HBM_PEAK_GBS, 5),
        "kernel": {-1: "scan_%s_kernel" % mode, 0: "scan2_%s_kernel" % mode, 1: "scan2v_kernel"}.get(variant, "scan3_kernel"),
        "avg_launch_ms": round(avg_ms, 4), "launches": scan_launches,
        "algorithmic_bytes_per_launch": refs_per_launch * sb,
        "note": (("%d resident queries = one query tile: the scan reads the packed planes of every reference exactly once "
                  "(nothing derived); HBM is the bound (DESIGN.md 4.1)") % n_query) if variant == 0 else
                (("at %d resident queries every reference byte is reused by every query tile: the scan is bound by instruction issue "
                  "(popcounts + item-stream bookkeeping), not by HBM; the HBM-bound regime is Q <= 16 (`sweep`, DESIGN.md 4.1)") % n_query),
    }
    if achieved > HBM_PEAK_GBS:
        r["frac_note"] = ("the SURVEY figure counts the whole packed record of every reference; this kernel reads only the derived planes of the "
                          "word groups some query needs (kernel_bytes_per_ref), so the nominal rate exceeds the peak: frac_on_kernel_bytes is the physical HBM fraction")
    return r


PMC_FILE = "r04_pmc_traffic.json"


def attach_pmc_traffic(roofline, n_query, refs, pool, mode):
    """HBM-side traffic and issue rate of the dominant kernel from the committed PMC passes (rocprofv3 cannot run inside this process):
    quoted only when those passes measured THIS build of the kernels (hash of the kernel sources) on this workload (queries, mode and
    the references of one launch); otherwise `traffic` stays null."""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
    except Exception:
        return
    if pm.get("kernel_source_hash") != kernel_source_hash():
        roofline["traffic_note"] = "profiles/%s was measured on another build of the kernels: not quoted" % PMC_FILE
        return
    for e in pm.get("entries", []):
        cfg = e.get("config", {})
        if e.get("kernel") != roofline["kernel"] or cfg.get("queries") != n_query or cfg.get("mode") != mode:
            continue
        per_launch = roofline.get("algorithmic_bytes_per_launch", 0.0)
        if abs(e.get("algorithmic_bytes_per_launch", 0.0) - per_launch) > 0.02 * max(per_launch, 1.0):
            continue                                                    # launches of another length
        roofline["traffic"] = e["hbm_side_read_bytes_per_launch"] + e["write_bytes_per_launch"]
        roofline["traffic_over_algorithmic"] = round(roofline["traffic"] / max(per_launch, 1.0), 3)
        roofline["traffic_note"] = ("FETCH_SIZE x2 (gfx950) + WRITE_SIZE per launch, separate --pmc passes of this build "
                                    "(profiles/%s, kernel source hash %s); L2 misses incl. Infinity-Cache hits" % (PMC_FILE, pm["kernel_source_hash"]))
        if e.get("issue"):
            iss = dict(e["issue"])
            if roofline.get("avg_launch_ms"):                           # the rate of THIS run: the passes' instruction count over this run's launch time
                iss["achieved"] = round(iss["wave_instructions_per_launch"] / (roofline["avg_launch_ms"] * 1e-3) / 1e9, 1)
                iss["frac"] = round(iss["achieved"] / iss["peak"], 3)
            roofline["issue"] = iss
        if e.get("waits"):
            roofline["waits"] = e["waits"]
        return


def single_gpu_workload(hostlib, n_query, refs, mode, nbest, pool, steps, warmup, nchar, seed, preset, device, qt=0, search_only=False, parity_refs=0, cpu_refs=0):
    """One resident-database workload on one GPU (no exchange): returns the figures of a sweep entry.  parity_refs > 0: a sample of
    that many references of the same database goes through the same sequence of calls and is compared with the oracle."""
    gen = hostlib.Synth(nchar, seed=seed, preset=preset)
    qseqs, _ = gen.generate_bytes(QUERY_INDEX0, n_query)
    qnames = ["query_%d" % i for i in range(n_query)]
    t0 = time.time()
    pq = hostlib.PreparedQuery(qseqs, qnames, acgt=(mode == "acgt"))
    t1 = time.time()
    pool = min(pool, refs)
    eng = pq.open_engine(nbest=nbest, max_pool=pool, device=device)
    t2 = time.time()
    if qt:
        eng.set_query_tile(qt)
    eng.db_reserve(refs)
    for a in range(0, refs, 8192):
        n = min(8192, refs - a)
        rows, non_n = gen.generate(a, n)
        eng.db_append_block(rows, non_n)
    load_s = time.time() - t2

    def step(derive=not search_only):
        eng.reset()
        if derive:
            eng.db_rederive()
        eng.search_resident(pool, ordinal0=0, want_entered=False)
        eng.sync()

    step()
    for _ in range(warmup):
        step()
    eng.scan_stats(reset=True)
    import gc
    gc.collect(); gc.disable()
    t_a = time.perf_counter()
    for _ in range(steps):
        step()
    elapsed = time.perf_counter() - t_a
    gc.enable()
    scan_ms, scan_launches, _ = eng.scan_stats(reset=True)
    ms = 1e3 * elapsed / max(1, steps)
    sb = survey_bytes_per_ref(nchar, mode)
    out = {
        "workload": "%d queries x %d refs x %d cols, %s, top-k %d, pool %d" % (pq.ntax, refs, nchar, "4-bit IUPAC planes" if mode == "iupac" else "2-bit + validity planes (--acgt)", nbest, pool),
        "queries": pq.ntax, "refs": refs, "mode": mode, "steps": steps, "warmup": warmup,
        "value": round(refs * steps / elapsed, 2), "unit": "ref-seqs/s", "ms_per_step": round(ms, 3),
        "whole_step_GBps": round(refs * sb / (ms * 1e-3) / 1e9, 1), "whole_step_frac_of_hbm_peak": round(refs * sb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "roofline": roofline_of(eng, scan_ms, scan_launches, refs * steps, nchar, mode, pq.ntax),
        "db_load_s": round(load_s, 2), "query_prepare_s": round(t1 - t0, 2), "engine_open_s": round(t2 - t1, 2),
    }
    attach_pmc_traffic(out["roofline"], pq.ntax, refs, pool, mode)
    eng.close()
    if cpu_refs > 0:      # the CPU restatement on a bounded sample of the same workload (the oracle: the checker and the baseline, never the thing timed above)
        # (no single-thread sample here: samples are whole batches of 64 x threads references, and one such batch on one thread against
        # 10 000 queries is two minutes -- the headline's cpu_baseline carries the single-thread figure)
        out["cpu_baseline"] = cpu_baseline(oracle_module(), gen, 0, qseqs, qnames, mode, pool, nbest, 2 * cpu_refs, cpu_refs, 0)
    if parity_refs > 0:
        from uvaia_amd import capi
        n_s = min(parity_refs, refs)
        out["parity"] = parity_on_timed_path(oracle_module(), capi, pq, gen, 0, qseqs, qnames, mode, pool, nbest, n_s, device, qt)
        out["parity_sample"] = "the first %d references of this database through db_append -> db_rederive -> search_resident: heaps, tolerances and dump flags equal the oracle's" % n_s
    return out


def ball_workload(hostlib, n_query, refs, dist, mode, steps, nchar, seed, preset, device, parity_refs=0, tuning=None):
    """uvaiaball's radius search (src/ball.c:248-259, src/fastaseq.c:660-696) over an HBM-resident database: references per second,
    and how many of them the search had to compare with the queries themselves (the others stop at the queries' consensus)."""
    gen = hostlib.Synth(nchar, seed=seed, preset=preset)
    qseqs, _ = gen.generate_bytes(QUERY_INDEX0, n_query)
    t0 = time.time()
    pq = hostlib.PreparedQuery(qseqs, ["query_%d" % i for i in range(n_query)], dist=dist, acgt=(mode == "acgt"), is_ball=True)
    t1 = time.time()
    eng = pq.open_engine(nbest=2, max_pool=65536, device=device, tuning=tuning)
    eng.db_reserve(refs)
    for a in range(0, refs, 8192):
        n = min(8192, refs - a)
        rows, non_n = gen.generate(a, n)
        eng.db_append_block(rows, non_n)
    md = eng.ball_resident(dist + 1)                      # warm-up, and the answer
    eng.ball_asked(reset=True)
    eng.ball_kernel_ms(reset=True)
    t_a = time.perf_counter()
    for _ in range(steps):
        eng.ball_resident(dist + 1, want=False)
    elapsed = time.perf_counter() - t_a
    asked = eng.ball_asked(reset=True) // max(1, steps)
    k_ms = [v / max(1, steps) for v in eng.ball_kernel_ms(reset=True)]
    sb = survey_bytes_per_ref(nchar, mode)
    # per kernel (HIP events on the engine's stream): the consensus pass reads every reference's packed planes once and
    # (default) leaves their columns of query->idx gathered into dense words; the references that go on are moved into dense tiles; the
    # pair scan is vector work on the gathered word groups, and a query leaves it once every reference of a tile has reached its limit
    n_planes = 3 if mode == "acgt" else 4
    plane_bytes = n_planes * ((nchar + 127) // 128) * 16
    n_idx = int(len(pq.idx)) if getattr(pq, "idx", None) is not None else 0
    n_hot = 256 if n_idx >= 512 else 0
    ng4 = max(1, n_hot // 128 + (n_idx - n_hot + 127) // 128)
    gathered_bytes = n_planes * ng4 * 16
    fused = (tuning or {}).get("ball_gather", 2) == 2
    def gbps(b, ms): return round(b / (ms * 1e-3) / 1e9, 1) if ms > 0 else None
    def hbm(name, ms, nbytes, **more):
        g = gbps(nbytes, ms)
        return {"kernel": name, "ms": round(ms, 3), "bound": "hbm", "bytes": int(nbytes), "GBps": g, "frac": round((g or 0) / HBM_PEAK_GBS, 4), **more}
    kernels = [
        hbm("ball_stage1_kernel", k_ms[0], refs * (plane_bytes + (gathered_bytes if fused else 0)),
            note="planes read once" + (" + the gathered columns of every reference written" if fused else "")),
        hbm("ball_compact_kernel", k_ms[1], asked * gathered_bytes * 2, note="gathered rows of the references that go on, read (scattered 16-byte reads) and written; with the read-back of their number") if fused else
        hbm("ball_gather_cols_kernel", k_ms[1], asked * (plane_bytes + gathered_bytes), note="useful bytes; scattered 16-byte reads; with the read-back of the number of references that go on"),
        {"kernel": "ball_scan_kernel", "ms": round(k_ms[2], 3), "bound": "vector issue",
         "lane_operations_if_every_query_stayed": int(asked * pq.ntax * ng4 * 4 * (4 if mode == "acgt" else 7)),
         "note": "a query leaves the scan of a tile once all 64 references have reached their limits against it"}]
    out = {"workload": "uvaiaball: %d queries (%d after pruning) x %d refs x %d cols, %s, radius %d" % (n_query, pq.ntax, refs, nchar, mode, dist),
           "value": round(refs * steps / elapsed, 1), "unit": "ref-seqs/s", "ms_per_search": round(1e3 * elapsed / steps, 3), "steps": steps,
           "kept": int((md <= dist).sum()), "compared_with_the_queries": int(asked),
           "whole_search_GBps": round(refs * sb * steps / elapsed / 1e9, 1), "whole_search_frac_of_hbm_peak": round(refs * sb * steps / elapsed / 1e9 / HBM_PEAK_GBS, 4),
           "kernels": kernels, "query_prepare_s": round(t1 - t0, 2)}
    eng.close()
    if parity_refs > 0:
        O = oracle_module()
        n_s = min(parity_refs, refs)
        sample, _ = gen.generate_bytes(0, n_s)
        oq = O.Query(qseqs, ["query_%d" % i for i in range(n_query)], dist=dist, acgt=(mode == "acgt"), is_ball=True)
        md_want, _keep = oq.ball(sample, ambig_r=0.001)
        out["parity"] = bool(oq.ntax == pq.ntax and np.array_equal(md[:n_s], md_want))
        out["parity_sample"] = "cq->mindist of the first %d references (src/ball.c:248-251) equals the oracle's" % n_s
    return out


def unaligned_from_rows(rows, rng):
    """Aligned generator rows -> unaligned sequences as uvaialign receives them: gap characters dropped, then a few short
    deletions and insertions (Poisson 2 each, 1..12 sites) so that the alignment has real gaps to find."""
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for r in rows:
        s = r[r != ord("-")]
        for _ in range(rng.poisson(2.0)):
            a, k = int(rng.integers(0, len(s) - 16)), int(rng.integers(1, 13))
            s = np.delete(s, slice(a, a + k))
        for _ in range(rng.poisson(2.0)):
            a, k = int(rng.integers(0, len(s))), int(rng.integers(1, 13))
            s = np.insert(s, a, acgt[rng.integers(0, 4, size=k)])
        out.append(s.tobytes())
    return out


def align_workload(hostlib, n_query, steps, nchar, seed, preset, device, n_cpu, n_check=48):
    """uvaialign (src/align.c:224-233,357-390; BASELINE config[4]): n_query unaligned sequences against one reference on one GPU, the
    pool resident in HBM; a step = uvaia_align_run (all wavefront passes + backtrace + projected rows, results left on the device)."""
    from uvaia_amd import align
    gen = hostlib.Synth(nchar, seed=seed, preset=1)                       # the reference: a clean sequence, what is not ACGT filled in
    ref_row, _ = gen.generate(7, 1)
    ref = np.array(ref_row[0], dtype=np.uint8)
    bad = ~np.isin(ref, np.frombuffer(b"ACGT", dtype=np.uint8))
    ref[bad] = ord("A")
    ref = ref.tobytes()
    gen = hostlib.Synth(nchar, seed=seed, preset=preset)
    rng = np.random.default_rng(seed)
    seqs = []
    for a in range(0, n_query, 2048):
        rows, _ = gen.generate(QUERY_INDEX0 + a, min(2048, n_query - a))
        seqs += unaligned_from_rows(np.asarray(rows, dtype=np.uint8), rng)
    al = align.Aligner(ref, device=device)
    al.load(seqs)
    al.run()                                                              # warm-up (workspace allocation) and the answer
    score, rows = al.fetch()
    t_a = time.perf_counter()
    kernel_ms = 0.0
    for _ in range(steps):
        al.run()
        kernel_ms += al.stats()["kernel_ms"]
    elapsed = time.perf_counter() - t_a
    st = al.stats()
    al.close()
    out = {"workload": ("BASELINE config[4]: " if (n_query, nchar) == (10000, 29903) else "") +
           "uvaialign: %d unaligned queries (%d..%d characters) vs one reference of %d sites, penalties 0/4/6/2, reduced wavefronts 128/512" % (n_query, min(map(len, seqs)), max(map(len, seqs)), nchar),
           "value": round(n_query * steps / elapsed, 1), "unit": "queries/s", "ms_per_pool": round(1e3 * elapsed / steps, 3), "steps": steps,
           "kernel_ms_per_pool": round(kernel_ms / steps, 3), "kernel": "wfa_align_kernel", "passes": st["passes"],
           "cells_per_query": round(st["cells"] / n_query), "median_score": int(np.median(score)), "max_score": int(score.max()),
           "cell_updates_per_s": round(st["cells"] * steps / elapsed), "bound": "issue",
           "lds_plus_hbm_bytes_per_cell": 33, "lds_plus_hbm_GBps": round(st["wavefront_bytes"] * steps / elapsed / 1e9, 1),
           "hbm_bytes_per_cell": 5, "hbm_GBps": round(5.0 * st["cells"] * steps / elapsed / 1e9, 1), "frac_of_hbm_peak": round(5.0 * st["cells"] * steps / elapsed / 1e9 / HBM_PEAK_GBS, 5),
           "note": "a query is a chain of one dependent step per score (thousands; N runs cost 4 per site), 1 000-2 500 diagonals wide for most of them: one block of four waves "
                   "per query, the wavefronts of the last steps in LDS, a step = five earlier offsets per cell + extension + one LDS barrier; per cell 5 bytes go to memory for the "
                   "backtrace (M offset + provenance byte: hbm_GBps / frac_of_hbm_peak), 8 are I/D offsets and 20 are read, all of which stay in LDS (lds_plus_hbm_GBps counts all 33); the kernel is bound by instruction issue, not by HBM.  Rows and scores equal the CPU restatement's (oracle/wfa_oracle.c; parity unpinned beyond "
                   "the optimal gap-affine score)"}
    if os.path.join(ROOT, "tests") not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    pick = list(range(0, n_query, max(1, n_query // n_check)))[:n_check]
    g_score, g_rows = O.uvaialign_batch(ref, [seqs[i] for i in pick])
    out["parity_on_sample"] = bool(np.array_equal(g_score, score[pick]) and np.array_equal(g_rows, rows[pick]))
    if n_cpu > 0:
        omp_max = O.lib().orc_max_threads()
        threads = min(usable_cores(), omp_max)
        # at least 64 queries per thread (or the whole pool), started dearest first as the GPU path starts them: with a handful of
        # queries per thread the few N-rich ones (scores of tens of thousands) set the wall time
        n_s = min(n_query, max(n_cpu, 64 * threads))
        cost = np.array([len(q_) - sum(q_.count(b) for b in (b"A", b"C", b"G", b"T")) for q_ in seqs[:n_s]])
        sample = [seqs[i] for i in np.argsort(-cost, kind="stable")]
        t0 = time.perf_counter(); O.uvaialign_batch(ref, sample, threads=threads); t_all = time.perf_counter() - t0
        one = seqs[:max(16, min(64, n_s // 64))]
        t0 = time.perf_counter(); O.uvaialign_batch(ref, one, threads=1); t_one = time.perf_counter() - t0
        O.lib().orc_set_threads(threads)
        out["cpu_baseline"] = {"value": round(len(sample) / t_all, 1), "unit": "queries/s", "cores": threads, "kind": "port", "host_hardware_threads": omp_max,
                               "sample": "the first %d queries of the same pool, dearest first, OpenMP (dynamic schedule) over %d threads = the cores this process may use, %.1f s; "
                                         "an unoptimised restatement of the published algorithm, not the WFA library" % (len(sample), threads, t_all),
                               "value_1_thread": round(len(one) / t_one, 1), "value_1_thread_x_cores": round(threads * len(one) / t_one, 1),
                               "sample_1_thread": "the first %d queries, 1 thread, %.1f s" % (len(one), t_one)}
    return out


def cpu_baseline(O, gen, first, qseqs, qnames, mode, pool, nbest, n_warm, n_timed, n_one):
    """The oracle's restatement of the reference loops (src/nearest.c:249-330) on the host cores, in the reference's configuration:
    batches of 64 x threads references (its default --pool, src/nearest.c:81) capped at the workload's own --pool, one OpenMP thread
    per core this process may use.  The heaps are in the state a long run spends its time in: `n_warm` references are fed untimed
    (heaps fill, tolerances settle: the early exits of src/nearest.c:488-496 fire), then `n_timed` references are timed on all
    threads and `n_one` more on one thread (same warm state, same batch size)."""
    L = O.lib()
    oq = O.Query(qseqs, qnames, acgt=(mode == "acgt"))
    omp_max = L.orc_max_threads()
    cores = min(usable_cores(), omp_max)
    L.orc_set_threads(cores)
    bpool = max(64, min(pool, 64 * cores))
    s = L.orc_search_new(oq.ptr, bpool, nbest, 0.5, 0)

    def feed(a, n):
        t_total = 0.0
        for b in range(a, a + n, 8192):
            m = min(8192, a + n - b)
            seqs, _ = gen.generate_bytes(first + b, m)
            names = ["ref_%d" % (b + i) for i in range(m)]
            arr_s, arr_n = O._cstr_array(seqs), O._cstr_array(names)
            t0 = time.perf_counter()
            rc = L.orc_search_feed(s, m, arr_s, arr_n, None)
            t_total += time.perf_counter() - t0
            assert rc == 0
        return t_total

    n_warm, n_timed, n_one = [(x + bpool - 1) // bpool * bpool for x in (n_warm, n_timed, n_one)]      # whole batches only
    try:
        feed(0, n_warm)
        t_all = feed(n_warm, n_timed)
        out = {"value": round(n_timed / t_all, 2), "unit": "ref-seqs/s", "cores": cores, "kind": "port",
               "sample": "%d references of the same database vs the same %d queries after %d untimed warm-up references (heaps full), "
                         "batches of %d (the reference's default pool of 64 x threads, src/nearest.c:81), OpenMP over %d threads = the cores this "
                         "process may use (the host reports %d hardware threads), %.1f s" % (n_timed, oq.ntax, n_warm, bpool, cores, omp_max, t_all),
               "host_hardware_threads": omp_max}
        if n_one > 0:
            L.orc_set_threads(1)
            t_one = feed(n_warm + n_timed, n_one)
            L.orc_set_threads(cores)
            out["value_1_thread"] = round(n_one / t_one, 2)
            out["value_1_thread_x_cores"] = round(cores * n_one / t_one, 2)
            out["sample_1_thread"] = "the next %d references, same warm state and batch size, 1 thread, %.1f s" % (n_one, t_one)
    finally:
        L.orc_search_del(s)
    try:
        out["host"] = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    return out


def parity_on_timed_path(O, capi, pq, gen, first, qseqs, qnames, mode, pool, nbest, n_s, device, qt):
    """A sample of the benchmark database through the sequence of calls the timed step makes (resident database, rebuild of the
    derived planes on its own stream, sub-sliced search) must give the oracle's heaps, tolerances and dump flags."""
    sample, _ = gen.generate_bytes(first, n_s)
    oq = O.Query(qseqs, qnames, acgt=(mode == "acgt"))
    p = min(pool, n_s)
    gold = O.search(oq, sample, ["ref_%d" % i for i in range(n_s)], pool=p, nbest=nbest, ambig_r=0.5)
    with pq.open_engine(nbest=nbest, max_pool=p, device=device) as e2:
        if qt:
            e2.set_query_tile(qt)
        e2.db_reserve(n_s)
        rows, non_n = gen.generate(first, n_s)
        e2.db_append_block(rows, non_n)
        e2.reset()
        e2.db_rederive()
        ent = e2.search_resident(p)
        e2.sync()
        n, T, sc, od = e2.drain()
    got = capi.finalise_heaps(n, sc, od)
    return bool(list(T) == gold.final_T and list(np.nonzero(ent)[0]) == list(gold.saved)
                and all(got[iq] == [(tuple(s), o) for o, _, s in gold.rows[iq]] for iq in range(oq.ntax)))


def parity_on_refshard(O, capi, refshard, dist, pq, gen, qseqs, qnames, mode, nbest, world, rank, device, on_gpu):
    """N > 1: a sample of the stream through the protocol the timed step runs -- reference shards, pieces of two tiles so that there
    are several stripes, the exchange over the process group in use (RCCL on the node) -- and every rank compares the heaps and
    tolerances of its own query shard with the oracle's; the verdict is the AND over the ranks."""
    import torch
    piece, per_rank = 128, 256
    n_s = per_rank * world
    plan = refshard.Plan(world, rank, per_rank, pq.ntax, pool=None, piece=piece)
    sample, _ = gen.generate_bytes(0, n_s)
    oq = O.Query(qseqs, qnames, acgt=(mode == "acgt"))
    gold = O.search(oq, sample, ["ref_%d" % i for i in range(n_s)], pool=n_s, nbest=nbest, ambig_r=0.5)
    ok = True
    with pq.open_engine(nbest=nbest, max_pool=plan.slice_refs, device=device) as e2:
        e2.db_set_shard(rank, world, plan.piece)
        e2.db_reserve(n_s)
        rows, non_n = gen.generate(0, n_s)
        e2.db_append_block(rows, non_n)                  # handed the whole stream: keeps the references of its own pieces
        x2 = refshard.TorchExchange(dist, plan, e2, "cuda" if on_gpu else "cpu", pinned=not on_gpu)
        x2.connect_peers(e2)
        e2.reset()
        e2.db_rederive()
        refshard.run(e2, plan, x2, len(pq.idx_c) > 0)
        e2.sync()
        n, T, sc, od = e2.drain()
        got = capi.finalise_heaps(n, sc, od)
        for iq in range(plan.q0, plan.q1):
            ok = ok and got[iq] == [(tuple(s_), o) for o, _, s_ in gold.rows[iq]] and int(T[iq]) == gold.final_T[iq]
        x2.disconnect_peers(e2)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()) == 1)


def emulate_refshard(args):
    """N contexts on one GPU run the protocol an N-GPU run uses (uvaia_gpu_group_*): every context derives and scans its pieces of the
    N x --refs references against all queries, the rows of each query shard are copied to the context that replays them.  The
    contexts share one card, so a step takes the SUM of the ranks' work: step / N estimates one rank's step on its own GPU (the
    copies are device-local here; on the node they cross xGMI)."""
    import torch
    from uvaia_amd import capi, hostlib, refshard
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    n = args.emulate_refshard
    gen = hostlib.Synth(args.nchar, seed=args.seed, preset=args.preset)
    qseqs, _ = gen.generate_bytes(QUERY_INDEX0, args.queries)
    pq = hostlib.PreparedQuery(qseqs, ["query_%d" % i for i in range(args.queries)], acgt=(args.mode == "acgt"))
    plan = refshard.Plan(n, 0, args.refs, pq.ntax, pool=args.pool)
    total = n * args.refs
    g = capi.Group(pq, [0] * n, nbest=args.nbest, max_pool=plan.piece, piece_refs=plan.piece)
    g.db_reserve(total)
    for a in range(0, total, 4096):
        m = min(4096, total - a)
        rows, non_n = gen.generate(a, m)
        g.db_append([rows[i].tobytes() for i in range(m)], non_n)

    def step():
        g.reset()
        g.db_rederive()
        g.search_resident(min(args.pool, total), want_entered=False)
        g.sync()

    step()
    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    elapsed = time.perf_counter() - t0
    ms = 1e3 * elapsed / max(1, args.steps)
    # what one rank of BASELINE config[3] (10 M references over 8 GPUs: 1.25 M per rank) would hold, from this engine's own per-reference sizes
    packed_b, derived_b = g.member_bytes_per_ref()
    w4 = (args.nchar + 127) // 128
    rows = ((pq.ntax + 127) // 128) * 128
    per_rank_refs, cols = 1250000, plan.piece + 64
    myrows = max(1, (pq.ntax + n - 1) // n)
    sizing = {"references_per_rank": per_rank_refs,
              "packed_planes_bytes": per_rank_refs * packed_b, "derived_planes_bytes": per_rank_refs * (derived_b + w4 * 16),
              "side_rows_and_counts_bytes": per_rank_refs * (256 + 12), "dump_flags_bytes_whole_stream": n * per_rank_refs,
              "scan_counter_ring_bytes": 4 * rows * cols * 4 + 4 * rows * (cols // 64) * 8,
              "exchange_buffers_bytes": 2 * (rows * cols * 4 + rows * (cols // 64) * 8 + n * myrows * cols * 4 + n * myrows * (cols // 64) * 8 + (n + 1) * cols * 20),
              "note": "per-reference sizes are this engine's (uvaia_gpu_packed_bytes_per_ref, uvaia_gpu_derived_bytes_per_ref + the V plane); the exchange buffers are those of "
                      "uvaia_gpu_group_open (two send and two receive sets per member); a piece is %d references" % plan.piece}
    sizing["total_bytes"] = sum(v for k, v in sizing.items() if k.endswith("_bytes") or k.endswith("_stream"))
    sizing["total_GB"] = round(sizing["total_bytes"] / 1e9, 2)
    print(json.dumps({"metric": "ref-seqs scored/sec", "per_rank_hbm_at_config3": sizing, "emulated": "reference shards: %d contexts on one GPU (uvaia_gpu_group_*), %d references in all; the contexts share the card: "
                      "step_ms / %d estimates one rank's step on its own GPU" % (n, total, n),
                      "n_contexts": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step_all_contexts_on_one_gpu": round(ms, 3), "estimated_ms_per_rank_step": round(ms / n, 3),
                      "estimated_value_on_%d_gpus" % n: round(total / (ms / n * 1e-3), 1), "unit": "ref-seqs/s", "piece_refs": plan.piece,
                      "config": {"workload": "%d queries x %d refs/GPU x %d cols, %s, top-k %d" % (pq.ntax, args.refs, args.nchar, args.mode, args.nbest)}}))
    g.close()


def main():
    args = parse()
    if args.emulate_refshard > 1:
        return emulate_refshard(args)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        launch_ranks(args)                       # does not return
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d: refusing to report a line for a different number of ranks" % (world, args.gpus))

    import torch
    from uvaia_amd import capi, hostlib
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # UVAIA_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (exchanges via host memory)
    backend = os.environ.get("UVAIA_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if args.align_only:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        print(json.dumps({"align": align_workload(hostlib, args.align_queries, args.steps, args.nchar, args.seed, args.preset, local_rank, args.align_cpu_queries)}))
        return
    if args.ball_only:
        print(json.dumps({"ball": ball_workload(hostlib, args.queries, args.sweep_refs, 2, args.mode, args.steps, args.nchar, args.seed, args.preset, local_rank,
                                                parity_refs=0 if args.no_parity else 8192,
                                                tuning=({kv.split("=", 1)[0]: int(kv.split("=", 1)[1]) for kv in args.tuning} or None))}))
        return
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- inputs: queries (host C preparation, as the command line does) and the resident packed database
    gen = hostlib.Synth(args.nchar, seed=args.seed, preset=args.preset)
    qseqs, _ = gen.generate_bytes(QUERY_INDEX0, args.queries)
    qnames = ["query_%d" % i for i in range(args.queries)]
    t_q0 = time.time()
    pq = hostlib.PreparedQuery(qseqs, qnames, acgt=(args.mode == "acgt"))
    t_q1 = time.time()
    emu = args.emulate_shard_of if (world == 1 and args.emulate_shard_of > 1) else 0
    multi = args.multi if world > 1 else ("shards" if emu else None)
    shard_mode = multi == "shards"
    # query shards: every rank holds (and scans) the whole stream; reference shards: the stream has world x refs references, a rank keeps,
    # derives and scans only its own pieces (the capacity below is the stream's length: a context sizes its arrays for its share)
    local_refs = (emu or world) * args.refs if (shard_mode or multi == "refshard") else args.refs
    pool = min(args.pool, local_refs)
    from uvaia_amd import ring, shards
    on_gpu = backend == "nccl"
    plan = None
    if multi == "refshard":
        from uvaia_amd import refshard
        plan = refshard.Plan(world, rank, args.refs, pq.ntax, pool=args.pool)
        pool = plan.slice_refs
    eng = pq.open_engine(nbest=args.nbest, max_pool=pool, device=local_rank, tuning=({k: v for k, v in (("rederive_streams", args.rederive_streams), ("subslice_refs", args.subslice), ("scan", args.scan if args.scan != "auto" else 0))
                                      + tuple((kv.split("=", 1)[0], int(kv.split("=", 1)[1])) for kv in args.tuning) if v} or None))
    t_q2 = time.time()
    # whatever ends this run -- an exception in a step included -- every rank first unmaps the other ranks' arrays (and meets them), then
    # frees its own: an owner must not free memory another rank's replay still reads in place
    import atexit, contextlib
    cleanup = contextlib.ExitStack()
    atexit.register(cleanup.close)
    cleanup.callback(eng.close)
    if args.qt:
        eng.set_query_tile(args.qt)
    if multi == "refshard":
        eng.db_set_shard(rank, world, plan.piece)
    eng.db_reserve(local_refs)
    t0 = time.time()
    # ring: block-cyclic shard, stripe s (= one pool of world*pool references of the stream) = slice s of rank 0, 1, ...
    # shards: the whole stream on every rank;  refshard: slice s of the stream = rank-major pieces, see uvaia_amd/refshard.py
    if multi == "refshard":
        slices = []
        for pc in plan.stream_pieces():              # its own pieces are generated and packed, the others only counted
            if pc.owner == rank:
                for a in range(0, pc.n, 8192):
                    n = min(8192, pc.n - a)
                    rows, non_n = gen.generate(pc.first + a, n)
                    eng.db_append_block(rows, non_n)
            else:
                eng.db_skip(pc.n)
    elif shard_mode or world == 1:
        slices = [ring.Slice(0, local_refs, 0)]
    else:
        slices = ring.block_cyclic_layout(args.refs, pool, rank, world)
    first = slices[0].ordinal0 if slices else 0
    chunk = 8192
    for sl in slices:
        for a in range(0, sl.n, chunk):
            n = min(chunk, sl.n - a)
            rows, non_n = gen.generate(sl.ordinal0 + a, n)
            eng.db_append_block(rows, non_n)
    load_s = time.time() - t0
    bytes_per_ref = eng.packed_bytes_per_ref()
    comm = ring.TorchRingComm(dist, rank, world, cuda=on_gpu) if (dist is not None and multi == "ring") else None
    nbytes = eng.state_bytes()
    cons = len(pq.idx_c) > 0
    q0, q1 = shards.query_shard(pq.ntax, rank, emu or world) if shard_mode else (0, pq.ntax)
    allmax = shards.TorchMax(dist, "cuda" if on_gpu else "cpu") if (shard_mode and cons and dist is not None) else None
    xchg = refshard.TorchExchange(dist, plan, eng, "cuda" if on_gpu else "cpu", pinned=not on_gpu) if multi == "refshard" else None
    if xchg is not None:
        xchg.connect_peers(eng)
        cleanup.callback(lambda: xchg.disconnect_peers(eng))

    # ---- timed region
    # One step = everything one search of the resident database costs for this query set: the planes derived from the packed
    # records for the query set (built by the appends while loading; rebuilt here so that the step holds them), the pair scan,
    # the consensus pre-score where it applies, and the ordered replay into the heaps.
    def step(derive=not args.search_only):
        eng.reset()
        if derive:
            eng.db_rederive()
        if world == 1 and not emu:
            eng.search_resident(pool, ordinal0=0, want_entered=False)
        elif shard_mode:   # no data-path exchange (one all-reduced int per pool if the query set has complete constant columns)
            shards.run_query_shard(eng, q0, q1, local_refs, pool, cons, allmax)
        elif multi == "refshard":
            refshard.run(eng, plan, xchg, cons)
        else:   # scans run concurrently on all ranks; the heap state visits the ranks in stream order, pipelined by query group
            ring.run_ring_grouped(eng, comm, rank, world, slices, pq.ntax, cons,
                                  lambda nb: ring.TorchStateBuffer(nb, "cuda" if on_gpu else "cpu"))
        eng.sync()

    step()                             # setup, not a warmup step: the engine sizes its counter buffers on the first search
    for _ in range(args.warmup):
        step()
    eng.scan_stats(reset=True)
    barrier()
    import gc
    gc.collect()
    gc.disable()                       # a collection in the middle of a 4 ms step would be measured as GPU time
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        t_s = time.perf_counter()
        step()
        step_ms.append(round(1e3 * (time.perf_counter() - t_s), 3))
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    scan_ms, scan_launches, scan_bytes = eng.scan_stats(reset=True)
    admitted, demanded, dense_rescans = eng.replay_stats(reset=True)
    # the two parts of a step on their own (untimed for `value`): derived planes only, scan + replay only
    derive_ms = search_only_ms = 0.0
    if world == 1:
        gc.disable()
        t_d = time.perf_counter()
        for _ in range(args.steps):
            eng.db_rederive()
            eng.sync()
        derive_ms = 1e3 * (time.perf_counter() - t_d) / max(1, args.steps)
        t_d = time.perf_counter()
        for _ in range(args.steps):
            step(derive=False)
        search_only_ms = 1e3 * (time.perf_counter() - t_d) / max(1, args.steps)
        gc.enable()
        eng.scan_stats(reset=True)
        eng.replay_stats(reset=True)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / max(1, args.steps)
    value = (emu or world) * args.refs * args.steps / elapsed

    # ---- roofline of the dominant kernel (pair scan): algorithmic bytes per launch / mean launch time (HIP events on its stream)
    roofline = roofline_of(eng, scan_ms, scan_launches, float(args.refs if multi == "refshard" else local_refs) * args.steps, args.nchar, args.mode, pq.ntax)
    if world == 1 and not emu:
        attach_pmc_traffic(roofline, pq.ntax, args.refs, pool, args.mode)
    derived_bytes = eng.derived_bytes_per_ref()
    packed_bytes = bytes_per_ref

    # ---- CPU baseline and the in-run parity check (rank 0, N=1 only)
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not emu and (args.cpu_refs > 0 or not args.no_parity):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        if args.cpu_refs > 0:
            cpu = cpu_baseline(O, gen, first, qseqs, qnames, args.mode, pool, args.nbest, args.cpu_warm, args.cpu_refs, args.cpu_refs_1thread)
        if not args.no_parity:
            parity = parity_on_timed_path(O, capi, pq, gen, first, qseqs, qnames, args.mode, pool, args.nbest, min(args.parity_refs, args.refs), local_rank, args.qt)
    if world > 1 and multi == "refshard" and not args.no_parity:      # every rank takes part; the oracle is the checker, as at N = 1
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        parity = parity_on_refshard(O, capi, refshard, dist, pq, gen, qseqs, qnames, args.mode, args.nbest, world, rank, local_rank, on_gpu)
    cleanup.close()                    # disconnect_peers, then eng.close

    # ---- other resident-query counts, driver-timed in the same run (rank 0, N=1 only)
    sweep = ball = aligned = None
    if rank == 0 and world == 1 and not emu and not args.no_sweep:
        sweep = []
        # (queries, mode, steps, references of the in-run oracle check: the oracle's cost grows with the query count)
        for nq_s, mode_s, steps_s, par_s in ((1, "iupac", 5, 4096), (4, "iupac", 5, 4096), (16, "iupac", 5, 4096), (64, "iupac", 5, 2048), (10000, "acgt", 2, 768)):
            e = single_gpu_workload(hostlib, nq_s, args.sweep_refs, mode_s, args.nbest, args.sweep_refs if nq_s <= 64 else args.pool,
                                    steps_s, 1, args.nchar, args.seed, args.preset, local_rank, parity_refs=0 if args.no_parity else par_s,
                                    cpu_refs=(256 if (nq_s == 10000 and args.cpu_refs > 0) else 0))
            if (nq_s, args.sweep_refs, args.nchar, args.nbest, mode_s) == (10000, 1000000, 29903, 100, "acgt"):
                e["workload"] = "BASELINE config[2]: " + e["workload"]
            sweep.append(e)
        ball = ball_workload(hostlib, 1000, args.sweep_refs, 2, "iupac", 3, args.nchar, args.seed, args.preset, local_rank, parity_refs=0 if args.no_parity else 8192)
        if args.align_queries > 0:
            aligned = align_workload(hostlib, args.align_queries, 3, args.nchar, args.seed, args.preset, local_rank, args.align_cpu_queries)

    if rank == 0:
        out = {
            "metric": "ref-seqs scored/sec", "value": round(value, 2), "unit": "ref-seqs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "step_ms_rank0": step_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # what the process group itself reports: the ranks the collectives of the timed steps ran over, and over what
            "rccl_ranks_seen": (dist.get_world_size() if dist is not None else 1), "collectives_backend": (backend if dist is not None else None),
            "emulated": ("rank 0 of %d query shards on one GPU: `value` is what %d GPUs would reach if every rank took this long" % (emu, emu)) if emu else None,
            "multi_gpu": None if (world == 1 and not emu) else
            ("query shards: every GPU holds all %d references and the heaps of %d of the %d queries (column classes from the whole set); no data-path exchange%s; exact"
             % (local_refs, q1 - q0, pq.ntax, ", one all-reduced int per pool" if cons else "")) if shard_mode else
            plan.describe() if multi == "refshard" else
            "block-cyclic slices of %d refs, concurrent scans, heap state (%d B in %d per-query-group blobs) pipelined rank to rank (RCCL isend/irecv), exact" % (pool, nbytes, world),
            "dtype": "u32 bit-planes / int32 counts", "data": "synthetic (seed %d, preset %d)" % (args.seed, args.preset),
            "config": {"workload": (("BASELINE config[1]: " if (pq.ntax, args.refs, args.nchar, args.nbest) == (1000, 100000, 29903, 100) else
                                     "BASELINE config[2]: " if (pq.ntax, args.refs, args.nchar, args.nbest, args.mode) == (10000, 1000000, 29903, 100, "acgt") else "")
                                    + "%d queries x %d refs/GPU x %d cols, %s, top-k %d, pool %d")
                                   % (pq.ntax, args.refs, args.nchar, "4-bit IUPAC planes" if args.mode == "iupac" else "2-bit + validity planes (--acgt)", args.nbest, pool),
                       "queries": pq.ntax, "refs_per_gpu": args.refs, "nchar": args.nchar, "nbest": args.nbest, "pool": pool,
                       "mode": args.mode, "packed_bytes_per_ref": packed_bytes, "db_load_s": round(load_s, 2),
                       "query_prepare_s": round(t_q1 - t_q0, 2), "engine_open_s": round(t_q2 - t_q1, 2)},
            "step_parts": {"includes_derived_planes": not args.search_only, "derived_planes_ms": round(derive_ms, 3), "scan_and_replay_ms": round(search_only_ms, 3),
                           "derived_planes_kernel": {"kernel": "derive_all_kernel", "bound": "hbm", "read_bytes_per_ref": packed_bytes, "written_bytes_per_ref": derived_bytes,
                                                     "achieved": round(local_refs * (packed_bytes + derived_bytes) / (derive_ms * 1e-3) / 1e9, 1) if (derive_ms > 0 and derived_bytes) else None,
                                                     "peak": HBM_PEAK_GBS, "unit": "GB/s (reads + writes, host-timed over its launches)"},
                           "note": "a step = planes derived from the resident packed records for this query set (uvaia_gpu_db_rederive) + pair scan + ordered replay; "
                                   "the two parts timed on their own after the timed region (N = 1 only)"},
            "roofline": roofline,
            "replay": {"admissions_per_step": admitted // max(1, args.steps + args.warmup + 1), "on_demand_per_step": demanded // max(1, args.steps + args.warmup + 1),
                       "dense_rescans_per_step": dense_rescans // max(1, args.steps + args.warmup + 1)},
            "cpu_baseline": cpu,
            "parity_check_on_timed_path": parity,
            "sweep": sweep,
            "ball": ball,
            "align": aligned,
        }
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()
    failed = []
    if parity is False:
        failed.append("headline")
    if rank == 0:
        failed += ["sweep[%d]" % i for i, e in enumerate(sweep or []) if e.get("parity") is False]
        if ball and ball.get("parity") is False:
            failed.append("ball")
        if aligned and aligned.get("parity_on_sample") is False:
            failed.append("align")
    if failed:
        raise SystemExit("bench.py: the GPU path disagrees with the oracle on the sample of: " + ", ".join(failed))


if __name__ == "__main__":
    main()
