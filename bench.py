#!/usr/bin/env python3
"""bench.py -- uvaia nearest-neighbour hot path on MI355X: reference sequences scored per second.

One "step" = one complete nearest-neighbour search: every reference of the HBM-resident packed database is scored
against the whole resident query set and passed through the ordered gate + top-k heaps (what `uvaia` does for one
reference file, src/nearest.c:249-330 of the reference), heaps reset between steps.  Inputs are synthetic
SARS-CoV-2-shaped alignments (uvaia_amd/csrc/host/synth.c, seed 20241008) and are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N=1 runs BASELINE.json config[1]: 1 000 queries x 100 000 references x 29 903 columns, 4-bit IUPAC planes, top-k 100.
With N>1 every rank holds its own 100 000-reference shard of the database (weak scaling, block-cyclic in stream order).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
QUERY_INDEX0 = 1 << 40           # queries come from the same process, disjoint sequence numbers


def parse():
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
    ap.add_argument("--cpu-refs", type=int, default=1536, help="references in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-refs-1thread", type=int, default=48, help="sample of the single-thread CPU baseline")
    ap.add_argument("--multi", choices=["shards", "ring"], default="shards",
                    help="N > 1: 'shards' = every GPU holds the whole database and a range of the queries (no data-path exchange); "
                         "'ring' = every GPU holds 1/N of the database, heap state handed rank to rank (uvaia_amd/ring.py)")
    ap.add_argument("--emulate-shard-of", type=int, default=0, metavar="N",
                    help="single process: do the work of rank 0 of N query shards (whole stream of N x --refs references, 1/N of the queries) "
                         "to measure the per-rank time of an N-GPU run on one GPU; the line is marked as emulated")
    ap.add_argument("--search-only", action="store_true",
                    help="leave the per-query-set planes of the resident database as the load built them (the timed step then "
                         "holds scan + replay only); by default every step rebuilds them first")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run GPU-vs-oracle check on the sample")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    from uvaia_amd import capi, hostlib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # UVAIA_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (state blobs via host memory)
    backend = os.environ.get("UVAIA_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
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
    shard_mode = (world > 1 and args.multi == "shards") or emu > 0
    local_refs = (emu or world) * args.refs if shard_mode else args.refs      # query shards: every rank holds (and scans) the whole stream
    pool = min(args.pool, local_refs)
    eng = pq.open_engine(nbest=args.nbest, max_pool=pool, device=local_rank)
    t_q2 = time.time()
    if args.qt:
        eng.set_query_tile(args.qt)
    eng.db_reserve(local_refs)
    t0 = time.time()
    from uvaia_amd import ring, shards
    # ring: block-cyclic shard, stripe s (= one pool of world*pool references of the stream) = slice s of rank 0, 1, ...
    # shards: the whole stream on every rank
    slices = [ring.Slice(0, local_refs, 0)] if shard_mode else ring.block_cyclic_layout(args.refs, pool, rank, world)
    first = slices[0].ordinal0
    chunk = 8192
    for sl in slices:
        for a in range(0, sl.n, chunk):
            n = min(chunk, sl.n - a)
            rows, non_n = gen.generate(sl.ordinal0 + a, n)
            eng.db_append_block(rows, non_n)
    load_s = time.time() - t0
    bytes_per_ref = eng.packed_bytes_per_ref()
    on_gpu = backend == "nccl"
    comm = ring.TorchRingComm(dist, rank, world, cuda=on_gpu) if (dist is not None and not shard_mode) else None
    nbytes = eng.state_bytes()
    cons = len(pq.idx_c) > 0
    q0, q1 = shards.query_shard(pq.ntax, rank, emu or world) if shard_mode else (0, pq.ntax)
    allmax = shards.TorchMax(dist, "cuda" if on_gpu else "cpu") if (shard_mode and cons and dist is not None) else None

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

    # ---- roofline of the dominant kernel (pair scan): algorithmic bytes per launch / mean launch time (HIP events)
    launches = max(1, scan_launches)
    avg_ms = scan_ms / launches
    W = (args.nchar + 31) // 32
    variant = eng.scan_variant()        # 2 column-compressed, 0 / 1 two counters over the packed planes, -1 four counters
    fullscan = variant == -1
    ops_per_pair_word = (15 if args.mode == "iupac" else 8) if fullscan else 6
    valu_ops = float(local_refs) * (q1 - q0) * W * ops_per_pair_word * args.steps    # lane-ops in the timed region (this rank)
    valu_rate = valu_ops / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
    # The column-compressed scan does not read the 4-bit records themselves but planes derived from them for this query set
    # (E and V planes + the gathered polymorphic columns): fewer bytes per reference than the packed record.
    kernel_bytes_per_ref = bytes_per_ref if variant != 2 else eng.scan_bytes_per_ref()
    # `achieved` is SURVEY 8d's implementation-independent figure: ceil(L*b/8) bytes per reference (14 952 at 4 bits, 11 214 with
    # 2 bits + validity plane), each reference byte once per launch.  The bytes THIS kernel has to read (derived planes, only the
    # word groups some query tile needs) are fewer; the rate on those is given next to it and is the one PMC FETCH_SIZE verifies.
    survey_bytes_per_ref = (args.nchar * 4 + 7) // 8 if args.mode == "iupac" else (args.nchar * 2 + 7) // 8 + (args.nchar + 7) // 8
    refs_per_launch = float(local_refs) * args.steps / launches
    achieved = refs_per_launch * survey_bytes_per_ref / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    on_kernel_bytes = refs_per_launch * kernel_bytes_per_ref / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
        "algorithmic_bytes_per_ref": survey_bytes_per_ref, "kernel_bytes_per_ref": kernel_bytes_per_ref,
        "achieved_on_kernel_bytes": round(on_kernel_bytes, 2), "frac_on_kernel_bytes": round(on_kernel_bytes / HBM_PEAK_GBS, 5),
        "kernel": {-1: "scan_%s_kernel" % args.mode, 0: "scan2_%s_kernel" % args.mode, 1: "scan2v_kernel"}.get(variant, "scan3_kernel"),
        "dense_equivalent_ops_per_pair_word": ops_per_pair_word,
        "avg_launch_ms": round(avg_ms, 4), "launches": scan_launches,
        "algorithmic_bytes_per_launch": refs_per_launch * survey_bytes_per_ref,
        "dense_equivalent_tlaneops_per_s": round(valu_rate, 2),
        "note": (("%d resident queries = one query tile: the scan reads the packed planes of every reference exactly once "
                  "(two-counter kernel, nothing derived); HBM is the bound (DESIGN.md 4.1)") % pq.ntax) if variant == 0 else
                (("at %d resident queries every reference byte is reused by every query tile: the scan is bound by instruction issue "
                  "(VALU popcounts + scalar bookkeeping), not by HBM; the HBM-bound regime is Q <= 16 (profiles/r01_sweep_q*.json, DESIGN.md 4.1)") % pq.ntax),
    }

    if achieved > HBM_PEAK_GBS:
        roofline["frac_note"] = ("the SURVEY figure counts the whole packed record of every reference; with %d queries this kernel reads only the derived "
                                 "planes of the word groups some query needs (kernel_bytes_per_ref), so the nominal rate exceeds the peak: "
                                 "frac_on_kernel_bytes is the physical HBM fraction (PMC-verified, profiles/r01_pmc_traffic.json)") % pq.ntax
    # HBM-side traffic of the scan from the committed PMC passes (rocprofv3 cannot run inside this process); only quoted
    # when the run is the configuration those passes measured
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["scan3_kernel"]
        same = all(pm["config"][k] == v for k, v in (("queries", pq.ntax), ("refs_per_gpu", args.refs), ("pool", pool), ("mode", args.mode)))
        if same and not fullscan and world == 1 and os.environ.get("UVAIA_GPU_SCAN", "") == pm.get("variant", ""):
            roofline["traffic"] = pm["hbm_side_read_bytes_per_launch"] + pm["write_bytes_per_launch"]
            roofline["traffic_note"] = "FETCH_SIZE x2 (gfx950) + WRITE_SIZE per launch, profiles/r01_pmc_traffic.json; L2 misses incl. Infinity-Cache hits"
            if pm.get("wave_instructions_all_dispatches") and pm.get("instruction_mix_all_dispatches", {}).get("SQ_WAVES"):
                # instruction issue: wave-instructions per (16 queries x 64 references) wave from the committed SQ passes, times the
                # waves of the timed region, over the measured scan time; peak = tools/issue_rate.hip (VALU + SALU mixed, whole chip)
                per_wave = pm["wave_instructions_all_dispatches"] / pm["instruction_mix_all_dispatches"]["SQ_WAVES"]
                waves = ((q1 - q0 + 15) // 16) * ((local_refs + 63) // 64) * args.steps
                rate = per_wave * waves / (scan_ms * 1e-3) / 1e9
                roofline["issue"] = {"wave_instructions_per_wave": round(per_wave), "achieved": round(rate, 1), "peak": 1037.0, "unit": "G wave-instr/s",
                                     "frac": round(rate / 1037.0, 3), "source": "profiles/r01_pmc_traffic.json, profiles/r01_issue_rate_microbench.txt"}
    except Exception:
        pass

    # ---- CPU baseline (rank 0, N=1 only): the oracle's restatement of the reference loops on a bounded sample
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not emu and args.cpu_refs > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        n_s = min(args.cpu_refs, args.refs)
        sample, _ = gen.generate_bytes(first, n_s)
        snames = ["ref_%d" % i for i in range(n_s)]
        oq = O.Query(qseqs, qnames, acgt=(args.mode == "acgt"))
        cores = O.lib().orc_max_threads()
        t0 = time.perf_counter()
        gold = O.search(oq, sample, snames, pool=min(pool, n_s), nbest=args.nbest, ambig_r=0.5)
        cpu_s = time.perf_counter() - t0
        cpu = {"value": round(n_s / cpu_s, 2), "unit": "ref-seqs/s", "cores": cores, "kind": "port",
               "sample": "first %d references of the same database vs the same %d queries, pool %d, OpenMP over %d threads, %.1f s"
                         % (n_s, oq.ntax, min(pool, n_s), cores, cpu_s)}
        try:        # the same restatement on one thread, on a smaller sample
            import ctypes
            gomp = ctypes.CDLL("libgomp.so.1")
            n_1 = max(1, min(n_s, args.cpu_refs_1thread))
            gomp.omp_set_num_threads(1)
            t0 = time.perf_counter()
            O.search(oq, sample[:n_1], snames[:n_1], pool=min(pool, n_1), nbest=args.nbest, ambig_r=0.5)
            t1 = time.perf_counter() - t0
            gomp.omp_set_num_threads(cores)
            cpu["value_1_thread"] = round(n_1 / t1, 2)
            cpu["sample_1_thread"] = "first %d references, 1 thread, %.1f s" % (n_1, t1)
        except OSError:
            pass
        try:
            cpu["host"] = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            pass
        if not args.no_parity:      # the same sample through the GPU engine must give the same heaps
            eng.reset()
            with pq.open_engine(nbest=args.nbest, max_pool=min(pool, n_s), device=local_rank) as e2:
                if args.qt:
                    e2.set_query_tile(args.qt)
                e2.push(sample)
                n, T, sc, od = e2.drain()
            rows = capi.finalise_heaps(n, sc, od)
            ok = list(T) == gold.final_T and all(
                rows[iq] == [(tuple(s), o) for o, _, s in gold.rows[iq]] for iq in range(oq.ntax))
            parity = bool(ok)

    if rank == 0:
        out = {
            "metric": "ref-seqs scored/sec", "value": round(value, 2), "unit": "ref-seqs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "step_ms_rank0": step_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "emulated": ("rank 0 of %d query shards on one GPU: `value` is what %d GPUs would reach if every rank took this long" % (emu, emu)) if emu else None,
            "multi_gpu": None if (world == 1 and not emu) else
            ("query shards: every GPU holds all %d references and the heaps of %d of the %d queries (column classes from the whole set); no data-path exchange%s; exact"
             % (local_refs, q1 - q0, pq.ntax, ", one all-reduced int per pool" if cons else "")) if shard_mode else
            "block-cyclic slices of %d refs, concurrent scans, heap state (%d B in %d per-query-group blobs) pipelined rank to rank (RCCL isend/irecv), exact" % (pool, nbytes, world),
            "dtype": "u32 bit-planes / int32 counts", "data": "synthetic (seed %d, preset %d)" % (args.seed, args.preset),
            "config": {"workload": (("BASELINE config[1]: " if (pq.ntax, args.refs, args.nchar, args.nbest) == (1000, 100000, 29903, 100) else
                                     "BASELINE config[2]: " if (pq.ntax, args.refs, args.nchar, args.nbest, args.mode) == (10000, 1000000, 29903, 100, "acgt") else "")
                                    + "%d queries x %d refs/GPU x %d cols, %s, top-k %d, pool %d")
                                   % (pq.ntax, args.refs, args.nchar, "4-bit IUPAC planes" if args.mode == "iupac" else "2-bit + validity planes (--acgt)", args.nbest, pool),
                       "queries": pq.ntax, "refs_per_gpu": args.refs, "nchar": args.nchar, "nbest": args.nbest, "pool": pool,
                       "mode": args.mode, "packed_bytes_per_ref": bytes_per_ref, "db_load_s": round(load_s, 2),
                       "query_prepare_s": round(t_q1 - t_q0, 2), "engine_open_s": round(t_q2 - t_q1, 2)},
            "step_parts": {"includes_derived_planes": not args.search_only, "derived_planes_ms": round(derive_ms, 3), "scan_and_replay_ms": round(search_only_ms, 3),
                           "derived_planes_kernel": {"kernel": "derive_all_kernel", "bound": "hbm", "read_bytes_per_ref": bytes_per_ref, "written_bytes_per_ref": eng.derived_bytes_per_ref(),
                                                     "achieved": round(local_refs * (bytes_per_ref + eng.derived_bytes_per_ref()) / (derive_ms * 1e-3) / 1e9, 1) if (derive_ms > 0 and eng.derived_bytes_per_ref()) else None,
                                                     "peak": HBM_PEAK_GBS, "unit": "GB/s (reads + writes, host-timed over its launches)"},
                           "note": "a step = planes derived from the resident packed records for this query set (uvaia_gpu_db_rederive) + pair scan + ordered replay; "
                                   "the two parts timed on their own after the timed region"},
            "roofline": roofline,
            "replay": {"admissions_per_step": admitted // max(1, args.steps + args.warmup + 1), "on_demand_per_step": demanded // max(1, args.steps + args.warmup + 1),
                       "dense_rescans_per_step": dense_rescans // max(1, args.steps + args.warmup + 1)},
            "cpu_baseline": cpu,
            "parity_check_on_sample": parity,
        }
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
