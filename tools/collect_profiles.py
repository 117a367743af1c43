#!/usr/bin/env python3
"""Copies the results of tools/measure_round.sh (gpurun_out/measure/) into profiles/ under round-4 names and writes
profiles/r04_pmc_traffic.json, the block bench.py quotes for `roofline.traffic`, `roofline.issue` and `roofline.waits` (only for the
build of the kernels it was measured on: it carries the hash of the kernel sources).  Usage: python tools/collect_profiles.py"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = os.path.join(ROOT, "gpurun_out", "measure")
P = os.path.join(ROOT, "profiles")
R = "r04"
ISSUE_PEAK = 1037.0       # G wave-instructions/s, VALU + SALU mixed, whole chip: profiles/r01_issue_rate_microbench.txt


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def first(pattern):
    g = sorted(glob.glob(os.path.join(M, pattern), recursive=True))
    return g[0] if g else None


def main():
    import bench
    copies = [("bench_default.json", "bench_default.json"), ("bench_acgt_c1.json", "bench_c1_acgt.json"),
              ("stats/**/c1_kernel_stats.csv", "bench_c1_kernel_stats.csv"), ("stats2/**/c2_kernel_stats.csv", "config2_kernel_stats.csv"),
              ("q4/**/q4_kernel_stats.csv", "q4_1Mrefs_kernel_stats.csv"), ("q4/**/q4_kernel_trace.csv", "q4_1Mrefs_kernel_trace.csv"),
              ("q16/**/q16_kernel_stats.csv", "q16_1Mrefs_kernel_stats.csv"), ("q16/**/q16_kernel_trace.csv", "q16_1Mrefs_kernel_trace.csv"),
              ("c1trace/**/c1_kernel_trace.csv", "config1_kernel_trace.csv"),
              ("ball/**/ball_kernel_stats.csv", "ball_kernel_stats.csv"), ("ball.json", "ball.json"),
              ("hbm_read.txt", "hbm_read_ceiling.txt"), ("push_rate.json", "push_rate.json"), ("ingest.json", "ingest_text_vs_packed.json"),
              ("emu_refshard_2.json", "emulated_reference_shards_2.json"), ("emu_refshard_4.json", "emulated_reference_shards_4.json"),
              ("emu_refshard_8.json", "emulated_reference_shards_8.json"), ("emu_refshard_8_config3.json", "emulated_reference_shards_8_config3_regime.json"),
              ("load_latency.txt", "load_latency_next_to_a_stream.txt"), ("probe_q4.txt", "small_query_replay_tiles_opened.txt"),
              ("probe_q4_timing.txt", "small_query_replay_time_split.txt"), ("probe_c1_timing.txt", "config1_replay_time_split.txt")]
    for d in ("c1_fetch", "c1_write", "c1_insts_a", "c1_waits", "c1_units", "c1_tcc", "c2_fetch", "c2_write", "c2_insts_a", "c2_waits", "c2_tcc", "q4_fetch", "q16_insts", "q16_waits"):
        copies.append(("pmc_%s/**/*counter_collection.csv" % d, "pmc_%s_counter_collection.csv" % d))
    for src, dst in copies:
        f = first(src)
        if f:
            shutil.copyfile(f, os.path.join(P, "%s_%s" % (R, dst)))
    if not os.path.exists(os.path.join(M, "pmc_summary.json")):
        print("no PMC passes to summarise")
        return
    summ = json.load(open(os.path.join(M, "pmc_summary.json")))
    b = last_json(os.path.join(M, "bench_default.json"))

    def pick(block, name):
        for k, v in summ.get(block, {}).items():
            if name in k:
                return v
        return {}

    out = {"note": "rocprofv3 --pmc, one counter group per pass (tools/measure_round.sh). FETCH_SIZE/WRITE_SIZE are KB per dispatch; gfx950 FETCH_SIZE "
                   "reports half of wide coalesced reads (MI355X_MICROARCH.md), hence x2; checked on the packed-plane scan of the Q = 4 run below, which has to read "
                   "every packed byte exactly once. FETCH_SIZE counts L2 misses (Infinity-Cache hits included). SQ counters are summed over the chip per dispatch.",
           "kernel_source_hash": bench.kernel_source_hash(), "entries": [], "raw": summ}

    def entry(tag, kernel, cfg, algorithmic_bytes_per_launch, avg_launch_ms):
        f, w = pick(tag + "_fetch", kernel), pick(tag + "_write", kernel)
        if not f or not w:
            return
        # the mean over the dispatches, like the launch time and the algorithmic bytes it is set against (a pool's first slice is shorter than the others)
        e = {"kernel": "scan3_kernel", "config": cfg, "hbm_side_read_bytes_per_launch": f["FETCH_SIZE"]["mean"] * 1024 * 2, "write_bytes_per_launch": w["WRITE_SIZE"]["mean"] * 1024,
             "algorithmic_bytes_per_launch": algorithmic_bytes_per_launch}
        ia = pick(tag + "_insts_a", kernel)
        if ia:
            n_disp = ia["SQ_INSTS_VALU"]["n"]
            tot = {k: v["sum"] for k, v in ia.items() if k.startswith("SQ_INSTS_")}
            per_launch = sum(tot.values()) / n_disp
            e["issue"] = {"wave_instructions_per_launch": round(per_launch), "peak": ISSUE_PEAK, "unit": "G wave-instr/s",
                          "mix": {k[9:]: round(v / sum(tot.values()), 3) for k, v in tot.items()},
                          "source": "SQ_INSTS_* passes of profiles/%s_pmc_%s_insts_a_counter_collection.csv; peak: profiles/r01_issue_rate_microbench.txt (VALU + SALU mixed, whole chip)" % (R, tag)}
            if avg_launch_ms:
                e["issue"]["achieved"] = round(per_launch / (avg_launch_ms * 1e-3) / 1e9, 1)
                e["issue"]["frac"] = round(e["issue"]["achieved"] / ISSUE_PEAK, 3)
        wt, tc = pick(tag + "_waits", kernel), pick(tag + "_tcc", kernel)
        if wt:
            wc = wt["SQ_WAVE_CYCLES"]["sum"]
            e["waits"] = {"of_wave_cycles": {"waiting_at_any_counter": round(wt["SQ_WAIT_ANY"]["sum"] / wc, 3), "waiting_for_an_issue_slot": round(wt["SQ_WAIT_INST_ANY"]["sum"] / wc, 3),
                                             "of_that_lds": round(wt["SQ_WAIT_INST_LDS"]["sum"] / wc, 3), "issuing": round(wt["SQ_ACTIVE_INST_ANY"]["sum"] / wc, 3),
                                             "issuing_valu": round(wt["SQ_ACTIVE_INST_VALU"]["sum"] / wc, 3), "issuing_scalar": round(wt["SQ_ACTIVE_INST_SCA"]["sum"] / wc, 3)},
                          "source": "profiles/%s_pmc_%s_waits_counter_collection.csv" % (R, tag)}
        if tc:
            req = tc["TCC_REQ_sum"]["mean"]
            e["l2"] = {"requests_per_launch": round(req), "hit_rate": round(tc["TCC_HIT_sum"]["mean"] / max(req, 1), 3), "miss_bytes_per_launch_128B": round(tc["TCC_MISS_sum"]["mean"] * 128),
                       "source": "profiles/%s_pmc_%s_tcc_counter_collection.csv" % (R, tag)}
        out["entries"].append(e)

    entry("c1", "scan3_kernel", {"queries": b["config"]["queries"], "mode": b["config"]["mode"], "refs_per_gpu": b["config"]["refs_per_gpu"], "pool": b["config"]["pool"]},
          b["roofline"]["algorithmic_bytes_per_launch"], b["roofline"]["avg_launch_ms"])
    c2 = [e_ for e_ in (b.get("sweep") or []) if e_.get("queries") == 10000]
    if c2:
        entry("c2", "scan3_kernel", {"queries": 10000, "mode": "acgt", "measured_on": "125 000 references (five launches of the sweep entry's length)"},
              c2[0]["roofline"]["algorithmic_bytes_per_launch"], c2[0]["roofline"]["avg_launch_ms"])
    q4f = pick("q4_fetch", "scan2_iupac_kernel")
    if q4f:
        total_kb = q4f["FETCH_SIZE"]["sum"] / 6.0        # the pass runs six steps (set-up, one warm-up, two timed, two search-only), each reading the whole database once
        out["q4_1Mrefs_check"] = {"kernel": "scan2_iupac_kernel", "fetch_bytes_corrected_per_step": total_kb * 1024 * 2, "packed_bytes_per_step": 1000000 * 14976,
                                  "ratio": total_kb * 1024 * 2 / (1000000 * 14976.0)}
    q16i, q16w = pick("q16_insts", "scan2_iupac_kernel"), pick("q16_waits", "scan2_iupac_kernel")
    if q16i and q16w:
        waves_per_step = 15625 * 4                       # one block of four waves per tile of 64 references, one query tile
        n_steps = q16i["SQ_WAVES"]["sum"] / waves_per_step
        valu = q16i["SQ_INSTS_VALU"]["sum"] / n_steps
        out["q16_1Mrefs_scan"] = {"kernel": "scan2_iupac_kernel<16>", "steps_in_pass": round(n_steps, 2), "valu_wave_instructions_per_step": round(valu),
                                  "salu_per_step": round(q16i["SQ_INSTS_SALU"]["sum"] / n_steps), "smem_per_step": round(q16i["SQ_INSTS_SMEM"]["sum"] / n_steps),
                                  "valu_pipe_ms_per_step": round(valu * 4 / (1024 * 2.4e9) * 1e3, 3),
                                  "note": "a wave64 vector instruction occupies its SIMD16 for four cycles: 1 024 SIMDs at 2.4 GHz; the launches of a step take 3.8 ms (profiles/%s_q16_1Mrefs_kernel_stats.csv)" % R,
                                  "of_wave_cycles": {"waiting_at_any_counter": round(q16w["SQ_WAIT_ANY"]["sum"] / q16w["SQ_WAVE_CYCLES"]["sum"], 3),
                                                     "waiting_for_an_issue_slot": round(q16w["SQ_WAIT_INST_ANY"]["sum"] / q16w["SQ_WAVE_CYCLES"]["sum"], 3),
                                                     "issuing_valu": round(q16w["SQ_ACTIVE_INST_VALU"]["sum"] / q16w["SQ_WAVE_CYCLES"]["sum"], 3)}}
    json.dump(out, open(os.path.join(P, "%s_pmc_traffic.json" % R), "w"), indent=1)
    print("profiles updated:", len(out["entries"]), "PMC entries")


if __name__ == "__main__":
    main()
