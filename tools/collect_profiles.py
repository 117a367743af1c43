#!/usr/bin/env python3
"""Copies the results of tools/measure_round.sh (gpurun_out/measure/) into profiles/ under round-3 names and writes
profiles/r03_pmc_traffic.json, the block bench.py quotes for `roofline.traffic` (only for the build of the kernels it was measured
on: it carries the hash of the kernel sources).  Usage: python tools/collect_profiles.py"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = os.path.join(ROOT, "gpurun_out", "measure")
P = os.path.join(ROOT, "profiles")


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def main():
    import bench
    for src, dst in (("bench_default.json", "r03_bench_default.json"), ("bench_acgt_c1.json", "r03_bench_c1_acgt.json"),
                     ("stats/c1_kernel_stats.csv", "r03_bench_c1_kernel_stats.csv"),
                     ("pmc_fetch/f_counter_collection.csv", "r03_pmc_fetch_counter_collection.csv"), ("pmc_write/w_counter_collection.csv", "r03_pmc_write_counter_collection.csv"),
                     ("pmc_fetch_q4/f_counter_collection.csv", "r03_pmc_fetch_q4_counter_collection.csv"),
                     ("pmc_sqa/a_counter_collection.csv", "r03_pmc_sq_a_counter_collection.csv"), ("pmc_sqb/b_counter_collection.csv", "r03_pmc_sq_b_counter_collection.csv"),
                     ("q4/q4_kernel_stats.csv", "r03_q4_1Mrefs_kernel_stats.csv"), ("q4/q4_kernel_trace.csv", "r03_q4_1Mrefs_kernel_trace.csv"),
                     ("hbm_read.txt", "r03_hbm_read_ceiling.txt"), ("push_rate.json", "r03_push_rate.json"), ("ingest.json", "r03_ingest_text_vs_packed.json"),
                     ("emu_refshard_2.json", "r03_emulated_reference_shards_2.json"), ("emu_refshard_4.json", "r03_emulated_reference_shards_4.json"),
                     ("emu_refshard_8.json", "r03_emulated_reference_shards_8.json"), ("align_cli.json", "r03_uvaialign_cli_10k.json"),
                     ("prune_timing.txt", "r03_query_preparation_timing.txt"),
                     ("ball/ball_kernel_stats.csv", "r03_ball_kernel_stats.csv"), ("ball.json", "r03_ball.json"),
                     ("c1trace/c1_kernel_trace.csv", "r03_config1_kernel_trace.csv")):
        if os.path.exists(os.path.join(M, src)):
            shutil.copyfile(os.path.join(M, src), os.path.join(P, dst))
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
                   "every packed byte exactly once. FETCH_SIZE counts L2 misses (Infinity-Cache hits included).",
           "kernel_source_hash": bench.kernel_source_hash(), "raw": summ}
    scan_f, scan_w = pick("config1_fetch", "scan3_kernel"), pick("config1_write", "scan3_kernel")
    if scan_f and scan_w:
        # full-size launches only (a step may end with a shorter slice): the maximum per dispatch
        fetch_kb, write_kb = scan_f["FETCH_SIZE"]["max"], scan_w["WRITE_SIZE"]["max"]
        e = {"config": {"queries": b["config"]["queries"], "refs_per_gpu": b["config"]["refs_per_gpu"], "pool": b["config"]["pool"], "mode": b["config"]["mode"]},
             "hbm_side_read_bytes_per_launch": fetch_kb * 1024 * 2, "write_bytes_per_launch": write_kb * 1024,
             "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"],
             "kernel_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"] * b["roofline"]["kernel_bytes_per_ref"] / b["roofline"]["algorithmic_bytes_per_ref"]}
        sqa, sqb = pick("config1_sq_a", "scan3_kernel"), pick("config1_sq_b", "scan3_kernel")
        if sqa and sqb:
            tot = {k: v["sum"] for k, v in {**sqa, **sqb}.items()}
            n_instr = sum(tot.get(k, 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
            n_disp = sqa["SQ_INSTS_VALU"]["n"]
            per_launch = n_instr / n_disp
            rate = per_launch / (b["roofline"]["avg_launch_ms"] * 1e-3) / 1e9
            e["instruction_mix_all_dispatches"] = tot
            e["wave_instructions_per_launch"] = per_launch
            e["issue"] = {"wave_instructions_per_launch": round(per_launch), "achieved": round(rate, 1), "peak": 1037.0, "unit": "G wave-instr/s", "frac": round(rate / 1037.0, 3),
                          "mix": {k[9:]: round(tot[k] / n_instr, 3) for k in tot if k.startswith("SQ_INSTS_")},
                          "source": "SQ passes of this file over avg_launch_ms of the bench line; peak: profiles/r01_issue_rate_microbench.txt (VALU + SALU mixed, whole chip)"}
        out["scan3_kernel"] = e
    q4f = pick("q4_1Mrefs_fetch", "scan2_iupac_kernel")
    if q4f:
        total_kb = q4f["FETCH_SIZE"]["sum"] / (q4f["FETCH_SIZE"]["n"] / 5.0)        # five launches per step (short head and tail slices)
        out["q4_1Mrefs_check"] = {"kernel": "scan2_iupac_kernel", "fetch_bytes_corrected_per_step": total_kb * 1024 * 2, "packed_bytes_per_step": 1000000 * 14976,
                                  "ratio": total_kb * 1024 * 2 / (1000000 * 14976.0)}
    json.dump(out, open(os.path.join(P, "r03_pmc_traffic.json"), "w"), indent=1)
    print("profiles updated")


if __name__ == "__main__":
    main()
