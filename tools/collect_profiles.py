#!/usr/bin/env python3
"""Copies the results of tools/measure_round.sh (gpurun_out/measure/) into profiles/ under round-1 names and writes
profiles/r01_pmc_traffic.json, the block bench.py quotes for `roofline.traffic`.  Usage: python tools/collect_profiles.py"""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M = os.path.join(ROOT, "gpurun_out", "measure")
P = os.path.join(ROOT, "profiles")


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def main():
    for src, dst in (("bench_c2.json", "r01_bench_c2_final.json"), ("bench_acgt_c2.json", "r01_bench_c2_acgt.json"), ("bench_c3.json", "r01_bench_c3_acgt_10kx1M.json"),
                     ("sweep_q1.json", "r01_sweep_q1_1Mrefs.json"), ("sweep_q4.json", "r01_sweep_q4_1Mrefs.json"), ("sweep_q16.json", "r01_sweep_q16_1Mrefs.json"),
                     ("sweep_q64.json", "r01_sweep_q64_1Mrefs.json"), ("stats/c2_kernel_stats.csv", "r01_bench_c2_kernel_stats_final.csv"),
                     ("pmc_fetch/f_counter_collection.csv", "r01_pmc_fetch_counter_collection.csv"), ("pmc_write/w_counter_collection.csv", "r01_pmc_write_counter_collection.csv"),
                     ("pmc_fetch_q4/f_counter_collection.csv", "r01_pmc_fetch_q4_counter_collection.csv"),
                     ("pmc_sqa/a_counter_collection.csv", "r01_pmc_sq_a_counter_collection.csv"), ("pmc_sqb/b_counter_collection.csv", "r01_pmc_sq_b_counter_collection.csv"),
                     ("hbm_read.txt", "r01_hbm_read_ceiling.txt"), ("emu_2.json", "r01_emulated_query_shard_of_2.json"), ("emu_4.json", "r01_emulated_query_shard_of_4.json"),
                     ("emu_8.json", "r01_emulated_query_shard_of_8.json")):
        if os.path.exists(os.path.join(M, src)):
            shutil.copyfile(os.path.join(M, src), os.path.join(P, dst))
    summ = json.load(open(os.path.join(M, "pmc_summary.json")))
    bench = last_json(os.path.join(M, "bench_c2.json"))
    q4 = last_json(os.path.join(M, "sweep_q4.json"))

    def pick(block, name):
        for k, v in summ.get(block, {}).items():
            if name in k:
                return v
        return {}

    scan_f, scan_w = pick("config1_fetch", "scan3_kernel"), pick("config1_write", "scan3_kernel")
    # full-size launches only (a step may end with a shorter slice): use the maximum per dispatch
    fetch_kb, write_kb = scan_f["FETCH_SIZE"]["max"], scan_w["WRITE_SIZE"]["max"]
    q4_kb = pick("q4_1Mrefs_fetch", "scan3_kernel")["FETCH_SIZE"]["mean"]
    # the PMC pass on 4 queries measures the column-compressed scan (forced: 4 queries get the packed-plane scan by default), whose
    # bytes per reference at 4 queries are 4 148 (uvaia_gpu_scan_bytes_per_ref); the sweep file may describe the other kernel
    q4_bpr = q4["roofline"]["kernel_bytes_per_ref"] if q4["roofline"]["kernel"] == "scan3_kernel" else 4148
    q4_alg = q4_bpr * q4["config"]["refs_per_gpu"]
    out = {
        "note": "rocprofv3 --pmc, one counter group per pass (tools/measure_round.sh). FETCH_SIZE/WRITE_SIZE are KB per dispatch; gfx950 FETCH_SIZE "
                "reports half of wide coalesced reads (MI355X_MICROARCH.md), hence x2. Check on the one-launch Q=4 run below: corrected fetch / bytes the kernel "
                "has to read. FETCH_SIZE counts L2 misses (Infinity-Cache hits included).",
        "scan3_kernel": {
            "config": {"queries": bench["config"]["queries"], "refs_per_gpu": bench["config"]["refs_per_gpu"], "pool": bench["config"]["pool"], "mode": bench["config"]["mode"]},
            "variant": "",
            "hbm_side_read_bytes_per_launch": fetch_kb * 1024 * 2, "write_bytes_per_launch": write_kb * 1024,
            "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
            "kernel_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"] * bench["roofline"]["kernel_bytes_per_ref"] / bench["roofline"]["algorithmic_bytes_per_ref"],
        },
        "q4_1Mrefs_one_launch_check": {"kernel": "scan3_kernel (UVAIA_GPU_SCAN=compressed)", "fetch_bytes_corrected": q4_kb * 1024 * 2, "kernel_bytes": q4_alg, "ratio": q4_kb * 1024 * 2 / q4_alg},
        "raw": summ,
    }
    sqa, sqb = pick("config1_sq_a", "scan3_kernel"), pick("config1_sq_b", "scan3_kernel")
    if sqa and sqb:
        tot = {k: v["sum"] for k, v in {**sqa, **sqb}.items()}
        n_instr = sum(tot.get(k, 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
        out["scan3_kernel"]["instruction_mix_all_dispatches"] = tot
        out["scan3_kernel"]["wave_instructions_all_dispatches"] = n_instr
    json.dump(out, open(os.path.join(P, "r01_pmc_traffic.json"), "w"), indent=1)
    print("profiles updated")


if __name__ == "__main__":
    main()
