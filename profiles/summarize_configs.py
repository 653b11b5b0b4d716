"""Derived memory-system figures of the configurations whose scene is read through the caches
(profiles/pmc_configs.sh): L1 and L2 hit rates, bytes fetched beyond L2, wait share, lane utilisation,
next to the algorithmic bytes the path counters imply.  Sums over every render dispatch of the run
(warm-up launch included) against the run's total of samples.
Usage: python profiles/summarize_configs.py gpurun_out/prof_<tag>_<case> gpurun_out/prof_<tag>_<case>_case.json"""
import glob
import json
import os
import sqlite3
import sys


def sums(root):
    out, ms, n = {}, 0.0, 0
    for db_path in sorted(glob.glob(os.path.join(root, "*", "*_results.db"))):
        cur = sqlite3.connect(db_path).cursor()
        for name, value in cur.execute(
                "select counter_name, sum(value) from counters_collection where kernel_name like "
                "'%renderWavefront<false, false, false,%' group by counter_name"):
            out[name] = float(value)
        if not n:
            for total, calls in cur.execute("select total_duration, total_calls from top_kernels where name like "
                                            "'%renderWavefront<false, false, false,%'"):
                ms, n = total / 1e3, calls
    return out, ms, n


def main(root, case_path):
    c, ms, n = sums(root)
    case = json.load(open(case_path))
    w, h = case["size"]
    spp_total = case["spp"] + min(case["spp"], 2)          # bench_configs.py runs a warm-up launch of min(spp, 2)
    samples = w * h * spp_total
    rays = samples * case["rays_per_sample"]
    tris = case["n_triangles"] > 0
    alg = rays * (case["node_tests_per_ray"] * 32 + case["prim_tests_per_ray"] * (48 if tris else 16))
    sec = ms / 1e3
    rd32 = c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    rd = c.get("TCC_EA0_RDREQ_sum", 0.0)
    out = {
        "case": case["case"], "dispatches": n, "kernel_ms_total": round(ms, 2), "samples": samples,
        "l1_hit_rate": round(1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_ACCESSES_sum"], 4),
        "l2_hit_rate": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4),
        "l1_requests_per_ray": round(c["TCP_TOTAL_ACCESSES_sum"] / rays, 1),
        "l2_read_requests_per_ray": round(c["TCP_TCC_READ_REQ_sum"] / rays, 1),
        "l2_read_latency_cycles": round(c["TCP_TCC_READ_REQ_LATENCY_sum"] / max(c["TCP_TCC_READ_REQ_sum"], 1.0), 0),
        "beyond_l2_read_GB": round((rd32 * 32 + (rd - rd32) * 64) / 1e9, 2),
        "beyond_l2_read_GBps": round((rd32 * 32 + (rd - rd32) * 64) / 1e9 / sec, 0),
        "fetch_size_GB_as_counted": round(c["FETCH_SIZE"] * 1024 / 1e9, 2),
        "write_size_GB": round(c["WRITE_SIZE"] * 1024 / 1e9, 2),
        "algorithmic_fetch_GB": round(alg / 1e9, 1),
        "algorithmic_fetch_GBps": round(alg / 1e9 / sec, 0),
        "hbm_peak_GBps": 8000,
        "frac_of_hbm_peak_beyond_l2": round((rd32 * 32 + (rd - rd32) * 64) / 1e9 / sec / 8000, 3),
        "wait_share_of_wave_time": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
        "issue_share_of_wave_time": round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
        "lane_utilisation": round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 3),
        "valu_insts_per_ray": round(c["SQ_INSTS_VALU"] * 64 * (c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])) / rays, 0),
        "vmem_reads_in_flight_per_wave": round(c["SQ_INST_LEVEL_VMEM"] / c["SQ_WAVE_CYCLES"], 2),
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
