"""Writes profiles/hbm_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of profiles/run_profile.sh.

rocprofv3 reports both counters in kilobytes.  On gfx950 FETCH_SIZE tallies the 128-byte read
requests of wide coalesced loads (16 B per lane: the colour-buffer and accumulation-buffer reads
here) at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
bytes = (2 * fetch + write) * 1024, the kernels of one launch: first pass, loop, ordered sum.
Usage: python profiles/summarize_traffic.py gpurun_out/prof_<tag>"""
import glob
import json
import os
import sqlite3
import sys


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402


def per_dispatch(root, counter):
    """Average per dispatch of each kernel of a render launch (first pass, loop, ordered sum), added up."""
    for db_path in sorted(glob.glob(os.path.join(root, "*", "*_results.db"))):
        cur = sqlite3.connect(db_path).cursor()
        total, found = 0.0, False
        for pattern in ("%primaryPass<false,%", "%renderWavefront<true, false, false,%", "%accumulateColors%"):
            try:
                rows = list(cur.execute(
                    "select sum(value) / count(distinct dispatch_id) from counters_collection "
                    "where counter_name = ? and kernel_name like ?", (counter, pattern)))
            except sqlite3.Error:
                rows = []
            if rows and rows[0][0] is not None:
                total += float(rows[0][0])
                found = True
        if found:
            return total
    return None


def render_kernel(root, counter):
    """Average per dispatch of the two render kernels of a launch (first pass + loop), added up."""
    for db_path in sorted(glob.glob(os.path.join(root, "*", "*_results.db"))):
        cur = sqlite3.connect(db_path).cursor()
        total, found = 0.0, False
        for pattern in ("%primaryPass<false,%", "%renderWavefront<true, false, false,%"):
            try:
                rows = list(cur.execute(
                    "select sum(value) / count(distinct dispatch_id) from counters_collection "
                    "where counter_name = ? and kernel_name like ?", (counter, pattern)))
            except sqlite3.Error:
                rows = []
            if rows and rows[0][0] is not None:
                total += float(rows[0][0])
                found = True
        if found:
            return total
    return None


def main(root):
    rd, wr = per_dispatch(root, "FETCH_SIZE"), per_dispatch(root, "WRITE_SIZE")
    valu, active, lanes = (render_kernel(root, c) for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"))
    out = {"source": os.path.basename(os.path.normpath(root)), "fetch_kb_per_launch": rd, "write_kb_per_launch": wr,
           "hbm_bytes_per_launch": None if rd is None or wr is None else int((2.0 * rd + wr) * 1024),
           "valu_insts_per_launch": valu,
           "lane_utilisation": None if not (active and lanes) else lanes / (64.0 * active),
           "workload": "bench.py default: cover scene 1920x1080, 64 spp, 1 GPU",
           # bench.py reports these figures only while the kernel sources still hash to this
           "kernel_source_sha256": kernel_source_hash()}
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "hbm_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out)


if __name__ == "__main__":
    main(sys.argv[1])
