"""Turns the rocprofv3 result databases of profiles/run_profile.sh into a text summary.
Usage: python profiles/summarize.py gpurun_out/prof_<tag> > profiles/<tag>_summary.txt"""
import glob
import os
import sqlite3
import sys


def main(root):
    for db_path in sorted(glob.glob(os.path.join(root, "*", "*_results.db"))):
        db = sqlite3.connect(db_path)
        cur = db.cursor()
        print(f"== {os.path.relpath(db_path, root)}")
        print("-- kernel stats (name, calls, total_us, avg_us, pct)")
        for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            print(f"{r[0][:90]:90s} calls={r[1]:4d} total_us={r[2]:12.1f} avg_us={r[3]:12.1f} pct={r[4]:6.2f}")
        # every dispatch of the render kernels, in order: the first launch of a process also pays for the first use of
        # its freshly allocated buffers, which the average above includes and the median does not
        try:
            tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
            kd = [t for t in tabs if "kernel_dispatch" in t][0]
            sym = [t for t in tabs if "info_kernel_symbol" in t][0]
            per = {}
            for name, t0, t1 in cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {sym} s on d.kernel_id = s.id "
                                            f"where s.kernel_name like '%rt%' order by d.start"):
                per.setdefault(name, []).append((t1 - t0) / 1e3)
            print("-- durations per dispatch in us (mangled name, median, all)")
            for name, d in per.items():
                med = sorted(d)[len(d) // 2]
                print(f"{name[:60]:60s} median_us={med:10.1f}  " + " ".join(f"{x:.0f}" for x in d[:12]))
        except (sqlite3.Error, IndexError) as e:
            print("dispatch query failed:", e)
        try:
            rows = list(cur.execute(
                "select kernel_name, counter_name, count(distinct dispatch_id), sum(value), "
                "sum(value) / count(distinct dispatch_id), max(vgpr_count), max(sgpr_count), "
                "max(lds_block_size), max(workgroup_size), max(grid_size) from counters_collection "
                "where (kernel_name like '%render%' or kernel_name like '%primaryPass%') group by kernel_name, counter_name"))
        except sqlite3.Error as e:
            rows = []
            print("pmc query failed:", e)
        if rows:
            print("-- PMC per render kernel (counter, dispatches, sum, avg per dispatch)")
            for r in rows:
                print(f"{r[0][:50]:50s} {r[1]:28s} n={r[2]:3d} sum={r[3]:18.0f} avg={r[4]:18.1f} "
                      f"vgpr={r[5]} sgpr={r[6]} lds={r[7]} wg={r[8]} grid={r[9]}")
        db.close()


if __name__ == "__main__":
    main(sys.argv[1])
