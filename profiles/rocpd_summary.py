#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd SQLite database (the default output of
`rocprofv3 --kernel-trace --stats`) into the small CSV summaries committed here.

  python profiles/rocpd_summary.py gpurun_out/prof/x_results.db profiles/r01_name
writes  <prefix>_kernel_stats.csv (per-kernel calls / total / average / %)
        <prefix>_dispatches.csv   (every dispatch of our kernels with launch geometry and registers)
"""
import csv
import sqlite3
import sys


def main(db_path, prefix):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    with open(prefix + "_kernel_stats.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_us", "average_us", "percent"])
        for name, calls, total, avg, pct in cur.execute(
                "select name,total_calls,total_duration,average,percentage from top_kernels"):
            w.writerow([name[:160], calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.4f}"])
    with open(prefix + "_dispatches.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "duration_us", "grid_x", "grid_y", "workgroup_x", "lds_bytes", "scratch", "vgpr",
                    "accum_vgpr", "sgpr"])
        for r in cur.execute("select name,duration,grid_x,grid_y,workgroup_x,lds_size,scratch_size,vgpr_count,"
                             "accum_vgpr_count,sgpr_count from kernels where name like '%zpq%' order by start"):
            w.writerow([r[0][:100], f"{r[1] / 1e3:.3f}"] + list(r[2:]))
    print("wrote", prefix + "_kernel_stats.csv", prefix + "_dispatches.csv")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
