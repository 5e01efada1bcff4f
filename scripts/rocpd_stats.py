#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (ROCm 7.2 default --kernel-trace output) as a per-kernel
stats CSV: name, calls, avg/min/max/total duration (us), % of GPU kernel time, grid, VGPRs."""
import csv
import sqlite3
import sys


def main(db_path, out_csv):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
         f"sum(d.end-d.start), max(d.grid_size_x), max(d.workgroup_size_x), max(s.arch_vgpr_count), "
         f"max(d.group_segment_size) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc")
    rows = cur.execute(q).fetchall()
    tot = sum(r[5] for r in rows) or 1
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "avg_us", "min_us", "max_us", "total_us", "pct", "grid_x", "wg_x", "vgprs", "lds_bytes"])
        for r in rows:
            w.writerow([r[0], r[1], f"{r[2]/1e3:.2f}", f"{r[3]/1e3:.2f}", f"{r[4]/1e3:.2f}", f"{r[5]/1e3:.1f}",
                        f"{100*r[5]/tot:.2f}", r[6], r[7], r[8], r[9]])
    print(f"{out_csv}: {len(rows)} kernels, {tot/1e3:.0f} us total")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
