#!/usr/bin/env python3
"""Timeline of the last N kernel dispatches in a rocprofv3 rocpd database: start offset, duration and
the idle gap before each (us) -- shows where a launch-bound step leaves the GPU dry."""
import sqlite3
import sys


def main(db_path, n=140, out=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select d.start, d.end, s.kernel_name, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id "
                       f"order by d.start desc limit {int(n)}").fetchall()[::-1]
    t0, prev_end, busy = rows[0][0], rows[0][0], 0
    lines = []
    for st, en, name, q in rows:
        lines.append(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f} {max(0, st - prev_end) / 1e3:7.1f} q{q} {name[:90]}")
        busy += en - st
        prev_end = max(prev_end, en)
    lines.append(f"# span {(prev_end - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us")
    text = "\n".join(["# start_us   dur_us  gap_us queue kernel"] + lines)
    if out:
        open(out, "w").write(text + "\n")
    else:
        print(text)


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:]))
