#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (`*_results.db`) as text: per-kernel call count,
average / total duration and, when present, PMC counter averages.  Used to turn the scratch
output under gpurun_out/ into the committed summaries under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    meta = [t for t in tabs if t.startswith("rocpd_metadata")]
    suffix = meta[0][len("rocpd_metadata"):] if meta else ""
    kd, ks = "rocpd_kernel_dispatch" + suffix, "rocpd_info_kernel_symbol" + suffix
    pe, ip = "rocpd_pmc_event" + suffix, "rocpd_info_pmc" + suffix
    rows = list(cur.execute(
        f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start), "
        f"avg(d.grid_size_x), avg(d.workgroup_size_x) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc"))
    total = sum(r[5] for r in rows) or 1
    print("# source: %s" % path)
    print("%-100s %8s %10s %9s %9s %12s %6s %10s" % ("kernel", "calls", "avg_ns", "min_ns", "max_ns", "total_ns", "pct", "avg_grid"))
    for r in rows:
        print("%-100s %8d %10.0f %9d %9d %12d %6.2f %10.0f" % (r[0][:100], r[1], r[2], r[3], r[4], r[5], 100.0 * r[5] / total, r[6]))
    try:
        prow = list(cur.execute(
            f"select s.kernel_name, p.name, count(*), avg(e.value), sum(e.value) from {pe} e join {ip} p on e.pmc_id=p.id "
            f"join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by s.kernel_name, p.name order by 5 desc"))
    except sqlite3.Error:
        prow = []
    if prow:
        print()
        print("%-100s %-14s %8s %14s %16s" % ("kernel", "counter", "samples", "avg_per_launch", "sum"))
        for r in prow:
            print("%-100s %-14s %8d %14.2f %16.2f" % (r[0][:100], r[1], r[2], r[3], r[4]))


if __name__ == "__main__":
    main(sys.argv[1])
