#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database by default; this turns its kernel-dispatch table into the
--stats CSV layout (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev), optionally split by grid
size so that one kernel template used at several shapes shows one row per shape.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db profiles/r02_rocprofv3_kernel_stats.csv [--by-grid]"""
import csv
import sqlite3
import statistics
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    by_grid = "--by-grid" in sys.argv
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, grid_x, workgroup_x, duration from kernels").fetchall()
    groups: dict = {}
    for name, gx, wx, d in rows:
        key = (name, gx // max(wx, 1)) if by_grid else (name,)
        groups.setdefault(key, []).append(d)
    total = sum(sum(v) for v in groups.values())
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"] + (["Workgroups"] if by_grid else []))
        for key, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([key[0], len(v), sum(v), round(sum(v) / len(v), 3), round(100 * sum(v) / total, 2), min(v), max(v),
                        round(statistics.pstdev(v), 3)] + ([key[1]] if by_grid else []))


if __name__ == "__main__":
    main()
