#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd sqlite .db, the default output of ROCm 7.2) into a small markdown/CSV pair that can
be committed under profiles/.  Usage: rocprof_summary.py <results.db> <out_prefix> [note]"""
import csv
import sqlite3
import sys


def main():
    db, prefix = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(workgroup_x), max(grid_x), "
                          "max(lds_size), max(vgpr_count), max(sgpr_count), max(scratch_size) from kernels group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows) or 1
    with open(prefix + ".csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "WorkgroupX", "GridX", "LDS", "VGPR", "SGPR", "Scratch"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], "%.1f" % r[3], r[4], r[5], "%.3f" % (100.0 * r[2] / total)] + list(r[6:]))
    with open(prefix + ".md", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary\n\n%s\n\n" % note)
        f.write("| kernel | calls | avg (us) | min (us) | max (us) | % of GPU time | wg | grid | LDS B | VGPR | SGPR |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write("| `%s` | %d | %.1f | %.1f | %.1f | %.2f | %d | %d | %d | %d | %d |\n" % (
                r[0][:90], r[1], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10]))
    print(open(prefix + ".md").read())


if __name__ == "__main__":
    main()
