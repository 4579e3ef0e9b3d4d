#!/usr/bin/env python
"""Per-kernel statistics of a rocprofv3 run (rocpd SQLite output of `rocprofv3 --kernel-trace --stats`) as CSV:
name, calls, total / average / min / max duration in microseconds, share of the GPU time, launch geometry and register counts.
Usage: python tools/rocpd_stats.py RESULTS.db [OUT.csv]"""
import csv
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ov_plane_amd.build import source_tree_hash  # noqa: E402


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(grid_x), max(workgroup_x), "
        "max(lds_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(scratch_size) from kernels group by name "
        "order by 3 desc"))
    tot = float(sum(r[2] for r in rows)) or 1.0
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    out.write("# source_hash %s (ov_plane_amd/build.py: source_tree_hash - the kernel sources this run was taken on)\n" % source_tree_hash())
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "MinUs", "MaxUs", "Percentage", "GridX", "WorkgroupX", "LDS",
                "VGPR", "AGPR", "SGPR", "Scratch"])
    for r in rows:
        w.writerow([r[0], r[1], "%.1f" % (r[2] / 1e3), "%.2f" % (r[3] / 1e3), "%.2f" % (r[4] / 1e3), "%.2f" % (r[5] / 1e3),
                    "%.2f" % (100.0 * r[2] / tot)] + list(r[6:]))


if __name__ == "__main__":
    main()
