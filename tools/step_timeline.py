#!/usr/bin/env python
"""The last update step of a rocprofv3 --kernel-trace run (rocpd SQLite) as a timeline: per dispatch the start relative to the step's
first dispatch, the duration and the idle gap in front of it (us).  A step is found as the dispatches behind the last gap > 150 us.
Usage: python tools/step_timeline.py RESULTS.db [OUT.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, start, end from kernels order by start"))
    # the last complete step: walk back from the end to the second-last large gap
    cuts = [i for i in range(1, len(rows)) if rows[i][1] - rows[i - 1][2] > 150e3]
    lo, hi = (cuts[-2], cuts[-1]) if len(cuts) >= 2 else (0, len(rows))
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    t0 = rows[lo][1]
    prev_end = None
    busy = 0.0
    for name, s, e in rows[lo:hi]:
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        short = name.split("(")[0].replace("void ovp::", "").replace("ovp::", "")[:40]
        out.write("%9.1f  dur %7.2f  gap %6.2f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, gap, short))
        busy += (e - s) / 1e3
        prev_end = max(prev_end, e) if prev_end is not None else e
    out.write("# dispatches %d, span %.1f us, busy %.1f us\n" % (hi - lo, (rows[hi - 1][2] - t0) / 1e3, busy))


if __name__ == "__main__":
    main()
