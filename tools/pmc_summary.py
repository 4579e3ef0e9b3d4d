#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into per-kernel averages (KB per launch).

Usage: pmc_summary.py OUT.json NOTE DIR [DIR ...]      (each DIR = one `rocprofv3 --pmc <counter> -d DIR` pass)
Reads every *counter_collection.csv below the directories; counters are averaged per kernel name and launch.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ov_plane_amd.build import source_tree_hash  # noqa: E402


def main():
    out, note, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    a = acc[r["Kernel_Name"]][r["Counter_Name"]]
                    a[0] += float(r["Counter_Value"])
                    a[1].add((f, r["Dispatch_Id"]))
    kernels = {}
    for k, cs in acc.items():
        e = {}
        for cname, (tot, disp) in cs.items():
            e["%s_KB_avg_per_launch" % cname if cname.endswith("_SIZE") else "%s_avg_per_launch" % cname] = tot / max(len(disp), 1)
            e["launches"] = len(disp)
        kernels[k] = e
    with open(out, "w") as fh:
        # source_hash: identity of the kernel sources the passes ran on (ov_plane_amd/build.py); bench.py quotes this file only when
        # it equals the running tree's
        json.dump({"note": note, "source_hash": source_tree_hash(), "kernels": kernels}, fh, indent=1)
    for k, e in kernels.items():
        print(k[:70], e)


if __name__ == "__main__":
    main()
