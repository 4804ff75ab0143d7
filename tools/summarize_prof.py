#!/usr/bin/env python3
"""Condense rocprofv3 output directories (kernel-trace stats + PMC csv) into a
short text summary: per-kernel time, and per-kernel counter sums."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("pylda::", "")
    return name[:70]


def main(root):
    for path in sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)):
        print("== kernel stats:", os.path.relpath(path, root))
        rows = list(csv.DictReader(open(path)))
        for r in rows[:14]:
            print("  %-70s calls %6s  total %10.3f ms  avg %10.3f us  %5s%%" % (
                short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                float(r["AverageNs"]) / 1e3, r["Percentage"]))
    for d in sorted(glob.glob(os.path.join(root, "pmc*"))):
        if not os.path.isdir(d):
            continue
        for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            acc = defaultdict(lambda: defaultdict(float))
            calls = defaultdict(set)
            for r in csv.DictReader(open(path)):
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[k].add(r["Dispatch_Id"])
            print("== counters:", os.path.relpath(path, root))
            tot = {k: sum(v.values()) for k, v in acc.items()}
            for k in sorted(acc, key=lambda k: -tot[k])[:6]:
                n = max(1, len(calls[k]))
                print("  %s  (%d dispatches; per-dispatch averages)" % (k, n))
                for c, v in sorted(acc[k].items()):
                    print("      %-24s %16.1f" % (c, v / n))


if __name__ == "__main__":
    main(sys.argv[1])
