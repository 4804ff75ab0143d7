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
            for k in sorted(acc, key=lambda k: -tot[k])[:10]:
                n = max(1, len(calls[k]))
                print("  %s  (%d dispatches; per-dispatch averages)" % (k, n))
                for c, v in sorted(acc[k].items()):
                    print("      %-24s %16.1f" % (c, v / n))


def estep_traffic(root):
    """HBM bytes of one E-step (document kernels + statistics pass) from the FETCH_SIZE / WRITE_SIZE passes:
    the counters of ALL dispatches of the E-step's kernels divided by the number of E-steps in the pass (a kernel
    may run several times per E-step: launch classes, rounds of the statistics gather), FETCH_SIZE doubled (gfx950
    counts the 128-byte requests of wide coalesced reads at 64 bytes: MI355X_MICROARCH.md, HBM)."""
    import json
    import sys as _sys
    _sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    mine = ("estep_", "sstats_", "doc_terms")
    out, esteps = {}, {}
    for counter, sub in (("FETCH_SIZE", "pmc3"), ("WRITE_SIZE", "pmc4")):
        acc, once = defaultdict(float), set()
        for path in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                k = short(r["Kernel_Name"]).replace("void ", "")
                if k.startswith("eta_rowsum_psi"):
                    once.add(r["Dispatch_Id"])              # one per E-step (table preparation)
                if r["Counter_Name"] != counter:
                    continue
                if k.startswith(mine) and "logspace" not in k:
                    acc[k] += float(r["Counter_Value"])
        n = max(1, len(once))
        esteps[counter] = n
        out[counter] = {k: acc[k] / n for k in acc}
    if not out["FETCH_SIZE"]:
        return
    fetch_kb = sum(out["FETCH_SIZE"].values())
    write_kb = sum(out["WRITE_SIZE"].values())
    try:
        from bench import kernel_source_hash
        source_hash = kernel_source_hash()
    except Exception:
        source_hash = None
    doc = {"FETCH_SIZE_KB_per_kernel": out["FETCH_SIZE"], "WRITE_SIZE_KB_per_kernel": out["WRITE_SIZE"],
           "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "esteps_in_pass": esteps,
           "kernel_source_hash": source_hash,
           "correction": "FETCH_SIZE * 2 (gfx950: 128-byte requests of wide coalesced reads are tallied at 64 bytes, "
                         "MI355X_MICROARCH.md HBM section); WRITE_SIZE as is; KB = 1024 bytes; per E-step = all dispatches "
                         "of the E-step's kernels / E-steps in the pass",
           "hbm_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024)}
    print("== E-step traffic (per E-step):", json.dumps(doc))
    with open(os.path.join(root, "traffic.json"), "w") as fh:
        json.dump(doc, fh, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
    estep_traffic(sys.argv[1])
