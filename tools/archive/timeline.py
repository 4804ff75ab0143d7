#!/usr/bin/env python3
"""Dispatch timeline of ONE E-step from a rocprofv3 --kernel-trace run (on the GPU box):

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python tools/class_ab.py cfg3 1 1000
    python tools/timeline.py /tmp/tl [estep-index-from-the-end, default 1] > profiles/r04_cfg3_timeline.txt

Prints, for the E-step's document kernels and statistics pass, begin / end offsets (ms from the first dispatch of that
E-step), queue, grid, and three totals: sum of the durations, the union (wall) and the chip-idle gaps inside it."""
import csv, glob, os, sys


def main():
    root = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    path = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # an E-step starts at its eta_rowsum_psi_kernel (table preparation) dispatch
    starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("pylda::eta_rowsum_psi") or "eta_rowsum_psi_kernel" in r["Kernel_Name"]]
    if len(starts) < back + 1:
        a, b = starts[-1], len(rows)
    else:
        a, b = starts[-back - 1], starts[-back]
    sel = rows[a:b]
    t0 = int(sel[0]["Start_Timestamp"])
    print("# %s: E-step %d from the end, %d dispatches; offsets in ms from its first dispatch" % (os.path.basename(path), back, len(sel)))
    print("# %-58s %5s %9s %9s %9s  %s" % ("kernel", "queue", "begin", "end", "duration", "grid x workgroup"))
    spans = []
    for r in sel:
        name = r["Kernel_Name"].replace("pylda::", "").replace("void ", "")
        if not (name.startswith(("estep_", "sstats_", "doc_terms"))):
            continue
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
        spans.append((s, e, name))
        grid = "%s x %s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
        print("  %-58s %5s %9.3f %9.3f %9.3f  %s" % (name[:58], r.get("Queue_Id", "?"), s, e, e - s, grid))
    docs = [x for x in spans if x[2].startswith("estep_") and "logspace" not in x[2]]      # (the safety net's kernel runs behind the statistics pass)
    for label, group in (("document kernels", docs), ("document kernels + doc_terms + statistics", spans)):
        if not group:
            continue
        group = sorted(group)
        lo, hi = group[0][0], max(e for _, e, _ in group)
        covered, cur_s, cur_e = 0.0, group[0][0], group[0][1]
        for s, e, _ in group[1:]:
            if s > cur_e:
                covered += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        covered += cur_e - cur_s
        print("# %s: sum of durations %.3f ms, wall %.3f ms (%.3f .. %.3f), no kernel running for %.3f ms of it"
              % (label, sum(e - s for s, e, _ in group), hi - lo, lo, hi, (hi - lo) - covered))


if __name__ == "__main__":
    main()
