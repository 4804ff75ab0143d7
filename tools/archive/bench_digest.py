#!/usr/bin/env python3
"""One-screen digest of a bench.py JSON line: python tools/bench_digest.py gpurun_out/bench_x.json"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
def line(tag, r, value, ms):
    rf = r["roofline"]
    print("%-5s %12.0f docs/s  %8.2f ms/step  documents %8.3f ms  statistics %7.3f ms  frac %.4f" %
          (tag, value, ms, rf["kernel_ms_documents"], rf["kernel_ms_sstats"], rf["frac"]))
line("cfg3", d, d["value"], d["ms_per_step"])
if "synth1m" in d:
    s = d["synth1m"]; line("cfg4", s, s["value"], s["ms_per_step"])
if "nips_k500" in d:
    n = d["nips_k500"]; line("cfg5", n, n["docs_per_s"], n["ms_per_step"]); print("      held-out rel delta %.2e" % n["heldout_rel_delta"])
if "ap_k10" in d:
    a = d["ap_k10"]; print("ap    %12.0f docs/s  estep %.3f ms  ll delta %.2e  iters equal %.3f" % (a["gpu_docs_per_s"], a["estep_ms"], a["max_rel_ll_delta_vs_reference"], a["iters_equal_fraction"]))
for k in ("ll_delta",):
    if k in d: print("cfg3 ll_delta", d[k])
