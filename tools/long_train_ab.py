"""Long training run with and without the live-topic kernel: python tools/long_train_ab.py cfg3|cfg4 [docs] [iterations] [a:name=value | b:name=value ...]

`iterations` learning() iterations (device M-step, alpha update every iteration) from the seeded start, twice - option
compact = 1 (run a) and 0 (run b; further options per run: a:name=value, b:name=value - e.g. `a:compact=0 b:gather_sweep=0`
compares two runs of the dense kernels that differ in the summation order of the statistics only): the joint log-likelihood traces must agree (the bench's window is iterations 4-8 only; alpha and the live
sets keep moving long after it), and no document may be flagged.  Prints the time per iteration of both runs."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    cfg = sys.argv[1]
    docs = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else None
    iterations = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    import torch
    from pylda_amd.variational_bayes import VariationalBayes
    wl = bench.build_workload({"cfg3": "synth100k", "cfg4": "synth1m", "nips": "nips"}[cfg], 0, 1, torch.device("cuda", 0), docs)
    ptr, ids, cts, V, K = wl["ptr"], wl["ids"], wl["cts"], wl["V"], wl["K"]
    np.random.seed(0)
    eta0 = wl.get("eta")
    if eta0 is None:
        eta0 = np.random.gamma(100., 1. / 100., (K, V))
    out = {"cfg": cfg, "docs": len(ptr) - 1, "K": K, "iterations": iterations}
    traces = {}
    for mode in (1, 0):
        vb = VariationalBayes(hyper_parameter_optimize_interval=1, device=0)
        vb._verbose = False
        vb._initialize_parsed(ptr, ids, cts, V, K, 1.0 / K, 1.0 / V, eta=eta0.copy())
        ctx = vb._context()
        ctx.set_option("compact", mode)
        for opt in sys.argv[4:]:
            run, kv = opt.split(":")
            if run == ("a" if mode == 1 else "b"):
                k, v = kv.split("=")
                ctx.set_option(k, int(v))
        ctx.set_profiling(True)
        ll, ms, flagged, handed = [], [], [], []
        for _ in range(iterations):
            ctx.work_counters()
            t0 = time.perf_counter()
            ll.append(vb.learning())
            ctx.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
            flagged.append(int(ctx.estep_results(vb._train_corpus)[2]))
            ctx.work_counters()
            handed.append(int(ctx.executed_work()[1]))
        traces[mode] = dict(ll=np.array(ll), ms=ms, flagged=flagged, handed=handed, alpha=vb._alpha_alpha.copy())
        vb._train_corpus.close()
        vb._ctx.close()
    a, b = traces[1], traces[0]
    out["ll_trace_max_rel"] = float(np.max(np.abs(a["ll"] - b["ll"]) / np.abs(b["ll"])))
    out["ll_rel_per_iteration"] = ["%.1e" % x for x in np.abs(a["ll"] - b["ll"]) / np.abs(b["ll"])]
    out["alpha_max_rel"] = float(np.max(np.abs(a["alpha"] - b["alpha"]) / b["alpha"]))
    out["alpha_min_max"] = [float(b["alpha"].min()), float(b["alpha"].max())]
    out["flagged_documents"] = [int(sum(a["flagged"])), int(sum(b["flagged"]))]
    out["handed_over_per_iteration"] = a["handed"]
    out["ms_per_iteration_live"] = [round(x, 2) for x in a["ms"]]
    out["ms_per_iteration_dense"] = [round(x, 2) for x in b["ms"]]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
