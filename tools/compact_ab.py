"""Live-topic kernel against the dense kernels on the same E-step: python tools/compact_ab.py cfg3|cfg4 [docs] [outer]

Runs `outer` learning() iterations (default 3) on the first `docs` documents of the corpus, then ONE E-step of the next
outer iteration twice - option compact = 0 and 1 - with per-document values, and compares: iteration counts (must be
equal), gamma, per-document log-likelihood, sufficient statistics.  Then times both on the training fast path.
"""
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
    outer = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    import torch
    from pylda_amd.variational_bayes import VariationalBayes
    name = {"cfg3": "synth100k", "cfg4": "synth1m"}[cfg]
    wl = bench.build_workload(name, 0, 1, torch.device("cuda", 0), docs)
    ptr, ids, cts, V, K = wl["ptr"], wl["ids"], wl["cts"], wl["V"], wl["K"]
    np.random.seed(0)
    eta0 = np.random.gamma(100., 1. / 100., (K, V))
    vb = VariationalBayes(hyper_parameter_optimize_interval=1, device=0)
    vb._verbose = False
    vb._initialize_parsed(ptr, ids, cts, V, K, 1.0 / K, 1.0 / V, eta=eta0)
    ctx = vb._context()
    for opt in sys.argv[4:]:
        k, v = opt.split("=")
        ctx.set_option(k, int(v))
    for _ in range(outer):
        vb.learning()
    vb._push_model()
    corpus = vb._train_corpus
    out = {"cfg": cfg, "docs": len(ptr) - 1, "K": K}
    res = {}
    for mode in (0, 1):
        ctx.set_option("compact", mode)
        ctx.set_option("doc_values", 1)
        ctx.estep(corpus, 50, 1e-6, False)
        ll, _, flagged = ctx.estep_results(corpus)
        doc_ll, _, iters = ctx.get_doc_values(corpus)
        res[mode] = dict(ll=ll, flagged=flagged, doc_ll=doc_ll, iters=iters, gamma=ctx.get_gamma(corpus), sstats=ctx.get_sstats())
    a, b = res[0], res[1]
    flips = int((a["iters"] != b["iters"]).sum())
    gd = np.abs(a["gamma"] - b["gamma"]) / np.abs(a["gamma"])
    out.update({"iteration_flips": flips, "flagged": [a["flagged"], b["flagged"]],
                "gamma_max_rel": float(gd.max()), "gamma_entries_differing": int((a["gamma"] != b["gamma"]).sum()),
                "gamma_entries": int(a["gamma"].size),
                "doc_ll_max_rel": float(np.max(np.abs(a["doc_ll"] - b["doc_ll"]) / np.abs(a["doc_ll"]))),
                "ll_rel": abs(a["ll"] - b["ll"]) / abs(a["ll"]),
                "sstats_max_abs": float(np.abs(a["sstats"] - b["sstats"]).max()),
                "sstats_max": float(np.abs(a["sstats"]).max()), "mean_iters": float(a["iters"].mean())})
    del res
    # timing on the training fast path
    ctx.set_option("doc_values", 0)
    for mode in (0, 1, 0, 1):
        ctx.set_option("compact", mode)
        ctx.estep(corpus, 50, 1e-6, False)
        ctx.synchronize()
        ctx.set_profiling(True)
        ctx.kernel_time()
        ctx.work_counters()
        ctx.executed_work()
        corpus.plan()
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.estep(corpus, 50, 1e-6, False)
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / 3 * 1e3
        doc_ms, ss_ms, calls = ctx.kernel_time()
        its, terms = ctx.work_counters()
        entries, handed = ctx.executed_work()
        ctx.set_profiling(False)
        out.setdefault("timing", []).append({
            "compact": mode, "estep_wall_ms": wall, "doc_kernels_ms": doc_ms / calls, "sstats_ms": ss_ms / calls,
            "live_fraction": entries / (K * terms) if terms else None, "handed_over": handed / calls,
            "classes": [(c["kernel"], c["geometry"], c["documents"], round(c["kernel_ms"] / calls, 3)) for c in corpus.plan()]})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
