"""Live-topic census of the E-step: how many topics of a document still have gamma_k != alpha_k
(bitwise) after i inner iterations, inside the bench's timed window.

    python tools/live_probe.py synth1m|synth100k|nips|ap [docs] [outer]

Runs `outer` (default 3) learning() iterations from the seeded start, then E-steps of the NEXT outer iteration capped
at i = 1, 2, ... inner iterations (the state after i iterations of a document is the same whatever the cap), reads
gamma back and counts.  Prints per cap: mean live fraction, and the distribution of the live count; and per document
the first iteration at which the live count is <= 16 / 24 / 32 / 48 / 64.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    name = sys.argv[1]
    docs = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else None
    outer = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    import torch
    from pylda_amd.variational_bayes import VariationalBayes
    dev = torch.device("cuda", 0)
    wl = bench.build_workload(name, 0, 1, dev, docs)
    ptr, ids, cts, V, K = wl["ptr"], wl["ids"], wl["cts"], wl["V"], wl["K"]
    np.random.seed(0)
    eta0 = wl.get("eta")
    if eta0 is None:
        eta0 = np.random.gamma(100., 1. / 100., (K, V))
    vb = VariationalBayes(hyper_parameter_optimize_interval=1, device=0)
    vb._verbose = False
    vb._initialize_parsed(ptr, ids, cts, V, K, 1.0 / K, 1.0 / V, eta=eta0)
    if "alpha" in wl:
        vb._alpha_alpha = wl["alpha"].copy()
    ctx = vb._context()
    for _ in range(outer):
        vb.learning()
    vb._push_model()
    alpha = vb._alpha_alpha.copy()
    corpus = vb._train_corpus
    D = len(ptr) - 1
    nterms = np.diff(ptr)
    caps = [1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 15, 20, 30, 50]
    bounds = [8, 16, 24, 32, 48, 64]
    first_at = {b: np.full(D, 99, dtype=np.int32) for b in bounds}
    out = {"workload": name, "docs": D, "K": K, "outer_iterations_before": outer,
           "alpha_min": float(alpha.min()), "alpha_max": float(alpha.max()), "caps": []}
    for cap in caps:
        ctx.estep(corpus, cap, 1e-6, False)
        gamma = ctx.get_gamma(corpus)
        _, _, iters = ctx.get_doc_values(corpus, want_ll=False)
        live = (gamma != alpha[None, :]).sum(axis=1)
        for b in bounds:
            hit = (live <= b) & (first_at[b] == 99)
            first_at[b][hit] = cap
        q = np.percentile(live, [1, 10, 50, 90, 99, 100])
        rec = {"cap": cap, "mean_iters": float(iters.mean()), "stopped_before_cap": float((iters < cap).mean()),
               "live_mean": float(live.mean()), "live_frac": float(live.mean() / K),
               "live_p1_p10_p50_p90_p99_max": [float(x) for x in q],
               "docs_live_le": {str(b): float((live <= b).mean()) for b in bounds}}
        out["caps"].append(rec)
        print(json.dumps(rec), flush=True)
        del gamma
    # N x L of the documents at their hand-off (first cap with live <= 32): how large is the compact tile
    for b in bounds:
        vals, counts = np.unique(first_at[b], return_counts=True)
        out["first_iteration_live_le_%d" % b] = {str(int(v)): int(c) for v, c in zip(vals, counts)}
    out["terms_p50_p99_max"] = [float(x) for x in np.percentile(nterms, [50, 99, 100])]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
