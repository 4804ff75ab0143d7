"""Long documents outside the quad kernel's reach: E-step time of (a) the nips.88-05 training split at K = 100 / 200
(a third of its documents has more than 256 distinct terms) and (b) synthetic long documents at K = 64 / 128 / 256,
with this tree's library.  Run from the tree to measure:  python tools/longdoc_ab.py  (or from a copy under .ab/)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from pylda_amd import _capi
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(root, "tests", "golden", "nips_trace_k500.npz"))


def time_estep(K, V, ptr, ids, cts, label, options=()):
    rng = np.random.default_rng(K)
    eta = rng.gamma(100.0, 0.01, (K, V))
    ctx = _capi.Context(K, V)
    for name, value in options:
        ctx.set_option(name, value)
    ctx.set_option("doc_values", 0)
    corpus = ctx.corpus(ptr, ids, cts)
    ctx.set_alpha(np.full(K, 1.0 / K))
    ctx.set_eta(eta)
    for _ in range(2):
        ctx.estep(corpus)
    ctx.synchronize()
    ctx.set_profiling(True); ctx.kernel_time(); corpus.plan()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.estep(corpus)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    doc_ms, ss_ms, calls = ctx.kernel_time()
    ll = ctx.estep_results(corpus)[0]
    print("%-34s K=%4d D=%5d: E-step %.3f ms (document kernels %.3f), ll %.10e, classes %s"
          % (label, K, len(ptr) - 1, wall, doc_ms / calls, ll, [(c["kernel"], c["geometry"], c["documents"], round(c["kernel_ms"] / calls, 3)) for c in corpus.plan()]))
    corpus.close(); ctx.close()


ptr, ids, cts = g["doc_ptr"], g["term_id"], g["term_ct"]
V = len(g["words"])
for K in (100, 200, 64):
    time_estep(K, V, ptr, ids, cts, "nips.88-05 train")
rng = np.random.default_rng(1)
for K, n_lo, n_hi in ((128, 300, 900), (256, 300, 900), (64, 300, 900), (32, 400, 900)):
    lens = rng.integers(n_lo, n_hi, 1500)
    V2 = 20000
    p2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    i2 = np.concatenate([np.sort(rng.choice(V2, n, replace=False)) for n in lens]).astype(np.int32)
    c2 = rng.integers(1, 4, i2.size).astype(np.int32)
    time_estep(K, V2, p2, i2, c2, "synthetic %d-%d terms" % (n_lo, n_hi))
