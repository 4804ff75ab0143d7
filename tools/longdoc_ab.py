"""Long documents at a given K (wide / hybrid / streaming kernels): python tools/longdoc_ab.py K V D mean_len"""
import sys, numpy as np
sys.path.insert(0, ".")
from pylda_amd import _capi
K, V, D, mean_len = (int(x) for x in sys.argv[1:5])
rng = np.random.default_rng(2)
ptr, ids, cts = [0], [], []
for _ in range(D):
    n = max(8, int(rng.normal(mean_len, mean_len * 0.15)))
    u = np.sort(rng.choice(V, size=n, replace=False)).astype(np.int32)
    ids.append(u); cts.append(rng.integers(1, 4, size=n).astype(np.int32)); ptr.append(ptr[-1] + n)
ptr = np.array(ptr, np.int64); ids = np.concatenate(ids); cts = np.concatenate(cts)
eta = rng.gamma(100., 0.01, (K, V))
ctx = _capi.Context(K, V)
ctx.set_option("doc_values", 0)
corpus = ctx.corpus(ptr, ids, cts)
ctx.set_alpha(np.full(K, 1.0 / K)); ctx.set_eta(eta)
for _ in range(2):
    ctx.estep(corpus)
ctx.synchronize()
ctx.set_profiling(True); ctx.kernel_time(); corpus.plan()
for _ in range(5):
    ctx.estep(corpus)
ctx.synchronize()
doc_ms, ss_ms, calls = ctx.kernel_time()
print("K=%d V=%d D=%d mean N=%d: doc kernels %.3f ms (%.1f ns/doc), ll %.4f; classes %s"
      % (K, V, D, mean_len, doc_ms / calls, doc_ms / calls * 1e6 / D, ctx.estep_results(corpus)[0],
         [(c["kernel"], c["geometry"], c["documents"], round(c["kernel_ms"] / calls, 3)) for c in corpus.plan()]))
