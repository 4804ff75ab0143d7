"""E-step kernel time on the associated-press K=10 fixture (cfg 2): python tools/ap_ab.py [name=value ...]"""
import os, sys, numpy as np
sys.path.insert(0, ".")
from pylda_amd import _capi
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
g = np.load(os.path.join(root, "tests/golden/ap_train_k10.npz"), allow_pickle=True)
K, V = g["eta"].shape
ctx = _capi.Context(K, V)
for kv in sys.argv[1:]:
    name, value = kv.split("=")
    ctx.set_option(name, int(value))
ctx.set_option("doc_values", 0)
corpus = ctx.corpus(g["doc_ptr"], g["term_id"], g["term_ct"])
ctx.set_alpha(g["alpha"]); ctx.set_eta(g["eta"])
for _ in range(3):
    ctx.estep(corpus)
ctx.synchronize()
ctx.set_profiling(True); ctx.kernel_time(); corpus.plan()
import time
t0 = time.perf_counter()
for _ in range(20):
    ctx.estep(corpus)
ctx.synchronize()
wall = (time.perf_counter() - t0) / 20
doc_ms, ss_ms, calls = ctx.kernel_time()
_, _, iters = ctx.get_doc_values(corpus, want_ll=False)
n = np.diff(g["doc_ptr"])
print("AP K=%d: %d docs, N mean %.0f max %d, iterations mean %.1f max %d; E-step wall %.3f ms, doc kernels %.3f ms, sstats %.3f ms; classes %s"
      % (K, len(n), n.mean(), n.max(), iters.mean(), iters.max(), wall * 1e3, doc_ms / calls, ss_ms / calls,
         [(c["kernel"], c["geometry"], c["documents"], round(c["kernel_ms"] / calls, 3)) for c in corpus.plan()]))
