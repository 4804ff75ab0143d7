"""E-step of the associated-press K=10 fixture (cfg 2): launch plan, kernel time, wall time per E-step.
    python tools/ap_ab.py [name=value ...]"""
import os, sys, time, numpy as np
sys.path.insert(0, ".")
from pylda_amd import _capi
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
g = np.load(os.path.join(root, "tests/golden/ap_train_k10.npz"))
ptr, ids, cts = g["doc_ptr"].astype(np.int64), g["term_id"].astype(np.int32), g["term_ct"].astype(np.int32)
ctx = _capi.Context(10, g["eta"].shape[1])
for kv in sys.argv[1:]:
    name, value = kv.split("=")
    ctx.set_option(name, int(value))
ctx.set_option("doc_values", 0)
corpus = ctx.corpus(ptr, ids, cts)
ctx.set_alpha(g["alpha"]); ctx.set_eta(g["eta"])
for _ in range(3):
    ctx.estep(corpus)
ctx.synchronize()
reps = 50
t0 = time.perf_counter()
for _ in range(reps):
    ctx.estep(corpus); ctx.estep_results(corpus)
wall = (time.perf_counter() - t0) / reps * 1e3
ctx.set_profiling(True); ctx.kernel_time(); corpus.plan()
for _ in range(reps):
    ctx.estep(corpus); ctx.estep_results(corpus)
doc_ms, ss_ms, calls = ctx.kernel_time()
print("AP K=10 %s: E-step %.3f ms wall (unprofiled), document kernels %.3f ms, statistics %.3f ms; classes %s"
      % (sys.argv[1:], wall, doc_ms / calls, ss_ms / calls,
         [(c["kernel"], c["geometry"], c["documents"], round(c["kernel_ms"] / calls, 3)) for c in corpus.plan()]))
