"""E-step kernel time on the nips.88-05 K=500 corpus of tests/golden (cfg 5 shape): python tools/nips_ab.py [K] [name=value ...]"""
import os, sys, numpy as np
sys.path.insert(0, ".")
from pylda_amd import _capi
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
g = np.load(os.path.join(root, "tests/golden/nips_trace_k500.npz"), allow_pickle=True)
args = sys.argv[1:]
K = int(args.pop(0)) if args and args[0].isdigit() else int(g["K"])
V = len(g["words"])
ptr, ids, cts = g["doc_ptr"].astype(np.int64), g["term_id"].astype(np.int32), g["term_ct"].astype(np.int32)
np.random.seed(0)
eta = np.random.gamma(100., 0.01, (K, V))
ctx = _capi.Context(K, V)
for kv in args:
    name, value = kv.split("=")
    ctx.set_option(name, int(value))
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
print("nips K=%d: %d docs, nnz %d, doc kernels %.3f ms, sstats %.3f ms; classes %s"
      % (K, len(ptr) - 1, ptr[-1], doc_ms / calls, ss_ms / calls,
         [(c["kernel"], c["geometry"], c["documents"], round(c["kernel_ms"] / calls, 3)) for c in corpus.plan()]))
