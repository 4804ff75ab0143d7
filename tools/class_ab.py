"""A/B of kernel classes on slices of the cfg-3 / cfg-4 corpus: python tools/class_ab.py cfg3 nmin nmax name=value ..."""
import sys, time, json, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pylda_amd import _capi
from pylda_amd.corpus import synthetic_lda_shard
cfg, nmin, nmax = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
opts = [kv.split("=") for kv in sys.argv[4:]]
D, V, K, seed = (100000, 50000, 128, 1234) if cfg == "cfg3" else (200000, 100000, 256, 5678)
ptr, ids, cts = synthetic_lda_shard(D if cfg == "cfg3" else 1000000, V, 0, D, 128, 200, seed, chunk=25000, device="cuda", workers=8)
n = np.diff(ptr)
sel = np.nonzero((n >= nmin) & (n <= nmax))[0]
newptr = np.concatenate([[0], np.cumsum(n[sel])]).astype(np.int64)
idx = np.concatenate([np.arange(ptr[d], ptr[d + 1]) for d in sel]) if len(sel) < 300000 else None
ids2, cts2 = ids[idx], cts[idx]
import os
if os.environ.get("CLASS_AB_VMOD"):      # probe: fold the vocabulary so that the table rows stay in L2 (timing only)
    ids2 = (ids2 % int(os.environ["CLASS_AB_VMOD"])).astype(ids2.dtype)
np.random.seed(0)
eta = np.random.gamma(100., 0.01, (K, V))
ctx = _capi.Context(K, V)
for name, value in opts:
    ctx.set_option(name, int(value))
ctx.set_option("doc_values", 0)
corpus = ctx.corpus(newptr, ids2, cts2)
ctx.set_alpha(np.full(K, 1.0 / K)); ctx.set_eta(eta)
for _ in range(2):
    ctx.estep(corpus)
ctx.synchronize()
ctx.set_profiling(True); ctx.kernel_time(); corpus.plan()
for _ in range(int(os.environ.get("CLASS_AB_STEPS", "5"))):      # (long runs: sustained clocks)
    ctx.estep(corpus)
ctx.synchronize()
doc_ms, ss_ms, calls = ctx.kernel_time()
_, _, iters = ctx.get_doc_values(corpus, want_ll=False)
print("%s N in [%d,%d]: %d docs, nnz %d, mean iters %.1f, doc kernels %.3f ms (%.1f us/doc-iteration x CU... %.2f ns/doc), sstats %.3f ms; classes %s; opts %s"
      % (cfg, nmin, nmax, len(sel), newptr[-1], iters.mean(), doc_ms / calls, 0.0, doc_ms / calls * 1e6 / len(sel), ss_ms / calls,
         [(c["kernel"], c["geometry"], c["documents"], round(c["kernel_ms"] / calls, 3)) for c in corpus.plan()], opts))
