import sys, time, numpy as np
sys.path.insert(0, ".")
from pylda_amd import _capi
from pylda_amd.corpus import synthetic_lda_shard
ptr, ids, cts = synthetic_lda_shard(1000000, 100000, 0, 1000000, 128, 200, 5678, chunk=25000, device="cuda", workers=8)
np.random.seed(0)
K, V = 256, 100000
eta = np.random.gamma(100., 0.01, (K, V))
ctx = _capi.Context(K, V)
ctx.set_option("doc_values", 0)
t0 = time.perf_counter()
corpus = ctx.corpus(ptr, ids, cts)
t1 = time.perf_counter()
ctx.set_alpha(np.full(K, 1.0 / K)); ctx.set_eta(eta)
ctx.synchronize()
t2 = time.perf_counter()
ctx.estep(corpus); ctx.synchronize()
t3 = time.perf_counter()
ctx.estep(corpus); ctx.synchronize()
t4 = time.perf_counter()
print("corpus() %.3f s, set model %.3f s, first E-step %.3f s, second %.3f s" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3))
