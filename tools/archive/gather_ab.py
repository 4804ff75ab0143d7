"""Statistics pass alone on the cfg-3 / cfg-4 corpus for several settings: python tools/gather_ab.py cfg3 name=v,name=v ...
(each argument one configuration: a new context + corpus, 2 + 5 E-steps, statistics-pass time)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pylda_amd import _capi
from pylda_amd.corpus import synthetic_lda_shard
cfg = sys.argv[1]
D, V, K, seed = (100000, 50000, 128, 1234) if cfg == "cfg3" else (1000000, 100000, 256, 5678)
ptr, ids, cts = synthetic_lda_shard(D, V, 0, D, 128, 200, seed, chunk=25000, device="cuda", workers=8)
np.random.seed(0)
eta = np.random.gamma(100., 0.01, (K, V))
for spec in sys.argv[2:]:
    ctx = _capi.Context(K, V)
    ctx.set_option("doc_values", 0)
    for kv in spec.split(","):
        if kv:
            name, value = kv.split("=")
            ctx.set_option(name, int(value))
    corpus = ctx.corpus(ptr, ids, cts)
    ctx.set_alpha(np.full(K, 1.0 / K)); ctx.set_eta(eta)
    for _ in range(2):
        ctx.estep(corpus)
    ctx.synchronize()
    ctx.set_profiling(True); ctx.kernel_time(); corpus.plan()
    for _ in range(4):
        ctx.estep(corpus)
    ctx.synchronize()
    doc_ms, ss_ms, calls = ctx.kernel_time()
    print("%s [%s]: statistics pass %.3f ms (documents %.2f ms); blocks %d, sweep passes %d, partial rows %d"
          % (cfg, spec, ss_ms / calls, doc_ms / calls, corpus.layout("gather_blocks"), corpus.layout("gather_sweep_passes"),
             corpus.layout("gather_partial_rows")), flush=True)
    corpus.close(); ctx.close()
