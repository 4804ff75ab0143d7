"""Sweep a runtime option of the document kernels on the full cfg-3 / cfg-4 corpus (one process, one corpus):

    python tools/tune_sweep.py cfg3 quad_tune 0 20 40 60 0x10000 0x20000 0x10028

prints the document-kernel time per E-step (HIP events on the launch streams) for every value."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pylda_amd import _capi
from pylda_amd.corpus import synthetic_lda_shard
cfg, name, values = sys.argv[1], sys.argv[2], [int(v, 0) for v in sys.argv[3:]]
D, V, K, seed = (100000, 50000, 128, 1234) if cfg == "cfg3" else (200000, 100000, 256, 5678)
ptr, ids, cts = synthetic_lda_shard(D if cfg == "cfg3" else 1000000, V, 0, D, 128, 200, seed, chunk=25000, device="cuda", workers=8)
np.random.seed(0)
eta = np.random.gamma(100., 0.01, (K, V))
ctx = _capi.Context(K, V)
ctx.set_option("doc_values", 0)
corpus = ctx.corpus(ptr, ids, cts)
ctx.set_alpha(np.full(K, 1.0 / K)); ctx.set_eta(eta)
for _ in range(3):
    ctx.estep(corpus)
ctx.synchronize()
steps = int(os.environ.get("SWEEP_STEPS", "6"))
for rep in range(2):
    for v in values:
        ctx.set_option(name, v)
        ctx.estep(corpus); ctx.synchronize()
        ctx.set_profiling(True); ctx.kernel_time(); corpus.plan()
        for _ in range(steps):
            ctx.estep(corpus)
        ctx.synchronize()
        doc_ms, ss_ms, calls = ctx.kernel_time()
        ctx.set_profiling(False)
        print("%s %s=%#x: doc kernels %.3f ms, sstats %.3f ms (%d E-steps)" % (cfg, name, v, doc_ms / calls, ss_ms / calls, calls), flush=True)
