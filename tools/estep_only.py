"""E-steps of the bench's timed window alone, for a profiler: python tools/estep_only.py cfg3|cfg4 [docs] [esteps] [name=value ...] [max_iter=N] [outer=N]
(`outer` learning() iterations - default 3 - from the seeded start, then `esteps` training E-steps of the next outer iteration, fast path)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

cfg = sys.argv[1]
docs = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else None
esteps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
import torch
from pylda_amd.variational_bayes import VariationalBayes
wl = bench.build_workload({"cfg3": "synth100k", "cfg4": "synth1m", "nips": "nips"}[cfg], 0, 1, torch.device("cuda", 0), docs)
ptr, ids, cts, V, K = wl["ptr"], wl["ids"], wl["cts"], wl["V"], wl["K"]
np.random.seed(0)
eta0 = wl.get("eta")
if eta0 is None:
    eta0 = np.random.gamma(100., 1. / 100., (K, V))
vb = VariationalBayes(hyper_parameter_optimize_interval=1, device=0)
vb._verbose = False
vb._initialize_parsed(ptr, ids, cts, V, K, 1.0 / K, 1.0 / V, eta=eta0)
ctx = vb._context()
max_iter, outer = 50, 3
for opt in sys.argv[4:]:
    k, v = opt.split("=")
    if k == "max_iter":         # (the timed E-steps only: the model is the one the full learning() iterations leave)
        max_iter = int(v)
    elif k == "outer":          # learning() iterations before the timed E-steps (default 3: the bench's window starts there)
        outer = int(v)
    else:
        ctx.set_option(k, int(v))
for _ in range(outer):
    vb.learning()
vb._push_model()
corpus = vb._train_corpus
ctx.set_profiling(True)
ctx.kernel_time()
for _ in range(esteps):
    ctx.estep(corpus, max_iter, 1e-6, False)
ctx.synchronize()
doc_ms, ss_ms, calls = ctx.kernel_time()
print("doc kernels %.3f ms, statistics %.3f ms per E-step (%d E-steps, inner iteration cap %d)" % (doc_ms / calls, ss_ms / calls, calls, max_iter))
