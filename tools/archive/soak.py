"""Soak run (development tool): 300 outer iterations at cfg 3 - finite, non-decreasing joint likelihood, no device-memory drift."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import torch
from pylda_amd.variational_bayes import VariationalBayes
from pylda_amd.corpus import synthetic_lda_shard
D, V, K = 100000, 50000, 128
ptr, ids, cts = synthetic_lda_shard(D, V, 0, D, K, 200, 1234, chunk=25000, device="cuda", workers=8)
np.random.seed(0)
vb = VariationalBayes(); vb._verbose = False
vb._initialize_parsed(ptr, ids, cts, V, K, 1.0 / K, 1.0 / V)
free0 = None
lls = []
t0 = time.time()
for it in range(300):
    lls.append(vb.learning())
    if it == 5:
        free0 = torch.cuda.mem_get_info()[0]
free1 = torch.cuda.mem_get_info()[0]
lls = np.array(lls)
print("300 outer iterations in %.1f s; joint LL first %.6e last %.6e; finite %s; non-decreasing after iteration 3: %s (min step %.3e);"
      " device memory drift %d bytes; alpha range [%.4g, %.4g]"
      % (time.time() - t0, lls[0], lls[-1], bool(np.all(np.isfinite(lls))), bool(np.all(np.diff(lls[3:]) > -1e-6 * abs(lls[-1]))),
         np.diff(lls[3:]).min(), free0 - free1, vb._alpha_alpha.min(), vb._alpha_alpha.max()))
