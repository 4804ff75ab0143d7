"""Where one learning() iteration goes: device time of the E-step and M-step spans (stream marks), the host-side
Newton update, the wall time.  python tools/step_breakdown.py [nips|ap|cfg3]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pylda_amd.variational_bayes import VariationalBayes
which = sys.argv[1] if len(sys.argv) > 1 else "nips"
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
np.random.seed(0)
m = VariationalBayes()
m._verbose = False
if which == "nips":
    g = np.load(os.path.join(root, "tests/golden/nips_trace_k500.npz"))
    m._initialize_parsed(g["doc_ptr"].astype(np.int64), g["term_id"].astype(np.int32), g["term_ct"].astype(np.int32), len(g["words"]), int(g["K"]), 1.0 / int(g["K"]), 1.0 / len(g["words"]))
elif which == "ap":
    g = np.load(os.path.join(root, "tests/golden/ap_train_k10.npz"))
    m._initialize_parsed(g["doc_ptr"].astype(np.int64), g["term_id"].astype(np.int32), g["term_ct"].astype(np.int32), g["eta"].shape[1], 10, 0.1, 1.0 / g["eta"].shape[1])
else:
    from pylda_amd.corpus import synthetic_lda_shard
    ptr, ids, cts = synthetic_lda_shard(100000, 50000, 0, 100000, 128, 200, 1234, chunk=25000, device="cuda", workers=8)
    m._initialize_parsed(ptr, ids, cts, 50000, 128, 1.0 / 128, 1.0 / 50000)
ctx = m._context()
for _ in range(3):
    m.learning()
orig = m.optimize_hyperparameters
newton = []
def timed(*a, **k):
    t0 = time.perf_counter(); orig(*a, **k); newton.append(time.perf_counter() - t0)
m.optimize_hyperparameters = timed
m._verbose = True
import io, contextlib
walls, e_ms, m_ms = [], [], []
for _ in range(10):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        m.learning()
    walls.append(time.perf_counter() - t0)
    e_ms.append(ctx.elapsed_ms(0, 1)); m_ms.append(ctx.elapsed_ms(1, 2))
print("%s: wall %.3f ms/iteration; device E-step span %.3f ms, M-step span %.3f ms; host Newton update %.3f ms"
      % (which, np.median(walls) * 1e3, np.median(e_ms), np.median(m_ms), np.median(newton) * 1e3))
