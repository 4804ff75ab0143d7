"""One process of bench.py's all-cores CPU baseline leg (TEST / MEASUREMENT INFRASTRUCTURE ONLY).

    python oracle/cpu_pool_worker.py <problem.npz> <first_doc> <last_doc> <budget_seconds>

Runs the numpy restatement of variational_bayes.py:132-216 (oracle/vb_numpy.py) single-threaded on documents
[first_doc, last_doc) of the problem until the budget is used up and prints "<documents> <seconds>"."""
import os
import sys
import time

for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[_k] = "1"
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vb_numpy


def main():
    path, first, last, budget = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    z = np.load(path)
    alpha, ptr, ids, cts = z["alpha"], z["ptr"], z["ids"], z["cts"]
    E_log_eta = vb_numpy.compute_dirichlet_expectation(z["eta"])
    t0 = time.perf_counter()
    n = 0
    for d in range(first, last):
        lo, hi = int(ptr[d]), int(ptr[d + 1])
        vb_numpy.e_step_document(alpha, E_log_eta, ids[lo:hi].astype(np.int64), cts[lo:hi])
        n += 1
        if time.perf_counter() - t0 > budget and n >= 5:
            break
    print("%d %.6f" % (n, time.perf_counter() - t0), flush=True)


if __name__ == "__main__":
    main()
