import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        # a fixture that failed to ship must not turn a parity test into a silent skip
        pytest.fail("golden fixture %s is missing from tests/golden/" % name, pytrace=False)
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def tiny():
    return load_golden("tiny_k2.npz")


@pytest.fixture(scope="session")
def ap_train():
    g = load_golden("ap_train_k10.npz")
    g["term_id"] = g["term_id"].astype(np.int32)
    g["term_ct"] = g["term_ct"].astype(np.int32)
    return g


@pytest.fixture(scope="session")
def ap_test():
    g = load_golden("ap_test_k10.npz")
    g["term_id"] = g["term_id"].astype(np.int32)
    g["term_ct"] = g["term_ct"].astype(np.int32)
    return g


def csr_slice(doc_ptr, term_id, term_ct, docs):
    """CSR restricted to the listed documents (in that order)."""
    ptr = [0]
    ids, cts = [], []
    for d in docs:
        lo, hi = int(doc_ptr[d]), int(doc_ptr[d + 1])
        ids.append(term_id[lo:hi])
        cts.append(term_ct[lo:hi])
        ptr.append(ptr[-1] + hi - lo)
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    return np.array(ptr, dtype=np.int64), cat(ids, np.int32), cat(cts, np.int32)


def rel_err(a, b, floor=1e-9):
    """Largest |a-b| / max(|b|, floor): relative error with an absolute floor, so that
    quantities that are analytically zero (K=1 log-likelihoods) compare absolutely."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)) if a.size else 0.0
