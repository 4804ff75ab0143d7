"""ctypes wrapper of oracle/vb_oracle.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle_vb.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    src = os.path.join(HERE, "vb_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-Wall", "-shared", "-o", LIB, src, "-lm"])
    return LIB


def load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(LIB)
        for name in ("vb_oracle_digamma", "vb_oracle_trigamma", "vb_oracle_lgamma"):
            getattr(lib, name).restype = ctypes.c_double
            getattr(lib, name).argtypes = [ctypes.c_double]
        lib.vb_oracle_dirichlet_expectation.restype = ctypes.c_int
        lib.vb_oracle_dirichlet_expectation.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp]
        lib.vb_oracle_estep.restype = ctypes.c_int
        lib.vb_oracle_estep.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_int64, _i64p,
                                        _i32p, _i32p, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                        _dp, _dp, _dp, _i32p, _dp]
        _lib = lib
    return _lib


def digamma(x):
    lib = load()
    return np.array([lib.vb_oracle_digamma(float(v)) for v in np.ravel(x)]).reshape(np.shape(x))


def trigamma(x):
    lib = load()
    return np.array([lib.vb_oracle_trigamma(float(v)) for v in np.ravel(x)]).reshape(np.shape(x))


def lgamma(x):
    lib = load()
    return np.array([lib.vb_oracle_lgamma(float(v)) for v in np.ravel(x)]).reshape(np.shape(x))


def dirichlet_expectation(eta):
    lib = load()
    eta = np.ascontiguousarray(eta, dtype=np.float64)
    out = np.empty_like(eta)
    lib.vb_oracle_dirichlet_expectation(eta.shape[0], eta.shape[1], eta.ctypes.data_as(_dp),
                                        out.ctypes.data_as(_dp))
    return out


def e_step(alpha, eta, doc_ptr, term_id, term_ct, max_iter=50, tol=1e-6, heldout=False):
    lib = load()
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    eta = np.ascontiguousarray(eta, dtype=np.float64)
    doc_ptr = np.ascontiguousarray(doc_ptr, dtype=np.int64)
    term_id = np.ascontiguousarray(term_id, dtype=np.int32)
    term_ct = np.ascontiguousarray(term_ct, dtype=np.int32)
    K, V = eta.shape
    D = doc_ptr.size - 1
    gamma = np.zeros((D, K))
    doc_ll = np.zeros(D)
    words_ll = np.zeros(D)
    iters = np.zeros(D, dtype=np.int32)
    sstats = np.zeros((K, V))
    rc = lib.vb_oracle_estep(K, V, alpha.ctypes.data_as(_dp), eta.ctypes.data_as(_dp), D,
                             doc_ptr.ctypes.data_as(_i64p), term_id.ctypes.data_as(_i32p),
                             term_ct.ctypes.data_as(_i32p), int(max_iter), float(tol),
                             1 if heldout else 0, gamma.ctypes.data_as(_dp),
                             doc_ll.ctypes.data_as(_dp), words_ll.ctypes.data_as(_dp),
                             iters.ctypes.data_as(_i32p), sstats.ctypes.data_as(_dp))
    if rc != 0:
        raise MemoryError("vb_oracle_estep failed")
    return {"document_log_likelihood": float(doc_ll.sum()),
            "words_log_likelihood": float(words_ll.sum()), "sstats": sstats, "gamma": gamma,
            "doc_ll": doc_ll, "doc_words_ll": words_ll, "iters": iters}
