"""ctypes binding of libpylda_hip.so (the C ABI in include/pylda_hip.h).

This is the only place Python touches the native library.  There is no CPU
fallback anywhere in pylda_amd: if the shared library is missing, or no HIP
device is visible, the failure is raised here, loudly.
"""
import collections
import ctypes
import os
import threading

import numpy as np

ABI_VERSION = 3          # == PYLDA_ABI_VERSION of include/pylda_hip.h (checked at load time)
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpylda_hip.so")
_lib = None

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int32_p = ctypes.POINTER(ctypes.c_int32)
_c_int64_p = ctypes.POINTER(ctypes.c_int64)
_vp = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/pylda_hip.h declares
SIGNATURES = {
    "pylda_version": (ctypes.c_char_p, []),
    "pylda_abi_version": (ctypes.c_int, []),
    "pylda_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "pylda_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    "pylda_destroy": (None, [_vp]),
    "pylda_last_error": (ctypes.c_char_p, [_vp]),
    "pylda_set_stream": (ctypes.c_int, [_vp, _vp]),
    "pylda_use_own_stream": (ctypes.c_int, [_vp]),
    "pylda_synchronize": (ctypes.c_int, [_vp]),
    "pylda_corpus_create": (ctypes.c_int, [_vp, ctypes.c_int64, _c_int64_p, _c_int32_p, _c_int32_p,
                                           ctypes.POINTER(_vp)]),
    "pylda_corpus_destroy": (None, [_vp]),
    "pylda_corpus_info": (ctypes.c_int, [_vp, _c_int64_p, _c_int64_p, _c_int64_p, _c_int32_p]),
    "pylda_set_eta": (ctypes.c_int, [_vp, _c_double_p]),
    "pylda_get_eta": (ctypes.c_int, [_vp, _c_double_p]),
    "pylda_set_alpha": (ctypes.c_int, [_vp, _c_double_p]),
    "pylda_estep": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_double, ctypes.c_int]),
    "pylda_estep_results": (ctypes.c_int, [_vp, _vp, _c_double_p, _c_double_p, _c_int64_p]),
    "pylda_get_sstats": (ctypes.c_int, [_vp, _c_double_p]),
    "pylda_set_sstats": (ctypes.c_int, [_vp, _c_double_p]),
    "pylda_get_gamma": (ctypes.c_int, [_vp, _vp, _c_double_p]),
    "pylda_get_doc_values": (ctypes.c_int, [_vp, _vp, _c_double_p, _c_double_p, _c_int32_p]),
    "pylda_estep_host": (ctypes.c_int, [_vp, _vp, _c_double_p, _c_double_p, ctypes.c_int,
                                        ctypes.c_double, ctypes.c_int, _c_double_p, _c_double_p,
                                        _c_double_p, _c_double_p, _c_int32_p, _c_double_p]),
    "pylda_table_stride": (ctypes.c_int, [_vp]),
    "pylda_sstats_device": (_vp, [_vp]),
    "pylda_eta_device": (_vp, [_vp]),
    "pylda_gamma_device": (_vp, [_vp]),
    "pylda_mark_device_state": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "pylda_comm_unique_id": (ctypes.c_int, [_vp]),
    "pylda_comm_init": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int]),
    "pylda_comm_destroy": (ctypes.c_int, [_vp]),
    "pylda_allreduce_sstats": (ctypes.c_int, [_vp]),
    "pylda_allreduce_doubles": (ctypes.c_int, [_vp, _c_double_p, ctypes.c_int64]),
    "pylda_mstep": (ctypes.c_int, [_vp, _vp, _c_double_p, _c_double_p, _c_double_p]),
    "pylda_mstep_enqueue": (ctypes.c_int, [_vp, _vp, _c_double_p, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                           ctypes.c_double]),
    "pylda_outer_device": (_vp, [_vp, _c_int64_p]),
    "pylda_allreduce_outer": (ctypes.c_int, [_vp]),
    "pylda_outer_fetch": (ctypes.c_int, [_vp, _c_double_p, _c_double_p, _c_int64_p, _c_double_p, _c_double_p,
                                         _c_double_p]),
    "pylda_host_alloc": (ctypes.c_int, [ctypes.c_int64, ctypes.POINTER(_vp)]),
    "pylda_host_free": (ctypes.c_int, [_vp]),
    "pylda_model_checkpoint": (ctypes.c_int, [_vp, ctypes.c_int]),
    "pylda_mark_time": (ctypes.c_int, [_vp, ctypes.c_int]),
    "pylda_elapsed_ms": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _c_double_p]),
    "pylda_work_counters": (ctypes.c_int, [_vp, _c_double_p, _c_double_p]),
    "pylda_executed_work": (ctypes.c_int, [_vp, _c_double_p, _c_double_p]),
    "pylda_clock_counters": (ctypes.c_int, [_vp, _c_double_p, _c_double_p, _c_double_p]),
    "pylda_set_profiling": (ctypes.c_int, [_vp, ctypes.c_int]),
    "pylda_kernel_time": (ctypes.c_int, [_vp, _c_double_p, _c_double_p, _c_int64_p]),
    "pylda_corpus_layout": (ctypes.c_int64, [_vp, ctypes.c_char_p]),
    "pylda_corpus_plan": (ctypes.c_int, [_vp, ctypes.c_int32, _c_int32_p, _c_int32_p, _c_int64_p, _c_int64_p,
                                         _c_double_p]),
    "pylda_set_option": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int64]),
    "pylda_parse_corpus": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int, ctypes.c_char_p,
                                          ctypes.c_int64, ctypes.c_int, _c_int64_p, _c_int64_p, _c_int64_p,
                                          _c_int32_p, _c_int32_p, _c_int64_p]),
    "pylda_test_alpha_update": (ctypes.c_int, [_vp, _c_double_p, _c_double_p, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                               ctypes.c_int, ctypes.c_double, _c_double_p]),
    "pylda_test_expdigamma": (ctypes.c_int, [_vp, ctypes.c_int64, _c_double_p, ctypes.c_double, _c_double_p]),
    "pylda_test_special": (ctypes.c_int, [_vp, ctypes.c_int64, _c_double_p, _c_double_p, _c_double_p]),
}


def library_path():
    return _LIB_PATH


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same
    SONAME as the system one).  If this library pulled in the system copy first, a later
    `import torch` would load the bundled copy as a SECOND runtime and fail with "no
    ROCm-capable device".  Loading torch's copy first (when torch is installed; torch itself is
    not imported) makes both resolve to the same runtime, whatever the import order - which is
    also what lets the context run on torch's streams for the RCCL all-reduce.
    PYLDA_HIP_RUNTIME=system skips this (a host process that never imports torch; tested in
    tests/test_gpu_estep.py::test_runs_on_the_system_hip_runtime_without_torch)."""
    if os.environ.get("PYLDA_HIP_RUNTIME", "") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        bundled = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(bundled):
            ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)
    except Exception:           # no torch, or an unexpected layout: the system runtime is used
        pass


def load():
    """Load libpylda_hip.so; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            "pylda_amd: %s is missing - build it with `python -m pylda_amd.build` "
            "(or __graft_entry__.build()).  pylda_amd has no CPU fallback." % _LIB_PATH)
    _preload_torch_hip_runtime()
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError => ABI/header mismatch, fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.pylda_abi_version() != ABI_VERSION:
        raise RuntimeError("pylda_amd: %s implements ABI %d, this binding was written for ABI %d - rebuild it "
                           "(python -m pylda_amd.build --force)" % (_LIB_PATH, lib.pylda_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(_c_double_p) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(_c_int32_p) if a is not None else None


def _f64(a, shape=None, name="array"):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, a.shape, tuple(shape)))
    return a


class _PinnedBlock(object):
    """A page-locked host allocation exposed to numpy (array interface); goes back to the pool when the last
    array viewing it is garbage-collected."""

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    def __del__(self):
        try:
            _pinned_release(self.ptr, self.nbytes)
        except Exception:
            pass


# Idle page-locked blocks, oldest first: (pointer, nbytes).  The pool is bounded by BYTES, not by count per size - a
# long-running process that asks for many different shapes (inference() on corpora of varying D) must not accumulate
# page-locked memory - and arrays above _PINNED_MAX_ARRAY (the 2 GB gamma of a 1M-document corpus) are never pinned.
# __del__ of a block may run at any garbage-collection point, on any thread, also while pinned_empty() scans the list:
# one re-entrant lock around every access (re-entrant: a collection triggered inside the locked region may release a
# block on the same thread).
_pinned_idle = collections.deque()
_pinned_idle_bytes = 0
_pinned_lock = threading.RLock()
_PINNED_POOL_CAP = 1 << 30          # idle page-locked bytes kept for reuse (least recently released go first)
_PINNED_MAX_ARRAY = 1 << 29         # larger arrays come from ordinary (pageable) memory


def _pinned_release(ptr, nbytes):
    global _pinned_idle_bytes
    evicted = []
    with _pinned_lock:
        _pinned_idle.append((ptr, nbytes))
        _pinned_idle_bytes += nbytes
        while _pinned_idle and _pinned_idle_bytes > _PINNED_POOL_CAP:
            old_ptr, old_bytes = _pinned_idle.popleft()
            _pinned_idle_bytes -= old_bytes
            evicted.append(old_ptr)
    for old_ptr in evicted:                 # (the driver call outside the lock)
        if _lib is not None:
            _lib.pylda_host_free(_vp(old_ptr))


def pinned_pool_bytes():
    """Page-locked bytes currently idle in the pool (for tests and diagnostics)."""
    return _pinned_idle_bytes


def pinned_empty(shape, dtype=np.float64):
    """numpy.empty in page-locked host memory (pylda_host_alloc): the arrays e_step() / m_step() hand back and
    forth move at the PCIe rate.  Recycled through a small pool - page-locking 50 MB costs milliseconds - that is
    capped at _PINNED_POOL_CAP idle bytes; very large arrays and failed allocations fall back to numpy.empty."""
    global _pinned_idle_bytes
    lib = load()
    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
    if nbytes == 0 or nbytes > _PINNED_MAX_ARRAY:
        return np.empty(shape, dtype=dtype)
    ptr = None
    with _pinned_lock:
        for i in range(len(_pinned_idle) - 1, -1, -1):      # most recently released block of exactly this size
            if _pinned_idle[i][1] == nbytes:
                ptr = _pinned_idle[i][0]
                del _pinned_idle[i]
                _pinned_idle_bytes -= nbytes
                break
    if ptr is None:
        handle = _vp()
        rc = lib.pylda_host_alloc(nbytes, ctypes.byref(handle))
        if rc != 0:                          # (no page-locked memory left: ordinary memory works everywhere)
            return np.empty(shape, dtype=dtype)
        ptr = handle.value
    return np.asarray(_PinnedBlock(ptr, nbytes)).view(dtype).reshape(shape)


class PyldaError(RuntimeError):
    def __init__(self, status, message):
        RuntimeError.__init__(self, "pylda_hip error %d: %s" % (status, message))
        self.status = status


def _join_documents(lines):
    """One bytes object + the separator byte: a byte that occurs in no document (so a document
    that itself contains line breaks stays ONE document, as in the reference's per-string loop)."""
    for sep in ("\x00", "\x01", "\x02", "\x03"):      # never white space, (almost) never in text
        text = sep.join(lines)
        if text.count(sep) == max(len(lines) - 1, 0):
            return text.encode("utf-8", "surrogatepass"), ord(sep)
    # every candidate occurs somewhere: 0xFF is not a byte of any UTF-8 sequence
    return b"\xff".join(l.encode("utf-8", "surrogatepass") for l in lines), 0xFF


def parse_corpus(lines, vocabulary, lowercase=False):
    """Native parse_data (variational_bayes.py:98-130): documents (iterable of str) and the
    vocabulary in id order -> CSR (doc_ptr int64, term_id int32, term_ct int32), #dropped.

    A vocabulary entry that is empty or contains white space can never equal a token of
    str.split(); it keeps its id (a placeholder no token can match takes its line)."""
    lib = load()
    lines = list(lines)
    text, sep = _join_documents(lines)
    vocab = b"\n".join(w.encode("utf-8", "surrogatepass") if len(w.split()) == 1 and w == w.strip()
                       else b"\xfe%d" % i for i, w in enumerate(vocabulary))
    n_docs, nnz = ctypes.c_int64(0), ctypes.c_int64(0)
    rc = lib.pylda_parse_corpus(text, len(text), sep, vocab, len(vocab), 1 if lowercase else 0,
                                ctypes.byref(n_docs), ctypes.byref(nnz), None, None, None, None)
    if rc != 0:
        raise PyldaError(rc, "pylda_parse_corpus (sizing pass)")
    doc_ptr = np.zeros(n_docs.value + 1, dtype=np.int64)
    term_id = np.zeros(max(nnz.value, 1), dtype=np.int32)
    term_ct = np.zeros(max(nnz.value, 1), dtype=np.int32)
    rc = lib.pylda_parse_corpus(text, len(text), sep, vocab, len(vocab), 1 if lowercase else 0,
                                ctypes.byref(n_docs), ctypes.byref(nnz),
                                doc_ptr.ctypes.data_as(_c_int64_p), term_id.ctypes.data_as(_c_int32_p),
                                term_ct.ctypes.data_as(_c_int32_p), None)
    if rc != 0:
        raise PyldaError(rc, "pylda_parse_corpus")
    return doc_ptr, term_id[:nnz.value], term_ct[:nnz.value], len(lines) - n_docs.value


def comm_unique_id():
    """128 opaque bytes naming a new RCCL communicator (call on rank 0, hand to every rank)."""
    lib = load()
    buf = ctypes.create_string_buffer(128)
    rc = lib.pylda_comm_unique_id(ctypes.cast(buf, _vp))
    if rc != 0:
        raise PyldaError(rc, (lib.pylda_last_error(None) or b"").decode())
    return buf.raw


def device_count():
    n = ctypes.c_int(0)
    load().pylda_device_count(ctypes.byref(n))
    return n.value


class Context(object):
    """A model-sized device context: K topics over V word types on one GPU."""

    def __init__(self, number_of_topics, number_of_types, device=0):
        lib = load()
        handle = _vp()
        rc = lib.pylda_create(int(device), int(number_of_topics), int(number_of_types),
                              ctypes.byref(handle))
        if rc != 0:
            raise PyldaError(rc, (lib.pylda_last_error(None) or b"").decode())
        self._lib = lib
        self._h = handle
        self.K = int(number_of_topics)
        self.V = int(number_of_types)
        self.device = int(device)

    def _check(self, rc):
        if rc != 0:
            raise PyldaError(rc, (self._lib.pylda_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pylda_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- model state ----
    def set_eta(self, eta):
        self._check(self._lib.pylda_set_eta(self._h, _dp(_f64(eta, (self.K, self.V), "eta"))))

    def get_eta(self):
        out = pinned_empty((self.K, self.V))
        self._check(self._lib.pylda_get_eta(self._h, _dp(out)))
        return out

    def set_alpha(self, alpha):
        self._check(self._lib.pylda_set_alpha(self._h, _dp(_f64(alpha, (self.K,), "alpha"))))

    def set_stream(self, hip_stream):
        """Run on the HIP stream with this handle; 0 / None is the device's default (null) stream."""
        self._check(self._lib.pylda_set_stream(self._h, _vp(hip_stream) if hip_stream else None))

    def use_own_stream(self):
        self._check(self._lib.pylda_use_own_stream(self._h))

    def synchronize(self):
        self._check(self._lib.pylda_synchronize(self._h))

    def set_option(self, name, value):
        self._check(self._lib.pylda_set_option(self._h, name.encode(), int(value)))

    # ---- corpus ----
    def corpus(self, doc_ptr, term_id, term_ct):
        return Corpus(self, doc_ptr, term_id, term_ct)

    # ---- hot path ----
    def estep(self, corpus, max_iter=50, tol=1e-6, heldout=False):
        self._check(self._lib.pylda_estep(self._h, corpus._h, int(max_iter), float(tol),
                                          1 if heldout else 0))

    def estep_results(self, corpus):
        ll, wll, nlog = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0)
        self._check(self._lib.pylda_estep_results(self._h, corpus._h, ctypes.byref(ll),
                                                  ctypes.byref(wll), ctypes.byref(nlog)))
        return ll.value, wll.value, nlog.value

    def get_sstats(self):
        out = pinned_empty((self.K, self.V))
        self._check(self._lib.pylda_get_sstats(self._h, _dp(out)))
        return out

    def set_sstats(self, sstats):
        self._check(self._lib.pylda_set_sstats(self._h, _dp(_f64(sstats, (self.K, self.V), "sstats"))))

    def get_gamma(self, corpus):
        out = pinned_empty((corpus.D, self.K))
        self._check(self._lib.pylda_get_gamma(self._h, corpus._h, _dp(out)))
        return out

    def get_doc_values(self, corpus, want_ll=True):
        """(doc_ll, doc_words_ll, iters) of the last E-step; want_ll=False skips the per-document
        likelihoods (unavailable after an E-step on the training fast path, option doc_values=0)."""
        ll = np.empty(corpus.D, dtype=np.float64) if want_ll else None
        wll = np.empty(corpus.D, dtype=np.float64) if want_ll else None
        iters = np.empty(corpus.D, dtype=np.int32)
        self._check(self._lib.pylda_get_doc_values(self._h, corpus._h, _dp(ll), _dp(wll), _ip(iters)))
        return ll, wll, iters

    def estep_host(self, corpus, alpha, eta, max_iter=50, tol=1e-6, heldout=False,
                   want_doc_values=True):
        """One-shot form: returns a dict of host arrays."""
        alpha = _f64(alpha, (self.K,), "alpha")
        eta = _f64(eta, (self.K, self.V), "eta")
        gamma = np.empty((corpus.D, self.K), dtype=np.float64)
        sstats = None if heldout else np.empty((self.K, self.V), dtype=np.float64)
        ll = np.empty(corpus.D) if want_doc_values else None
        wll = np.empty(corpus.D) if want_doc_values else None
        iters = np.empty(corpus.D, dtype=np.int32) if want_doc_values else None
        scal = np.zeros(2)
        self.set_option("doc_values", 1 if want_doc_values else 0)
        self._check(self._lib.pylda_estep_host(
            self._h, corpus._h, _dp(alpha), _dp(eta), int(max_iter), float(tol),
            1 if heldout else 0, _dp(gamma), _dp(sstats), _dp(ll), _dp(wll), _ip(iters), _dp(scal)))
        return {"document_log_likelihood": scal[0], "words_log_likelihood": scal[1],
                "gamma": gamma, "sstats": sstats, "doc_ll": ll, "doc_words_ll": wll, "iters": iters}

    def mstep(self, corpus, beta, want_alpha_ss=True):
        beta = _f64(beta, (self.V,), "beta")
        tll = ctypes.c_double(0)
        ass = np.empty(self.K, dtype=np.float64) if want_alpha_ss else None
        self._check(self._lib.pylda_mstep(self._h, corpus._h if corpus is not None else None,
                                          _dp(beta), ctypes.byref(tll), _dp(ass)))
        return tll.value, ass

    def mstep_enqueue(self, corpus, beta, hyper_parameter_iteration=0, hyper_parameter_decay_factor=0.9,
                      hyper_parameter_maximum_decay=10, hyper_parameter_converge_threshold=1e-6):
        """The device half of m_step, nothing waited for (see outer_fetch); hyper_parameter_iteration > 0 also
        schedules the alpha update (optimize_hyperparameters) on the device."""
        beta = _f64(beta, (self.V,), "beta")
        self._check(self._lib.pylda_mstep_enqueue(self._h, corpus._h, _dp(beta), int(hyper_parameter_iteration),
                                                  float(hyper_parameter_decay_factor), int(hyper_parameter_maximum_decay),
                                                  float(hyper_parameter_converge_threshold)))

    def outer_device(self):
        """(device pointer, elements, leading elements that are rank-local sums) of the packed outer-iteration values."""
        n = ctypes.c_int64(0)
        ptr = self._lib.pylda_outer_device(self._h, ctypes.byref(n))
        return int(ptr or 0), 3 * self.K + 4, int(n.value)

    def allreduce_outer(self):
        self._check(self._lib.pylda_allreduce_outer(self._h))

    def outer_fetch(self):
        """(document_log_likelihood, number_of_documents, logspace_documents, topic_log_likelihood, alpha_ss,
        alpha): the one host wait of an outer iteration; alpha is what the device holds now (updated there when
        mstep_enqueue asked for it)."""
        ll, nd, tll, nlog = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0)
        ass, alpha = np.empty(self.K, dtype=np.float64), np.empty(self.K, dtype=np.float64)
        self._check(self._lib.pylda_outer_fetch(self._h, ctypes.byref(ll), ctypes.byref(nd), ctypes.byref(nlog),
                                                ctypes.byref(tll), _dp(ass), _dp(alpha)))
        return ll.value, int(round(nd.value)), nlog.value, tll.value, ass, alpha

    def model_checkpoint(self, restore=False):
        self._check(self._lib.pylda_model_checkpoint(self._h, 1 if restore else 0))

    def mark_time(self, slot):
        self._check(self._lib.pylda_mark_time(self._h, int(slot)))

    def elapsed_ms(self, slot_from, slot_to):
        ms = ctypes.c_double(0)
        self._check(self._lib.pylda_elapsed_ms(self._h, int(slot_from), int(slot_to), ctypes.byref(ms)))
        return ms.value

    def work_counters(self):
        """(sum_d I_d, sum_d I_d N_d) of the profiled E-steps since the last call."""
        a, b = ctypes.c_double(0), ctypes.c_double(0)
        self._check(self._lib.pylda_work_counters(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def executed_work(self):
        """(tile entries the kernels ran through the FMA pipes, documents handed to the live-topic kernel) of the E-steps
        the LAST work_counters() call read (no device access)."""
        a, b = ctypes.c_double(0), ctypes.c_double(0)
        self._check(self._lib.pylda_executed_work(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def shader_clock_mhz(self):
        """Sustained shader clock under the load of the E-steps the last work_counters() call read, or None (no span was
        sampled): resident kernels time themselves in shader cycles and in ticks of the constant-rate counter."""
        a, b, hz = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
        self._check(self._lib.pylda_clock_counters(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(hz)))
        return a.value / b.value * hz.value / 1e6 if b.value > 0 else None

    # ---- device-resident interop ----
    def sstats_elements(self):
        """Number of doubles behind sstats_device_ptr(): V * table stride."""
        return self.V * int(self._lib.pylda_table_stride(self._h))

    def sstats_device_ptr(self):
        return int(self._lib.pylda_sstats_device(self._h) or 0)

    def eta_device_ptr(self):
        return int(self._lib.pylda_eta_device(self._h) or 0)

    def mark_device_state(self, have_eta=-1, have_sstats=-1):
        self._check(self._lib.pylda_mark_device_state(self._h, int(have_eta), int(have_sstats)))

    # ---- multi-GPU exchange through the C ABI (RCCL bound at run time; no torch involved) ----
    def comm_init(self, unique_id, rank, world_size):
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        self._check(self._lib.pylda_comm_init(self._h, ctypes.cast(buf, _vp), int(rank), int(world_size)))

    def comm_destroy(self):
        self._check(self._lib.pylda_comm_destroy(self._h))

    def allreduce_sstats(self):
        self._check(self._lib.pylda_allreduce_sstats(self._h))

    def allreduce_doubles(self, values):
        values = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._check(self._lib.pylda_allreduce_doubles(self._h, _dp(values), values.size))
        return values

    # ---- profiling ----
    def set_profiling(self, enabled):
        self._check(self._lib.pylda_set_profiling(self._h, 1 if enabled else 0))

    def kernel_time(self):
        """(document-kernel ms, statistics-pass ms, E-steps) accumulated since the last call."""
        ms, ss, calls = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0)
        self._check(self._lib.pylda_kernel_time(self._h, ctypes.byref(ms), ctypes.byref(ss), ctypes.byref(calls)))
        return ms.value, ss.value, calls.value

    def test_alpha_update(self, alpha, alpha_ss, number_of_documents, iterations=100, decay_factor=0.9, maximum_decay=10,
                          threshold=1e-6):
        alpha, alpha_ss = _f64(alpha, (self.K,), "alpha"), _f64(alpha_ss, (self.K,), "alpha_ss")
        out = np.empty(self.K, dtype=np.float64)
        self._check(self._lib.pylda_test_alpha_update(self._h, _dp(alpha), _dp(alpha_ss), float(number_of_documents),
                                                      int(iterations), float(decay_factor), int(maximum_decay),
                                                      float(threshold), _dp(out)))
        return out

    def test_expdigamma(self, x, c):
        x = _f64(x)
        out = np.empty_like(x)
        self._check(self._lib.pylda_test_expdigamma(self._h, x.size, _dp(x), float(c), _dp(out)))
        return out

    def test_special(self, x):
        x = _f64(x)
        dg, lg = np.empty_like(x), np.empty_like(x)
        self._check(self._lib.pylda_test_special(self._h, x.size, _dp(x), _dp(dg), _dp(lg)))
        return dg, lg


class Corpus(object):
    """A parsed corpus resident on the device (CSR of distinct term ids/counts)."""

    def __init__(self, ctx, doc_ptr, term_id, term_ct):
        doc_ptr = np.ascontiguousarray(doc_ptr, dtype=np.int64)
        term_id = np.ascontiguousarray(term_id, dtype=np.int32)
        term_ct = np.ascontiguousarray(term_ct, dtype=np.int32)
        if doc_ptr.ndim != 1 or doc_ptr.size < 1:
            raise ValueError("doc_ptr must hold D+1 offsets")
        if term_id.shape != term_ct.shape or term_id.ndim != 1 or term_id.size != doc_ptr[-1]:
            raise ValueError("term_id/term_ct must be 1-D of length doc_ptr[-1]")
        handle = _vp()
        rc = ctx._lib.pylda_corpus_create(
            ctx._h, doc_ptr.size - 1, doc_ptr.ctypes.data_as(_c_int64_p),
            term_id.ctypes.data_as(_c_int32_p), term_ct.ctypes.data_as(_c_int32_p),
            ctypes.byref(handle))
        ctx._check(rc)
        self._ctx = ctx
        self._h = handle
        self.D = int(doc_ptr.size - 1)
        self.nnz = int(term_id.size)
        tokens = ctypes.c_int64(0)
        ctx._lib.pylda_corpus_info(handle, None, None, ctypes.byref(tokens), None)      # (summed while validating)
        self.tokens = int(tokens.value)

    def gamma_device_ptr(self):
        return int(self._ctx._lib.pylda_gamma_device(self._h) or 0)

    VARIANT_NAMES = {0: "generic64", 1: "generic256", 2: "generic512", 3: "generic_global", 4: "slab",
                     6: "quilt", 9: "qgroup", 10: "quad", 11: "qfuse", 12: "generic_huge", 13: "qfusek"}

    def plan(self):
        """Launch classes of this corpus: list of dicts (kernel, geometry, documents, terms, kernel_ms)."""
        lib = self._ctx._lib
        n = lib.pylda_corpus_plan(self._h, 0, None, None, None, None, None)
        if n < 0:
            self._ctx._check(n)
        variant, geometry = np.zeros(n, np.int32), np.zeros(n, np.int32)
        documents, terms, ms = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.float64)
        rc = lib.pylda_corpus_plan(self._h, n, _ip(variant), _ip(geometry), documents.ctypes.data_as(_c_int64_p),
                                   terms.ctypes.data_as(_c_int64_p), _dp(ms))
        if rc < 0:
            self._ctx._check(rc)
        return [{"kernel": self.VARIANT_NAMES.get(int(variant[i]), str(int(variant[i]))), "geometry": int(geometry[i]),
                 "documents": int(documents[i]), "terms": int(terms[i]), "kernel_ms": float(ms[i])}
                for i in range(n)]

    def layout(self, name):
        """Layout facts: "gather_blocks", "gather_segments" (include/pylda_hip.h)."""
        v = self._ctx._lib.pylda_corpus_layout(self._h, name.encode())
        if v < 0:
            self._ctx._check(int(v))
        return int(v)

    def close(self):
        if getattr(self, "_h", None) and getattr(self._ctx, "_h", None):
            self._ctx._lib.pylda_corpus_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
