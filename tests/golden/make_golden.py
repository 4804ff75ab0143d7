#!/usr/bin/env python3
"""Generate the committed golden fixtures by RUNNING THE REFERENCE ITSELF.

Container-only: needs /root/reference (imported through _ref_loader's
in-memory lib2to3 translation) and its associated-press.tar.gz data archive.
Outputs are small .npz files of inputs and expected outputs (data, no
source) written next to this script; they are what travels to the GPU box.

    PYTHONHASHSEED=0 python tests/golden/make_golden.py                   # tiny + AP + nips fixtures
    PYTHONHASHSEED=0 python tests/golden/make_golden.py --only ap --trace 100   # 100-iteration AP trace (~30 min)
    PYTHONHASHSEED=0 python tests/golden/make_golden.py --only nipstrace  # nips K=500 trace (~5 min/iteration)

(The reference derives word ids from Python's set() order; PYTHONHASHSEED pins it so that a re-run
reproduces the committed files.  Every fixture stores the ids it was made with, so the tests do
not depend on the hash seed.)

Fixtures
  tiny_k2.npz        D=3, V=7, K=2 hand-checkable corpus, training + held-out
  tiny_exports.npz   the bytes export_beta / export_gamma write for that corpus after two
                     learning() iterations (full and top_display=2), with the eta / gamma they came from
  ap_train_k10.npz   AP train split (2000 docs) K=10: state after 2 learning()
                     iterations, then per-document goldens of the 3rd e_step,
                     the m_step outputs and the alpha Newton update
  ap_test_k10.npz    AP test split (221 docs, 30 word types unseen in
                     training) in held-out mode against the same model
  ap_trace_k10.npz   per-iteration joint log-likelihood / alpha trace
  special_fn.npz     scipy psi / gammaln / polygamma samples used to pin the
                     C oracle's and the HIP kernel's special functions
  nips_k500.npz      parsed/nips.88-05, K=500, 48 training + 16 held-out documents
                     (up to ~480 distinct terms): first E-step from the seeded eta
"""
import argparse
import io
import os
import sys
import tarfile
import contextlib

import numpy as np
import scipy.special

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_loader import REFERENCE_ROOT, load_reference  # noqa: E402


def read_ap():
    tf = tarfile.open(os.path.join(REFERENCE_ROOT, "associated-press.tar.gz"))
    def lines(name):
        data = tf.extractfile("associated-press/" + name).read().decode("utf-8")
        return data.splitlines()
    train = [l.strip().lower() for l in lines("train.dat")]          # launch_train.py:106
    test = [l.strip().lower() for l in lines("test.dat")]            # launch_test.py:66
    vocab = [l.strip().lower().split()[0] for l in lines("voc.dat")]  # launch_train.py:114
    return train, test, vocab


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **kw)


def csr_of(parsed):
    ids, cts = parsed
    ptr = np.zeros(len(ids) + 1, dtype=np.int64)
    for d, w in enumerate(ids):
        ptr[d + 1] = ptr[d] + len(w)
    tid = np.concatenate([np.asarray(w).ravel() for w in ids]).astype(np.int32)
    tct = np.concatenate([np.asarray(c).ravel() for c in cts]).astype(np.int32)
    return ptr, tid, tct


class PsiCounter:
    """Counts scipy.special.psi calls to recover the inner-iteration count."""

    def __init__(self):
        self.real = scipy.special.psi
        self.calls = 0

    def __enter__(self):
        def counted(x, *a, **kw):
            self.calls += 1
            return self.real(x, *a, **kw)
        scipy.special.psi = counted
        return self

    def __exit__(self, *exc):
        scipy.special.psi = self.real


def per_document(model, parsed, heldout):
    """Per-document goldens via single-document e_step calls."""
    ids, cts = parsed
    D, K = len(ids), model._number_of_topics
    gamma = np.zeros((D, K))
    val = np.zeros(D)
    iters = np.zeros(D, dtype=np.int32)
    saved_corpus, saved_gamma = model._parsed_corpus, model._gamma
    for d in range(D):
        one = ([ids[d]], [cts[d]])
        with PsiCounter() as pc:
            if heldout:
                v, g = quiet(model.e_step, one)
                gamma[d] = g[0]
            else:
                model._parsed_corpus = one
                v, _ = quiet(model.e_step)
                gamma[d] = model._gamma[0]
        val[d] = v
        iters[d] = pc.calls - 2          # 2 psi calls in compute_dirichlet_expectation
    model._parsed_corpus, model._gamma = saved_corpus, saved_gamma
    return gamma, val, iters


def make_tiny(vb):
    docs = ["a b b c", "c c d e e e f", "g a a g b"]
    vocab = ["a", "b", "c", "d", "e", "f", "g"]
    np.random.seed(7)
    m = vb.VariationalBayes()
    quiet(m._initialize, docs, vocab, 2, 0.5, 0.1)
    words = [m._index_to_type[i] for i in range(len(vocab))]
    ptr, tid, tct = csr_of(m._parsed_corpus)
    quiet(m.learning)
    alpha, eta = m._alpha_alpha.copy(), m._eta.copy()
    gamma, doc_ll, iters = per_document(m, m._parsed_corpus, heldout=False)
    np.random.seed(11)
    ll, sstats = quiet(m.e_step)
    gamma_corpus = m._gamma.copy()
    hgamma, hwll, hiters = per_document(m, m._parsed_corpus, heldout=True)
    np.random.seed(12)
    wll, hg = quiet(m.e_step, m._parsed_corpus)
    np.savez_compressed(
        os.path.join(HERE, "tiny_k2.npz"), words=np.array(words), docs=np.array(docs),
        doc_ptr=ptr, term_id=tid, term_ct=tct, alpha=alpha, eta=eta,
        gamma=gamma, doc_ll=doc_ll, iters=iters, corpus_ll=ll, sstats=sstats,
        gamma_corpus=gamma_corpus, heldout_gamma=hgamma, heldout_words_ll=hwll,
        heldout_iters=hiters, heldout_corpus_words_ll=wll, heldout_corpus_gamma=hg)
    print("tiny_k2: corpus_ll=%r words_ll=%r iters=%s" % (ll, wll, iters))


def make_tiny_exports(vb):
    """The reference's export_beta / export_gamma text (variational_bayes.py:326-356) on the tiny
    case, full and with top_display=2: the model state it was written from and the exact bytes."""
    import tempfile
    docs = ["a b b c", "c c d e e e f", "g a a g b"]
    vocab = ["a", "b", "c", "d", "e", "f", "g"]
    np.random.seed(7)
    m = vb.VariationalBayes()
    quiet(m._initialize, docs, vocab, 2, 0.5, 0.1)
    quiet(m.learning)
    quiet(m.learning)
    words = [m._index_to_type[i] for i in range(len(vocab))]
    texts = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, fn, top in (("exp_beta", m.export_beta, -1), ("exp_beta_top2", m.export_beta, 2),
                              ("exp_gamma", m.export_gamma, -1), ("exp_gamma_top2", m.export_gamma, 2)):
            path = os.path.join(tmp, name)
            fn(path, top)
            with open(path, "rb") as fh:
                texts[name] = np.frombuffer(fh.read(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "tiny_exports.npz"), words=np.array(words), eta=m._eta.copy(),
                        gamma=m._gamma.copy(), **texts)
    print("tiny_exports: %s" % {k: v.size for k, v in texts.items()})


def make_ap(vb, trace_iters):
    train, test, vocab = read_ap()
    vocab = list(dict.fromkeys(vocab))
    K = 10
    np.random.seed(0)
    m = vb.VariationalBayes()
    quiet(m._initialize, train, vocab, K, 1.0 / K, 1.0 / len(vocab))  # launch_train.py:119-124,194
    words = np.array([m._index_to_type[i] for i in range(len(vocab))])
    ptr, tid, tct = csr_of(m._parsed_corpus)
    eta0 = m._eta.copy()
    trace_ll, trace_alpha = [], []
    for _ in range(2):
        trace_ll.append(quiet(m.learning))
        trace_alpha.append(m._alpha_alpha.copy())
    alpha, eta = m._alpha_alpha.copy(), m._eta.copy()
    beta = m._alpha_beta.copy()

    gamma, doc_ll, iters = per_document(m, m._parsed_corpus, heldout=False)
    np.random.seed(100)
    ll, sstats = quiet(m.e_step)
    gamma_corpus = m._gamma.copy()
    topic_ll, alpha_ss = m.m_step(sstats)
    eta_after = m._eta.copy()
    quiet(m.optimize_hyperparameters, alpha_ss)
    alpha_after = m._alpha_alpha.copy()
    m._counter += 1
    trace_ll.append(ll + topic_ll)
    trace_alpha.append(alpha_after.copy())
    np.savez_compressed(
        os.path.join(HERE, "ap_train_k10.npz"), words=words, doc_ptr=ptr,
        term_id=tid.astype(np.int16), term_ct=tct.astype(np.int16), alpha=alpha, eta=eta,
        beta=beta, gamma=gamma, doc_ll=doc_ll, iters=iters, corpus_ll=ll, sstats=sstats,
        gamma_corpus=gamma_corpus, topic_ll=topic_ll, alpha_ss=alpha_ss,
        eta_after=eta_after, alpha_after=alpha_after)
    print("ap_train_k10: corpus_ll=%r mean iters=%.2f tokens=%d" % (ll, iters.mean(), tct.sum()))

    # held-out split against the model after 3 iterations
    parsed_test = quiet(m.parse_data, test)
    tptr, ttid, ttct = csr_of(parsed_test)
    hgamma, hwll, hiters = per_document(m, parsed_test, heldout=True)
    np.random.seed(101)
    wll, hg = quiet(m.e_step, parsed_test)
    seen = np.zeros(len(vocab), bool)
    seen[tid] = True
    np.savez_compressed(
        os.path.join(HERE, "ap_test_k10.npz"), doc_ptr=tptr,
        term_id=ttid.astype(np.int16), term_ct=ttct.astype(np.int16),
        alpha=m._alpha_alpha.copy(), eta=m._eta.copy(), gamma=hgamma, words_ll=hwll,
        iters=hiters, corpus_words_ll=wll, corpus_gamma=hg,
        unseen_types=int((~seen[np.unique(ttid)]).sum()))
    print("ap_test_k10: words_ll=%r unseen types=%d" % (wll, (~seen[np.unique(ttid)]).sum()))

    if trace_iters > 3:
        for it in range(3, trace_iters):
            trace_ll.append(quiet(m.learning))
            trace_alpha.append(m._alpha_alpha.copy())
            print("trace iteration %d: %r" % (it + 1, trace_ll[-1]), flush=True)
    wll_end, _ = quiet(m.inference, test)
    np.savez_compressed(
        os.path.join(HERE, "ap_trace_k10.npz"), eta0=eta0.astype(np.float64),
        joint_ll=np.array(trace_ll), alpha=np.array(trace_alpha),
        heldout_words_ll_end=wll_end, seed=0)
    print("ap_trace_k10: %d iterations, final held-out words_ll=%r" % (len(trace_ll), wll_end))


def make_nips(vb):
    """BASELINE.json cfg 5 in miniature: parsed/nips.88-05, K=500 (documents with up to ~480
    distinct terms: the tile does not fit on chip, so this pins the large-K / long-document
    kernels against the reference itself).  eta is NOT stored (500 x 3209 doubles): it is the
    reference's own seeded draw, numpy.random.seed(5); gamma(100, 0.01, (K, V)) - numpy's legacy
    stream is stable across versions - and the test regenerates it."""
    tf = tarfile.open(os.path.join(REFERENCE_ROOT, "parsed", "nips.88-05.tar.gz"))
    docs = tf.extractfile("nips.88-05/doc.dat").read().decode("utf-8").splitlines()
    vocab = [l.strip().lower().split()[0] for l in
             tf.extractfile("nips.88-05/voc.dat").read().decode("utf-8").splitlines() if l.strip()]
    vocab = list(dict.fromkeys(vocab))
    docs = [l.strip().lower() for l in docs]
    train, test = docs[:48], docs[-16:]
    K = 500
    np.random.seed(5)
    m = vb.VariationalBayes()
    quiet(m._initialize, train, vocab, K, 1.0 / K, 1.0 / len(vocab))
    words = np.array([m._index_to_type[i] for i in range(len(vocab))])
    ptr, tid, tct = csr_of(m._parsed_corpus)
    gamma, doc_ll, iters = per_document(m, m._parsed_corpus, heldout=False)
    np.random.seed(50)
    ll, sstats = quiet(m.e_step)
    parsed_test = quiet(m.parse_data, test)
    tptr, ttid, ttct = csr_of(parsed_test)
    hgamma, hwll, hiters = per_document(m, parsed_test, heldout=True)
    # sstats are K x V = 12.8 MB: keep column sums, row sums and a strided sample instead
    np.savez_compressed(
        os.path.join(HERE, "nips_k500.npz"), words=words, seed=5, K=K, doc_ptr=ptr,
        term_id=tid.astype(np.int16), term_ct=tct.astype(np.int16), gamma=gamma, doc_ll=doc_ll, iters=iters,
        corpus_ll=ll, sstats_rowsum=sstats.sum(axis=1), sstats_colsum=sstats.sum(axis=0),
        sstats_sample=sstats[::7, ::11].copy(), test_doc_ptr=tptr, test_term_id=ttid.astype(np.int16),
        test_term_ct=ttct.astype(np.int16), heldout_gamma=hgamma, heldout_words_ll=hwll, heldout_iters=hiters)
    print("nips_k500: %d train docs, max terms %d, corpus_ll=%r, mean iters %.1f"
          % (len(ptr) - 1, np.diff(ptr).max(), ll, iters.mean()))


def make_nips_trace(vb, iterations, out=None):
    """BASELINE.json cfg 5: parsed/nips.88-05, K=500, train = first 2,235 documents, test = last 248
    (SURVEY 8d), seed 0, `iterations` learning() calls of the reference (about 5 minutes each on one
    core), then inference() on the test split.  The trace is rewritten after every iteration so that
    a partial run is still a usable fixture."""
    tf = tarfile.open(os.path.join(REFERENCE_ROOT, "parsed", "nips.88-05.tar.gz"))
    docs = [l.strip().lower() for l in tf.extractfile("nips.88-05/doc.dat").read().decode("utf-8").splitlines()]
    vocab = [l.strip().lower().split()[0] for l in
             tf.extractfile("nips.88-05/voc.dat").read().decode("utf-8").splitlines() if l.strip()]
    vocab = list(dict.fromkeys(vocab))
    train, test = docs[:2235], docs[-248:]
    K = 500
    np.random.seed(0)
    m = vb.VariationalBayes()
    quiet(m._initialize, train, vocab, K, 1.0 / K, 1.0 / len(vocab))
    words = np.array([m._index_to_type[i] for i in range(len(vocab))])
    ptr, tid, tct = csr_of(m._parsed_corpus)
    parsed_test = quiet(m.parse_data, test)
    tptr, ttid, ttct = csr_of(parsed_test)
    joint, alphas, heldout = [], [], []
    out = out or os.path.join(HERE, "nips_trace_k500.npz")
    for it in range(iterations):
        joint.append(quiet(m.learning))
        alphas.append(m._alpha_alpha.copy())
        if (it + 1) % 10 == 0 or it + 1 == iterations:
            wll, _ = quiet(m.e_step, parsed_test)
            heldout.append((it + 1, wll))
        np.savez_compressed(out, words=words, seed=0, K=K, doc_ptr=ptr, term_id=tid.astype(np.int16),
                            term_ct=tct.astype(np.int16), test_doc_ptr=tptr, test_term_id=ttid.astype(np.int16),
                            test_term_ct=ttct.astype(np.int16), joint_ll=np.array(joint),
                            alpha_mean=np.array([a.mean() for a in alphas]), alpha_last=alphas[-1],
                            heldout=np.array(heldout, dtype=np.float64).reshape(-1, 2),
                            test_tokens=int(ttct.sum()))
        print("nips trace iteration %d: %r" % (it + 1, joint[-1]), flush=True)


def make_special():
    rng = np.random.default_rng(3)
    x = np.concatenate([
        10.0 ** rng.uniform(-6, 4, 4000), rng.uniform(0.0, 12.0, 4000) + 1e-9,
        np.array([1e-5, 1.0 / 6806, 0.1, 0.5, 1.0, 1.4616321449683623, 2.0, 5.999, 6.0,
                  9.999, 10.0, 10.001, 100.0, 1234.5, 1e6])])
    np.savez_compressed(os.path.join(HERE, "special_fn.npz"), x=x, psi=scipy.special.psi(x),
                        gammaln=scipy.special.gammaln(x),
                        trigamma=scipy.special.polygamma(1, x))
    print("special_fn: %d samples" % x.size)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", type=int, default=3, help="AP K=10 trace length (iterations)")
    ap.add_argument("--only", default="", help="comma list of: tiny,exports,ap,special,nips,nipstrace")
    ap.add_argument("--nips-iterations", type=int, default=50)
    ap.add_argument("--nips-trace-out", default=None, help="write the nips trace here instead of over the committed fixture")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    _, vb = load_reference()
    if not only or "special" in only:
        make_special()
    if not only or "tiny" in only:
        make_tiny(vb)
    if not only or "exports" in only:
        make_tiny_exports(vb)
    if not only or "ap" in only:
        make_ap(vb, args.trace)
    if not only or "nips" in only:
        make_nips(vb)
    if "nipstrace" in only:                 # hours of CPU: only on request
        make_nips_trace(vb, args.nips_iterations, args.nips_trace_out)
