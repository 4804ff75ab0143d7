"""Pins the CPU oracles (oracle/vb_numpy.py, oracle/vb_oracle.c) against the
golden vectors produced by running the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import csr_slice, load_golden, rel_err
from oracle import c_oracle, vb_numpy


def test_special_functions_c_oracle_vs_scipy():
    g = load_golden("special_fn.npz")
    x = g["x"]
    dg = c_oracle.digamma(x)
    # psi has a zero near 1.4616: compare absolutely, scaled by max(1, |psi|)
    assert np.max(np.abs(dg - g["psi"]) / np.maximum(1.0, np.abs(g["psi"]))) < 2e-14
    tg = c_oracle.trigamma(x)
    assert rel_err(tg, g["trigamma"]) < 1e-13
    lg = c_oracle.lgamma(x)
    assert np.max(np.abs(lg - g["gammaln"]) / np.maximum(1.0, np.abs(g["gammaln"]))) < 2e-14


def test_tiny_numpy_oracle_matches_reference(tiny):
    t = tiny
    out = vb_numpy.e_step(t["alpha"], t["eta"], t["doc_ptr"], t["term_id"], t["term_ct"])
    assert np.array_equal(out["iters"], t["iters"])
    assert rel_err(out["gamma"], t["gamma"]) < 1e-13
    assert rel_err(out["doc_ll"], t["doc_ll"]) < 1e-12
    assert np.max(np.abs(out["sstats"] - t["sstats"])) < 1e-13
    assert abs(out["document_log_likelihood"] - float(t["corpus_ll"])) < 1e-12
    assert abs(out["sstats"].sum() - t["term_ct"].sum()) < 1e-10      # SURVEY 8a a6 invariant
    held = vb_numpy.e_step(t["alpha"], t["eta"], t["doc_ptr"], t["term_id"], t["term_ct"],
                           heldout=True)
    assert rel_err(held["gamma"], t["heldout_gamma"]) < 1e-13
    assert rel_err(held["doc_words_ll"], t["heldout_words_ll"]) < 1e-12
    assert abs(held["words_log_likelihood"] - float(t["heldout_corpus_words_ll"])) < 1e-11


def test_tiny_c_oracle_matches_reference(tiny):
    t = tiny
    out = c_oracle.e_step(t["alpha"], t["eta"], t["doc_ptr"], t["term_id"], t["term_ct"])
    assert np.array_equal(out["iters"], t["iters"])
    assert rel_err(out["gamma"], t["gamma"]) < 1e-12
    assert rel_err(out["doc_ll"], t["doc_ll"]) < 1e-11
    assert np.max(np.abs(out["sstats"] - t["sstats"])) < 1e-12
    held = c_oracle.e_step(t["alpha"], t["eta"], t["doc_ptr"], t["term_id"], t["term_ct"],
                           heldout=True)
    assert rel_err(held["doc_words_ll"], t["heldout_words_ll"]) < 1e-11
    assert rel_err(held["gamma"], t["heldout_gamma"]) < 1e-12


def test_ap_train_numpy_oracle_subset(ap_train):
    g = ap_train
    docs = list(range(0, 2000, 25))                     # 80 documents, a few seconds
    ptr, tid, tct = csr_slice(g["doc_ptr"], g["term_id"], g["term_ct"], docs)
    out = vb_numpy.e_step(g["alpha"], g["eta"], ptr, tid, tct)
    assert np.array_equal(out["iters"], g["iters"][docs])
    assert rel_err(out["gamma"], g["gamma"][docs]) < 1e-12
    assert rel_err(out["doc_ll"], g["doc_ll"][docs]) < 1e-11


def test_ap_train_c_oracle_full(ap_train):
    g = ap_train
    out = c_oracle.e_step(g["alpha"], g["eta"], g["doc_ptr"], g["term_id"], g["term_ct"])
    assert np.array_equal(out["iters"], g["iters"])         # every one of the 2000 documents stops where the reference does
    assert rel_err(out["gamma"], g["gamma"]) < 1e-10
    assert rel_err(out["doc_ll"], g["doc_ll"]) < 1e-10
    assert np.max(np.abs(out["sstats"] - g["sstats"])) < 1e-7
    assert abs(out["document_log_likelihood"] - float(g["corpus_ll"])) < 1e-6 * abs(float(g["corpus_ll"]))
    assert abs(out["sstats"].sum() - g["term_ct"].sum()) < 1e-6
    assert rel_err(g["gamma_corpus"], g["gamma"]) < 1e-12   # per-doc goldens == corpus run


def test_ap_heldout_c_oracle(ap_test):
    g = ap_test
    assert int(g["unseen_types"]) == 30                     # SURVEY 8c fixture (2)
    out = c_oracle.e_step(g["alpha"], g["eta"], g["doc_ptr"], g["term_id"], g["term_ct"],
                          heldout=True)
    assert np.array_equal(out["iters"], g["iters"])
    assert rel_err(out["gamma"], g["gamma"]) < 1e-10
    assert rel_err(out["doc_words_ll"], g["words_ll"]) < 1e-10
    assert abs(out["words_log_likelihood"] - float(g["corpus_words_ll"])) < 1e-7 * abs(float(g["corpus_words_ll"]))


def test_ap_heldout_numpy_oracle_subset(ap_test):
    g = ap_test
    docs = list(range(0, 221, 8))
    ptr, tid, tct = csr_slice(g["doc_ptr"], g["term_id"], g["term_ct"], docs)
    out = vb_numpy.e_step(g["alpha"], g["eta"], ptr, tid, tct, heldout=True)
    assert np.array_equal(out["iters"], g["iters"][docs])
    assert rel_err(out["doc_words_ll"], g["words_ll"][docs]) < 1e-12
    assert rel_err(out["gamma"], g["gamma"][docs]) < 1e-12


def test_mstep_and_alpha_update_numpy_oracle(ap_train):
    g = ap_train
    topic_ll, alpha_ss, new_eta = vb_numpy.m_step(g["eta"], g["beta"], g["sstats"], g["gamma_corpus"])
    assert abs(topic_ll - float(g["topic_ll"])) < 1e-12 * abs(float(g["topic_ll"]))
    assert rel_err(alpha_ss, g["alpha_ss"]) < 1e-13
    assert np.array_equal(new_eta, g["eta_after"])
    alpha = vb_numpy.optimize_hyperparameters(g["alpha"], g["alpha_ss"], 2000)
    assert rel_err(alpha, g["alpha_after"]) < 1e-12
