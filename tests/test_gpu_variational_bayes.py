"""The drop-in class: pylda_amd.variational_bayes.VariationalBayes against the
reference's own traces and return contracts.  Needs an MI355X."""
import os
import pickle

import numpy as np
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def documents_from_csr(words, ptr, ids, cts):
    docs = []
    for d in range(len(ptr) - 1):
        toks = []
        for n in range(int(ptr[d]), int(ptr[d + 1])):
            toks += [str(words[ids[n]])] * int(cts[n])
        docs.append(" ".join(toks))
    return docs


@pytest.fixture(scope="module")
def ap_model(ap_train):
    from pylda_amd.variational_bayes import VariationalBayes
    g = ap_train
    words = [str(w) for w in g["words"]]
    docs = documents_from_csr(words, g["doc_ptr"], g["term_id"], g["term_ct"])
    np.random.seed(0)                                   # the seed the goldens were made with
    m = VariationalBayes()
    m._verbose = False
    m._initialize(docs, words, 10, 1.0 / 10, 1.0 / len(words))
    return m


def test_learning_trace_matches_reference(ap_model, ap_train):
    tr = load_golden("ap_trace_k10.npz")
    m = ap_model
    assert m._number_of_documents == 2000 and m._number_of_types == 6806
    assert np.array_equal(m._eta, tr["eta0"])           # same RNG draw as variational_bayes.py:95
    n = min(6, len(tr["joint_ll"]))
    for it in range(n):
        joint = m.learning()
        assert abs(joint - tr["joint_ll"][it]) < 1e-8 * abs(tr["joint_ll"][it]), it
        assert rel_err(m._alpha_alpha, tr["alpha"][it]) < 1e-8, it
        if it == 1:                                     # state the per-document goldens start from
            assert rel_err(m._alpha_alpha, ap_train["alpha"]) < 1e-9
            assert rel_err(m._eta, ap_train["eta"]) < 1e-8
    assert m._counter == n


def test_learning_honours_an_overridden_e_step(ap_train):
    """The reference's learning() dispatches through self.e_step() / self.m_step() (variational_bayes.py:243-247) and
    hybrid.py:23,85 overrides e_step alone: a subclass that wraps e_step (or an instance that patches m_step) must see
    one call per learning(), and - perturbing nothing - the reference's trace must come out unchanged."""
    from pylda_amd.variational_bayes import VariationalBayes
    tr = load_golden("ap_trace_k10.npz")

    class Counting(VariationalBayes):
        calls = 0

        def e_step(self, parsed_corpus=None, local_parameter_iteration=50, local_parameter_converge_threshold=1e-6):
            if parsed_corpus is None:
                Counting.calls += 1
            return VariationalBayes.e_step(self, parsed_corpus, local_parameter_iteration, local_parameter_converge_threshold)

    g = ap_train
    words = [str(w) for w in g["words"]]
    docs = documents_from_csr(words, g["doc_ptr"], g["term_id"], g["term_ct"])
    np.random.seed(int(tr["seed"]))
    m = Counting()
    m._verbose = False
    m._initialize(docs, words, 10, 1.0 / 10, 1.0 / len(words))
    assert m._seam_is_overridden()
    m_calls = []
    for it in range(3):
        if it == 2:                     # ... and a method patched on the INSTANCE counts too
            inner = m.m_step
            m.m_step = lambda sstats: (m_calls.append(sstats.shape), inner(sstats))[1]
        joint = m.learning()
        assert abs(joint - tr["joint_ll"][it]) < 1e-8 * abs(tr["joint_ll"][it]), it
        assert rel_err(m._alpha_alpha, tr["alpha"][it]) < 1e-8, it
    assert Counting.calls == 3 and m._counter == 3
    assert m_calls == [(10, 6806)]
    plain = VariationalBayes()
    assert not plain._seam_is_overridden()


def test_hundred_iteration_trace_and_heldout(ap_train, ap_test):
    """BASELINE.json cfg 1/2 end to end: 100 learning() iterations on AP K=10 from the reference's
    seeded initial state reproduce its joint log-likelihood and alpha traces, then its held-out
    words log-likelihood (launch_train + launch_test flow)."""
    from pylda_amd.variational_bayes import VariationalBayes
    tr = load_golden("ap_trace_k10.npz")
    if len(tr["joint_ll"]) < 100:
        pytest.skip("short trace fixture")
    g = ap_train
    words = [str(w) for w in g["words"]]
    docs = documents_from_csr(words, g["doc_ptr"], g["term_id"], g["term_ct"])
    np.random.seed(int(tr["seed"]))
    m = VariationalBayes()
    m._verbose = False
    m._initialize(docs, words, 10, 1.0 / 10, 1.0 / len(words))
    joint = np.array([m.learning() for _ in range(100)])
    assert rel_err(joint, tr["joint_ll"]) < 1e-7
    assert rel_err(m._alpha_alpha, tr["alpha"][-1]) < 1e-6
    h = ap_test
    test_docs = documents_from_csr(words, h["doc_ptr"], h["term_id"], h["term_ct"])
    wll, gamma = m.inference(test_docs)
    assert abs(wll - float(tr["heldout_words_ll_end"])) < 1e-7 * abs(float(tr["heldout_words_ll_end"]))
    assert gamma.shape == (221, 10)


def test_e_step_m_step_contract(ap_model, ap_train):
    """Public e_step()/m_step() keep the reference's host-array contract (:212-216, :218-235)."""
    from pylda_amd.variational_bayes import VariationalBayes
    g = ap_train
    m = VariationalBayes()
    m._verbose = False
    words = [str(w) for w in g["words"]]
    np.random.seed(1)
    m._initialize(documents_from_csr(words, g["doc_ptr"][:301], g["term_id"], g["term_ct"]),
                  words, 10, 0.1, 1.0 / len(words))
    m._alpha_alpha = g["alpha"].copy()
    m._eta = g["eta"].copy()
    ll, sstats = m.e_step()
    assert isinstance(sstats, np.ndarray) and sstats.shape == (10, 6806) and sstats.dtype == np.float64
    assert m._gamma.shape == (300, 10)
    assert rel_err(m._gamma, g["gamma"][:300]) < 1e-9
    assert abs(ll - g["doc_ll"][:300].sum()) < 1e-9 * abs(g["doc_ll"][:300].sum())
    from oracle import vb_numpy
    topic_ll_ref, alpha_ss_ref, eta_ref = vb_numpy.m_step(g["eta"], m._alpha_beta, sstats, m._gamma)
    topic_ll, alpha_ss = m.m_step(sstats)
    assert abs(topic_ll - topic_ll_ref) < 1e-10 * abs(topic_ll_ref)
    assert rel_err(alpha_ss, alpha_ss_ref) < 1e-10
    assert rel_err(m._eta, eta_ref) < 1e-13
    alpha_ref = vb_numpy.optimize_hyperparameters(m._alpha_alpha, alpha_ss_ref, 300)
    m.optimize_hyperparameters(alpha_ss)
    assert rel_err(m._alpha_alpha, alpha_ref) < 1e-9


def test_inference_and_pickle_round_trip(ap_model, ap_test, tmp_path):
    g = ap_test
    m = ap_model
    path = tmp_path / "model-x"
    with open(path, "wb") as fh:
        pickle.dump(m, fh)                               # launch_train.py:203-204
    with open(path, "rb") as fh:
        m2 = pickle.load(fh)                             # launch_test.py:92
    assert m2._ctx is None and np.array_equal(m2._eta, m._eta)
    m2._verbose = False
    # held-out documents through the text interface, model state of the goldens
    m2._alpha_alpha = g["alpha"].copy()
    m2._eta = g["eta"].copy()
    words = [m2._index_to_type[i] for i in range(m2._number_of_types)]
    docs = documents_from_csr(words, g["doc_ptr"], g["term_id"], g["term_ct"])
    gamma_before = m2._gamma.copy()
    wll, gamma = m2.inference(docs)
    assert gamma.shape == (221, 10)
    assert abs(wll - float(g["corpus_words_ll"])) < 1e-9 * abs(float(g["corpus_words_ll"]))
    assert rel_err(gamma, g["gamma"]) < 1e-8
    assert np.array_equal(m2._gamma, gamma_before)       # :212-216: _gamma untouched in held-out mode


def test_exports(ap_model, tmp_path):
    m = ap_model
    m.export_beta(str(tmp_path / "exp_beta"), top_display=5)
    lines = open(tmp_path / "exp_beta").read().splitlines()
    assert lines[0] == "==========\t0\t==========" and len(lines) == 10 * 6
    word, prob = lines[1].split("\t")
    assert word in m._type_to_index and 0.0 < float(prob) <= 1.0
    m.export_gamma(str(tmp_path / "exp_gamma"))
    rows = open(tmp_path / "exp_gamma").read().splitlines()
    assert len(rows) == 2000 and len(rows[0].split("\t")) == 10
    probs = [float(x.split(":")[1]) for x in rows[0].split("\t")]
    assert probs == sorted(probs, reverse=True) and abs(sum(probs) - 1.0) < 1e-4


def test_launch_train_and_launch_test_drivers(ap_train, ap_test, tmp_path, capsys):
    """The launch_train / launch_test command lines end to end on a small on-disk corpus:
    output layout (launch_train.py:127-162,199-204) and the held-out flow (launch_test.py:90-97)."""
    from pylda_amd import launch_test, launch_train
    g = ap_train
    words = [str(w) for w in g["words"]]
    corpus_dir = tmp_path / "mini-press"
    corpus_dir.mkdir()
    docs = documents_from_csr(words, g["doc_ptr"][:121], g["term_id"], g["term_ct"])
    (corpus_dir / "train.dat").write_text("\n".join(d.upper() for d in docs[:100]) + "\n")     # lower-cased on load
    (corpus_dir / "test.dat").write_text("\n".join(docs[100:120]) + "\n")
    (corpus_dir / "voc.dat").write_text("".join("%s\t1\t1\n" % w for w in words))
    out_dir = tmp_path / "out"
    np.random.seed(3)
    rc = launch_train.main(["--input_directory=%s/" % corpus_dir, "--output_directory=%s" % out_dir,
                            "--number_of_topics=5", "--training_iterations=4", "--snapshot_interval=2"])
    assert rc == 0
    runs = list((out_dir / "mini-press").iterdir())
    assert len(runs) == 1 and "-lda-I4-S2-K5-aa0.200000-ab" in runs[0].name and runs[0].name.endswith("-im2")
    names = sorted(p.name for p in runs[0].iterdir())
    assert names == ["exp_beta-2", "exp_beta-4", "exp_gamma-2", "exp_gamma-4", "model-4", "option.txt"]
    opts = dict(l.split("=", 1) for l in (runs[0] / "option.txt").read_text().splitlines())
    assert opts["number_of_topics"] == "5" and opts["inference_mode"] == "2" and opts["corpus_name"] == "mini-press"
    assert len((runs[0] / "exp_gamma-4").read_text().splitlines()) == 100
    assert launch_train.main(["--input_directory=%s" % corpus_dir, "--output_directory=%s" % out_dir,
                              "--number_of_topics=5", "--training_iterations=1", "--inference_mode=0"]) == 2
    capsys.readouterr()
    rc = launch_test.main(["--input_directory=%s" % corpus_dir, "--model_directory=%s" % runs[0],
                           "--snapshot_index=4"])
    assert rc == 0
    printed = capsys.readouterr().out
    assert "held-out likelihood of snapshot" in printed
    gamma = np.loadtxt(runs[0] / "test-4")
    assert gamma.shape == (20, 5) and np.all(gamma > 0)


def test_nips_k500_trace_and_heldout_likelihood():
    """BASELINE.json cfg 5 (parsed/nips.88-05, K=500, train = first 2,235 documents, test = last 248):
    joint log-likelihood per iteration and held-out words log-likelihood every 10 iterations against
    the reference's own trace (tests/golden/make_golden.py --only nipstrace; as many iterations as the
    committed fixture holds)."""
    from pylda_amd.variational_bayes import VariationalBayes
    g = load_golden("nips_trace_k500.npz")
    K, V = int(g["K"]), len(g["words"])
    n_iter = len(g["joint_ll"])
    np.random.seed(int(g["seed"]))
    m = VariationalBayes()
    m._verbose = False
    m._initialize_parsed(g["doc_ptr"], g["term_id"].astype(np.int32), g["term_ct"].astype(np.int32),
                         V, K, 1.0 / K, 1.0 / V)            # eta: the same seeded draw as variational_bayes.py:95
    test = (g["test_doc_ptr"], g["test_term_id"].astype(np.int32), g["test_term_ct"].astype(np.int32))
    heldout = {int(it): v for it, v in g["heldout"]}
    for it in range(1, n_iter + 1):
        joint = m.learning()
        ref = g["joint_ll"][it - 1]
        assert abs(joint - ref) < 1e-7 * abs(ref), (it, joint, ref)
        if it in heldout:
            wll, gamma = m.e_step(test)
            assert abs(wll - heldout[it]) < 1e-7 * abs(heldout[it]), (it, wll, heldout[it])
            assert gamma.shape == (len(test[0]) - 1, K)
    assert rel_err(m._alpha_alpha, g["alpha_last"]) < 1e-6
    print("nips K=500: %d iterations match; held-out per-token log-likelihood %.6f"
          % (n_iter, (heldout[max(heldout)] / int(g["test_tokens"])) if heldout else float("nan")))


def test_tiny_exports_end_to_end_match_reference_bytes(tiny, tmp_path):
    """text -> _initialize (seed 7, the reference's eta draw) -> two learning() iterations on the
    GPU -> export_beta / export_gamma: the files equal, byte for byte, what the reference wrote after the
    same calls (tests/golden/make_golden.py::make_tiny_exports; `%g` keeps six significant digits)."""
    from pylda_amd.variational_bayes import VariationalBayes
    g = load_golden("tiny_exports.npz")
    np.random.seed(7)
    m = VariationalBayes()
    m._verbose = False
    m._initialize([str(d) for d in tiny["docs"]], [str(w) for w in g["words"]], 2, 0.5, 0.1)
    export_before = tmp_path / "before"
    m.export_gamma(str(export_before))                  # usable before the first learning() (:92's gamma)
    assert len(export_before.read_text().splitlines()) == 3
    m.learning()
    m.learning()
    assert rel_err(m._eta, g["eta"]) < 1e-10 and rel_err(m._gamma, g["gamma"]) < 1e-10
    for name, fn, top in (("exp_beta", m.export_beta, -1), ("exp_beta_top2", m.export_beta, 2),
                          ("exp_gamma", m.export_gamma, -1), ("exp_gamma_top2", m.export_gamma, 2)):
        path = tmp_path / name
        fn(str(path), top)
        assert path.read_bytes() == bytes(g[name]), name


def test_m_step_uses_a_host_gamma_the_caller_assigned(ap_train):
    """m_step reads self._gamma (:232-233) - also when the caller assigned it, or it comes from
    _initialize / a snapshot, instead of from the last training e_step() on the device."""
    from oracle import vb_numpy
    from pylda_amd.variational_bayes import VariationalBayes
    g = ap_train
    ptr = g["doc_ptr"][:201]
    m = VariationalBayes()
    m._verbose = False
    m._initialize_parsed(ptr, g["term_id"][:ptr[-1]], g["term_ct"][:ptr[-1]], 6806, 10, 0.1, 1.0 / 6806,
                         eta=g["eta"].copy())
    sstats = np.random.default_rng(0).gamma(1.0, 1.0, (10, 6806))
    # straight after initialisation: gamma is the constant matrix of :92
    gamma0 = m._gamma.copy()
    assert gamma0.shape == (200, 10) and np.allclose(gamma0, 0.1 + 6806 / 10.0)
    tll, ass = m.m_step(sstats)
    tll_ref, ass_ref, eta_ref = vb_numpy.m_step(g["eta"], m._alpha_beta, sstats, gamma0)
    assert abs(tll - tll_ref) < 1e-10 * abs(tll_ref) and rel_err(ass, ass_ref) < 1e-12
    assert rel_err(m._eta, eta_ref) < 1e-13
    # a gamma the caller assigns
    mine = np.random.default_rng(1).gamma(2.0, 1.0, (200, 10))
    m._gamma = mine
    _, ass = m.m_step(sstats)
    assert rel_err(ass, vb_numpy.m_step(eta_ref, m._alpha_beta, sstats, mine)[1]) < 1e-12
    # after a training e_step the device copy is the one in use again
    m.e_step()
    _, ass = m.m_step(sstats)
    assert rel_err(ass, vb_numpy.m_step(eta_ref, m._alpha_beta, sstats, m._gamma)[1]) < 1e-10


def test_reference_rng_stream_and_progress_lines(ap_train, capsys):
    """Optional side effects of the reference's E-step loop: with _reference_rng_stream the global numpy RNG
    advances exactly as in the reference (one permutation(D) draw per e_step, variational_bayes.py:159), and
    _progress_lines prints its every-1000-documents lines (:209-210)."""
    from pylda_amd.variational_bayes import VariationalBayes
    g = ap_train
    m = VariationalBayes()
    m._verbose = False
    m._initialize_parsed(g["doc_ptr"], g["term_id"], g["term_ct"], 6806, 10, 0.1, 1.0 / 6806, eta=g["eta"].copy())
    np.random.seed(123)
    m.e_step()
    untouched = np.random.random()
    m._reference_rng_stream = True
    m._progress_lines = True
    np.random.seed(123)
    capsys.readouterr()
    m.e_step()
    after = np.random.random()
    np.random.seed(123)
    np.random.permutation(2000)
    assert after == np.random.random() and after != untouched
    assert capsys.readouterr().out.splitlines() == ["successfully processed 1000 documents...",
                                                    "successfully processed 2000 documents..."]


@pytest.mark.parametrize("K,docs,seed", [(1, 50, 0), (10, 2000, 1), (128, 100000, 2), (500, 2235, 3), (1500, 300, 4), (10, 2000, 5)])
def test_alpha_update_on_the_device(ap_train, K, docs, seed):
    """learning()'s alpha update runs on the device (alpha_newton_kernel): the reference's Newton iteration with its
    decaying step and its element-wise 1 / hessian (variational_bayes.py:277-324), against the numpy restatement that
    is pinned to the reference's own update (tests/test_host_logic.py); seed 5 starts from an alpha whose first steps
    are refused (the decay path), K = 10 / docs = 2000 with seed 1 is the associated-press golden itself."""
    from oracle import vb_numpy
    from pylda_amd import _capi
    rng = np.random.default_rng(seed)
    if seed == 1:
        alpha, stats = ap_train["alpha"].copy(), ap_train["alpha_ss"].copy()
    else:
        alpha = rng.uniform(0.01, 2.0, K) if seed != 5 else np.full(K, 1e-3)
        gamma = rng.gamma(0.3 if seed != 5 else 5.0, 1.0, (min(docs, 400), K)) + 1e-3
        stats = np.sum(vb_numpy.compute_dirichlet_expectation(gamma), axis=0) * (docs / gamma.shape[0])
    ctx = _capi.Context(K, 4)
    got = ctx.test_alpha_update(alpha, stats, docs)
    ctx.close()
    with np.errstate(all="ignore"):
        want = vb_numpy.optimize_hyperparameters(alpha, stats, docs)
    if K == 1:      # the reference's vector c is sum_g_h / (1/z + 1/h) with h = -z at K = 1: 0/0, alpha becomes NaN - here too
        assert np.isnan(want).all() and np.isnan(got).all()
        return
    assert np.all(got > 0) and rel_err(got, want) < 1e-10, (got[:4], want[:4])
    if seed == 1:
        assert rel_err(got, ap_train["alpha_after"]) < 1e-10


def test_pinned_pool_is_bounded_by_bytes():
    """the page-locked pool behind e_step() / get_gamma() must not grow with the number of distinct
    shapes a long-running process asks for; very large arrays are not pinned at all."""
    from pylda_amd import _capi
    _capi.load()
    cap, big = _capi._PINNED_POOL_CAP, _capi._PINNED_MAX_ARRAY
    try:
        _capi._PINNED_POOL_CAP, _capi._PINNED_MAX_ARRAY = 4 << 20, 2 << 20
        for n in range(1, 200):                               # 199 different sizes, 8 KB .. 1.6 MB
            a = _capi.pinned_empty((n, 1024))
            a[:] = n
            assert a.sum() == n * n * 1024
            del a
            assert _capi.pinned_pool_bytes() <= _capi._PINNED_POOL_CAP
        assert _capi.pinned_pool_bytes() > 0                  # recent blocks are kept for reuse
        before = _capi.pinned_pool_bytes()
        b = _capi.pinned_empty((300, 1024))                   # 2.4 MB > _PINNED_MAX_ARRAY: ordinary memory
        del b
        assert _capi.pinned_pool_bytes() == before
    finally:
        _capi._PINNED_POOL_CAP, _capi._PINNED_MAX_ARRAY = cap, big
