"""The live-topic E-step (estep_compact.h): the inner loop of variational_bayes.py:174-190 on the topics of a document
whose gamma still differs from alpha, behind a dense prefix of the quad kernel (estep_quad.h).

Same results as the dense kernels to rounding (another summation order), the same iteration counts, the oracle's
per-document values - at every shape of the register tile (term slots per lane x columns), through the shrinking of the
tile, in held-out mode, with the hand-over forced early, and with the exactness guard failing on purpose (the log-space
kernel then redoes the document).  Needs an MI355X."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

# against the dense kernels: rounding of another summation order (measured 3e-12).  Against the oracle gamma gets the bar
# of the suite's randomised sweep: a topic that is still decaying when the iteration cap ends the document carries fifty
# iterations of amplified rounding - the dense kernels differ from the oracle by 1.1e-9 on such an entry, before any of this
GAMMA_RTOL, GAMMA_RTOL_ORACLE, LL_RTOL, SSTATS_ATOL = 1e-9, 1e-7, 1e-9, 1e-8


@pytest.fixture(scope="module")
def capi():
    from pylda_amd import _capi
    return _capi


def topical_corpus(rng, D, V, K, mean_len, true_topics=24, concentration=0.1):
    """Documents drawn from a few topics each, and a model that knows topics with vocabularies of their own: with
    alpha = 1 / K most of a document's K topics die within a dozen iterations."""
    beta = rng.dirichlet(np.full(V, 0.02), size=true_topics)
    ptr, ids, cts = [0], [], []
    for _ in range(D):
        theta = rng.dirichlet(np.full(true_topics, concentration))
        n = max(1, int(rng.poisson(mean_len)))
        words = rng.choice(V, size=n, p=theta @ beta)
        u, c = np.unique(words, return_counts=True)
        ids.append(u.astype(np.int32))
        cts.append(c.astype(np.int32))
        ptr.append(ptr[-1] + len(u))
    eta = rng.gamma(100.0, 0.01, (K, V))
    for k in range(K):
        eta[k] += 40.0 * V * beta[k % true_topics] * rng.uniform(0.2, 1.0)
    return np.array(ptr, np.int64), np.concatenate(ids), np.concatenate(cts), eta


def run(capi, K, V, ptr, ids, cts, alpha, eta, options=(), heldout=False, tol=1e-6, max_iter=50):
    ctx = capi.Context(K, V)
    for name, value in options:
        ctx.set_option(name, value)
    corpus = ctx.corpus(ptr, ids, cts)
    ctx.set_profiling(True)
    ctx.work_counters()
    out = ctx.estep_host(corpus, alpha, eta, max_iter, tol, heldout)
    ctx.work_counters()
    out["tile_entries"], out["handed_over"] = ctx.executed_work()
    out["flagged"] = ctx.estep_results(corpus)[2]
    out["clock_mhz"] = ctx.shader_clock_mhz()
    corpus.close()
    ctx.close()
    return out


@pytest.mark.parametrize("K,mean_len", [(256, 200), (128, 200), (256, 120), (128, 60), (200, 240), (100, 30), (256, 215),
                                        (500, 230), (384, 150), (500, 420), (450, 330)])
def test_live_topic_kernel_matches_dense_kernels_and_oracle(capi, K, mean_len):
    """Every lane shape of the kernel (1 .. 8 term slots per lane by document length; behind the quad kernel with its
    streamed classes, and - K > 256 - behind the fused streaming kernel, from which it gathers its tile itself) against
    the dense kernels on the same E-step and against the C oracle: iteration counts identical, gamma / per-document
    log-likelihood / statistics within the suite's bars (measured: 1e-12)."""
    from oracle import c_oracle
    rng = np.random.default_rng(11 * K + mean_len)
    V, D = 3000, 160
    ptr, ids, cts, eta = topical_corpus(rng, D, V, K, mean_len)
    alpha = np.full(K, 1.0 / K)
    ref = c_oracle.e_step(alpha, eta, ptr, ids, cts)
    dense = run(capi, K, V, ptr, ids, cts, alpha, eta, [("compact", 0)])
    live = run(capi, K, V, ptr, ids, cts, alpha, eta)
    assert dense["handed_over"] == 0 and live["handed_over"] >= (0.8 if mean_len < 300 else 0.3) * D, (live["handed_over"], D)
    assert live["flagged"] == 0 and dense["flagged"] == 0
    assert np.array_equal(live["iters"], ref["iters"]) and np.array_equal(dense["iters"], ref["iters"])
    for other, name in ((dense, "dense kernels"), (ref, "oracle")):
        assert rel_err(live["gamma"], other["gamma"]) < (GAMMA_RTOL if other is dense else GAMMA_RTOL_ORACLE), name
        assert rel_err(live["doc_ll"], other["doc_ll"]) < LL_RTOL, name
        assert np.max(np.abs(live["sstats"] - other["sstats"])) < SSTATS_ATOL, name
    # the work it saved: the executed tile entries are a fraction of the dense kernels'
    assert live["tile_entries"] < (0.7 if mean_len < 300 else 0.95) * dense["tile_entries"]      # (long documents: few columns per wavefront)
    assert live["clock_mhz"] is not None and 500.0 < live["clock_mhz"] < 3000.0


@pytest.mark.parametrize("K,cap", [(256, 8), (256, 12), (128, 17), (256, 24), (128, 4), (200, 32), (256, 40), (128, 56), (256, 33)])
def test_hand_over_at_other_live_counts_and_shrinking_tiles(capi, K, cap):
    """Option compact_cap hands a document over at most at `cap` live topics: above one wavefront's columns the two-wavefront
    body (compact_pair_body) and its hand-down, below them the smaller instantiations of the tile (8, 16, 24 columns),
    entered directly and by shrinking - all must give what the oracle gives."""
    from oracle import c_oracle
    rng = np.random.default_rng(1000 + 7 * K + cap)
    V, D = 2500, 120
    ptr, ids, cts, eta = topical_corpus(rng, D, V, K, 190, concentration=0.02 if cap <= 12 else 0.1 if cap <= 32 else 0.4)
    alpha = np.full(K, 1.0 / K)
    ref = c_oracle.e_step(alpha, eta, ptr, ids, cts)
    for phase in (0, 1):
        out = run(capi, K, V, ptr, ids, cts, alpha, eta, [("compact_cap", cap), ("compact_phase", phase)])
        assert out["handed_over"] > 0 and out["flagged"] == 0
        assert np.array_equal(out["iters"], ref["iters"])
        assert rel_err(out["gamma"], ref["gamma"]) < GAMMA_RTOL_ORACLE and rel_err(out["doc_ll"], ref["doc_ll"]) < LL_RTOL
        assert np.max(np.abs(out["sstats"] - ref["sstats"])) < SSTATS_ATOL


@pytest.mark.parametrize("K", [256, 128])
def test_heldout_mode_iteration_cap_and_thresholds(capi, K):
    """:202-204 (words log-likelihood, gamma returned, statistics untouched) through the live-topic kernel; a small
    iteration cap that ends documents inside it; a loose threshold that stops them early."""
    from oracle import c_oracle
    rng = np.random.default_rng(77 + K)
    V, D = 2500, 100
    ptr, ids, cts, eta = topical_corpus(rng, D, V, K, 180)
    alpha = np.full(K, 1.0 / K)
    for heldout, tol, cap in ((True, 1e-6, 50), (False, 1e-3, 50), (False, 1e-6, 17), (True, 1e-4, 23)):
        ref = c_oracle.e_step(alpha, eta, ptr, ids, cts, cap, tol, heldout=heldout)
        out = run(capi, K, V, ptr, ids, cts, alpha, eta, heldout=heldout, tol=tol, max_iter=cap)
        assert out["handed_over"] > 0
        assert np.array_equal(out["iters"], ref["iters"]), (heldout, tol, cap)
        assert rel_err(out["gamma"], ref["gamma"]) < GAMMA_RTOL_ORACLE
        if heldout:
            assert rel_err(out["doc_words_ll"], ref["doc_words_ll"]) < LL_RTOL
        else:
            assert rel_err(out["doc_ll"], ref["doc_ll"]) < LL_RTOL
            assert np.max(np.abs(out["sstats"] - ref["sstats"])) < SSTATS_ATOL


def test_exactness_guard_failure_goes_to_the_log_space_kernel(capi):
    """The guard of estep_compact.h (dead topics add < 2^-60 to a normaliser and cannot come back to life) is evaluated
    per document from live quantities; option compact_guard_fail makes it fail for every document: they are flagged and
    redone by the log-space kernel - the reference's own formulation - like a document whose normaliser left the range."""
    from oracle import c_oracle
    rng = np.random.default_rng(5)
    K, V, D = 256, 2000, 60
    ptr, ids, cts, eta = topical_corpus(rng, D, V, K, 150)
    alpha = np.full(K, 1.0 / K)
    ref = c_oracle.e_step(alpha, eta, ptr, ids, cts)
    out = run(capi, K, V, ptr, ids, cts, alpha, eta, [("compact_guard_fail", 1)])
    assert out["handed_over"] > 0 and out["flagged"] == out["handed_over"]
    assert np.array_equal(out["iters"], ref["iters"])
    assert rel_err(out["gamma"], ref["gamma"]) < 1e-8 and rel_err(out["doc_ll"], ref["doc_ll"]) < 1e-8
    assert np.max(np.abs(out["sstats"] - ref["sstats"])) < SSTATS_ATOL
    # a few topics with a large alpha (the alpha update of a trained model) never die - they stay columns to the end -
    # and must not trip the guard of the documents around them: the bound is over the DEAD topics' alpha
    alpha = np.full(K, 1.0 / K)
    alpha[rng.permutation(K)[:6]] = [0.08, 0.2, 0.05, 0.5, 0.03, 0.1]
    ref = c_oracle.e_step(alpha, eta, ptr, ids, cts)
    out = run(capi, K, V, ptr, ids, cts, alpha, eta)
    assert out["handed_over"] > 0 and out["flagged"] == 0
    assert np.array_equal(out["iters"], ref["iters"]) and rel_err(out["doc_ll"], ref["doc_ll"]) < LL_RTOL
    assert rel_err(out["gamma"], ref["gamma"]) < GAMMA_RTOL_ORACLE and np.max(np.abs(out["sstats"] - ref["sstats"])) < SSTATS_ATOL
    # ... and alpha too large for any topic to die bitwise: nothing is handed over, nothing flagged
    alpha = np.full(K, 0.3)
    ref = c_oracle.e_step(alpha, eta, ptr, ids, cts)
    out = run(capi, K, V, ptr, ids, cts, alpha, eta)
    assert out["handed_over"] == 0 and out["flagged"] == 0
    assert np.array_equal(out["iters"], ref["iters"]) and rel_err(out["doc_ll"], ref["doc_ll"]) < LL_RTOL


def test_learning_trace_with_and_without_the_live_topic_kernel(capi):
    """Five learning() iterations (device M-step, alpha Newton update) with the hand-over on and off: the same joint
    log-likelihood trace and the same alpha to 1e-10 - the model the bench trains is the model the dense kernels train."""
    from pylda_amd.variational_bayes import VariationalBayes
    rng = np.random.default_rng(3)
    K, V, D = 128, 3000, 400
    ptr, ids, cts, eta = topical_corpus(rng, D, V, K, 150)
    traces = {}
    for mode in (0, 1):
        vb = VariationalBayes()
        vb._verbose = False
        vb._initialize_parsed(ptr, ids, cts, V, K, 1.0 / K, 1.0 / V, eta=eta.copy())
        vb._context().set_option("compact", mode)
        traces[mode] = ([vb.learning() for _ in range(5)], vb._alpha_alpha.copy())
        vb._train_corpus.close()
        vb._ctx.close()
    assert rel_err(np.array(traces[1][0]), np.array(traces[0][0])) < 1e-11
    assert rel_err(traces[1][1], traces[0][1]) < 1e-10


def test_alpha_grown_past_the_mortality_bound_switches_the_hand_over_off_and_back(capi):
    """The alpha update of a training run pushes alpha_k past the bound beyond which a topic never counts as dead
    (kMortalT); with as many such topics as the widest tile has columns no document can be handed over: the corpus then
    runs as with compact = 0 - its postings go back to the row layout (the walk over lists would add whole rows per
    posting) - and returns to the lists when alpha shrinks again.  Same corpus object throughout; the oracle's values in
    every state."""
    from oracle import c_oracle
    rng = np.random.default_rng(21)
    K, V, D = 128, 2500, 150
    ptr, ids, cts, eta = topical_corpus(rng, D, V, K, 170)
    ctx = capi.Context(K, V)
    corpus = ctx.corpus(ptr, ids, cts)
    ctx.set_profiling(True)
    small = np.full(K, 1.0 / K)
    grown = small.copy()
    grown[rng.permutation(K)[:70]] = rng.uniform(0.03, 0.3, 70)        # 70 topics that never die: more than any tile holds
    few = small.copy()
    few[rng.permutation(K)[:5]] = rng.uniform(0.03, 0.3, 5)            # five: they are columns of every tile, the rest dies
    for alpha, hands_over in ((small, True), (grown, False), (grown, False), (few, True), (small, True)):
        ref = c_oracle.e_step(alpha, eta, ptr, ids, cts)
        ctx.work_counters()
        out = ctx.estep_host(corpus, alpha, eta)
        ctx.work_counters()
        handed = ctx.executed_work()[1]
        assert (handed > 0.5 * D) if hands_over else handed == 0, handed
        assert corpus.layout("gather_live") == (1 if hands_over else 0)
        assert corpus.layout("live_off_by_alpha") == (0 if hands_over else 1)
        assert ctx.estep_results(corpus)[2] == 0
        assert np.array_equal(out["iters"], ref["iters"])
        assert rel_err(out["gamma"], ref["gamma"]) < GAMMA_RTOL_ORACLE and rel_err(out["doc_ll"], ref["doc_ll"]) < LL_RTOL
        assert np.max(np.abs(out["sstats"] - ref["sstats"])) < SSTATS_ATOL
    corpus.close()
    ctx.close()
