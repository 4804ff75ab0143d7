"""Parity of the HIP E-step (through the C ABI) with the oracle and with the
golden vectors produced by the reference itself.  Needs an MI355X.

Tolerances: the bar BASELINE.json states is 1e-5 relative on the per-document
log-likelihood; fp64 end to end lets these tests hold 1e-9 or better.
"""
import numpy as np
import pytest

from conftest import csr_slice, load_golden, rel_err

pytestmark = pytest.mark.gpu

LL_RTOL = 1e-9        # per-document log-likelihood, relative (bar: 1e-5)
LL_ATOL = 1e-11       # ... plus an absolute floor: K=1 log-likelihoods are analytically 0
GAMMA_RTOL = 1e-9
SSTATS_ATOL = 1e-8


@pytest.fixture(scope="module")
def capi():
    from pylda_amd import _capi
    _capi.load()
    assert _capi.device_count() >= 1, "no HIP device visible"
    return _capi


def run(capi, alpha, eta, ptr, tid, tct, heldout=False, max_iter=50, tol=1e-6, options=()):
    K, V = eta.shape
    ctx = capi.Context(K, V)
    for name, value in options:
        ctx.set_option(name, value)
    corpus = ctx.corpus(ptr, tid, tct)
    out = ctx.estep_host(corpus, alpha, eta, max_iter, tol, heldout)
    out["logspace_docs"] = ctx.estep_results(corpus)[2]
    corpus.close()
    ctx.close()
    return out


def check_against(out, ref_gamma, ref_ll, ref_iters, ll_key="doc_ll", min_same=1.0):
    same = out["iters"] == ref_iters
    # the stop test (:187-189) must fall on the same inner iteration as the reference's for EVERY document of the
    # fixed-seed cases (tools/fuzz_parity.py: no flip in 90 000 random documents either); the first one that does
    # not is named
    if not same.all():
        d = int(np.nonzero(~same)[0][0])
        print("inner-iteration counts differ on %d of %d documents; first: document %d ran %d, reference %d"
              % ((~same).sum(), same.size, d, out["iters"][d], ref_iters[d]))
    assert np.mean(same) >= min_same, "inner-iteration counts differ on %d documents" % (~same).sum()
    assert rel_err(out["gamma"][same], ref_gamma[same]) < GAMMA_RTOL
    assert np.all(np.abs(out[ll_key][same] - ref_ll[same]) <= LL_RTOL * np.abs(ref_ll[same]) + LL_ATOL)
    # documents that stop one iteration apart sit on the threshold: still within the 1e-5 bar
    if (~same).any():
        assert rel_err(out[ll_key][~same], ref_ll[~same]) < 1e-5


def test_device_special_functions(capi):
    g = load_golden("special_fn.npz")
    ctx = capi.Context(2, 2)
    dg, lg = ctx.test_special(g["x"])
    ctx.close()
    assert np.max(np.abs(dg - g["psi"]) / np.maximum(1.0, np.abs(g["psi"]))) < 5e-15
    assert np.max(np.abs(lg - g["gammaln"]) / np.maximum(1.0, np.abs(g["gammaln"]))) < 5e-14


def test_device_fused_exp_digamma(capi):
    """t[k] = exp(psi(gamma_k) - c): the fused, log-free form against scipy over the range gamma takes."""
    g = load_golden("special_fn.npz")
    x = g["x"][(g["x"] > 1e-4) & (g["x"] < 1e5)]
    ctx = capi.Context(2, 2)
    for c in (0.0, 3.5, -1.25, 9.0):
        got = ctx.test_expdigamma(x, c)
        want = np.exp(g["psi"][(g["x"] > 1e-4) & (g["x"] < 1e5)] - c)
        ok = want > 1e-290
        psi = g["psi"][(g["x"] > 1e-4) & (g["x"] < 1e5)]
        # the exponent psi - c carries ~1 ulp of ITS magnitude: |psi| ~ 1e3 at x ~ 1e-3 costs 3 digits
        assert np.all(np.abs(got[ok] - want[ok]) / want[ok] < 2e-15 * (4.0 + np.abs(psi[ok] - c)))
    # arguments far below anything gamma takes in practice (gamma_k >= alpha_k, and pylda_set_alpha accepts any
    # positive alpha): exp(psi(x) - c) = exp(-1/x - ...) is a clean 0 in BOTH fused forms, never inf / NaN
    tiny_x = np.array([1e-300, 1e-250, 1e-170, 1e-100, 1e-60, 1e-40, 1e-20, 1e-10, 1e-5])
    for c in (0.0, 4.0, 2000.0 + 3.0):          # (c > 1e3 selects the level-ordered form the kernels' gamma phase uses)
        got = ctx.test_expdigamma(tiny_x, c)
        assert np.array_equal(got, np.zeros_like(tiny_x)), (c, got)
    ctx.close()


def test_tiny_training_and_heldout(capi, tiny):
    t = tiny
    out = run(capi, t["alpha"], t["eta"], t["doc_ptr"], t["term_id"], t["term_ct"])
    assert np.array_equal(out["iters"], t["iters"])
    assert rel_err(out["gamma"], t["gamma"]) < 1e-12
    assert rel_err(out["doc_ll"], t["doc_ll"]) < 1e-11
    assert np.max(np.abs(out["sstats"] - t["sstats"])) < 1e-12
    assert abs(out["document_log_likelihood"] - float(t["corpus_ll"])) < 1e-11
    held = run(capi, t["alpha"], t["eta"], t["doc_ptr"], t["term_id"], t["term_ct"], heldout=True)
    assert np.array_equal(held["iters"], t["heldout_iters"])
    assert rel_err(held["gamma"], t["heldout_gamma"]) < 1e-12
    assert rel_err(held["doc_words_ll"], t["heldout_words_ll"]) < 1e-11
    assert held["sstats"] is None


def test_ap_train_k10_matches_reference_goldens(capi, ap_train):
    g = ap_train
    out = run(capi, g["alpha"], g["eta"], g["doc_ptr"], g["term_id"], g["term_ct"])
    assert out["logspace_docs"] == 0
    check_against(out, g["gamma"], g["doc_ll"], g["iters"])
    assert np.max(np.abs(out["sstats"] - g["sstats"])) < SSTATS_ATOL
    assert abs(out["sstats"].sum() - g["term_ct"].sum()) < 1e-6          # tokens conserved
    assert abs(out["document_log_likelihood"] - float(g["corpus_ll"])) < 1e-9 * abs(float(g["corpus_ll"]))
    # the headline number: worst per-document relative log-likelihood delta
    worst = rel_err(out["doc_ll"], g["doc_ll"])
    print("AP K=10 max per-document relative LL delta: %.3e" % worst)
    assert worst < 1e-5


def test_ap_heldout_k10_matches_reference_goldens(capi, ap_test):
    g = ap_test
    out = run(capi, g["alpha"], g["eta"], g["doc_ptr"], g["term_id"], g["term_ct"], heldout=True)
    check_against(out, g["gamma"], g["words_ll"], g["iters"], ll_key="doc_words_ll")
    assert abs(out["words_log_likelihood"] - float(g["corpus_words_ll"])) < 1e-9 * abs(float(g["corpus_words_ll"]))


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 6, 9, 10, 11])
def test_every_kernel_variant_agrees(capi, ap_train, variant):
    g = ap_train
    docs = list(range(0, 2000, 10))
    ptr, tid, tct = csr_slice(g["doc_ptr"], g["term_id"], g["term_ct"], docs)
    out = run(capi, g["alpha"], g["eta"], ptr, tid, tct, options=[("force_variant", variant)])
    check_against(out, g["gamma"][docs], g["doc_ll"][docs], g["iters"][docs])


def test_logspace_safety_net_kernel_matches(capi, ap_train, ap_test):
    g = ap_train
    docs = list(range(0, 2000, 20))
    ptr, tid, tct = csr_slice(g["doc_ptr"], g["term_id"], g["term_ct"], docs)
    out = run(capi, g["alpha"], g["eta"], ptr, tid, tct, options=[("force_logspace", 1)])
    assert out["logspace_docs"] == len(docs)
    check_against(out, g["gamma"][docs], g["doc_ll"][docs], g["iters"][docs])
    from oracle import c_oracle
    ref = c_oracle.e_step(g["alpha"], g["eta"], ptr, tid, tct)
    assert np.max(np.abs(out["sstats"] - ref["sstats"])) < SSTATS_ATOL
    h = ap_test
    held = run(capi, h["alpha"], h["eta"], h["doc_ptr"], h["term_id"], h["term_ct"], heldout=True,
               options=[("force_logspace", 1)])
    check_against(held, h["gamma"], h["words_ll"], h["iters"], ll_key="doc_words_ll")


def test_collapsed_alpha_triggers_safety_net(capi):
    """alpha_k ~ 1e-4 makes exp(psi(gamma_k) - max psi) underflow for topics that own
    words: the fast kernel must flag those documents and the log-space kernel finish them."""
    from oracle import c_oracle
    rng = np.random.default_rng(5)
    K, V, D = 4, 40, 12
    eta = np.full((K, V), 1e-3)
    for k in range(K):
        eta[k, k * 10:(k + 1) * 10] = 50.0            # disjoint topics
    alpha = np.array([1e-4, 1e-4, 1e-4, 5.0])
    ptr, ids, cts = [0], [], []
    for d in range(D):
        k = d % 3                                      # words only from the collapsed topics
        w = rng.choice(np.arange(k * 10, (k + 1) * 10), size=3, replace=False)
        ids += list(w)
        cts += [1, 1, 1]
        ptr.append(len(ids))
    ptr, ids, cts = np.array(ptr), np.array(ids, np.int32), np.array(cts, np.int32)
    ref = c_oracle.e_step(alpha, eta, ptr, ids, cts, max_iter=3)
    out = run(capi, alpha, eta, ptr, ids, cts, max_iter=3)
    assert np.all(np.isfinite(out["gamma"])) and np.all(np.isfinite(out["doc_ll"]))
    assert rel_err(out["gamma"], ref["gamma"]) < 1e-9
    assert rel_err(out["doc_ll"], ref["doc_ll"]) < 1e-9
    assert np.max(np.abs(out["sstats"] - ref["sstats"])) < 1e-10


def random_corpus(rng, D, V, mean_len, zipf=1.1):
    p = 1.0 / np.arange(1, V + 1) ** zipf
    p /= p.sum()
    ptr, ids, cts = [0], [], []
    for _ in range(D):
        n = max(1, rng.poisson(mean_len))
        w = rng.choice(V, size=n, p=p)
        u, c = np.unique(w, return_counts=True)
        ids.append(u)
        cts.append(c)
        ptr.append(ptr[-1] + u.size)
    return np.array(ptr, np.int64), np.concatenate(ids).astype(np.int32), np.concatenate(cts).astype(np.int32)


@pytest.mark.parametrize("K,V,D,mean_len", [(128, 2000, 48, 200), (64, 500, 64, 40), (500, 800, 12, 300),
                                            (3, 50, 100, 5), (256, 3000, 16, 250), (1, 20, 5, 10),
                                            (256, 4000, 40, 420), (200, 2500, 24, 700), (130, 3000, 30, 1200),
                                            (192, 3000, 24, 300), (150, 2500, 20, 170), (256, 3000, 12, 100)])
def test_random_corpora_against_c_oracle(capi, K, V, D, mean_len):
    from oracle import c_oracle
    rng = np.random.default_rng(K * 1000 + V)
    ptr, ids, cts = random_corpus(rng, D, V, mean_len)
    eta = rng.gamma(100.0, 0.01, (K, V))
    eta[:, rng.choice(V, V // 3, replace=False)] = 1.0 / V          # rows that look "unseen"
    alpha = rng.uniform(0.05, 1.5, K)
    ref = c_oracle.e_step(alpha, eta, ptr, ids, cts)
    out = run(capi, alpha, eta, ptr, ids, cts)
    check_against(out, ref["gamma"], ref["doc_ll"], ref["iters"])
    assert np.max(np.abs(out["sstats"] - ref["sstats"])) < SSTATS_ATOL
    assert abs(out["sstats"].sum() - cts.sum()) < 1e-7 * cts.sum()
    held_ref = c_oracle.e_step(alpha, eta, ptr, ids, cts, heldout=True)
    held = run(capi, alpha, eta, ptr, ids, cts, heldout=True)
    check_against(held, held_ref["gamma"], held_ref["doc_words_ll"], held_ref["iters"],
                  ll_key="doc_words_ll")


@pytest.mark.parametrize("K,V,mean_len", [(10, 300, 20), (16, 300, 150), (17, 400, 90), (32, 500, 260),
                                          (50, 600, 120), (64, 700, 330), (100, 900, 200), (128, 900, 150),
                                          (128, 1200, 230), (128, 1200, 40), (128, 1500, 185), (110, 1500, 200)])
def test_register_resident_slab_kernels(capi, K, V, mean_len):
    """The slab kernels (tile in VGPRs) over every (wavefronts, slab width, words per lane) geometry,
    against the C oracle and against the generic LDS kernel."""
    from oracle import c_oracle
    rng = np.random.default_rng(K * 7 + mean_len)
    ptr, ids, cts = random_corpus(rng, 40, V, mean_len, zipf=0.8)
    eta = rng.gamma(100.0, 0.01, (K, V))
    eta[:, rng.choice(V, V // 4, replace=False)] = 1.0 / V
    alpha = rng.uniform(0.05, 1.5, K)
    ref = c_oracle.e_step(alpha, eta, ptr, ids, cts)
    gen = run(capi, alpha, eta, ptr, ids, cts, options=[("force_variant", 1)])
    held_ref = c_oracle.e_step(alpha, eta, ptr, ids, cts, heldout=True)
    for variant in (4, 6, 9, 10):          # slab, quilt (register tiles), group-fused streaming, quad
        out = run(capi, alpha, eta, ptr, ids, cts, options=[("force_variant", variant)])
        check_against(out, ref["gamma"], ref["doc_ll"], ref["iters"])
        assert np.max(np.abs(out["sstats"] - ref["sstats"])) < SSTATS_ATOL
        assert np.array_equal(out["iters"], gen["iters"])
        assert rel_err(out["gamma"], gen["gamma"]) < 1e-9
        held = run(capi, alpha, eta, ptr, ids, cts, heldout=True, options=[("force_variant", variant)])
        check_against(held, held_ref["gamma"], held_ref["doc_words_ll"], held_ref["iters"],
                      ll_key="doc_words_ll")
        # bitwise reproducible: fixed summation order (and an order-independent convergence sum)
        again = run(capi, alpha, eta, ptr, ids, cts, options=[("force_variant", variant)])
        assert np.array_equal(out["gamma"], again["gamma"]) and np.array_equal(out["sstats"], again["sstats"])
        assert np.array_equal(out["doc_ll"], again["doc_ll"])


@pytest.mark.parametrize("K,V,mean_len", [(256, 2500, 190), (256, 2500, 120), (200, 2500, 175), (129, 2000, 60)])
def test_wide_table_kernels_agree(capi, K, V, mean_len):
    """128 < K <= 256 (table stride 256): the 8-wavefront quad kernel (all words on chip) and the group-fused streaming
    kernel against the C oracle and the generic kernel, training and held-out, bitwise repeatable."""
    from oracle import c_oracle
    rng = np.random.default_rng(K * 11 + mean_len)
    ptr, ids, cts = random_corpus(rng, 36, V, mean_len, zipf=0.8)
    eta = rng.gamma(100.0, 0.01, (K, V))
    eta[:, rng.choice(V, V // 4, replace=False)] = 1.0 / V
    alpha = rng.uniform(0.05, 1.5, K)
    ref = c_oracle.e_step(alpha, eta, ptr, ids, cts)
    gen = run(capi, alpha, eta, ptr, ids, cts, options=[("force_variant", 1)])
    held_ref = c_oracle.e_step(alpha, eta, ptr, ids, cts, heldout=True)
    for variant in (10, 9):
        out = run(capi, alpha, eta, ptr, ids, cts, options=[("force_variant", variant)])
        check_against(out, ref["gamma"], ref["doc_ll"], ref["iters"])
        assert np.max(np.abs(out["sstats"] - ref["sstats"])) < SSTATS_ATOL
        assert np.array_equal(out["iters"], gen["iters"])
        assert rel_err(out["gamma"], gen["gamma"]) < 1e-9
        held = run(capi, alpha, eta, ptr, ids, cts, heldout=True, options=[("force_variant", variant)])
        check_against(held, held_ref["gamma"], held_ref["doc_words_ll"], held_ref["iters"],
                      ll_key="doc_words_ll")
        again = run(capi, alpha, eta, ptr, ids, cts, options=[("force_variant", variant)])
        assert np.array_equal(out["gamma"], again["gamma"]) and np.array_equal(out["sstats"], again["sstats"])
        assert np.array_equal(out["doc_ll"], again["doc_ll"])


def corpus_of_lengths(rng, V, lengths, zipf=0.8):
    """Documents with exactly the given numbers of distinct terms."""
    p = 1.0 / np.arange(1, V + 1) ** zipf
    p /= p.sum()
    ptr, ids, cts = [0], [], []
    for n in lengths:
        u = np.sort(rng.choice(V, size=n, replace=False, p=p))
        ids.append(u)
        cts.append(rng.integers(1, 5, size=n))
        ptr.append(ptr[-1] + n)
    return np.array(ptr, np.int64), np.concatenate(ids).astype(np.int32), np.concatenate(cts).astype(np.int32)


@pytest.mark.parametrize("K", [256, 128, 200, 100])
def test_quad_kernel_with_streamed_slots(capi, K):
    """Documents of 225-256 distinct terms at table strides 128 / 256: the quad kernel with three / four word slots
    streamed from the table (estep_quad.h, SWL), every boundary length, against the C oracle - equal iteration
    counts, training and held-out, bitwise repeatable - and against the plan without the streamed classes."""
    from oracle import c_oracle
    rng = np.random.default_rng(K + 5)
    V = 3000
    lengths = [224, 225, 226, 231, 232, 233, 239, 240, 241, 242, 247, 248, 249, 254, 255, 256, 257, 230, 236, 250]
    ptr, ids, cts = corpus_of_lengths(rng, V, lengths)
    eta = rng.gamma(100.0, 0.01, (K, V))
    eta[:, rng.choice(V, V // 4, replace=False)] = 1.0 / V
    alpha = rng.uniform(0.05, 1.5, K)
    ctx = capi.Context(K, V)
    corpus = ctx.corpus(ptr, ids, cts)
    plan = {(c["kernel"], c["geometry"]): c["documents"] for c in corpus.plan()}
    short, long_ = (2160904, 3160904) if K <= 128 else (3320804, 4320804)     # SWL, TL, RWL, TWL
    assert plan[("quad", short)] == 9 and plan[("quad", long_)] == 9, plan
    corpus.close()
    ctx.close()
    ref = c_oracle.e_step(alpha, eta, ptr, ids, cts)
    out = run(capi, alpha, eta, ptr, ids, cts)
    check_against(out, ref["gamma"], ref["doc_ll"], ref["iters"])
    assert np.max(np.abs(out["sstats"] - ref["sstats"])) < SSTATS_ATOL
    old = run(capi, alpha, eta, ptr, ids, cts, options=[("quad_stream", 0)])
    assert np.array_equal(out["iters"], old["iters"])
    assert rel_err(out["gamma"], old["gamma"]) < 1e-9
    held_ref = c_oracle.e_step(alpha, eta, ptr, ids, cts, heldout=True)
    held = run(capi, alpha, eta, ptr, ids, cts, heldout=True)
    check_against(held, held_ref["gamma"], held_ref["doc_words_ll"], held_ref["iters"], ll_key="doc_words_ll")
    again = run(capi, alpha, eta, ptr, ids, cts)
    assert np.array_equal(out["gamma"], again["gamma"]) and np.array_equal(out["sstats"], again["sstats"])
    assert np.array_equal(out["doc_ll"], again["doc_ll"])
    for mi, tol in [(1, 1e-6), (2, 1e-6), (50, 1e-2)]:
        ref = c_oracle.e_step(alpha, eta, ptr, ids, cts, max_iter=mi, tol=tol)
        out = run(capi, alpha, eta, ptr, ids, cts, max_iter=mi, tol=tol)
        assert np.array_equal(out["iters"], ref["iters"]), (mi, tol)
        assert rel_err(out["gamma"], ref["gamma"]) < 1e-9
    # the fast path (doc_values = 0) leaves through another exit of the kernel
    ctx = capi.Context(K, V)
    corpus = ctx.corpus(ptr, ids, cts)
    ctx.set_alpha(alpha)
    ctx.set_eta(eta)
    ctx.set_option("doc_values", 1)
    ctx.estep(corpus)
    full_ll = ctx.estep_results(corpus)[0]
    sst_full = ctx.get_sstats()
    ctx.set_option("doc_values", 0)
    ctx.estep(corpus)
    assert abs(ctx.estep_results(corpus)[0] - full_ll) < 1e-11 * abs(full_ll)
    assert np.array_equal(ctx.get_sstats(), sst_full)
    corpus.close()
    ctx.close()


@pytest.mark.parametrize("K", [32, 64, 128, 256, 20])
def test_long_documents_beyond_the_register_kernels(capi, K):
    """Documents of 260-1030 distinct terms: the group-fused streaming kernel (estep_qgroup.h; table strides 64 / 128 /
    256, up to 1024 terms) and, beyond it or at narrower strides, the generic kernels up to the LDS limit (a tile that
    just fits: 540 terms at K = 32) and past it - against the C oracle, training and held-out."""
    from oracle import c_oracle
    rng = np.random.default_rng(K)
    V = 4000
    lengths = [260, 300, 385, 390, 512, 530, 540, 544, 545, 560, 700, 1000, 1023, 1024, 1025, 1030]
    ptr, ids, cts = corpus_of_lengths(rng, V, lengths)
    eta = rng.gamma(100.0, 0.01, (K, V))
    eta[:, rng.choice(V, V // 4, replace=False)] = 1.0 / V
    alpha = rng.uniform(0.05, 1.5, K)
    ctx = capi.Context(K, V)
    corpus = ctx.corpus(ptr, ids, cts)
    plan = {c["kernel"]: c["documents"] for c in corpus.plan()}
    corpus.close()
    ctx.close()
    if K >= 64:
        assert plan.get("qgroup", 0) >= 12, plan
    ref = c_oracle.e_step(alpha, eta, ptr, ids, cts)
    out = run(capi, alpha, eta, ptr, ids, cts)
    check_against(out, ref["gamma"], ref["doc_ll"], ref["iters"])
    assert np.max(np.abs(out["sstats"] - ref["sstats"])) < SSTATS_ATOL
    held_ref = c_oracle.e_step(alpha, eta, ptr, ids, cts, heldout=True)
    held = run(capi, alpha, eta, ptr, ids, cts, heldout=True)
    check_against(held, held_ref["gamma"], held_ref["doc_words_ll"], held_ref["iters"], ll_key="doc_words_ll")
    again = run(capi, alpha, eta, ptr, ids, cts)
    assert np.array_equal(out["gamma"], again["gamma"]) and np.array_equal(out["sstats"], again["sstats"])


@pytest.mark.parametrize("K,V,mean_len", [(500, 1500, 230), (512, 1200, 60), (449, 1500, 700), (480, 1000, 100),
                                          (300, 1500, 210), (384, 1200, 90), (257, 1500, 500), (385, 900, 150)])
def test_fused_streaming_kernel_agrees(capi, K, V, mean_len):
    """256 < K <= 512 (table stride 384 / 512): the fused single-pass streaming kernel (registers + LDS rows + four
    row buffers in flight) against the C oracle and the generic kernel;
    documents from a handful of terms (no streamed slots) to several hundred (many trips of the row pipeline)."""
    from oracle import c_oracle
    rng = np.random.default_rng(K * 13 + mean_len)
    ptr, ids, cts = random_corpus(rng, 28, V, mean_len, zipf=0.8)
    eta = rng.gamma(100.0, 0.01, (K, V))
    eta[:, rng.choice(V, V // 4, replace=False)] = 1.0 / V
    alpha = rng.uniform(0.05, 1.5, K)
    ref = c_oracle.e_step(alpha, eta, ptr, ids, cts)
    gen = run(capi, alpha, eta, ptr, ids, cts, options=[("force_variant", 3)])
    held_ref = c_oracle.e_step(alpha, eta, ptr, ids, cts, heldout=True)
    for variant in (11,):
        out = run(capi, alpha, eta, ptr, ids, cts, options=[("force_variant", variant)])
        check_against(out, ref["gamma"], ref["doc_ll"], ref["iters"])
        assert np.max(np.abs(out["sstats"] - ref["sstats"])) < SSTATS_ATOL
        assert np.array_equal(out["iters"], gen["iters"])
        held = run(capi, alpha, eta, ptr, ids, cts, heldout=True, options=[("force_variant", variant)])
        check_against(held, held_ref["gamma"], held_ref["doc_words_ll"], held_ref["iters"],
                      ll_key="doc_words_ll")
        again = run(capi, alpha, eta, ptr, ids, cts, options=[("force_variant", variant)])
        assert np.array_equal(out["gamma"], again["gamma"]) and np.array_equal(out["sstats"], again["sstats"])
        assert np.array_equal(out["doc_ll"], again["doc_ll"])
    for mi, tol in [(1, 1e-6), (3, 1e-6), (50, 1e-2)]:
        ref = c_oracle.e_step(alpha, eta, ptr, ids, cts, max_iter=mi, tol=tol)
        out = run(capi, alpha, eta, ptr, ids, cts, max_iter=mi, tol=tol)
        assert np.array_equal(out["iters"], ref["iters"]), (mi, tol)
        assert rel_err(out["gamma"], ref["gamma"]) < 1e-9


@pytest.mark.parametrize("variant", [1, 3, 4, 6, 9, 10])
def test_training_fast_path_corpus_likelihood(capi, ap_train, variant):
    """Option doc_values=0 (what learning() uses): the corpus-level document_log_likelihood must equal
    the sum of the complete per-document values, and the reference's own corpus value."""
    g = ap_train
    rng = np.random.default_rng(variant)
    for K, eta, alpha, ptr, tid, tct, ref_ll in (
            (10, g["eta"], g["alpha"], g["doc_ptr"], g["term_id"], g["term_ct"], float(g["corpus_ll"])),
            (100, None, None, g["doc_ptr"][:301], g["term_id"], g["term_ct"], None)):
        if eta is None:
            eta = rng.gamma(100.0, 0.01, (K, 6806))
            alpha = rng.uniform(0.05, 1.0, K)
            tid, tct = tid[:ptr[-1]], tct[:ptr[-1]]
        ctx = capi.Context(K, 6806)
        ctx.set_option("force_variant", variant)
        corpus = ctx.corpus(ptr, tid, tct)
        ctx.set_alpha(alpha)
        ctx.set_eta(eta)
        ctx.set_option("doc_values", 1)
        ctx.estep(corpus)
        full_ll = ctx.estep_results(corpus)[0]
        doc_ll = ctx.get_doc_values(corpus)[0]
        sst_full = ctx.get_sstats()
        ctx.set_option("doc_values", 0)
        ctx.estep(corpus)
        fast_ll = ctx.estep_results(corpus)[0]
        assert abs(fast_ll - full_ll) < 1e-11 * abs(full_ll)
        assert abs(doc_ll.sum() - full_ll) < 1e-11 * abs(full_ll)
        assert np.array_equal(ctx.get_sstats(), sst_full)
        if ref_ll is not None:
            assert abs(fast_ll - ref_ll) < 1e-9 * abs(ref_ll)
        with pytest.raises(capi.PyldaError) as e:
            ctx.get_doc_values(corpus)
        assert e.value.status == -4
        corpus.close()
        ctx.close()


@pytest.mark.parametrize("K,alpha0", [(100, 6e-4), (256, 6e-4), (400, 0.3), (128, 0.02)])
def test_document_terms_pass_with_underflowing_topics(capi, ap_train, K, alpha0):
    """Training fast path: the register kernels leave variational_bayes.py:195-199 to doc_terms_kernel, which
    rebuilds log t_k from the stored t_k.  With alpha ~ 6e-4 the topics a document does not use end at
    t_k = exp(psi(alpha) - psi(sum gamma)) = 0 and gamma_k - alpha_k = 0 exactly: their term must drop out,
    not turn into 0 * log 0.  (K = 400: the fused streaming kernel's exit.)"""
    g = ap_train
    rng = np.random.default_rng(K)
    ptr = g["doc_ptr"][:201]
    tid, tct = g["term_id"][:ptr[-1]], g["term_ct"][:ptr[-1]]
    eta = rng.gamma(100.0, 0.01, (K, 6806))
    eta[rng.choice(K, K // 2, replace=False)] *= 1e-3          # topics nobody wants
    alpha = np.full(K, alpha0)
    ctx = capi.Context(K, 6806)
    corpus = ctx.corpus(ptr, tid, tct)
    ctx.set_alpha(alpha)
    ctx.set_eta(eta)
    ctx.set_option("doc_values", 1)
    ctx.estep(corpus)
    full_ll, _, nlog = ctx.estep_results(corpus)
    gamma_full = ctx.get_gamma(corpus)
    ctx.set_option("doc_values", 0)
    ctx.estep(corpus)
    fast_ll, _, nlog_fast = ctx.estep_results(corpus)
    print("K=%d alpha=%g: kernels %s, %d documents through the log-space safety net, topics at gamma == alpha: %d"
          % (K, alpha0, sorted({c["kernel"] for c in corpus.plan()}), nlog, int((gamma_full == alpha0).sum())))
    assert np.isfinite(fast_ll) and nlog == nlog_fast
    assert abs(fast_ll - full_ll) < 1e-11 * abs(full_ll)
    assert np.array_equal(ctx.get_gamma(corpus), gamma_full)
    corpus.close()
    ctx.close()


@pytest.mark.parametrize("K,blocks,rows", [(128, 8, 2), (128, 16, 2), (256, 8, 2), (100, 24, 1), (128, 0, 1), (64, 8, 2)])
def test_document_blocked_statistics_gather(capi, ap_train, K, blocks, rows):
    """variational_bayes.py:207 as a gather over the postings: cutting a term's segments at document-block
    boundaries and running them in XCD order (option gather_blocks, automatic only for corpora whose t rows
    exceed the L2) must give the same statistics as the unblocked pass - same terms, another summation order -
    and the oracle's."""
    from oracle import c_oracle
    g = ap_train
    rng = np.random.default_rng(K + blocks)
    ptr = g["doc_ptr"][:401]
    tid, tct = g["term_id"][:ptr[-1]], g["term_ct"][:ptr[-1]]
    eta = rng.gamma(100.0, 0.01, (K, 6806))
    alpha = rng.uniform(0.05, 1.0, K)
    out = {}
    for nb in (0, blocks):
        ctx = capi.Context(K, 6806)
        ctx.set_option("gather_live", 0)                 # (the row gathers; the pass over the live lists: test_statistics_from_live_lists)
        ctx.set_option("gather_rows", rows)
        ctx.set_option("gather_blocks", nb)
        corpus = ctx.corpus(ptr, tid, tct)
        res = ctx.estep_host(corpus, alpha, eta)
        out[nb] = (res["sstats"], res["document_log_likelihood"])
        assert corpus.layout("gather_blocks") == max(1, nb)
        ctx.set_option("doc_values", 0)                  # the corpus entropy term comes from the same pass
        ctx.estep(corpus)
        assert abs(ctx.estep_results(corpus)[0] - res["document_log_likelihood"]) < 1e-11 * abs(res["document_log_likelihood"])
        corpus.close()
        ctx.close()
    ref = c_oracle.e_step(alpha, eta, ptr, tid, tct)
    assert np.max(np.abs(out[blocks][0] - out[0][0])) < 1e-11
    assert np.max(np.abs(out[blocks][0] - ref["sstats"])) < SSTATS_ATOL
    assert abs(out[blocks][0].sum() - tct.sum()) < 1e-7
    assert out[blocks][1] == out[0][1]


@pytest.mark.parametrize("K,blocks", [(128, 16), (256, 8), (100, 24)])
def test_statistics_gather_in_rounds(capi, ap_train, K, blocks):
    """A (term, document block) pair costs a partial row; beyond a budget the gather runs in rounds over term ranges
    that reuse the rows (cfg 4: 45 GB in one go).  With a 1 MiB budget a small corpus takes several rounds: the
    statistics and the corpus likelihood are bitwise those of the single round."""
    g = ap_train
    rng = np.random.default_rng(K + blocks)
    ptr = g["doc_ptr"][:601]
    tid, tct = g["term_id"][:ptr[-1]], g["term_ct"][:ptr[-1]]
    eta = rng.gamma(100.0, 0.01, (K, 6806))
    alpha = rng.uniform(0.05, 1.0, K)
    got = {}
    for budget in (0, 1):
        ctx = capi.Context(K, 6806)
        ctx.set_option("gather_live", 0)
        ctx.set_option("gather_sweep", 0)
        ctx.set_option("gather_blocks", blocks)
        ctx.set_option("gather_round_mb", budget)
        corpus = ctx.corpus(ptr, tid, tct)
        res = ctx.estep_host(corpus, alpha, eta)
        rounds, rows = corpus.layout("gather_rounds"), corpus.layout("gather_partial_rows")
        ctx.set_option("doc_values", 0)
        ctx.estep(corpus)
        got[budget] = (res["sstats"], ctx.estep_results(corpus)[0], rounds, rows, corpus.layout("gather_segments"))
        corpus.close()
        ctx.close()
    assert got[0][2] == 1 and got[0][3] == got[0][4]
    assert got[1][2] >= 3 and got[1][3] <= got[0][3] // 3, got[1][2:]      # (a round holds at least one of the cut's 8 pieces)
    assert np.array_equal(got[0][0], got[1][0])
    assert abs(got[0][1] - got[1][1]) <= 1e-13 * abs(got[0][1])          # (the entropy partials are cut differently)


@pytest.mark.parametrize("K,blocks,wide", [(128, 16, 0), (256, 8, 0), (100, 24, 1), (200, 40, 0)])
def test_statistics_gather_as_a_persistent_sweep(capi, ap_train, K, blocks, wide):
    """sstats_sweep.h: every wavefront owns a few terms and keeps their accumulators in registers while all
    workgroups walk the document blocks together - no partial rows.  Against the dispatch-paced gather (same terms,
    another summation order), the oracle, bitwise repeatable; the corpus likelihood of the training fast path comes
    out of the same pass."""
    from oracle import c_oracle
    g = ap_train
    rng = np.random.default_rng(K * 3 + blocks)
    ptr = g["doc_ptr"][:801]
    tid, tct = g["term_id"][:ptr[-1]], g["term_ct"][:ptr[-1]]
    eta = rng.gamma(100.0, 0.01, (K, 6806))
    alpha = rng.uniform(0.05, 1.0, K)
    got = {}
    for sweep in (0, 1, 1):
        ctx = capi.Context(K, 6806)
        ctx.set_option("gather_live", 0)
        ctx.set_option("gather_sweep", 2 * sweep)            # (2: whatever the size of the partial rows)
        ctx.set_option("gather_blocks", blocks)
        ctx.set_option("wide_postings", wide)
        corpus = ctx.corpus(ptr, tid, tct)
        res = ctx.estep_host(corpus, alpha, eta)
        assert (corpus.layout("gather_sweep_passes") >= 1) == bool(sweep)
        assert corpus.layout("gather_partial_rows") == (0 if sweep else corpus.layout("gather_segments"))
        ctx.set_option("doc_values", 0)
        ctx.estep(corpus)
        fast = ctx.estep_results(corpus)[0]
        assert abs(fast - res["document_log_likelihood"]) < 1e-11 * abs(res["document_log_likelihood"])
        got.setdefault(sweep, []).append((res["sstats"], fast))
        corpus.close()
        ctx.close()
    ref = c_oracle.e_step(alpha, eta, ptr, tid, tct)
    assert np.max(np.abs(got[1][0][0] - got[0][0][0])) < 1e-11
    assert np.max(np.abs(got[1][0][0] - ref["sstats"])) < SSTATS_ATOL
    assert abs(got[1][0][0].sum() - tct.sum()) < 1e-7
    assert np.array_equal(got[1][0][0], got[1][1][0]) and got[1][0][1] == got[1][1][1]
    # pacing and scheduling options change WHEN things run, never a bit of the result: the sweep's rendezvous per XCD or
    # chip-wide, the sub-steps it walks a block in, the document-terms pass beside the gather or in front of it, the
    # launch order of the classes
    for options in ([("gather_sweep", 2), ("sweep_xcd", 0)], [("terms_overlap", 0)], [("launch_order", 0)],
                    [("gather_sweep", 2), ("sweep_sub", 1)], [("gather_sweep", 2), ("sweep_sub", 7)],     # sub-steps of a block
                    [("terms_overlap", 0), ("launch_order", 0), ("gather_sweep", 2), ("sweep_xcd", 0)]):
        ctx = capi.Context(K, 6806)
        ctx.set_option("gather_live", 0)
        ctx.set_option("gather_blocks", blocks)
        ctx.set_option("wide_postings", wide)
        for name, value in options:
            ctx.set_option(name, value)
        corpus = ctx.corpus(ptr, tid, tct)
        ctx.set_alpha(alpha)
        ctx.set_eta(eta)
        ctx.set_option("doc_values", 0)
        ctx.estep(corpus)
        fast = ctx.estep_results(corpus)[0]
        want = got[1][0] if ("gather_sweep", 2) in options else got[0][0]
        assert np.array_equal(ctx.get_sstats(), want[0]) and fast == want[1], options
        corpus.close()
        ctx.close()


@pytest.mark.parametrize("K,wide", [(128, 0), (256, 0), (200, 1), (100, 0)])
def test_statistics_from_live_lists(capi, ap_train, K, wide):
    """sstats_live.h: a document the live-topic kernel finished leaves a list of its live topics and their t instead of
    a row of t; the statistics pass adds r t per list entry into LDS accumulators, one wavefront per posting segment.
    With alpha = 1 / K most topics of a document die (gamma_k == alpha_k bitwise): against the row gather of the same
    E-step, the oracle, bitwise repeatable, 64-bit posting positions; documents without a list (the dense kernels
    finished them) are mixed in."""
    from oracle import c_oracle
    g = ap_train
    rng = np.random.default_rng(5 * K + wide)
    ptr = g["doc_ptr"][:701]
    tid, tct = g["term_id"][:ptr[-1]], g["term_ct"][:ptr[-1]]
    eta = rng.gamma(100.0, 0.01, (K, 6806))
    eta[:, rng.permutation(6806)[:3000]] *= rng.uniform(0.01, 30.0, (K, 3000))       # topics with a vocabulary of their own
    alpha = np.full(K, 1.0 / K)
    got = {}
    for live in (0, 1, 1):
        ctx = capi.Context(K, 6806)
        ctx.set_option("gather_live", live)
        ctx.set_option("wide_postings", wide)
        corpus = ctx.corpus(ptr, tid, tct)
        res = ctx.estep_host(corpus, alpha, eta)
        assert corpus.layout("gather_live") == live
        ctx.set_option("doc_values", 0)
        ctx.set_profiling(True)
        ctx.work_counters()
        ctx.estep(corpus)
        fast = ctx.estep_results(corpus)[0]
        ctx.work_counters()
        handed = ctx.executed_work()[1]
        assert abs(fast - res["document_log_likelihood"]) < 1e-11 * abs(res["document_log_likelihood"])
        got.setdefault(live, []).append((res["sstats"], fast, res["iters"], handed))
        corpus.close()
        ctx.close()
    ref = c_oracle.e_step(alpha, eta, ptr, tid, tct)
    assert 0 < got[1][0][3] <= 700, "documents must have been handed to the live-topic kernel"
    assert np.array_equal(got[1][0][2], ref["iters"])
    assert np.max(np.abs(got[1][0][0] - got[0][0][0])) < 1e-11
    assert np.max(np.abs(got[1][0][0] - ref["sstats"])) < SSTATS_ATOL
    assert abs(got[1][0][0].sum() - tct.sum()) < 1e-7
    assert np.array_equal(got[1][0][0], got[1][1][0]) and got[1][0][1] == got[1][1][1]


def test_runs_on_the_system_hip_runtime_without_torch():
    """The library does not need PyTorch: with PYLDA_HIP_RUNTIME=system the loader leaves torch's bundled HIP
    runtime alone, and a process that never imports torch runs the smoke E-step against the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as g; g.smoke(); "
            "assert 'torch' not in sys.modules, 'torch was imported'; print('no-torch ok')" % root)
    env = dict(os.environ, PYLDA_HIP_RUNTIME="system")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "no-torch ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_nips_k500_matches_reference_goldens(capi):
    """BASELINE.json cfg 5 in miniature (parsed/nips.88-05, K=500, documents up to 482 distinct terms,
    goldens from the reference itself): exercises the streaming large-K kernel."""
    g = load_golden("nips_k500.npz")
    K, V = int(g["K"]), len(g["words"])
    np.random.seed(int(g["seed"]))
    eta = np.random.gamma(100., 1. / 100., (K, V))                  # the reference's draw, variational_bayes.py:95
    alpha = np.full(K, 1.0 / K)
    ptr, tid, tct = g["doc_ptr"], g["term_id"].astype(np.int32), g["term_ct"].astype(np.int32)
    for options in ([], [("force_variant", 3)]):                      # automatic (streaming) and generic
        out = run(capi, alpha, eta, ptr, tid, tct, options=options)
        assert out["logspace_docs"] == 0
        check_against(out, g["gamma"], g["doc_ll"], g["iters"])
        assert abs(out["document_log_likelihood"] - float(g["corpus_ll"])) < 1e-9 * abs(float(g["corpus_ll"]))
        assert np.max(np.abs(out["sstats"].sum(axis=1) - g["sstats_rowsum"])) < 1e-8
        assert np.max(np.abs(out["sstats"].sum(axis=0) - g["sstats_colsum"])) < 1e-8
        assert np.max(np.abs(out["sstats"][::7, ::11] - g["sstats_sample"])) < 1e-9
    held = run(capi, alpha, eta, g["test_doc_ptr"], g["test_term_id"].astype(np.int32),
               g["test_term_ct"].astype(np.int32), heldout=True)
    check_against(held, g["heldout_gamma"], g["heldout_words_ll"], g["heldout_iters"], ll_key="doc_words_ll")


@pytest.mark.parametrize("K,V,mean_len", [(100, 900, 150), (128, 1200, 210), (256, 3000, 190), (256, 3000, 330),
                                          (256, 3000, 150), (200, 3000, 120)])
def test_iteration_cap_and_threshold_on_register_kernels(capi, K, V, mean_len):
    """The register-resident kernels evaluate the stop test of iteration i behind the first half of
    iteration i+1 (an integer compare on a fixed-point sum): caps, loose / zero / negative thresholds
    must give the reference's iteration counts (variational_bayes.py:174,187-190)."""
    from oracle import c_oracle
    rng = np.random.default_rng(K + mean_len)
    ptr, ids, cts = random_corpus(rng, 24, V, mean_len)
    eta = rng.gamma(100.0, 0.01, (K, V))
    alpha = rng.uniform(0.05, 1.0, K)
    # (7, 0.0), (5, -1.0), (9, 1e-13), (50, 2000.0): outside the fixed-point stop test's range - the library
    # routes those E-steps to the kernels that compare in floating point (plan.hip choose_variant)
    for mi, tol in [(1, 1e-6), (2, 1e-6), (50, 1e-1), (7, 0.0), (5, -1.0), (50, 1e-3), (9, 1e-13), (50, 2000.0)]:
        ref = c_oracle.e_step(alpha, eta, ptr, ids, cts, max_iter=mi, tol=tol)
        out = run(capi, alpha, eta, ptr, ids, cts, max_iter=mi, tol=tol)
        assert np.array_equal(out["iters"], ref["iters"]), (mi, tol)
        assert rel_err(out["gamma"], ref["gamma"]) < 1e-9
        assert np.max(np.abs(out["doc_ll"] - ref["doc_ll"]) / (np.abs(ref["doc_ll"]) + 1.0)) < 1e-9
        assert np.max(np.abs(out["sstats"] - ref["sstats"])) < SSTATS_ATOL


def test_edge_cases_empty_ragged_and_limits(capi):
    from oracle import c_oracle
    rng = np.random.default_rng(0)
    K, V = 7, 30
    eta = rng.gamma(100.0, 0.01, (K, V))
    alpha = np.full(K, 1.0 / K)
    # ragged: an empty document, a one-term document, a document using every type
    ptr = np.array([0, 0, 1, 1 + V, 1 + V + 2], np.int64)
    ids = np.concatenate([[4], np.arange(V), [0, V - 1]]).astype(np.int32)
    cts = np.concatenate([[1000], rng.integers(1, 9, V), [1, 1]]).astype(np.int32)
    ref = c_oracle.e_step(alpha, eta, ptr, ids, cts)
    out = run(capi, alpha, eta, ptr, ids, cts)
    assert np.array_equal(out["iters"], ref["iters"])
    assert out["iters"][0] == 1 and abs(out["doc_ll"][0]) < 1e-12      # empty doc: gamma = alpha
    assert rel_err(out["gamma"], ref["gamma"]) < 1e-11
    assert np.max(np.abs(out["doc_ll"] - ref["doc_ll"])) < 1e-9
    assert np.max(np.abs(out["sstats"] - ref["sstats"])) < 1e-9
    # iteration cap of 1 and a loose threshold
    for mi, tol in [(1, 1e-6), (50, 1e-1), (7, 0.0)]:
        ref = c_oracle.e_step(alpha, eta, ptr, ids, cts, max_iter=mi, tol=tol)
        out = run(capi, alpha, eta, ptr, ids, cts, max_iter=mi, tol=tol)
        assert np.array_equal(out["iters"], ref["iters"])
        assert rel_err(out["gamma"], ref["gamma"]) < 1e-11
    # zero documents
    out = run(capi, alpha, eta, np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert out["gamma"].shape == (0, K) and out["document_log_likelihood"] == 0.0
    assert np.all(out["sstats"] == 0.0)


def test_error_reporting(capi):
    ctx = capi.Context(3, 5)
    with pytest.raises(capi.PyldaError) as e:
        ctx.corpus(np.array([0, 2]), np.array([1, 7], np.int32), np.array([1, 1], np.int32))
    assert e.value.status == -1 and "outside" in str(e.value)
    with pytest.raises(capi.PyldaError):
        ctx.corpus(np.array([0, 1]), np.array([1], np.int32), np.array([0], np.int32))     # zero count
    corpus = ctx.corpus(np.array([0, 1]), np.array([1], np.int32), np.array([2], np.int32))
    with pytest.raises(capi.PyldaError) as e:
        ctx.estep(corpus)                                   # eta / alpha never set
    assert e.value.status == -4
    with pytest.raises(capi.PyldaError):
        ctx.set_alpha(np.array([0.1, -1.0, 0.1]))
    ctx.set_alpha(np.full(3, 0.1))
    ctx.set_eta(np.ones((3, 5)))
    with pytest.raises(capi.PyldaError):
        ctx.estep(corpus, max_iter=0)
    with pytest.raises(capi.PyldaError) as e:
        ctx.get_sstats()                                    # before any training E-step
    assert e.value.status == -4
    ctx.estep(corpus)
    assert ctx.get_sstats().shape == (3, 5)
    with pytest.raises(capi.PyldaError):
        capi.Context(0, 5)
    ctx.close()
    # documents of any length run (round 2 refused anything above ~5,700 distinct terms) ...
    big = capi.Context(8, 20000)
    big.set_alpha(np.full(8, 0.2))
    big.set_eta(np.random.default_rng(0).gamma(100.0, 0.01, (8, 20000)))
    for n, kernel in ((4000, "generic_global"), (12000, "generic_huge")):
        ok = big.corpus(np.array([0, n]), np.arange(n, dtype=np.int32), np.ones(n, np.int32))
        assert [c["kernel"] for c in ok.plan()] == [kernel]
        big.estep(ok)
        ll, _, _ = big.estep_results(ok)
        assert np.isfinite(ll) and abs(big.get_sstats().sum() - n) < 1e-8
        ok.close()
    big.close()
    # ... what is refused, with a clear message, is a K whose K-sized per-document arrays exceed the LDS
    wide = capi.Context(6000, 4)
    with pytest.raises(capi.PyldaError) as e:
        wide.corpus(np.array([0, 1]), np.array([1], np.int32), np.array([2], np.int32))
    assert e.value.status == -1 and "LDS" in str(e.value)
    wide.close()


def test_randomised_parity_sweep(capi):
    """tools/fuzz_parity.py for 30 s: K from 1 to 1100, documents of 1 to ~600 terms, alpha from 0.005 to 1.5, three
    thresholds, random settings of the statistics gather - every document stops on the C oracle's inner iteration
    (the tolerances on log-likelihood, gamma, statistics and the fast path are asserted inside the sweep)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    got = mod.sweep(30.0, seed=3, verbose=False)
    assert got["cases"] >= 20 and got["documents"] >= 200, got
    assert got["flips"] == 0, got


def _random_corpus(rng, V, lengths):
    ptr, ids, cts = [0], [], []
    for n in lengths:
        ids.append(np.sort(rng.choice(V, size=n, replace=False)).astype(np.int32))
        cts.append(rng.integers(1, 5, size=n).astype(np.int32))
        ptr.append(ptr[-1] + n)
    return np.array(ptr, np.int64), np.concatenate(ids), np.concatenate(cts)


def test_a_document_of_twenty_thousand_terms_at_k700(capi):
    """The reference accepts any document length and any K (variational_bayes.py:98-130, :132).  One document
    of 20,000 distinct terms - 560 KB of per-term scalars alone, more than the LDS holds - beside ordinary ones,
    K = 700: training and held-out mode against the C oracle."""
    from oracle import c_oracle
    rng = np.random.default_rng(5)
    K, V = 700, 30000
    ptr, ids, cts = _random_corpus(rng, V, [20000, 5, 130, 6100, 1])
    eta = rng.gamma(100.0, 0.01, (K, V))
    alpha = np.full(K, 0.05)
    ref = c_oracle.e_step(alpha, eta, ptr, ids, cts)
    ctx = capi.Context(K, V)
    corpus = ctx.corpus(ptr, ids, cts)
    kernels = {c["kernel"] for c in corpus.plan()}
    assert "generic_huge" in kernels, kernels
    out = ctx.estep_host(corpus, alpha, eta)
    check_against(out, ref["gamma"], ref["doc_ll"], ref["iters"])
    assert np.max(np.abs(out["sstats"] - ref["sstats"])) < SSTATS_ATOL
    assert abs(out["sstats"].sum() - cts.sum()) < 1e-6
    held_ref = c_oracle.e_step(alpha, eta, ptr, ids, cts, heldout=True)
    held = ctx.estep_host(corpus, alpha, eta, heldout=True)
    check_against(held, held_ref["gamma"], held_ref["doc_words_ll"], held_ref["iters"], ll_key="doc_words_ll")
    # the training fast path (what learning() runs) gives the same corpus likelihood
    ctx.set_option("doc_values", 0)
    ctx.estep(corpus)
    assert abs(ctx.estep_results(corpus)[0] - out["document_log_likelihood"]) < 1e-10 * abs(out["document_log_likelihood"])
    corpus.close()
    ctx.close()


@pytest.mark.parametrize("K,V,lengths", [(700, 1500, (230, 1, 9, 64, 410, 1024, 1025, 17, 2)),
                                         (1000, 2000, (300, 8, 16, 7, 1000, 129)),
                                         (520, 1200, (5, 250, 250, 33, 900)),
                                         (800, 1300, (180, 2, 640, 15, 16)),
                                         (1024, 1100, (40, 1024, 3))])
def test_fused_streaming_kernel_above_512_topics(capi, K, V, lengths):
    """512 < K <= 1024 (table stride 640 .. 1024): every row streamed once per inner iteration, two buffers in flight
    (estep_qfusek.h) - against the C oracle and the generic kernel, training and held-out, early stops, bitwise
    repeatable; documents of 1 term, of exactly 8 slots x 128 (the kernel's capacity) and one beyond it (generic)."""
    from oracle import c_oracle
    rng = np.random.default_rng(K + V)
    ptr, ids, cts = _random_corpus(rng, V, lengths)
    eta = rng.gamma(100.0, 0.01, (K, V))
    eta[:, rng.choice(V, V // 4, replace=False)] = 1.0 / V
    alpha = rng.uniform(0.05, 1.5, K)
    ref = c_oracle.e_step(alpha, eta, ptr, ids, cts)
    ctx = capi.Context(K, V)
    corpus = ctx.corpus(ptr, ids, cts)
    kernels = {c["kernel"] for c in corpus.plan()}
    assert "qfusek" in kernels and (max(lengths) <= 1024 or kernels & {"generic_global", "generic512", "generic_huge"}), kernels
    corpus.close()
    ctx.close()
    gen = run(capi, alpha, eta, ptr, ids, cts, options=[("force_variant", 3)])
    out = run(capi, alpha, eta, ptr, ids, cts)
    check_against(out, ref["gamma"], ref["doc_ll"], ref["iters"])
    assert np.max(np.abs(out["sstats"] - ref["sstats"])) < SSTATS_ATOL
    assert np.array_equal(out["iters"], gen["iters"]) and rel_err(out["gamma"], gen["gamma"]) < 1e-9
    held_ref = c_oracle.e_step(alpha, eta, ptr, ids, cts, heldout=True)
    held = run(capi, alpha, eta, ptr, ids, cts, heldout=True)
    check_against(held, held_ref["gamma"], held_ref["doc_words_ll"], held_ref["iters"], ll_key="doc_words_ll")
    again = run(capi, alpha, eta, ptr, ids, cts)
    assert np.array_equal(out["gamma"], again["gamma"]) and np.array_equal(out["sstats"], again["sstats"])
    assert np.array_equal(out["doc_ll"], again["doc_ll"])
    for mi, tol in [(1, 1e-6), (3, 1e-6), (50, 1e-2)]:
        ref = c_oracle.e_step(alpha, eta, ptr, ids, cts, max_iter=mi, tol=tol)
        out = run(capi, alpha, eta, ptr, ids, cts, max_iter=mi, tol=tol)
        assert np.array_equal(out["iters"], ref["iters"]), (mi, tol)
        assert rel_err(out["gamma"], ref["gamma"]) < 1e-9
    # the training fast path (doc_terms pass) agrees with the complete per-document values
    ctx = capi.Context(K, V)
    corpus = ctx.corpus(ptr, ids, cts)
    full = ctx.estep_host(corpus, alpha, eta)
    ctx.set_option("doc_values", 0)
    ctx.estep(corpus)
    assert abs(ctx.estep_results(corpus)[0] - full["document_log_likelihood"]) < 1e-10 * abs(full["document_log_likelihood"])
    corpus.close()
    ctx.close()


@pytest.mark.parametrize("K,rows,blocks", [(10, 2, 0), (20, 2, 0), (50, 1, 0), (128, 2, 16), (128, 1, 8), (256, 2, 8),
                                           (256, 0, 0), (300, 2, 0)])
def test_sixty_four_bit_posting_positions(capi, ap_train, K, rows, blocks):
    """Corpora of 2^31 or more (document, term) pairs address r_dn through 64-bit CSR positions in the postings
    (automatic from that size; such a corpus does not fit a test).  Option wide_postings runs the same code -
    device radix sort of (term, int64 position) pairs, every gather kernel - on a small corpus: the statistics are
    bitwise those of the 32-bit path."""
    g = ap_train
    rng = np.random.default_rng(K)
    ptr = g["doc_ptr"][:301]
    tid, tct = g["term_id"][:ptr[-1]], g["term_ct"][:ptr[-1]]
    eta = rng.gamma(100.0, 0.01, (K, 6806))
    alpha = rng.uniform(0.05, 1.0, K)
    got = []
    for wide in (0, 1):
        ctx = capi.Context(K, 6806)
        ctx.set_option("gather_rows", rows)
        ctx.set_option("gather_blocks", blocks)
        ctx.set_option("wide_postings", wide)
        corpus = ctx.corpus(ptr, tid, tct)
        res = ctx.estep_host(corpus, alpha, eta)
        got.append(res)
        corpus.close()
        ctx.close()
    assert np.array_equal(got[0]["sstats"], got[1]["sstats"])
    assert got[0]["document_log_likelihood"] == got[1]["document_log_likelihood"]
    assert abs(got[1]["sstats"].sum() - tct.sum()) < 1e-7
