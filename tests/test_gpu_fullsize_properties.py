"""Parity at BASELINE.json's FULL sizes (cfg 3: 100k documents K=128, cfg 4: 1M documents K=256)
through size-independent properties (the oracle cannot finish 100k documents): conservation laws of the algorithm, fast-path / per-document agreement,
shard-additivity of the sufficient statistics, bitwise reproducibility, and an oracle spot check
on documents drawn from the full-size run."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


CONFIGS = {
    # name: (bench workload, documents, V, K, seed)
    "cfg3": ("synth100k", 100000, 50000, 128, 1234),      # bench.py's primary workload
    "cfg4": ("synth1m", 1000000, 100000, 256, 5678),      # BASELINE.json configs[3] at its full size on one GPU
}


@pytest.fixture(scope="module", params=["cfg3", "cfg4"])
def cfg3(request):
    """cfg 3: synthetic LDA corpus, 100,000 documents, V=50,000, K=128 (quilt kernels);
    cfg 4: 1,000,000 documents, V=100,000, K=256 (8-wavefront quad kernels up to 224 distinct terms on chip, with streamed
    word slots up to 256, the group-fused streaming kernel for the 29 documents beyond).  Same generator, seeds and sizes as bench.py."""
    import bench
    from pylda_amd import _capi
    from pylda_amd.corpus import corpus_checksum, synthetic_lda_shard
    workload, D, V, K, seed = CONFIGS[request.param]
    ptr, ids, cts = synthetic_lda_shard(D, V, 0, D, 128, 200, seed, chunk=25000, device="cuda", workers=8)
    assert corpus_checksum(ptr, ids, cts) == bench.EXPECTED_CHECKSUM[workload]     # the corpus bench.py times
    np.random.seed(0)
    eta = np.random.gamma(100., 1. / 100., (K, V))
    alpha = np.full(K, 1.0 / K)
    ctx = _capi.Context(K, V)
    corpus = ctx.corpus(ptr, ids, cts)
    ctx.set_alpha(alpha)
    ctx.set_eta(eta)
    ctx.set_option("doc_values", 1)
    ctx.estep(corpus)
    ll, _, nlog = ctx.estep_results(corpus)
    out = dict(name=request.param, ctx=ctx, corpus=corpus, ptr=ptr, ids=ids, cts=cts, K=K, V=V, D=D, eta=eta,
               alpha=alpha, ll=ll, nlog=nlog, gamma=ctx.get_gamma(corpus), sstats=ctx.get_sstats())
    out["doc_ll"], _, out["iters"] = ctx.get_doc_values(corpus)
    print("%s: launch classes %s" % (request.param, [(c["kernel"], c["geometry"], c["documents"]) for c in corpus.plan()]))
    yield out
    corpus.close()
    ctx.close()


def test_conservation_laws_at_full_size(cfg3):
    c = cfg3
    tokens = c["cts"].sum()
    assert c["nlog"] == 0
    # sstats.sum() == #tokens: every phi row sums to one
    assert abs(c["sstats"].sum() - tokens) < 1e-9 * tokens
    # column sums of sstats == corpus word counts
    word_tokens = np.bincount(c["ids"], weights=c["cts"], minlength=c["V"])
    assert np.max(np.abs(c["sstats"].sum(axis=0) - word_tokens)) < 1e-8 * word_tokens.max()
    # sum_k gamma_dk - sum alpha == tokens of document d, for every document (:185)
    doc_tokens = np.add.reduceat(c["cts"].astype(np.float64), c["ptr"][:-1])
    assert rel_err(c["gamma"].sum(axis=1) - c["alpha"].sum(), doc_tokens) < 1e-12
    # topic mass: sum_d (gamma_dk - alpha_k) == sum_w sstats[k][w] only at a fixed point; what always holds is
    # the total: both equal #tokens
    assert abs((c["gamma"] - c["alpha"]).sum() - tokens) < 1e-9 * tokens
    assert np.all(c["gamma"] > 0) and np.all(np.isfinite(c["doc_ll"]))
    assert c["iters"].min() >= 1 and c["iters"].max() <= 50
    # corpus value == sum of the per-document values
    assert abs(c["doc_ll"].sum() - c["ll"]) < 1e-11 * abs(c["ll"])


def test_fast_path_and_reproducibility_at_full_size(cfg3):
    c, ctx, corpus = cfg3, cfg3["ctx"], cfg3["corpus"]
    ctx.set_option("doc_values", 0)                       # what learning() / bench.py run
    ctx.estep(corpus)
    ll_fast = ctx.estep_results(corpus)[0]
    assert abs(ll_fast - c["ll"]) < 1e-11 * abs(c["ll"])
    assert np.array_equal(ctx.get_sstats(), c["sstats"])  # bitwise: fixed summation order everywhere
    assert np.array_equal(ctx.get_gamma(corpus), c["gamma"])
    ctx.set_option("doc_values", 1)


def test_live_topic_kernel_against_the_dense_kernels_at_full_size(cfg3):
    """on the FULL cfg 3 / cfg 4 corpora the E-step with the hand-over to the live-topic kernel
    (estep_compact.h, the default the fixture ran) and the dense kernels alone (option compact = 0) execute the same
    number of inner iterations for EVERY document; gamma, the per-document log-likelihoods and the statistics agree
    to rounding - another summation order inside normalisers and topic sums, not another result.  The counts of
    entries that differ at all are printed."""
    c, ctx, corpus = cfg3, cfg3["ctx"], cfg3["corpus"]
    ctx.set_profiling(True)
    ctx.work_counters()
    ctx.estep(corpus)                                     # (the fixture's E-step again, for the work counters)
    its, terms = ctx.work_counters()
    entries, handed = ctx.executed_work()
    ctx.set_profiling(False)
    assert handed > 0.99 * c["D"], "nearly every document goes to the live-topic kernel: %d of %d" % (handed, c["D"])
    live_fraction = entries / (c["K"] * terms)
    ctx.set_option("compact", 0)
    ctx.estep(corpus)
    ll, _, nlog = ctx.estep_results(corpus)
    doc_ll, _, iters = ctx.get_doc_values(corpus)
    gamma = ctx.get_gamma(corpus)
    sstats = ctx.get_sstats()
    ctx.set_option("compact", 1)
    flips = int((iters != c["iters"]).sum())
    differing = int((gamma != c["gamma"]).sum())
    g_rel = float(np.max(np.abs(gamma - c["gamma"]) / gamma))
    ll_rel = float(np.max(np.abs(doc_ll - c["doc_ll"]) / np.abs(doc_ll)))
    ss_abs = float(np.max(np.abs(sstats - c["sstats"])))
    print("%s: live-topic vs dense kernels: %d iteration-count flips in %d documents; %d of %d gamma entries differ, max rel %.2e; "
          "per-document log-likelihood max rel %.2e; statistics max abs %.2e; executed fraction of the dense tile work %.3f"
          % (c["name"], flips, c["D"], differing, gamma.size, g_rel, ll_rel, ss_abs, live_fraction))
    assert nlog == 0 and flips == 0
    assert g_rel < 1e-9 and ll_rel < 1e-11 and ss_abs < 1e-9
    assert abs(ll - c["ll"]) < 1e-13 * abs(c["ll"])
    assert live_fraction < 0.75


def test_shard_additivity_at_full_size(cfg3):
    """Document sharding (the multi-GPU decomposition): statistics of the shards add up to the whole."""
    from pylda_amd.corpus import shard_csr
    c, ctx = cfg3, cfg3["ctx"]
    total = np.zeros_like(c["sstats"])
    ll = 0.0
    for rank in range(2):
        sp, si, sc, (lo, hi) = shard_csr(c["ptr"], c["ids"], c["cts"], 2, rank)
        shard = ctx.corpus(sp, si, sc)
        ctx.estep(shard)
        ll += ctx.estep_results(shard)[0]
        total += ctx.get_sstats()
        assert np.array_equal(ctx.get_gamma(shard), c["gamma"][lo:hi])     # documents are independent
        shard.close()
    assert np.max(np.abs(total - c["sstats"])) < 1e-9
    assert abs(ll - c["ll"]) < 1e-11 * abs(c["ll"])


def test_oracle_spot_check_on_full_size_run(cfg3):
    from oracle import c_oracle
    c = cfg3
    rng = np.random.default_rng(0)
    docs = np.sort(rng.choice(c["D"], 24, replace=False))
    order = np.argsort(np.diff(c["ptr"]))
    docs = np.unique(np.concatenate([docs, order[:2], order[-2:]]))      # plus the shortest and longest
    if c["name"] == "cfg4":             # the longest have > 256 distinct terms: the group-fused streaming kernel
        assert np.diff(c["ptr"])[order[-1]] > 256
        kernels = {(p["kernel"], p["geometry"]) for p in c["corpus"].plan()}
        assert ("qgroup", 0) in kernels, kernels                                      # > 256 distinct terms
        assert ("quad", 3320804) in kernels and ("quad", 4320804) in kernels, kernels # 225..256: streamed word slots
        long_docs = order[np.diff(c["ptr"])[order] > 224]
        docs = np.unique(np.concatenate([docs, long_docs[:: max(1, len(long_docs) // 6)]]))
        assert ("quad", 321003) in kernels and ("quad", 321002) in kernels, kernels   # the bulk: all words on chip
    from conftest import csr_slice
    ptr, tid, tct = csr_slice(c["ptr"], c["ids"], c["cts"], docs)
    # the sampled documents against BOTH restatements: the numpy/scipy one executes the reference's own operations
    # in the reference's order (bit-pinned to its goldens, tests/test_oracle_golden.py), the C one is the checker of
    # the random-shape tests; every sampled document must stop on the reference's inner iteration
    from oracle import vb_numpy
    for name, ref in (("numpy", vb_numpy.e_step(c["alpha"], c["eta"], ptr, tid, tct)),
                      ("c", c_oracle.e_step(c["alpha"], c["eta"], ptr, tid, tct))):
        assert np.array_equal(ref["iters"], c["iters"][docs]), (name, ref["iters"], c["iters"][docs])
        assert rel_err(c["gamma"][docs], ref["gamma"]) < 1e-9, name
        assert rel_err(c["doc_ll"][docs], ref["doc_ll"]) < 1e-9, name      # (the stated bar is 1e-5)


def test_oracle_check_of_full_size_statistics(cfg3):
    """sstats[:, w] of the FULL-size run (variational_bayes.py:207) against the oracle for terms whose every posting
    the oracle can afford: ~20 terms with 5..50 postings each - the oracle runs exactly the documents that contain
    them, so its column w is the complete corpus sum - plus the single most frequent term on a contiguous document
    shard run through the same context.  cfg 3 takes the dispatch-paced, document-blocked gather, cfg 4 the
    persistent sweep (3 passes): a wrong-but-conserving accumulation would pass the invariants above, not this."""
    from oracle import c_oracle
    from conftest import csr_slice
    c, ctx = cfg3, cfg3["ctx"]
    postings = np.bincount(c["ids"], minlength=c["V"])
    rng = np.random.default_rng(7)
    budget = 600 if c["K"] <= 128 else 300                 # documents the C oracle gets (seconds, not minutes)
    pool = np.nonzero((postings >= 5) & (postings <= 50))[0]
    assert pool.size >= 40
    cand = rng.choice(pool, 40, replace=False)
    pos = np.nonzero(np.isin(c["ids"], cand))[0]           # one pass over the corpus for all candidates
    pos_term = c["ids"][pos]
    pos_doc = np.searchsorted(c["ptr"], pos, side="right") - 1
    terms, docs = [], np.zeros(0, np.int64)
    for w in cand:
        d = pos_doc[pos_term == w]
        assert d.size == postings[w]
        merged = np.union1d(docs, d)
        if merged.size > budget:
            continue
        terms.append(int(w))
        docs = merged
        if len(terms) == 20:
            break
    assert len(terms) >= 10, (len(terms), docs.size)
    ptr, tid, tct = csr_slice(c["ptr"], c["ids"], c["cts"], docs)
    ref = c_oracle.e_step(c["alpha"], c["eta"], ptr, tid, tct)
    assert np.array_equal(ref["iters"], c["iters"][docs])
    got, want = c["sstats"][:, terms], ref["sstats"][:, terms]
    assert np.max(np.abs(got - want)) < 1e-8, np.max(np.abs(got - want))
    assert np.all(want.sum(axis=0) > 0)
    print("%s: %d terms (%d..%d postings) on %d documents, max |sstats - oracle| %.2e"
          % (c["name"], len(terms), postings[terms].min(), postings[terms].max(), docs.size, np.max(np.abs(got - want))))
    # the most frequent term, on a contiguous shard the oracle can run completely
    top = int(np.argmax(postings))
    n_shard = 1200 if c["K"] <= 128 else 600
    lo = int(rng.integers(0, c["D"] - n_shard))
    sel = np.arange(lo, lo + n_shard)
    sp, si, sc = csr_slice(c["ptr"], c["ids"], c["cts"], sel)
    shard = ctx.corpus(sp, si, sc)
    ctx.estep(shard)
    shard_stats = ctx.get_sstats()
    shard.close()
    sref = c_oracle.e_step(c["alpha"], c["eta"], sp, si, sc)
    in_shard = int((si == top).sum())
    assert in_shard >= 5, in_shard
    assert np.max(np.abs(shard_stats[:, top] - sref["sstats"][:, top])) < 1e-8
    assert np.max(np.abs(shard_stats - sref["sstats"])) < 1e-8
    # restore the fixture's state for the tests that follow (the context's statistics are those of the last E-step)
    ctx.estep(c["corpus"])
    assert np.array_equal(ctx.get_sstats(), c["sstats"])


def test_heldout_mode_at_full_size(cfg3):
    """Held-out mode (variational_bayes.py:133-138, :202-204, :216) over the full-size corpus: the inner loop is the
    training one (same gamma, same iteration counts, bit for bit), the sufficient statistics stay untouched, the corpus
    value is the sum of the per-document values, and sampled documents - incl. the streamed and the longest classes -
    match the oracle's words log-likelihood."""
    from conftest import csr_slice
    from oracle import c_oracle
    c = cfg3
    ctx, corpus = c["ctx"], c["corpus"]
    before = ctx.get_sstats()
    ctx.estep(corpus, 50, 1e-6, True)
    _, words_ll, nlog = ctx.estep_results(corpus)
    _, doc_wll, iters = ctx.get_doc_values(corpus)
    gamma = ctx.get_gamma(corpus)
    assert nlog == 0
    assert np.array_equal(iters, c["iters"]) and np.array_equal(gamma, c["gamma"])
    assert np.array_equal(ctx.get_sstats(), before)                      # :216 - the statistics of the training E-step stay
    assert np.all(np.isfinite(doc_wll)) and np.all(doc_wll < 0)
    assert abs(doc_wll.sum() - words_ll) < 1e-11 * abs(words_ll)
    order = np.argsort(np.diff(c["ptr"]))
    rng = np.random.default_rng(1)
    docs = np.unique(np.concatenate([rng.choice(c["D"], 10, replace=False), order[:2], order[-3:],
                                     order[np.searchsorted(np.diff(c["ptr"])[order], [225, 233, 241])]]))
    ptr, tid, tct = csr_slice(c["ptr"], c["ids"], c["cts"], docs)
    ref = c_oracle.e_step(c["alpha"], c["eta"], ptr, tid, tct, heldout=True)
    assert np.array_equal(ref["iters"], iters[docs])
    assert rel_err(doc_wll[docs], ref["doc_words_ll"]) < 1e-9 and rel_err(gamma[docs], ref["gamma"]) < 1e-9
    # leave the fixture as the other tests expect it: the training E-step's results
    ctx.estep(corpus)
