"""Randomised parity sweep: python tools/fuzz_parity.py [seconds] [seed]
Random K (1 .. 1100), V, document lengths, alpha and settings of the statistics pass; per-document log-likelihood / gamma / iterations against the C oracle,
and the training fast path (doc_values=0: document-terms pass) against the complete per-document values.
tests/test_gpu_estep.py::test_randomised_parity_sweep runs sweep(30 s) in the GPU suite."""
import sys, time, numpy as np
sys.path.insert(0, ".")


def sweep(budget=60.0, seed=0, verbose=True):
    from pylda_amd import _capi
    from oracle import c_oracle
    rng = np.random.default_rng(seed)
    t0 = time.time()
    cases = worst_ll = worst_g = worst_fast = worst_ss = 0
    flips = docs = handed = 0
    while time.time() - t0 < budget:
        K = int(rng.choice([1, 2, 7, 10, 16, 31, 33, 64, 65, 100, 127, 128, 129, 160, 192, 200, 255, 256, 257, 300, 384, 400, 500, 512,
                            513, 640, 700, 768, 800, 1000, 1024, 1100]))
        V = int(rng.integers(max(20, K // 4), 5000))
        D = int(rng.integers(1, 40))
        mean_len = float(rng.choice([3, 20, 80, 150, 200, 215, 232, 240, 250, 300, 500, 800]))
        ptr, ids, cts = [0], [], []
        for _ in range(D):
            n = int(min(V, max(1, rng.poisson(mean_len))))
            u = np.sort(rng.choice(V, size=n, replace=False)).astype(np.int32)
            ids.append(u); cts.append(rng.integers(1, 6, size=n).astype(np.int32)); ptr.append(ptr[-1] + n)
        ptr = np.array(ptr, np.int64); ids = np.concatenate(ids); cts = np.concatenate(cts)
        eta = rng.gamma(100.0, 0.01, (K, V))
        if rng.random() < 0.5:
            eta[:, rng.choice(V, V // 3, replace=False)] = 1.0 / V
        topical = rng.random() < 0.4
        if topical:             # a model that knows topics with vocabularies of their own: most topics of a document die, documents are handed over
            true_topics = int(rng.choice([4, 12, 24, 40]))
            beta = rng.dirichlet(np.full(V, 0.02), size=true_topics)
            for k in range(K):
                eta[k] += 40.0 * V * beta[k % true_topics] * rng.uniform(0.2, 1.0)
            if rng.random() < 0.7:      # ... and documents drawn from it
                ptr, ids, cts = [0], [], []
                for _ in range(D):
                    theta = rng.dirichlet(np.full(true_topics, float(rng.choice([0.02, 0.1, 0.4]))))
                    words = rng.choice(V, size=max(1, int(rng.poisson(mean_len))), p=theta @ beta)
                    u, c = np.unique(words, return_counts=True)
                    ids.append(u.astype(np.int32)); cts.append(c.astype(np.int32)); ptr.append(ptr[-1] + len(u))
                ptr = np.array(ptr, np.int64); ids = np.concatenate(ids); cts = np.concatenate(cts)
        # (small alpha: topics die - gamma_k == alpha_k bitwise - and documents go to the live-topic kernel)
        alpha = rng.uniform(0.02, 1.5, K) if rng.random() < 0.5 else np.full(K, float(rng.choice([0.005, 0.05, 1.0 / K, 1.0 / K, 0.5 / K])))
        tol = float(rng.choice([1e-6, 1e-6, 1e-4, 1e-8]))
        ref = c_oracle.e_step(alpha, eta, ptr, ids, cts, 50, tol)
        ctx = _capi.Context(K, V)
        ctx.set_option("gather_blocks", int(rng.choice([-1, 0, 8, 16, 40])))      # statistics gather: automatic / unblocked / forced blocks
        ctx.set_option("gather_rows", int(rng.choice([0, 1, 2, 2])))
        ctx.set_option("gather_sweep", int(rng.choice([0, 1, 2, 2])))               # persistent sweep where the gather is blocked
        ctx.set_option("gather_round_mb", int(rng.choice([0, 0, 1])))               # ... or rounds over term ranges
        ctx.set_option("slab_uber", int(rng.choice([0, 1, 1])))
        ctx.set_option("compact", int(rng.choice([0, 1, 1, 1])))                    # hand-over to the live-topic kernel (64 < K <= 256) ...
        ctx.set_option("compact_cap", int(rng.choice([0, 0, 8, 12, 17])))           # ... at its capacity, or earlier shapes of it
        ctx.set_option("compact_phase", int(rng.choice([0, 1])))
        ctx.set_option("compact_pair", int(rng.choice([-1, -1, 0, 1])))             # ... the two-wavefront stage in front of it
        ctx.set_option("compact_stream", int(rng.choice([0, 1, 1])))                # ... behind the fused streaming kernel (256 < K <= 512)
        ctx.set_option("gather_live", int(rng.choice([0, 1, 1])))                   # statistics from the lists of live topics
        ctx.set_option("wide_postings", int(rng.choice([0, 0, 1])))
        corpus = ctx.corpus(ptr, ids, cts)
        out = ctx.estep_host(corpus, alpha, eta, 50, tol, False)
        ctx.set_option("doc_values", 0)
        ctx.set_profiling(True)
        ctx.work_counters()
        ctx.estep(corpus, 50, tol, False)
        fast = ctx.estep_results(corpus)[0]
        ctx.work_counters()
        handed += int(ctx.executed_work()[1])
        same = out["iters"] == ref["iters"]
        flips += int((~same).sum()); docs += D
        if same.any():
            worst_ll = max(worst_ll, float(np.max(np.abs(out["doc_ll"][same] - ref["doc_ll"][same]) / np.maximum(1.0, np.abs(ref["doc_ll"][same])))))
            worst_g = max(worst_g, float(np.max(np.abs(out["gamma"][same] - ref["gamma"][same]) / ref["gamma"][same])))
        if (~same).any():
            assert np.max(np.abs(out["doc_ll"][~same] - ref["doc_ll"][~same]) / np.maximum(1.0, np.abs(ref["doc_ll"][~same]))) < 1e-5
        if same.all():          # (a document that stops one iteration apart carries a different phi into the statistics)
            worst_ss = max(worst_ss, float(np.max(np.abs(out["sstats"] - ref["sstats"]))))
            assert worst_ss < 1e-8, (K, V, D, mean_len, worst_ss)
        full = out["document_log_likelihood"]
        worst_fast = max(worst_fast, abs(fast - full) / max(1.0, abs(full)))
        assert worst_ll < 1e-9 and worst_g < 1e-6 and worst_fast < 1e-10, (K, V, D, mean_len, worst_ll, worst_g, worst_fast, [c["kernel"] for c in corpus.plan()])
        if same.any():
            g_case = float(np.max(np.abs(out["gamma"][same] - ref["gamma"][same]) / ref["gamma"][same]))
            if g_case > 5e-9 and verbose:
                d, k = np.unravel_index(np.argmax(np.abs(out["gamma"] - ref["gamma"]) / ref["gamma"]), ref["gamma"].shape)
                print("  gamma rel %.2e: K=%d V=%d D=%d mean_len=%g tol=%g kernels=%s; doc %d (N=%d, iters %d) topic %d: %.17g vs %.17g, alpha %.4g"
                      % (g_case, K, V, D, mean_len, tol, sorted({c["kernel"] for c in corpus.plan()}), d, ptr[d + 1] - ptr[d], ref["iters"][d], k,
                         out["gamma"][d, k], ref["gamma"][d, k], alpha[k]))
        corpus.close(); ctx.close()
        cases += 1
    summary = {"cases": cases, "documents": docs, "handed_to_live_topic_kernel": handed, "flips": flips, "worst_rel_doc_ll": worst_ll, "worst_rel_gamma": worst_g,
               "worst_fast_path_corpus_ll": worst_fast, "worst_abs_statistics": worst_ss}
    print("fuzz: %d cases, %d documents (%d through the live-topic kernel), %d iteration-count flips; worst rel doc-LL %.2e, gamma %.2e, fast-path corpus LL %.2e, statistics abs %.2e"
          % (cases, docs, handed, flips, worst_ll, worst_g, worst_fast, worst_ss))
    return summary


if __name__ == "__main__":
    sweep(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
