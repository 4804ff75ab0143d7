/*
 * CPU oracle (plain C, float64) for PyLDA's variational-Bayes E-step hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker, never the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it.  Nothing under pylda_amd/ links or calls this file.
 *
 * It restates in log space, in the reference's own operation order, the
 * algorithm of /root/reference/variational_bayes.py:132-216 (e_step) and
 * /root/reference/inferencer.py:15-18 (compute_dirichlet_expectation).
 * The transcendental functions the reference takes from SciPy (psi, gammaln,
 * logsumexp; module SciPy, version unpinned by the reference, README.md:13)
 * are restated here from their published definitions: digamma by upward
 * recurrence + the Bernoulli asymptotic series, ln-Gamma from libm.
 *
 * Parity pinning: tests/test_oracle_golden.py checks this file against the
 * golden vectors produced by running the reference itself in the build
 * container (the npz fixtures under tests/golden) and against scipy samples
 * (tests/golden/special_fn.npz).
 *
 * Build: gcc -O2 -fPIC -shared -o liboracle_vb.so vb_oracle.c -lm
 *        (no -ffast-math: summation order is part of the restatement).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* psi(x) for x > 0.  psi(x) = psi(x+1) - 1/x until x >= 10, then
 * ln x - 1/(2x) - sum_n B_2n / (2n x^2n). */
double vb_oracle_digamma(double x)
{
    double shift = 0.0;
    while (x < 10.0) {
        shift += 1.0 / x;
        x += 1.0;
    }
    double inv = 1.0 / x, inv2 = inv * inv;
    /* B_2n/(2n): 1/12, -1/120, 1/252, -1/240, 1/132, -691/32760, 1/12 */
    double series = inv2 * (1.0 / 12.0 - inv2 * (1.0 / 120.0 - inv2 * (1.0 / 252.0 -
                    inv2 * (1.0 / 240.0 - inv2 * (1.0 / 132.0 - inv2 * (691.0 / 32760.0 -
                    inv2 * (1.0 / 12.0)))))));
    return log(x) - 0.5 * inv - series - shift;
}

/* psi'(x) for x > 0 (used by the alpha Newton update, :286,:294). */
double vb_oracle_trigamma(double x)
{
    double shift = 0.0;
    while (x < 12.0) {
        shift += 1.0 / (x * x);
        x += 1.0;
    }
    double inv = 1.0 / x, inv2 = inv * inv;
    /* 1/x + 1/(2x^2) + sum B_2n / x^(2n+1) */
    double series = inv * (1.0 + inv * (0.5 + inv * (1.0 / 6.0 - inv2 * (1.0 / 30.0 -
                    inv2 * (1.0 / 42.0 - inv2 * (1.0 / 30.0 - inv2 * (5.0 / 66.0 -
                    inv2 * (691.0 / 2730.0 - inv2 * (7.0 / 6.0)))))))));
    return series + shift;
}

double vb_oracle_lgamma(double x)
{
    return lgamma(x);
}

static double logsumexp(const double *v, int n)
{
    double m = v[0];
    for (int i = 1; i < n; ++i)
        if (v[i] > m) m = v[i];
    if (isinf(m)) return m;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += exp(v[i] - m);
    return m + log(s);
}

/* inferencer.py:18 : E_log_eta[k][v] = psi(eta[k][v]) - psi(sum_v eta[k][v]) */
int vb_oracle_dirichlet_expectation(int K, int V, const double *eta, double *E_log_eta)
{
    for (int k = 0; k < K; ++k) {
        const double *row = eta + (size_t)k * V;
        double s = 0.0;
        for (int v = 0; v < V; ++v) s += row[v];
        double ps = vb_oracle_digamma(s);
        for (int v = 0; v < V; ++v) E_log_eta[(size_t)k * V + v] = vb_oracle_digamma(row[v]) - ps;
    }
    return 0;
}

/*
 * Whole-corpus E-step, documents visited in index order (the reference's
 * permutation at :159 only reorders two floating-point sums).
 *
 * eta      K x V row-major (numpy layout of self._eta)
 * gamma    D x K out; doc_ll, words_ll, iters: D out; sstats K x V out (+=0 first)
 * heldout  0: training (:212-214), 1: held-out (:154-155, :202-204)
 */
int vb_oracle_estep(int K, int V, const double *alpha, const double *eta, int64_t D,
                    const int64_t *doc_ptr, const int32_t *term_id, const int32_t *term_ct,
                    int max_iter, double tol, int heldout, double *gamma, double *doc_ll,
                    double *words_ll, int32_t *iters, double *sstats)
{
    double *E = (double *)malloc(sizeof(double) * (size_t)K * V);
    double *lse_row = (double *)calloc((size_t)K, sizeof(double));
    double *psi_g = (double *)malloc(sizeof(double) * K);
    double *g_new = (double *)malloc(sizeof(double) * K);
    int64_t max_n = 1;
    for (int64_t d = 0; d < D; ++d)
        if (doc_ptr[d + 1] - doc_ptr[d] > max_n) max_n = doc_ptr[d + 1] - doc_ptr[d];
    double *log_phi = (double *)malloc(sizeof(double) * (size_t)max_n * K);
    if (!E || !lse_row || !psi_g || !g_new || !log_phi) return -1;

    vb_oracle_dirichlet_expectation(K, V, eta, E);                       /* :152 */
    if (heldout)
        for (int k = 0; k < K; ++k) lse_row[k] = logsumexp(E + (size_t)k * V, V);  /* :155 */
    memset(sstats, 0, sizeof(double) * (size_t)K * V);                   /* :147 */

    double alpha_sum = 0.0, alpha_lg = 0.0;
    for (int k = 0; k < K; ++k) {
        alpha_sum += alpha[k];
        alpha_lg += lgamma(alpha[k]);
    }
    double alpha_term = lgamma(alpha_sum) - alpha_lg;                    /* :195 */

    for (int64_t d = 0; d < D; ++d) {
        const int32_t *ids = term_id + doc_ptr[d];
        const int32_t *cts = term_ct + doc_ptr[d];
        int n_terms = (int)(doc_ptr[d + 1] - doc_ptr[d]);
        double *g = gamma + (size_t)d * K;
        double total = 0.0;
        for (int n = 0; n < n_terms; ++n) total += cts[n];               /* :162 */
        for (int k = 0; k < K; ++k) g[k] = alpha[k] + total / K;         /* :165 */

        int it = 0;
        while (it < max_iter) {                                          /* :174 */
            for (int k = 0; k < K; ++k) psi_g[k] = vb_oracle_digamma(g[k]);
            for (int k = 0; k < K; ++k) g_new[k] = 0.0;
            for (int n = 0; n < n_terms; ++n) {
                double *lp = log_phi + (size_t)n * K;
                for (int k = 0; k < K; ++k) lp[k] = E[(size_t)k * V + ids[n]] + psi_g[k];  /* :177 */
                double z = logsumexp(lp, K);                             /* :182 */
                double lc = log((double)cts[n]);
                for (int k = 0; k < K; ++k) {
                    lp[k] -= z;
                    g_new[k] += exp(lp[k] + lc);                         /* :185 */
                }
            }
            double change = 0.0;
            for (int k = 0; k < K; ++k) {
                g_new[k] += alpha[k];
                change += fabs(g_new[k] - g[k]);                         /* :187 */
                g[k] = g_new[k];                                         /* :188 */
            }
            ++it;
            if (change / K <= tol) break;                                /* :189 */
        }
        iters[d] = it;

        double ll = alpha_term, gsum = 0.0;
        for (int k = 0; k < K; ++k) {
            ll += lgamma(g[k]);
            gsum += g[k];
        }
        ll -= lgamma(gsum);                                              /* :197 */
        double ent = 0.0, wll = 0.0;
        for (int n = 0; n < n_terms; ++n) {
            const double *lp = log_phi + (size_t)n * K;
            double lc = log((double)cts[n]);
            double row = 0.0;
            for (int k = 0; k < K; ++k) {
                double phi = exp(lp[k]);
                row += phi * lp[k];
                double phic = exp(lp[k] + lc);
                sstats[(size_t)k * V + ids[n]] += phic;                  /* :207 */
                if (heldout) wll += phic * (E[(size_t)k * V + ids[n]] - lse_row[k]);  /* :204 */
            }
            ent += cts[n] * row;                                         /* :199 */
        }
        doc_ll[d] = ll - ent;
        words_ll[d] = wll;
    }
    free(E); free(lse_row); free(psi_g); free(g_new); free(log_phi);
    return 0;
}
