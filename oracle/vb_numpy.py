"""CPU oracle (numpy/scipy) for PyLDA's variational-Bayes E-step hot path.

TEST INFRASTRUCTURE ONLY.  This module is the checker, never the product:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it.  pylda_amd/ must never import anything under oracle/.

It restates, in log space and in the reference's own operation order, the
algorithm of /root/reference/variational_bayes.py:132-216 (e_step),
:218-235 (m_step), :277-324 (optimize_hyperparameters) and
/root/reference/inferencer.py:15-18 (compute_dirichlet_expectation).  The
corpus container is CSR (doc_ptr, term_id, term_ct) instead of the
reference's two Python lists; everything numeric is float64.

Parity pinning: checked against golden vectors produced by importing the
reference itself in the build container (tests/golden/make_golden.py,
tests/test_oracle_golden.py).  The transcendental functions are SciPy's
(psi, gammaln, logsumexp, polygamma) exactly as in the reference, which
names SciPy without pinning a version (README.md:13); goldens were made
with scipy 1.15.3 / numpy 2.2.6.
"""
import numpy as np
from scipy.special import gammaln, logsumexp, polygamma, psi


def compute_dirichlet_expectation(dirichlet_parameter):
    """E[log x] for x ~ Dir(parameter); row-wise for 2-D input.

    Follows inferencer.py:15-18.
    """
    p = np.asarray(dirichlet_parameter, dtype=np.float64)
    if p.ndim == 1:
        return psi(p) - psi(p.sum())
    return psi(p) - psi(p.sum(axis=1))[:, None]


def e_step_document(alpha, E_log_eta, ids, cts, max_iter=50, tol=1e-6,
                    E_log_prob_eta=None):
    """One document of the E-step (variational_bayes.py:162-207).

    Returns (gamma (K,), doc_ll, words_ll, iterations, phi_times_count (N,K)).
    phi is the LAST COMPUTED one, i.e. half a step behind gamma (:177-188).
    """
    K = alpha.shape[0]
    cts = np.asarray(cts, dtype=np.float64).reshape(1, -1)          # (1, N) as at :121
    gamma = alpha + cts.sum() / K                                    # :165
    gathered = E_log_eta[:, ids].T                                   # (N, K)  :177
    log_cts_col = np.log(cts.T)                                      # (N, 1)  :185
    log_phi = None
    iterations = 0
    for _ in range(max_iter):                                        # :174
        log_phi = gathered + psi(gamma)[None, :]                     # :177
        log_phi = log_phi - logsumexp(log_phi, axis=1)[:, None]      # :182
        gamma_update = alpha + np.exp(log_phi + log_cts_col).sum(axis=0)   # :185
        mean_change = np.mean(np.abs(gamma_update - gamma))          # :187
        gamma = gamma_update                                         # :188
        iterations += 1
        if mean_change <= tol:                                       # :189
            break
    doc_ll = gammaln(alpha.sum()) - gammaln(alpha).sum()             # :195
    doc_ll += gammaln(gamma).sum() - gammaln(gamma.sum())            # :197
    doc_ll -= np.sum(np.dot(cts, np.exp(log_phi) * log_phi))         # :199
    words_ll = 0.0
    if E_log_prob_eta is not None:                                   # :202-204
        words_ll = np.sum(np.exp(log_phi.T + np.log(cts)) * E_log_prob_eta[:, ids])
    phi_c = np.exp(log_phi + log_cts_col)                            # :207 (N, K)
    return gamma, float(doc_ll), float(words_ll), iterations, phi_c


def e_step(alpha, eta, doc_ptr, term_id, term_ct, max_iter=50, tol=1e-6,
           heldout=False, order=None):
    """Whole-corpus E-step (variational_bayes.py:132-216) over a CSR corpus.

    `order` is the document visiting order (the reference draws a random
    permutation at :159; it only changes floating-point summation order).
    Returns a dict with the corpus-level values the reference returns and the
    per-document values the parity tests need.
    """
    alpha = np.asarray(alpha, dtype=np.float64)
    eta = np.asarray(eta, dtype=np.float64)
    K, V = eta.shape
    D = len(doc_ptr) - 1
    E_log_eta = compute_dirichlet_expectation(eta)                   # :152
    E_log_prob_eta = None
    if heldout:                                                      # :154-155
        E_log_prob_eta = E_log_eta - logsumexp(E_log_eta, axis=1)[:, None]
    sstats = np.zeros((K, V))                                        # :147
    gamma = np.zeros((D, K))
    doc_ll = np.zeros(D)
    words_ll = np.zeros(D)
    iters = np.zeros(D, dtype=np.int32)
    total_ll = 0.0
    total_words_ll = 0.0
    visit = range(D) if order is None else order
    for d in visit:
        lo, hi = int(doc_ptr[d]), int(doc_ptr[d + 1])
        ids = np.asarray(term_id[lo:hi], dtype=np.int64)
        g, ll, wll, it, phi_c = e_step_document(
            alpha, E_log_eta, ids, term_ct[lo:hi], max_iter, tol, E_log_prob_eta)
        gamma[d] = g
        doc_ll[d] = ll
        words_ll[d] = wll
        iters[d] = it
        total_ll += ll
        total_words_ll += wll
        sstats[:, ids] += phi_c.T                                    # :207
    return {
        "document_log_likelihood": total_ll,
        "words_log_likelihood": total_words_ll,
        "sstats": sstats,
        "gamma": gamma,
        "doc_ll": doc_ll,
        "doc_words_ll": words_ll,
        "iters": iters,
    }


def m_step(eta, alpha_beta, sstats, gamma):
    """M-step (variational_bayes.py:218-235).

    Topic log-likelihood uses the PRE-update eta (:222-224), then
    eta <- sstats + beta (:226) and the alpha sufficient statistics come from
    the gamma the E-step just stored (:232-233).
    Returns (topic_log_likelihood, alpha_sufficient_statistics (K,), new_eta).
    """
    K = eta.shape[0]
    topic_ll = K * (gammaln(np.sum(alpha_beta)) - np.sum(gammaln(alpha_beta)))   # :222
    topic_ll += np.sum(np.sum(gammaln(eta), axis=1) - gammaln(np.sum(eta, axis=1)))  # :224
    new_eta = sstats + alpha_beta                                    # :226
    alpha_ss = psi(gamma) - psi(np.sum(gamma, axis=1)[:, None])      # :232
    alpha_ss = np.sum(alpha_ss, axis=0)                              # :233
    return float(topic_ll), alpha_ss, new_eta


def optimize_hyperparameters(alpha, alpha_ss, number_of_documents,
                             hyper_parameter_iteration=100,
                             hyper_parameter_decay_factor=0.9,
                             hyper_parameter_maximum_decay=10,
                             hyper_parameter_converge_threshold=1e-6):
    """Newton update of alpha (variational_bayes.py:277-324).

    Reproduces the reference's quirk: `1/hessian` is kept as a VECTOR where
    the textbook form has a sum (:292), so the correction `c` is a vector.
    """
    alpha = np.array(alpha, dtype=np.float64)
    D = number_of_documents
    alpha_update = alpha
    decay = 0
    for _ in range(hyper_parameter_iteration):
        alpha_sum = np.sum(alpha)
        gradient = D * (psi(alpha_sum) - psi(alpha)) + alpha_ss       # :285
        hessian = -D * polygamma(1, alpha)                            # :286
        sum_g_h = np.sum(gradient / hessian)                          # :291
        sum_1_h = 1.0 / hessian                                       # :292 (vector!)
        z = D * polygamma(1, alpha_sum)                               # :294
        c = sum_g_h / (1.0 / z + sum_1_h)                             # :295
        while True:                                                   # :298-315
            step = np.power(hyper_parameter_decay_factor, decay) * (gradient - c) / hessian
            if np.any(alpha <= step):
                decay += 1
                if decay > hyper_parameter_maximum_decay:
                    break
            else:
                alpha_update = alpha - step
                break
        mean_change = np.mean(np.abs(alpha_update - alpha))           # :319
        alpha = alpha_update
        if mean_change <= hyper_parameter_converge_threshold:
            break
    return alpha


def lists_to_csr(word_ids, word_cts):
    """Reference corpus container (two lists, :120-121) -> CSR arrays."""
    D = len(word_ids)
    doc_ptr = np.zeros(D + 1, dtype=np.int64)
    for d in range(D):
        doc_ptr[d + 1] = doc_ptr[d] + len(word_ids[d])
    term_id = np.concatenate([np.asarray(w).ravel() for w in word_ids]).astype(np.int32) \
        if D else np.zeros(0, np.int32)
    term_ct = np.concatenate([np.asarray(c).ravel() for c in word_cts]).astype(np.int32) \
        if D else np.zeros(0, np.int32)
    return doc_ptr, term_id, term_ct
