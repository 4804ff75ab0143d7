// Generic E-step kernel: one workgroup per document, any K, any N_d.
//
// Implements variational_bayes.py:162-207 of the reference for one document
// per workgroup in the exp-hoisted ("linear space") form:
//
//   B[n][k] = exp(E_log_eta[k][w_n] - shift[w_n])       (table row, gathered once)
//   t[k]    = exp(psi(gamma_k) - max_k psi(gamma))      (:177, K digammas)
//   nrm[n]  = sum_k B[n][k] t[k]                        (:182 logsumexp, linear form)
//   gamma'  = alpha + t[k] * sum_n (c_n / nrm[n]) B[n][k]   (:185)
//
// which is algebraically the reference's log-space update with the
// per-word shift and the psi-max factored out of the normalisation.  The
// final pass uses phi from the LAST EXECUTED iteration (half a step behind
// gamma, :177-188) for the entropy term (:199), the held-out word
// likelihood (:204) and the sufficient statistics (:207).
//
// MODE 0: the N_d x K tile of B is staged once in LDS (odd row stride =>
// conflict-free ds_read_b64 in both passes) and every inner iteration runs
// out of LDS.  MODE 1: the tile does not fit the 160 KiB LDS; rows are
// re-read from the table (L2 / Infinity Cache).  MODE 2: neither do the
// per-term scalars (28 bytes per distinct term: above ~5,000 terms) - r_n
// lives in the corpus' rfinal array, the normalisers in a per-corpus scratch
// array, ids and counts are read where they lie: a document of ANY length
// runs (the reference accepts any, variational_bayes.py:98-130), slowly.
#pragma once
#include "estep_common.h"
#include "special_device.h"
#include "estep_limits.h"

namespace pylda {

template <int NT, int MODE>
__global__ __launch_bounds__(NT) void estep_generic_kernel(EstepParams p)
{
    constexpr bool TILE_GLOBAL = MODE >= 1;
    constexpr bool TERMS_GLOBAL = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = p.K;
    const int tid = threadIdx.x;
    const int doc = p.order[blockIdx.x];
    const int64_t lo = p.doc_ptr[doc];
    const int N = (int)(p.doc_ptr[doc + 1] - lo);
    const int stride = TILE_GLOBAL ? p.ldk : p.tile_stride;

    const GenericLds L = generic_lds_layout(K, TERMS_GLOBAL ? 0 : p.n_cap, p.tile_stride, NT, TILE_GLOBAL);
    double* tile = reinterpret_cast<double*>(smem + L.tile);
    double* t = reinterpret_cast<double*>(smem + L.t);
    double* lt = reinterpret_cast<double*>(smem + L.lt);
    double* gam = reinterpret_cast<double*>(smem + L.gam);
    // per-term scalars: LDS, or (MODE 2) global memory - written and read by this workgroup only, between
    // __syncthreads() (workgroup-scope release / acquire; one CU, one L1)
    double* r = TERMS_GLOBAL ? p.rfinal + lo : reinterpret_cast<double*>(smem + L.r);
    double* lognrm = TERMS_GLOBAL ? p.term_scratch + lo : reinterpret_cast<double*>(smem + L.lognrm);
    double* cts = reinterpret_cast<double*>(smem + L.cts);
    int* ids = reinterpret_cast<int*>(smem + L.ids);
    auto id_of = [&](int n) -> int { if constexpr (TERMS_GLOBAL) return p.term_id[lo + n]; else return ids[n]; };
    auto ct_of = [&](int n) -> double { if constexpr (TERMS_GLOBAL) return (double)p.term_ct[lo + n]; else return cts[n]; };
    double* red = reinterpret_cast<double*>(smem + L.red);
    double* scratch = reinterpret_cast<double*>(smem + L.scratch);

    // lane mapping shared by pass 2 and the final pass: KL lanes along topics
    // (contiguous k => coalesced table reads / atomics, conflict-free LDS),
    // G = NT / KL groups along words.
    int KL = 1;
    while (KL < K && KL < NT) KL <<= 1;
    const int G = NT / KL;
    const int kl = tid & (KL - 1);
    const int g = tid / KL;

    // ---- stage ids / counts, total token count (:162) ----
    double local = 0.0;
    for (int n = tid; n < N; n += NT) {
        const int id = p.term_id[lo + n];
        const double c = (double)p.term_ct[lo + n];
        if constexpr (!TERMS_GLOBAL) {
            ids[n] = id;
            cts[n] = c;
        }
        local += c;
    }
    const double total = block_sum<NT>(local, scratch);
    __syncthreads();

    // ---- gather the B tile: rows are contiguous K doubles in the table ----
    if constexpr (!TILE_GLOBAL) {
        for (int n = g; n < N; n += G) {
            const double* src = p.expElog + (size_t)ids[n] * p.ldk;
            double* dst = tile + (size_t)n * stride;
            for (int k = kl; k < K; k += KL) dst[k] = src[k];
        }
    }
    for (int k = tid; k < K; k += NT) gam[k] = p.alpha[k] + total / K;       // :165
    __syncthreads();

    auto row = [&](int n) -> const double* {
        if constexpr (TILE_GLOBAL) return p.expElog + (size_t)id_of(n) * p.ldk;
        else return tile + (size_t)n * stride;
    };

    int it = 0;
    int bad = 0;
    while (it < p.max_iter) {                                                 // :174
        // t[k] = exp(psi(gamma_k) - max psi)
        double lmax = -INFINITY;
        for (int k = tid; k < K; k += NT) {
            const double ps = digamma(gam[k]);
            lt[k] = ps;
            lmax = fmax(lmax, ps);
        }
        const double pmax = block_max<NT>(lmax, scratch);
        for (int k = tid; k < K; k += NT) {
            const double d = lt[k] - pmax;
            lt[k] = d;
            t[k] = exp(d);
        }
        __syncthreads();

        // pass 1: nrm[n] = B[n][:] . t   (lane <-> word)
        for (int n = tid; n < N; n += NT) {
            const double* b = row(n);
            double a0 = 0.0, a1 = 0.0;
            int k = 0;
            for (; k + 1 < K; k += 2) {
                a0 = fma(b[k], t[k], a0);
                a1 = fma(b[k + 1], t[k + 1], a1);
            }
            if (k < K) a0 = fma(b[k], t[k], a0);
            const double nrm = a0 + a1;
            if (!(nrm > 1e-280 && nrm < 1e300)) bad = 1;
            r[n] = ct_of(n) / nrm;
            lognrm[n] = nrm;          // the log is taken once, after the loop
        }
        __syncthreads();

        // pass 2: s[k] = sum_n r[n] B[n][k]   (lane <-> topic, groups over words)
        for (int kk = kl; kk < K; kk += KL) {
            double a0 = 0.0, a1 = 0.0;
            int n = g;
            for (; n + G < N; n += 2 * G) {
                a0 = fma(r[n], row(n)[kk], a0);
                a1 = fma(r[n + G], row(n + G)[kk], a1);
            }
            if (n < N) a0 = fma(r[n], row(n)[kk], a0);
            red[g * K + kk] = a0 + a1;
        }
        __syncthreads();
        double diff = 0.0;
        for (int k = tid; k < K; k += NT) {
            double s = 0.0;
            for (int gg = 0; gg < G; ++gg) s += red[gg * K + k];
            const double gnew = fma(t[k], s, p.alpha[k]);                     // :185
            diff += fabs(gnew - gam[k]);                                      // :187
            gam[k] = gnew;                                                    // :188
        }
        const double change = block_sum<NT>(diff, scratch) / K;
        ++it;
        __syncthreads();
        if (change <= p.tol) break;                                           // :189
    }

    // A document whose linear-space normaliser left the fp64 range is not
    // finished here: it is flagged and redone by the log-space kernel.
    bad = __syncthreads_or(bad);
    if (bad) {
        if (!p.heldout) {      // contributes nothing to the gather pass; the log-space kernel adds it
            for (int n = tid; n < N; n += NT) p.rfinal[lo + n] = 0.0;
            for (int k = tid; k < p.ldk; k += NT) p.tfinal[(size_t)doc * p.ldk + k] = 0.0;
        }
        if (tid == 0) p.status[doc] = 1;
        return;
    }

    // ---- final pass with the last phi ----
    for (int n = tid; n < N; n += NT) lognrm[n] = log(lognrm[n]);
    __syncthreads();
    double ent = 0.0, wll = 0.0;
    for (int n = g; n < N; n += G) {
        const double* b = row(n);
        const double rn = r[n], ln = lognrm[n];
        const double sh = p.heldout ? p.shift[id_of(n)] : 0.0;
        for (int kk = kl; kk < K; kk += KL) {
            const double bv = b[kk];
            const double pc = bv * t[kk] * rn;             // phi * count
            ent = fma(pc, lt[kk] - ln, ent);               // :199 without the log B part
            if (bv > 0.0 && (p.heldout || p.want_doc_ll)) {
                const double lb = log(bv);
                ent = fma(pc, lb, ent);                    // :199, log B part
                if (p.heldout) wll = fma(pc, lb + sh - p.topic_lse[kk], wll);   // :204
            }
        }
    }
    ent = block_sum<NT>(ent, scratch);
    wll = block_sum<NT>(wll, scratch);
    // The sufficient statistics (:207) are phi*count = B[w][k] * t[k] * r[n]: the
    // factors t (per document) and r (per term) are handed to the gather pass
    // (sstats_kernels.h), which sums them word by word without atomics.
    if (!p.heldout) {
        if constexpr (!TERMS_GLOBAL)
            for (int n = tid; n < N; n += NT) p.rfinal[lo + n] = r[n];
        for (int k = tid; k < p.ldk; k += NT) p.tfinal[(size_t)doc * p.ldk + k] = k < K ? t[k] : 0.0;
    }

    double lg = 0.0, gs = 0.0;
    for (int k = tid; k < K; k += NT) {
        const double gk = gam[k];
        p.gamma[(size_t)doc * K + k] = gk;
        lg += lgamma_pos(gk);
        gs += gk;
    }
    lg = block_sum<NT>(lg, scratch);
    gs = block_sum<NT>(gs, scratch);
    if (tid == 0) {
        p.doc_ll[doc] = p.alpha_term + lg - lgamma_pos(gs) - ent;            // :195-199
        p.doc_words_ll[doc] = wll;
        p.iters[doc] = it;
        p.status[doc] = 0;
    }
}

}  // namespace pylda
