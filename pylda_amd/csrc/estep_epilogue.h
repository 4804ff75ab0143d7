// The tail every register / streaming document kernel ends with (estep_quad.h, estep_quilt.h, estep_qgroup.h,
// estep_qfuse.h, estep_qfusek.h): the per-topic share of the document terms, the sums over the workgroup and the
// document's two likelihood values (variational_bayes.py:195-204).  One copy instead of five: every operation and
// every summation order is what each of the kernels had of its own, so results are bitwise what they were.
#pragma once
#include "estep_common.h"
#include "special_device.h"

namespace pylda {

struct TopicShare {
    double term2 = 0.0;         // sum_k log t_k (gamma_k - alpha_k)                               (:199, second identity)
    double lse_term = 0.0;      // sum_k logsumexp_v E_log_eta[k] (gamma_k - alpha_k)              (:204, held-out)
    double lgam = 0.0;          // sum_k lnG(gamma_k)                                              (:197)
    double gsum = 0.0;          // sum_k gamma_k
};

// The thread that owns topic k adds its share and writes gamma_k / t_k (`live`: k < K; `owns` without `live`: a padding
// column of the table stride, whose t is written as 0 for the statistics pass).  t_last is t of the last executed
// iteration, gam_prev the gamma it was computed from (log t_k = psi(gam_prev) - psi(sum gamma)).
__device__ __forceinline__ void topic_share(const EstepParams& p, int doc, int k, int ldk, bool live, bool owns, double gam, double alpha_k,
                                            double gam_prev, double t_last, double psi_total, TopicShare& s)
{
    if (live) {
        const double mass = gam - alpha_k;                                // = t_last * sum_n r_n B[w_n][k]
        const double ltv = digamma(gam_prev) - psi_total;
        s.term2 = fma(ltv, mass, s.term2);
        if (p.heldout) s.lse_term = fma(p.topic_lse[k], mass, s.lse_term);
        s.lgam += lgamma_pos(gam);
        s.gsum += gam;
        p.gamma[(size_t)doc * p.K + k] = gam;
        if (!p.heldout) p.tfinal[(size_t)doc * ldk + k] = t_last;
    } else if (owns && !p.heldout) {
        p.tfinal[(size_t)doc * ldk + k] = 0.0;
    }
}

// term1: sum_n r_n sum_k G[w_n][k] t_k, term3: sum_n c_n log(normaliser_n), shift_term: sum_n c_n shift[w_n] - each thread's
// partial.  misc: 7 * W doubles of LDS.  Thread 0 writes the document's values; `it` inner iterations were executed.
template <int W>
__device__ __forceinline__ void finish_document(const EstepParams& p, int doc, int it, double* misc, int lane, int wave, int tid,
                                                double term1, double term3, double shift_term, TopicShare s)
{
    term1 = wave_sum(term1);
    s.term2 = wave_sum(s.term2);
    s.lse_term = wave_sum(s.lse_term);
    s.lgam = wave_sum(s.lgam);
    s.gsum = wave_sum(s.gsum);
    term3 = wave_sum(term3);
    shift_term = wave_sum(shift_term);
    __syncthreads();                                        // (misc may still be read by a slower wavefront's prologue sums)
    if (lane == 0) {
        misc[0 * W + wave] = term1;
        misc[1 * W + wave] = s.term2;
        misc[2 * W + wave] = s.lse_term;
        misc[3 * W + wave] = s.lgam;
        misc[4 * W + wave] = s.gsum;
        misc[5 * W + wave] = term3;
        misc[6 * W + wave] = shift_term;
    }
    __syncthreads();
    if (tid == 0) {
        double t1 = 0.0, t2 = 0.0, tl = 0.0, lg = 0.0, gs = 0.0, t3 = 0.0, sh = 0.0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            t1 += misc[0 * W + w];
            t2 += misc[1 * W + w];
            tl += misc[2 * W + w];
            lg += misc[3 * W + w];
            gs += misc[4 * W + w];
            t3 += misc[5 * W + w];
            sh += misc[6 * W + w];
        }
        const double ent = t1 + t2 - t3;
        p.doc_ll[doc] = p.alpha_term + lg - lgamma_pos(gs) - ent;        // :195-199
        p.doc_words_ll[doc] = p.heldout ? t1 + sh - tl : 0.0;            // :204
        p.iters[doc] = it;
        p.status[doc] = 0;
    }
}

// Hand-over of a document from a STREAMING dense kernel (estep_qfuse.h, estep_qfusek.h) to the live-topic kernel
// (estep_compact.h): gamma after `it` updates and the live topics in ascending order - no tile, these kernels keep none
// on chip (the live-topic kernel gathers its N x L entries from the table).  Thread tid owns topic tid (tid < padded K);
// `counts`: one unsigned per wavefront of the workgroup in LDS.  Every thread of the workgroup calls it.
__device__ __forceinline__ void stream_hand_over(const EstepParams& p, int doc, int tid, bool topic_live, double gam, double alpha_k, unsigned* counts,
                                                 int waves, int it)
{
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    const bool alive = topic_live && gam != alpha_k;
    const unsigned long long mask = __ballot(alive);
    if (lane == 0) counts[wave] = (unsigned)__builtin_popcountll(mask);
    __syncthreads();
    int at = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) at += (int)counts[w];
    if (alive) *live_idx_at(live_list_of(p.live_list, doc), at) = (uint16_t)tid;
    if (topic_live) p.gamma[(size_t)doc * p.K + tid] = gam;
    if (tid == 0) {
        unsigned total = 0;
        for (int w = 0; w < waves; ++w) total += counts[w];
        p.live_n[doc] = (int)total;
        p.handoff_it[doc] = it;
        p.iters[doc] = it;
        p.status[doc] = 4;
    }
}

}  // namespace pylda
