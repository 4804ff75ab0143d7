// Document terms of the likelihood (variational_bayes.py:195-199) for the documents whose E-step kernel left
// them out (status 3), as a pass of its own.
//
// The register kernels hold a CU's whole register file per document (or half of it): their epilogue - K
// lgamma, K + 2 N_d logarithms on a handful of active wavefronts, ~8000 cycles - costs the same slot time as
// one and a half inner iterations while the tile registers sit idle.  On the training fast path (corpus-level
// likelihood only, EstepParams::want_doc_ll == 0) everything the epilogue needs is in memory anyway for the
// statistics pass: gamma, t of the last iteration (tfinal) and r_n = c_n / normaliser_n (rfinal).  Here one
// wavefront per document recomputes the terms from those at full occupancy (cfg 3: 0.1 ms against 0.65 ms
// inside the document kernels).
//
//   doc_ll[d] = alpha_term + sum_k lnG(gamma_k) - lnG(sum_k gamma_k)
//               - ( sum_k log t_k (gamma_k - alpha_k) - sum_n c_n (log c_n - log r_n) )
// (the remaining term of :199, sum_n c_n sum_k phi log B, comes once per corpus from the statistics pass;
// log t_k = psi(gamma_k before the last update) - psi(sum gamma) is taken from the stored t_k.)
#pragma once
#include "estep_common.h"
#include "special_device.h"

namespace pylda {

// ONE launch per E-step over all D documents (p.order = nullptr, count = D), behind the join of the launch classes:
// on an auxiliary stream beside the dispatch-paced statistics gather (fp64-bound next to L2-bound), in front of the
// persistent sweep on the main stream (the sweep needs every CU to itself).  A launch per class right behind the
// class' kernel was measured and not kept.  (p.order is honoured for callers that pass a document list.)
__global__ __launch_bounds__(256) void doc_terms_kernel(EstepParams p, int64_t count)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t slot = (int64_t)blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
    if (slot >= count) return;
    const int64_t doc = p.order ? p.order[slot] : slot;
    if (p.status[doc] != 3) return;
    const int K = p.K;
    const double* gamma = p.gamma + (size_t)doc * K;
    const double* t = p.tfinal + (size_t)doc * p.ldk;
    double lgam = 0.0, gsum = 0.0, term2 = 0.0, term3 = 0.0;
    // a document the live-topic kernel finished keeps t as a list of its live topics; every other topic sits at alpha_k
    // EXACTLY (mass 0: no term), so the two sums over the K topics are the corpus constants plus the live topics' shares:
    //   sum_k lnG(gamma_k) = sum_k lnG(alpha_k) + sum_live (lnG(gamma_j) - lnG(alpha_j)),  sum_k gamma_k likewise
    // - a dozen lnG evaluations per document instead of K, and the 2-KiB gamma row is not read
    const int listed = p.live_stats ? p.live_n[doc] : -1;
    if (listed >= 0) {
        if (lane < listed) {
            char* list = live_list_of(p.live_list, doc);
            const int k = *live_idx_at(list, lane);
            const double g = gamma[k], a = p.alpha[k], mass = g - a;      // mass = t_k * sum_n r_n B[w_n][k]
            lgam = lgamma_pos(g) - lgamma_pos(a);
            gsum = mass;
            if (mass != 0.0) term2 = log(*live_t_at(list, lane)) * mass;  // (t_k may have underflowed where the mass did)
        }
    } else {
        for (int k = lane; k < K; k += kWave) {
            const double g = gamma[k], mass = g - p.alpha[k];
            lgam += lgamma_pos(g);
            gsum += g;
            if (mass != 0.0) term2 = fma(log(t[k]), mass, term2);
        }
    }
    const int64_t lo = p.doc_ptr[doc], hi = p.doc_ptr[doc + 1];
    for (int64_t n = lo + lane; n < hi; n += kWave) {
        const double c = (double)p.term_ct[n];
        term3 = fma(c, log(c) - log(p.rfinal[n]), term3);                 // c_n log(normaliser_n)
    }
    lgam = wave_sum(lgam);
    gsum = wave_sum(gsum);
    if (listed >= 0) {
        lgam += p.alpha_lgamma_sum;
        gsum += p.alpha_sum;
    }
    term2 = wave_sum(term2);
    term3 = wave_sum(term3);
    if (lane == 0) {
        p.doc_ll[doc] = p.alpha_term + lgam - lgamma_pos(gsum) - (term2 - term3);
        p.doc_words_ll[doc] = 0.0;
        p.status[doc] = 0;
    }
}

// Profiling: work[0] += sum_d I_d, work[1] += sum_d I_d N_d (inner iterations executed, and their terms), work[2] +=
// sum_d N_d (K x dense iterations + tile columns x live-topic iterations) (the tile entries the kernels really ran
// through the FMA pipes, twice per iteration), work[3] += documents handed to the live-topic kernel, work[4 .. 5] += this
// kernel's span in shader cycles and in ticks of the constant-rate counter (the live-topic kernel adds samples of its
// own) - single workgroup, so the accumulation over E-steps needs no atomics.
__global__ __launch_bounds__(1024) void work_count_kernel(const int32_t* __restrict__ iters, const int64_t* __restrict__ doc_ptr,
                                                          int64_t D, double* __restrict__ work, const int32_t* __restrict__ handoff_it,
                                                          const int32_t* __restrict__ col_iters, int K)
{
    __shared__ double scratch[16];
    const long long tick0 = clock64(), wall0 = wall_clock64();          // work[4], work[5]: this kernel's own span in both clocks
    double a = 0.0, b = 0.0, e = 0.0, h = 0.0;
    for (int64_t d = threadIdx.x; d < D; d += 1024) {
        const double it = (double)iters[d], n = (double)(doc_ptr[d + 1] - doc_ptr[d]);
        a += it;
        b += it * n;
        const int at = handoff_it ? handoff_it[d] : -1;
        if (at >= 0) {
            e += n * ((double)K * at + (double)col_iters[d]);
            h += 1.0;
        } else {
            e += n * (double)K * it;
        }
    }
    a = block_sum<1024>(a, scratch);
    b = block_sum<1024>(b, scratch);
    e = block_sum<1024>(e, scratch);
    h = block_sum<1024>(h, scratch);
    if (threadIdx.x == 0) {
        work[0] += a;
        work[1] += b;
        work[2] += e;
        work[3] += h;
        work[4] += (double)(clock64() - tick0);
        work[5] += (double)(wall_clock64() - wall0);
    }
}

}  // namespace pylda
