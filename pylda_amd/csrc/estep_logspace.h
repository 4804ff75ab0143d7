// Log-space E-step kernel: the reference's formulation, evaluated on the
// device without the exp-hoisting.  It is the safety net of the hot path:
// a document whose linear-space normaliser leaves the fp64 range (possible
// only when some alpha_k has collapsed below ~1e-3, so that
// exp(psi(gamma_k) - max psi) underflows for the very topic that carries a
// word) is flagged by the fast kernels and redone here.  It recomputes
// log phi = E_log_eta[:, w] + psi(gamma) - logsumexp(...) every inner
// iteration exactly as variational_bayes.py:177-185 does (N_d*K exps per
// iteration), so it is slow and never on the measured path.
//
// One workgroup (4 wavefronts) per document; a wavefront owns a word at a
// time, lanes stride over topics.  Its sufficient statistics have no t*r
// factorisation in linear space: it runs AFTER the gather pass and adds them
// (fp64 atomics, rare) straight into the finished V x ldk table.
#pragma once
#include "estep_common.h"
#include "special_device.h"
#include "estep_limits.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace pylda {

__device__ inline void logspace_document(const EstepParams& p, const double* __restrict__ elog_wk,
                                         double* __restrict__ sstats_extra, int doc, char* smem)
{
    constexpr int NT = 256, NW = 4;
    const int K = p.K;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int64_t lo = p.doc_ptr[doc];
    const int N = (int)(p.doc_ptr[doc + 1] - lo);

    double* psi = reinterpret_cast<double*>(smem);
    double* gam = psi + K;
    double* gacc = gam + K;                 // NW x K
    double* scratch = gacc + NW * K;

    double local = 0.0;
    for (int n = tid; n < N; n += NT) local += (double)p.term_ct[lo + n];
    const double total = block_sum<NT>(local, scratch);                        // :162
    for (int k = tid; k < K; k += NT) gam[k] = p.alpha[k] + total / K;          // :165
    __syncthreads();

    int it = 0;
    while (it < p.max_iter) {                                                   // :174
        for (int k = tid; k < K; k += NT) psi[k] = digamma(gam[k]);
        for (int k = lane; k < K; k += kWave) gacc[wave * K + k] = 0.0;
        __syncthreads();
        for (int n = wave; n < N; n += NW) {
            const double* row = elog_wk + (size_t)p.term_id[lo + n] * p.ldk;
            double m = -INFINITY;
            for (int k = lane; k < K; k += kWave) m = fmax(m, row[k] + psi[k]);   // :177
            m = wave_max(m);
            double s = 0.0;
            for (int k = lane; k < K; k += kWave) s += exp(row[k] + psi[k] - m);
            s = wave_sum(s);
            const double lse = m + log(s);                                      // :182
            const double lc = log((double)p.term_ct[lo + n]);
            for (int k = lane; k < K; k += kWave)
                gacc[wave * K + k] += exp(row[k] + psi[k] - lse + lc);          // :185
        }
        __syncthreads();
        double diff = 0.0;
        for (int k = tid; k < K; k += NT) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += gacc[w * K + k];
            const double gnew = p.alpha[k] + s;
            diff += fabs(gnew - gam[k]);                                        // :187
            gam[k] = gnew;                                                      // :188
        }
        const double change = block_sum<NT>(diff, scratch) / K;
        ++it;
        __syncthreads();
        if (change <= p.tol) break;                                             // :189
    }

    // final pass: psi[] still holds the digammas of the pre-update gamma.
    double ent = 0.0, wll = 0.0;
    for (int n = wave; n < N; n += NW) {
        const int id = p.term_id[lo + n];
        const double* row = elog_wk + (size_t)id * p.ldk;
        double m = -INFINITY;
        for (int k = lane; k < K; k += kWave) m = fmax(m, row[k] + psi[k]);
        m = wave_max(m);
        double s = 0.0;
        for (int k = lane; k < K; k += kWave) s += exp(row[k] + psi[k] - m);
        s = wave_sum(s);
        const double lse = m + log(s);
        const double c = (double)p.term_ct[lo + n];
        const double lc = log(c);
        const double sh = p.heldout ? p.shift[id] : 0.0;
        for (int k = lane; k < K; k += kWave) {
            const double lp = row[k] + psi[k] - lse;
            ent = fma(c, exp(lp) * lp, ent);                                    // :199
            const double pc = exp(lp + lc);
            if (p.heldout) wll = fma(pc, row[k] + sh - p.topic_lse[k], wll);     // :204
            else if (pc != 0.0) unsafeAtomicAdd(&sstats_extra[(size_t)id * p.ldk + k], pc);   // :207
        }
    }
    ent = block_sum<NT>(ent, scratch);
    wll = block_sum<NT>(wll, scratch);
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
        p.doc_ll[doc] = p.alpha_term + lg - lgamma_pos(gs) - ent;              // :195-199
        p.doc_words_ll[doc] = wll;
        p.iters[doc] = it;
        p.status[doc] = 2;        // finished by the log-space kernel
    }
}

// Documents flagged (status == 1) by the fast kernels are found by this kernel itself: the workgroups stride over the
// status array in chunks of up to 256 documents, collect a chunk's flagged documents in LDS and redo them one after the other
// (the whole workgroup works on a document).  Neither the host nor another kernel builds a list: with nothing flagged -
// every E-step of every BASELINE configuration - the safety net costs ONE dispatch of a few loads per thread.
//   smem: logspace_lds_bytes(K) bytes for a document, then 256 + 1 ints for the chunk's list.
// `chunk` (<= 256) documents per workgroup and step: 256 for a large corpus; a small one (associated-press: 2000 documents)
// is cut finer so that a step with many flagged documents still spreads over the chip (logspace_chunk below).
__global__ __launch_bounds__(256) void estep_logspace_kernel(EstepParams p,
                                                             const double* __restrict__ elog_wk,
                                                             double* __restrict__ sstats_extra,
                                                             const int32_t* __restrict__ status, int64_t D,
                                                             size_t list_offset, int chunk)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* found = reinterpret_cast<int32_t*>(smem + list_offset);
    int32_t* nfound = found + 256;
    for (int64_t base = (int64_t)blockIdx.x * chunk; base < D; base += (int64_t)gridDim.x * chunk) {
        if (threadIdx.x == 0) *nfound = 0;
        __syncthreads();
        const int64_t d = base + threadIdx.x;
        if ((int)threadIdx.x < chunk && d < D && status[d] == 1) found[atomicAdd(nfound, 1)] = (int32_t)d;
        __syncthreads();
        const int n = *nfound;
        for (int i = 0; i < n; ++i) {                  // (documents are independent: the order inside a chunk does not matter)
            logspace_document(p, elog_wk, sstats_extra, found[i], smem);
            __syncthreads();
        }
    }
}

}  // namespace pylda
