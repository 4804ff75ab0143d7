// Device M-step (variational_bayes.py:218-235) on the resident buffers.
#pragma once
#include "estep_common.h"
#include "special_device.h"
#include "estep_limits.h"

namespace pylda {

// Topic log-likelihood of the PRE-update eta (:224), two deterministic stages:
//   part[k][c] = (sum_{v in chunk c} lnG(eta[k][v]), sum_{v in chunk c} eta[k][v])     grid (K, kTopicChunks)
//   per_topic[k] = sum_c part.lg - lnG(sum_c part.s)
constexpr int kTopicChunks = 8;
__global__ __launch_bounds__(256) void mstep_topic_ll_kernel(const double* __restrict__ eta, int K,
                                                             int V, double* __restrict__ part)
{
    __shared__ double scratch[4];
    const int k = blockIdx.x, chunk = blockIdx.y;
    const int per = (V + kTopicChunks - 1) / kTopicChunks;
    const int v0 = chunk * per, v1 = min(V, v0 + per);
    const double* row = eta + (size_t)k * V;
    double lg = 0.0, s = 0.0;
    for (int v = v0 + threadIdx.x; v < v1; v += 256) {
        const double e = row[v];
        lg += lgamma_pos(e);
        s += e;
    }
    lg = block_sum<256>(lg, scratch);
    s = block_sum<256>(s, scratch);
    if (threadIdx.x == 0) {
        part[((size_t)k * kTopicChunks + chunk) * 2] = lg;
        part[((size_t)k * kTopicChunks + chunk) * 2 + 1] = s;
    }
}

__global__ __launch_bounds__(256) void mstep_topic_ll_finish_kernel(const double* __restrict__ part, int K,
                                                                    double* __restrict__ per_topic)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    double lg = 0.0, s = 0.0;
#pragma unroll
    for (int c = 0; c < kTopicChunks; ++c) {
        lg += part[((size_t)k * kTopicChunks + c) * 2];
        s += part[((size_t)k * kTopicChunks + c) * 2 + 1];
    }
    per_topic[k] = lg - lgamma_pos(s);
}

// eta[k][v] = sstats_wk[v][k] + beta[v]                              (:226)
__global__ __launch_bounds__(256) void mstep_update_eta_kernel(const double* __restrict__ sstats_wk,
                                                               const double* __restrict__ beta,
                                                               int K, int V, int ldk,
                                                               double* __restrict__ eta)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int k0 = blockIdx.x * 32, v0 = blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int v = v0 + ty + j * 8, k = k0 + tx;
        if (v < V && k < K) tile[ty + j * 8][tx] = sstats_wk[(size_t)v * ldk + k];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + ty + j * 8, v = v0 + tx;
        if (v < V && k < K) eta[(size_t)k * V + v] = tile[tx][ty + j * 8] + beta[v];
    }
}

// partial[b][k] = sum over this block's documents of psi(gamma_dk) - psi(sum_k gamma_dk)
// (:232-233).  One wavefront per document at a time; fixed document->block
// assignment and fixed summation order => bitwise reproducible.
__global__ __launch_bounds__(256) void mstep_alpha_ss_kernel(const double* __restrict__ gamma,
                                                             int64_t D, int K,
                                                             double* __restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* acc = reinterpret_cast<double*>(smem);        // 4 x K
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (int k = lane; k < K; k += kWave) acc[wave * K + k] = 0.0;
    // two documents per trip: the second one's gamma row is in flight while the first one's digammas run
    // (per document a chain of row fetch -> wavefront sum -> digamma; with 1024 workgroups = 4 wavefronts per
    // SIMD the whole device M-step went from 3.4 to 2.7 ms at cfg 4 - the rest is the 2.6e8 digammas themselves)
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t d = (int64_t)blockIdx.x * 4 + wave; d < D; d += 2 * stride) {
        const double* g0 = gamma + (size_t)d * K;
        const bool two = d + stride < D;
        const double* g1 = two ? gamma + (size_t)(d + stride) * K : g0;
        double s0 = 0.0, s1 = 0.0;
        for (int k = lane; k < K; k += kWave) {
            s0 += g0[k];
            s1 += g1[k];
        }
        s0 = wave_sum(s0);
        s1 = wave_sum(s1);
        const double ps0 = digamma(s0), ps1 = digamma(s1);
        for (int k = lane; k < K; k += kWave) {
            double a = acc[wave * K + k] + (digamma(g0[k]) - ps0);
            if (two) a += digamma(g1[k]) - ps1;
            acc[wave * K + k] = a;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256)
        partial[(size_t)blockIdx.x * K + k] = acc[k] + acc[K + k] + acc[2 * K + k] + acc[3 * K + k];
}

// out[k] = sum_b partial[b][k]: 64 topics per workgroup, 4 row groups (b mod 4) summed in a fixed order.
__global__ __launch_bounds__(256) void column_sum_kernel(const double* __restrict__ partial,
                                                         int nblocks, int K,
                                                         double* __restrict__ out)
{
    __shared__ double part[4][64];
    const int kk = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kk;
    double s = 0.0;
    if (k < K)
        for (int b = grp; b < nblocks; b += 4) s += partial[(size_t)b * K + k];
    part[grp][kk] = s;
    __syncthreads();
    if (grp == 0 && k < K) out[k] = (part[0][kk] + part[1][kk]) + (part[2][kk] + part[3][kk]);
}

// Everything the host half of learning() reads after one outer iteration (:244-252), packed for ONE copy:
//   out[0] document log-likelihood (training fast path: the corpus-level entropy term subtracted here, exactly the
//          host's former `sc[0] -= sc[2]`), out[1] #documents, out[2] documents redone in log space, out[3] 0,
//   out[4 .. 4+K) alpha sufficient statistics        - these K + 4 values are summed over the ranks -
//   out[4+K .. 4+2K) per-topic terms of the topic log-likelihood (:224; identical on every rank)
//   out[4+2K .. 4+3K) alpha (replicated): the Newton update of the outer iteration works on it in place
__global__ __launch_bounds__(256) void outer_pack_kernel(const double* __restrict__ scalars, const int32_t* __restrict__ flag_count,
                                                         int doc_values, double n_docs, const double* __restrict__ alpha_ss,
                                                         const double* __restrict__ per_topic, const double* __restrict__ alpha,
                                                         int K, double* __restrict__ out)
{
    if (threadIdx.x == 0) {
        out[0] = doc_values ? scalars[0] : scalars[0] - scalars[2];
        out[1] = n_docs;
        out[2] = (double)flag_count[0];
        out[3] = 0.0;
    }
    for (int k = threadIdx.x; k < K; k += 256) {
        out[4 + k] = alpha_ss[k];
        out[4 + K + k] = per_topic[k];
        out[4 + 2 * K + k] = alpha[k];          // the alpha of this iteration; alpha_newton_kernel updates it in place
    }
}

// The alpha update of learning() on the device: optimize_hyperparameters (variational_bayes.py:277-324), the
// reference's Newton iteration with a decaying step - including its element-wise 1 / hessian where Blei's closed form
// has a sum (:292-295), which end-to-end likelihood traces only match with.  One workgroup; alpha is K doubles, the
// iteration is a chain of K-wide steps with four reductions each (numpy's pairwise sums become fixed-order block sums:
// 1e-16 apart).  It runs between the (all-reduced) pack and the one read-back of the outer iteration, so that the
// host neither computes (0.1 ms at K = 10, 0.26 ms at K = 500 per iteration, scipy calls) nor uploads alpha.
//   io[0 .. K)      alpha: in, and out (also written to `alpha_device`, what the next E-step reads)
//   stats[0 .. K)   alpha sufficient statistics (:232-233), summed over the ranks;  docs: #documents (ditto)
//   work            4 K scratch doubles
__global__ __launch_bounds__(1024) void alpha_newton_kernel(double* __restrict__ io, const double* __restrict__ stats,
                                                            const double* __restrict__ docs_ptr, int K, NewtonParams np,
                                                            double* __restrict__ work, double* __restrict__ alpha_device)
{
    __shared__ double scratch[16];
    const int tid = threadIdx.x;
    const double docs = docs_ptr[0];
    double* alpha = work;                // current alpha (self._alpha_alpha)
    double* update = work + K;           // alpha_update: survives an iteration whose step was never accepted
    double* grad = work + 2 * (size_t)K;
    double* hess = work + 3 * (size_t)K;
    for (int k = tid; k < K; k += 1024) alpha[k] = update[k] = io[k];
    __syncthreads();
    int decay = 0;
    for (int it = 0; it < np.iterations; ++it) {
        double part = 0.0;
        for (int k = tid; k < K; k += 1024) part += alpha[k];
        const double alpha_sum = block_sum<1024>(part, scratch);                                  // :284
        const double psi_sum = digamma(alpha_sum);
        part = 0.0;
        for (int k = tid; k < K; k += 1024) {
            const double a = alpha[k];
            const double g = docs * (psi_sum - digamma(a)) + stats[k];                            // :285
            const double h = -docs * trigamma(a);                                                 // :286
            grad[k] = g;
            hess[k] = h;
            part += g / h;
        }
        const double sum_g_h = block_sum<1024>(part, scratch);                                    // :291
        const double z = docs * trigamma(alpha_sum);                                              // :294
        bool accepted_step = false;
        for (;;) {                                                                                // :298-315
            const double scale = np.decay_power[decay];
            int singular = 0;
            for (int k = tid; k < K; k += 1024) {
                const double c = sum_g_h / (1.0 / z + 1.0 / hess[k]);                             // :292-295 (vector c)
                const double step = scale * (grad[k] - c) / hess[k];                              // :301
                if (alpha[k] <= step) singular = 1;                                               // :305
            }
            if (__syncthreads_or(singular)) {
                decay += 1;
                if (decay > np.maximum_decay) break;
            } else {
                accepted_step = true;
                break;
            }
        }
        part = 0.0;
        for (int k = tid; k < K; k += 1024) {
            if (accepted_step) {
                const double c = sum_g_h / (1.0 / z + 1.0 / hess[k]);
                update[k] = alpha[k] - np.decay_power[decay] * (grad[k] - c) / hess[k];           // :308
            }
            part += fabs(update[k] - alpha[k]);                                                   // :319
            alpha[k] = update[k];                                                                 // :320
        }
        // (a step refused at every decay leaves alpha_update where it was: the change is 0 and the loop ends here,
        //  as in the reference)
        const double mean_change = block_sum<1024>(part, scratch) / K;
        if (mean_change <= np.threshold) break;                                                   // :321
    }
    __syncthreads();
    for (int k = tid; k < K; k += 1024) {
        io[k] = alpha[k];
        alpha_device[k] = alpha[k];
    }
}

}  // namespace pylda
