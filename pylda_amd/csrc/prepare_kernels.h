// Per-outer-iteration table preparation: eta (K x V, numpy layout) ->
// word-major tables the document kernels gather from.
//
// Reference: compute_dirichlet_expectation (inferencer.py:15-18), called at
// variational_bayes.py:152; the held-out normaliser of :155.
//
//   Elog[w][k]    = psi(eta[k][w]) - psi(sum_v eta[k][v]) - shift[w]
//   shift[w]      = max_k (psi(eta[k][w]) - psi(sum_v eta[k][v]))
//   expElog[w][k] = exp(Elog[w][k])                   in (0, 1], row max == 1
//   topic_lse[k]  = logsumexp_v E_log_eta[k][v]       (held-out only)
//
// The per-word shift is what keeps the linear-space inner loop inside the
// fp64 range: E_log_eta spans [-V, 0] (psi(1/V) ~ -V for unseen pairs).
#pragma once
#include "estep_common.h"
#include "special_device.h"

namespace pylda {

// psi_rowsum[k] = psi(sum_v eta[k][v]); one workgroup per topic row.
__global__ __launch_bounds__(256) void eta_rowsum_psi_kernel(const double* __restrict__ eta,
                                                             int K, int V,
                                                             double* __restrict__ psi_rowsum)
{
    __shared__ double scratch[4];
    const int k = blockIdx.x;
    const double* row = eta + (size_t)k * V;
    double s = 0.0;
    for (int v = threadIdx.x; v < V; v += 256) s += row[v];
    s = block_sum<256>(s, scratch);
    if (threadIdx.x == 0) psi_rowsum[k] = digamma(s);
}

// Transposing pass: reads eta coalesced along v, writes the un-shifted
// E_log_eta word-major, coalesced along k, through a 32x33 LDS tile.
__global__ __launch_bounds__(256) void elog_transpose_kernel(const double* __restrict__ eta,
                                                             const double* __restrict__ psi_rowsum,
                                                             int K, int V, int ldk,
                                                             double* __restrict__ elog_wk)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int v0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + ty + j * 8, v = v0 + tx;
        if (k < K && v < V) tile[ty + j * 8][tx] = digamma(eta[(size_t)k * V + v]) - psi_rowsum[k];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int v = v0 + ty + j * 8, k = k0 + tx;
        if (k < K && v < V) elog_wk[(size_t)v * ldk + k] = tile[tx][ty + j * 8];
    }
}

// One wavefront per word row: max-shift and exponentiate in place.
__global__ __launch_bounds__(256) void row_shift_exp_kernel(double* __restrict__ elog_wk, int K,
                                                            int V, int ldk,
                                                            double* __restrict__ expElog,
                                                            double* __restrict__ expElog_elog,
                                                            double* __restrict__ shift)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int w = blockIdx.x * 4 + threadIdx.x / kWave;
    if (w >= V) return;
    double* row = elog_wk + (size_t)w * ldk;
    double m = -INFINITY;
    for (int k = lane; k < K; k += kWave) m = fmax(m, row[k]);
    m = wave_max(m);
    for (int k = lane; k < ldk; k += kWave) {
        const bool real = k < K;                     // columns K..ldk-1 are zero padding
        const double e = real ? row[k] - m : 0.0;
        row[k] = e;
        const double b = real ? exp(e) : 0.0;
        expElog[(size_t)w * ldk + k] = b;
        expElog_elog[(size_t)w * ldk + k] = b > 0.0 ? b * e : 0.0;    // 0 * (-745..) -> 0, as exp(lp)*lp does at :199
    }
    if (lane == 0) shift[w] = m;
}

// The two passes above in one for a SMALL table with K <= 64 (associated-press K = 10: 68 000 entries): one wavefront per
// word, lane k reads eta[k][w] straight from the topic-major matrix (the strided read costs nothing at this size) -
// one kernel boundary less in an E-step that is a chain of them.  The same operations in the same order: bitwise the
// tables of the two-pass path.
__global__ __launch_bounds__(256) void elog_rows_small_kernel(const double* __restrict__ eta,
                                                              const double* __restrict__ psi_rowsum, int K, int V, int ldk,
                                                              double* __restrict__ elog_wk, double* __restrict__ expElog,
                                                              double* __restrict__ expElog_elog, double* __restrict__ shift)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int w = blockIdx.x * 4 + threadIdx.x / kWave;
    if (w >= V) return;
    const bool real = lane < K;
    const double raw = real ? digamma(eta[(size_t)lane * V + w]) - psi_rowsum[lane] : -INFINITY;
    const double m = wave_max(raw);
    if (lane < ldk) {
        const double e = real ? raw - m : 0.0;
        const double b = real ? exp(e) : 0.0;
        elog_wk[(size_t)w * ldk + lane] = e;
        expElog[(size_t)w * ldk + lane] = b;
        expElog_elog[(size_t)w * ldk + lane] = b > 0.0 ? b * e : 0.0;
    }
    if (lane == 0) shift[w] = m;
}

// topic_lse[k] = logsumexp_v (Elog[v][k] + shift[v]); one workgroup per topic.
__global__ __launch_bounds__(256) void topic_lse_kernel(const double* __restrict__ elog_wk,
                                                        const double* __restrict__ shift, int K,
                                                        int V, int ldk,
                                                        double* __restrict__ topic_lse)
{
    __shared__ double scratch[4];
    const int k = blockIdx.x;
    double m = -INFINITY;
    for (int v = threadIdx.x; v < V; v += 256) m = fmax(m, elog_wk[(size_t)v * ldk + k] + shift[v]);
    m = block_max<256>(m, scratch);
    double s = 0.0;
    for (int v = threadIdx.x; v < V; v += 256) s += exp(elog_wk[(size_t)v * ldk + k] + shift[v] - m);
    s = block_sum<256>(s, scratch);
    if (threadIdx.x == 0) topic_lse[k] = m + log(s);
}

// Deterministic sum of a length-n vector into out[0] (single workgroup).
__global__ __launch_bounds__(1024) void vector_sum_kernel(const double* __restrict__ x, int64_t n,
                                                          double* __restrict__ out)
{
    __shared__ double scratch[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += x[i];
    s = block_sum<1024>(s, scratch);
    if (threadIdx.x == 0) out[0] = s;
}

// The corpus-level sums of an E-step (document log-likelihood, words log-likelihood, entropy partials of the statistics
// pass) and the count of documents the log-space safety net redid (status 2) in ONE launch: workgroup b does job b (a
// small corpus' E-step is a chain of kernel boundaries).
struct SumJob {
    const double* x;
    int64_t n;
    double* out;
};
__global__ __launch_bounds__(1024) void vector_sum3_kernel(SumJob a, SumJob b, SumJob c, const int32_t* __restrict__ status,
                                                           int64_t D, int32_t* __restrict__ redone)
{
    __shared__ double scratch[16];
    if (blockIdx.x == 3) {
        double n = 0.0;
        for (int64_t i = threadIdx.x; i < D; i += 1024) n += status[i] == 2 ? 1.0 : 0.0;
        n = block_sum<1024>(n, scratch);                  // (exact: integers below 2^53)
        if (threadIdx.x == 0) redone[0] = (int32_t)n;
        return;
    }
    const SumJob job = blockIdx.x == 0 ? a : blockIdx.x == 1 ? b : c;
    if (!job.out) return;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < job.n; i += 1024) s += job.x[i];
    s = block_sum<1024>(s, scratch);
    if (threadIdx.x == 0) job.out[0] = s;
}

// alpha with the sign bit set where the topic may never count as dead (estep_common.h kMortalT): its t at gamma = alpha,
// exp(psi(alpha_k) - psi(sum gamma)), is not negligible in a short document (sum gamma = sum alpha + kMortalTokens).
// One workgroup, once per E-step.
__global__ __launch_bounds__(256) void alpha_mortality_kernel(const double* __restrict__ alpha, int K, double* __restrict__ alpha_sgn)
{
    __shared__ double scratch[4];
    double a = 0.0;
    for (int k = threadIdx.x; k < K; k += 256) a += alpha[k];
    const double psi_shortest = digamma(block_sum<256>(a, scratch) + kMortalTokens);
    for (int k = threadIdx.x; k < K; k += 256) {
        const double ak = alpha[k];
        alpha_sgn[k] = exp_digamma_minus(ak, psi_shortest) < kMortalT ? ak : -ak;
    }
}

}  // namespace pylda
