// libpylda_hip.so - device M-step, pack, alpha update and the one read-back of an outer iteration (variational_bayes.py:218-324).
// (host side of the C ABI declared in include/pylda_hip.h; see host_internal.h for the map of the translation units)
#include "host_internal.h"
#include "mstep_kernels.h"

extern "C" {

namespace {
// The device half of m_step (:218-235): kernels only, nothing is read back.
static int enqueue_mstep(pylda_ctx* ctx, pylda_corpus* c, const double* beta_v, bool want_alpha_ss, const char* who)
{
    if (!beta_v) return fail(ctx, PYLDA_ERR_INVALID, "%s: beta is NULL", who);
    if (!ctx->have_eta || !ctx->have_sstats)
        return fail(ctx, PYLDA_ERR_STATE, "%s: needs eta and the sufficient statistics of a training E-step", who);
    if (want_alpha_ss && (!c || c->ctx != ctx || !c->estep_done))
        return fail(ctx, PYLDA_ERR_STATE, "%s: alpha statistics need the corpus of the last E-step", who);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int K = ctx->K, V = ctx->V;
    // beta is constant over a run: its lgamma sums (V host lgamma calls) and the device copy are
    // refreshed only when the caller hands over different values
    if (ctx->h_beta.size() != (size_t)V || memcmp(ctx->h_beta.data(), beta_v, (size_t)V * sizeof(double)) != 0) {
        double bsum = 0.0, blg = 0.0;
        for (int v = 0; v < V; ++v) {
            if (!(beta_v[v] > 0.0)) return fail(ctx, PYLDA_ERR_INVALID, "%s: beta[%d]=%g", who, v, beta_v[v]);
            bsum += beta_v[v];
            blg += std::lgamma(beta_v[v]);
        }
        ctx->h_beta.clear();            // stays empty if the copy below fails
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_beta, beta_v, (size_t)V * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // host buffer is not retained
        ctx->h_beta.assign(beta_v, beta_v + V);
        ctx->beta_sum = bsum;
        ctx->beta_lgamma_sum = blg;
    }
    double* d_per_topic = ctx->d_small;          // K
    double* d_alpha_ss = ctx->d_small + K;       // K
    hipLaunchKernelGGL(mstep_topic_ll_kernel, dim3(K, kTopicChunks), dim3(256), 0, ctx->stream, ctx->d_eta, K, V,
                       ctx->d_partial);                                                                             // :224 (old eta)
    hipLaunchKernelGGL(mstep_topic_ll_finish_kernel, dim3((K + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_partial, K,
                       d_per_topic);
    hipLaunchKernelGGL(mstep_update_eta_kernel, dim3((K + 31) / 32, (V + 31) / 32), dim3(256), 0, ctx->stream,
                       ctx->d_sstats, ctx->d_beta, K, V, ctx->ldk, ctx->d_eta);                                               // :226
    if (want_alpha_ss) {
        const int nblocks = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (c->D + 3) / 4));     // (ctx->d_partial holds 1024 rows)
        hipLaunchKernelGGL(mstep_alpha_ss_kernel, dim3(nblocks), dim3(256), (size_t)4 * K * sizeof(double),
                           ctx->stream, c->d_gamma, c->D, K, ctx->d_partial);                                       // :232
        hipLaunchKernelGGL(column_sum_kernel, dim3((K + 63) / 64), dim3(256), 0, ctx->stream, ctx->d_partial,
                           nblocks, K, d_alpha_ss);                                                                 // :233
    }
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

static double topic_ll_from(const pylda_ctx* ctx, const double* per_topic)
{
    double ll = ctx->K * (std::lgamma(ctx->beta_sum) - ctx->beta_lgamma_sum);                                       // :222
    for (int k = 0; k < ctx->K; ++k) ll += per_topic[k];
    return ll;
}
}  // namespace

int pylda_mstep(pylda_ctx* ctx, pylda_corpus* c, const double* beta_v, double* topic_log_likelihood,
                double* alpha_ss_k)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    const int rc = enqueue_mstep(ctx, c, beta_v, alpha_ss_k != nullptr, "mstep");
    if (rc != PYLDA_OK) return rc;
    const int K = ctx->K;
    std::vector<double> per_topic((size_t)K);
    HIP_TRY(ctx, hipMemcpyAsync(per_topic.data(), ctx->d_small, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (alpha_ss_k)
        HIP_TRY(ctx, hipMemcpyAsync(alpha_ss_k, ctx->d_small + K, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (topic_log_likelihood) *topic_log_likelihood = topic_ll_from(ctx, per_topic.data());
    return PYLDA_OK;
}

int pylda_mstep_enqueue(pylda_ctx* ctx, pylda_corpus* c, const double* beta_v, int hyper_parameter_iteration,
                        double hyper_parameter_decay_factor, int hyper_parameter_maximum_decay,
                        double hyper_parameter_converge_threshold)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!c || c->ctx != ctx || !c->estep_done || c->last_heldout)
        return fail(ctx, PYLDA_ERR_STATE, "mstep_enqueue: needs the corpus of the last training-mode E-step");
    if (hyper_parameter_iteration < 0 || hyper_parameter_maximum_decay < 0 || hyper_parameter_maximum_decay > 16)
        return fail(ctx, PYLDA_ERR_INVALID, "mstep_enqueue: hyper_parameter_iteration=%d, hyper_parameter_maximum_decay=%d (0..16)",
                    hyper_parameter_iteration, hyper_parameter_maximum_decay);
    ctx->newton_pending = hyper_parameter_iteration > 0;
    if (ctx->newton_pending) {
        ctx->newton.iterations = hyper_parameter_iteration;
        ctx->newton.maximum_decay = hyper_parameter_maximum_decay;
        ctx->newton.threshold = hyper_parameter_converge_threshold;
        for (int d = 0; d <= 16; ++d) ctx->newton.decay_power[d] = std::pow(hyper_parameter_decay_factor, (double)d);   // numpy.power
    }
    const int rc = enqueue_mstep(ctx, c, beta_v, true, "mstep_enqueue");
    if (rc != PYLDA_OK) return rc;
    const int K = ctx->K;
    hipLaunchKernelGGL(outer_pack_kernel, dim3(1), dim3(256), 0, ctx->stream, c->d_scalars, c->d_flag_count,
                       c->last_doc_values ? 1 : 0, (double)c->D, ctx->d_small + K, ctx->d_small, ctx->d_alpha, K, ctx->d_outer);
    HIP_TRY(ctx, hipGetLastError());
    ctx->outer_ready = true;
    return PYLDA_OK;
}

void* pylda_outer_device(pylda_ctx* ctx, int64_t* n_reduce)
{
    if (!ctx) return nullptr;
    if (n_reduce) *n_reduce = ctx->K + 4;
    return ctx->d_outer;
}

int pylda_allreduce_outer(pylda_ctx* ctx)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!ctx->comm) return fail(ctx, PYLDA_ERR_STATE, "allreduce_outer: pylda_comm_init has not been called");
    if (!ctx->outer_ready) return fail(ctx, PYLDA_ERR_STATE, "allreduce_outer: pylda_mstep_enqueue has not run");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    const int rc = pylda::comm_allreduce_sum_f64(ctx->comm, ctx->d_outer, (size_t)ctx->K + 4, ctx->stream, &err);
    return rc == PYLDA_OK ? rc : fail(ctx, rc, "allreduce_outer: %s", err.c_str());
}

int pylda_outer_fetch(pylda_ctx* ctx, double* document_log_likelihood, double* number_of_documents,
                      int64_t* logspace_documents, double* topic_log_likelihood, double* alpha_ss_k, double* alpha_k)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!ctx->outer_ready) return fail(ctx, PYLDA_ERR_STATE, "outer_fetch: pylda_mstep_enqueue has not run");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int K = ctx->K;
    if (ctx->newton_pending) {
        // behind the all-reduce of the packed values (the statistics and #documents are the global ones on every rank)
        hipLaunchKernelGGL(alpha_newton_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_outer + 4 + 2 * (size_t)K,
                           ctx->d_outer + 4, ctx->d_outer + 1, K, ctx->newton, ctx->d_newton_work, ctx->d_alpha);
        HIP_TRY(ctx, hipGetLastError());
    }
    double* host = ctx->h_pin + (size_t)2 * K;
    HIP_TRY(ctx, hipMemcpyAsync(host, ctx->d_outer, (size_t)(3 * K + 4) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));          // the ONE wait of an outer iteration
    ctx->outer_ready = false;
    if (ctx->newton_pending) ctx->h_alpha.assign(host + 4 + 2 * (size_t)K, host + 4 + 3 * (size_t)K);   // what d_alpha holds now
    ctx->newton_pending = false;
    if (alpha_k) memcpy(alpha_k, host + 4 + 2 * (size_t)K, (size_t)K * sizeof(double));
    if (document_log_likelihood) *document_log_likelihood = host[0];
    if (number_of_documents) *number_of_documents = host[1];
    if (logspace_documents) *logspace_documents = (int64_t)std::llround(host[2]);
    if (alpha_ss_k) memcpy(alpha_ss_k, host + 4, (size_t)K * sizeof(double));
    if (topic_log_likelihood) *topic_log_likelihood = topic_ll_from(ctx, host + 4 + K);
    return PYLDA_OK;
}

int pylda_test_alpha_update(pylda_ctx* ctx, const double* alpha_k, const double* alpha_ss_k, double number_of_documents,
                            int hyper_parameter_iteration, double hyper_parameter_decay_factor, int hyper_parameter_maximum_decay,
                            double hyper_parameter_converge_threshold, double* alpha_out_k)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!alpha_k || !alpha_ss_k || !alpha_out_k || hyper_parameter_iteration < 1 || hyper_parameter_maximum_decay < 0 ||
        hyper_parameter_maximum_decay > 16)
        return fail(ctx, PYLDA_ERR_INVALID, "test_alpha_update: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int K = ctx->K;
    double* d = nullptr;                       // [alpha io (K) | statistics (K) | #documents | scratch alpha out (K)]
    int rc = dev_alloc(ctx, &d, (size_t)3 * K + 1);
    if (rc != PYLDA_OK) return rc;
    NewtonParams np;
    np.iterations = hyper_parameter_iteration;
    np.maximum_decay = hyper_parameter_maximum_decay;
    np.threshold = hyper_parameter_converge_threshold;
    for (int i = 0; i <= 16; ++i) np.decay_power[i] = std::pow(hyper_parameter_decay_factor, (double)i);
    hipError_t e = hipMemcpy(d, alpha_k, (size_t)K * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + K, alpha_ss_k, (size_t)K * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + 2 * (size_t)K, &number_of_documents, sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(alpha_newton_kernel, dim3(1), dim3(1024), 0, ctx->stream, d, d + K, d + 2 * (size_t)K, K, np,
                           ctx->d_newton_work, d + 2 * (size_t)K + 1);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(alpha_out_k, d, (size_t)K * sizeof(double), hipMemcpyDeviceToHost);
    dev_free(d);
    if (e != hipSuccess) return fail(ctx, PYLDA_ERR_HIP, "test_alpha_update: %s", hipGetErrorString(e));
    return PYLDA_OK;
}

}  // extern "C"
