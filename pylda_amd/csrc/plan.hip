// libpylda_hip.so - the launch plan of a corpus: the planner (host_plan.cpp) fed from the context, and its read-outs.
// (host side of the C ABI declared in include/pylda_hip.h; see host_internal.h for the map of the translation units)
#include "host_internal.h"

namespace pylda_host {

PlanConfig plan_config(const pylda_ctx* ctx)
{
    PlanConfig cfg;
    cfg.K = ctx->K;
    cfg.V = ctx->V;
    cfg.ldk = ctx->ldk;
    cfg.num_cu = ctx->num_cu;
    cfg.lds_limit = ctx->lds_limit;
    cfg.force_variant = ctx->force_variant;
    cfg.exact_stop = ctx->exact_stop;
    cfg.quad = ctx->quad;
    cfg.quad_stream = ctx->quad_stream;
    cfg.quilt12 = ctx->quilt12;
    cfg.quilt_odd = ctx->quilt_odd;
    cfg.slab_uber = ctx->slab_uber;
    return cfg;
}

void build_plan(pylda_corpus* c)
{
    pylda_ctx* ctx = c->ctx;
    c->plan_epoch = ctx->plan_epoch;
    c->plan_exact = ctx->exact_stop;
    c->plan = build_launch_classes(plan_config(ctx), c->h_terms_sorted.data(), c->D);
}

int slab_uber_from(const pylda_ctx* ctx, const pylda_corpus* c) { return pylda_plan::slab_uber_from(plan_config(ctx), c->plan); }

}  // namespace pylda_host

extern "C" {

int64_t pylda_corpus_layout(pylda_corpus* c, const char* name)
{
    if (!c || !name) return PYLDA_ERR_INVALID;
    if (!strcmp(name, "gather_blocks")) return c->have_postings ? c->gather_blocks : 0;
    if (!strcmp(name, "gather_segments")) return c->have_postings ? c->nseg : 0;
    if (!strcmp(name, "gather_rounds")) return c->have_postings ? (int64_t)c->rounds.size() : 0;
    if (!strcmp(name, "gather_sweep_passes")) return c->have_postings && c->sweep ? c->sweep_passes : 0;
    if (!strcmp(name, "gather_partial_rows")) return c->have_postings ? c->partial_rows : 0;
    if (!strcmp(name, "gather_live")) return c->have_postings && c->live_stats ? 1 : 0;
    if (!strcmp(name, "live_off_by_alpha")) return c->live_off_by_alpha ? 1 : 0;
    return fail(c->ctx, PYLDA_ERR_INVALID, "corpus_layout: unknown name '%s'", name);
}

int pylda_corpus_plan(pylda_corpus* c, int32_t capacity, int32_t* variant, int32_t* geometry, int64_t* documents,
                      int64_t* terms, double* kernel_ms)
{
    if (!c || !c->ctx || capacity < 0) return PYLDA_ERR_INVALID;
    pylda_ctx* ctx = c->ctx;
    if (c->plan_epoch != ctx->plan_epoch || c->plan_exact != ctx->exact_stop) build_plan(c);
    if (kernel_ms) {
        if (hipSetDevice(ctx->device) == hipSuccess) drain_events(ctx);
    }
    const int n = (int)std::min<size_t>(c->plan.size(), (size_t)capacity);
    for (int i = 0; i < n; ++i) {
        const Launch& L = c->plan[(size_t)i];
        if (variant) variant[i] = L.variant;
        if (geometry) geometry[i] = L.rn;
        if (documents) documents[i] = L.count;
        if (terms) {
            int64_t t = 0;
            for (int64_t j = L.first; j < L.first + L.count; ++j) t += c->h_terms_sorted[(size_t)j];
            terms[i] = t;
        }
        if (kernel_ms) {
            kernel_ms[i] = (size_t)i < ctx->class_ms.size() ? ctx->class_ms[(size_t)i] : 0.0;
            if ((size_t)i < ctx->class_ms.size()) ctx->class_ms[(size_t)i] = 0.0;
        }
    }
    return (int)c->plan.size();
}

}  // extern "C"
