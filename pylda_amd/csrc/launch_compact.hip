// libpylda_hip.so - the live-topic document kernel (estep_compact.h): hand-over buffers, instantiations and launcher.
// (host side of the C ABI declared in include/pylda_hip.h; see host_internal.h for the map of the translation units)
#include "host_internal.h"
#include "estep_compact.h"

namespace pylda_host {

namespace {

// Term slots per lane and the columns the register tile of that shape holds (2 S LT VGPRs of 256, two wavefronts
// per SIMD): the live-topic count at which a launch class hands its documents over.
int slots_for(int n_cap) { return std::max(1, (n_cap + kWave - 1) / kWave); }
int columns_for(int slots) { return slots <= 2 ? 32 : slots == 3 ? 28 : slots == 4 ? 20 : 0; }

template <int S, int LTMAX>
int launch_compact_as(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    auto kern = estep_compact_kernel<S, LTMAX>;
    const size_t lds = compact_lds_bytes(p.ldk);
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(kWave), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

}  // namespace

int compact_handoff_for(const pylda_ctx* ctx, const Launch& L)
{
    if (L.variant != kQuad) return 0;
    int cap = columns_for(slots_for(L.n_cap));
    if (ctx->compact_cap > 0) cap = std::min(cap, ctx->compact_cap);
    return cap;
}

int prepare_compact(pylda_ctx* ctx, pylda_corpus* c)
{
    c->compact_ready = false;
    if (!ctx->compact || c->compact_failed || ctx->exact_stop || ctx->force_logspace || ctx->ldk > 1024) return PYLDA_OK;
    bool any = false;
    for (const Launch& L : c->plan) any = any || compact_handoff_for(ctx, L) > 0;
    if (!any) return PYLDA_OK;
    if (c->d_live_tile && c->compact_plan_epoch == c->plan_epoch && c->compact_cap_used == ctx->compact_cap) {
        c->compact_ready = true;
        return PYLDA_OK;
    }
    // a document's tile: N_d x (columns of its class), at an offset of its own
    std::vector<int64_t> tile_ptr((size_t)c->D, 0);
    int64_t total = 0;
    for (const Launch& L : c->plan) {
        const int cap = compact_handoff_for(ctx, L);
        if (cap <= 0) continue;
        for (int64_t i = L.first; i < L.first + L.count; ++i) {
            tile_ptr[(size_t)c->h_order[(size_t)i]] = total;
            total += (int64_t)c->h_terms_sorted[(size_t)i] * cap;
        }
    }
    dev_free(c->d_live_tile);
    int rc = PYLDA_OK;
    auto A = [&](int r) { if (rc == PYLDA_OK) rc = r; };
    if (!c->d_live_n) {
        A(dev_alloc(ctx, &c->d_live_n, (size_t)c->D));
        A(dev_alloc(ctx, &c->d_live_list, (size_t)c->D * kLiveListBytes));
        A(dev_alloc(ctx, &c->d_tile_ptr, (size_t)c->D));
        A(dev_alloc(ctx, &c->d_handoff_it, (size_t)c->D));
        A(dev_alloc(ctx, &c->d_col_iters, (size_t)c->D));
    }
    if (rc == PYLDA_OK && hipMalloc(reinterpret_cast<void**>(&c->d_live_tile), (size_t)std::max<int64_t>(1, total) * sizeof(double)) != hipSuccess) {
        // no room for the tiles (cfg 4: 38 GB): the dense kernels run every iteration themselves, results are the same
        (void)hipGetLastError();
        c->d_live_tile = nullptr;
        c->compact_failed = true;
        ctx->err.clear();
        return PYLDA_OK;
    }
    if (rc != PYLDA_OK) return rc;
    HIP_TRY(ctx, hipMemcpy(c->d_tile_ptr, tile_ptr.data(), (size_t)c->D * sizeof(int64_t), hipMemcpyHostToDevice));
    c->compact_plan_epoch = c->plan_epoch;
    c->compact_cap_used = ctx->compact_cap;
    c->compact_ready = true;
    return PYLDA_OK;
}

int launch_compact(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    switch (slots_for(L.n_cap)) {
    case 1: return launch_compact_as<1, 32>(ctx, p, L);
    case 2: return launch_compact_as<2, 32>(ctx, p, L);
    case 3: return launch_compact_as<3, 28>(ctx, p, L);
    case 4: return launch_compact_as<4, 20>(ctx, p, L);
    }
    return fail(ctx, PYLDA_ERR_STATE, "no live-topic kernel for documents of %d terms", L.n_cap);
}

}  // namespace pylda_host
