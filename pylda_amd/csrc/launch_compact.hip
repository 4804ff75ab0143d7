// libpylda_hip.so - the live-topic document kernel (estep_compact.h): hand-over buffers, instantiations and launcher.
// (host side of the C ABI declared in include/pylda_hip.h; see host_internal.h for the map of the translation units)
#include "host_internal.h"
#include <cmath>
#include "estep_compact.h"

namespace pylda_host {

namespace {

// Term slots per lane of a document of n terms, and the columns ONE wavefront's register tile holds at that shape
// (2 S LT VGPRs of 256, two wavefronts per SIMD).  While more are alive the document's workgroup is two wavefronts
// with that many columns each (compact_pair_body): a dense kernel hands a document over at twice the figure.
int slots_for(int n) { return std::max(1, (n + kWave - 1) / kWave); }
int columns_for(int slots)
{
    static const int columns[9] = {0, 28, 28, 28, 20, 16, 12, 8, 8};      // (two wavefronts of 28: 56 of the list's 60 entries)
    return slots <= 8 ? columns[slots] : 0;
}

// TPW: columns per wavefront of the two-wavefront first stage, or 0: one wavefront per document from the start
template <int S, int LTMAX, int TPW>
int launch_compact_shape(pylda_ctx* ctx, const EstepParams& p, int64_t count)
{
    auto kern = estep_compact_kernel<S, LTMAX, TPW>;
    const size_t lds = compact_lds_bytes(p.ldk, S, TPW);
    if (lds > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)count), dim3(TPW > 0 ? 2 * kWave : kWave), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

template <int S, int LTMAX>
int launch_compact_as(pylda_ctx* ctx, const EstepParams& p, int64_t count)
{
    // (documents are handed over at p.handoff_caps: beyond one wavefront's columns only the pair kernel can take them)
    return p.handoff_caps[S] > LTMAX ? launch_compact_shape<S, LTMAX, LTMAX>(ctx, p, count) : launch_compact_shape<S, LTMAX, 0>(ctx, p, count);
}

}  // namespace

// does the launch class hand documents to the live-topic kernel, and how: 1 with their tile columns (the quad kernel holds the
// tile on chip), 2 without (the fused streaming kernels: the live-topic kernel gathers its tile from the table), 0 not at all
int compact_handoff_for(const pylda_ctx* ctx, const Launch& L)
{
    if (L.variant == kQuad) return 1;
    if (L.variant == kQfuse && ctx->compact_stream) return 2;      // (estep_qfusek.h, 512 < K <= 1024, keeps its documents: two topics per thread)
    return 0;
}

void compact_caps(const pylda_ctx* ctx, int (&caps)[9])
{
    caps[0] = 0;
    // Two wavefronts (twice the columns) from table stride 256 on.  At stride 128 the dense kernel runs two documents per
    // CU and an iteration of it costs less than one of the pair body with its barrier (measured, cfg 3: document kernels
    // 12.4 ms handing over at one wavefront's columns, 13.0 ms at twice as many); at stride 256 the two break even in
    // the bench's window (cfg 4, 200k documents: 39.7 ms either way) and the pair is what keeps documents with 20-40 live
    // topics - a trained model - out of the dense kernel at all.  Option compact_pair: 0 never, 1 always.
    const bool pair = ctx->compact_pair < 0 ? ctx->ldk >= 256 : ctx->compact_pair != 0;
    for (int slots = 1; slots <= 8; ++slots) {
        caps[slots] = (pair ? 2 : 1) * columns_for(slots);
        if (ctx->compact_cap > 0) caps[slots] = std::min(caps[slots], ctx->compact_cap);
    }
}

// psi(x), x > 0, on the host: recurrence up to 10, then the asymptotic series (the rule below has orders of magnitude to spare)
static double host_digamma(double x)
{
    double shift = 0.0;
    while (x < 10.0) {
        shift -= 1.0 / x;
        x += 1.0;
    }
    const double f = 1.0 / (x * x);
    return shift + std::log(x) - 0.5 / x - f * (1.0 / 12.0 - f * (1.0 / 120.0 - f * (1.0 / 252.0 - f * (1.0 / 240.0 - f * (1.0 / 132.0)))));
}

// alpha_mortality_kernel's rule (prepare_kernels.h) on the host's copy of alpha: the topics whose t at gamma = alpha is not
// negligible.  They are live in EVERY document, so they are columns of every tile.  (A topic on the boundary may be classed
// differently here and on the device: this count only decides whether the hand-over is worth setting up.)
int immortal_topics(const pylda_ctx* ctx)
{
    double sum = 0.0;
    for (double a : ctx->h_alpha) sum += a;
    const double psi_shortest = host_digamma(sum + kMortalTokens), bound = std::log(kMortalT);
    int n = 0;
    for (double a : ctx->h_alpha) n += !(host_digamma(a) - psi_shortest < bound);
    return n;
}

// Once alpha has grown (the Newton update of a training run: cfg 3 past iteration 10, cfg 4 past 18) as many topics never
// die as the widest tile has columns: no document can leave the dense kernels, and a corpus set up for the hand-over
// pays for it - the plain walk of the postings adds whole rows for documents without a list (14 ms against the sweep's 8
// per 200k cfg-4 documents).  From then on the corpus runs as with compact = 0 (and comes back only at half that count).
bool alpha_allows_live(const pylda_ctx* ctx, bool was_off)
{
    if (!ctx->compact || ctx->h_alpha.empty()) return true;
    int caps[9], widest = 0;
    compact_caps(ctx, caps);
    for (int s = 1; s <= 8; ++s) widest = std::max(widest, caps[s]);
    const int immortal = immortal_topics(ctx);
    return was_off ? 2 * immortal <= widest : immortal < widest;
}

int prepare_compact(pylda_ctx* ctx, pylda_corpus* c)
{
    c->compact_ready = false;
    if (!ctx->compact || c->compact_failed || ctx->exact_stop || ctx->force_logspace || ctx->ldk > 1024) return PYLDA_OK;
    if (c->live_off_by_alpha) {
        dev_free(c->d_live_tile);           // (the largest buffer of the corpus: cfg 4 69 GB)
        return PYLDA_OK;
    }
    bool any = false;
    for (const Launch& L : c->plan) any = any || compact_handoff_for(ctx, L) > 0;
    if (!any) return PYLDA_OK;
    if (c->d_live_tile && c->compact_plan_epoch == c->plan_epoch && c->compact_cap_used == ctx->compact_cap && c->compact_stream_used == ctx->compact_stream &&
        c->compact_pair_used == ctx->compact_pair) {
        c->compact_ready = true;
        return PYLDA_OK;
    }
    // a document's tile (quad classes): N_d x (live topics it is handed over at), at an offset of its own; and the
    // schedule ranges per lane shape (the schedule is sorted by length: the documents of a shape are contiguous in a class)
    std::vector<int64_t> tile_ptr((size_t)c->D, -1);
    int caps[9];
    compact_caps(ctx, caps);
    int64_t total = 0;
    c->compact_ranges.clear();
    for (size_t li = 0; li < c->plan.size(); ++li) {
        const Launch& L = c->plan[li];
        const int mode = compact_handoff_for(ctx, L);
        if (mode == 0) continue;
        for (int64_t i = L.first; i < L.first + L.count;) {
            const int slots = slots_for(c->h_terms_sorted[(size_t)i]);
            int64_t j = i;
            while (j < L.first + L.count && slots_for(c->h_terms_sorted[(size_t)j]) == slots) {
                if (mode == 1 && slots <= 8) {
                    tile_ptr[(size_t)c->h_order[(size_t)j]] = total;
                    total += (int64_t)c->h_terms_sorted[(size_t)j] * caps[slots];
                }
                ++j;
            }
            if (slots <= 8 && caps[slots] > 0) c->compact_ranges.push_back(pylda_corpus::CompactRange{(int)li, slots, mode == 2, i, j - i});
            i = j;
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
    c->compact_stream_used = ctx->compact_stream;
    c->compact_pair_used = ctx->compact_pair;
    c->compact_ready = true;
    return PYLDA_OK;
}

// the live-topic kernel over `count` schedule slots from `first` on, all of `slots` term slots per lane
int launch_compact(pylda_ctx* ctx, const pylda_corpus* c, EstepParams p, int slots, bool from_table, int64_t first, int64_t count)
{
    p.order = c->d_order + first;
    p.tile_from_table = from_table ? 1 : 0;
    switch (slots) {
    case 1: return launch_compact_as<1, 28>(ctx, p, count);
    case 2: return launch_compact_as<2, 28>(ctx, p, count);
    case 3: return launch_compact_as<3, 28>(ctx, p, count);
    case 4: return launch_compact_as<4, 20>(ctx, p, count);
    case 5: return launch_compact_as<5, 16>(ctx, p, count);
    case 6: return launch_compact_as<6, 12>(ctx, p, count);
    case 7: return launch_compact_as<7, 8>(ctx, p, count);
    case 8: return launch_compact_as<8, 8>(ctx, p, count);
    }
    return fail(ctx, PYLDA_ERR_STATE, "no live-topic kernel for %d term slots per lane", slots);
}

}  // namespace pylda_host
