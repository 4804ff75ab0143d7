// libpylda_hip.so - the launch plan: kernel variant and geometry per distinct-term count (host code only).
// (host side of the C ABI declared in include/pylda_hip.h; see host_internal.h for the map of the translation units)
#include "host_internal.h"

namespace pylda_host {
namespace {

// Slab (register-resident) kernel geometry for a document with n distinct terms:
// prefer 32-topic slabs (fewer wavefronts per document, so the per-wavefront
// digamma / reduction overhead is amortised over more FMAs) while the slab
// fits the 256 architectural VGPRs (RN <= 3), else 16-topic slabs (RN <= 6).
struct SlabGeom { int W, RK, RN; };
SlabGeom slab_geom_for(const pylda_ctx* ctx, int n)
{
    const int need = std::max(1, (n + 63) / 64), ldk = ctx->ldk;
    if ((ldk == 32 || ldk == 64 || ldk == 128) && need <= 2) return {ldk / 32, 32, need};
    if (ldk == 16 || ldk == 32 || ldk == 64 || ldk == 128) {
        if (need <= 4) return {ldk / 16, 16, need};
        if (need <= 6 && ldk <= 64) return {ldk / 16, 16, 6};
    }
    return {0, 0, 0};
}

// Quilt (2-D lanes, register-resident) kernel geometry: wavefronts per document and
// words per lane (W * 4 * RWL >= n), or W = 0.  12 wavefronts x 4 words per lane keeps a
// 129..192-term document at 149 VGPRs = 3 wavefronts per SIMD instead of 2.
struct QuiltGeom { int W, RWL; };
QuiltGeom quilt_geom_for(const pylda_ctx* ctx, int n)
{
    if (ctx->ldk != 64 && ctx->ldk != 128) return {0, 0};
    if (n <= 64) return {8, 2};
    if (n <= 128) return {8, 4};
    if (n <= 192 && ctx->quilt12) return {12, 4};
    if (n <= 192 && ctx->quilt_odd) return {8, 6};
    if (n <= 224 && ctx->quilt_odd) return {8, 7};
    if (n <= 256) return {8, 8};
    return {0, 0};
}
// Quad kernel (register + LDS tile on 16 word groups; estep_quad.h): K <= 128 (table stride 128): 4
// wavefronts per document, two documents per CU; 128 < K <= 256 (stride 256): 8 wavefronts, one per CU.
// Register, LDS and streamed slots per word group, N <= 16 * (RWL + TWL + SWL) <= 256; code SWL * 1000000 +
// TL * 10000 + RWL * 100 + TWL, or 0.  TWL <= 3: two workgroups inside a CU's 160 KiB of LDS at stride 128, one at stride 256.
int quad_geom_for(const pylda_ctx* ctx, int n)
{
    if ((ctx->ldk != 128 && ctx->ldk != 256) || !ctx->quad || ctx->lds_limit < 160 * 1024) return 0;
    const int tl = ctx->ldk / 8 * 10000;
    if (n <= 128) return tl + 800;
    if (n <= 160) return tl + 1000;
    if (n <= 176) return tl + 1001;
    if (n <= 192) return tl + 1002;
    if (n <= 208) return tl + 1003;
    if (n <= 224) return tl + 1004;
    // + SWL streamed slots (estep_quad.h), addressed by 32-bit byte offsets into the table
    if (!ctx->quad_stream || (uint64_t)ctx->V * (uint64_t)ctx->ldk * 8 >= (1ull << 32)) return 0;
    // (stride 128: nine register slots + 2 / 3 streamed - 319 ns per document on cfg 3's 225-256-term class against 326 for the
    //  quilt kernel and 332 with eight; stride 256: eight + 3 / 4 - 591 against 595 with nine, and no scratch)
    if (ctx->ldk == 128) return n <= 240 ? 2000000 + tl + 904 : n <= 256 ? 3000000 + tl + 904 : 0;
    return n <= 240 ? 3000000 + tl + 804 : n <= 256 ? 4000000 + tl + 804 : 0;
}

int quilt_rwl_for(const pylda_ctx* ctx, int n) { const QuiltGeom q = quilt_geom_for(ctx, n); return q.W * 100 + q.RWL; }

// Group-fused streaming kernel (estep_qgroup.h): table stride 64 / 128 / 256 (32-bit byte offsets into the table),
// documents up to 1024 distinct terms.
bool qgroup_ok(const pylda_ctx* ctx, int n)
{
    return (ctx->ldk == 64 || ctx->ldk == 128 || ctx->ldk == 256) && n <= kQgMaxWords && (uint64_t)ctx->V * (uint64_t)ctx->ldk * 8 < (1ull << 32);
}

// Decide the kernel variant for a document with n distinct terms.
int choose_variant(const pylda_ctx* ctx, int n, size_t* lds_bytes)
{
    // The register-resident and streaming kernels decide convergence on a 2^-40 fixed-point sum of
    // |delta gamma_k|, each clipped to 1024 (estep_common.h change_fixed): equivalent to the
    // reference's floating-point `mean <= threshold` (:187-189) while 2^-28 <= threshold*K < 1024.
    // Outside that range (threshold 0: "run until nothing moves at all"; huge thresholds) the
    // generic kernels, which compare in floating point, take the documents.
    if (ctx->exact_stop) goto generic;
    if ((ctx->force_variant < 0 || ctx->force_variant == kQuad) && quad_geom_for(ctx, n) > 0) {
        *lds_bytes = 0;
        return kQuad;
    }
    if ((ctx->force_variant < 0 || ctx->force_variant == kQfuse) && (ctx->ldk == 384 || ctx->ldk == 512) && n <= 8 * kQfMaxSlots - 32 &&
        ctx->lds_limit >= 160 * 1024) {
        *lds_bytes = 0;
        return kQfuse;
    }
    if ((ctx->force_variant < 0 || ctx->force_variant == kQfusek) && ctx->ldk > 512 && ctx->ldk <= 1024 && ctx->ldk % 128 == 0 &&
        n <= 8 * kQfMaxSlots && ctx->lds_limit >= 160 * 1024) {
        *lds_bytes = 0;
        return kQfusek;
    }
    if ((ctx->force_variant < 0 || ctx->force_variant == kQuilt) && quilt_geom_for(ctx, n).W > 0) {
        *lds_bytes = 0;
        return kQuilt;
    }
    if ((ctx->force_variant < 0 || ctx->force_variant == kSlab) && slab_geom_for(ctx, n).W > 0) {
        *lds_bytes = 0;
        return kSlab;
    }
    if ((ctx->force_variant < 0 || ctx->force_variant == kQgroup) && qgroup_ok(ctx, n)) {
        *lds_bytes = 0;
        return kQgroup;
    }
generic:
    // (a request within 3 KiB of the CU's 160 KiB is refused by hipFuncSetAttribute - found with 540-term documents at
    //  K = 32, 162 608 bytes; the quad kernel's 160 512 are accepted)
    constexpr size_t kLdsMargin = 3072;
    const int K = ctx->K, stride = tile_stride_for(K);
    const size_t l64 = generic_lds_layout(K, n, stride, 64, false).total;
    const size_t l256 = generic_lds_layout(K, n, stride, 256, false).total;
    const size_t l512 = generic_lds_layout(K, n, stride, 512, false).total;
    int v;
    if (ctx->force_variant >= 0 && ctx->force_variant < kSlab) v = ctx->force_variant;
    else if (ctx->force_variant == kGenericHuge) v = kGenericGlobal;
    else if (l64 <= 20 * 1024) v = kGeneric64;
    else if (l256 <= 64 * 1024) v = kGeneric256;
    else if (l512 + kLdsMargin <= ctx->lds_limit) v = kGeneric512;
    else v = kGenericGlobal;
    // a forced LDS variant that does not fit degrades to the global-tile kernel
    const size_t need = v == kGeneric64 ? l64 : v == kGeneric256 ? l256 : l512;
    if (v != kGenericGlobal && need + kLdsMargin > ctx->lds_limit) v = kGenericGlobal;
    // ... and a document whose per-term scalars (28 bytes per distinct term) do not fit either keeps those in
    // global memory as well: any length runs
    if (v == kGenericGlobal && (ctx->force_variant == kGenericHuge || generic_lds_layout(K, n, stride, 256, true).total > ctx->lds_limit))
        v = kGenericHuge;
    switch (v) {
    case kGeneric64: *lds_bytes = l64; break;
    case kGeneric256: *lds_bytes = l256; break;
    case kGeneric512: *lds_bytes = l512; break;
    case kGenericHuge: *lds_bytes = generic_lds_layout(K, 0, stride, 256, true).total; break;
    default: *lds_bytes = generic_lds_layout(K, n, stride, 256, true).total; break;
    }
    return v;
}

}  // namespace

void build_plan(pylda_corpus* c)
{
    pylda_ctx* ctx = c->ctx;
    c->plan.clear();
    c->plan_epoch = ctx->plan_epoch;
    c->plan_exact = ctx->exact_stop;
    const int64_t D = c->D;
    // Documents are sorted by distinct-term count, descending, and the kernel choice depends on that count only:
    // walk the RUNS of equal counts (a few hundred at most), not the documents (10^6 at cfg 4).
    struct Run { int64_t first, count; int n; int variant; size_t lds; int sub; int rk; };
    std::vector<Run> runs;
    for (int64_t i = 0; i < D;) {
        const int n = c->h_terms_sorted[(size_t)i];
        int64_t j = i + 1;
        while (j < D && c->h_terms_sorted[(size_t)j] == n) ++j;
        Run r{i, j - i, n, 0, 0, 0, 0};
        r.variant = choose_variant(ctx, n, &r.lds);
        r.sub = r.variant == kQuilt ? quilt_rwl_for(ctx, n) : r.variant == kQuad ? quad_geom_for(ctx, n)
              : r.variant == kSlab ? slab_geom_for(ctx, n).RN : 0;
        r.rk = r.variant == kSlab ? slab_geom_for(ctx, n).RK : 0;
        runs.push_back(r);
        i = j;
    }
    for (size_t a = 0; a < runs.size();) {
        // a launch is a maximal sequence of runs with the same variant and geometry whose LDS request (sized for
        // its first, largest document) is not more than ~25 % above what its last needs
        const Run& first = runs[a];
        size_t b = a + 1;
        int64_t docs = first.count;
        while (b < runs.size()) {
            const Run& r = runs[b];
            if (r.variant != first.variant || r.sub != first.sub || r.rk != first.rk) break;
            if (first.variant != kGenericGlobal && first.lds > 4096 && r.lds * 5 < first.lds * 4 && docs >= 4 * (int64_t)ctx->num_cu)
                break;
            docs += r.count;
            ++b;
        }
        Launch L;
        L.variant = first.variant;
        L.first = first.first;
        L.count = docs;
        L.n_cap = std::max(1, first.n);
        L.tile_stride = tile_stride_for(ctx->K);
        L.lds_bytes = first.lds;
        L.rn = first.sub;
        L.rk = first.rk;
        c->plan.push_back(L);
        a = b;
    }
}

// The slab classes of a small corpus as ONE dispatch (estep_slab.h, estep_slab_uber_kernel): the classes from index
// `from` to the end of the plan, or -1.  Eligible: at least two classes, all of the slab family with the same slab
// width, few enough wavefronts to be resident at once at two per SIMD (with documents of 6 words per lane in the
// launch the kernel needs more than 256 registers - one wavefront per SIMD, two rounds of residency at most - and still
// beats a second stream).
int slab_uber_from(const pylda_ctx* ctx, const pylda_corpus* c)
{
    if (!ctx->slab_uber || c->plan.size() < 2) return -1;
    int from = (int)c->plan.size();
    const int rk = c->plan.back().rk;
    int64_t docs = 0;
    while (from > 0) {
        const Launch& L = c->plan[(size_t)from - 1];
        if (L.variant != kSlab || L.rk != rk || (rk == 16 && L.rn > 6) || (rk == 32 && L.rn > 2)) break;
        docs += L.count;
        --from;
    }
    const int W = ctx->ldk / std::max(1, rk);
    // (8 wavefronts x 16-topic slabs: the combined kernel spills)
    if ((int)c->plan.size() - from < 2 || (int)c->plan.size() - from > 6 || W > 4 || docs * W > (int64_t)ctx->num_cu * 4 * 2) return -1;
    return from;
}

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
