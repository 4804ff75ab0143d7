// libpylda_hip.so - the host-side planner: launch classes of the document kernels and the layout of the statistics
// pass, as pure functions over host arrays (host_plan.h).  No HIP in this file: it is also built with
// g++ -fsanitize=address,undefined and fuzzed on the CPU (tests/test_planner_sanitizers.py).
#include "host_plan.h"

#include <algorithm>
#include <cmath>
#include <numeric>

using namespace pylda;

namespace pylda_plan {

// ---- launch classes ----------------------------------------------------------------------------------------------
namespace {

// Slab (register-resident) kernel geometry for a document with n distinct terms: prefer 32-topic slabs (fewer
// wavefronts per document, so the per-wavefront digamma / reduction overhead is amortised over more FMAs) while the
// slab fits the 256 architectural VGPRs (RN <= 3), else 16-topic slabs (RN <= 6).
struct SlabGeom { int W, RK, RN; };
SlabGeom slab_geom_for(const PlanConfig& cfg, int n)
{
    const int need = std::max(1, (n + 63) / 64), ldk = cfg.ldk;
    if ((ldk == 32 || ldk == 64 || ldk == 128) && need <= 2) return {ldk / 32, 32, need};
    if (ldk == 16 || ldk == 32 || ldk == 64 || ldk == 128) {
        if (need <= 4) return {ldk / 16, 16, need};
        if (need <= 6 && ldk <= 64) return {ldk / 16, 16, 6};
    }
    return {0, 0, 0};
}

// Quilt (2-D lanes, register-resident) kernel geometry: wavefronts per document and words per lane
// (W * 4 * RWL >= n), or W = 0.
struct QuiltGeom { int W, RWL; };
QuiltGeom quilt_geom_for(const PlanConfig& cfg, int n)
{
    if (cfg.ldk != 64 && cfg.ldk != 128) return {0, 0};
    if (n <= 64) return {8, 2};
    if (n <= 128) return {8, 4};
    if (n <= 192 && cfg.quilt12) return {12, 4};
    if (n <= 192 && cfg.quilt_odd) return {8, 6};
    if (n <= 224 && cfg.quilt_odd) return {8, 7};
    if (n <= 256) return {8, 8};
    return {0, 0};
}

bool table_below_4gib(const PlanConfig& cfg) { return (uint64_t)cfg.V * (uint64_t)cfg.ldk * 8 < (1ull << 32); }

// Quad kernel (register + LDS tile on 16 word groups; estep_quad.h): K <= 128 (table stride 128): 4 wavefronts per
// document, two documents per CU; 128 < K <= 256 (stride 256): 8 wavefronts, one per CU.  Register, LDS and streamed
// slots per word group, N <= 16 * (RWL + TWL + SWL) <= 256; code SWL * 1000000 + TL * 10000 + RWL * 100 + TWL, or 0.
int quad_geom_for(const PlanConfig& cfg, int n)
{
    if ((cfg.ldk != 128 && cfg.ldk != 256) || !cfg.quad || cfg.lds_limit < 160 * 1024) return 0;
    const int tl = cfg.ldk / 8 * 10000;
    if (n <= 128) return tl + 800;
    if (n <= 160) return tl + 1000;
    if (n <= 176) return tl + 1001;
    if (n <= 192) return tl + 1002;
    if (n <= 208) return tl + 1003;
    if (n <= 224) return tl + 1004;
    // + SWL streamed slots (estep_quad.h), addressed by 32-bit byte offsets into the table
    if (!cfg.quad_stream || !table_below_4gib(cfg)) return 0;
    // (stride 128: nine register slots + 2 / 3 streamed - 319 ns per document on cfg 3's 225-256-term class against 326 for the
    //  quilt kernel and 332 with eight; stride 256: eight + 3 / 4 - 591 against 595 with nine, and no scratch)
    if (cfg.ldk == 128) return n <= 240 ? 2000000 + tl + 904 : n <= 256 ? 3000000 + tl + 904 : 0;
    return n <= 240 ? 3000000 + tl + 804 : n <= 256 ? 4000000 + tl + 804 : 0;
}

// Group-fused streaming kernel (estep_qgroup.h): table stride 64 / 128 / 256 (32-bit byte offsets into the table),
// documents up to 1024 distinct terms.
bool qgroup_ok(const PlanConfig& cfg, int n)
{
    return (cfg.ldk == 64 || cfg.ldk == 128 || cfg.ldk == 256) && n <= kQgMaxWords && table_below_4gib(cfg);
}
bool qfuse_ok(const PlanConfig& cfg, int n)
{
    return (cfg.ldk == 384 || cfg.ldk == 512) && n <= 8 * kQfMaxSlots - 32 && cfg.lds_limit >= 160 * 1024;
}
bool qfusek_ok(const PlanConfig& cfg, int n)
{
    return cfg.ldk > 512 && cfg.ldk <= 1024 && cfg.ldk % 128 == 0 && n <= 8 * kQfMaxSlots && cfg.lds_limit >= 160 * 1024;
}

// (a request within 3 KiB of the CU's 160 KiB is refused by hipFuncSetAttribute - found with 540-term documents at
//  K = 32, 162 608 bytes; the quad kernel's 160 512 are accepted)
constexpr size_t kLdsMargin = 3072;

int choose_generic(const PlanConfig& cfg, int n, size_t* lds_bytes)
{
    const int K = cfg.K, stride = tile_stride_for(K);
    const size_t l64 = generic_lds_layout(K, n, stride, 64, false).total;
    const size_t l256 = generic_lds_layout(K, n, stride, 256, false).total;
    const size_t l512 = generic_lds_layout(K, n, stride, 512, false).total;
    int v;
    if (cfg.force_variant >= 0 && cfg.force_variant < kSlab) v = cfg.force_variant;
    else if (cfg.force_variant == kGenericHuge) v = kGenericGlobal;
    else if (l64 <= 20 * 1024) v = kGeneric64;
    else if (l256 <= 64 * 1024) v = kGeneric256;
    else if (l512 + kLdsMargin <= cfg.lds_limit) v = kGeneric512;
    else v = kGenericGlobal;
    // a forced LDS variant that does not fit degrades to the global-tile kernel
    const size_t need = v == kGeneric64 ? l64 : v == kGeneric256 ? l256 : l512;
    if (v != kGenericGlobal && need + kLdsMargin > cfg.lds_limit) v = kGenericGlobal;
    // ... and a document whose per-term scalars (28 bytes per distinct term) do not fit either keeps those in
    // global memory as well: any length runs
    if (v == kGenericGlobal &&
        (cfg.force_variant == kGenericHuge || generic_lds_layout(K, n, stride, 256, true).total + kLdsMargin > cfg.lds_limit))
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

// Decide the kernel variant for a document with n distinct terms.
int choose_variant(const PlanConfig& cfg, int n, size_t* lds_bytes)
{
    // The register-resident and streaming kernels decide convergence on a 2^-40 fixed-point sum of |delta gamma_k|,
    // each clipped to 1024 (estep_common.h change_fixed): equivalent to the reference's floating-point
    // `mean <= threshold` (:187-189) while 2^-28 <= threshold*K < 1024.  Outside that range (threshold 0: "run until
    // nothing moves at all"; huge thresholds) the generic kernels, which compare in floating point, take the documents.
    if (cfg.exact_stop) return choose_generic(cfg, n, lds_bytes);
    auto wanted = [&](int v) { return cfg.force_variant < 0 || cfg.force_variant == v; };
    *lds_bytes = 0;
    if (wanted(kQuad) && quad_geom_for(cfg, n) > 0) return kQuad;
    if (wanted(kQfuse) && qfuse_ok(cfg, n)) return kQfuse;
    if (wanted(kQfusek) && qfusek_ok(cfg, n)) return kQfusek;
    if (wanted(kQuilt) && quilt_geom_for(cfg, n).W > 0) return kQuilt;
    if (wanted(kSlab) && slab_geom_for(cfg, n).W > 0) return kSlab;
    if (wanted(kQgroup) && qgroup_ok(cfg, n)) return kQgroup;
    return choose_generic(cfg, n, lds_bytes);
}

int geometry_for(const PlanConfig& cfg, int variant, int n, int* rk)
{
    *rk = 0;
    switch (variant) {
    case kQuilt: { const QuiltGeom q = quilt_geom_for(cfg, n); return q.W * 100 + q.RWL; }
    case kQuad: return quad_geom_for(cfg, n);
    case kSlab: { const SlabGeom s = slab_geom_for(cfg, n); *rk = s.RK; return s.RN; }
    default: return 0;
    }
}

int64_t capacity_of(const PlanConfig& cfg, int variant, int rn, int /*rk*/, size_t lds_bytes)
{
    switch (variant) {
    case kQuad: { const int swl = rn / 1000000, rwl = rn % 10000 / 100, twl = rn % 100; return 16 * (rwl + twl + swl); }
    case kQuilt: return (int64_t)(rn / 100) * 4 * (rn % 100);
    case kSlab: return 64 * (int64_t)rn;
    case kQgroup: return kQgMaxWords;
    case kQfuse: return 8 * kQfMaxSlots - 32;
    case kQfusek: return 8 * kQfMaxSlots;
    case kGenericHuge: return INT32_MAX;
    case kGenericGlobal: {      // the per-term scalars of the largest document fit the request
        int64_t n = 0;
        while (generic_lds_layout(cfg.K, (int)n + 1, tile_stride_for(cfg.K), 256, true).total <= lds_bytes) ++n;
        return n;
    }
    default: {                  // tile in LDS
        const int nt = variant == kGeneric64 ? 64 : variant == kGeneric256 ? 256 : 512;
        int64_t n = 0;
        while (generic_lds_layout(cfg.K, (int)n + 1, tile_stride_for(cfg.K), nt, false).total <= lds_bytes) ++n;
        return n;
    }
    }
}

bool geometry_is_instantiated(const PlanConfig& cfg, int variant, int rn, int rk)
{
    switch (variant) {
    case kQuad: {
        static const int codes[] = {160800, 161000, 161001, 161002, 161003, 161004, 320800, 321000, 321001, 321002, 321003, 321004,
                                    2160904, 3160904, 3320804, 4320804};
        return std::find(std::begin(codes), std::end(codes), rn) != std::end(codes) && rn % 1000000 / 10000 == cfg.ldk / 8;
    }
    case kQuilt: {
        static const int codes[] = {802, 804, 806, 807, 808, 1204};
        return (cfg.ldk == 64 || cfg.ldk == 128) && std::find(std::begin(codes), std::end(codes), rn) != std::end(codes);
    }
    case kSlab: {
        if (rk != 16 && rk != 32) return false;
        const int W = cfg.ldk / rk;
        if (rk == 32) return (W == 1 || W == 2 || W == 4) && (rn == 1 || rn == 2);
        if (rn >= 1 && rn <= 4) return W == 1 || W == 2 || W == 4 || W == 8;
        return rn == 6 && (W == 1 || W == 2 || W == 4);
    }
    case kQgroup: return cfg.ldk == 64 || cfg.ldk == 128 || cfg.ldk == 256;
    case kQfuse: return cfg.ldk == 384 || cfg.ldk == 512;
    case kQfusek: return cfg.ldk >= 640 && cfg.ldk <= 1024 && cfg.ldk % 128 == 0;
    case kRetired5: case kRetired7: case kRetired8: return false;
    default: return variant >= 0 && variant <= kVariantLast;
    }
}

std::vector<Launch> build_launch_classes(const PlanConfig& cfg, const int32_t* terms_sorted, int64_t D)
{
    std::vector<Launch> plan;
    // Documents are sorted by distinct-term count, descending, and the kernel choice depends on that count only:
    // walk the RUNS of equal counts (a few hundred at most), not the documents (10^6 at cfg 4).
    struct Run { int64_t first, count; int n; int variant; size_t lds; int sub; int rk; };
    std::vector<Run> runs;
    for (int64_t i = 0; i < D;) {
        const int n = terms_sorted[(size_t)i];
        int64_t j = i + 1;
        while (j < D && terms_sorted[(size_t)j] == n) ++j;
        Run r{i, j - i, n, 0, 0, 0, 0};
        r.variant = choose_variant(cfg, n, &r.lds);
        r.sub = geometry_for(cfg, r.variant, n, &r.rk);
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
            if (first.variant != kGenericGlobal && first.lds > 4096 && r.lds * 5 < first.lds * 4 && docs >= 4 * (int64_t)cfg.num_cu)
                break;
            docs += r.count;
            ++b;
        }
        Launch L;
        L.variant = first.variant;
        L.first = first.first;
        L.count = docs;
        L.n_cap = std::max(1, first.n);
        L.tile_stride = tile_stride_for(cfg.K);
        L.lds_bytes = first.lds;
        L.rn = first.sub;
        L.rk = first.rk;
        plan.push_back(L);
        a = b;
    }
    return plan;
}

// The slab classes of a small corpus as ONE dispatch: the classes from index `from` to the end of the plan, or -1.
// Eligible: at least two classes, all of the slab family with the same slab width, few enough wavefronts to be resident
// at once at two per SIMD (with documents of 6 words per lane in the launch the kernel needs more than 256 registers -
// one wavefront per SIMD, two rounds of residency at most - and still beats a second stream).
int slab_uber_from(const PlanConfig& cfg, const std::vector<Launch>& plan)
{
    if (!cfg.slab_uber || plan.size() < 2) return -1;
    int from = (int)plan.size();
    const int rk = plan.back().rk;
    int64_t docs = 0;
    while (from > 0) {
        const Launch& L = plan[(size_t)from - 1];
        if (L.variant != kSlab || L.rk != rk || (rk == 16 && L.rn > 6) || (rk == 32 && L.rn > 2)) break;
        docs += L.count;
        --from;
    }
    const int W = cfg.ldk / std::max(1, rk);
    // (8 wavefronts x 16-topic slabs: the combined kernel spills)
    if ((int)plan.size() - from < 2 || (int)plan.size() - from > 6 || W > 4 || docs * W > (int64_t)cfg.num_cu * 4 * 2) return -1;
    return from;
}

// ---- statistics pass ------------------------------------------------------------------------------------------------

int document_blocks(const GatherConfig& g)
{
    // Document-blocked gather (sstats_kernels.h): NB contiguous document blocks whose t rows fit an XCD's L2, NB a
    // multiple of the 8 XCDs; only for the whole-row kernel, when all of t exceeds one L2 and a (term, block) pair
    // still holds >= 8 postings on average.
    const double t_bytes = (double)g.D * g.ldk * sizeof(double);
    const bool rows_kernel = g.gather_rows >= 1 && (g.ldk == 64 || g.ldk == 128 || g.ldk == 256);
    const bool bulk_kernel = g.gather_rows == 2 && (g.ldk == 128 || g.ldk == 256);   // (short segments need it)
    if (g.gather_blocks > 1 && rows_kernel && g.V > 0) return g.gather_blocks;          // forced (tests, A/B runs)
    if (g.gather_blocks < 0 && bulk_kernel && t_bytes > 8.6e6 && g.V > 0) {
        // automatic: blocks of about one L2 (cfg 3 sweep: 16 -> 1.78 ms, 24 -> 1.62, 32 -> ~1.8, 64 -> 3.1; unblocked 3.03),
        // but no more than leave a (term, block) pair 8 postings on average - every pair costs a partial row
        // (cfg 4, t = 2 GB: 64 blocks 55 ms, 128 53, 256 48, unblocked 65; 240 by this rule)
        const int by_l2 = std::max(8, 8 * (int)std::lround(t_bytes / (8 * 4.3e6)));
        const int by_pairs = (int)std::min<double>(1e6, (double)g.nnz / (8.0 * g.V)) / 8 * 8;
        const int NB = std::min(by_l2, by_pairs);
        return NB < 8 ? 1 : NB;
    }
    return 1;
}

double decision_budget(const GatherConfig& g)
{
    return g.gather_round_mb > 0 ? (double)g.gather_round_mb * 1048576.0 : 4.0 * 1073741824.0;
}

double round_budget(const GatherConfig& g, size_t free_device_bytes)
{
    double budget = decision_budget(g);
    // never more than a quarter of the device memory that is free right now: a shared or nearly full device gets
    // more, smaller rounds instead of an allocation failure (the rounds do not change a bit of the result: the partial
    // rows of a term are summed in segment order whatever round they belong to, and the entropy partials are laid
    // out by blocks of 256 statistics of the WHOLE table - round boundaries are multiples of 16 terms)
    if (g.gather_round_mb <= 0 && free_device_bytes > 0) budget = std::min(budget, (double)free_device_bytes / 4.0);
    return budget;
}

// Geometry of the persistent sweep (sstats_sweep.h) for V terms: terms per wavefront and wavefronts per workgroup (one
// workgroup per CU) such that the fewest passes over the document blocks cover all terms.
SweepGeom sweep_geometry(const GatherConfig& g)
{
    const int64_t cus = g.num_cu;
    auto passes = [&](int T, int WPB) { return (int)((g.V + cus * WPB * T - 1) / (cus * WPB * T)); };
    if (g.ldk == 128) {
        if (passes(12, 16) == 1) return {12, 16, 1};
        return {16, 16, passes(16, 16)};
    }
    const int a = passes(8, 16), b = passes(12, 12);       // stride 256: 8 VGPRs per term
    return b < a ? SweepGeom{12, 12, b} : SweepGeom{8, 16, a};
}

bool sweep_wanted(const GatherConfig& g, int NB, bool resident)
{
    if (!(NB > 1 && g.nnz > 0 && g.gather_sweep && (g.ldk == 128 || g.ldk == 256) && g.gather_rows == 2)) return false;
    // (mode 1: only where the (term, block) partial rows - about V x NB of them - would not fit their budget)
    const double rows_bytes = ((double)std::min<int64_t>((int64_t)g.V * NB, g.nnz) + (double)g.nnz / kGatherSegment) * g.ldk * sizeof(double);
    return resident && (g.gather_sweep == 2 || rows_bytes > decision_budget(g));
}

int sweep_blocks(const GatherConfig& g, int NB)
{
    if (g.gather_blocks > 1 || NB <= 8) return NB;          // forced, or nothing to merge
    return std::max(8, NB * 2 / 3 / 8 * 8);
}

namespace {

// the terms whose postings start in thread t's 1 / nthreads share of the posting range, on multiples of 16 terms
int piece_first_term(const int64_t* col_ptr, int V, int64_t nnz, int t, int nthreads)
{
    if (t <= 0) return 0;
    if (t >= nthreads) return V;
    const int64_t from = nnz * t / nthreads;
    const int v = (int)(std::lower_bound(col_ptr, col_ptr + V, from) - col_ptr);
    return std::min(V, (v + 15) & ~15);
}

}  // namespace

const char* cut_segments_blocked(const int64_t* col_ptr, const int32_t* post_doc, int V, int64_t D, int64_t nnz, int NB,
                                 int64_t cap, int nthreads, SegmentCut* out)
{
    const int64_t per_block = (D + NB - 1) / NB;
    if (per_block <= 0 || cap <= 0 || nthreads <= 0) return "segment cut: bad geometry";
    out->pieces.assign((size_t)nthreads, CutPiece());
    run_on_threads(nthreads, [&](int t) {
        CutPiece& piece = out->pieces[(size_t)t];
        piece.per_block.assign((size_t)NB, 0);
        const int v0 = piece_first_term(col_ptr, V, nnz, t, nthreads), v1 = piece_first_term(col_ptr, V, nnz, t + 1, nthreads);
        piece.v0 = v0;
        piece.per_word.reserve((size_t)std::max(0, v1 - v0));
        for (int v = v0; v < v1; ++v) {
            int64_t b = col_ptr[(size_t)v], n = 0;
            while (b < col_ptr[(size_t)v + 1]) {
                const int32_t blk = (int32_t)(post_doc[(size_t)b] / per_block);
                const int64_t block_end = ((int64_t)blk + 1) * per_block;       // first document of the next block
                const int64_t stop = std::min<int64_t>(col_ptr[(size_t)v + 1], b + cap);
                int64_t e = b + 1;
                while (e < stop && post_doc[(size_t)e] < block_end) ++e;
                piece.begin.push_back(b);
                piece.end.push_back(e);
                piece.block.push_back(blk);
                piece.per_block[(size_t)blk] += 1;
                b = e;
                ++n;
            }
            piece.per_word.push_back(n);
        }
    });
    int64_t total = 0;
    int covered = 0;
    for (CutPiece& piece : out->pieces) {          // the pieces cover the terms in order
        if (piece.v0 != covered) return "segment cut: the pieces lost terms";
        piece.base = total;
        total += (int64_t)piece.begin.size();
        covered += (int)piece.per_word.size();
    }
    if (covered != V) return "segment cut: the pieces do not cover the vocabulary";
    out->seg_begin.resize((size_t)total);
    out->seg_end.resize((size_t)total);
    out->word_seg_ptr.assign((size_t)V + 1, 0);
    run_on_threads(nthreads, [&](int t) {
        const CutPiece& piece = out->pieces[(size_t)t];
        std::copy(piece.begin.begin(), piece.begin.end(), out->seg_begin.begin() + piece.base);
        std::copy(piece.end.begin(), piece.end.end(), out->seg_end.begin() + piece.base);
        int64_t at = piece.base;
        for (size_t i = 0; i < piece.per_word.size(); ++i) {
            at += piece.per_word[i];
            out->word_seg_ptr[(size_t)piece.v0 + i + 1] = at;
        }
    });
    return nullptr;
}

void cut_segments_plain(const int64_t* col_ptr, int V, SegmentCut* out)
{
    out->pieces.clear();
    out->seg_begin.clear();
    out->seg_end.clear();
    out->word_seg_ptr.assign((size_t)V + 1, 0);
    for (int v = 0; v < V; ++v) {
        for (int64_t b = col_ptr[v]; b < col_ptr[v + 1]; b += kGatherSegment) {
            out->seg_begin.push_back(b);
            out->seg_end.push_back(std::min<int64_t>(b + kGatherSegment, col_ptr[v + 1]));
        }
        out->word_seg_ptr[(size_t)v + 1] = (int64_t)out->seg_begin.size();
    }
}

std::vector<int32_t> deal_terms(const int64_t* col_ptr, int V, int64_t nwaves, int T, int passes)
{
    // terms by posting count, largest first, dealt boustrophedon over the wavefronts: equal work per block
    std::vector<int32_t> by_df((size_t)V);
    std::iota(by_df.begin(), by_df.end(), 0);
    std::stable_sort(by_df.begin(), by_df.end(), [&](int32_t a, int32_t b) {
        return col_ptr[(size_t)a + 1] - col_ptr[(size_t)a] > col_ptr[(size_t)b + 1] - col_ptr[(size_t)b];
    });
    std::vector<int32_t> term_of((size_t)passes * nwaves * T, -1);
    for (int64_t j = 0; j < V; ++j) {
        const int64_t row = j / nwaves, col = (row & 1) ? nwaves - 1 - j % nwaves : j % nwaves;
        term_of[(size_t)(((row / T) * nwaves + col) * T + row % T)] = by_df[(size_t)j];
    }
    return term_of;
}

RoundPlan single_round(int64_t nseg, int V, int ldk)
{
    RoundPlan plan;
    plan.rounds.push_back(Round{0, nseg, 0, V, 0, nseg, 0, finalize_blocks(V, ldk)});
    plan.partial_rows = nseg;
    plan.ent_blocks = finalize_blocks(V, ldk);
    return plan;
}

RoundPlan plan_rounds(const SegmentCut& cut, int NB, int64_t max_rows, int ldk)
{
    RoundPlan plan;
    const std::vector<CutPiece>& pieces = cut.pieces;
    max_rows = std::max<int64_t>(1, max_rows);
    // rounds: groups of consecutive pieces (contiguous term ranges), each within the budget of partial rows
    std::vector<std::pair<size_t, size_t>> groups;        // [first piece, last piece + 1)
    for (size_t t = 0; t < pieces.size();) {
        size_t u = t + 1;
        int64_t rows = (int64_t)pieces[t].begin.size();
        // (a piece without segments joins the round in front of it: no round starts at the end of the vocabulary)
        while (u < pieces.size() && (pieces[u].begin.empty() || rows + (int64_t)pieces[u].begin.size() <= max_rows))
            rows += (int64_t)pieces[u++].begin.size();
        groups.emplace_back(t, u);
        t = u;
    }
    // XCD x works through the segments of blocks x, x + 8, ... block after block; workgroup g (4 wavefronts) takes
    // slots 4 * (g / 8) .. + 3 of the list of XCD g % 8.  A block's segments keep their order (term by term); a block
    // starts on a multiple of 4 slots (a workgroup never mixes two blocks' rows).  One such order per round.
    std::vector<int64_t> round_slot0(groups.size() + 1, 0);
    std::vector<std::vector<int64_t>> cursor(pieces.size(), std::vector<int64_t>((size_t)NB, 0));
    const int V = (int)cut.word_seg_ptr.size() - 1;
    for (size_t g = 0; g < groups.size(); ++g) {
        int64_t list_len[kXcd] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; b < NB; ++b) {
            int64_t at = list_len[b % kXcd];
            for (size_t t = groups[g].first; t < groups[g].second; ++t) {
                cursor[t][(size_t)b] = at;                // where piece t's segments of block b go: behind the earlier pieces'
                at += pieces[t].per_block[(size_t)b];
            }
            list_len[b % kXcd] = (at + 3) / 4 * 4;
        }
        const int64_t longest = *std::max_element(list_len, list_len + kXcd);
        round_slot0[g + 1] = round_slot0[g] + longest * kXcd;
        Round r;
        const CutPiece& head = pieces[groups[g].first];
        const CutPiece& tail = pieces[groups[g].second - 1];
        r.seg_lo = head.base;
        r.seg_hi = tail.base + (int64_t)tail.begin.size();
        r.w_first = head.v0;
        r.n_words = tail.v0 + (int)tail.per_word.size() - head.v0;
        r.slot_lo = round_slot0[g];
        r.slot_count = longest * kXcd;
        // blocks of 256 statistics of the WHOLE table (a round starts on a multiple of 16 terms and ends on one, or at V)
        r.ent_first = (int64_t)r.w_first * ldk / 256;
        r.ent_blocks = finalize_blocks(r.n_words, ldk);
        plan.rounds.push_back(r);
        plan.partial_rows = std::max(plan.partial_rows, r.seg_hi - r.seg_lo);
    }
    plan.ent_blocks = finalize_blocks(V, ldk);
    plan.order.assign((size_t)round_slot0.back(), -1);
    std::vector<size_t> group_of(pieces.size(), 0);
    for (size_t g = 0; g < groups.size(); ++g)
        for (size_t t = groups[g].first; t < groups[g].second; ++t) group_of[t] = g;
    run_on_threads(std::max(1, (int)pieces.size()), [&](int t) {
        if ((size_t)t >= pieces.size()) return;
        const CutPiece& piece = pieces[(size_t)t];
        std::vector<int64_t>& cur = cursor[(size_t)t];
        const int64_t slot0 = round_slot0[group_of[(size_t)t]];
        for (size_t k = 0; k < piece.block.size(); ++k) {
            const int32_t b = piece.block[k];
            const int64_t i = cur[(size_t)b]++;
            plan.order[(size_t)(slot0 + ((i / 4) * kXcd + b % kXcd) * 4 + i % 4)] = (int32_t)(piece.base + (int64_t)k);
        }
    });
    return plan;
}

}  // namespace pylda_plan
