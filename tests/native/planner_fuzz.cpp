// Sanitizer driver for the host-side planner (pylda_amd/csrc/host_plan.cpp): built by tests/test_planner_sanitizers.py
// with g++ -fsanitize=address,undefined and run on the CPU.  Random corpora, vocabularies, table strides, options,
// document-block counts, thread counts and row budgets; every input array lives in an exactly-sized heap buffer (a read
// past its end is an ASan error) and every result is held against the invariants the kernels rely on:
//   * launch classes: the classes tile the schedule, every document sits in exactly one class, its length is within what
//     the class's kernel geometry holds, the geometry code has an instantiation, the LDS request is within the limit;
//   * segment cut: a term's segments partition its postings in order, none crosses a document block or the cap;
//   * rounds: contiguous in segments and terms, every segment exactly once in its round's execution order, a
//     workgroup's four slots in ONE document block of the XCD the workgroup lands on, partial-row indices inside the
//     rows allocated, entropy partials laid out by blocks of 256 statistics of the whole table whatever the rounds;
//   * sweep: every term dealt to exactly one (pass, wavefront, slot).
#include "../../pylda_amd/csrc/host_plan.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace pylda_plan;

static uint64_t state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd()
{
    state ^= state << 13; state ^= state >> 7; state ^= state << 17;
    return (uint32_t)(state >> 32);
}
static int rnd_in(int lo, int hi) { return lo + (int)(rnd() % (uint32_t)(hi - lo + 1)); }

#define REQUIRE(cond, ...)                                          \
    do {                                                            \
        if (!(cond)) {                                              \
            fprintf(stderr, "planner fuzz: %s:%d: %s: ", __FILE__, __LINE__, #cond); \
            fprintf(stderr, __VA_ARGS__);                           \
            fprintf(stderr, "\n");                                  \
            return 1;                                               \
        }                                                           \
    } while (0)

template <typename T>
struct Exact {          // exactly-sized heap copy: ASan red zones sit right behind the last element
    T* p;
    explicit Exact(const std::vector<T>& v) : p((T*)malloc(v.size() ? v.size() * sizeof(T) : 1)) { if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(T)); }
    ~Exact() { free(p); }
};

static int check_launch_classes()
{
    PlanConfig cfg;
    static const int ks[] = {1, 3, 10, 16, 17, 32, 50, 64, 100, 128, 129, 192, 200, 256, 257, 300, 384, 500, 512, 513, 700, 1000, 1024, 1100, 2000};
    cfg.K = rnd() % 3 ? ks[rnd() % (sizeof ks / sizeof ks[0])] : rnd_in(1, 1200);
    cfg.ldk = table_stride_for(cfg.K);
    cfg.V = rnd() % 8 == 0 ? rnd_in(2000000, 40000000) : rnd_in(1, 200000);       // (large: the table passes 4 GiB)
    cfg.num_cu = rnd() % 4 ? 256 : rnd_in(1, 304);
    cfg.lds_limit = rnd() % 4 ? 160 * 1024 : 64 * 1024;
    cfg.exact_stop = rnd() % 6 == 0;
    cfg.force_variant = -1;
    if (rnd() % 3 == 0) {
        static const int forced[] = {0, 1, 2, 3, 4, 6, 9, 10, 11, 12, 13};
        cfg.force_variant = forced[rnd() % (sizeof forced / sizeof forced[0])];
    }
    cfg.quad = rnd() % 8 != 0;
    cfg.quad_stream = rnd() % 4 != 0;
    cfg.quilt12 = rnd() % 4 == 0;
    cfg.quilt_odd = rnd() % 4 != 0;
    cfg.slab_uber = rnd() % 4 != 0;
    const int64_t D = rnd() % 16 == 0 ? 0 : rnd_in(1, 6000);
    std::vector<int32_t> terms((size_t)D);
    for (auto& n : terms) {
        switch (rnd() % 6) {
        case 0: n = rnd_in(1, 60); break;
        case 1: n = rnd_in(100, 230); break;
        case 2: n = rnd_in(220, 260); break;
        case 3: n = rnd_in(250, 1100); break;
        case 4: n = rnd_in(380, 400); break;
        default: n = rnd() % 50 ? rnd_in(1, 400) : rnd_in(1000, 40000); break;
        }
    }
    std::sort(terms.begin(), terms.end(), [](int a, int b) { return a > b; });
    Exact<int32_t> t(terms);
    const std::vector<Launch> plan = build_launch_classes(cfg, t.p, D);
    int64_t at = 0;
    for (const Launch& L : plan) {
        REQUIRE(L.first == at && L.count > 0, "classes tile the schedule (first %lld, expected %lld)", (long long)L.first, (long long)at);
        REQUIRE(L.variant >= 0 && L.variant <= kVariantLast && geometry_is_instantiated(cfg, L.variant, L.rn, L.rk),
                "K %d ldk %d: no kernel for variant %d geometry %d rk %d", cfg.K, cfg.ldk, L.variant, L.rn, L.rk);
        REQUIRE(L.lds_bytes <= cfg.lds_limit, "LDS request %zu above the limit %zu (variant %d)", L.lds_bytes, cfg.lds_limit, L.variant);
        REQUIRE(L.n_cap == std::max(1, terms[(size_t)L.first]), "n_cap is the first (longest) document's");
        const int64_t cap = capacity_of(cfg, L.variant, L.rn, L.rk, L.lds_bytes);
        for (int64_t d = L.first; d < L.first + L.count; ++d)
            REQUIRE(terms[(size_t)d] <= cap && terms[(size_t)d] <= L.n_cap,
                    "K %d ldk %d: a document of %d terms in variant %d geometry %d (holds %lld)", cfg.K, cfg.ldk, terms[(size_t)d], L.variant,
                    L.rn, (long long)cap);
        if (cfg.exact_stop) REQUIRE(L.variant <= kGenericGlobal || L.variant == kGenericHuge, "exact stop test: generic kernels only");
        at += L.count;
    }
    REQUIRE(at == D, "the classes cover %lld of %lld documents", (long long)at, (long long)D);
    const int from = slab_uber_from(cfg, plan);
    if (from >= 0) {
        REQUIRE((size_t)from + 2 <= plan.size(), "the one-dispatch slab group has at least two classes");
        for (size_t i = (size_t)from; i < plan.size(); ++i) REQUIRE(plan[i].variant == kSlab && plan[i].rk == plan.back().rk, "slab classes only");
    }
    return 0;
}

static int check_statistics_plan()
{
    GatherConfig g;
    g.V = rnd() % 10 == 0 ? rnd_in(1, 20) : rnd_in(1, 3000);
    g.D = rnd() % 10 == 0 ? rnd_in(1, 40) : rnd_in(1, 20000);
    static const int strides[] = {16, 32, 64, 128, 256, 384, 512, 1024};
    g.ldk = strides[rnd() % 8];
    g.num_cu = rnd() % 4 ? 256 : rnd_in(8, 304);
    // postings: per term a sorted set of distinct documents
    std::vector<int64_t> col_ptr((size_t)g.V + 1, 0);
    std::vector<int32_t> post_doc;
    for (int v = 0; v < g.V; ++v) {
        const int64_t want = rnd() % 5 == 0 ? 0 : rnd() % 20 == 0 ? rnd_in(1, (int)std::min<int64_t>(g.D, 2000)) : rnd_in(1, (int)std::min<int64_t>(g.D, 40));
        const int64_t stride = std::max<int64_t>(1, g.D / std::max<int64_t>(1, want));
        int64_t d = rnd() % stride;
        for (int64_t i = 0; i < want && d < g.D; ++i) {
            post_doc.push_back((int32_t)d);
            d += 1 + rnd() % (2 * stride);
        }
        col_ptr[(size_t)v + 1] = (int64_t)post_doc.size();
    }
    g.nnz = (int64_t)post_doc.size();
    g.gather_rows = rnd_in(0, 2);
    g.gather_blocks = rnd() % 3 == 0 ? -1 : 8 * rnd_in(0, 8);
    g.gather_sweep = rnd_in(0, 2);
    g.gather_round_mb = rnd() % 3 == 0 ? rnd_in(1, 64) : 0;
    Exact<int64_t> cp(col_ptr);
    Exact<int32_t> pd(post_doc);

    const int automatic = document_blocks(g);
    REQUIRE(automatic >= 1 && (automatic == 1 || automatic % 8 == 0 || g.gather_blocks > 1), "document blocks %d", automatic);
    REQUIRE(round_budget(g, 0) == decision_budget(g) && round_budget(g, 1 << 20) <= decision_budget(g), "budgets");
    const SweepGeom sg = sweep_geometry(g);
    REQUIRE((int64_t)sg.passes * g.num_cu * sg.WPB * sg.T >= g.V, "the sweep's geometry holds every term");
    (void)sweep_wanted(g, automatic, true);
    const int swept = sweep_blocks(g, automatic);
    REQUIRE(swept >= 1 && swept <= automatic && (swept == automatic || swept % 8 == 0), "sweep blocks %d of %d", swept, automatic);

    // ---- blocked cut ----
    const int NB = 8 * rnd_in(1, 9);
    const int64_t cap = rnd() % 2 ? kSweepSegmentCap : kGatherSegment;
    const int nthreads = rnd_in(1, 9);
    SegmentCut cut;
    const char* err = cut_segments_blocked(cp.p, pd.p, g.V, g.D, g.nnz, NB, cap, nthreads, &cut);
    REQUIRE(err == nullptr, "%s", err ? err : "");
    const int64_t per_block = (g.D + NB - 1) / NB;
    const int64_t nseg = (int64_t)cut.seg_begin.size();
    REQUIRE(cut.seg_end.size() == (size_t)nseg && cut.word_seg_ptr.size() == (size_t)g.V + 1 && cut.word_seg_ptr[0] == 0 &&
            cut.word_seg_ptr[(size_t)g.V] == nseg, "segment arrays");
    std::vector<int32_t> seg_block((size_t)nseg, -1);
    int covered = 0;
    for (const CutPiece& piece : cut.pieces) {
        REQUIRE(piece.v0 == covered && (piece.v0 % 16 == 0 || piece.v0 == g.V), "a piece starts at term %d", piece.v0);
        REQUIRE(piece.begin.size() == piece.block.size() && piece.per_block.size() == (size_t)NB, "piece arrays");
        for (size_t k = 0; k < piece.block.size(); ++k) seg_block[(size_t)piece.base + k] = piece.block[k];
        covered += (int)piece.per_word.size();
    }
    REQUIRE(covered == g.V, "pieces cover the terms");
    for (int v = 0; v < g.V; ++v) {
        int64_t at = col_ptr[(size_t)v];
        REQUIRE(cut.word_seg_ptr[(size_t)v] <= cut.word_seg_ptr[(size_t)v + 1], "word_seg_ptr is monotone");
        for (int64_t s = cut.word_seg_ptr[(size_t)v]; s < cut.word_seg_ptr[(size_t)v + 1]; ++s) {
            REQUIRE(cut.seg_begin[(size_t)s] == at && cut.seg_end[(size_t)s] > at && cut.seg_end[(size_t)s] - at <= cap, "segment %lld of term %d", (long long)s, v);
            const int32_t blk = seg_block[(size_t)s];
            REQUIRE(blk >= 0 && blk < NB, "segment block");
            for (int64_t i = at; i < cut.seg_end[(size_t)s]; ++i)
                REQUIRE(post_doc[(size_t)i] / per_block == blk, "posting %lld outside its segment's document block", (long long)i);
            at = cut.seg_end[(size_t)s];
        }
        REQUIRE(at == col_ptr[(size_t)v + 1], "the segments of term %d partition its postings", v);
    }

    // ---- rounds ----
    const int64_t max_rows = rnd() % 4 == 0 ? 1 : rnd() % 3 == 0 ? nseg + 5 : rnd_in(1, (int)std::max<int64_t>(1, nseg));
    const RoundPlan plan = plan_rounds(cut, NB, max_rows, g.ldk);
    REQUIRE(plan.ent_blocks == finalize_blocks(g.V, g.ldk), "entropy partials: blocks of the whole table");
    int64_t seg_at = 0, slot_at = 0;
    int w_at = 0;
    std::vector<int> seen((size_t)nseg, 0);
    for (const Round& r : plan.rounds) {
        REQUIRE(r.seg_lo == seg_at && r.seg_hi >= r.seg_lo && r.w_first == w_at && r.n_words >= 0, "rounds are contiguous");
        REQUIRE(r.seg_hi - r.seg_lo <= plan.partial_rows, "partial rows of a round within the allocation");
        REQUIRE(r.slot_lo == slot_at && r.slot_count % (4 * kXcd) == 0 && r.slot_lo + r.slot_count <= (int64_t)plan.order.size(), "slots of a round");
        REQUIRE(((int64_t)r.w_first * g.ldk) % 256 == 0 && r.ent_first == (int64_t)r.w_first * g.ldk / 256 &&
                r.ent_blocks == finalize_blocks(r.n_words, g.ldk) && r.ent_first + r.ent_blocks <= plan.ent_blocks, "entropy blocks of a round");
        if (r.w_first + r.n_words < g.V) REQUIRE(((int64_t)r.n_words * g.ldk) % 256 == 0, "a round inside the table ends on a block of 256 statistics");
        for (int64_t s = r.seg_lo; s < r.seg_hi; ++s)
            REQUIRE(s >= cut.word_seg_ptr[(size_t)r.w_first] && s < cut.word_seg_ptr[(size_t)(r.w_first + r.n_words)], "a round's segments are its terms'");
        for (int64_t wg = 0; wg < r.slot_count / 4; ++wg) {
            int32_t blk = -1;
            for (int x = 0; x < 4; ++x) {
                const int32_t s = plan.order[(size_t)(r.slot_lo + 4 * wg + x)];
                if (s < 0) continue;
                REQUIRE(s >= r.seg_lo && s < r.seg_hi, "slot holds a segment of another round");
                REQUIRE(s - r.seg_lo < plan.partial_rows, "partial row index");
                seen[(size_t)s] += 1;
                REQUIRE(blk < 0 || blk == seg_block[(size_t)s], "a workgroup's four slots mix document blocks");
                blk = seg_block[(size_t)s];
                REQUIRE(blk % kXcd == wg % kXcd, "block %d on the list of XCD %d", blk, (int)(wg % kXcd));
            }
        }
        seg_at = r.seg_hi;
        slot_at += r.slot_count;
        w_at += r.n_words;
    }
    REQUIRE(seg_at == nseg && w_at == g.V && slot_at == (int64_t)plan.order.size(), "the rounds cover segments, terms and slots");
    for (int64_t s = 0; s < nseg; ++s) REQUIRE(seen[(size_t)s] == 1, "segment %lld appears %d times in the execution order", (long long)s, seen[(size_t)s]);
    if (max_rows > 0 && plan.rounds.size() > 1)
        for (const Round& r : plan.rounds) {          // a round above the budget is a single piece (cannot be split further)
            if (r.seg_hi - r.seg_lo <= max_rows) continue;
            int pieces_in = 0;
            for (const CutPiece& piece : cut.pieces)
                if (piece.base >= r.seg_lo && piece.base + (int64_t)piece.begin.size() <= r.seg_hi && !piece.begin.empty()) ++pieces_in;
            REQUIRE(pieces_in <= 1, "a round of %lld rows above the budget %lld holds %d pieces", (long long)(r.seg_hi - r.seg_lo), (long long)max_rows, pieces_in);
        }

    // ---- unblocked cut, single round ----
    SegmentCut plain;
    cut_segments_plain(cp.p, g.V, &plain);
    for (int v = 0; v < g.V; ++v) {
        int64_t at = col_ptr[(size_t)v];
        for (int64_t s = plain.word_seg_ptr[(size_t)v]; s < plain.word_seg_ptr[(size_t)v + 1]; ++s) {
            REQUIRE(plain.seg_begin[(size_t)s] == at && plain.seg_end[(size_t)s] - at <= kGatherSegment && plain.seg_end[(size_t)s] > at, "plain cut");
            at = plain.seg_end[(size_t)s];
        }
        REQUIRE(at == col_ptr[(size_t)v + 1], "plain cut partitions term %d", v);
    }
    const RoundPlan one = single_round((int64_t)plain.seg_begin.size(), g.V, g.ldk);
    REQUIRE(one.rounds.size() == 1 && one.partial_rows == (int64_t)plain.seg_begin.size() && one.order.empty(), "single round");

    // ---- sweep dealing ----
    const int64_t nwaves = (int64_t)g.num_cu * sg.WPB;
    const std::vector<int32_t> term_of = deal_terms(cp.p, g.V, nwaves, sg.T, sg.passes);
    REQUIRE(term_of.size() == (size_t)sg.passes * nwaves * sg.T, "term_of size");
    std::vector<int> dealt((size_t)g.V, 0);
    for (int32_t t : term_of)
        if (t >= 0) {
            REQUIRE(t < g.V, "term id");
            dealt[(size_t)t] += 1;
        }
    for (int v = 0; v < g.V; ++v) REQUIRE(dealt[(size_t)v] == 1, "term %d dealt %d times", v, dealt[(size_t)v]);
    return 0;
}

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    for (int i = 0; i < 8 * rounds; ++i)
        if (check_launch_classes()) return 1;
    for (int i = 0; i < rounds; ++i)
        if (check_statistics_plan()) return 1;
    printf("planner sanitizer run: ok (%d launch plans, %d statistics plans)\n", 8 * rounds, rounds);
    return 0;
}
