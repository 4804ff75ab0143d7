// libpylda_hip.so - postings, segments and the sufficient-statistics pass (variational_bayes.py:206-207).
// (host side of the C ABI declared in include/pylda_hip.h; see host_internal.h for the map of the translation units)
#include "host_internal.h"
#include "sstats_kernels.h"
#include "sstats_sweep.h"

namespace pylda_host {

// Geometry of the persistent sweep (sstats_sweep.h) for V terms: terms per wavefront and wavefronts per workgroup (one
// workgroup per CU) such that the fewest passes over the document blocks cover all terms.
struct SweepGeom { int T, WPB, passes; };
static SweepGeom sweep_geom_for(const pylda_ctx* ctx, int V)
{
    const int64_t cus = ctx->num_cu;
    auto passes = [&](int T, int WPB) { return (int)((V + cus * WPB * T - 1) / (cus * WPB * T)); };
    if (ctx->ldk == 128) {
        if (passes(12, 16) == 1) return {12, 16, 1};
        return {16, 16, passes(16, 16)};
    }
    const int a = passes(8, 16), b = passes(12, 12);       // stride 256: 8 VGPRs per term
    return b < a ? SweepGeom{12, 12, b} : SweepGeom{8, 16, a};
}

#define PYLDA_SWEEP_DISPATCH(ctx, c, DO)                                                             \
    do {                                                                                             \
        const int t_ = (c)->sweep_terms, w_ = (c)->sweep_wpb;                                        \
        if ((ctx)->ldk == 128 && t_ == 12) { if ((c)->wide_pos) { DO(2, 12, 16, int64_t); } else { DO(2, 12, 16, int32_t); } } \
        else if ((ctx)->ldk == 128) { if ((c)->wide_pos) { DO(2, 16, 16, int64_t); } else { DO(2, 16, 16, int32_t); } }       \
        else if (w_ == 12) { if ((c)->wide_pos) { DO(4, 12, 12, int64_t); } else { DO(4, 12, 12, int32_t); } }                 \
        else { if ((c)->wide_pos) { DO(4, 8, 16, int64_t); } else { DO(4, 8, 16, int32_t); } }                                \
    } while (0)

// One host thread's share of the segment cut (build_postings): the segments of a contiguous range of terms.
struct CutPiece {
    std::vector<int64_t> begin, end, per_word;
    std::vector<int32_t> block, per_block;      // per_block[b]: this piece's segments in document block b
    int v0 = 0;
    int64_t base = 0;                           // index of its first segment in the whole list
};

template <typename F>
void run_on_threads(int nthreads, F&& fn)
{
    std::vector<std::thread> workers;
    for (int t = 1; t < nthreads; ++t) workers.emplace_back(fn, t);
    fn(0);
    for (auto& w : workers) w.join();
}

// Postings (CSC) of the corpus, built once, on the first training E-step, on the device (postings.hip):
// for every word the (document, CSR position) pairs in document order, cut into segments.
int build_postings(pylda_corpus* c)
{
    if (c->have_postings) return PYLDA_OK;
    pylda_ctx* ctx = c->ctx;
    const int V = ctx->V;
    const int64_t nnz = c->nnz;
    int rc = PYLDA_OK;
    auto A = [&](int r) { if (rc == PYLDA_OK) rc = r; };
    // every exit below that is not the last line leaves the corpus without postings AND without their arrays: a
    // retry (the next training E-step) starts from scratch instead of leaking nnz * 8 bytes or more per attempt
    struct Undo {
        pylda_corpus* c;
        bool keep = false;
        ~Undo()
        {
            if (keep) return;
            dev_free(c->d_post_doc);
            if (c->d_post_pos) (void)hipFree(c->d_post_pos);
            c->d_post_pos = nullptr;
            dev_free(c->d_exec_order); dev_free(c->d_seg_begin); dev_free(c->d_seg_end); dev_free(c->d_word_seg_ptr); dev_free(c->d_partial);
            dev_free(c->d_seg_block); dev_free(c->d_term_of); dev_free(c->d_rendezvous);
            c->sweep = false;
            c->nseg = 0;
            c->exec_slots = 0;
            c->rounds.clear();
        }
    } undo{c};
    PhaseTimer timer;
    c->wide_pos = ctx->wide_postings || nnz > INT32_MAX;
    A(dev_alloc(ctx, &c->d_post_doc, (size_t)nnz));
    if (rc == PYLDA_OK) {
        const size_t bytes = (size_t)std::max<int64_t>(nnz, 1) * (c->wide_pos ? sizeof(int64_t) : sizeof(int32_t));
        const hipError_t ea = hipMalloc(&c->d_post_pos, bytes);
        if (ea != hipSuccess) {
            c->d_post_pos = nullptr;
            rc = fail(ctx, ea == hipErrorOutOfMemory ? PYLDA_ERR_OOM : PYLDA_ERR_HIP, "postings: hipMalloc: %s", hipGetErrorString(ea));
        }
    }
    if (rc != PYLDA_OK) return rc;
    std::vector<int64_t> col_ptr((size_t)V + 1, 0);
    const char* what = "";
    const hipError_t e = build_postings_device(ctx->stream, V, c->D, nnz, c->d_doc_ptr, c->d_term_id, c->d_post_doc,
                                               c->d_post_pos, c->wide_pos, col_ptr.data(), &what);
    if (e != hipSuccess)
        return fail(ctx, e == hipErrorOutOfMemory ? PYLDA_ERR_OOM : PYLDA_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    timer.lap("postings on the device");
    std::vector<int64_t> seg_begin, seg_end, word_seg_ptr((size_t)V + 1, 0);
    seg_begin.reserve((size_t)(nnz / kSegment + V));
    seg_end.reserve((size_t)(nnz / kSegment + V));
    // Document-blocked gather (sstats_kernels.h): NB contiguous document blocks whose t rows fit an XCD's L2,
    // NB a multiple of the 8 XCDs; only for the whole-row kernel, when all of t exceeds one L2 and a
    // (term, block) pair still holds >= 8 postings on average.
    int NB = 1;
    {
        const double t_bytes = (double)c->D * ctx->ldk * sizeof(double);
        const bool rows_kernel = ctx->gather_rows >= 1 && (ctx->ldk == 64 || ctx->ldk == 128 || ctx->ldk == 256);
        const bool bulk_kernel = ctx->gather_rows == 2 && (ctx->ldk == 128 || ctx->ldk == 256);   // (short segments need it)
        if (ctx->gather_blocks > 1 && rows_kernel && V > 0) {
            NB = ctx->gather_blocks;                                  // forced (tests, A/B runs)
        } else if (ctx->gather_blocks < 0 && bulk_kernel && t_bytes > 8.6e6 && V > 0) {
            // automatic: blocks of about one L2 (cfg 3 sweep: 16 -> 1.78 ms, 24 -> 1.62, 32 -> ~1.8, 64 -> 3.1; unblocked 3.03),
            // but no more than leave a (term, block) pair 8 postings on average - every pair costs a partial row
            // (cfg 4, t = 2 GB: 64 blocks 55 ms, 128 53, 256 48, unblocked 65; 240 by this rule); the rows themselves are
            // budgeted per round below
            const int by_l2 = std::max(8, 8 * (int)std::lround(t_bytes / (8 * 4.3e6)));
            const int by_pairs = (int)std::min<double>(1e6, (double)nnz / (8.0 * V)) / 8 * 8;
            NB = std::min(by_l2, by_pairs);
            if (NB < 8) NB = 1;
        }
    }
    // budget of the gather's partial rows per round: option gather_round_mb, else 4 GiB but never more than a quarter of
    // the device memory that is free right now (a shared or nearly full device gets more, smaller rounds instead of an
    // allocation failure)
    double round_budget = 4.0 * 1073741824.0;
    if (ctx->gather_round_mb > 0) {
        round_budget = (double)ctx->gather_round_mb * 1048576.0;
    } else {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) round_budget = std::min(round_budget, (double)free_b / 4.0);
    }
    // (a) the gather kernel family is fixed here, with the postings: the partial rows, the rounds and seg_lo below are
    // sized for it, so a later change of the option must not change the kernel that walks them
    c->gather_rows = ctx->gather_rows;
    using Round = pylda_corpus::Round;
    std::vector<CutPiece> pieces;
    int cut_threads = 1;
    // the persistent sweep (sstats_sweep.h) instead of partial rows: its geometry must be resident, one workgroup per CU
    bool want_sweep = false;
    if (NB > 1 && nnz > 0 && ctx->gather_sweep && (ctx->ldk == 128 || ctx->ldk == 256) && ctx->gather_rows == 2) {
        const SweepGeom g = sweep_geom_for(ctx, V);
        c->sweep_terms = g.T;
        c->sweep_wpb = g.WPB;
        c->sweep_passes = g.passes;
        // (mode 1: only where the (term, block) partial rows - about V x NB of them - would not fit their budget)
        const double budget = round_budget;
        const double rows_bytes = ((double)std::min<int64_t>((int64_t)V * NB, nnz) + (double)nnz / kSegment) * ctx->ldk * sizeof(double);
        int per_cu = 0;
#define SWEEP_OCC(NCH, T, WPB, P) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sstats_sweep_kernel<NCH, T, WPB, P>, kWave * WPB, 0)
        PYLDA_SWEEP_DISPATCH(ctx, c, SWEEP_OCC);
#undef SWEEP_OCC
        want_sweep = per_cu >= 1 && (ctx->gather_sweep == 2 || rows_bytes > budget);
    }
    const int64_t segment_cap = want_sweep ? kSweepSegment : kSegment;
    if (NB > 1) {
        // the documents of the postings come back through a page-locked buffer (0.8 GB at cfg 4: 16 ms instead of the
        // pageable copy's 0.3 s) and the cut runs on all host threads, term ranges side by side (it took 0.4 s)
        int32_t* post_doc = nullptr;
        if (hipHostMalloc(reinterpret_cast<void**>(&post_doc), (size_t)std::max<int64_t>(nnz, 1) * sizeof(int32_t), hipHostMallocDefault) != hipSuccess)
            return fail(ctx, PYLDA_ERR_OOM, "postings: page-locked staging buffer");
        struct Pinned { int32_t* p; ~Pinned() { (void)hipHostFree(p); } } pinned{post_doc};
        timer.lap("page-locked staging buffer");
        if (nnz > 0 && hipMemcpy(post_doc, c->d_post_doc, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(ctx, PYLDA_ERR_HIP, "postings: D2H copy failed");
        timer.lap("documents of the postings D2H");
        const int64_t per_block = (c->D + NB - 1) / NB;
        // (a forced round budget - tests - cuts at least 8 pieces so that small corpora get several rounds as well)
        const int nthreads = (int)std::max<int64_t>(ctx->gather_round_mb > 0 ? 8 : 1,
                                                    std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), 32, nnz / 2000000 + 1}));
        using Piece = CutPiece;
        cut_threads = nthreads;
        pieces.assign((size_t)nthreads, Piece());
        run_on_threads(nthreads, [&](int t) {
            // thread t: the terms whose postings start in its 1 / nthreads share of the posting range
            Piece& out = pieces[(size_t)t];
            out.per_block.assign((size_t)NB, 0);
            const int64_t from = nnz * t / nthreads, to = nnz * (t + 1) / nthreads;
            const int v0 = (int)(std::lower_bound(col_ptr.begin(), col_ptr.end() - 1, from) - col_ptr.begin());
            const int v1 = t + 1 == nthreads ? V : (int)(std::lower_bound(col_ptr.begin(), col_ptr.end() - 1, to) - col_ptr.begin());
            out.v0 = v0;
            out.per_word.reserve((size_t)std::max(0, v1 - v0));
            for (int v = v0; v < v1; ++v) {
                int64_t b = col_ptr[(size_t)v], n = 0;
                while (b < col_ptr[(size_t)v + 1]) {
                    const int32_t blk = (int32_t)(post_doc[(size_t)b] / per_block);
                    const int64_t block_end = ((int64_t)blk + 1) * per_block;       // first document of the next block
                    const int64_t cap = std::min<int64_t>(col_ptr[(size_t)v + 1], b + segment_cap);
                    int64_t e = b + 1;
                    while (e < cap && post_doc[(size_t)e] < block_end) ++e;
                    out.begin.push_back(b);
                    out.end.push_back(e);
                    out.block.push_back(blk);
                    out.per_block[(size_t)blk] += 1;
                    b = e;
                    ++n;
                }
                out.per_word.push_back(n);
            }
        });
        int64_t total = 0;
        int covered = 0;
        for (Piece& piece : pieces) {               // the pieces cover the terms in order
            if (piece.v0 != covered) return fail(ctx, PYLDA_ERR_STATE, "postings: the segment cut lost terms at %d", covered);
            piece.base = total;
            total += (int64_t)piece.begin.size();
            covered += (int)piece.per_word.size();
        }
        if (covered != V) return fail(ctx, PYLDA_ERR_STATE, "postings: the segment cut covered %d of %d terms", covered, V);
        seg_begin.resize((size_t)total);
        seg_end.resize((size_t)total);
        run_on_threads(nthreads, [&](int t) {
            const Piece& piece = pieces[(size_t)t];
            std::copy(piece.begin.begin(), piece.begin.end(), seg_begin.begin() + piece.base);
            std::copy(piece.end.begin(), piece.end.end(), seg_end.begin() + piece.base);
            int64_t at = piece.base;
            for (size_t i = 0; i < piece.per_word.size(); ++i) {
                at += piece.per_word[i];
                word_seg_ptr[(size_t)piece.v0 + i + 1] = at;
            }
        });
        timer.lap("segment cut");
    } else {
        for (int v = 0; v < V; ++v) {
            for (int64_t b = col_ptr[v]; b < col_ptr[v + 1]; b += kSegment) {
                seg_begin.push_back(b);
                seg_end.push_back(std::min<int64_t>(b + kSegment, col_ptr[v + 1]));
            }
            word_seg_ptr[v + 1] = (int64_t)seg_begin.size();
        }
    }
    c->nseg = (int64_t)seg_begin.size();
    c->gather_blocks = NB;
    c->rounds.clear();
    c->sweep = false;
    if (want_sweep && c->nseg > 0) {
        // the persistent sweep: every wavefront owns a few terms, all workgroups walk the document blocks together
        const int T = c->sweep_terms;
        const int64_t nwaves = (int64_t)ctx->num_cu * c->sweep_wpb;
        const int passes = c->sweep_passes;
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
        std::vector<int32_t> seg_block_all((size_t)c->nseg);
        run_on_threads(cut_threads, [&](int t) {
            std::copy(pieces[(size_t)t].block.begin(), pieces[(size_t)t].block.end(), seg_block_all.begin() + pieces[(size_t)t].base);
        });
        A(dev_alloc(ctx, &c->d_seg_block, (size_t)c->nseg));
        A(dev_alloc(ctx, &c->d_term_of, term_of.size()));
        A(dev_alloc(ctx, &c->d_rendezvous, (size_t)kSweepCounters * 32));      // one counter per XCD, a cache line apart
        dev_free(c->d_entropy_partial);
        A(dev_alloc(ctx, &c->d_entropy_partial, (size_t)passes * nwaves));
        if (rc != PYLDA_OK) return rc;
        if (hipMemcpy(c->d_seg_block, seg_block_all.data(), seg_block_all.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(c->d_term_of, term_of.data(), term_of.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
            return fail(ctx, PYLDA_ERR_HIP, "postings: H2D copy failed");
        c->sweep = true;
        c->ent_blocks = (int64_t)passes * nwaves;
        c->partial_rows = 0;
    }
    const int64_t ldk_rows = ctx->ldk;
    auto blocks_of = [&](int n_words) { return ((int64_t)n_words * ldk_rows + 255) / 256; };
    if (c->sweep) {
        // (no partial rows, no execution order)
    } else if (NB > 1 && c->nseg > 0) {
        // rounds: groups of consecutive pieces (contiguous term ranges), each within the budget of partial rows
        const double row_bytes = (double)ctx->ldk * sizeof(double);
        const double budget = round_budget;
        const int64_t max_rows = std::max<int64_t>(1, (int64_t)(budget / row_bytes));
        std::vector<std::pair<size_t, size_t>> groups;        // [first piece, last piece + 1)
        for (size_t t = 0; t < pieces.size();) {
            size_t u = t + 1;
            int64_t rows = (int64_t)pieces[t].begin.size();
            while (u < pieces.size() && rows + (int64_t)pieces[u].begin.size() <= max_rows) rows += (int64_t)pieces[u++].begin.size();
            groups.emplace_back(t, u);
            t = u;
        }
        // XCD x works through the segments of blocks x, x + 8, ... block after block; workgroup g (4 wavefronts)
        // takes slots 4 * (g / 8) .. + 3 of the list of XCD g % 8.  A block's segments keep their order (term by term);
        // a block starts on a multiple of 4 slots (a workgroup never mixes two blocks' rows).  One such order per round.
        constexpr int kXcd = 8;
        std::vector<int64_t> round_slot0(groups.size() + 1, 0);
        std::vector<std::vector<int64_t>> cursor(pieces.size(), std::vector<int64_t>((size_t)NB, 0));
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
            r.ent_first = c->rounds.empty() ? 0 : c->rounds.back().ent_first + c->rounds.back().ent_blocks;
            r.ent_blocks = blocks_of(r.n_words);
            c->rounds.push_back(r);
        }
        std::vector<int32_t> order((size_t)round_slot0.back(), -1);
        std::vector<size_t> group_of(pieces.size(), 0);
        for (size_t g = 0; g < groups.size(); ++g)
            for (size_t t = groups[g].first; t < groups[g].second; ++t) group_of[t] = g;
        run_on_threads(cut_threads, [&](int t) {
            const CutPiece& piece = pieces[(size_t)t];
            std::vector<int64_t>& cur = cursor[(size_t)t];
            const int64_t slot0 = round_slot0[group_of[(size_t)t]];
            for (size_t k = 0; k < piece.block.size(); ++k) {
                const int32_t b = piece.block[k];
                const int64_t i = cur[(size_t)b]++;
                order[(size_t)(slot0 + ((i / 4) * kXcd + b % kXcd) * 4 + i % 4)] = (int32_t)(piece.base + (int64_t)k);
            }
        });
        c->exec_slots = (int64_t)order.size();
        A(dev_alloc(ctx, &c->d_exec_order, order.size()));
        if (rc != PYLDA_OK) return rc;
        if (hipMemcpy(c->d_exec_order, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
            return fail(ctx, PYLDA_ERR_HIP, "postings: H2D copy failed");
    } else {
        c->rounds.push_back(Round{0, c->nseg, 0, V, 0, c->nseg, 0, blocks_of(V)});
    }
    if (!c->sweep) {
        c->partial_rows = 0;
        for (const Round& r : c->rounds) c->partial_rows = std::max(c->partial_rows, r.seg_hi - r.seg_lo);
        c->ent_blocks = c->rounds.back().ent_first + c->rounds.back().ent_blocks;
    }
    timer.lap("XCD execution order");
    A(dev_alloc(ctx, &c->d_seg_begin, (size_t)c->nseg));
    A(dev_alloc(ctx, &c->d_seg_end, (size_t)c->nseg));
    A(dev_alloc(ctx, &c->d_word_seg_ptr, (size_t)V + 1));
    timer.lap("segment array allocations");
    if (!c->sweep) {
        A(dev_alloc(ctx, &c->d_partial, (size_t)c->partial_rows * ctx->ldk));
        dev_free(c->d_entropy_partial);
        A(dev_alloc(ctx, &c->d_entropy_partial, (size_t)c->ent_blocks));
    }
    if (rc != PYLDA_OK) return rc;
    timer.lap("partial rows allocation");
    auto H2D = [&](void* dst, const void* src, size_t bytes) {
        if (rc == PYLDA_OK && bytes && hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(ctx, PYLDA_ERR_HIP, "postings: H2D copy failed");
    };
    H2D(c->d_seg_begin, seg_begin.data(), (size_t)c->nseg * sizeof(int64_t));
    H2D(c->d_seg_end, seg_end.data(), (size_t)c->nseg * sizeof(int64_t));
    H2D(c->d_word_seg_ptr, word_seg_ptr.data(), ((size_t)V + 1) * sizeof(int64_t));
    if (rc != PYLDA_OK) return rc;
    timer.lap("segment arrays H2D");
    c->have_postings = true;
    undo.keep = true;
    return PYLDA_OK;
}

#ifndef PYLDA_GATHER_U
#define PYLDA_GATHER_U 4        // rows in flight per wavefront (cfg 3, 24 blocks: 4 -> 1.62 ms, 8 -> 1.71, 16 -> 2.3: occupancy)
#endif
template <typename P>
void launch_gather(pylda_ctx* ctx, pylda_corpus* c, const pylda_corpus::Round& r)
{
    const int ldk = ctx->ldk;
    const P* pos = static_cast<const P*>(c->d_post_pos);
    const dim3 grid((unsigned)((c->nseg + 3) / 4), (unsigned)((ldk + 63) / 64));
    const int32_t* order = c->d_exec_order ? c->d_exec_order + r.slot_lo : nullptr;
    const dim3 g1((unsigned)((r.slot_count + 3) / 4));
#define GATHER_ARGS c->d_seg_begin, c->d_seg_end, c->nseg, c->d_post_doc, pos, c->d_tfinal, c->d_rfinal
    if (ldk == 16)
        hipLaunchKernelGGL((sstats_gather_kernel<16, P>), grid, dim3(256), 0, ctx->stream, GATHER_ARGS, ldk, c->d_partial);
    else if (ldk == 32)
        hipLaunchKernelGGL((sstats_gather_kernel<32, P>), grid, dim3(256), 0, ctx->stream, GATHER_ARGS, ldk, c->d_partial);
    else if (c->gather_rows && (ldk == 64 || ldk == 128 || ldk == 256)) {
        if (ldk == 128 && c->gather_rows == 2)
            hipLaunchKernelGGL((sstats_gather_bulk_kernel<2, PYLDA_GATHER_U, P>), g1, dim3(256), 0, ctx->stream, GATHER_ARGS, c->d_partial, order, r.seg_lo);
        else if (ldk == 256 && c->gather_rows == 2)
            hipLaunchKernelGGL((sstats_gather_bulk_kernel<4, PYLDA_GATHER_U, P>), g1, dim3(256), 0, ctx->stream, GATHER_ARGS, c->d_partial, order, r.seg_lo);
        else if (ldk == 64)
            hipLaunchKernelGGL((sstats_gather_rows_kernel<1, P>), g1, dim3(256), 0, ctx->stream, GATHER_ARGS, c->d_partial, order, r.seg_lo);
        else if (ldk == 128)
            hipLaunchKernelGGL((sstats_gather_rows_kernel<2, P>), g1, dim3(256), 0, ctx->stream, GATHER_ARGS, c->d_partial, order, r.seg_lo);
        else
            hipLaunchKernelGGL((sstats_gather_rows_kernel<4, P>), g1, dim3(256), 0, ctx->stream, GATHER_ARGS, c->d_partial, order, r.seg_lo);
    } else
        hipLaunchKernelGGL((sstats_gather_kernel<64, P>), grid, dim3(256), 0, ctx->stream, GATHER_ARGS, ldk, c->d_partial);
#undef GATHER_ARGS
}

static void fill_sweep_params(pylda_ctx* ctx, pylda_corpus* c, SweepParams& sp)
{
    sp.seg_begin = c->d_seg_begin;
    sp.seg_end = c->d_seg_end;
    sp.seg_block = c->d_seg_block;
    sp.word_seg_ptr = c->d_word_seg_ptr;
    sp.post_doc = c->d_post_doc;
    sp.post_pos = c->d_post_pos;
    sp.tfinal = c->d_tfinal;
    sp.rfinal = c->d_rfinal;
    sp.expElog = ctx->d_expElog;
    sp.expElog_elog = ctx->d_expElog_elog;
    sp.sstats = ctx->d_sstats;
    sp.entropy_partial = c->d_entropy_partial;
    sp.term_of = c->d_term_of;
    sp.passes = c->sweep_passes;
    sp.NB = c->gather_blocks;
    sp.rendezvous = c->d_rendezvous;
    sp.per_xcd = ctx->sweep_xcd;
    sp.spin_limit = (unsigned)ctx->sweep_spin;       // (4000 ~ 5 ms: a rendezvous that does not complete costs L2 locality, nothing else)
}

int enqueue_sstats_gather(pylda_ctx* ctx, pylda_corpus* c)
{
    const int ldk = ctx->ldk;
    if (c->sweep) {
        SweepParams sp;
        fill_sweep_params(ctx, c, sp);
        (void)hipMemsetAsync(c->d_rendezvous, 0, sizeof(unsigned) * kSweepCounters * 32, ctx->stream);
#define SWEEP_LAUNCH(NCH, T, WPB, P) \
    hipLaunchKernelGGL((sstats_sweep_kernel<NCH, T, WPB, P>), dim3((unsigned)ctx->num_cu), dim3(kWave * WPB), 0, ctx->stream, sp)
        PYLDA_SWEEP_DISPATCH(ctx, c, SWEEP_LAUNCH);
#undef SWEEP_LAUNCH
        HIP_TRY(ctx, hipGetLastError());       // (the entropy partials are summed with the likelihoods: pylda_estep)
        return PYLDA_OK;
    }
    for (const pylda_corpus::Round& r : c->rounds) {
        if (r.seg_hi > r.seg_lo) {
            if (c->wide_pos) launch_gather<int64_t>(ctx, c, r);
            else launch_gather<int32_t>(ctx, c, r);
        }
        if (r.ent_blocks > 0)
            hipLaunchKernelGGL(sstats_finalize_kernel, dim3((unsigned)r.ent_blocks), dim3(256), 0, ctx->stream,
                               c->d_word_seg_ptr, c->d_partial, ctx->d_expElog, ctx->d_expElog_elog, r.w_first, r.n_words, ldk,
                               r.seg_lo, ctx->d_sstats, c->d_entropy_partial + r.ent_first);
    }
    HIP_TRY(ctx, hipGetLastError());           // (the entropy partials are summed with the likelihoods: pylda_estep)
    return PYLDA_OK;
}

}  // namespace pylda_host

