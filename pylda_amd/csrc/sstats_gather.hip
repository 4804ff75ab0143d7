// libpylda_hip.so - postings, segments and the sufficient-statistics pass (variational_bayes.py:206-207).
// (host side of the C ABI declared in include/pylda_hip.h; see host_internal.h for the map of the translation units)
#include "host_internal.h"
#include "sstats_kernels.h"
#include "sstats_sweep.h"
#include "sstats_live.h"

namespace pylda_host {

#define PYLDA_SWEEP_DISPATCH(ctx, c, DO)                                                             \
    do {                                                                                             \
        const int t_ = (c)->sweep_terms, w_ = (c)->sweep_wpb;                                        \
        if ((ctx)->ldk == 128 && t_ == 12) { if ((c)->wide_pos) { DO(2, 12, 16, int64_t); } else { DO(2, 12, 16, int32_t); } } \
        else if ((ctx)->ldk == 128) { if ((c)->wide_pos) { DO(2, 16, 16, int64_t); } else { DO(2, 16, 16, int32_t); } }       \
        else if (w_ == 12) { if ((c)->wide_pos) { DO(4, 12, 12, int64_t); } else { DO(4, 12, 12, int32_t); } }                 \
        else { if ((c)->wide_pos) { DO(4, 8, 16, int64_t); } else { DO(4, 8, 16, int32_t); } }                                \
    } while (0)

static_assert(kGatherSegment == kSegment && kSweepSegmentCap == kSweepSegment, "the planner cuts what the kernels walk");

namespace {

GatherConfig gather_config(const pylda_ctx* ctx, const pylda_corpus* c)
{
    GatherConfig g;
    g.V = ctx->V;
    g.ldk = ctx->ldk;
    g.num_cu = ctx->num_cu;
    g.D = c->D;
    g.nnz = c->nnz;
    g.gather_rows = ctx->gather_rows;
    g.gather_blocks = ctx->gather_blocks;
    g.gather_sweep = ctx->gather_sweep;
    g.gather_round_mb = ctx->gather_round_mb;
    return g;
}

// every exit of build_postings that is not its last line leaves the corpus without postings AND without their arrays: a
// retry (the next training E-step) starts from scratch instead of leaking nnz * 8 bytes or more per attempt
struct PostingsUndo {
    pylda_corpus* c;
    bool keep = false;
    ~PostingsUndo()
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
};

}  // namespace

// (the buffers are idle: hipFree waits for the device)
void release_postings(pylda_corpus* c)
{
    { PostingsUndo undo{c}; }
    c->have_postings = false;
    c->live_stats = false;
}

namespace {

template <typename T>
int upload(pylda_ctx* ctx, T** dst, const std::vector<T>& src, size_t at_least = 0)
{
    const int rc = dev_alloc(ctx, dst, std::max(src.size(), at_least));
    if (rc != PYLDA_OK) return rc;
    if (!src.empty() && hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
        return fail(ctx, PYLDA_ERR_HIP, "postings: H2D copy failed");
    return PYLDA_OK;
}

// the CSC index itself, on the device (postings.hip); col_ptr comes back to the host
int device_postings(pylda_ctx* ctx, pylda_corpus* c, std::vector<int64_t>* col_ptr)
{
    const int64_t nnz = c->nnz;
    c->wide_pos = ctx->wide_postings || nnz > INT32_MAX;
    int rc = dev_alloc(ctx, &c->d_post_doc, (size_t)nnz);
    if (rc != PYLDA_OK) return rc;
    const size_t bytes = (size_t)std::max<int64_t>(nnz, 1) * (c->wide_pos ? sizeof(int64_t) : sizeof(int32_t));
    const hipError_t ea = hipMalloc(&c->d_post_pos, bytes);
    if (ea != hipSuccess) {
        c->d_post_pos = nullptr;
        return fail(ctx, ea == hipErrorOutOfMemory ? PYLDA_ERR_OOM : PYLDA_ERR_HIP, "postings: hipMalloc: %s", hipGetErrorString(ea));
    }
    col_ptr->assign((size_t)ctx->V + 1, 0);
    const char* what = "";
    const hipError_t e = build_postings_device(ctx->stream, ctx->V, c->D, nnz, c->d_doc_ptr, c->d_term_id, c->d_post_doc,
                                               c->d_post_pos, c->wide_pos, col_ptr->data(), &what);
    if (e != hipSuccess)
        return fail(ctx, e == hipErrorOutOfMemory ? PYLDA_ERR_OOM : PYLDA_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return PYLDA_OK;
}

// the documents of the postings on the host, through a page-locked buffer (0.8 GB at cfg 4: 16 ms instead of the
// pageable copy's 0.3 s)
struct PinnedDocs {
    int32_t* p = nullptr;
    ~PinnedDocs() { if (p) (void)hipHostFree(p); }
    int fetch(pylda_ctx* ctx, const pylda_corpus* c)
    {
        if (hipHostMalloc(reinterpret_cast<void**>(&p), (size_t)std::max<int64_t>(c->nnz, 1) * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) {
            p = nullptr;
            return fail(ctx, PYLDA_ERR_OOM, "postings: page-locked staging buffer");
        }
        if (c->nnz > 0 && hipMemcpy(p, c->d_post_doc, (size_t)c->nnz * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(ctx, PYLDA_ERR_HIP, "postings: D2H copy failed");
        return PYLDA_OK;
    }
};

// the persistent sweep: every wavefront owns a few terms, all workgroups walk the document blocks together
int install_sweep(pylda_ctx* ctx, pylda_corpus* c, const SegmentCut& cut, const std::vector<int64_t>& col_ptr)
{
    const int64_t nwaves = (int64_t)ctx->num_cu * c->sweep_wpb;
    const std::vector<int32_t> term_of = deal_terms(col_ptr.data(), ctx->V, nwaves, c->sweep_terms, c->sweep_passes);
    std::vector<int32_t> seg_block((size_t)c->nseg);
    run_on_threads((int)cut.pieces.size(), [&](int t) {
        std::copy(cut.pieces[(size_t)t].block.begin(), cut.pieces[(size_t)t].block.end(), seg_block.begin() + cut.pieces[(size_t)t].base);
    });
    int rc = upload(ctx, &c->d_seg_block, seg_block);
    if (rc == PYLDA_OK) rc = upload(ctx, &c->d_term_of, term_of);
    if (rc == PYLDA_OK) rc = dev_alloc(ctx, &c->d_rendezvous, (size_t)kSweepCounters * 32);      // one counter per XCD, a cache line apart
    dev_free(c->d_entropy_partial);
    if (rc == PYLDA_OK) rc = dev_alloc(ctx, &c->d_entropy_partial, (size_t)c->sweep_passes * nwaves);
    if (rc != PYLDA_OK) return rc;
    c->sweep = true;
    c->ent_blocks = (int64_t)c->sweep_passes * nwaves;
    c->partial_rows = 0;
    return PYLDA_OK;
}

// the dispatch-paced gather: rounds, their XCD execution order, the partial rows
int install_rounds(pylda_ctx* ctx, pylda_corpus* c, const RoundPlan& plan)
{
    c->rounds = plan.rounds;
    c->partial_rows = plan.partial_rows;
    c->ent_blocks = plan.ent_blocks;
    c->exec_slots = (int64_t)plan.order.size();
    int rc = PYLDA_OK;
    if (!plan.order.empty()) rc = upload(ctx, &c->d_exec_order, plan.order);
    if (rc == PYLDA_OK) rc = dev_alloc(ctx, &c->d_partial, (size_t)c->partial_rows * ctx->ldk);
    dev_free(c->d_entropy_partial);
    if (rc == PYLDA_OK) rc = dev_alloc(ctx, &c->d_entropy_partial, (size_t)c->ent_blocks);
    return rc;
}

}  // namespace

// Postings (CSC) of the corpus, built once, on the first training E-step, on the device (postings.hip): for every word
// the (document, CSR position) pairs in document order, cut into segments.  The layout decisions - document blocks,
// segment cut, sweep or rounds, execution order - are the planner's (host_plan.cpp); this function moves the data.
int build_postings(pylda_corpus* c)
{
    if (c->have_postings) return PYLDA_OK;
    pylda_ctx* ctx = c->ctx;
    const int V = ctx->V;
    const int64_t nnz = c->nnz;
    PostingsUndo undo{c};
    PhaseTimer timer;
    std::vector<int64_t> col_ptr;
    int rc = device_postings(ctx, c, &col_ptr);
    if (rc != PYLDA_OK) return rc;
    timer.lap("postings on the device");

    const GatherConfig g = gather_config(ctx, c);
    // Documents the live-topic kernel finishes leave a LIST of their live topics instead of a row of t (sstats_live.h):
    // with that the pass is one plain walk of the postings - no document blocks, no sweep.  Decided here, with the
    // postings: the corpus hands documents over at this, its first training E-step (prepare_compact ran before us).
    c->live_stats = ctx->gather_live && c->compact_ready && (size_t)4 * ctx->ldk * sizeof(double) <= 64 * 1024;
    int NB = c->live_stats ? 1 : document_blocks(g);
    // (the gather kernel family is fixed here, with the postings: the partial rows, the rounds and seg_lo are sized for it,
    //  so a later change of the option must not change the kernel that walks them)
    c->gather_rows = ctx->gather_rows;
    // the persistent sweep (sstats_sweep.h) instead of partial rows: its geometry must be resident, one workgroup per CU
    const SweepGeom sg = sweep_geometry(g);
    c->sweep_terms = sg.T;
    c->sweep_wpb = sg.WPB;
    c->sweep_passes = sg.passes;
    int per_cu = 0;
    if (NB > 1 && (ctx->ldk == 128 || ctx->ldk == 256)) {
#define SWEEP_OCC(NCH, T, WPB, P) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sstats_sweep_kernel<NCH, T, WPB, P>, kWave * WPB, 0)
        PYLDA_SWEEP_DISPATCH(ctx, c, SWEEP_OCC);
#undef SWEEP_OCC
    }
    const bool want_sweep = !c->live_stats && sweep_wanted(g, document_blocks(g), per_cu >= 1);
    if (want_sweep) NB = sweep_blocks(g, NB);

    SegmentCut cut;
    if (NB > 1) {
        PinnedDocs docs;
        rc = docs.fetch(ctx, c);
        if (rc != PYLDA_OK) return rc;
        timer.lap("documents of the postings D2H");
        // (a forced round budget - tests - cuts at least 8 pieces so that small corpora get several rounds as well)
        const int nthreads = (int)std::max<int64_t>(ctx->gather_round_mb > 0 ? 8 : 1,
                                                    std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), 32, nnz / 2000000 + 1}));
        const char* err = cut_segments_blocked(col_ptr.data(), docs.p, V, c->D, nnz, NB, want_sweep ? kSweepSegment : kSegment, nthreads, &cut);
        if (err) return fail(ctx, PYLDA_ERR_STATE, "postings: %s", err);
        timer.lap("segment cut");
    } else {
        cut_segments_plain(col_ptr.data(), V, &cut);
    }
    c->nseg = (int64_t)cut.seg_begin.size();
    c->gather_blocks = NB;
    c->rounds.clear();
    c->sweep = false;
    if (want_sweep && c->nseg > 0) {
        rc = install_sweep(ctx, c, cut, col_ptr);
    } else if (NB > 1 && c->nseg > 0) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        const int64_t max_rows = (int64_t)(round_budget(g, free_b) / ((double)ctx->ldk * sizeof(double)));
        rc = install_rounds(ctx, c, plan_rounds(cut, NB, max_rows, ctx->ldk));
    } else {
        rc = install_rounds(ctx, c, single_round(c->nseg, V, ctx->ldk));
    }
    if (rc != PYLDA_OK) return rc;
    timer.lap("execution order, partial rows");
    rc = upload(ctx, &c->d_seg_begin, cut.seg_begin);
    if (rc == PYLDA_OK) rc = upload(ctx, &c->d_seg_end, cut.seg_end);
    if (rc == PYLDA_OK) rc = upload(ctx, &c->d_word_seg_ptr, cut.word_seg_ptr);
    if (rc != PYLDA_OK) return rc;
    timer.lap("segment arrays H2D");
    c->have_postings = true;
    undo.keep = true;
    return PYLDA_OK;
}

#ifndef PYLDA_LIVE_U
#define PYLDA_LIVE_U 4         // postings whose lists are in flight together in the live-list pass (sstats_live.h)
#endif
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
    sp.per_block = (int)((c->D + c->gather_blocks - 1) / std::max(1, c->gather_blocks));
    sp.sub = std::max(1, ctx->sweep_sub);
    sp.spin_limit = (unsigned)ctx->sweep_spin;       // (4000 ~ 5 ms: a rendezvous that does not complete costs L2 locality, nothing else)
}

int enqueue_sstats_gather(pylda_ctx* ctx, pylda_corpus* c)
{
    const int ldk = ctx->ldk;
    if (c->live_stats) {
        // (every E-step of this corpus: a document without a list - live_n = -1, reset by pylda_estep - adds its row)
        const pylda_corpus::Round& r = c->rounds.front();
        const dim3 grid((unsigned)((c->nseg + 3) / 4));
        const size_t lds = (size_t)4 * ldk * sizeof(double);
        if (c->nseg > 0) {
#define LIVE_ARGS c->d_seg_begin, c->d_seg_end, c->nseg, c->d_post_doc
#define LIVE_TAIL c->d_tfinal, c->d_rfinal, c->d_live_n, c->d_live_list, ldk, c->d_partial
            if (c->wide_pos)
                hipLaunchKernelGGL((sstats_gather_live_kernel<PYLDA_LIVE_U, int64_t>), grid, dim3(256), lds, ctx->stream, LIVE_ARGS,
                                   static_cast<const int64_t*>(c->d_post_pos), LIVE_TAIL);
            else
                hipLaunchKernelGGL((sstats_gather_live_kernel<PYLDA_LIVE_U, int32_t>), grid, dim3(256), lds, ctx->stream, LIVE_ARGS,
                                   static_cast<const int32_t*>(c->d_post_pos), LIVE_TAIL);
#undef LIVE_ARGS
#undef LIVE_TAIL
        }
        if (r.ent_blocks > 0)
            hipLaunchKernelGGL(sstats_finalize_kernel, dim3((unsigned)r.ent_blocks), dim3(256), 0, ctx->stream,
                               c->d_word_seg_ptr, c->d_partial, ctx->d_expElog, ctx->d_expElog_elog, r.w_first, r.n_words, ldk,
                               r.seg_lo, ctx->d_sstats, c->d_entropy_partial + r.ent_first);
        HIP_TRY(ctx, hipGetLastError());
        return PYLDA_OK;
    }
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

