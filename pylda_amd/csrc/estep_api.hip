// libpylda_hip.so - corpora and the E-step entry points (variational_bayes.py:132-216).
// (host side of the C ABI declared in include/pylda_hip.h; see host_internal.h for the map of the translation units)
#include "host_internal.h"
#include "estep_logspace.h"
#include "doc_terms.h"
#include "prepare_kernels.h"

namespace {

int enqueue_prepare(pylda_ctx* ctx, bool heldout)
{
    const int K = ctx->K, V = ctx->V;
    hipLaunchKernelGGL(eta_rowsum_psi_kernel, dim3(K), dim3(256), 0, ctx->stream, ctx->d_eta, K, V,
                       ctx->d_psi_rowsum);
    if (K <= 64 && (int64_t)K * V <= (1 << 21)) {
        hipLaunchKernelGGL(elog_rows_small_kernel, dim3((V + 3) / 4), dim3(256), 0, ctx->stream, ctx->d_eta, ctx->d_psi_rowsum,
                           K, V, ctx->ldk, ctx->d_elog, ctx->d_expElog, ctx->d_expElog_elog, ctx->d_shift);
    } else {
        hipLaunchKernelGGL(elog_transpose_kernel, dim3((V + 31) / 32, (K + 31) / 32), dim3(256), 0,
                           ctx->stream, ctx->d_eta, ctx->d_psi_rowsum, K, V, ctx->ldk, ctx->d_elog);
        hipLaunchKernelGGL(row_shift_exp_kernel, dim3((V + 3) / 4), dim3(256), 0, ctx->stream,
                           ctx->d_elog, K, V, ctx->ldk, ctx->d_expElog, ctx->d_expElog_elog, ctx->d_shift);
    }
    if (heldout)
        hipLaunchKernelGGL(topic_lse_kernel, dim3(K), dim3(256), 0, ctx->stream, ctx->d_elog,
                           ctx->d_shift, K, V, ctx->ldk, ctx->d_topic_lse);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

}  // namespace

extern "C" {

int pylda_corpus_create(pylda_ctx* ctx, int64_t D, const int64_t* doc_ptr, const int32_t* term_id,
                        const int32_t* term_ct, pylda_corpus** out)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!out) return fail(ctx, PYLDA_ERR_INVALID, "corpus_create: out is NULL");
    *out = nullptr;
    if (D < 0 || D > INT32_MAX || !doc_ptr)
        return fail(ctx, PYLDA_ERR_INVALID, "corpus_create: D=%lld", (long long)D);
    if (doc_ptr[0] != 0) return fail(ctx, PYLDA_ERR_INVALID, "corpus_create: doc_ptr[0] != 0");
    PhaseTimer timer;
    int64_t max_terms = 0;
    for (int64_t d = 0; d < D; ++d) {
        const int64_t n = doc_ptr[d + 1] - doc_ptr[d];
        if (n < 0) return fail(ctx, PYLDA_ERR_INVALID, "corpus_create: doc_ptr not monotone at %lld", (long long)d);
        max_terms = std::max(max_terms, n);
    }
    const int64_t nnz = doc_ptr[D];
    {
        // the most general kernel (estep_generic.h MODE 2) needs only K-sized arrays in LDS: any document length
        const size_t need = generic_lds_layout(ctx->K, 0, tile_stride_for(ctx->K), 256, true).total;
        if (need > ctx->lds_limit || logspace_lds_bytes(ctx->K) > ctx->lds_limit)
            return fail(ctx, PYLDA_ERR_INVALID, "corpus_create: K=%d needs %zu bytes of LDS per document (limit %zu)", ctx->K,
                        std::max(need, logspace_lds_bytes(ctx->K)), ctx->lds_limit);
        if (max_terms > INT32_MAX)
            return fail(ctx, PYLDA_ERR_INVALID, "corpus_create: a document has %lld distinct terms", (long long)max_terms);
    }
    if (nnz > ((int64_t)1 << 36))       // (8 bytes of r_dn per pair alone: beyond one device's memory)
        return fail(ctx, PYLDA_ERR_INVALID, "corpus_create: %lld distinct (doc, term) pairs; shard the corpus", (long long)nnz);
    if (nnz > 0 && (!term_id || !term_ct))
        return fail(ctx, PYLDA_ERR_INVALID, "corpus_create: NULL term arrays");
    int64_t tokens = 0;
    {
        // term ids in range, counts >= 1, token total: on all host threads (198 M pairs at cfg 4)
        const int nthreads = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), 16, nnz / 4000000 + 1}));
        std::vector<int64_t> bad_at((size_t)nthreads, -1), part((size_t)nthreads, 0);
        auto check = [&](int t) {
            const int64_t from = nnz * t / nthreads, to = nnz * (t + 1) / nthreads;
            const int V = ctx->V;
            int64_t sum = 0;
            for (int64_t i = from; i < to; ++i) {
                if ((unsigned)term_id[i] >= (unsigned)V || term_ct[i] < 1) {
                    bad_at[(size_t)t] = i;
                    return;
                }
                sum += term_ct[i];
            }
            part[(size_t)t] = sum;
        };
        std::vector<std::thread> workers;
        for (int t = 1; t < nthreads; ++t) workers.emplace_back(check, t);
        check(0);
        for (auto& w : workers) w.join();
        for (int t = 0; t < nthreads; ++t) {
            const int64_t i = bad_at[(size_t)t];
            if (i >= 0) {
                if (term_id[i] < 0 || term_id[i] >= ctx->V)
                    return fail(ctx, PYLDA_ERR_INVALID, "corpus_create: term id %d at %lld outside [0,%d)",
                                term_id[i], (long long)i, ctx->V);
                return fail(ctx, PYLDA_ERR_INVALID, "corpus_create: count %d at %lld", term_ct[i], (long long)i);
            }
            tokens += part[(size_t)t];
        }
    }
    timer.lap("corpus validation");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pylda_corpus* c = new (std::nothrow) pylda_corpus;
    if (!c) return fail(ctx, PYLDA_ERR_OOM, "corpus_create: host allocation failed");
    c->ctx = ctx;
    c->D = D;
    c->nnz = nnz;
    c->tokens = tokens;
    c->max_terms = (int32_t)max_terms;

    // schedule: longest documents first (stable => deterministic)
    std::vector<int32_t> order((size_t)D);
    if (max_terms <= (int64_t)4 << 20) {
        // counting sort by distinct-term count, descending, documents of equal length in corpus order
        std::vector<int64_t> at((size_t)max_terms + 2, 0);
        for (int64_t d = 0; d < D; ++d) at[(size_t)(max_terms - (doc_ptr[d + 1] - doc_ptr[d])) + 1] += 1;
        for (int64_t n = 0; n <= max_terms; ++n) at[(size_t)n + 1] += at[(size_t)n];
        for (int64_t d = 0; d < D; ++d) order[(size_t)at[(size_t)(max_terms - (doc_ptr[d + 1] - doc_ptr[d]))]++] = (int32_t)d;
    } else {
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            return doc_ptr[a + 1] - doc_ptr[a] > doc_ptr[b + 1] - doc_ptr[b];
        });
    }
    c->h_terms_sorted.resize((size_t)D);
    for (int64_t i = 0; i < D; ++i)
        c->h_terms_sorted[i] = (int32_t)(doc_ptr[order[i] + 1] - doc_ptr[order[i]]);
    c->h_order = order;
    build_plan(c);
    timer.lap("schedule (sort + launch plan)");

    const int K = ctx->K;
    int rc = PYLDA_OK;
    auto A = [&](int r) { if (rc == PYLDA_OK) rc = r; };
    A(dev_alloc(ctx, &c->d_doc_ptr, (size_t)D + 1));
    A(dev_alloc(ctx, &c->d_term_id, (size_t)nnz));
    A(dev_alloc(ctx, &c->d_term_ct, (size_t)nnz));
    A(dev_alloc(ctx, &c->d_order, (size_t)D));
    A(dev_alloc(ctx, &c->d_gamma, (size_t)D * K));
    A(dev_alloc(ctx, &c->d_doc_ll, (size_t)D));
    A(dev_alloc(ctx, &c->d_doc_wll, (size_t)D));
    A(dev_alloc(ctx, &c->d_iters, (size_t)D));
    A(dev_alloc(ctx, &c->d_status, (size_t)D));
    A(dev_alloc(ctx, &c->d_scalars, (size_t)4));
    // (the count of flagged documents lives in the fourth scalar's bytes: ONE read-back of 32 bytes per E-step)
    c->d_flag_count = c->d_scalars ? reinterpret_cast<int32_t*>(c->d_scalars + 3) : nullptr;
    A(dev_alloc(ctx, &c->d_entropy_partial, (size_t)(((int64_t)ctx->V * ctx->ldk + 255) / 256)));
    A(dev_alloc(ctx, &c->d_tfinal, (size_t)D * ctx->ldk));
    A(dev_alloc(ctx, &c->d_rfinal, (size_t)nnz));
    if (rc != PYLDA_OK) {
        pylda_corpus_destroy(c);
        return rc;
    }
    auto H2D = [&](void* dst, const void* src, size_t bytes) {
        if (rc == PYLDA_OK && bytes)
            if (hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess)
                rc = fail(ctx, PYLDA_ERR_HIP, "corpus_create: H2D copy failed");
    };
    H2D(c->d_doc_ptr, doc_ptr, ((size_t)D + 1) * sizeof(int64_t));
    H2D(c->d_term_id, term_id, (size_t)nnz * sizeof(int32_t));
    H2D(c->d_term_ct, term_ct, (size_t)nnz * sizeof(int32_t));
    H2D(c->d_order, order.data(), (size_t)D * sizeof(int32_t));
    if (rc != PYLDA_OK) {
        pylda_corpus_destroy(c);
        return rc;
    }
    timer.lap("allocations + corpus H2D");
    *out = c;
    return PYLDA_OK;
}

void pylda_corpus_destroy(pylda_corpus* c)
{
    if (!c) return;
    if (c->ctx) {
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
    }
    dev_free(c->d_doc_ptr); dev_free(c->d_term_id); dev_free(c->d_term_ct); dev_free(c->d_order);
    dev_free(c->d_gamma); dev_free(c->d_doc_ll); dev_free(c->d_doc_wll); dev_free(c->d_iters);
    dev_free(c->d_status); c->d_flag_count = nullptr; dev_free(c->d_scalars); dev_free(c->d_entropy_partial);
    dev_free(c->d_tfinal); dev_free(c->d_rfinal); dev_free(c->d_term_scratch); dev_free(c->d_post_doc);
    if (c->d_post_pos) (void)hipFree(c->d_post_pos);
    c->d_post_pos = nullptr;
    dev_free(c->d_seg_begin); dev_free(c->d_seg_end); dev_free(c->d_word_seg_ptr); dev_free(c->d_partial); dev_free(c->d_exec_order);
    dev_free(c->d_seg_block); dev_free(c->d_term_of); dev_free(c->d_rendezvous);
    dev_free(c->d_live_n); dev_free(c->d_live_list); dev_free(c->d_tile_ptr); dev_free(c->d_live_tile);
    dev_free(c->d_handoff_it); dev_free(c->d_col_iters);
    delete c;
}

int pylda_corpus_info(const pylda_corpus* c, int64_t* D, int64_t* nnz, int64_t* tokens,
                      int32_t* max_terms)
{
    if (!c) return PYLDA_ERR_INVALID;
    if (D) *D = c->D;
    if (nnz) *nnz = c->nnz;
    if (tokens) *tokens = c->tokens;
    if (max_terms) *max_terms = c->max_terms;
    return PYLDA_OK;
}

int pylda_estep(pylda_ctx* ctx, pylda_corpus* c, int max_iter, double tol, int heldout)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!c || c->ctx != ctx) return fail(ctx, PYLDA_ERR_INVALID, "estep: corpus does not belong to this context");
    if (max_iter < 1) return fail(ctx, PYLDA_ERR_INVALID, "estep: local_parameter_iteration=%d (must be >= 1)", max_iter);
    if (!(tol >= 0.0) && !(tol < 0.0)) return fail(ctx, PYLDA_ERR_INVALID, "estep: threshold is NaN");
    if (!ctx->have_eta || !ctx->have_alpha)
        return fail(ctx, PYLDA_ERR_STATE, "estep: set_eta and set_alpha must be called first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int K = ctx->K, V = ctx->V;
    heldout = heldout ? 1 : 0;

    int rc = enqueue_prepare(ctx, heldout != 0);                      // :152-155
    if (rc != PYLDA_OK) return rc;
    // the launch plan of this E-step, then the hand-over buffers of the live-topic kernel (estep_compact.h), then - first
    // training E-step only - the postings, whose layout depends on whether the corpus hands documents over
    {
        const double span = tol * K;
        ctx->exact_stop = !(span >= 3.725290298461914e-09 /* 2^-28 */ && span < 1024.0);
    }
    if (c->plan_epoch != ctx->plan_epoch || c->plan_exact != ctx->exact_stop) build_plan(c);
    // alpha decides whether anything can be handed over at all (alpha_allows_live); when that changes, the postings change
    // their layout with it: lists of live topics <-> rows of t
    {
        const bool off = !alpha_allows_live(ctx, c->live_off_by_alpha);
        if (off != c->live_off_by_alpha) {
            c->live_off_by_alpha = off;
            if (c->have_postings && (c->live_stats || !off)) release_postings(c);
        }
    }
    if ((rc = prepare_compact(ctx, c)) != PYLDA_OK) return rc;
    if (!heldout && (rc = build_postings(c)) != PYLDA_OK) return rc;

    EstepParams p;
    p.K = K;
    p.V = V;
    p.ldk = ctx->ldk;
    p.expElog = ctx->d_expElog;
    p.expElog_elog = ctx->d_expElog_elog;
    p.shift = ctx->d_shift;
    p.topic_lse = ctx->d_topic_lse;
    p.alpha = ctx->d_alpha;
    p.alpha_sgn = ctx->d_alpha;       // (no document is handed over: every topic is its plain alpha)
    double asum = 0.0, alg = 0.0;
    for (double a : ctx->h_alpha) {
        asum += a;
        alg += std::lgamma(a);
    }
    p.alpha_term = std::lgamma(asum) - alg;                           // :195
    p.alpha_sum = asum;
    p.alpha_lgamma_sum = alg;
    p.doc_ptr = c->d_doc_ptr;
    p.term_id = c->d_term_id;
    p.term_ct = c->d_term_ct;
    p.max_iter = max_iter;
    p.tol = tol;
    p.heldout = heldout;
    p.want_doc_ll = (heldout || ctx->doc_values) ? 1 : 0;
    p.gamma = c->d_gamma;
    p.doc_ll = c->d_doc_ll;
    p.doc_words_ll = c->d_doc_wll;
    p.iters = c->d_iters;
    p.tfinal = c->d_tfinal;
    p.rfinal = c->d_rfinal;
    p.status = c->d_status;
    p.term_scratch = c->d_term_scratch;

    if (!c->d_term_scratch)
        for (const Launch& L : c->plan)
            if (L.variant == kGenericHuge) {
                rc = dev_alloc(ctx, &c->d_term_scratch, (size_t)c->nnz);
                if (rc != PYLDA_OK) return rc;
                p.term_scratch = c->d_term_scratch;
                break;
            }
    // hand-over to the live-topic kernel: its buffers, and the per-document work counters of this E-step
    p.handoff_on = 0;
    p.handoff_live = 0;
    p.tile_from_table = 0;
    compact_caps(ctx, p.handoff_caps);
    p.live_n = c->d_live_n;
    p.live_list = c->d_live_list;
    p.live_stats = (!heldout && c->live_stats) ? 1 : 0;
    p.live_tile = c->d_live_tile;
    p.tile_ptr = c->d_tile_ptr;
    p.handoff_it = c->d_handoff_it;
    p.col_iters = c->d_col_iters;
    p.clock_acc = ctx->profiling ? ctx->d_work + 4 : nullptr;
    p.alpha_min = ctx->compact_guard_fail ? 0.0 : *std::min_element(ctx->h_alpha.begin(), ctx->h_alpha.end());
    if (c->compact_ready) {
        // which topics may count as dead at all (kMortalT): alpha with a sign bit, for the kernels that hand documents over
        hipLaunchKernelGGL(alpha_mortality_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->d_alpha, K, ctx->d_alpha_sgn);
        p.alpha_sgn = ctx->d_alpha_sgn;
        HIP_TRY(ctx, hipMemsetAsync(c->d_handoff_it, 0xff, (size_t)c->D * sizeof(int32_t), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(c->d_col_iters, 0, (size_t)c->D * sizeof(int32_t), ctx->stream));
    }
    // (-1: the document's t is its dense row - until the live-topic kernel finishes it and leaves a list)
    if (p.live_stats) HIP_TRY(ctx, hipMemsetAsync(c->d_live_n, 0xff, (size_t)c->D * sizeof(int32_t), ctx->stream));
    auto open_bracket = [&](int slot, hipStream_t st) -> int {      // index into pending_events, or -1
        if (!ctx->profiling) return -1;
        pylda_ctx::Bracket br{take_event(ctx), take_event(ctx), slot};
        if (!br.a || !br.b || hipEventRecord(br.a, st) != hipSuccess) return -1;
        ctx->pending_events.push_back(br);
        return (int)ctx->pending_events.size() - 1;
    };
    auto close_bracket = [&](int at, hipStream_t st) {
        if (at >= 0) (void)hipEventRecord(ctx->pending_events[(size_t)at].b, st);
    };
    if (ctx->profiling && ctx->class_ms.size() != c->plan.size()) ctx->class_ms.assign(c->plan.size(), 0.0);
    const int doc_bracket = open_bracket(-1, ctx->stream);
    if (ctx->force_logspace) {
        // test hook: mark every document for the log-space kernel (a device fill: no host buffer, no wait)
        HIP_TRY(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->d_status), 1, (size_t)c->D, ctx->stream));
    } else {
        hipStream_t main_stream = ctx->stream;
        // a small corpus' slab classes go out as one dispatch on the main stream (no fork / join at all when that is
        // the whole plan); everything else: one launch per class, fanned out over the auxiliary streams
        const int uber_from = slab_uber_from(ctx, c);
        const size_t separate = uber_from >= 0 ? (size_t)uber_from : c->plan.size();
        const bool fan_out = separate > (uber_from >= 0 ? 0u : 1u);
        const int used = fan_out ? (int)std::min<size_t>(pylda_ctx::kAux, separate) : 0;
        if (fan_out) {
            HIP_TRY(ctx, hipEventRecord(ctx->fork_event, main_stream));
            for (int i = 0; i < used; ++i) HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux_stream[i], ctx->fork_event, 0));
        }
        // the auxiliary streams rejoin the main stream on every path out of here, failures included
        auto join = [&]() {
            ctx->stream = main_stream;
            for (int i = 0; i < used; ++i)
                if (hipEventRecord(ctx->join_event[i], ctx->aux_stream[i]) == hipSuccess)
                    (void)hipStreamWaitEvent(main_stream, ctx->join_event[i], 0);
        };
        // launch order over the classes (the plan lists them longest documents first): option launch_order 1 sends the
        // classes with the FEWEST documents first, so that the kernels that end the E-step are the large ones
        std::vector<size_t> launch_seq(separate);
        std::iota(launch_seq.begin(), launch_seq.end(), (size_t)0);
        if (ctx->launch_order == 1)
            std::stable_sort(launch_seq.begin(), launch_seq.end(), [&](size_t a, size_t b) { return c->plan[a].count < c->plan[b].count; });
        size_t launch_index = 0;
        for (const size_t plan_index : launch_seq) {
            const Launch& L = c->plan[plan_index];
            const int slot = (int)plan_index;
            if (fan_out) ctx->stream = ctx->aux_stream[launch_index % pylda_ctx::kAux];
            ++launch_index;
            p.order = c->d_order + L.first;
            p.n_cap = L.n_cap;
            p.tile_stride = L.tile_stride;
            p.handoff_on = c->compact_ready && compact_handoff_for(ctx, L) > 0 ? 1 : 0;
            // (a quad class holds one lane shape - its shortest documents at K <= 128 two, of equal capacity)
            p.handoff_live = p.handoff_on && L.variant == kQuad ? p.handoff_caps[std::min(8, std::max(1, (L.n_cap + kWave - 1) / kWave))] : 0;
            const int class_bracket = open_bracket(slot, ctx->stream);
            // PYLDA_DEBUG_SYNC=1: fault localisation - a line before every launch, a wait and the stream's status behind it
            static const bool debug_sync = getenv("PYLDA_DEBUG_SYNC") != nullptr;
            if (debug_sync)
                fprintf(stderr, "[pylda debug] launching class %d variant %d geometry %d documents %lld n_cap %d handoff %d\n", slot, L.variant, L.rn,
                        (long long)L.count, L.n_cap, p.handoff_on);
            switch (L.variant) {
            case kSlab: rc = launch_slab_any(ctx, p, L); break;
            case kQuilt: rc = launch_quilt_any(ctx, p, L); break;
            case kQgroup: rc = launch_qgroup(ctx, p, L); break;
            case kQuad: rc = launch_quad_any(ctx, p, L); break;
            case kQfuse: rc = launch_qfuse(ctx, p, L); break;
            case kQfusek: rc = launch_qfusek(ctx, p, L); break;
            default: rc = launch_generic_any(ctx, p, L); break;      // the generic family (tile in LDS / re-read from the table)
            }
            if (debug_sync) {
                const hipError_t e = hipStreamSynchronize(ctx->stream);
                fprintf(stderr, "[pylda debug] class %d variant %d geometry %d documents %lld handoff %d: %s\n", slot, L.variant, L.rn,
                        (long long)L.count, p.handoff_on, hipGetErrorString(e));
            }
            // ... and behind it, on the same stream, the live-topic kernel for the documents the class handed over
            // (option compact_phase 0; the default runs them as a phase of their own behind all dense kernels, below)
            if (p.handoff_on && !ctx->compact_phase)
                for (const pylda_corpus::CompactRange& r : c->compact_ranges)
                    if (rc == PYLDA_OK && r.plan_index == slot) rc = launch_compact(ctx, c, p, r.slots, r.from_table, r.first, r.count);
            if (debug_sync && p.handoff_on && !ctx->compact_phase) {
                const hipError_t e = hipStreamSynchronize(ctx->stream);
                fprintf(stderr, "[pylda debug] class %d live-topic kernel: %s\n", slot, hipGetErrorString(e));
            }
            close_bracket(class_bracket, ctx->stream);
            if (rc != PYLDA_OK) {
                join();
                return rc;
            }
        }
        ctx->stream = main_stream;
        p.handoff_on = 0;
        p.handoff_live = 0;
        if (uber_from >= 0) {
            const Launch& L = c->plan[(size_t)uber_from];
            p.order = c->d_order + L.first;
            p.n_cap = L.n_cap;
            p.tile_stride = L.tile_stride;
            const int class_bracket = open_bracket(uber_from, main_stream);      // (the group's time is booked on its first class)
            rc = launch_slab_uber_any(ctx, p, c, uber_from);
            close_bracket(class_bracket, main_stream);
            if (rc != PYLDA_OK) {
                join();
                return rc;
            }
        }
        join();
        // The live-topic kernels as a phase of their own.  A dense quad workgroup needs a whole CU (at K = 256: its eight
        // wavefronts hold the CU's register file), a live-topic wavefront an eighth of one: side by side on the chip, a
        // CU that holds even one live-topic wavefront cannot take a dense document, and the dense kernels - 90 % of
        // the document time - ran on what was left (measured on 200k cfg 4 documents: 38.7 ms mixed).  Behind the join
        // the two never meet.  One launch per lane shape (term slots per lane): the classes of a shape are contiguous in
        // the schedule.
        if (c->compact_ready && ctx->compact_phase) {
            // one launch per lane shape (term slots per lane): the ranges of a shape are adjacent across the classes
            for (size_t a = 0; a < c->compact_ranges.size();) {
                const pylda_corpus::CompactRange& r = c->compact_ranges[a];
                int64_t count = r.count;
                size_t b = a + 1;
                while (b < c->compact_ranges.size() && c->compact_ranges[b].slots == r.slots && c->compact_ranges[b].from_table == r.from_table &&
                       c->compact_ranges[b].first == r.first + count) {
                    count += c->compact_ranges[b].count;
                    ++b;
                }
                if ((rc = launch_compact(ctx, c, p, r.slots, r.from_table, r.first, count)) != PYLDA_OK) return rc;
                a = b;
            }
        }
    }
    // the document terms the register kernels left out on the training fast path (status 3; doc_terms.h): one wavefront
    // per document, fp64-VALU bound.  Beside the dispatch-paced statistics gather (L2-bound) it runs on an auxiliary
    // stream, CONCURRENTLY with it - neither needs the other's output; the persistent sweep needs every CU to itself
    // (one workgroup per CU, all resident), so there the pass stays in front of it.
    bool leaves_terms = false;          // (slab and generic kernels always finish their documents themselves)
    for (const Launch& L : c->plan)
        leaves_terms = leaves_terms || L.variant == kQuad || L.variant == kQuilt || L.variant == kQgroup || L.variant == kQfuse || L.variant == kQfusek;
    const bool terms_pass = !heldout && !p.want_doc_ll && c->D > 0 && leaves_terms && !ctx->force_logspace;
    const bool terms_beside_gather = terms_pass && c->have_postings && !c->sweep && ctx->terms_overlap;
    bool terms_forked = false;          // the pass runs on aux_stream[0]: the main stream joins it inside the statistics bracket
    if (terms_pass) {
        p.order = nullptr;
        hipStream_t st = ctx->stream;
        // (a failure to fork keeps the pass on the main stream: no exit of this function leaves work on the auxiliary
        //  stream that the main stream does not wait for)
        if (terms_beside_gather && hipEventRecord(ctx->fork_event, ctx->stream) == hipSuccess &&
            hipStreamWaitEvent(ctx->aux_stream[0], ctx->fork_event, 0) == hipSuccess)
            st = ctx->aux_stream[0];
        hipLaunchKernelGGL(doc_terms_kernel, dim3((unsigned)((c->D + 3) / 4)), dim3(256), 0, st, p, c->D);
        terms_forked = st != ctx->stream;
        if (terms_forked && hipEventRecord(ctx->join_event[0], st) != hipSuccess) {
            (void)hipStreamSynchronize(st);         // cannot order by event: order by the host, once
            terms_forked = false;
        }
    }
    close_bracket(doc_bracket, ctx->stream);
    if (ctx->profiling) ctx->estep_calls += 1;
    if (ctx->profiling && c->D > 0)       // inner iterations actually executed, for the fp64 roofline and doc-iterations/s
        hipLaunchKernelGGL(work_count_kernel, dim3(1), dim3(1024), 0, ctx->stream, c->d_iters, c->d_doc_ptr, c->D, ctx->d_work,
                           c->compact_ready ? c->d_handoff_it : nullptr, c->d_col_iters, K);

    // sufficient statistics (:207): gather pass over the postings, no atomics
    if (!heldout) {
        if (ctx->force_logspace) {
            HIP_TRY(ctx, hipMemsetAsync(c->d_rfinal, 0, (size_t)c->nnz * sizeof(double), ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(c->d_tfinal, 0, (size_t)c->D * ctx->ldk * sizeof(double), ctx->stream));
        }
        const int ss_bracket = open_bracket(-2, ctx->stream);
        rc = enqueue_sstats_gather(ctx, c);
        // (inside the bracket: kernel_time()'s `statistics` figure is the wall time of the PAIR gather + document terms
        //  whenever the pass is forked - bench.py's roofline adds documents + statistics, so nothing is lost or counted twice)
        if (terms_forked) (void)hipStreamWaitEvent(ctx->stream, ctx->join_event[0], 0);
        close_bracket(ss_bracket, ctx->stream);
        if (rc != PYLDA_OK) return rc;
    }
    // safety net: documents the linear-space kernels flagged are redone in log space (the kernel finds them itself)
    if (c->D > 0) {
        p.order = nullptr;
        // documents per workgroup and step: as many as keep 4 workgroups per CU busy, 256 at most (a small corpus with many
        // flagged documents - or the force_logspace hook - used to sit on D / 256 workgroups: 8 of 256 CUs at 2000 documents)
        const int64_t slots = 4 * (int64_t)ctx->num_cu;
        const int chunk = (int)std::min<int64_t>(256, std::max<int64_t>(1, (c->D + slots - 1) / slots));
        const unsigned grid = (unsigned)std::min<int64_t>((c->D + chunk - 1) / chunk, slots);
        const size_t list_offset = (logspace_lds_bytes(K) + 15) & ~(size_t)15, lds = list_offset + 257 * sizeof(int32_t);
        if (lds > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(estep_logspace_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(estep_logspace_kernel, dim3(grid), dim3(256), lds, ctx->stream, p, ctx->d_elog, ctx->d_sstats,
                           c->d_status, c->D, list_offset, chunk);
    }
    // the corpus-level sums and the number of documents redone, into the four scalars pylda_estep_results reads back
    hipLaunchKernelGGL(vector_sum3_kernel, dim3(4), dim3(1024), 0, ctx->stream, SumJob{c->d_doc_ll, c->D, c->d_scalars},
                       SumJob{c->d_doc_wll, c->D, c->d_scalars + 1},
                       SumJob{c->d_entropy_partial, heldout ? 0 : c->ent_blocks, heldout ? nullptr : c->d_scalars + 2},
                       c->d_status, c->D, c->d_flag_count);
    HIP_TRY(ctx, hipGetLastError());
    c->estep_done = true;
    c->last_heldout = heldout;
    c->last_doc_values = p.want_doc_ll != 0;
    if (!heldout) ctx->have_sstats = true;
    return PYLDA_OK;
}

int pylda_estep_results(pylda_ctx* ctx, pylda_corpus* c, double* document_log_likelihood,
                        double* words_log_likelihood, int64_t* logspace_documents)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!c || c->ctx != ctx) return fail(ctx, PYLDA_ERR_INVALID, "estep_results: bad corpus");
    if (!c->estep_done) return fail(ctx, PYLDA_ERR_STATE, "estep_results: no E-step has run on this corpus");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // through the context's page-locked staging area (its last four doubles): a copy into pageable memory is staged and
    // waited for by the runtime, twice (associated-press: 35 % of the E-step's wall time was this read-back)
    double* sc = ctx->h_pin + (size_t)5 * ctx->K + 4;
    int32_t* nflag_pin = reinterpret_cast<int32_t*>(sc + 3);
    HIP_TRY(ctx, hipMemcpyAsync(sc, c->d_scalars, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int32_t nflag = *nflag_pin;
    // training fast path: the log B entropy term comes once per corpus from the statistics
    if (!c->last_doc_values) sc[0] -= sc[2];
    if (document_log_likelihood) *document_log_likelihood = sc[0];
    if (words_log_likelihood) *words_log_likelihood = sc[1];
    if (logspace_documents) *logspace_documents = nflag;
    return PYLDA_OK;
}

int pylda_get_gamma(pylda_ctx* ctx, pylda_corpus* c, double* gamma_dk)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!c || c->ctx != ctx || !gamma_dk) return fail(ctx, PYLDA_ERR_INVALID, "get_gamma: bad argument");
    if (!c->estep_done) return fail(ctx, PYLDA_ERR_STATE, "get_gamma: no E-step has run on this corpus");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(gamma_dk, c->d_gamma, (size_t)c->D * ctx->K * sizeof(double),
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PYLDA_OK;
}

int pylda_get_doc_values(pylda_ctx* ctx, pylda_corpus* c, double* doc_ll, double* doc_words_ll,
                         int32_t* iters)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!c || c->ctx != ctx) return fail(ctx, PYLDA_ERR_INVALID, "get_doc_values: bad corpus");
    if (!c->estep_done) return fail(ctx, PYLDA_ERR_STATE, "get_doc_values: no E-step has run on this corpus");
    if (doc_ll && !c->last_doc_values)
        return fail(ctx, PYLDA_ERR_STATE, "get_doc_values: the last E-step ran with option doc_values=0 (corpus-level likelihood only)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (doc_ll)
        HIP_TRY(ctx, hipMemcpyAsync(doc_ll, c->d_doc_ll, (size_t)c->D * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (doc_words_ll)
        HIP_TRY(ctx, hipMemcpyAsync(doc_words_ll, c->d_doc_wll, (size_t)c->D * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (iters)
        HIP_TRY(ctx, hipMemcpyAsync(iters, c->d_iters, (size_t)c->D * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PYLDA_OK;
}

int pylda_estep_host(pylda_ctx* ctx, pylda_corpus* c, const double* alpha_k, const double* eta_kv,
                     int max_iter, double tol, int heldout, double* gamma_dk, double* sstats_kv,
                     double* doc_ll, double* doc_words_ll, int32_t* iters, double* scalars_out)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    int rc;
    if ((rc = pylda_set_alpha(ctx, alpha_k)) != PYLDA_OK) return rc;
    if ((rc = pylda_set_eta(ctx, eta_kv)) != PYLDA_OK) return rc;
    if ((rc = pylda_estep(ctx, c, max_iter, tol, heldout)) != PYLDA_OK) return rc;
    double sc[2];
    if ((rc = pylda_estep_results(ctx, c, &sc[0], &sc[1], nullptr)) != PYLDA_OK) return rc;
    if (scalars_out) {
        scalars_out[0] = sc[0];
        scalars_out[1] = sc[1];
    }
    if (gamma_dk && (rc = pylda_get_gamma(ctx, c, gamma_dk)) != PYLDA_OK) return rc;
    if (sstats_kv && !heldout && (rc = pylda_get_sstats(ctx, sstats_kv)) != PYLDA_OK) return rc;
    if (doc_ll || doc_words_ll || iters)
        if ((rc = pylda_get_doc_values(ctx, c, doc_ll, doc_words_ll, iters)) != PYLDA_OK) return rc;
    return PYLDA_OK;
}

}  // extern "C"
