// Internal header of libpylda_hip.so: the context / corpus objects behind the opaque handles of include/pylda_hip.h
// and the host-side functions its translation units share.  Nothing here is part of the ABI.
//
//   context.hip        handles, options, model tables in and out, host memory, marks, RCCL glue, test hooks
//   host_plan.cpp      the planner: launch classes, segment cut, rounds, sweep dealing - pure host functions, no HIP
//   plan.hip           the planner's configuration from the context; pylda_corpus_plan / pylda_corpus_layout
//   launch_small.hip   document kernels, generic / slab / quilt families
//   launch_quad.hip    ... the register + LDS tile kernel (strides 128 / 256), handing documents to the live-topic kernel
//   launch_quad_dense.hip  ... the same without the hand-over (launch_quad.hip compiled again with the flag off)
//   launch_compact.hip ... the live-topic kernel behind them, its buffers, and whether alpha still lets topics die
//   launch_stream.hip  ... the fused streaming families (qfuse, qfusek, qgroup)
//   sstats_gather.hip  postings, segments and the statistics pass (dispatch-paced gather, persistent sweep)
//   estep_api.hip      corpus upload, pylda_estep and its read-backs
//   mstep_api.hip      device M-step, pack, alpha update, the outer iteration's one read-back
#pragma once
#include "../../include/pylda_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "comm.h"
#include "estep_common.h"
#include "estep_limits.h"
#include "host_plan.h"
#include "postings.h"

using namespace pylda;

namespace pylda_host __attribute__((visibility("hidden"))) {
using namespace pylda_plan;     // Variant, Launch, the planner (host_plan.h: HIP-free, sanitizer-tested)
}  // namespace pylda_host

using namespace pylda_host;

struct pylda_ctx {
    int device = 0;
    int K = 0, V = 0;
    int ldk = 0;                    // row stride of the word-major tables
    hipStream_t own_stream = nullptr;
    // A corpus whose documents fall into several launch classes (different words-per-lane
    // instantiations) has independent launches: they are fanned out over these streams so a
    // small corpus pays one kernel latency (50 serial inner iterations), not one per class.
    static constexpr int kAux = 4;
    hipStream_t aux_stream[kAux] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork_event = nullptr;
    hipEvent_t join_event[kAux] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t stream = nullptr;
    size_t lds_limit = 64 * 1024;
    int num_cu = 256;

    double* d_eta = nullptr;        // K x V (numpy layout)
    double* d_elog = nullptr;       // V x ldk shifted E_log_eta
    double* d_expElog = nullptr;    // V x ldk
    double* d_expElog_elog = nullptr; // V x ldk
    double* d_shift = nullptr;      // V
    double* d_psi_rowsum = nullptr; // K
    double* d_topic_lse = nullptr;  // K
    double* d_alpha = nullptr;      // K
    double* d_alpha_sgn = nullptr;  // K: alpha, sign bit set where the topic never counts as dead (alpha_mortality_kernel, per E-step)
    double* d_sstats = nullptr;     // V x ldk
    double* d_kv_scratch = nullptr; // K x V (export transposes)
    double* d_beta = nullptr;       // V
    double* d_small = nullptr;      // scalars + K-vectors scratch
    double* d_partial = nullptr;    // alpha-ss partials
    std::vector<double> h_beta;     // the beta last handed to pylda_mstep (its lgamma sums are cached)
    double beta_sum = 0.0, beta_lgamma_sum = 0.0;

    std::vector<double> h_alpha;
    // pinned host staging (one allocation, 5K + 8 doubles): two alpha slots (K each), the outer-iteration read-back (3K + 4),
    // the E-step's scalars (4)
    double* h_pin = nullptr;
    hipEvent_t alpha_event[2] = {nullptr, nullptr};
    bool alpha_event_used[2] = {false, false};
    int alpha_slot = 0;
    double* d_outer = nullptr;      // [doc ll, #documents, log-space documents, 0, alpha ss (K) | per-topic ll (K)]
    bool outer_ready = false;
    bool newton_pending = false;    // pylda_mstep_enqueue asked for the alpha update: pylda_outer_fetch runs it
    NewtonParams newton;
    double* d_newton_work = nullptr;   // 4 K
    double* d_eta_ckpt = nullptr;   // pylda_model_checkpoint
    double* d_work = nullptr;       // profiling, accumulated over E-steps: [sum_d I_d, sum_d I_d N_d, tile entries executed, documents handed
                                    // over, shader-clock ticks, constant-rate ticks of the same spans]
    double work_cache[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // the last read of d_work (pylda_work_counters)
    hipEvent_t mark_event[4] = {nullptr, nullptr, nullptr, nullptr};
    void* comm = nullptr;           // RCCL communicator of pylda_comm_init (multi-GPU through the C ABI)
    int comm_world = 1;
    double* d_comm_small = nullptr; // staging buffer of pylda_allreduce_doubles
    size_t comm_small_cap = 0;
    bool have_eta = false, have_alpha = false, have_sstats = false;
    int force_logspace = 0;
    int force_variant = -1;
    int quilt12 = 0;
    int gather_rows = 2;            // 0: 64-topic chunks; 1: whole rows (ldk 64 / 128 / 256); 2: + postings in bulk (ldk 128 / 256)
    int gather_blocks = -1;         // document blocks of the gather: -1 automatic, 0 / 1 off, n forced (multiple of 8)
    int sweep_xcd = 1;              // the sweep's rendezvous per XCD (32 workgroups) instead of chip-wide (256)
    int sweep_sub = 4;              // sub-blocks a document block is walked in by the sweep (L2 working set 1 / sub of a block)
    int sweep_spin = 4000;          // polls of a rendezvous of the sweep before a workgroup goes on alone
    int gather_sweep = 1;           // the persistent sweep (sstats_sweep.h) at stride 128 / 256: 0 never, 1 when the partial rows of the
                                    // dispatch-paced gather would exceed their budget (rounds), 2 whenever the gather is blocked
    int gather_round_mb = 0;        // budget of the gather's partial rows per round, MiB (0: 4 GiB for the sweep-or-gather decision;
                                    // the rounds themselves are also capped by a quarter of the free device memory)
    int launch_order = 1;           // 0: launch classes in plan order (longest documents first), 1: fewest documents first (cfg 3: -0.5 %)
    int terms_overlap = 1;          // doc_terms_kernel on an auxiliary stream beside the dispatch-paced statistics gather
    int slab_uber = 1;              // small corpora: all slab launch classes in one dispatch
    int wide_postings = 0;          // test hook: 64-bit CSR positions in the postings whatever nnz (automatic from 2^31 pairs)
    int lds_pad = 0;                // A/B: extra dynamic LDS per quad workgroup (forces one workgroup per CU)
    int quad = 1;                   // register + LDS tile kernel (estep_quad.h) for table strides 128 / 256, N <= 208
    int quad_stream = 1;            // ... with streamed word slots for documents of 225-256 terms
    int quilt_odd = 1;              // words-per-lane 6 / 7 instantiations (less padding for 129..224-term documents)
    int doc_values = 1;             // 1: per-document log-likelihoods complete (see EstepParams::want_doc_ll)
    int compact = 1;                // the dense quad kernel hands a document to the live-topic kernel (estep_compact.h) once few topics move
    int gather_live = 1;            // the statistics pass reads the documents' lists of live topics (sstats_live.h) where the corpus hands documents over
    int compact_phase = 1;          // the live-topic kernels run behind ALL dense kernels of the E-step (0: behind their class, on its stream)
    int compact_stream = 1;         // ... and so do the fused streaming kernels (384 <= table stride <= 1024), without a tile
    int compact_pair = -1;          // hand over at twice one wavefront's columns (two-wavefront body): -1 from table stride 256 on, 0 never, 1 always
    int compact_cap = 0;            // test hook: hand over at this many live topics at most (0: what the class' register tile holds)
    int compact_guard_fail = 0;     // test hook: the live-topic kernel's exactness guard fails for every document
    int plan_epoch = 0;
    bool exact_stop = false;        // this E-step's threshold is outside the fixed-point stop test's range

    // profiling (pylda_set_profiling): HIP events on the launch streams
    struct Bracket { hipEvent_t a, b; int slot; };   // slot -1: document kernels, -2: statistics pass, >= 0: launch class
    bool profiling = false;
    std::vector<Bracket> pending_events;
    std::vector<hipEvent_t> event_pool;
    double doc_kernel_ms = 0.0, sstats_kernel_ms = 0.0;
    std::vector<double> class_ms;   // per launch class of the last profiled corpus
    int64_t estep_calls = 0;

    std::string err;
};

struct pylda_corpus {
    pylda_ctx* ctx = nullptr;
    int64_t D = 0, nnz = 0, tokens = 0;
    int32_t max_terms = 0;
    int64_t* d_doc_ptr = nullptr;
    int32_t* d_term_id = nullptr;
    int32_t* d_term_ct = nullptr;
    int32_t* d_order = nullptr;
    double* d_gamma = nullptr;
    double* d_doc_ll = nullptr;
    double* d_doc_wll = nullptr;
    int32_t* d_iters = nullptr;
    int32_t* d_status = nullptr;
    int32_t* d_flag_count = nullptr;   // documents the safety net redid in the last E-step over THIS corpus (in d_scalars[3])
    double* d_scalars = nullptr;   // [0] doc ll, [1] words ll, [2] corpus entropy term (fast path)
    double* d_entropy_partial = nullptr;
    bool last_doc_values = true;
    double* d_tfinal = nullptr;    // D x ldk
    double* d_rfinal = nullptr;    // nnz
    double* d_term_scratch = nullptr;   // nnz, only when a launch class needs it (kGenericHuge)
    // postings (CSC) of the corpus for the sufficient-statistics gather pass
    bool have_postings = false;
    int32_t* d_post_doc = nullptr; // nnz
    void* d_post_pos = nullptr;    // nnz: position in CSR order (int32, or int64 when wide_pos)
    bool wide_pos = false;         // nnz >= 2^31 (or option wide_postings): 64-bit CSR positions in the postings
    int64_t* d_seg_begin = nullptr;
    int64_t* d_seg_end = nullptr;
    int32_t* d_exec_order = nullptr;   // document-blocked gather: segment of every (workgroup, wavefront) slot, or -1
    int64_t exec_slots = 0;
    // The gather runs in ROUNDS over contiguous term ranges that share one set of partial rows (a (term, block)
    // pair costs a row: 45 GB at cfg 4 in one go - and a second for the allocation alone): gather round r, finalize
    // its terms, reuse the rows.  One round unless the rows would exceed the budget.
    using Round = pylda_plan::Round;
    std::vector<Round> rounds;
    int64_t partial_rows = 0, ent_blocks = 0;
    // ... or the persistent sweep (sstats_sweep.h): no partial rows at all
    bool sweep = false;
    int sweep_passes = 0, sweep_terms = 0, sweep_wpb = 0;   // passes over the document blocks, terms per wavefront, wavefronts per workgroup
    int32_t* d_seg_block = nullptr;             // document block of every segment
    int32_t* d_term_of = nullptr;               // [passes][wavefronts][terms per wavefront]
    unsigned* d_rendezvous = nullptr;
    int gather_blocks = 1;
    int gather_rows = 2;            // the context's option at the time the postings were built (it selects the gather kernel)
    int64_t* d_word_seg_ptr = nullptr;  // V+1
    double* d_partial = nullptr;   // nseg x ldk
    int64_t nseg = 0;
    std::vector<int32_t> h_terms_sorted;  // distinct-term counts in schedule order
    std::vector<int32_t> h_order;         // the schedule: document of every slot
    // hand-over to the live-topic kernel (launch_compact.hip): per document the live topics and their tile columns
    int32_t* d_live_n = nullptr;          // D
    char* d_live_list = nullptr;          // D x kLiveListBytes: a document's live topics and their t
    bool live_stats = false;              // the statistics pass of this corpus reads those lists (decided with the postings)
    int64_t* d_tile_ptr = nullptr;        // D
    double* d_live_tile = nullptr;        // sum over the quad classes' documents of N_d x (live topics the class hands over at)
    int32_t* d_handoff_it = nullptr;      // D: inner iterations the dense kernel ran before the hand-over, or -1
    int32_t* d_col_iters = nullptr;       // D: tile columns x iterations the live-topic kernel executed
    bool compact_ready = false;
    bool compact_failed = false;          // the tile buffer did not fit: dense kernels only, for good
    bool live_off_by_alpha = false;       // alpha has grown: too many topics never count as dead for any document to be handed over (alpha_allows_live)
    int compact_plan_epoch = -1, compact_cap_used = -1, compact_stream_used = -1, compact_pair_used = -2;
    // schedule ranges of one lane shape (term slots per lane) inside a launch class that hands documents over
    struct CompactRange { int plan_index, slots; bool from_table; int64_t first, count; };
    std::vector<CompactRange> compact_ranges;
    std::vector<Launch> plan;
    int plan_epoch = 0;
    bool plan_exact = false;       // the plan avoids the kernels with the fixed-point stop test
    bool estep_done = false;
    int last_heldout = 0;
};

namespace pylda_host __attribute__((visibility("hidden"))) {

extern std::string g_create_error;
int fail(pylda_ctx* ctx, int code, const char* fmt, ...);

#define HIP_TRY(ctx, expr)                                                              \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail((ctx), e_ == hipErrorOutOfMemory ? PYLDA_ERR_OOM : PYLDA_ERR_HIP, \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,  \
                        __LINE__);                                                      \
    } while (0)

template <typename T>
int dev_alloc(pylda_ctx* ctx, T** p, size_t n)
{
    *p = nullptr;
    if (n == 0) n = 1;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
    return PYLDA_OK;
}

template <typename T>
void dev_free(T*& p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

// PYLDA_TIMING=1: wall time of the one-off phases (corpus upload, postings, segment cut) on stderr
struct PhaseTimer {
    bool on = getenv("PYLDA_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char* what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[pylda timing] %-34s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t0).count());
        t0 = now;
    }
};

// ---- plan.hip ----
PlanConfig plan_config(const pylda_ctx* ctx);
void build_plan(pylda_corpus* c);
int slab_uber_from(const pylda_ctx* ctx, const pylda_corpus* c);    // first class of the one-dispatch slab group, or -1

// ---- launch_*.hip: one launch class of the plan on ctx->stream ----
int launch_generic_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L);
int launch_slab_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L);
int launch_slab_uber_any(pylda_ctx* ctx, const EstepParams& p, const pylda_corpus* c, int from);
int launch_quilt_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L);
int launch_quad_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L);         // (hands on to the next one where the class hands nothing over)
int launch_quad_dense_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L);   // launch_quad_dense.hip: the kernels without the hand-over
int launch_qfuse(pylda_ctx* ctx, const EstepParams& p, const Launch& L);
int launch_qfusek(pylda_ctx* ctx, const EstepParams& p, const Launch& L);
int launch_qgroup(pylda_ctx* ctx, const EstepParams& p, const Launch& L);

// ---- launch_compact.hip: the live-topic kernel behind a quad launch class ----
int compact_handoff_for(const pylda_ctx* ctx, const Launch& L);      // 0: the class keeps its documents, 1: hands over with tile columns, 2: without
void compact_caps(const pylda_ctx* ctx, int (&caps)[9]);             // live topics at which a document of s term slots per lane is handed over
int prepare_compact(pylda_ctx* ctx, pylda_corpus* c);                // buffers and ranges of the hand-over (sets c->compact_ready)
int immortal_topics(const pylda_ctx* ctx);                           // topics whose alpha keeps them from ever counting as dead (kMortalT), by the host's alpha
bool alpha_allows_live(const pylda_ctx* ctx, bool was_off);          // fewer of them than the widest tile has columns (with hysteresis)
void release_postings(pylda_corpus* c);                              // sstats_gather.hip: the next training E-step builds them again
int launch_compact(pylda_ctx* ctx, const pylda_corpus* c, EstepParams p, int slots, bool from_table, int64_t first, int64_t count);

// ---- sstats_gather.hip ----
int build_postings(pylda_corpus* c);
int enqueue_sstats_gather(pylda_ctx* ctx, pylda_corpus* c);

// ---- context.hip: profiling events ----
hipEvent_t take_event(pylda_ctx* ctx);
void drain_events(pylda_ctx* ctx);

}  // namespace pylda_host
