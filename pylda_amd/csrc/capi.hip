// libpylda_hip.so - host side of the C ABI declared in include/pylda_hip.h.
//
// Owns the device-resident model tables and corpora, builds the launch
// schedule and enqueues the gfx950 kernels.  No CPU fallback exists: every
// compute entry point requires a HIP device.
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
#include "estep_generic.h"
#include "estep_logspace.h"
#include "estep_slab.h"
#include "estep_quilt.h"
#include "estep_quad.h"
#include "doc_terms.h"
#include "estep_qfuse.h"
#include "estep_qfusek.h"
#include "estep_qstream.h"
#include "estep_qhybrid.h"
#include "estep_qwide.h"
#include "mstep_kernels.h"
#include "postings.h"
#include "prepare_kernels.h"
#include "sstats_kernels.h"
#include "sstats_sweep.h"

using namespace pylda;

namespace {

std::string g_create_error;

enum Variant : int {
    kGeneric64 = 0,    // 1 wavefront / document, tile in LDS
    kGeneric256 = 1,   // 4 wavefronts / document, tile in LDS
    kGeneric512 = 2,   // 8 wavefronts / document, tile in LDS (up to the whole 160 KiB)
    kGenericGlobal = 3, // tile larger than LDS: rows re-read from the table
    kSlab = 4,          // tile in registers, word-major lanes (estep_slab.h)
    kRetired5 = 5,      // (the topic-major column kernel of round 1: measured 2x slower than the quilt layout, removed)
    kQuilt = 6,         // tile in registers, 4 x 16 word-group x topic lanes (estep_quilt.h)
    kQstream = 7,       // tile streamed from L2 twice per iteration, quilt lanes (estep_qstream.h)
    kQhybrid = 8,       // tile split over registers / LDS / streamed remainder (estep_qhybrid.h)
    kQwide = 9,         // the same three tiers on a 2 x 32 lane grid with prefetched tail rows (estep_qwide.h)
    kQuad = 10,         // 16 word groups / document, tile in registers + LDS rows (estep_quad.h)
    kQfuse = 11,        // table stride 512: rows streamed ONCE per iteration, normaliser and topic sums fused (estep_qfuse.h)
    kGenericHuge = 12,  // a document too long even for its per-term scalars in LDS: those in global memory too (estep_generic.h MODE 2)
    kQfusek = 13,       // table stride 640 .. 1024: every row streamed once per iteration, fused (estep_qfusek.h)
    kVariantLast = kQfusek
};

struct Launch {
    int variant;
    int64_t first;   // offset into the sorted order
    int64_t count;   // documents (= workgroups)
    int n_cap;       // largest distinct-term count in the launch
    int tile_stride;
    size_t lds_bytes;
    int rn;          // slab kernels: words per lane
    int rk;          // slab kernels: topics per wavefront
};

}  // namespace

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
    double* d_sstats = nullptr;     // V x ldk
    double* d_kv_scratch = nullptr; // K x V (export transposes)
    double* d_beta = nullptr;       // V
    double* d_small = nullptr;      // scalars + K-vectors scratch
    double* d_partial = nullptr;    // alpha-ss partials
    std::vector<double> h_beta;     // the beta last handed to pylda_mstep (its lgamma sums are cached)
    double beta_sum = 0.0, beta_lgamma_sum = 0.0;

    std::vector<double> h_alpha;
    // pinned host staging (one allocation): two alpha slots (K each) + the outer-iteration read-back (2K + 8)
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
    double* d_work = nullptr;       // profiling: [sum_d I_d, sum_d I_d N_d] accumulated over E-steps
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
    int sweep_spin = 4000;          // polls of a rendezvous of the sweep before a workgroup goes on alone
    int gather_sweep = 1;           // the persistent sweep (sstats_sweep.h) at stride 128 / 256: 0 never, 1 when the partial rows of the
                                    // dispatch-paced gather would exceed their budget (rounds), 2 whenever the gather is blocked
    int gather_round_mb = 0;        // budget of the gather's partial rows per round, MiB (0: 4 GiB)
    int slab_uber = 1;              // small corpora: all slab launch classes in one dispatch
    int wide_postings = 0;          // test hook: 64-bit CSR positions in the postings whatever nnz (automatic from 2^31 pairs)
    int lds_pad = 0;                // A/B: extra dynamic LDS per quad workgroup (forces one workgroup per CU)
    int quad = 1;                   // register + LDS tile kernel (estep_quad.h) for table strides 128 / 256, N <= 208
    int quilt_odd = 1;              // words-per-lane 6 / 7 instantiations (less padding for 129..224-term documents)
    int doc_values = 1;             // 1: per-document log-likelihoods complete (see EstepParams::want_doc_ll)
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
    int32_t* d_flag_list = nullptr;
    int32_t* d_flag_count = nullptr;   // documents the safety net redid in the last E-step over THIS corpus
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
    struct Round { int64_t seg_lo, seg_hi; int w_first, n_words; int64_t slot_lo, slot_count; int64_t ent_first, ent_blocks; };
    std::vector<Round> rounds;
    int64_t partial_rows = 0, ent_blocks = 0;
    // ... or the persistent sweep (sstats_sweep.h): no partial rows at all
    bool sweep = false;
    int sweep_passes = 0, sweep_terms = 0, sweep_wpb = 0;   // passes over the document blocks, terms per wavefront, wavefronts per workgroup
    int32_t* d_seg_block = nullptr;             // document block of every segment
    int32_t* d_term_of = nullptr;               // [passes][wavefronts][terms per wavefront]
    unsigned* d_rendezvous = nullptr;
    int gather_blocks = 1;
    int64_t* d_word_seg_ptr = nullptr;  // V+1
    double* d_partial = nullptr;   // nseg x ldk
    int64_t nseg = 0;
    std::vector<int32_t> h_terms_sorted;  // distinct-term counts in schedule order
    std::vector<Launch> plan;
    int plan_epoch = 0;
    bool plan_exact = false;       // the plan avoids the kernels with the fixed-point stop test
    bool estep_done = false;
    int last_heldout = 0;
};

namespace {

int fail(pylda_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

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

int tile_stride_for(int K) { return K | 1; }   // odd => conflict-free ds_read_b64 along words

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
// Register slots and LDS slots per word group, N <= 16 * (RWL + TWL); code TL * 10000 + RWL * 100 + TWL,
// or 0.  TWL <= 3: two workgroups inside a CU's 160 KiB of LDS at stride 128, one at stride 256.
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
    return 0;
}

int quilt_rwl_for(const pylda_ctx* ctx, int n) { const QuiltGeom q = quilt_geom_for(ctx, n); return q.W * 100 + q.RWL; }

// Hybrid kernel: 128 < K <= 256 (ldk 192 / 256), or long documents at ldk 64 / 128.
bool qhybrid_ok(const pylda_ctx* ctx, int n)
{
    return ctx->ldk % 64 == 0 && ctx->ldk <= 256 && n <= 128 + 8 * (kQhMaxTail - 4) && ctx->lds_limit >= 160 * 1024;
}

// Wide tiered kernel: ldk 128 / 192 / 256, documents up to 624 distinct terms.
bool qwide_ok(const pylda_ctx* ctx, int n)
{
    return (ctx->ldk == 128 || ctx->ldk == 192 || ctx->ldk == 256) && n <= kQwRegWords + 8 * (kQwMaxTail - 2) &&
           ctx->lds_limit >= 160 * 1024;
}

// 1: one round of tail steps per wavefront (at most 8 steps = 16 words: N <= 256), 2: several.
int qwide_rounds_for(int n) { return n <= kQwRegWords + 8 * 16 ? 1 : 2; }

// Streaming quilt kernel: any ldk that is a multiple of 64 up to 512, documents up to 1000 terms.
bool qstream_ok(const pylda_ctx* ctx, int n) { return ctx->ldk % 64 == 0 && ctx->ldk <= 512 && n <= 1000; }

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
    if ((ctx->force_variant < 0 || ctx->force_variant == kQwide) && qwide_ok(ctx, n)) {
        *lds_bytes = 0;
        return kQwide;
    }
    if ((ctx->force_variant < 0 || ctx->force_variant == kQhybrid) && qhybrid_ok(ctx, n)) {
        *lds_bytes = 0;
        return kQhybrid;
    }
    if ((ctx->force_variant < 0 || ctx->force_variant == kQstream) && qstream_ok(ctx, n)) {
        *lds_bytes = 0;
        return kQstream;
    }
generic:
    const int K = ctx->K, stride = tile_stride_for(K);
    const size_t l64 = generic_lds_layout(K, n, stride, 64, false).total;
    const size_t l256 = generic_lds_layout(K, n, stride, 256, false).total;
    const size_t l512 = generic_lds_layout(K, n, stride, 512, false).total;
    int v;
    if (ctx->force_variant >= 0 && ctx->force_variant < kSlab) v = ctx->force_variant;
    else if (ctx->force_variant == kGenericHuge) v = kGenericGlobal;
    else if (l64 <= 20 * 1024) v = kGeneric64;
    else if (l256 <= 64 * 1024) v = kGeneric256;
    else if (l512 <= ctx->lds_limit) v = kGeneric512;
    else v = kGenericGlobal;
    // a forced LDS variant that does not fit degrades to the global-tile kernel
    const size_t need = v == kGeneric64 ? l64 : v == kGeneric256 ? l256 : l512;
    if (v != kGenericGlobal && need > ctx->lds_limit) v = kGenericGlobal;
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
              : r.variant == kQwide ? qwide_rounds_for(n) : r.variant == kSlab ? slab_geom_for(ctx, n).RN : 0;
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

template <int NT, int MODE>
int launch_generic(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    auto kern = estep_generic_kernel<NT, MODE>;
    if (L.lds_bytes > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)L.lds_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(NT), L.lds_bytes, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

template <int W, int RK, int RN>
int launch_slab(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    auto kern = estep_slab_kernel<W, RK, RN>;
    const size_t lds = SlabLds<W, RK, RN>::total;
    if (lds > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(kWave * W), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

int launch_slab_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    const int W = ctx->ldk / L.rk;
#define SLAB_CASE(w_, rk_, rn_) \
    if (W == w_ && L.rk == rk_ && L.rn == rn_) return launch_slab<w_, rk_, rn_>(ctx, p, L);
#define SLAB_RN32(w) SLAB_CASE(w, 32, 1) SLAB_CASE(w, 32, 2)
#define SLAB_RN16(w) SLAB_CASE(w, 16, 1) SLAB_CASE(w, 16, 2) SLAB_CASE(w, 16, 3) SLAB_CASE(w, 16, 4)
    SLAB_RN32(1) SLAB_RN32(2) SLAB_RN32(4)
    SLAB_RN16(1) SLAB_RN16(2) SLAB_RN16(4) SLAB_RN16(8)
    SLAB_CASE(1, 16, 6) SLAB_CASE(2, 16, 6) SLAB_CASE(4, 16, 6)
#undef SLAB_RN16
#undef SLAB_RN32
#undef SLAB_CASE
    return fail(ctx, PYLDA_ERR_STATE, "no slab kernel for W=%d RK=%d RN=%d", W, L.rk, L.rn);
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

template <int W, int RK>
int launch_slab_uber(pylda_ctx* ctx, const EstepParams& p, const pylda_corpus* c, int from)
{
    SlabUberClasses cls;
    memset(&cls, 0, sizeof cls);
    cls.n = (int)c->plan.size() - from;
    size_t lds = 0;
    int64_t docs = 0;
    for (int i = 0; i < cls.n; ++i) {
        const Launch& L = c->plan[(size_t)(from + i)];
        cls.first[i] = (int)(L.first - c->plan[(size_t)from].first);
        cls.rn[i] = L.rn;
        docs += L.count;
        // (the largest class comes first: documents are scheduled longest first)
        const size_t need = RK == 32 ? (L.rn == 1 ? SlabLds<W, RK, 1>::total : SlabLds<W, RK, 2>::total)
                          : L.rn == 1 ? SlabLds<W, RK, 1>::total : L.rn == 2 ? SlabLds<W, RK, 2>::total
                          : L.rn == 3 ? SlabLds<W, RK, 3>::total : L.rn == 4 ? SlabLds<W, RK, 4>::total : SlabLds<W, RK, 6>::total;
        lds = std::max(lds, need);
    }
    cls.first[cls.n] = (int)docs;
    bool has_long = false;
    for (int i = 0; i < cls.n; ++i) has_long = has_long || cls.rn[i] > 4;
    auto kern = (RK == 16 && has_long) ? estep_slab_uber_kernel<W, RK, true> : estep_slab_uber_kernel<W, RK, false>;
    if (lds > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)docs), dim3(kWave * W), lds, ctx->stream, p, cls);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

int launch_slab_uber_any(pylda_ctx* ctx, const EstepParams& p, const pylda_corpus* c, int from)
{
    const int rk = c->plan[(size_t)from].rk, W = ctx->ldk / rk;
#define UBER_CASE(w_, rk_) if (W == w_ && rk == rk_) return launch_slab_uber<w_, rk_>(ctx, p, c, from);
    UBER_CASE(1, 32) UBER_CASE(2, 32) UBER_CASE(4, 32)
    UBER_CASE(1, 16) UBER_CASE(2, 16) UBER_CASE(4, 16)
#undef UBER_CASE
    return fail(ctx, PYLDA_ERR_STATE, "no slab uber kernel for W=%d RK=%d", W, rk);
}

template <int W, int KRL, int RWL>
int launch_quilt(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    auto kern = estep_quilt_kernel<W, KRL, RWL>;
    const size_t lds = QuiltLds<W, KRL, RWL>::total;
    if (lds > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(kWave * W), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

int launch_quilt_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    const int KRL = ctx->ldk / 16;
#define QUILT_CASE(w_, krl_, rwl_) \
    if (KRL == krl_ && L.rn == w_ * 100 + rwl_) return launch_quilt<w_, krl_, rwl_>(ctx, p, L);
    QUILT_CASE(8, 4, 2) QUILT_CASE(8, 4, 4) QUILT_CASE(8, 4, 8) QUILT_CASE(8, 8, 2) QUILT_CASE(8, 8, 4) QUILT_CASE(8, 8, 8)
    QUILT_CASE(8, 4, 6) QUILT_CASE(8, 4, 7) QUILT_CASE(8, 8, 6) QUILT_CASE(8, 8, 7)
    QUILT_CASE(12, 4, 4) QUILT_CASE(12, 8, 4)
#undef QUILT_CASE
    return fail(ctx, PYLDA_ERR_STATE, "no quilt kernel for KRL=%d RWL=%d", KRL, L.rn);
}

template <int TL, int RWL, int TWL>
int launch_quad(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    auto kern = estep_quad_kernel<TL, RWL, TWL>;
    const size_t lds = QuadLds<TL, RWL, TWL>::total + (size_t)ctx->lds_pad;
    if (lds > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(kWave * (TL / 4)), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

int launch_quad_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    switch (L.rn) {
    case 160800: return launch_quad<16, 8, 0>(ctx, p, L);
    case 161000: return launch_quad<16, 10, 0>(ctx, p, L);
    case 161001: return launch_quad<16, 10, 1>(ctx, p, L);
    case 161002: return launch_quad<16, 10, 2>(ctx, p, L);
    case 161003: return launch_quad<16, 10, 3>(ctx, p, L);
    case 161004: return launch_quad<16, 10, 4>(ctx, p, L);
    case 320800: return launch_quad<32, 8, 0>(ctx, p, L);
    case 321000: return launch_quad<32, 10, 0>(ctx, p, L);
    case 321001: return launch_quad<32, 10, 1>(ctx, p, L);
    case 321002: return launch_quad<32, 10, 2>(ctx, p, L);
    case 321003: return launch_quad<32, 10, 3>(ctx, p, L);
    case 321004: return launch_quad<32, 10, 4>(ctx, p, L);
    }
    return fail(ctx, PYLDA_ERR_STATE, "no quad kernel for geometry %d", L.rn);
}

template <int NP, int RWL, int TWL>
int launch_qfuse_np(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    auto kern = estep_qfuse_kernel<NP, RWL, TWL>;
    const size_t lds = QfuseLds<NP, TWL>::total;
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(512), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

int launch_qfuse(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
#ifndef PYLDA_QF4_RWL
#define PYLDA_QF4_RWL 6
#endif
#ifndef PYLDA_QF4_TWL
#define PYLDA_QF4_TWL 2
#endif
    return ctx->ldk == 512 ? launch_qfuse_np<4, PYLDA_QF4_RWL, PYLDA_QF4_TWL>(ctx, p, L) : launch_qfuse_np<3, 8, 2>(ctx, p, L);
}

template <int NP>
int launch_qfusek_np(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    auto kern = estep_qfusek_kernel<NP>;
    const size_t lds = QfusekLds<NP>::total;
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(512), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

int launch_qfusek(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    switch (ctx->ldk / 128) {
    case 5: return launch_qfusek_np<5>(ctx, p, L);
    case 6: return launch_qfusek_np<6>(ctx, p, L);
    case 7: return launch_qfusek_np<7>(ctx, p, L);
    case 8: return launch_qfusek_np<8>(ctx, p, L);
    }
    return fail(ctx, PYLDA_ERR_STATE, "no fused streaming kernel for table stride %d", ctx->ldk);
}

template <int KRL>
int launch_qstream(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    auto kern = estep_qstream_kernel<8, KRL>;
    const size_t lds = QstreamLds<8, KRL>::total;
    if (lds > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(kWave * 8), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

template <int KRL>
int launch_qhybrid(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    using Lds = QhybridLds<8, KRL, 4>;
    auto kern = estep_qhybrid_kernel<8, KRL, 4>;
    const size_t limit = 160 * 1024;
    const int rows_per_wave = std::min(kQhMaxTail, Lds::rows_that_fit(limit) / 8);
    const size_t lds = Lds::fixed_total + (size_t)8 * rows_per_wave * Lds::kRowDoubles * 8;
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(kWave * 8), lds, ctx->stream, p, rows_per_wave);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

template <int JJ, bool MULTI>
int launch_qwide(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    using Lds = QwideLds<8, JJ>;
    auto kern = estep_qwide_kernel<8, JJ, MULTI>;
    const size_t limit = 160 * 1024;
    const int rows_per_wave = std::min(kQwMaxTail, Lds::rows_that_fit(limit) / 8) & ~1;
    const size_t lds = Lds::fixed_total + (size_t)8 * rows_per_wave * Lds::kRowDoubles * 8;
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(kWave * 8), lds, ctx->stream, p, rows_per_wave);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

int launch_qwide_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    const bool multi = L.rn != 1;       // more than one round of tail steps per wavefront
    switch (ctx->ldk / 64) {
    case 2: return multi ? launch_qwide<2, true>(ctx, p, L) : launch_qwide<2, false>(ctx, p, L);
    case 3: return multi ? launch_qwide<3, true>(ctx, p, L) : launch_qwide<3, false>(ctx, p, L);
    case 4: return multi ? launch_qwide<4, true>(ctx, p, L) : launch_qwide<4, false>(ctx, p, L);
    }
    return fail(ctx, PYLDA_ERR_STATE, "no wide tiered kernel for table stride %d", ctx->ldk);
}

int launch_qhybrid_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    switch (ctx->ldk / 16) {
    case 4: return launch_qhybrid<4>(ctx, p, L);
    case 8: return launch_qhybrid<8>(ctx, p, L);
    case 12: return launch_qhybrid<12>(ctx, p, L);
    case 16: return launch_qhybrid<16>(ctx, p, L);
    }
    return fail(ctx, PYLDA_ERR_STATE, "no hybrid kernel for table stride %d", ctx->ldk);
}

int launch_qstream_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    switch (ctx->ldk / 16) {
    case 4: return launch_qstream<4>(ctx, p, L);
    case 8: return launch_qstream<8>(ctx, p, L);
    case 12: return launch_qstream<12>(ctx, p, L);
    case 16: return launch_qstream<16>(ctx, p, L);
    case 20: return launch_qstream<20>(ctx, p, L);
    case 24: return launch_qstream<24>(ctx, p, L);
    case 28: return launch_qstream<28>(ctx, p, L);
    case 32: return launch_qstream<32>(ctx, p, L);
    }
    return fail(ctx, PYLDA_ERR_STATE, "no streaming kernel for table stride %d", ctx->ldk);
}

int enqueue_prepare(pylda_ctx* ctx, bool heldout)
{
    const int K = ctx->K, V = ctx->V;
    hipLaunchKernelGGL(eta_rowsum_psi_kernel, dim3(K), dim3(256), 0, ctx->stream, ctx->d_eta, K, V,
                       ctx->d_psi_rowsum);
    hipLaunchKernelGGL(elog_transpose_kernel, dim3((V + 31) / 32, (K + 31) / 32), dim3(256), 0,
                       ctx->stream, ctx->d_eta, ctx->d_psi_rowsum, K, V, ctx->ldk, ctx->d_elog);
    hipLaunchKernelGGL(row_shift_exp_kernel, dim3((V + 3) / 4), dim3(256), 0, ctx->stream,
                       ctx->d_elog, K, V, ctx->ldk, ctx->d_expElog, ctx->d_expElog_elog, ctx->d_shift);
    if (heldout)
        hipLaunchKernelGGL(topic_lse_kernel, dim3(K), dim3(256), 0, ctx->stream, ctx->d_elog,
                           ctx->d_shift, K, V, ctx->ldk, ctx->d_topic_lse);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

// Geometry of the persistent sweep (sstats_sweep.h) for V terms: terms per wavefront and wavefronts per workgroup (one
// workgroup per CU) such that the fewest passes over the document blocks cover all terms.
struct SweepGeom { int T, WPB, passes; };
SweepGeom sweep_geom_for(const pylda_ctx* ctx, int V)
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
            // (cfg 4, t = 2 GB: 64 blocks 55 ms, 128 53, 256 48, unblocked 65; 240 by this rule) - and no more
            // partial rows than fit a quarter of the free device memory
            const int by_l2 = std::max(8, 8 * (int)std::lround(t_bytes / (8 * 4.3e6)));
            const int by_pairs = (int)std::min<double>(1e6, (double)nnz / (8.0 * V)) / 8 * 8;
            NB = std::min(by_l2, by_pairs);
            if (NB < 8) NB = 1;
        }
    }
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
        const double budget = ctx->gather_round_mb > 0 ? (double)ctx->gather_round_mb * 1048576.0 : 4.0 * 1073741824.0;
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
        A(dev_alloc(ctx, &c->d_rendezvous, (size_t)1));
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
        const double budget = ctx->gather_round_mb > 0 ? (double)ctx->gather_round_mb * 1048576.0 : 4.0 * 1073741824.0;
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
    else if (ctx->gather_rows && (ldk == 64 || ldk == 128 || ldk == 256)) {
        if (ldk == 128 && ctx->gather_rows == 2)
            hipLaunchKernelGGL((sstats_gather_bulk_kernel<2, PYLDA_GATHER_U, P>), g1, dim3(256), 0, ctx->stream, GATHER_ARGS, c->d_partial, order, r.seg_lo);
        else if (ldk == 256 && ctx->gather_rows == 2)
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

void fill_sweep_params(pylda_ctx* ctx, pylda_corpus* c, SweepParams& sp)
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
    sp.spin_limit = (unsigned)ctx->sweep_spin;       // (4000 ~ 5 ms: a rendezvous that does not complete costs L2 locality, nothing else)
}

int enqueue_sstats_gather(pylda_ctx* ctx, pylda_corpus* c)
{
    const int ldk = ctx->ldk;
    if (c->sweep) {
        SweepParams sp;
        fill_sweep_params(ctx, c, sp);
        (void)hipMemsetAsync(c->d_rendezvous, 0, sizeof(unsigned), ctx->stream);
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

hipEvent_t take_event(pylda_ctx* ctx)
{
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void drain_events(pylda_ctx* ctx)
{
    for (auto& br : ctx->pending_events) {
        float ms = 0.f;
        if (hipEventSynchronize(br.b) == hipSuccess && hipEventElapsedTime(&ms, br.a, br.b) == hipSuccess) {
            if (br.slot == -1) ctx->doc_kernel_ms += ms;
            else if (br.slot == -2) ctx->sstats_kernel_ms += ms;
            else if ((size_t)br.slot < ctx->class_ms.size()) ctx->class_ms[(size_t)br.slot] += ms;
        }
        ctx->event_pool.push_back(br.a);
        ctx->event_pool.push_back(br.b);
    }
    ctx->pending_events.clear();
}

}  // namespace

extern "C" {

const char* pylda_version(void) { return "pylda_hip 0.3 (gfx950, abi 3)"; }
int pylda_abi_version(void) { return PYLDA_ABI_VERSION; }

int pylda_device_count(int* count)
{
    if (!count) return PYLDA_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return PYLDA_OK;
}

const char* pylda_last_error(const pylda_ctx* ctx)
{
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int pylda_create(int device, int K, int V, pylda_ctx** out)
{
    if (!out) return fail(nullptr, PYLDA_ERR_INVALID, "pylda_create: out is NULL");
    *out = nullptr;
    if (K < 1 || V < 1) return fail(nullptr, PYLDA_ERR_INVALID, "pylda_create: K=%d V=%d", K, V);
    if ((int64_t)K * V > ((int64_t)1 << 40))
        return fail(nullptr, PYLDA_ERR_INVALID, "pylda_create: K*V too large");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, PYLDA_ERR_HIP,
                    "pylda_create: no HIP device visible; this library has no CPU fallback");
    if (device < 0 || device >= ndev)
        return fail(nullptr, PYLDA_ERR_INVALID, "pylda_create: device %d of %d", device, ndev);
    pylda_ctx* ctx = new (std::nothrow) pylda_ctx;
    if (!ctx) return fail(nullptr, PYLDA_ERR_OOM, "pylda_create: host allocation failed");
    ctx->device = device;
    ctx->K = K;
    ctx->V = V;
    // table stride: K rounded up to 16 / 32 / a multiple of 64, from 257 to 1024 to a multiple of 128 (the fused
    // streaming kernels' rows are 64 lanes x 16-byte pieces)
    ctx->ldk = K <= 16 ? 16 : K <= 32 ? 32 : K <= 256 ? (K + 63) / 64 * 64 : K <= 1024 ? (K + 127) / 128 * 128 : (K + 63) / 64 * 64;
    auto bail = [&](int code) {
        g_create_error = ctx->err;
        pylda_destroy(ctx);
        return code;
    };
#define CREATE_TRY(expr)                     \
    do {                                     \
        int rc_ = (expr);                    \
        if (rc_ != PYLDA_OK) return bail(rc_); \
    } while (0)
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return (int)PYLDA_OK;
        return fail(ctx, e == hipErrorOutOfMemory ? PYLDA_ERR_OOM : PYLDA_ERR_HIP, "%s: %s", what,
                    hipGetErrorString(e));
    };
    CREATE_TRY(hip_ok(hipSetDevice(device), "hipSetDevice"));
    hipDeviceProp_t prop;
    CREATE_TRY(hip_ok(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties"));
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    size_t lds = prop.maxSharedMemoryPerMultiProcessor;
    if (lds < 64 * 1024) lds = 64 * 1024;
    if (lds > 160 * 1024) lds = 160 * 1024;
    ctx->lds_limit = lds;
    CREATE_TRY(hip_ok(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking),
                      "hipStreamCreate"));
    ctx->stream = ctx->own_stream;
    for (int i = 0; i < pylda_ctx::kAux; ++i) {
        CREATE_TRY(hip_ok(hipStreamCreateWithFlags(&ctx->aux_stream[i], hipStreamNonBlocking), "hipStreamCreate"));
        CREATE_TRY(hip_ok(hipEventCreateWithFlags(&ctx->join_event[i], hipEventDisableTiming), "hipEventCreate"));
    }
    CREATE_TRY(hip_ok(hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming), "hipEventCreate"));
    const size_t kv = (size_t)K * V, wk = (size_t)V * ctx->ldk;
    CREATE_TRY(dev_alloc(ctx, &ctx->d_eta, kv));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_elog, wk));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_expElog, wk));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_expElog_elog, wk));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_sstats, wk));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_kv_scratch, kv));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_shift, (size_t)V));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_beta, (size_t)V));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_psi_rowsum, (size_t)K));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_topic_lse, (size_t)K));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_alpha, (size_t)K));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_small, (size_t)(4 * K + 16)));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_partial, (size_t)1024 * K));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_outer, (size_t)(3 * K + 8)));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_newton_work, (size_t)4 * K));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_work, (size_t)2));
    CREATE_TRY(hip_ok(hipMemsetAsync(ctx->d_work, 0, 2 * sizeof(double), ctx->stream), "hipMemsetAsync"));
    CREATE_TRY(hip_ok(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pin), (size_t)(5 * K + 8) * sizeof(double), hipHostMallocDefault),
                      "hipHostMalloc"));
    for (int i = 0; i < 2; ++i)
        CREATE_TRY(hip_ok(hipEventCreateWithFlags(&ctx->alpha_event[i], hipEventDisableTiming), "hipEventCreate"));
    CREATE_TRY(hip_ok(hipMemsetAsync(ctx->d_sstats, 0, wk * sizeof(double), ctx->stream),
                      "hipMemsetAsync"));
#undef CREATE_TRY
    *out = ctx;
    return PYLDA_OK;
}

void pylda_destroy(pylda_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->comm) pylda::comm_destroy(ctx->comm);
    dev_free(ctx->d_comm_small);
    drain_events(ctx);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    dev_free(ctx->d_eta); dev_free(ctx->d_elog); dev_free(ctx->d_expElog); dev_free(ctx->d_expElog_elog); dev_free(ctx->d_sstats);
    dev_free(ctx->d_kv_scratch); dev_free(ctx->d_shift); dev_free(ctx->d_beta);
    dev_free(ctx->d_psi_rowsum); dev_free(ctx->d_topic_lse); dev_free(ctx->d_alpha);
    dev_free(ctx->d_small); dev_free(ctx->d_partial); dev_free(ctx->d_outer); dev_free(ctx->d_newton_work); dev_free(ctx->d_work); dev_free(ctx->d_eta_ckpt);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    for (int i = 0; i < 2; ++i)
        if (ctx->alpha_event[i]) (void)hipEventDestroy(ctx->alpha_event[i]);
    for (hipEvent_t e : ctx->mark_event)
        if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < pylda_ctx::kAux; ++i) {
        if (ctx->aux_stream[i]) { (void)hipStreamSynchronize(ctx->aux_stream[i]); (void)hipStreamDestroy(ctx->aux_stream[i]); }
        if (ctx->join_event[i]) (void)hipEventDestroy(ctx->join_event[i]);
    }
    if (ctx->fork_event) (void)hipEventDestroy(ctx->fork_event);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int pylda_set_stream(pylda_ctx* ctx, void* hip_stream)
{
    // As everywhere in HIP, a NULL handle is the device's default ("null") stream - which is
    // what torch.cuda.current_stream().cuda_stream reports for torch's default stream.
    if (!ctx) return PYLDA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return PYLDA_OK;
}

int pylda_use_own_stream(pylda_ctx* ctx)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = ctx->own_stream;
    return PYLDA_OK;
}

int pylda_synchronize(pylda_ctx* ctx)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PYLDA_OK;
}

int pylda_set_option(pylda_ctx* ctx, const char* name, int64_t value)
{
    if (!ctx || !name) return PYLDA_ERR_INVALID;
    if (!strcmp(name, "force_logspace")) ctx->force_logspace = value != 0;
    else if (!strcmp(name, "force_variant")) {
        if (value < -1 || value > kVariantLast || value == kRetired5)
            return fail(ctx, PYLDA_ERR_INVALID, "force_variant %lld is not a kernel variant", (long long)value);
        ctx->force_variant = (int)value;
        ctx->plan_epoch += 1;
    } else if (!strcmp(name, "gather_rows")) {
        ctx->gather_rows = (int)value;               // 0: 64-topic chunks, 1: whole rows, 2: whole rows, postings in bulk
    } else if (!strcmp(name, "gather_blocks")) {     // (takes effect for corpora created afterwards)
        if (value > 1 && value % 8) return fail(ctx, PYLDA_ERR_INVALID, "gather_blocks=%lld: a multiple of 8, or -1 / 0 / 1", (long long)value);
        ctx->gather_blocks = (int)value;
    } else if (!strcmp(name, "quilt_odd")) {
        ctx->quilt_odd = value != 0;
        ctx->plan_epoch += 1;
    } else if (!strcmp(name, "sweep_spin")) {
        ctx->sweep_spin = (int)std::max<int64_t>(0, value);
    } else if (!strcmp(name, "gather_sweep")) {      // (takes effect for corpora whose postings are built afterwards)
        ctx->gather_sweep = (int)std::min<int64_t>(2, std::max<int64_t>(0, value));
    } else if (!strcmp(name, "gather_round_mb")) {   // (takes effect for corpora whose postings are built afterwards)
        ctx->gather_round_mb = (int)std::max<int64_t>(0, value);
    } else if (!strcmp(name, "slab_uber")) {
        ctx->slab_uber = value != 0;
    } else if (!strcmp(name, "wide_postings")) {     // (takes effect for corpora whose postings are built afterwards)
        ctx->wide_postings = value != 0;
    } else if (!strcmp(name, "lds_pad")) {
        ctx->lds_pad = (int)value;
    } else if (!strcmp(name, "quad")) {
        ctx->quad = value != 0;
        ctx->plan_epoch += 1;
    } else if (!strcmp(name, "quilt12")) {
        ctx->quilt12 = value != 0;
        ctx->plan_epoch += 1;
    } else if (!strcmp(name, "doc_values")) {
        ctx->doc_values = value != 0;
    } else
        return fail(ctx, PYLDA_ERR_INVALID, "unknown option '%s'", name);
    return PYLDA_OK;
}

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
    A(dev_alloc(ctx, &c->d_flag_list, (size_t)D));
    A(dev_alloc(ctx, &c->d_flag_count, (size_t)1));
    A(dev_alloc(ctx, &c->d_scalars, (size_t)4));
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
    dev_free(c->d_status); dev_free(c->d_flag_list); dev_free(c->d_flag_count); dev_free(c->d_scalars); dev_free(c->d_entropy_partial);
    dev_free(c->d_tfinal); dev_free(c->d_rfinal); dev_free(c->d_term_scratch); dev_free(c->d_post_doc);
    if (c->d_post_pos) (void)hipFree(c->d_post_pos);
    c->d_post_pos = nullptr;
    dev_free(c->d_seg_begin); dev_free(c->d_seg_end); dev_free(c->d_word_seg_ptr); dev_free(c->d_partial); dev_free(c->d_exec_order);
    dev_free(c->d_seg_block); dev_free(c->d_term_of); dev_free(c->d_rendezvous);
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

int pylda_set_eta(pylda_ctx* ctx, const double* eta_kv)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!eta_kv) return fail(ctx, PYLDA_ERR_INVALID, "set_eta: NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_eta, eta_kv, (size_t)ctx->K * ctx->V * sizeof(double),
                                hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // host buffer is not retained
    ctx->have_eta = true;
    return PYLDA_OK;
}

int pylda_get_eta(pylda_ctx* ctx, double* eta_kv)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!eta_kv) return fail(ctx, PYLDA_ERR_INVALID, "get_eta: NULL");
    if (!ctx->have_eta) return fail(ctx, PYLDA_ERR_STATE, "get_eta: eta was never set");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(eta_kv, ctx->d_eta, (size_t)ctx->K * ctx->V * sizeof(double),
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PYLDA_OK;
}

int pylda_set_alpha(pylda_ctx* ctx, const double* alpha_k)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!alpha_k) return fail(ctx, PYLDA_ERR_INVALID, "set_alpha: NULL");
    for (int k = 0; k < ctx->K; ++k)
        if (!(alpha_k[k] > 0.0) || !std::isfinite(alpha_k[k]))
            return fail(ctx, PYLDA_ERR_INVALID, "set_alpha: alpha[%d]=%g is not positive", k, alpha_k[k]);
    // (the device already holds exactly these values: after an alpha update on the device - pylda_outer_fetch - the
    //  host hands back what it was handed)
    if (ctx->have_alpha && ctx->h_alpha.size() == (size_t)ctx->K &&
        memcmp(ctx->h_alpha.data(), alpha_k, (size_t)ctx->K * sizeof(double)) == 0)
        return PYLDA_OK;
    ctx->h_alpha.assign(alpha_k, alpha_k + ctx->K);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // no stream wait: the values go through one of two pinned slots (the copy is ordered on the stream behind the
    // kernels still reading the previous alpha); a slot is reused only when its last copy has left it
    const int slot = ctx->alpha_slot;
    ctx->alpha_slot ^= 1;
    if (ctx->alpha_event_used[slot]) HIP_TRY(ctx, hipEventSynchronize(ctx->alpha_event[slot]));
    double* pin = ctx->h_pin + (size_t)slot * ctx->K;
    memcpy(pin, alpha_k, (size_t)ctx->K * sizeof(double));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_alpha, pin, (size_t)ctx->K * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->alpha_event[slot], ctx->stream));
    ctx->alpha_event_used[slot] = true;
    ctx->have_alpha = true;
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
    if (!heldout && (rc = build_postings(c)) != PYLDA_OK) return rc;
    HIP_TRY(ctx, hipMemsetAsync(c->d_flag_count, 0, sizeof(int32_t), ctx->stream));

    EstepParams p;
    p.K = K;
    p.V = V;
    p.ldk = ctx->ldk;
    p.expElog = ctx->d_expElog;
    p.expElog_elog = ctx->d_expElog_elog;
    p.shift = ctx->d_shift;
    p.topic_lse = ctx->d_topic_lse;
    p.alpha = ctx->d_alpha;
    double asum = 0.0, alg = 0.0;
    for (double a : ctx->h_alpha) {
        asum += a;
        alg += std::lgamma(a);
    }
    p.alpha_term = std::lgamma(asum) - alg;                           // :195
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

    {
        const double span = tol * K;
        ctx->exact_stop = !(span >= 3.725290298461914e-09 /* 2^-28 */ && span < 1024.0);
    }
    if (c->plan_epoch != ctx->plan_epoch || c->plan_exact != ctx->exact_stop) build_plan(c);
    if (!c->d_term_scratch)
        for (const Launch& L : c->plan)
            if (L.variant == kGenericHuge) {
                rc = dev_alloc(ctx, &c->d_term_scratch, (size_t)c->nnz);
                if (rc != PYLDA_OK) return rc;
                p.term_scratch = c->d_term_scratch;
                break;
            }
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
        // test hook: mark every document for the log-space kernel
        std::vector<int32_t> ones((size_t)c->D, 1);
        HIP_TRY(ctx, hipMemcpyAsync(c->d_status, ones.data(), (size_t)c->D * sizeof(int32_t),
                                    hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
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
        size_t launch_index = 0;
        for (const Launch& L : c->plan) {
            const int slot = (int)launch_index;
            if (launch_index >= separate) break;
            if (fan_out) ctx->stream = ctx->aux_stream[launch_index % pylda_ctx::kAux];
            ++launch_index;
            p.order = c->d_order + L.first;
            p.n_cap = L.n_cap;
            p.tile_stride = L.tile_stride;
            const int class_bracket = open_bracket(slot, ctx->stream);
            switch (L.variant) {
            case kGeneric64: rc = launch_generic<64, 0>(ctx, p, L); break;
            case kGeneric256: rc = launch_generic<256, 0>(ctx, p, L); break;
            case kGeneric512: rc = launch_generic<512, 0>(ctx, p, L); break;
            case kGenericHuge: rc = launch_generic<256, 2>(ctx, p, L); break;
            case kSlab: rc = launch_slab_any(ctx, p, L); break;
            case kQuilt: rc = launch_quilt_any(ctx, p, L); break;
            case kQstream: rc = launch_qstream_any(ctx, p, L); break;
            case kQhybrid: rc = launch_qhybrid_any(ctx, p, L); break;
            case kQwide: rc = launch_qwide_any(ctx, p, L); break;
            case kQuad: rc = launch_quad_any(ctx, p, L); break;
            case kQfuse: rc = launch_qfuse(ctx, p, L); break;
            case kQfusek: rc = launch_qfusek(ctx, p, L); break;
            default: rc = launch_generic<256, 1>(ctx, p, L); break;
            }
            close_bracket(class_bracket, ctx->stream);
            if (rc != PYLDA_OK) {
                join();
                return rc;
            }
        }
        ctx->stream = main_stream;
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
    }
    // the document terms the register kernels left out on the training fast path (status 3; doc_terms.h)
    bool leaves_terms = false;          // (slab and generic kernels always finish their documents themselves)
    for (const Launch& L : c->plan)
        leaves_terms = leaves_terms || L.variant == kQuad || L.variant == kQuilt || L.variant == kQwide || L.variant == kQfuse || L.variant == kQfusek;
    if (!heldout && !p.want_doc_ll && c->D > 0 && leaves_terms)
        hipLaunchKernelGGL(doc_terms_kernel, dim3((unsigned)((c->D + 3) / 4)), dim3(256), 0, ctx->stream, p, c->D);
    close_bracket(doc_bracket, ctx->stream);
    if (ctx->profiling) ctx->estep_calls += 1;
    if (ctx->profiling && c->D > 0)       // inner iterations actually executed, for the fp64 roofline and doc-iterations/s
        hipLaunchKernelGGL(work_count_kernel, dim3(1), dim3(1024), 0, ctx->stream, c->d_iters, c->d_doc_ptr, c->D, ctx->d_work);

    // sufficient statistics (:207): gather pass over the postings, no atomics
    if (!heldout) {
        if (ctx->force_logspace) {
            HIP_TRY(ctx, hipMemsetAsync(c->d_rfinal, 0, (size_t)c->nnz * sizeof(double), ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(c->d_tfinal, 0, (size_t)c->D * ctx->ldk * sizeof(double), ctx->stream));
        }
        const int ss_bracket = open_bracket(-2, ctx->stream);
        rc = enqueue_sstats_gather(ctx, c);
        close_bracket(ss_bracket, ctx->stream);
        if (rc != PYLDA_OK) return rc;
    }
    // safety net: documents the linear-space kernels flagged are redone in log space
    if (c->D > 0) {
        hipLaunchKernelGGL(flagged_collect_kernel, dim3((unsigned)((c->D + 255) / 256)), dim3(256), 0,
                           ctx->stream, c->d_status, c->D, c->d_flag_list, c->d_flag_count);
        p.order = nullptr;
        const unsigned grid = (unsigned)std::min<int64_t>(c->D, 4 * (int64_t)ctx->num_cu);
        if (logspace_lds_bytes(K) > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(estep_logspace_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)logspace_lds_bytes(K)));
        hipLaunchKernelGGL(estep_logspace_kernel, dim3(grid), dim3(256), logspace_lds_bytes(K),
                           ctx->stream, p, ctx->d_elog, ctx->d_sstats, c->d_flag_list, c->d_flag_count);
    }
    hipLaunchKernelGGL(vector_sum3_kernel, dim3(heldout ? 2 : 3), dim3(1024), 0, ctx->stream, SumJob{c->d_doc_ll, c->D, c->d_scalars},
                       SumJob{c->d_doc_wll, c->D, c->d_scalars + 1},
                       SumJob{c->d_entropy_partial, heldout ? 0 : c->ent_blocks, heldout ? nullptr : c->d_scalars + 2});
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
    double sc[3] = {0, 0, 0};
    int32_t nflag = 0;
    HIP_TRY(ctx, hipMemcpyAsync(sc, c->d_scalars, sizeof sc, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&nflag, c->d_flag_count, sizeof nflag, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // training fast path: the log B entropy term comes once per corpus from the statistics
    if (!c->last_doc_values) sc[0] -= sc[2];
    if (document_log_likelihood) *document_log_likelihood = sc[0];
    if (words_log_likelihood) *words_log_likelihood = sc[1];
    if (logspace_documents) *logspace_documents = nflag;
    return PYLDA_OK;
}

int pylda_get_sstats(pylda_ctx* ctx, double* sstats_kv)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!sstats_kv) return fail(ctx, PYLDA_ERR_INVALID, "get_sstats: NULL");
    if (!ctx->have_sstats) return fail(ctx, PYLDA_ERR_STATE, "get_sstats: no training-mode E-step has run");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int K = ctx->K, V = ctx->V;
    // device layout is (V, K); hand back numpy's (K, V)
    hipLaunchKernelGGL(transpose_kernel, dim3((K + 31) / 32, (V + 31) / 32), dim3(256), 0, ctx->stream,
                       ctx->d_sstats, V, K, ctx->ldk, V, ctx->d_kv_scratch);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(sstats_kv, ctx->d_kv_scratch, (size_t)K * V * sizeof(double),
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PYLDA_OK;
}

int pylda_set_sstats(pylda_ctx* ctx, const double* sstats_kv)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!sstats_kv) return fail(ctx, PYLDA_ERR_INVALID, "set_sstats: NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int K = ctx->K, V = ctx->V;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_kv_scratch, sstats_kv, (size_t)K * V * sizeof(double),
                                hipMemcpyHostToDevice, ctx->stream));
    // numpy's (K, V) -> device layout (V, K)
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_sstats, 0, (size_t)V * ctx->ldk * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(transpose_kernel, dim3((V + 31) / 32, (K + 31) / 32), dim3(256), 0, ctx->stream,
                       ctx->d_kv_scratch, K, V, V, ctx->ldk, ctx->d_sstats);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_sstats = true;
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

int pylda_table_stride(const pylda_ctx* ctx) { return ctx ? ctx->ldk : 0; }
void* pylda_sstats_device(pylda_ctx* ctx) { return ctx ? ctx->d_sstats : nullptr; }
void* pylda_eta_device(pylda_ctx* ctx) { return ctx ? ctx->d_eta : nullptr; }
void* pylda_gamma_device(pylda_corpus* c) { return c ? c->d_gamma : nullptr; }

int pylda_mark_device_state(pylda_ctx* ctx, int have_eta, int have_sstats)
{
    // the caller wrote eta / sstats through the device pointers above
    if (!ctx) return PYLDA_ERR_INVALID;
    if (have_eta >= 0) ctx->have_eta = have_eta != 0;
    if (have_sstats >= 0) ctx->have_sstats = have_sstats != 0;
    return PYLDA_OK;
}

namespace {
// The device half of m_step (:218-235): kernels only, nothing is read back.
int enqueue_mstep(pylda_ctx* ctx, pylda_corpus* c, const double* beta_v, bool want_alpha_ss, const char* who)
{
    if (!beta_v) return fail(ctx, PYLDA_ERR_INVALID, "%s: beta is NULL", who);
    if (!ctx->have_eta || !ctx->have_sstats)
        return fail(ctx, PYLDA_ERR_STATE, "%s: needs eta and the sufficient statistics of a training E-step", who);
    if (want_alpha_ss && (!c || c->ctx != ctx || !c->estep_done))
        return fail(ctx, PYLDA_ERR_STATE, "%s: alpha statistics need the corpus of the last E-step", who);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int K = ctx->K, V = ctx->V;
    // beta is constant over a run: its lgamma sums (V host lgamma calls) and the device copy are
    // refreshed only when the caller hands over different values
    if (ctx->h_beta.size() != (size_t)V || memcmp(ctx->h_beta.data(), beta_v, (size_t)V * sizeof(double)) != 0) {
        double bsum = 0.0, blg = 0.0;
        for (int v = 0; v < V; ++v) {
            if (!(beta_v[v] > 0.0)) return fail(ctx, PYLDA_ERR_INVALID, "%s: beta[%d]=%g", who, v, beta_v[v]);
            bsum += beta_v[v];
            blg += std::lgamma(beta_v[v]);
        }
        ctx->h_beta.clear();            // stays empty if the copy below fails
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_beta, beta_v, (size_t)V * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // host buffer is not retained
        ctx->h_beta.assign(beta_v, beta_v + V);
        ctx->beta_sum = bsum;
        ctx->beta_lgamma_sum = blg;
    }
    double* d_per_topic = ctx->d_small;          // K
    double* d_alpha_ss = ctx->d_small + K;       // K
    hipLaunchKernelGGL(mstep_topic_ll_kernel, dim3(K, kTopicChunks), dim3(256), 0, ctx->stream, ctx->d_eta, K, V,
                       ctx->d_partial);                                                                             // :224 (old eta)
    hipLaunchKernelGGL(mstep_topic_ll_finish_kernel, dim3((K + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_partial, K,
                       d_per_topic);
    hipLaunchKernelGGL(mstep_update_eta_kernel, dim3((K + 31) / 32, (V + 31) / 32), dim3(256), 0, ctx->stream,
                       ctx->d_sstats, ctx->d_beta, K, V, ctx->ldk, ctx->d_eta);                                               // :226
    if (want_alpha_ss) {
        const int nblocks = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (c->D + 3) / 4));     // (ctx->d_partial holds 1024 rows)
        hipLaunchKernelGGL(mstep_alpha_ss_kernel, dim3(nblocks), dim3(256), (size_t)4 * K * sizeof(double),
                           ctx->stream, c->d_gamma, c->D, K, ctx->d_partial);                                       // :232
        hipLaunchKernelGGL(column_sum_kernel, dim3((K + 63) / 64), dim3(256), 0, ctx->stream, ctx->d_partial,
                           nblocks, K, d_alpha_ss);                                                                 // :233
    }
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

double topic_ll_from(const pylda_ctx* ctx, const double* per_topic)
{
    double ll = ctx->K * (std::lgamma(ctx->beta_sum) - ctx->beta_lgamma_sum);                                       // :222
    for (int k = 0; k < ctx->K; ++k) ll += per_topic[k];
    return ll;
}
}  // namespace

int pylda_mstep(pylda_ctx* ctx, pylda_corpus* c, const double* beta_v, double* topic_log_likelihood,
                double* alpha_ss_k)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    const int rc = enqueue_mstep(ctx, c, beta_v, alpha_ss_k != nullptr, "mstep");
    if (rc != PYLDA_OK) return rc;
    const int K = ctx->K;
    std::vector<double> per_topic((size_t)K);
    HIP_TRY(ctx, hipMemcpyAsync(per_topic.data(), ctx->d_small, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (alpha_ss_k)
        HIP_TRY(ctx, hipMemcpyAsync(alpha_ss_k, ctx->d_small + K, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (topic_log_likelihood) *topic_log_likelihood = topic_ll_from(ctx, per_topic.data());
    return PYLDA_OK;
}

int pylda_mstep_enqueue(pylda_ctx* ctx, pylda_corpus* c, const double* beta_v, int hyper_parameter_iteration,
                        double hyper_parameter_decay_factor, int hyper_parameter_maximum_decay,
                        double hyper_parameter_converge_threshold)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!c || c->ctx != ctx || !c->estep_done || c->last_heldout)
        return fail(ctx, PYLDA_ERR_STATE, "mstep_enqueue: needs the corpus of the last training-mode E-step");
    if (hyper_parameter_iteration < 0 || hyper_parameter_maximum_decay < 0 || hyper_parameter_maximum_decay > 16)
        return fail(ctx, PYLDA_ERR_INVALID, "mstep_enqueue: hyper_parameter_iteration=%d, hyper_parameter_maximum_decay=%d (0..16)",
                    hyper_parameter_iteration, hyper_parameter_maximum_decay);
    ctx->newton_pending = hyper_parameter_iteration > 0;
    if (ctx->newton_pending) {
        ctx->newton.iterations = hyper_parameter_iteration;
        ctx->newton.maximum_decay = hyper_parameter_maximum_decay;
        ctx->newton.threshold = hyper_parameter_converge_threshold;
        for (int d = 0; d <= 16; ++d) ctx->newton.decay_power[d] = std::pow(hyper_parameter_decay_factor, (double)d);   // numpy.power
    }
    const int rc = enqueue_mstep(ctx, c, beta_v, true, "mstep_enqueue");
    if (rc != PYLDA_OK) return rc;
    const int K = ctx->K;
    hipLaunchKernelGGL(outer_pack_kernel, dim3(1), dim3(256), 0, ctx->stream, c->d_scalars, c->d_flag_count,
                       c->last_doc_values ? 1 : 0, (double)c->D, ctx->d_small + K, ctx->d_small, ctx->d_alpha, K, ctx->d_outer);
    HIP_TRY(ctx, hipGetLastError());
    ctx->outer_ready = true;
    return PYLDA_OK;
}

void* pylda_outer_device(pylda_ctx* ctx, int64_t* n_reduce)
{
    if (!ctx) return nullptr;
    if (n_reduce) *n_reduce = ctx->K + 4;
    return ctx->d_outer;
}

int pylda_allreduce_outer(pylda_ctx* ctx)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!ctx->comm) return fail(ctx, PYLDA_ERR_STATE, "allreduce_outer: pylda_comm_init has not been called");
    if (!ctx->outer_ready) return fail(ctx, PYLDA_ERR_STATE, "allreduce_outer: pylda_mstep_enqueue has not run");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    const int rc = pylda::comm_allreduce_sum_f64(ctx->comm, ctx->d_outer, (size_t)ctx->K + 4, ctx->stream, &err);
    return rc == PYLDA_OK ? rc : fail(ctx, rc, "allreduce_outer: %s", err.c_str());
}

int pylda_outer_fetch(pylda_ctx* ctx, double* document_log_likelihood, double* number_of_documents,
                      int64_t* logspace_documents, double* topic_log_likelihood, double* alpha_ss_k, double* alpha_k)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!ctx->outer_ready) return fail(ctx, PYLDA_ERR_STATE, "outer_fetch: pylda_mstep_enqueue has not run");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int K = ctx->K;
    if (ctx->newton_pending) {
        // behind the all-reduce of the packed values (the statistics and #documents are the global ones on every rank)
        hipLaunchKernelGGL(alpha_newton_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_outer + 4 + 2 * (size_t)K,
                           ctx->d_outer + 4, ctx->d_outer + 1, K, ctx->newton, ctx->d_newton_work, ctx->d_alpha);
        HIP_TRY(ctx, hipGetLastError());
    }
    double* host = ctx->h_pin + (size_t)2 * K;
    HIP_TRY(ctx, hipMemcpyAsync(host, ctx->d_outer, (size_t)(3 * K + 4) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));          // the ONE wait of an outer iteration
    ctx->outer_ready = false;
    if (ctx->newton_pending) ctx->h_alpha.assign(host + 4 + 2 * (size_t)K, host + 4 + 3 * (size_t)K);   // what d_alpha holds now
    ctx->newton_pending = false;
    if (alpha_k) memcpy(alpha_k, host + 4 + 2 * (size_t)K, (size_t)K * sizeof(double));
    if (document_log_likelihood) *document_log_likelihood = host[0];
    if (number_of_documents) *number_of_documents = host[1];
    if (logspace_documents) *logspace_documents = (int64_t)std::llround(host[2]);
    if (alpha_ss_k) memcpy(alpha_ss_k, host + 4, (size_t)K * sizeof(double));
    if (topic_log_likelihood) *topic_log_likelihood = topic_ll_from(ctx, host + 4 + K);
    return PYLDA_OK;
}

int pylda_host_alloc(int64_t bytes, void** out)
{
    if (!out || bytes < 0) return PYLDA_ERR_INVALID;
    *out = nullptr;
    const hipError_t e = hipHostMalloc(out, (size_t)std::max<int64_t>(bytes, 1), hipHostMallocDefault);
    if (e != hipSuccess) {
        *out = nullptr;
        return fail(nullptr, e == hipErrorOutOfMemory ? PYLDA_ERR_OOM : PYLDA_ERR_HIP, "host_alloc: %s", hipGetErrorString(e));
    }
    return PYLDA_OK;
}

int pylda_host_free(void* p)
{
    if (p && hipHostFree(p) != hipSuccess) return fail(nullptr, PYLDA_ERR_HIP, "host_free: not a pylda_host_alloc pointer");
    return PYLDA_OK;
}

int pylda_model_checkpoint(pylda_ctx* ctx, int restore)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)ctx->K * ctx->V * sizeof(double);
    if (!restore) {
        if (!ctx->have_eta) return fail(ctx, PYLDA_ERR_STATE, "model_checkpoint: eta was never set");
        if (!ctx->d_eta_ckpt) {
            const int rc = dev_alloc(ctx, &ctx->d_eta_ckpt, (size_t)ctx->K * ctx->V);
            if (rc != PYLDA_OK) return rc;
        }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_eta_ckpt, ctx->d_eta, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        if (!ctx->d_eta_ckpt) return fail(ctx, PYLDA_ERR_STATE, "model_checkpoint: nothing was saved");
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_eta, ctx->d_eta_ckpt, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        ctx->have_eta = true;
    }
    return PYLDA_OK;
}

int pylda_mark_time(pylda_ctx* ctx, int slot)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (slot < 0 || slot >= 4) return fail(ctx, PYLDA_ERR_INVALID, "mark_time: slot %d", slot);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->mark_event[slot]) HIP_TRY(ctx, hipEventCreate(&ctx->mark_event[slot]));
    HIP_TRY(ctx, hipEventRecord(ctx->mark_event[slot], ctx->stream));
    return PYLDA_OK;
}

int pylda_elapsed_ms(pylda_ctx* ctx, int slot_from, int slot_to, double* ms)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (slot_from < 0 || slot_from >= 4 || slot_to < 0 || slot_to >= 4 || !ms || !ctx->mark_event[slot_from] || !ctx->mark_event[slot_to])
        return fail(ctx, PYLDA_ERR_INVALID, "elapsed_ms: slots %d, %d", slot_from, slot_to);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->mark_event[slot_to]));
    float f = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&f, ctx->mark_event[slot_from], ctx->mark_event[slot_to]));
    *ms = f;
    return PYLDA_OK;
}

int pylda_work_counters(pylda_ctx* ctx, double* inner_iterations, double* inner_iteration_terms)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    double w[2] = {0.0, 0.0};
    HIP_TRY(ctx, hipMemcpyAsync(w, ctx->d_work, sizeof w, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_work, 0, sizeof w, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (inner_iterations) *inner_iterations = w[0];
    if (inner_iteration_terms) *inner_iteration_terms = w[1];
    return PYLDA_OK;
}

int pylda_comm_unique_id(void* id_out)
{
    if (!id_out) return fail(nullptr, PYLDA_ERR_INVALID, "comm_unique_id: NULL");
    std::string err;
    const int rc = pylda::comm_unique_id(id_out, &err);
    return rc == PYLDA_OK ? rc : fail(nullptr, rc, "comm_unique_id: %s", err.c_str());
}

int pylda_comm_init(pylda_ctx* ctx, const void* id, int rank, int world_size)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!id || world_size < 1 || rank < 0 || rank >= world_size)
        return fail(ctx, PYLDA_ERR_INVALID, "comm_init: rank %d of %d", rank, world_size);
    if (ctx->comm) return fail(ctx, PYLDA_ERR_STATE, "comm_init: the context already has a communicator");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    const int rc = pylda::comm_init(&ctx->comm, id, rank, world_size, &err);
    if (rc != PYLDA_OK) return fail(ctx, rc, "comm_init: %s", err.c_str());
    ctx->comm_world = world_size;
    return PYLDA_OK;
}

int pylda_comm_destroy(pylda_ctx* ctx)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->comm) pylda::comm_destroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 1;
    return PYLDA_OK;
}

int pylda_allreduce_sstats(pylda_ctx* ctx)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!ctx->comm) return fail(ctx, PYLDA_ERR_STATE, "allreduce_sstats: pylda_comm_init has not been called");
    if (!ctx->have_sstats) return fail(ctx, PYLDA_ERR_STATE, "allreduce_sstats: no training-mode E-step has run");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    // on the context's stream: ordered behind the E-step's kernels and before the M-step's
    const int rc = pylda::comm_allreduce_sum_f64(ctx->comm, ctx->d_sstats, (size_t)ctx->V * ctx->ldk, ctx->stream, &err);
    return rc == PYLDA_OK ? rc : fail(ctx, rc, "allreduce_sstats: %s", err.c_str());
}

int pylda_allreduce_doubles(pylda_ctx* ctx, double* values, int64_t n)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!ctx->comm) return fail(ctx, PYLDA_ERR_STATE, "allreduce_doubles: pylda_comm_init has not been called");
    if (n < 0 || (n > 0 && !values)) return fail(ctx, PYLDA_ERR_INVALID, "allreduce_doubles: bad argument");
    if (n == 0) return PYLDA_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->comm_small_cap < (size_t)n) {
        dev_free(ctx->d_comm_small);
        ctx->comm_small_cap = 0;
        const int rc = dev_alloc(ctx, &ctx->d_comm_small, (size_t)n);
        if (rc != PYLDA_OK) return rc;
        ctx->comm_small_cap = (size_t)n;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_comm_small, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    std::string err;
    const int rc = pylda::comm_allreduce_sum_f64(ctx->comm, ctx->d_comm_small, (size_t)n, ctx->stream, &err);
    if (rc != PYLDA_OK) return fail(ctx, rc, "allreduce_doubles: %s", err.c_str());
    HIP_TRY(ctx, hipMemcpyAsync(values, ctx->d_comm_small, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PYLDA_OK;
}

int pylda_set_profiling(pylda_ctx* ctx, int enabled)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    ctx->profiling = enabled != 0;
    return PYLDA_OK;
}

int pylda_kernel_time(pylda_ctx* ctx, double* doc_kernel_ms, double* sstats_kernel_ms, int64_t* estep_calls)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    drain_events(ctx);
    if (doc_kernel_ms) *doc_kernel_ms = ctx->doc_kernel_ms;
    if (sstats_kernel_ms) *sstats_kernel_ms = ctx->sstats_kernel_ms;
    if (estep_calls) *estep_calls = ctx->estep_calls;
    ctx->doc_kernel_ms = ctx->sstats_kernel_ms = 0.0;
    ctx->estep_calls = 0;
    return PYLDA_OK;
}

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

namespace {
__global__ void special_test_kernel(const double* x, int64_t n, double* dg, double* lg)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        dg[i] = pylda::digamma(x[i]);
        lg[i] = pylda::lgamma_pos(x[i]);
    }
}
__global__ void expdigamma_test_kernel(const double* x, int64_t n, double c, double* out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = c > 1e3 ? pylda::exp_digamma_minus_levels(x[i], c - 2e3) : pylda::exp_digamma_minus(x[i], c);
}
}  // namespace

int pylda_test_special(pylda_ctx* ctx, int64_t n, const double* x, double* digamma_out, double* lgamma_out)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (n < 0 || !x || !digamma_out || !lgamma_out) return fail(ctx, PYLDA_ERR_INVALID, "test_special: bad argument");
    if (n == 0) return PYLDA_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    double *dx = nullptr, *dd = nullptr, *dl = nullptr;
    int rc = dev_alloc(ctx, &dx, (size_t)n);
    if (rc == PYLDA_OK) rc = dev_alloc(ctx, &dd, (size_t)n);
    if (rc == PYLDA_OK) rc = dev_alloc(ctx, &dl, (size_t)n);
    if (rc == PYLDA_OK) {
        hipError_t e = hipMemcpy(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(special_test_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dx, n, dd, dl);
            e = hipStreamSynchronize(ctx->stream);
        }
        if (e == hipSuccess) e = hipMemcpy(digamma_out, dd, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(lgamma_out, dl, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(ctx, PYLDA_ERR_HIP, "test_special: %s", hipGetErrorString(e));
    }
    dev_free(dx); dev_free(dd); dev_free(dl);
    return rc;
}

int pylda_test_alpha_update(pylda_ctx* ctx, const double* alpha_k, const double* alpha_ss_k, double number_of_documents,
                            int hyper_parameter_iteration, double hyper_parameter_decay_factor, int hyper_parameter_maximum_decay,
                            double hyper_parameter_converge_threshold, double* alpha_out_k)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!alpha_k || !alpha_ss_k || !alpha_out_k || hyper_parameter_iteration < 1 || hyper_parameter_maximum_decay < 0 ||
        hyper_parameter_maximum_decay > 16)
        return fail(ctx, PYLDA_ERR_INVALID, "test_alpha_update: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int K = ctx->K;
    double* d = nullptr;                       // [alpha io (K) | statistics (K) | #documents | scratch alpha out (K)]
    int rc = dev_alloc(ctx, &d, (size_t)3 * K + 1);
    if (rc != PYLDA_OK) return rc;
    NewtonParams np;
    np.iterations = hyper_parameter_iteration;
    np.maximum_decay = hyper_parameter_maximum_decay;
    np.threshold = hyper_parameter_converge_threshold;
    for (int i = 0; i <= 16; ++i) np.decay_power[i] = std::pow(hyper_parameter_decay_factor, (double)i);
    hipError_t e = hipMemcpy(d, alpha_k, (size_t)K * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + K, alpha_ss_k, (size_t)K * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + 2 * (size_t)K, &number_of_documents, sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(alpha_newton_kernel, dim3(1), dim3(1024), 0, ctx->stream, d, d + K, d + 2 * (size_t)K, K, np,
                           ctx->d_newton_work, d + 2 * (size_t)K + 1);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(alpha_out_k, d, (size_t)K * sizeof(double), hipMemcpyDeviceToHost);
    dev_free(d);
    if (e != hipSuccess) return fail(ctx, PYLDA_ERR_HIP, "test_alpha_update: %s", hipGetErrorString(e));
    return PYLDA_OK;
}

int pylda_test_expdigamma(pylda_ctx* ctx, int64_t n, const double* x, double c, double* out)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (n < 0 || !x || !out) return fail(ctx, PYLDA_ERR_INVALID, "test_expdigamma: bad argument");
    if (n == 0) return PYLDA_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    double *dx = nullptr, *dout = nullptr;
    int rc = dev_alloc(ctx, &dx, (size_t)n);
    if (rc == PYLDA_OK) rc = dev_alloc(ctx, &dout, (size_t)n);
    if (rc == PYLDA_OK) {
        hipError_t e = hipMemcpy(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(expdigamma_test_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dx, n, c, dout);
            e = hipStreamSynchronize(ctx->stream);
        }
        if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(ctx, PYLDA_ERR_HIP, "test_expdigamma: %s", hipGetErrorString(e));
    }
    dev_free(dx); dev_free(dout);
    return rc;
}

}  // extern "C"
