// Shared device helpers and launch parameter blocks for the E-step kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pylda {

constexpr int kWave = 64;   // CDNA4 wavefront width
typedef double f64x2 __attribute__((ext_vector_type(2)));

// Everything one E-step launch needs.  Tables are WORD-MAJOR (V x K): the
// reference's E_log_eta[:, ids] column gather (variational_bayes.py:177)
// becomes a gather of contiguous K-double rows, i.e. coalesced HBM reads.
struct EstepParams {
    int K;
    int V;
    int ldk;                  // row stride of the word-major tables (K rounded up; padding is 0)
    const double* expElog;    // V x ldk : exp(E_log_eta[k][w] - shift[w])
    const double* expElog_elog; // V x ldk : expElog * (E_log_eta - shift)   ("B log B", entropy term)
    const double* shift;      // V     : max_k E_log_eta[k][w]
    const double* topic_lse;  // K     : logsumexp_v E_log_eta[k][:]  (held-out only, :155)
    const double* alpha;      // K
    double alpha_term;        // lnG(sum alpha) - sum lnG(alpha)      (:195)
    double alpha_sum, alpha_lgamma_sum;   // sum_k alpha_k, sum_k lnG(alpha_k)
    const int64_t* doc_ptr;   // D+1
    const int32_t* term_id;   // nnz
    const int32_t* term_ct;   // nnz
    const int32_t* order;     // document ids of this launch, longest first
    int max_iter;             // local_parameter_iteration            (:132)
    double tol;               // local_parameter_converge_threshold   (:132)
    int heldout;              // 1: parsed_corpus given               (:133-138)
    int want_doc_ll;          // 1: doc_ll[] holds every term of :195-199.  0 (training fast path): the
                              //    sum_n c_n sum_k phi log B term is left out of doc_ll[] and taken once
                              //    per corpus from the sufficient statistics (sstats_finalize_kernel)
    double* gamma;            // D x K out                            (:188,:213)
    double* doc_ll;           // D out: this document's terms of :195-199
    double* doc_words_ll;     // D out: this document's term of :204
    int32_t* iters;           // D out: inner iterations executed
    double* tfinal;           // D x ldk out: t[k] of the last executed iteration (training)
    double* rfinal;           // nnz out: count / normaliser of the last executed iteration
    int32_t* status;          // D out: 0 ok, 1 = linear-space normaliser under/overflowed
    int n_cap;                // max distinct terms of any document in this launch
    int tile_stride;          // LDS row stride in doubles (odd)
    double* term_scratch;     // nnz scratch doubles, only for documents too long for the LDS (estep_generic.h MODE 2)
    // ---- hand-over to the live-topic kernel (estep_compact.h) ----
    int handoff_live;         // quad kernel (one lane shape per launch class): its documents leave at this many live topics (0: never)
    int handoff_on;           // 1: a document of N terms leaves the dense kernel once at most handoff_caps[ceil(N / 64)] topics have
    int handoff_caps[9];      //    gamma_k != alpha_k (0: never) - the columns the live-topic kernel's register tile holds at that many
                              //    term slots per lane
    int tile_from_table;      // live-topic kernel: 1: the launch's documents have no tile in live_tile - gather it from the table
    int32_t* live_n;          // D: live topics of a document handed over (status 4); after the E-step: entries of its list of t
                              //    (live_stats), -1: the document's t is the dense row tfinal[d]
    char* live_list;          // D x kLiveListBytes: a document's live topics and their t of the last iteration (layout below)
    int live_stats;           // 1: the statistics pass reads the lists (sstats_live.h): the live-topic kernel writes no dense row
    double* live_tile;        // the document's compact tile: value of term n, live topic j at live_tile[tile_ptr[d] + j * N_d + n]
    const int64_t* tile_ptr;  // D offsets into live_tile (doubles)
    double alpha_min;         // over the K topics (the exactness guard of the live-topic kernel)
    const double* alpha_sgn;  // K: alpha with the sign bit set where the topic never counts as dead (kMortalT; alpha_mortality_kernel)
    int32_t* handoff_it;      // D out: inner iterations the dense kernel ran before it handed the document over (else left at -1)
    int32_t* col_iters;       // D out: sum over the live-topic kernel's iterations of the tile columns it ran them on
    double* clock_acc;        // profiling (else NULL): [shader-clock ticks, constant-rate ticks] of sampled kernel spans, accumulated
};

// the live-topic count at which a document of N terms leaves a dense kernel (-1: never)
__device__ __forceinline__ int handoff_threshold(const EstepParams& p, int N)
{
    const int slots = (N + kWave - 1) / kWave;
    const int cap = p.handoff_on && slots <= 8 ? p.handoff_caps[slots > 0 ? slots : 1] : 0;
    return cap > 0 ? cap : -1;
}

// A topic may only ever count as DEAD (gamma_k == alpha_k bitwise, dropped from the live-topic kernel's tile) if its
// t_k = exp(psi(alpha_k) - psi(sum gamma)) is below this for a document of kMortalTokens tokens.  For a topic with a
// larger alpha_k - the alpha update of a trained model pushes the used topics' alpha beyond 0.01 - gamma_k can equal
// alpha_k bitwise in a document that does not use it (its B are tiny there) while t_k is only ~1e-20: nothing bounds
// its share of a normaliser below rounding then, and the guard of estep_compact.h (K t_dead < 2^-60 of the smallest
// normaliser, t_dead N r_max < 2^-54 alpha_min - evaluated per document, with the document's own psi(sum gamma)) would
// send every such document to the log-space kernel.  Such a topic stays a live column for good instead: the kernels
// hold its alpha with the sign bit set (EstepParams::alpha_sgn), so that `gamma != alpha` holds for it always (the
// arithmetic takes |alpha|).  With 1e-33 the guard passes for normalisers above ~1e-14 and r_n below ~1e12 - a document
// outside that (or much shorter than kMortalTokens) is what the guard and the log-space kernel are for.
constexpr double kMortalT = 1e-33;
constexpr double kMortalTokens = 8.0;

constexpr int kLiveStride = 60;   // entries per document's list: the largest live set the live-topic kernel takes over
// A document's list of live topics (EstepParams::live_list), 640 bytes = five 128-byte lines of twelve entries each:
//     line l        t[12 l .. 12 l + 12) (96 bytes), then topic[12 l .. 12 l + 12) as uint16 (24 bytes)
// The statistics pass reads a document's list once per posting, at random over the corpus: a document that finishes with
// at most twelve live topics costs it ONE line (cfg 4 after three outer iterations: nineteen documents in twenty), up to
// twenty-four TWO (the rule from outer iteration five on: the trained model's documents use more topics; with the tail
// of the list split into a t part and a topic part they cost three, and the pass 17-19 ms instead of 12.8).
constexpr int kLiveHead = 12;
constexpr int kLiveListBytes = 640;
__device__ __forceinline__ double* live_t_at(char* list, int j)
{
    return reinterpret_cast<double*>(list + 128 * (j / kLiveHead) + 8 * (j % kLiveHead));
}
__device__ __forceinline__ uint16_t* live_idx_at(char* list, int j)
{
    return reinterpret_cast<uint16_t*>(list + 128 * (j / kLiveHead) + 96 + 2 * (j % kLiveHead));
}
__device__ __forceinline__ char* live_list_of(char* live_list, int64_t doc) { return live_list + doc * kLiveListBytes; }
static_assert(kLiveStride % kLiveHead == 0 && 128 * (kLiveStride / kLiveHead) == kLiveListBytes && 10 * kLiveHead <= 128, "list layout");

// Sum over the 64 lanes, result in every lane, without the LDS crossbar (ds_bpermute costs an LDS
// round trip per level): four DPP levels inside each 16-lane row, then one permlane16 and one
// permlane32 swap level across the rows.  Fixed order, so results are run-to-run identical.
__device__ __forceinline__ double wave_sum(double v)
{
#define PYLDA_DPP_ADD(CTRL)                                                                     \
    v += __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false),   \
                          __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false))
    PYLDA_DPP_ADD(0xB1);        // quad_perm [1,0,3,2]
    PYLDA_DPP_ADD(0x4E);        // quad_perm [2,3,0,1]
    PYLDA_DPP_ADD(0x141);       // row_half_mirror
    PYLDA_DPP_ADD(0x140);       // row_mirror
#undef PYLDA_DPP_ADD
    {
        const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
        v = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);       // rows 0+1 | 0+1 | 2+3 | 2+3
    }
    {
        const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
        v = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);       // both halves
    }
    return v;
}

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int m = kWave / 2; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, kWave));
    return v;
}

// Deterministic block-wide reductions; every thread gets the result.
// `scratch` holds NT/64 doubles in LDS.
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* scratch)
{
    v = wave_sum(v);
    if constexpr (NT == kWave) return v;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NT / kWave; ++w) s += scratch[w];
    return s;
}

template <int NT>
__device__ __forceinline__ double block_max(double v, double* scratch)
{
    v = wave_max(v);
    if constexpr (NT == kWave) return v;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double s = scratch[0];
#pragma unroll
    for (int w = 1; w < NT / kWave; ++w) s = fmax(s, scratch[w]);
    return s;
}

// Exchange through LDS between the lanes of ONE wavefront.  The LDS executes a wavefront's
// DS instructions in issue order, so a later read sees an earlier write of another lane of the
// same wavefront without any wait; what is needed is only that the compiler keeps the two in
// program order.  (A workgroup-scope release/acquire fence pair does that too, but costs
// ~40-116 cycles each on gfx950 - MI355X_MICROARCH.md - twice per exchange, on the serial path.)
__device__ __forceinline__ void wave_lds_exchange()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// Workgroup barrier that orders LDS traffic only: outstanding global loads (a tile gather in
// flight) are NOT waited for, unlike __syncthreads(), whose release fence drains vmcnt.
__device__ __forceinline__ void lds_only_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- cross-lane helpers shared by the register-resident kernels ----
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// a value known to be the same in every lane, moved to scalar registers
__device__ __forceinline__ double uniform_f64(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// a' = [a.lo32 | b.lo32], b' = [a.hi32 | b.hi32] (halves of the wavefront); returns a' + b'.
__device__ __forceinline__ double swap32_add(double a, double b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// the same with 16-lane rows: swaps odd rows of a with even rows of b.
__device__ __forceinline__ double swap16_add(double a, double b)
{
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a constant expression
template <int I>
struct StaticIndex {
    static constexpr int value = I;
};
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(StaticIndex<I>());
        static_for<N, I + 1>(f);
    }
}

// Lane permutation inside a row of 16 lanes (DPP), on both halves of a double.  CTRL: quad_perm
// (0x00-0xFF), row_half_mirror 0x141, row_newbcast:L 0x150+L (lane L of each row to all 16 lanes).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int L>
__device__ __forceinline__ double row_bcast(double v) { return dpp_f64<0x150 + L>(v); }

// out[i] = v of lane i*STEP of the caller's row, i < N
template <int N, int STEP, int I = 0>
__device__ __forceinline__ void row_bcast_all(double v, double* out)
{
    if constexpr (I < N) {
        out[I] = row_bcast<I * STEP>(v);
        row_bcast_all<N, STEP, I + 1>(v, out);
    }
}

// q[j] += r(lane L of the caller's row) * b[j]: the row broadcast rides on the FMA's DPP operand
// (DP-ALU DPP supports row_newbcast; v_fmac_f64 has a DPP form, v_mul_f64 does not), no separate
// v_mov_dpp pair per word.  The DPP read of r needs two wait states behind the VALU write of r and
// the compiler's hazard recogniser does not look inside inline assembly: callers put other VALU
// work (row_bcast_matvec: the first word's plain multiplies) between the two.
// A value about to be read through the DPP operand of an asm FMA block: two wait states behind its
// (compiler-generated) VALU write, wherever the scheduler ends up placing that write.  Found the hard
// way: with the word count arriving late from memory the compiler sank `r = count * reciprocal` to
// right in front of the first v_fmac_f64_dpp, which then read the previous iteration's low half
// (errors of 1e-8 in the topics of the block's first chain only).
__device__ __forceinline__ void dpp_source_ready(double& r) { asm volatile("s_nop 1" : "+v"(r)); }

#define PYLDA_DPP_ROW(L) " row_newbcast:" #L " row_mask:0xf bank_mask:0xf\n\t"
template <int L>
__device__ __forceinline__ void row_bcast_fmac(double (&q)[8], double r, const double (&b)[8])
{
    static_assert(L >= 0 && L < 16, "lane of the row");
#define PYLDA_ROW8(LANE)                                                                                     \
    "v_fmac_f64_dpp %0, %8, %9" PYLDA_DPP_ROW(LANE) "v_fmac_f64_dpp %1, %8, %10" PYLDA_DPP_ROW(LANE)          \
    "v_fmac_f64_dpp %2, %8, %11" PYLDA_DPP_ROW(LANE) "v_fmac_f64_dpp %3, %8, %12" PYLDA_DPP_ROW(LANE)         \
    "v_fmac_f64_dpp %4, %8, %13" PYLDA_DPP_ROW(LANE) "v_fmac_f64_dpp %5, %8, %14" PYLDA_DPP_ROW(LANE)         \
    "v_fmac_f64_dpp %6, %8, %15" PYLDA_DPP_ROW(LANE) "v_fmac_f64_dpp %7, %8, %16" PYLDA_DPP_ROW(LANE)
#define PYLDA_ROW8_CASE(LANE)                                                                                \
    if constexpr (L == LANE)                                                                                 \
        asm(PYLDA_ROW8(LANE)                                                                                 \
            : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) \
            : "v"(r), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
    PYLDA_ROW8_CASE(0) PYLDA_ROW8_CASE(1) PYLDA_ROW8_CASE(2) PYLDA_ROW8_CASE(3) PYLDA_ROW8_CASE(4) PYLDA_ROW8_CASE(5)
    PYLDA_ROW8_CASE(6) PYLDA_ROW8_CASE(7) PYLDA_ROW8_CASE(8) PYLDA_ROW8_CASE(9) PYLDA_ROW8_CASE(10) PYLDA_ROW8_CASE(11)
    PYLDA_ROW8_CASE(12) PYLDA_ROW8_CASE(13) PYLDA_ROW8_CASE(14) PYLDA_ROW8_CASE(15)
#undef PYLDA_ROW8_CASE
#undef PYLDA_ROW8
}

template <int L>
__device__ __forceinline__ void row_bcast_fmac(double (&q)[4], double r, const double (&b)[4])
{
    static_assert(L >= 0 && L < 16, "lane of the row");
#define PYLDA_ROW4(LANE)                                                                                     \
    "v_fmac_f64_dpp %0, %4, %5" PYLDA_DPP_ROW(LANE) "v_fmac_f64_dpp %1, %4, %6" PYLDA_DPP_ROW(LANE)           \
    "v_fmac_f64_dpp %2, %4, %7" PYLDA_DPP_ROW(LANE) "v_fmac_f64_dpp %3, %4, %8" PYLDA_DPP_ROW(LANE)
#define PYLDA_ROW4_CASE(LANE)                                                                                \
    if constexpr (L == LANE)                                                                                 \
        asm(PYLDA_ROW4(LANE)                                                                                 \
            : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3])                                                 \
            : "v"(r), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    PYLDA_ROW4_CASE(0) PYLDA_ROW4_CASE(1) PYLDA_ROW4_CASE(2) PYLDA_ROW4_CASE(3) PYLDA_ROW4_CASE(4) PYLDA_ROW4_CASE(5)
    PYLDA_ROW4_CASE(6) PYLDA_ROW4_CASE(7) PYLDA_ROW4_CASE(8) PYLDA_ROW4_CASE(9) PYLDA_ROW4_CASE(10) PYLDA_ROW4_CASE(11)
    PYLDA_ROW4_CASE(12) PYLDA_ROW4_CASE(13) PYLDA_ROW4_CASE(14) PYLDA_ROW4_CASE(15)
#undef PYLDA_ROW4_CASE
#undef PYLDA_ROW4
}

// other row lengths: broadcast first, plain FMAs
template <int L, int KRL>
__device__ __forceinline__ void row_bcast_fmac(double (&q)[KRL], double r, const double (&b)[KRL])
{
    const double rb = row_bcast<L>(r);
#pragma unroll
    for (int j = 0; j < KRL; ++j) q[j] = fma(rb, b[j], q[j]);
}

// q[j] = sum_i r(lane i*STEP of the row) * B[i][j]
template <int RWL, int STEP, int KRL, int I = 0>
__device__ __forceinline__ void row_bcast_matvec(double (&q)[KRL], double r, const double (&B)[RWL][KRL])
{
    if constexpr (I == 0) {
        const double r0 = row_bcast<0>(r);
#pragma unroll
        for (int j = 0; j < KRL; ++j) q[j] = r0 * B[0][j];
        row_bcast_matvec<RWL, STEP, KRL, 1>(q, r, B);
    } else if constexpr (I < RWL) {
        row_bcast_fmac<I * STEP>(q, r, B[I]);
        row_bcast_matvec<RWL, STEP, KRL, I + 1>(q, r, B);
    }
}

// ---- hand-ordered FMA blocks -------------------------------------------------------------------
// hipcc schedules these kernels for register pressure (they sit at the 256-VGPR limit) and then
// interleaves only TWO accumulator chains; a lone wavefront issues a 2-chain fp64 stream at 6.4
// ticks per instruction against 4.9 for 8 chains (tools/valu_bench.hip).  An asm statement is
// scheduled as a unit, so the blocks below fix the order: eight independent chains per block.

// a[i] = b_i * t  /  a[i] += b_i * t   for the eight words a lane holds at one topic
__device__ __forceinline__ void col_mul8(double (&a)[8], double b0, double b1, double b2, double b3, double b4, double b5,
                                         double b6, double b7, double t)
{
    asm("v_mul_f64 %0, %8, %16\n\tv_mul_f64 %1, %9, %16\n\tv_mul_f64 %2, %10, %16\n\tv_mul_f64 %3, %11, %16\n\t"
        "v_mul_f64 %4, %12, %16\n\tv_mul_f64 %5, %13, %16\n\tv_mul_f64 %6, %14, %16\n\tv_mul_f64 %7, %15, %16"
        : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7])
        : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7), "v"(t));
}
__device__ __forceinline__ void col_fmac8(double (&a)[8], double b0, double b1, double b2, double b3, double b4, double b5,
                                          double b6, double b7, double t)
{
    asm("v_fmac_f64_e32 %0, %8, %16\n\tv_fmac_f64_e32 %1, %9, %16\n\tv_fmac_f64_e32 %2, %10, %16\n\t"
        "v_fmac_f64_e32 %3, %11, %16\n\tv_fmac_f64_e32 %4, %12, %16\n\tv_fmac_f64_e32 %5, %13, %16\n\t"
        "v_fmac_f64_e32 %6, %14, %16\n\tv_fmac_f64_e32 %7, %15, %16"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
        : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7), "v"(t));
}

// sum_j row[j] * t[j] over a lane's eight topics of ONE word: four chains of two, then a tree
__device__ __forceinline__ double dot8(const double (&row)[8], const double (&t)[8])
{
    double p0, p1, p2, p3;
    asm("v_mul_f64 %0, %4, %12\n\tv_mul_f64 %1, %6, %14\n\tv_mul_f64 %2, %8, %16\n\tv_mul_f64 %3, %10, %18\n\t"
        "v_fmac_f64_e32 %0, %5, %13\n\tv_fmac_f64_e32 %1, %7, %15\n\tv_fmac_f64_e32 %2, %9, %17\n\tv_fmac_f64_e32 %3, %11, %19\n\t"
        "v_add_f64 %0, %0, %1\n\tv_add_f64 %2, %2, %3\n\tv_add_f64 %0, %0, %2"
        : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
        : "v"(row[0]), "v"(row[1]), "v"(row[2]), "v"(row[3]), "v"(row[4]), "v"(row[5]), "v"(row[6]), "v"(row[7]),
          "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]));
    return p0;
}

// The same dot product as two chains of four and one add: 9 instructions instead of 11, dependency depth 5
// instead of 4.  With two documents per CU the kernel's time follows its instruction count (the co-resident
// document fills the longer chain); the one-document-per-CU kernels keep dot8.
__device__ __forceinline__ double dot8_two_chains(const double (&row)[8], const double (&t)[8])
{
    double p0, p1;
    asm("v_mul_f64 %0, %2, %10\n\tv_mul_f64 %1, %3, %11\n\t"
        "v_fmac_f64_e32 %0, %4, %12\n\tv_fmac_f64_e32 %1, %5, %13\n\t"
        "v_fmac_f64_e32 %0, %6, %14\n\tv_fmac_f64_e32 %1, %7, %15\n\t"
        "v_fmac_f64_e32 %0, %8, %16\n\tv_fmac_f64_e32 %1, %9, %17\n\t"
        "v_add_f64 %0, %0, %1"
        : "=&v"(p0), "=&v"(p1)
        : "v"(row[0]), "v"(row[1]), "v"(row[2]), "v"(row[3]), "v"(row[4]), "v"(row[5]), "v"(row[6]), "v"(row[7]),
          "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]));
    return p0;
}

// An LDS row (a lane's eight topics of one word: four 16-byte pieces, 256 bytes apart) requested NOW
// and waited for LATER: the compiler would sink the reads to their first use (register pressure) and
// serialise the round trips.  lds_row_request issues the four ds_read_b128; the values may only be
// used after lds_row_wait, which hands them over as its outputs.  (A wavefront's LDS operations return
// in order, so the compiler's own lgkmcnt waits stay sufficient with these extra reads in flight.)
struct LdsRow {
    f64x2 p[4];
    // valid after lds_row_wait only
    __device__ __forceinline__ void unpack(double (&row)[8]) const
    {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            row[2 * jj] = p[jj].x;
            row[2 * jj + 1] = p[jj].y;
        }
    }
};
// PIECE: bytes between the four 16-byte pieces (16 bytes x topic lanes of a group: 256 or 512)
template <int PIECE>
__device__ __forceinline__ void lds_row_request(LdsRow& r, const void* lds_ptr)
{
    static_assert(PIECE == 256 || PIECE == 512, "16 or 32 topic lanes");
    typedef __attribute__((address_space(3))) const char* lds_cptr;
    const unsigned addr = (unsigned)(uintptr_t)(lds_cptr)lds_ptr;      // LDS byte offset
    if constexpr (PIECE == 256)
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:256\n\tds_read_b128 %2, %4 offset:512\n\t"
                     "ds_read_b128 %3, %4 offset:768"
                     : "=&v"(r.p[0]), "=&v"(r.p[1]), "=&v"(r.p[2]), "=&v"(r.p[3])
                     : "v"(addr)
                     : "memory");
    else
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:512\n\tds_read_b128 %2, %4 offset:1024\n\t"
                     "ds_read_b128 %3, %4 offset:1536"
                     : "=&v"(r.p[0]), "=&v"(r.p[1]), "=&v"(r.p[2]), "=&v"(r.p[3])
                     : "v"(addr)
                     : "memory");
}
__device__ __forceinline__ void lds_row_wait(LdsRow& r)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.p[0]), "+v"(r.p[1]), "+v"(r.p[2]), "+v"(r.p[3]) : : "memory");
}

// The same row shape fetched from the TABLE (an L2 hit after the document's first iteration): the four
// global_load_dwordx4 go out now, table_row_wait hands the values over.  The address is a uniform base (scalar
// registers) plus a 32-bit byte offset per lane - one VGPR per row instead of a pointer pair, in kernels at the
// register limit; the host only selects these kernels while the table is below 4 GiB.  The wait is vmcnt(0): the
// loops that use it keep one such row in flight per wavefront, and any older load of the thread has landed by then.
// The s_nop: a VMEM instruction needs five wait states behind a VALU write of its scalar base, and the compiler's
// hazard recogniser does not look inside the asm - when the base had been spilled to a vector register's lanes, its
// v_readlane reload sat right in front of the load (found as a memory fault of the streamed classes in round 6).
template <int PIECE>
__device__ __forceinline__ void table_row_request(LdsRow& r, const void* base, unsigned byte_offset)
{
    static_assert(PIECE == 128 || PIECE == 256 || PIECE == 512, "8, 16 or 32 topic lanes");
    if constexpr (PIECE == 128)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:128\n\t"
                     "global_load_dwordx4 %2, %4, %5 offset:256\n\tglobal_load_dwordx4 %3, %4, %5 offset:384"
                     : "=&v"(r.p[0]), "=&v"(r.p[1]), "=&v"(r.p[2]), "=&v"(r.p[3])
                     : "v"(byte_offset), "s"(base)
                     : "memory");
    else if constexpr (PIECE == 256)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:256\n\t"
                     "global_load_dwordx4 %2, %4, %5 offset:512\n\tglobal_load_dwordx4 %3, %4, %5 offset:768"
                     : "=&v"(r.p[0]), "=&v"(r.p[1]), "=&v"(r.p[2]), "=&v"(r.p[3])
                     : "v"(byte_offset), "s"(base)
                     : "memory");
    else
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:512\n\t"
                     "global_load_dwordx4 %2, %4, %5 offset:1024\n\tglobal_load_dwordx4 %3, %4, %5 offset:1536"
                     : "=&v"(r.p[0]), "=&v"(r.p[1]), "=&v"(r.p[2]), "=&v"(r.p[3])
                     : "v"(byte_offset), "s"(base)
                     : "memory");
}
// one double to uniform base + 32-bit byte offset (the store's address is ONE vector register)
__device__ __forceinline__ void store_f64_uniform_base(double* base, unsigned byte_offset, double v)
{
    asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2" : : "v"(byte_offset), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void table_row_wait(LdsRow& r)
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.p[0]), "+v"(r.p[1]), "+v"(r.p[2]), "+v"(r.p[3]) : : "memory");
}

// sum over aligned groups of LPW (1, 2, 4 or 8) neighbouring lanes; every lane of a group gets it
template <int LPW>
__device__ __forceinline__ double lane_group_sum(double s)
{
    if constexpr (LPW >= 2) s += dpp_f64<0xB1>(s);      // quad_perm [1,0,3,2]
    if constexpr (LPW >= 4) s += dpp_f64<0x4E>(s);      // quad_perm [2,3,0,1]
    if constexpr (LPW >= 8) s += dpp_f64<0x141>(s);     // row_half_mirror: quad 0 <-> quad 1
    return s;
}

constexpr double kChangeScale = 1099511627776.0;   // 2^40 fixed point for sum_k |delta gamma_k|

// |delta gamma| (clipped to 1024) as a 2^40 fixed-point integer, rounded to nearest: adding 2^52
// leaves the integer in the low mantissa bits (one FMA and one AND instead of a 64-bit float -> int
// conversion sequence on the serial path).
__device__ __forceinline__ unsigned long long change_fixed(double diff)
{
    const double biased = fma(fmin(diff, 1024.0), kChangeScale, 4503599627370496.0);
    return (unsigned long long)__double_as_longlong(biased) & 0x000fffffffffffffull;
}

// the same with the 2^52 bias supplied in a (resident) vector register: one VOP3 FMA
__device__ __forceinline__ unsigned long long change_fixed(double diff, double bias52)
{
    double biased;
    const double clipped = fmin(diff, 1024.0);
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(biased) : "v"(clipped), "s"(kChangeScale), "v"(bias52));
    return (unsigned long long)__double_as_longlong(biased) & 0x000fffffffffffffull;
}

// *p = hi << 32, with the constant created HERE: a copy hoisted out of the inner loop of a kernel at the
// register limit is spilled to scratch and reloaded on the serial path (seen: scratch_load + vmcnt(0)
// in front of a one-lane LDS store).
__device__ __forceinline__ void store_u64_hi(unsigned long long* p, unsigned hi)
{
    unsigned lo = 0;
    asm volatile("" : "+v"(hi), "+v"(lo));
    *p = ((unsigned long long)hi << 32) | lo;
}

// Forces the N values to be live in registers at this point (loads that produce them are all
// issued before it; the scheduler otherwise recycles two registers and serialises the round trips).
template <int N>
__device__ __forceinline__ void keep_together(double (&v)[N])
{
    if constexpr (N == 8)
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
    else if constexpr (N == 4)
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
    else
        for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

__device__ __forceinline__ size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

}  // namespace pylda
