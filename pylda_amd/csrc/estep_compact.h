// Live-topic E-step kernel: the inner loop of variational_bayes.py:174-190 on the topics of a document that still move.
//
// Why.  With alpha_k ~ 1/K a topic the document does not use decays super-exponentially (exp(psi(gamma)) ~ exp(-1/gamma)):
// after a handful of iterations  gamma'_k = fma(t_k, S_k, alpha_k)  rounds to alpha_k BITWISE, and from then on
// t_k = exp(psi(alpha_k) - psi(sum gamma)) ~ 1e-114 (K = 256) is a constant that adds less than 2^-60 relative to any
// normaliser and nothing to its own gamma: the topic is dead, exactly.  Measured inside the bench's timed window
// (tools/live_probe.py, cfg 4): 256 live topics through iteration 4, 141 at 6, 58 at 8, 33 at 10, 23 at 12, 16 at 15,
// 9 at 30, 7 at 50 - nine tenths of the dense N x K tile a document kernel streams through the fp64 pipes 50 times are
// columns that cannot change a bit of the result.  The dense kernels (estep_quad.h) therefore hand a document over
// once at most `handoff_live` topics are alive: gamma, the live topics' indices and their columns of the tile.
//
// Here ONE wavefront runs the remaining iterations of a document on its N x L tile, eight documents per CU (the dense
// kernel at K = 256: one):
//   * a lane owns terms n = lane + 64 s, s < S, with their L tile values in registers: the normalisers are lane-local
//     FMAs with t_j as a SCALAR operand (no cross-lane step at all in the pass that has N outputs);
//   * the topic sums q_j = sum_n r_n C[n][j] are L values reduced over the 64 lanes by a reduce-scatter - each of the six
//     exchange levels halves the values a lane carries (permlane32/16 swaps, then DPP inside the 16-lane rows) - after
//     which lane l owns column j(l): gamma update, exp(psi(gamma) - psi(sum gamma)), |delta gamma| on L lanes at once;
//   * t_j goes back to scalar registers by v_readlane; no LDS, no barrier inside the loop.
// When the live set has shrunk to the next smaller instantiation the wavefront reloads the surviving columns (an L2 hit:
// it wrote or read them microseconds ago) and goes on with fewer.
//
// Exactness.  Same arithmetic as the dense kernels on the live topics (the same exp_digamma_minus_levels, the same
// 2^-40 fixed-point stop sum - a dead topic's |delta gamma| is exactly 0), another summation ORDER inside normalisers and
// topic sums: results differ from the dense kernel's by rounding (<= a few ulp), from the oracle by the same 1e-13 as
// before.  What is dropped - the dead topics' B t ~ 1e-114 in the normalisers - is guarded per document from live
// quantities only: with t_dead = exp(psi(max alpha over the dead topics) - psi(sum gamma)) >= every dead topic's t,
//     K t_dead             <  2^-60 min_n normaliser_n      (a normaliser moves by < 2^-60 relative)
//     t_dead N max_n r_n   <  2^-54 alpha_min               (fma(t_k, S_k, alpha_k) still rounds to alpha_k: dead stays dead)
// over ALL iterations run here (a topic whose t at gamma = alpha is not below kMortalT never counts as dead in the first
// place - estep_common.h - so the bound holds by a wide margin whenever the normalisers are sane);
// a document that fails either is flagged (status 1) and redone by the log-space kernel,
// the reference's own formulation - like a document whose normaliser leaves the fp64 range.
#pragma once
#include "estep_common.h"
#include "special_device.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace pylda {

// ---- the reduce-scatter over the 64 lanes --------------------------------------------------------------------------
// Level k exchanges between lanes that differ in bit 5 - k.  Of R values a lane keeps ceil(R / 2): the lanes with the
// bit clear the first half, the others the second (index + half; beyond R: padding, carried as zero); R == 1: both keep
// the sum (replicas).  After six levels one value is left: lane l holds column sum_k bit_k(l) half_k.
constexpr int compact_half(int r) { return (r + 1) / 2; }
constexpr int compact_count_at(int lt, int level)       // values per lane entering `level`
{
    int r = lt;
    for (int k = 0; k < level; ++k) r = r > 1 ? compact_half(r) : 1;
    return r;
}
// column of lane `lane`, or -1 (padding); `primary`: the replica with the lowest lane number
constexpr int compact_column_of(int lt, int lane, bool* primary)
{
    int m = 0;
    bool valid = true, first = true;
    for (int k = 5; k >= 0; --k) {                      // from the last level back to the first
        const int r = compact_count_at(lt, k), bit = (lane >> (5 - k)) & 1;
        if (r == 1) {
            if (bit) first = false;
            continue;
        }
        const int half = compact_half(r);
        if (bit) {
            m += half;
            if (m >= r) valid = false;
        }
    }
    if (primary) *primary = valid && first;
    return valid ? m : -1;
}
constexpr int compact_lane_of(int lt, int column)
{
    for (int lane = 0; lane < 64; ++lane) {
        bool primary = false;
        if (compact_column_of(lt, lane, &primary) == column && primary) return lane;
    }
    return -1;
}

template <int CTRL>
__device__ __forceinline__ double compact_dpp_add(double keep, double send) { return keep + dpp_f64<CTRL>(send); }

// one level: x[0 .. R) -> y[0 .. ceil(R / 2))
template <int LEVEL, int R>
__device__ __forceinline__ void compact_level(const double (&x)[R], double (&y)[(R + 1) / 2 > 0 ? (R + 1) / 2 : 1], bool upper)
{
    constexpr int H = (R + 1) / 2;
    if constexpr (R == 1) {
        if constexpr (LEVEL == 0) y[0] = swap32_add(x[0], x[0]);
        else if constexpr (LEVEL == 1) y[0] = swap16_add(x[0], x[0]);
        else if constexpr (LEVEL == 2) y[0] = compact_dpp_add<0x140>(x[0], x[0]);       // row_mirror
        else if constexpr (LEVEL == 3) y[0] = compact_dpp_add<0x141>(x[0], x[0]);       // row_half_mirror
        else if constexpr (LEVEL == 4) y[0] = compact_dpp_add<0x4E>(x[0], x[0]);        // quad_perm [2,3,0,1]
        else y[0] = compact_dpp_add<0xB1>(x[0], x[0]);                                  // quad_perm [1,0,3,2]
    } else {
#pragma unroll
        for (int m = 0; m < H; ++m) {
            const double a = x[m], b = m + H < R ? x[m + H < R ? m + H : 0] : 0.0;
            if constexpr (LEVEL == 0) y[m] = swap32_add(a, b);
            else if constexpr (LEVEL == 1) y[m] = swap16_add(a, b);
            else {
                const double keep = upper ? b : a, send = upper ? a : b;
                if constexpr (LEVEL == 2) y[m] = compact_dpp_add<0x140>(keep, send);
                else if constexpr (LEVEL == 3) y[m] = compact_dpp_add<0x141>(keep, send);
                else if constexpr (LEVEL == 4) y[m] = compact_dpp_add<0x4E>(keep, send);
                else y[m] = compact_dpp_add<0xB1>(keep, send);
            }
        }
    }
}

// levels LEVEL .. 5 on x[0 .. R): the one value left
template <int LEVEL, int R>
__device__ __forceinline__ double compact_reduce_from(const double (&x)[R], int lane)
{
    double y[(R + 1) / 2 > 0 ? (R + 1) / 2 : 1];
    compact_level<LEVEL, R>(x, y, ((lane >> (5 - LEVEL)) & 1) != 0);
    if constexpr (LEVEL == 5) {
        static_assert(R <= 2, "six levels reduce at most 64 values");
        return y[0];
    } else {
        return compact_reduce_from<LEVEL + 1, (R + 1) / 2>(y, lane);
    }
}

// ---- LDS of a document's workgroup: the hand-over state between bodies, and the epilogue's row of t ----
struct CompactLds {
    double gam[kLiveStride];        // gamma of the live topics (column order)
    double gprev[kLiveStride];      // ... before the last update
    double tlast[kLiveStride];      // t of the last executed iteration
    double alf[kLiveStride];        // alpha of the live topics; sign bit: never counts as dead (kMortalT)
    int idx[kLiveStride];           // topic of column j
    int col[kLiveStride];           // its column in the document's tile in memory
    unsigned member[32];            // bit k: topic k is a column (K <= 1024)
};

struct CompactState {
    int L;                  // live columns
    int it, left;           // iterations executed / left
    int cols;               // sum over the iterations run here of the tile columns they ran on (work counter)
    int bad;                // a normaliser left the fp64 range, or the exactness guard failed
    double nrm_min, r_max;  // over the iterations run here (per lane; reduced at the end)
};

enum CompactExit { kCompactDone = 0, kCompactShrink = 1 };

// Where a lane's tile values come from: the columns the dense quad kernel wrote (live_tile, term-minor), or - documents of
// the streaming kernels, which keep no tile on chip to write - the table itself (N x L scattered reads, once per body).
// A slot beyond the document repeats the document's LAST term: its count is 0, so r = 0 there, and the extremes the
// exactness guard takes over the normalisers see nothing new.
template <int S>
struct CompactSource {
    const double* tile;     // the document's columns (uniform), or nullptr
    const double* table;    // expElog
    int ldk, N;
    int wid[S];             // table mode: term ids of this lane's slots
    unsigned at[S];         // tile mode: byte offset of this lane's slots within a column
};

// The register tile of a body: C[s][m] = value of term lane + 64 s in column STRIDE m + first.  Addresses are a scalar
// column base plus a 32-bit lane offset (tile mode) or a lane's row pointer plus a scalar topic (table mode): two or three
// instructions per element, every load in flight at once.  A column beyond L repeats column 0 - nobody owns it, its
// t_j is 0 (finite values are all that is asked of it).
template <int S, int LT, int STRIDE>
__device__ __forceinline__ void compact_load_tile(double (&C)[S][LT], const CompactSource<S>& src, const CompactLds& lds, int L, int first)
{
    if (src.tile != nullptr) {
        const char* base = reinterpret_cast<const char*>(src.tile);
        const size_t pitch = (size_t)src.N * sizeof(double);
#pragma unroll
        for (int m = 0; m < LT; ++m) {
            const int j = STRIDE * m + first, from = j < L ? j : 0;
            const char* column = base + (size_t)__builtin_amdgcn_readfirstlane(lds.col[from]) * pitch;
#pragma unroll
            for (int s = 0; s < S; ++s) C[s][m] = *reinterpret_cast<const double*>(column + src.at[s]);
        }
    } else {
        const double* row[S];
#pragma unroll
        for (int s = 0; s < S; ++s) row[s] = src.table + (size_t)src.wid[s] * src.ldk;
#pragma unroll
        for (int m = 0; m < LT; ++m) {
            const int j = STRIDE * m + first, from = j < L ? j : 0;
            const int topic = __builtin_amdgcn_readfirstlane(lds.idx[from]);
#pragma unroll
            for (int s = 0; s < S; ++s) C[s][m] = row[s][topic];
        }
    }
}

// this lane's column after the reduce-scatter of LT values (the recursion of compact_column_of on the lane's bits, shape
// constants folded), or -1; `first`: the replica with the lowest lane number
template <int LT>
__device__ __forceinline__ int compact_my_column(int lane, bool* first_replica)
{
    int m = 0;
    bool valid = true, first = true;
    static_for<6>([&](auto idx) {
        constexpr int k = 5 - decltype(idx)::value;
        constexpr int rr = compact_count_at(LT, k), half = compact_half(rr);
        const bool bit = ((lane >> (5 - k)) & 1) != 0;
        if constexpr (rr == 1) {
            if (bit) first = false;
        } else {
            if (bit) {
                m += half;
                if (m >= rr) valid = false;
            }
        }
    });
    *first_replica = first;
    return valid ? m : -1;
}

// q_j = sum_s r_s C[s][j] per lane, then over the 64 lanes: the value of this lane's column
template <int S, int LT>
__device__ __forceinline__ double compact_topic_sums(const double (&C)[S][LT], const double (&r)[S], int lane)
{
    constexpr int H0 = (LT + 1) / 2;
    double y0[H0];
#pragma unroll
    for (int m = 0; m < H0; ++m) {
        double qa = r[0] * C[0][m], qb = m + H0 < LT ? r[0] * C[0][m + H0 < LT ? m + H0 : 0] : 0.0;
#pragma unroll
        for (int s = 1; s < S; ++s) {
            qa = fma(r[s], C[s][m], qa);
            if (m + H0 < LT) qb = fma(r[s], C[s][m + H0 < LT ? m + H0 : 0], qb);
        }
        y0[m] = swap32_add(qa, qb);
    }
    return compact_reduce_from<1, H0>(y0, lane);
}

// The columns still alive move up, in order (lane j < L looks after column j); a column that died keeps alpha_k for good.
// One wavefront; the state of all columns is in LDS.
__device__ __forceinline__ void compact_keep_alive(const EstepParams& p, int doc, CompactLds& lds, CompactState& st)
{
    const int lane = threadIdx.x & (kWave - 1);
    const bool mine = lane < st.L;
    const int at = mine ? lane : 0;
    const double g = lds.gam[at], a = lds.alf[at];
    const int topic = lds.idx[at], column = lds.col[at];
    const bool alive = mine && g != a;
    const unsigned long long mask = __ballot(alive);
    if (mine && !alive) p.gamma[(size_t)doc * p.K + topic] = g;
    wave_lds_exchange();
    if (alive) {
        const int to = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        lds.gam[to] = g;
        lds.alf[to] = a;
        lds.idx[to] = topic;
        lds.col[to] = column;
    }
    wave_lds_exchange();
    st.L = __builtin_popcountll(mask);
}

// The inner loop on an S x LT register tile, ONE wavefront.  Returns when the document is finished (stop test, iteration
// cap) or when at most `shrink_to` columns are still alive (0: never) - st.L, lds.idx / col / gam are then the survivors.
template <int S, int LT>
__device__ __forceinline__ int compact_body(const EstepParams& p, int doc, int64_t lo, int N, double psi_total, double thresh_f,
                                            CompactLds& lds, CompactState& st, int shrink_to, const CompactSource<S>& src, double (&r)[S])
{
    static_assert(LT >= 4 && LT <= kLiveStride && LT % 4 == 0, "columns of the register tile");
    const int lane = threadIdx.x & (kWave - 1);
    const int L = st.L;

    // the tile: C[s][j] = value of term lane + 64 s, column j
    double C[S][LT];
    double cnt[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int n = s * kWave + lane;
        cnt[s] = n < N ? (double)p.term_ct[lo + n] : 0.0;
    }
    compact_load_tile<S, LT, 1>(C, src, lds, L, 0);

    bool first_replica = false;
    int mycol = compact_my_column<LT>(lane, &first_replica);
    if (mycol >= L) mycol = -1;
    const bool owns = mycol >= 0, primary = owns && first_replica;
    double gam = owns ? lds.gam[owns ? mycol : 0] : 1.0;
    const double alpha_signed = owns ? lds.alf[owns ? mycol : 0] : 1.0;      // (sign bit: the topic never counts as dead)
    const double alpha_k = fabs(alpha_signed);
    double t;
    {
        ExpDigammaLevelsA coef_a;
        coef_a.load();
        t = exp_digamma_minus_levels(gam, psi_total, coef_a);
        if (!owns) t = 0.0;
    }
    double gprev = gam, tused = t;
    int it = st.it, left = st.left, bad = st.bad;
    double nrm_min = st.nrm_min, r_max = st.r_max;
    int exit_code = kCompactDone;

    for (;;) {                                                                // :174
        // t_j as scalars
        double ts[LT];
        static_for<LT>([&](auto idx) {
            constexpr int j = decltype(idx)::value;
            constexpr int from = compact_lane_of(LT, j);
            static_assert(from >= 0 && from < kWave, "every column has a lane");
            ts[j] = readlane_f64(t, from);
        });
        tused = t;
        // A. normalisers, lane-local                                            :177-182
        double nrm[S];
#pragma unroll
        for (int s = 0; s < S; ++s) nrm[s] = C[s][0] * ts[0];
#pragma unroll
        for (int j = 1; j < LT; ++j)
#pragma unroll
            for (int s = 0; s < S; ++s) nrm[s] = fma(C[s][j], ts[j], nrm[s]);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (cnt[s] > 0.0 && !(nrm[s] > 1e-280)) bad = 1;
            r[s] = cnt[s] * rcp_newton(nrm[s]);
            nrm_min = fmin(nrm_min, nrm[s]);
            r_max = fmax(r_max, r[s]);
        }
        // both coefficient tables of exp_digamma_minus_levels (62 scalar registers) are fetched again in every iteration,
        // here, where the t_j have just been used for the last time: resident they would push the t_j out of the scalar file
        ExpDigammaLevelsA coef_a;
        ExpDigammaLevelsB coef_b;
        coef_a.load();
        coef_b.load();
        // B. topic sums: per lane over its terms, then over the lanes                 :185
        const double q = compact_topic_sums<S, LT>(C, r, lane);
        // C. gamma update on the lanes that own a column                               :185-188
        const double gnew = fma(t, q, alpha_k);
        const double diff = fabs(gnew - gam);
        gprev = gam;
        gam = gnew;
        double moved = primary ? __builtin_rint(fmin(diff, 1024.0) * kChangeScale) : 0.0;
        t = exp_digamma_minus_levels<true>(gam, psi_total, coef_a, &coef_b);
        if (!owns) t = 0.0;
        // the stop sum in 2^-40 fixed point, as the dense kernels form it (integers in doubles: exact below 2^53, and
        // a sum that is not exact is far above any threshold of the fixed-point range)
        moved = wave_sum(moved);
        ++it;
        --left;
        if (moved <= thresh_f || left <= 0) break;                            // :189, :174
        if (shrink_to > 0) {
            const int alive = __builtin_popcountll(__ballot(primary && gam != alpha_signed));
            if (alive <= shrink_to) {
                exit_code = kCompactShrink;
                break;
            }
        }
    }

    // state back to LDS: gamma, the gamma before the last update and the t used last, per column
    if (primary) {
        lds.gam[mycol] = gam;
        lds.gprev[mycol] = gprev;
        lds.tlast[mycol] = tused;
    }
    wave_lds_exchange();
    if (exit_code == kCompactShrink) compact_keep_alive(p, doc, lds, st);
    st.cols += (it - st.it) * LT;
    st.it = it;
    st.left = left;
    st.bad = bad;
    st.nrm_min = nrm_min;
    st.r_max = r_max;
    return exit_code;
}

// The same loop on TWO wavefronts that split the COLUMNS (wavefront w owns columns 2 m + w, m < TPW): twice the register
// tile - documents of up to 2 TPW live topics, i.e. the dense kernel lets go five iterations earlier, and documents whose
// live set never falls to one wavefront's capacity (a trained model: 20-35 topics per document) leave it at all.
// Per iteration the wavefronts exchange ONE thing, their partial normalisers (S doubles per lane through LDS, one
// barrier): both then hold the same normalisers and r, bit for bit (the sum is wave 0's part + wave 1's part on both),
// and everything else - topic sums, gamma, t, the stop sum - is local to the columns a wavefront owns.  The stop sum and
// the live count of an iteration travel with the NEXT iteration's partials: the decision to stop is taken one exchange
// late, and the partial normalisers computed in between are dropped.  Returns like compact_body; with kCompactShrink
// (at most `hand_down_to` columns alive) the caller lets wavefront 1 go and wavefront 0 carries on alone.
template <int S, int TPW>
__device__ __forceinline__ int compact_pair_body(const EstepParams& p, int doc, int64_t lo, int N, double psi_total, double thresh_f,
                                                 CompactLds& lds, double* xchg, CompactState& st, int hand_down_to, const CompactSource<S>& src,
                                                 double (&r)[S])
{
    static_assert(TPW >= 4 && 2 * TPW <= kLiveStride && TPW % 4 == 0, "columns of a wavefront's register tile");
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int L = st.L;
    double* xn = xchg;                                      // [2][2][S][64] partial normalisers
    double* xs = xchg + 2 * 2 * S * kWave;                  // [2][2][2]     stop sum, live columns

    double C[S][TPW];
    double cnt[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int n = s * kWave + lane;
        cnt[s] = n < N ? (double)p.term_ct[lo + n] : 0.0;
    }
    compact_load_tile<S, TPW, 2>(C, src, lds, L, wave);
    bool first_replica = false;
    const int mine_local = compact_my_column<TPW>(lane, &first_replica);
    int mycol = mine_local >= 0 ? 2 * mine_local + wave : -1;
    if (mycol >= L) mycol = -1;
    const bool owns = mycol >= 0, primary = owns && first_replica;
    double gam = owns ? lds.gam[owns ? mycol : 0] : 1.0;
    const double alpha_signed = owns ? lds.alf[owns ? mycol : 0] : 1.0;      // (sign bit: the topic never counts as dead)
    const double alpha_k = fabs(alpha_signed);
    double t;
    {
        ExpDigammaLevelsA coef_a;
        coef_a.load();
        t = exp_digamma_minus_levels(gam, psi_total, coef_a);
        if (!owns) t = 0.0;
    }
    double gprev = gam, tused = t;
    int it = st.it, left = st.left, bad = st.bad;
    double nrm_min = st.nrm_min, r_max = st.r_max;
    double moved_mine = 0.0;
    int alive_mine = 0;
    int exit_code = kCompactDone;

    for (bool first = true;; first = false) {                                 // :174
        double ts[TPW];
        static_for<TPW>([&](auto idx) {
            constexpr int m = decltype(idx)::value;
            constexpr int from = compact_lane_of(TPW, m);
            static_assert(from >= 0 && from < kWave, "every column has a lane");
            ts[m] = readlane_f64(t, from);
        });
        // A. this wavefront's part of the normalisers, out to the other one - with the stop sum and the live columns of
        //    the iteration before
        double pn[S];
#pragma unroll
        for (int s = 0; s < S; ++s) pn[s] = C[s][0] * ts[0];
#pragma unroll
        for (int m = 1; m < TPW; ++m)
#pragma unroll
            for (int s = 0; s < S; ++s) pn[s] = fma(C[s][m], ts[m], pn[s]);
        const int buf = it & 1;
#pragma unroll
        for (int s = 0; s < S; ++s) xn[((buf * 2 + wave) * S + s) * kWave + lane] = pn[s];
        if (lane == 0) {
            xs[(buf * 2 + wave) * 2] = moved_mine;
            xs[(buf * 2 + wave) * 2 + 1] = (double)alive_mine;
        }
        lds_only_barrier();
        double nrm[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const double other = xn[((buf * 2 + (wave ^ 1)) * S + s) * kWave + lane];
            nrm[s] = wave == 0 ? pn[s] + other : other + pn[s];               // (wave 0's part + wave 1's part, on both)
        }
        if (!first) {
            const double moved = xs[(buf * 2 + 0) * 2] + xs[(buf * 2 + 1) * 2];
            const int alive = (int)(xs[(buf * 2 + 0) * 2 + 1] + xs[(buf * 2 + 1) * 2 + 1]);
            if (uniform_f64(moved) <= thresh_f || left <= 0) break;           // :189, :174
            if (__builtin_amdgcn_readfirstlane(alive) <= hand_down_to) {
                exit_code = kCompactShrink;
                break;
            }
        }
        tused = t;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (cnt[s] > 0.0 && !(nrm[s] > 1e-280)) bad = 1;
            r[s] = cnt[s] * rcp_newton(nrm[s]);
            nrm_min = fmin(nrm_min, nrm[s]);
            r_max = fmax(r_max, r[s]);
        }
        ExpDigammaLevelsA coef_a;
        ExpDigammaLevelsB coef_b;
        coef_a.load();
        coef_b.load();
        // B. topic sums of the columns this wavefront owns                                :185
        const double q = compact_topic_sums<S, TPW>(C, r, lane);
        // C. their gamma update                                                           :185-188
        const double gnew = fma(t, q, alpha_k);
        const double diff = fabs(gnew - gam);
        gprev = gam;
        gam = gnew;
        const double moved = primary ? __builtin_rint(fmin(diff, 1024.0) * kChangeScale) : 0.0;
        t = exp_digamma_minus_levels<true>(gam, psi_total, coef_a, &coef_b);
        if (!owns) t = 0.0;
        moved_mine = wave_sum(moved);
        alive_mine = __builtin_popcountll(__ballot(primary && gam != alpha_signed));
        ++it;
        --left;
    }

    // state to LDS, both wavefronts; the caller's barrier publishes it
    if (primary) {
        lds.gam[mycol] = gam;
        lds.gprev[mycol] = gprev;
        lds.tlast[mycol] = tused;
    }
    st.cols += (it - st.it) * 2 * TPW;
    st.it = it;
    st.left = left;
    st.bad = bad;
    st.nrm_min = nrm_min;
    st.r_max = r_max;
    return exit_code;
}

__device__ __forceinline__ double wave_min(double v) { return -wave_max(-v); }

// LDS of a workgroup: the state, the pair body's exchange area, the epilogue's row of t (row mode of the statistics only)
constexpr size_t compact_state_bytes() { return (sizeof(CompactLds) + 15) & ~(size_t)15; }
constexpr size_t compact_xchg_bytes(int slots, int tpw) { return tpw > 0 ? (size_t)(2 * 2 * slots * kWave + 8) * 8 : 0; }
constexpr size_t compact_lds_bytes(int ldk, int slots, int tpw) { return compact_state_bytes() + compact_xchg_bytes(slots, tpw) + (size_t)ldk * 8; }

// One workgroup per document of the launch; S = term slots per lane (N <= 64 S); LTMAX: columns one wavefront holds;
// TPW > 0: the workgroup is TWO wavefronts that hold TPW columns each while more than LTMAX are alive (compact_pair_body),
// then wavefront 1 leaves.  Documents the dense kernel finished itself (status != 4) are skipped.
template <int S, int LTMAX, int TPW>
__global__ __launch_bounds__(TPW > 0 ? 2 * kWave : kWave, 2) void estep_compact_kernel(EstepParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CompactLds& lds = *reinterpret_cast<CompactLds*>(smem);
    double* xchg = reinterpret_cast<double*>(smem + compact_state_bytes());
    double* trow = reinterpret_cast<double*>(smem + compact_state_bytes() + compact_xchg_bytes(S, TPW));     // [ldk]
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int doc = p.order[blockIdx.x];
    if (p.status[doc] != 4) return;
    // one workgroup in 64 times itself twice - shader cycles and the constant-rate counter: the sustained clock under
    // this load (pylda_clock_counters)
    const bool timed = p.clock_acc != nullptr && (blockIdx.x & 63) == 0 && wave == 0;
    const long long tick0 = timed ? clock64() : 0, wall0 = timed ? wall_clock64() : 0;
    auto clock_out = [&]() {
        if (timed && lane == 0) {
            unsafeAtomicAdd(p.clock_acc, (double)(clock64() - tick0));
            unsafeAtomicAdd(p.clock_acc + 1, (double)(wall_clock64() - wall0));
        }
    };
    const int K = p.K, ldk = p.ldk;
    const int64_t lo = p.doc_ptr[doc];
    const int N = (int)(p.doc_ptr[doc + 1] - lo);
    const int L0 = p.live_n[doc];

    if (wave == 0 && lane < L0) {
        const int topic = *live_idx_at(live_list_of(p.live_list, doc), lane);
        lds.idx[lane] = topic;
        lds.col[lane] = lane;
        lds.gam[lane] = p.gamma[(size_t)doc * K + topic];
        lds.alf[lane] = p.alpha_sgn[topic];
    }
    // total token count (:162) and psi(sum_k gamma_k), formed as the dense kernels form them (the same bits)
    double local = 0.0;
    for (int n = lane; n < N; n += kWave) local += (double)p.term_ct[lo + n];
    double asum = 0.0;
    for (int k = lane; k < K; k += kWave) asum += p.alpha[k];
    const double total = wave_sum(local);
    asum = wave_sum(asum);
    const double psi_total = uniform_f64(digamma(asum + total));
    const double thresh_f = p.tol * K * kChangeScale;
    CompactSource<S> src;
    src.tile = p.tile_from_table ? nullptr : p.live_tile + p.tile_ptr[doc];
    src.table = p.expElog;
    src.ldk = ldk;
    src.N = N;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int n = s * kWave + lane < N ? s * kWave + lane : N - 1;
        src.at[s] = (unsigned)n * (unsigned)sizeof(double);
        src.wid[s] = p.tile_from_table ? p.term_id[lo + n] : 0;
    }
    if constexpr (TPW > 0) lds_only_barrier();
    else wave_lds_exchange();

    CompactState st;
    st.L = L0;
    st.it = p.iters[doc];
    st.left = p.max_iter - st.it;
    st.cols = 0;
    st.bad = 0;
    st.nrm_min = 1e300;
    st.r_max = 0.0;
    double r[S];
    bool finished = false;
    if constexpr (TPW > 0) {
        if (st.L > LTMAX) {
            const int code = compact_pair_body<S, TPW>(p, doc, lo, N, psi_total, thresh_f, lds, xchg, st, LTMAX, src, r);
            lds_only_barrier();                             // both wavefronts' columns are in LDS
            if (wave != 0) return;
            finished = code == kCompactDone;
            if (!finished) compact_keep_alive(p, doc, lds, st);
        } else if (wave != 0) {
            return;
        }
    }
    while (!finished) {
        int code;
        // the smallest instantiation that holds the live columns; it runs until the next smaller one would do
        if constexpr (LTMAX > 24) {
            if (st.L > 24) {
                code = compact_body<S, LTMAX>(p, doc, lo, N, psi_total, thresh_f, lds, st, 24, src, r);
                finished = code == kCompactDone;
                continue;
            }
        }
        if constexpr (LTMAX > 16) {
            if (st.L > 16) {
                code = compact_body<S, (LTMAX < 24 ? LTMAX : 24)>(p, doc, lo, N, psi_total, thresh_f, lds, st, 16, src, r);
                finished = code == kCompactDone;
                continue;
            }
        }
        if constexpr (LTMAX > 8) {
            if (st.L > 8) {
                code = compact_body<S, (LTMAX < 16 ? LTMAX : 16)>(p, doc, lo, N, psi_total, thresh_f, lds, st, 8, src, r);
                finished = code == kCompactDone;
                continue;
            }
        }
        code = compact_body<S, 8>(p, doc, lo, N, psi_total, thresh_f, lds, st, 0, src, r);
        finished = true;
    }
    const int L = st.L;

    // ---- the exactness guard (header), from the extremes over every iteration run here.  The dead topics: every topic
    //      that is not a column now (a column is only ever dropped once its gamma IS alpha_k) - the largest alpha among
    //      THEM bounds their t (a topic with a large alpha never dies bitwise: it is a column to the end) ----
    if (lane < 32) lds.member[lane] = 0u;
    wave_lds_exchange();
    if (lane < L) atomicOr(&lds.member[lds.idx[lane] >> 5], 1u << (lds.idx[lane] & 31));
    wave_lds_exchange();
    {
        double alpha_dead = 0.0;
        for (int k = lane; k < K; k += kWave)
            if (!((lds.member[k >> 5] >> (k & 31)) & 1u)) alpha_dead = fmax(alpha_dead, p.alpha[k]);
        alpha_dead = wave_max(alpha_dead);
        const double nrm_min = wave_min(st.nrm_min), r_max = wave_max(st.r_max);
        const double t_dead = alpha_dead > 0.0 ? exp_digamma_minus(alpha_dead, psi_total) : 0.0;
        const bool safe = (double)K * t_dead < 8.673617379884035e-19 * nrm_min &&            // 2^-60
                          t_dead * (double)N * r_max < 5.551115123125783e-17 * p.alpha_min;     // 2^-54
        if (!safe) st.bad = 1;
    }
    if (__ballot(st.bad != 0) != 0ull) {
        if (!p.heldout) {      // contributes nothing to the gather pass; the log-space kernel adds it
            for (int n = lane; n < N; n += kWave) p.rfinal[lo + n] = 0.0;
            if (p.live_stats) {
                if (lane == 0) p.live_n[doc] = 0;
            } else {
                for (int k = lane; k < ldk; k += kWave) p.tfinal[(size_t)doc * ldk + k] = 0.0;
            }
        }
        if (lane == 0) p.status[doc] = 1;
        clock_out();
        return;
    }

    // t of the last executed iteration for the statistics pass: the document's LIST (live_stats: 10 bytes per live topic
    // instead of a 2-KiB row) or the dense row with 0 at the dead topics - whose 1e-114 is below the last bit of every
    // statistic it would touch either way: eta = statistics + beta is unchanged
    const bool mine = lane < L;
    const int at = mine ? lane : 0;
    const int topic = lds.idx[at];
    const double gam = lds.gam[at], gprev = lds.gprev[at], tlast = lds.tlast[at], alpha_k = fabs(lds.alf[at]);
    if (mine) p.gamma[(size_t)doc * K + topic] = gam;
    if (lane == 0) p.col_iters[doc] = st.cols;
    if (!p.heldout) {
        if (p.live_stats) {
            if (mine) {
                *live_idx_at(live_list_of(p.live_list, doc), lane) = (uint16_t)topic;
                *live_t_at(live_list_of(p.live_list, doc), lane) = tlast;
            }
            if (lane == 0) p.live_n[doc] = L;
        } else {
            for (int k = lane; k < ldk; k += kWave) trow[k] = 0.0;
            wave_lds_exchange();
            if (mine) trow[topic] = tlast;
            wave_lds_exchange();
            for (int k = lane; k < ldk; k += kWave) p.tfinal[(size_t)doc * ldk + k] = trow[k];
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int n = s * kWave + lane;
            if (n < N) p.rfinal[lo + n] = r[s];
        }
    }
    // ---- training fast path: the document terms are left to doc_terms_kernel (doc_terms.h) ----
    if (!p.heldout && !p.want_doc_ll) {
        if (lane == 0) {
            p.iters[doc] = st.it;
            p.status[doc] = 3;
        }
        clock_out();
        return;
    }

    // ---- document terms (:195-204) with the last phi = B t r, as the dense kernels' epilogue (estep_epilogue.h) ----
    double term1 = 0.0, term3 = 0.0, shift_term = 0.0;      // (lds.member: the columns, from the guard above)
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int n = s * kWave + lane;
        if (n < N) {
            const int w = p.term_id[lo + n];
            const double* row = p.expElog_elog + (size_t)w * ldk;
            double gsum = 0.0;
            for (int j = 0; j < L; ++j) gsum = fma(row[lds.idx[j]], lds.tlast[j], gsum);
            term1 = fma(r[s], gsum, term1);
            const double c = (double)p.term_ct[lo + n];
            term3 = fma(c, log(c) - log(r[s]), term3);                    // c_n log(normaliser_n)
            if (p.heldout) shift_term = fma(c, p.shift[w], shift_term);
        }
    }
    double term2 = 0.0, lse_term = 0.0, lgam = 0.0, gsum_all = 0.0;
    if (mine) {
        const double mass = gam - alpha_k;                                // = t_last * sum_n r_n B[w_n][k]
        term2 = (digamma(gprev) - psi_total) * mass;
        if (p.heldout) lse_term = p.topic_lse[topic] * mass;
        lgam = lgamma_pos(gam);
        gsum_all = gam;
    }
    for (int k = lane; k < K; k += kWave) {                               // the topics that are not columns sit at alpha_k
        if (!((lds.member[k >> 5] >> (k & 31)) & 1u)) {
            const double a = p.alpha[k];
            lgam += lgamma_pos(a);
            gsum_all += a;
        }
    }
    term1 = wave_sum(term1);
    term2 = wave_sum(term2);
    term3 = wave_sum(term3);
    shift_term = wave_sum(shift_term);
    lse_term = wave_sum(lse_term);
    lgam = wave_sum(lgam);
    gsum_all = wave_sum(gsum_all);
    if (lane == 0) {
        const double ent = term1 + term2 - term3;
        p.doc_ll[doc] = p.alpha_term + lgam - lgamma_pos(gsum_all) - ent;        // :195-199
        p.doc_words_ll[doc] = p.heldout ? term1 + shift_term - lse_term : 0.0;   // :204
        p.iters[doc] = st.it;
        p.status[doc] = 0;
    }
    clock_out();
}

}  // namespace pylda
