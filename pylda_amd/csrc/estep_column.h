// Register-resident E-step kernel, topic-major ("column") layout, for
// 32 < K <= 128: the N_d x K tile of B = exp(E_log_eta - shift) lives in VGPRs.
//
// One workgroup = one document = W wavefronts.
//   lane l       owns topics  l, l+64, ... (KR = ldk/64 per lane)
//   wavefront w  owns words   [w*RNW, (w+1)*RNW)
//   => lane registers hold B[RNW][KR]; table rows are read coalesced across
//      lanes (64 consecutive doubles), the layout the HBM/L2 path likes best.
//
// Compared with the word-major slab kernel (estep_slab.h) this layout
//   * computes r[n] = c_n / nrm_n once per word (no per-wave redundancy),
//   * runs the K digamma+exp evaluations of an iteration on K distinct lanes
//     (wavefronts 0..KR-1), instead of 64/RK-fold redundantly in every wave,
//   * needs no v_readlane broadcast: t[k] is per-lane data.
// rocprof on cfg 3 showed the slab kernel issue-bound with only 30 % of its
// VALU instructions being tile FMAs; this layout removes most of the rest.
//
// One inner iteration (variational_bayes.py:177-190), exp-hoisted, with
// t[k] = exp(psi(gamma_k) - psi(sum_k gamma_k)):
//   A. p[n] = sum_{k in lane} B[n][k] t[k]            in-lane
//      nrm[n] = sum_lanes p[n]                         2 permlane-swap levels + LDS transpose
//      r[n] = c_n / nrm[n]                             once per word, shared through LDS
//   B. q[k] = sum_{n in wave} r[n] B[n][k]             in-lane -> LDS partial[w][k]
//      barrier
//   C. (threads 0..K-1)  s[k] = sum_w partial[w][k];  gamma'_k = alpha_k + t[k] s[k];
//      mean |gamma' - gamma| via a fixed-point LDS atomic (order-independent, so
//      the convergence decision is reproducible); next t[k]
//      barrier
#pragma once
#include "estep_common.h"
#include "special_device.h"

namespace pylda {

template <int W, int KR, int RNW>
struct ColumnLds {
    static constexpr int kTopics = kWave * KR;
    static constexpr size_t red = 0;                                             // [W][RNW][17]
    static constexpr size_t rr = red + (size_t)W * RNW * 17 * 8;                 // [W][RNW]
    static constexpr size_t sp = rr + (size_t)W * RNW * 8;                       // [W][kTopics]
    static constexpr size_t tt = sp + (size_t)W * kTopics * 8;                   // [2][kTopics]
    static constexpr size_t chg = tt + (size_t)2 * kTopics * 8;                  // u64[2]
    static constexpr size_t misc = chg + 16;                                     // [8][W]
    static constexpr size_t total = (misc + (size_t)8 * W * 8 + 15) & ~(size_t)15;
};

template <int W, int KR, int RNW>
__global__ __launch_bounds__(kWave* W) void estep_column_kernel(EstepParams p)
{
    using L = ColumnLds<W, KR, RNW>;
    constexpr int NT = kWave * W;
    constexpr int KT = kWave * KR;          // padded topic count (== ldk)
    constexpr int LPW = kWave / RNW;        // lanes that share one word after the reduction
    constexpr int Q = RNW / 4;              // values per lane after the two swap levels
    static_assert(RNW == 8 || RNW == 16 || RNW == 32, "words per wavefront");
    static_assert(KT <= NT, "one thread per topic in the gamma phase");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem + L::red);
    double* rr = reinterpret_cast<double*>(smem + L::rr);
    double* sp = reinterpret_cast<double*>(smem + L::sp);
    double* tt = reinterpret_cast<double*>(smem + L::tt);
    unsigned long long* chg = reinterpret_cast<unsigned long long*>(smem + L::chg);
    double* misc = reinterpret_cast<double*>(smem + L::misc);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int K = p.K, ldk = p.ldk;
    const int doc = p.order[blockIdx.x];
    const int64_t lo = p.doc_ptr[doc];
    const int N = (int)(p.doc_ptr[doc + 1] - lo);
    const int nb = wave * RNW;              // first word of this wavefront

    // ---- load this wavefront's rows: coalesced across lanes ----
    double B[RNW][KR];
#pragma unroll
    for (int i = 0; i < RNW; ++i) {
        const int n = nb + i;
        if (n < N) {
            const double* row = p.expElog + (size_t)p.term_id[lo + n] * ldk;
#pragma unroll
            for (int j = 0; j < KR; ++j) B[i][j] = row[lane + kWave * j];
        } else {
#pragma unroll
            for (int j = 0; j < KR; ++j) B[i][j] = 0.0;
        }
    }
    // the word this lane finishes in the normaliser reduction: nb + lane / LPW
    const int my_word = nb + lane / LPW;
    const bool word_live = my_word < N;
    const double my_cnt = word_live ? (double)p.term_ct[lo + my_word] : 0.0;

    // ---- total token count (:162) and the invariant sum_k gamma_k ----
    double local = 0.0;
    for (int n = tid; n < N; n += NT) local += (double)p.term_ct[lo + n];
    local = wave_sum(local);
    double asum = 0.0;
    for (int k = lane; k < K; k += kWave) asum += p.alpha[k];
    asum = wave_sum(asum);
    if (lane == 0) misc[wave] = local;
    if (tid == 0) chg[0] = chg[1] = 0ull;
    __syncthreads();
    double total = 0.0;
#pragma unroll
    for (int w = 0; w < W; ++w) total += misc[w];
    const double psi_total = digamma(asum + total);

    // ---- gamma phase state: thread k < KT owns topic k ----
    const bool topic_thread = tid < KT;
    const bool topic_live = tid < K;
    const double alpha_k = topic_live ? p.alpha[tid] : 1.0;
    double gam = alpha_k + total / K;                                     // :165
    double gam_prev = gam;
    double t_mine = 0.0;
    if (topic_thread) {
        t_mine = topic_live ? exp_digamma_minus(gam, psi_total) : 0.0;
        tt[tid] = t_mine;
    }
    __syncthreads();

    double r_mine = 0.0, nrm_mine = 1.0;
    int it = 0;
    int bad = 0;
    while (it < p.max_iter) {                                             // :174
        const int buf = it & 1;
        double tq[KR];
#pragma unroll
        for (int j = 0; j < KR; ++j) tq[j] = tt[buf * KT + lane + kWave * j];

        // A. per-lane partial normalisers, then the sum over lanes.  The two swap
        // levels pair word m with m + RNW/2, then with m + RNW/4 (see estep_slab.h).
        double u[RNW / 2];
#pragma unroll
        for (int m = 0; m < RNW / 2; ++m) {
            double a = B[m][0] * tq[0], b = B[m + RNW / 2][0] * tq[0];
#pragma unroll
            for (int j = 1; j < KR; ++j) {
                a = fma(B[m][j], tq[j], a);
                b = fma(B[m + RNW / 2][j], tq[j], b);
            }
            u[m] = swap32_add(a, b);
        }
        double v[Q];
#pragma unroll
        for (int m = 0; m < Q; ++m) v[m] = swap16_add(u[m], u[m + Q]);
        double* myred = red + (size_t)wave * RNW * 17;
        {
            const int r4 = lane >> 4, c = lane & 15;
            const int wbase = (r4 & 1) * Q + (r4 >> 1) * (RNW / 2);
#pragma unroll
            for (int m = 0; m < Q; ++m) myred[(wbase + m) * 17 + c] = v[m];
        }
        wave_lds_exchange();
        {
            const int part = lane % LPW;
            const double* src = myred + (lane / LPW) * 17 + part * (16 / LPW);
            double s = src[0];
#pragma unroll
            for (int x = 1; x < 16 / LPW; ++x) s += src[x];
#pragma unroll
            for (int m = 1; m < LPW; m <<= 1) s += __shfl_xor(s, m, kWave);
            nrm_mine = s;
            if (word_live && !(s > 1e-280 && s < 1e300)) bad = 1;
            r_mine = word_live ? my_cnt * rcp_newton(s) : 0.0;
            if (part == 0) rr[wave * RNW + lane / LPW] = r_mine;
        }
        wave_lds_exchange();

        // B. q[k] over this wavefront's words (r broadcast from LDS, two words per read)
        double q0[KR], q1[KR];
#pragma unroll
        for (int j = 0; j < KR; ++j) q0[j] = q1[j] = 0.0;
        const double2* rsrc = reinterpret_cast<const double2*>(rr + wave * RNW);
#pragma unroll
        for (int i = 0; i < RNW; i += 2) {
            const double2 r2 = rsrc[i / 2];
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                q0[j] = fma(r2.x, B[i][j], q0[j]);
                q1[j] = fma(r2.y, B[i + 1][j], q1[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < KR; ++j) sp[wave * KT + lane + kWave * j] = q0[j] + q1[j];
        __syncthreads();

        // C. gamma update by the topic threads
        if (topic_thread) {
            double s = sp[tid];
#pragma unroll
            for (int w = 1; w < W; ++w) s += sp[w * KT + tid];
            const double gnew = fma(t_mine, s, alpha_k);                  // :185
            const double diff = topic_live ? fabs(gnew - gam) : 0.0;      // :187
            gam_prev = gam;
            gam = gnew;                                                   // :188
            atomicAdd(&chg[buf], change_fixed(diff));
            t_mine = topic_live ? exp_digamma_minus(gam, psi_total) : 0.0;
            tt[(buf ^ 1) * KT + tid] = t_mine;
            if (tid == 0) chg[buf ^ 1] = 0ull;
        }
        ++it;
        __syncthreads();
        const double change = (double)chg[buf] * (1.0 / kChangeScale);
        if (change <= p.tol * K) break;                                   // :189 (mean <= tol)
    }
    const int last = (it - 1) & 1;          // tt[last] holds t of the last executed iteration

    bad = __syncthreads_or(bad);
    if (bad) {
        if (!p.heldout) {      // contributes nothing to the gather pass; the log-space kernel adds it
            for (int n = tid; n < N; n += NT) p.rfinal[lo + n] = 0.0;
            for (int k = tid; k < ldk; k += NT) p.tfinal[(size_t)doc * ldk + k] = 0.0;
        }
        if (tid == 0) p.status[doc] = 1;
        return;
    }

    // ---- document terms (:195-204) with the last phi = B t r (see estep_slab.h) ----
    double tq[KR];
#pragma unroll
    for (int j = 0; j < KR; ++j) tq[j] = tt[last * KT + lane + kWave * j];
    double term1 = 0.0;
    const bool do_term1 = p.heldout || p.want_doc_ll;     // else: taken per corpus from the statistics
#pragma unroll
    for (int i = 0; i < RNW; ++i) {
        const int n = nb + i;
        if (n < N && do_term1) {
            const double* row = p.expElog_elog + (size_t)p.term_id[lo + n] * ldk;
            double g = row[lane] * tq[0];
#pragma unroll
            for (int j = 1; j < KR; ++j) g = fma(row[lane + kWave * j], tq[j], g);
            term1 = fma(rr[wave * RNW + i], g, term1);
        }
    }
    const bool word_owner = word_live && (lane % LPW) == 0;
    double term3 = word_owner ? my_cnt * log(nrm_mine) : 0.0;
    double shift_term = (word_owner && p.heldout) ? my_cnt * p.shift[p.term_id[lo + my_word]] : 0.0;
    double term2 = 0.0, lse_term = 0.0, lgam = 0.0, gsum = 0.0;
    if (topic_live) {
        const double t_last = tt[last * KT + tid];
        const double moved = gam - alpha_k;                               // = t_last * s
        const double ltv = digamma(gam_prev) - psi_total;                 // log t of the last iteration
        term2 = ltv * moved;
        if (p.heldout) lse_term = p.topic_lse[tid] * moved;
        lgam = lgamma_pos(gam);
        gsum = gam;
        p.gamma[(size_t)doc * K + tid] = gam;
        if (!p.heldout) p.tfinal[(size_t)doc * ldk + tid] = t_last;
    } else if (topic_thread && !p.heldout) {
        p.tfinal[(size_t)doc * ldk + tid] = 0.0;
    }
    if (word_owner && !p.heldout) p.rfinal[lo + my_word] = r_mine;
    term1 = wave_sum(term1);
    term2 = wave_sum(term2);
    lse_term = wave_sum(lse_term);
    lgam = wave_sum(lgam);
    gsum = wave_sum(gsum);
    term3 = wave_sum(term3);
    shift_term = wave_sum(shift_term);
    if (lane == 0) {
        misc[0 * W + wave] = term1;
        misc[1 * W + wave] = term2;
        misc[2 * W + wave] = lse_term;
        misc[3 * W + wave] = lgam;
        misc[4 * W + wave] = gsum;
        misc[5 * W + wave] = term3;
        misc[6 * W + wave] = shift_term;
    }
    __syncthreads();
    if (tid == 0) {
        double t1 = 0.0, t2 = 0.0, tl = 0.0, lg = 0.0, gs = 0.0, t3 = 0.0, sh = 0.0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            t1 += misc[0 * W + w];
            t2 += misc[1 * W + w];
            tl += misc[2 * W + w];
            lg += misc[3 * W + w];
            gs += misc[4 * W + w];
            t3 += misc[5 * W + w];
            sh += misc[6 * W + w];
        }
        const double ent = t1 + t2 - t3;
        p.doc_ll[doc] = p.alpha_term + lg - lgamma_pos(gs) - ent;        // :195-199
        p.doc_words_ll[doc] = p.heldout ? t1 + sh - tl : 0.0;            // :204
        p.iters[doc] = it;
        p.status[doc] = 0;
    }
}

}  // namespace pylda
