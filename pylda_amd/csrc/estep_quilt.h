// Register-resident E-step kernel, 2-D ("quilt") lane layout, for 32 < K <= 128.
//
// One workgroup = one document = W wavefronts; the N_d x K tile of
// B = exp(E_log_eta - shift) lives in VGPRs.  Within a wavefront the 64 lanes
// form a 4 x 16 grid:
//   lane = 16*g + c :  word group g (0..3)  x  topic lane c (0..15)
//   lane owns words   nb + g*RWL + i   (i < RWL)        (nb = first word of the wave)
//        and topics   2c + 32*jj + {0,1}   (jj < KRL/2; KRL = ldk/16 topics per lane)
//   => registers B[RWL][KRL]; a table row is read with 16-byte loads as
//      16-lane x 256-byte contiguous pieces (8-byte global loads run at 0.5-0.7x
//      the 16-byte rate on this part, and the tile gather is per-CU bandwidth bound).
//
// Why 2-D: each inner iteration needs two reductions across lanes,
//   nrm[n] = sum_k B[n][k] t[k]   (over topic lanes)   and
//   s[k]   = sum_n r[n] B[n][k]   (over word groups, then over wavefronts),
// and the cost of a cross-lane reduction is proportional to the number of values
// each lane carries into it.  The column kernel (estep_column.h) carried 32
// normaliser partials per lane through two permlane-swap levels (100 of its 260
// VALU instructions per wave-iteration); here the normalisers need only a 16-lane
// sum (one LDS transpose, no swaps) and the topic sums enter the swap levels with
// KRL = 8 values instead of 32.
//
// Iteration structure, barriers and the gamma phase are those of the column kernel.
#pragma once
#include "estep_common.h"
#include "special_device.h"
#include "estep_epilogue.h"

namespace pylda {

template <int W, int KRL, int RWL>
struct QuiltLds {
    static constexpr int kTopics = 16 * KRL;
    static constexpr int kRowsPerGroup = RWL <= 2 ? 2 : RWL <= 4 ? 4 : 8;          // RWL padded to a power of two
    // row stride of the normaliser transpose, in doubles: 16-byte aligned rows whose ds_read_b128
    // pattern below (lane pair / quad / octet of a word reads interleaved 16-byte pieces) is bank
    // conflict free for the instruction's 16-lane groups (MI355X_MICROARCH.md, LDS)
    static constexpr int kRedStride = kRowsPerGroup == 8 ? 20 : kRowsPerGroup == 4 ? 24 : 16;
    static constexpr size_t red = 0;                                               // [W][4*kRowsPerGroup][kRedStride]
    static constexpr size_t sp = red + (size_t)W * 4 * kRowsPerGroup * kRedStride * 8;   // [W][kTopics]
    static constexpr size_t tt = sp + (size_t)W * kTopics * 8;                     // [2][kTopics]
    static constexpr size_t chg = tt + (size_t)2 * kTopics * 8;                    // u64[2]
    static constexpr size_t misc = chg + 16;                                       // [8][W]
    static constexpr size_t total = (misc + (size_t)8 * W * 8 + 15) & ~(size_t)15;
};

template <int W, int KRL, int RWL>
__global__ __launch_bounds__(kWave* W) void estep_quilt_kernel(EstepParams p)
{
    using L = QuiltLds<W, KRL, RWL>;
    constexpr int NT = kWave * W;
    constexpr int KT = 16 * KRL;            // padded topic count (== ldk)
    constexpr int RNW = 4 * RWL;            // words per wavefront
    constexpr int LPW = 16 / L::kRowsPerGroup;   // lanes (of the word's own 16-lane row) that finish one normaliser
    constexpr int PER = 16 / LPW;           // partials each of them adds
    constexpr int QV = KRL / 4;             // topic values per lane after the swap levels
    static_assert(KRL == 4 || KRL == 8, "ldk 64 or 128");
    static_assert(RWL >= 2 && RWL <= 8, "words per lane");
    static_assert(KT <= NT, "one thread per topic in the gamma phase");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem + L::red);
    double* sp = reinterpret_cast<double*>(smem + L::sp);
    double* tt = reinterpret_cast<double*>(smem + L::tt);
    unsigned long long* chg = reinterpret_cast<unsigned long long*>(smem + L::chg);
    double* misc = reinterpret_cast<double*>(smem + L::misc);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int g = lane >> 4, c = lane & 15;
    const int K = p.K, ldk = p.ldk;
    const int doc = p.order[blockIdx.x];
    const int64_t lo = p.doc_ptr[doc];
    const int N = (int)(p.doc_ptr[doc + 1] - lo);
    const int nb = wave * RNW;
    const int wb = nb + g * RWL;            // first word of this lane

    // ---- small loads first: they must not queue behind the tile gather (vmcnt retires in order) ----
    int wid[RWL];
#pragma unroll
    for (int i = 0; i < RWL; ++i) wid[i] = wb + i < N ? p.term_id[lo + wb + i] : -1;
    // the word whose normaliser this lane finishes, one of its own row's: wb + c / LPW.  Its r then
    // reaches the 16 lanes that hold the word's tile entries by a DPP row broadcast, not through LDS.
    const int my_slot = c / LPW;
    const int my_word = wb + my_slot;
    const bool word_live = my_slot < RWL && my_word < N;
    const double my_cnt = word_live ? (double)p.term_ct[lo + my_word] : 0.0;
    double local = 0.0;
    for (int n = tid; n < N; n += NT) local += (double)p.term_ct[lo + n];
    double asum = 0.0;
    for (int k = lane; k < K; k += kWave) asum += p.alpha[k];
    const bool topic_thread = tid < KT;
    const bool topic_live = tid < K;
    const double alpha_k = topic_live ? p.alpha[tid] : 1.0;

    // ---- the tile gather: issued now, first needed in the inner loop; the set-up below (token
    // total, psi(sum gamma), first t) runs while it is in flight, which is why the two barriers
    // of the set-up are raw s_barrier + LDS-only waits (a __syncthreads would wait for the gather) ----
    double B[RWL][KRL];
#pragma unroll
    for (int i = 0; i < RWL; ++i) {
        if (wid[i] >= 0) {
            const double2* row = reinterpret_cast<const double2*>(p.expElog + (size_t)wid[i] * ldk) + c;
#pragma unroll
            for (int jj = 0; jj < KRL / 2; ++jj) {
                const double2 v2 = row[16 * jj];
                B[i][2 * jj] = v2.x;
                B[i][2 * jj + 1] = v2.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < KRL; ++j) B[i][j] = 0.0;
        }
    }

    // ---- total token count (:162) and the invariant sum_k gamma_k ----
    local = wave_sum(local);
    asum = wave_sum(asum);
    if (lane == 0) misc[wave] = local;
    if (tid == 0) chg[0] = chg[1] = 0ull;
    lds_only_barrier();
    double total = 0.0;
#pragma unroll
    for (int w = 0; w < W; ++w) total += misc[w];
    const double psi_total = digamma(asum + total);

    // ---- gamma phase state: thread k < KT owns topic k ----
    double gam = topic_live ? alpha_k + total / K : alpha_k;              // :165 (padding topics never move)
    double gam_prev = gam;
    double t_mine = 0.0;
    if (topic_thread) {
        t_mine = topic_live ? exp_digamma_minus(gam, psi_total) : 0.0;
        tt[tid] = t_mine;
    }
    lds_only_barrier();

    double r_mine = 0.0, nrm_mine = 1.0;
    int it = 0;
    int bad = 0;
    constexpr int RP = L::kRowsPerGroup, RS = L::kRedStride;
    double* myred = red + (size_t)wave * 4 * RP * RS + (size_t)g * RP * RS;        // this lane group's rows
    const double2* mysrc = reinterpret_cast<const double2*>(myred + my_slot * RS) + (c % LPW);
    // The stop test of iteration i (:189) is evaluated AFTER the first half of iteration i+1 has been
    // issued: the sum of |delta gamma| and the new t are requested together right behind the
    // barrier, the tile FMAs start as t arrives and the decision rides along (one LDS round trip
    // and a dependent compare less on the serial path; the speculative half iteration writes only
    // the transpose scratch).  The test itself is an integer compare on the fixed-point sum:
    // moved * 2^-40 <= tol * K  <=>  moved <= floor(tol * K * 2^40).
    const double thresh_f = p.tol * K * kChangeScale;
    const long long thresh = !(thresh_f >= 0.0) ? -1ll : thresh_f >= 9.2e18 ? 0x7fffffffffffffffll : (long long)thresh_f;
    long long moved = 0x7fffffffffffffffll;
    int left = p.max_iter;                  // iterations still allowed (counted down: no reload of the cap per trip)
    double tq[KRL];
#pragma unroll
    for (int jj = 0; jj < KRL / 2; ++jj) {
        const double2 t2 = reinterpret_cast<const double2*>(tt)[c + 16 * jj];
        tq[2 * jj] = t2.x;
        tq[2 * jj + 1] = t2.y;
    }
#pragma unroll
    for (int j = 0; j < KRL; ++j) asm volatile("" : "+v"(tq[j]));   // (keeps the in-loop reads at the loop's end: the
                                                                     //  optimiser would merge both sets at the loop head)
    for (;;) {                                                            // :174
        const int buf = it & 1;

        // A. partial normalisers over this lane's topics -> LDS transpose -> sum over the 16 topic lanes
#pragma unroll
        for (int i = 0; i < RWL; ++i) {
            double a0 = B[i][0] * tq[0];                // one chain per word: RWL independent chains
#pragma unroll
            for (int j = 1; j < KRL; ++j) a0 = fma(B[i][j], tq[j], a0);
            myred[i * RS + c] = a0;
        }
        if (moved <= thresh || left <= 0) break;                          // :189 (mean <= tol), :174
        wave_lds_exchange();
        {
            double2 s2 = mysrc[0];
#pragma unroll
            for (int x = 1; x < PER / 2; ++x) {
                const double2 v2 = mysrc[x * LPW];
                s2.x += v2.x;
                s2.y += v2.y;
            }
            const double s0 = s2.x, s1 = s2.y;
            const double s = lane_group_sum<LPW>(s0 + s1);
            nrm_mine = s;
            if (word_live && !(s > 1e-280 && s < 1e300)) bad = 1;
            r_mine = word_live ? my_cnt * rcp_newton(s) : 0.0;
        }

        // B. q[k] over this lane's words, then over the 4 word groups (two swap levels)
        double q[KRL];
        row_bcast_matvec<RWL, LPW>(q, r_mine, B);      // r of word i sits in lane i*LPW of this lane's row
        double u[KRL / 2];
#pragma unroll
        for (int m = 0; m < KRL / 2; ++m) u[m] = swap32_add(q[m], q[m + KRL / 2]);
        // lane (row g, column c) now holds, for m < QV, topic slot m + (g&1)*QV + (g>>1)*KRL/2
#pragma unroll
        for (int m = 0; m < QV; ++m) {
            const double v = swap16_add(u[m], u[m + QV]);
            const int slot = m + (g & 1) * QV + (g >> 1) * (KRL / 2);      // register index j of the topic
            sp[wave * KT + 2 * c + (slot & 1) + 32 * (slot >> 1)] = v;
        }
        // both coefficient tables of exp_digamma_minus_levels, requested ahead of the barrier (estep_quad.h)
        ExpDigammaLevelsA coef_a;
        ExpDigammaLevelsB coef_b;
        if (topic_thread) {
            coef_a.load();
            coef_b.load();
        }
        __syncthreads();

        // C. gamma update by the topic threads
        if (topic_thread) {
            double part[W];
#pragma unroll
            for (int w = 0; w < W; ++w) part[w] = sp[w * KT + tid];
            keep_together(part);            // all W reads in flight at once (one LDS round trip, not W/2)
            double s0 = part[0], s1 = part[1];
#pragma unroll
            for (int w = 2; w < W; w += 2) {
                s0 += part[w];
                s1 += part[w + 1];
            }
            const double gnew = fma(t_mine, s0 + s1, alpha_k);            // :185
            const double diff = fabs(gnew - gam);                         // :187
            gam_prev = gam;
            gam = gnew;                                                   // :188
            atomicAdd(&chg[buf], change_fixed(diff));
            const double t_next = exp_digamma_minus_levels<true>(gam, psi_total, coef_a, &coef_b);
            t_mine = topic_live ? t_next : 0.0;
            tt[(buf ^ 1) * KT + tid] = t_mine;
            if (tid == 0) store_u64_hi(&chg[buf ^ 1], 0u);
        }
        ++it;
        --left;
        __syncthreads();
        moved = (long long)chg[buf];
#pragma unroll
        for (int jj = 0; jj < KRL / 2; ++jj) {
            const double2 t2 = reinterpret_cast<const double2*>(tt + (buf ^ 1) * KT)[c + 16 * jj];
            tq[2 * jj] = t2.x;
            tq[2 * jj + 1] = t2.y;
        }
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

    // ---- training fast path: the document terms are left to doc_terms_kernel (doc_terms.h; see estep_quad.h) ----
    if (!p.heldout && !p.want_doc_ll) {
        if (word_live && (c % LPW) == 0) p.rfinal[lo + my_word] = r_mine;
        if (topic_thread) {
            if (topic_live) p.gamma[(size_t)doc * K + tid] = gam;
            p.tfinal[(size_t)doc * ldk + tid] = topic_live ? tt[last * KT + tid] : 0.0;
        }
        if (tid == 0) {
            p.iters[doc] = it;
            p.status[doc] = 3;
        }
        return;
    }

    // ---- document terms (:195-204) with the last phi = B t r (see estep_slab.h) ----
#pragma unroll
    for (int jj = 0; jj < KRL / 2; ++jj) {
        const double2 t2 = reinterpret_cast<const double2*>(tt + last * KT)[c + 16 * jj];
        tq[2 * jj] = t2.x;
        tq[2 * jj + 1] = t2.y;
    }
    double term1 = 0.0;
    double rl[RWL];
    row_bcast_all<RWL, LPW>(r_mine, rl);
    const bool do_term1 = p.heldout || p.want_doc_ll;     // else: taken per corpus from the statistics
#pragma unroll
    for (int i = 0; i < RWL; ++i) {
        const int n = wb + i;
        if (n < N && do_term1) {
            const double2* row = reinterpret_cast<const double2*>(p.expElog_elog + (size_t)p.term_id[lo + n] * ldk) + c;
            double gsum2 = 0.0;
#pragma unroll
            for (int jj = 0; jj < KRL / 2; ++jj) {
                const double2 g2 = row[16 * jj];
                gsum2 = fma(g2.y, tq[2 * jj + 1], fma(g2.x, tq[2 * jj], gsum2));
            }
            term1 = fma(rl[i], gsum2, term1);
        }
    }
    const bool word_owner = word_live && (c % LPW) == 0;
    double term3 = word_owner ? my_cnt * log(nrm_mine) : 0.0;
    double shift_term = (word_owner && p.heldout) ? my_cnt * p.shift[p.term_id[lo + my_word]] : 0.0;
    TopicShare share;
    if (topic_thread)
        topic_share(p, doc, tid, ldk, topic_live, true, gam, alpha_k, gam_prev, tt[last * KT + tid], psi_total, share);
    if (word_owner && !p.heldout) p.rfinal[lo + my_word] = r_mine;
    finish_document<W>(p, doc, it, misc, lane, wave, tid, term1, term3, shift_term, share);
}

}  // namespace pylda
