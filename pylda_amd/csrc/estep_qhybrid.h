// Hybrid E-step kernel for 128 < K <= 256 (cfg 4: K = 256, N_d ~ 200, tile 400 KB):
// the tile is larger than a CU's registers + LDS, so it is split in three tiers
//
//   tier R   the first 32*RWL words         in VGPRs    (quilt layout, estep_quilt.h)
//   tier L   the next  NL words             in LDS      (whole rows, loaded once per document)
//   tier S   whatever is left               streamed from L2 / Infinity Cache twice per iteration
//                                           (estep_qstream.h)
//
// Tail words (tiers L and S) are processed FUSED, four at a time: a row is loaded into
// registers once per inner iteration (from LDS or from the table), used for the normaliser
// partial, the 16-lane sum goes through a tiny LDS exchange, and the same registers then
// feed the topic sums - one row read and one memory round trip per word per iteration.
//
// Every tier uses the same lane grid (lane = 16*g + c: word slot g, topic lane c, topics
// 2c + 32*jj + {0,1}), so the two reductions and the gamma phase are shared.  The fully
// streamed kernel moves 2*N_d*K*8 B per inner iteration (3.8 TB per outer iteration at
// cfg 4 - Infinity-Cache / HBM bound, 5.1 s); keeping ~85 % of the words on chip cuts
// that ~6x, which is about where the fp64 work takes over.
#pragma once
#include "estep_common.h"
#include "special_device.h"
#include "estep_limits.h"

namespace pylda {

constexpr int kQhSpan = 16;                 // tier L/S words whose normalisers are finished together (>= 4*RWL)

template <int W, int KRL, int RWL>
struct QhybridLds {
    static constexpr int kTopics = 16 * KRL;
    static constexpr int kRowDoubles = kTopics + 2;                                    // +16 B: rows of a group differ in bank
    static constexpr size_t red = 0;                                                   // [W][kQhSpan][17] (tier R uses 4*RWL rows)
    static constexpr size_t rr = red + (size_t)W * kQhSpan * 17 * 8;                   // [W][4*RWL + kQhMaxTail]
    static constexpr size_t nrm = rr + (size_t)W * (4 * RWL + kQhMaxTail) * 8;         // [W][kQhMaxTail]
    static constexpr size_t cnt = nrm + (size_t)W * kQhMaxTail * 8;                    // [W][kQhMaxTail]
    static constexpr size_t sp = cnt + (size_t)W * kQhMaxTail * 8;                     // [W][kTopics]
    static constexpr size_t tt = sp + (size_t)W * kTopics * 8;                         // [2][kTopics]
    static constexpr size_t ids = tt + (size_t)2 * kTopics * 8;                        // int [W][kQhMaxTail]
    static constexpr size_t chg = ids + (size_t)W * kQhMaxTail * 4;                    // u64[2]
    static constexpr size_t misc = chg + 16;                                           // [8][W]
    static constexpr size_t rows = (misc + (size_t)8 * W * 8 + 15) & ~(size_t)15;      // tier L rows start here
    static constexpr size_t fixed_total = rows;
    static constexpr int rows_that_fit(size_t lds_limit)
    {
        return lds_limit > fixed_total ? (int)((lds_limit - fixed_total) / ((size_t)kRowDoubles * 8)) : 0;
    }
};

template <int W, int KRL, int RWL>
__global__ __launch_bounds__(kWave* W) void estep_qhybrid_kernel(EstepParams p, int lds_rows_per_wave)
{
    using L = QhybridLds<W, KRL, RWL>;
    constexpr int NT = kWave * W;
    constexpr int KT = 16 * KRL;
    constexpr int RNW = 4 * RWL;            // tier R words per wavefront
    constexpr int LPW = kWave / RNW;
    constexpr int PER = 16 / LPW;
    constexpr int QV = KRL / 4;
    constexpr int ROW = L::kRowDoubles;
    static_assert(KRL % 4 == 0 && KRL >= 4 && KRL <= 16, "ldk a multiple of 64, at most 256");
    static_assert(RWL == 2 || RWL == 4, "tier R words per lane");
    static_assert(KT <= NT, "one thread per topic in the gamma phase");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem + L::red);
    double* rr = reinterpret_cast<double*>(smem + L::rr);
    double* nrmv = reinterpret_cast<double*>(smem + L::nrm);
    double* cntv = reinterpret_cast<double*>(smem + L::cnt);
    double* sp = reinterpret_cast<double*>(smem + L::sp);
    double* tt = reinterpret_cast<double*>(smem + L::tt);
    int* ids = reinterpret_cast<int*>(smem + L::ids);
    unsigned long long* chg = reinterpret_cast<unsigned long long*>(smem + L::chg);
    double* misc = reinterpret_cast<double*>(smem + L::misc);
    double* rows = reinterpret_cast<double*>(smem + L::rows);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int g = lane >> 4, c = lane & 15;
    const int K = p.K, ldk = p.ldk;
    const int doc = p.order[blockIdx.x];
    const int64_t lo = p.doc_ptr[doc];
    const int N = (int)(p.doc_ptr[doc + 1] - lo);

    // ---- word assignment ----
    // tier R: wave w owns words [w*RNW, (w+1)*RNW); tail (tiers L, S): words >= W*RNW are dealt
    // to the wavefronts in contiguous blocks of NTW (a multiple of 4); the first NLW of a
    // wavefront's tail words live in LDS, the rest are streamed.
    const int nbR = wave * RNW;
    const int wbR = nbR + g * RWL;
    const int tail = max(0, N - W * RNW);
    const int NTW = ((tail + W - 1) / W + 3) & ~3;
    const int nbT = W * RNW + wave * NTW;
    const int nmineT = max(0, min(NTW, N - nbT));
    const int NLW = min(NTW, lds_rows_per_wave);
    double* myred = red + (size_t)wave * kQhSpan * 17;
    double* myrr = rr + wave * (RNW + kQhMaxTail);        // [0, RNW): tier R, then the tail
    double* myrrT = myrr + RNW;
    double* mynrmT = nrmv + wave * kQhMaxTail;
    double* mycntT = cntv + wave * kQhMaxTail;
    int* myidsT = ids + wave * kQhMaxTail;
    double* myrows = rows + (size_t)wave * lds_rows_per_wave * ROW;
    const double2* table = reinterpret_cast<const double2*>(p.expElog);
    const int ldk2 = ldk / 2;

    // ---- tier R: registers ----
    double B[RWL][KRL];
#pragma unroll
    for (int i = 0; i < RWL; ++i) {
        const int n = wbR + i;
        if (n < N) {
            const double2* row = table + (size_t)p.term_id[lo + n] * ldk2 + c;
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
    const int my_word = nbR + lane / LPW;
    const bool word_live = my_word < N;
    const double my_cnt = word_live ? (double)p.term_ct[lo + my_word] : 0.0;

    // ---- tail ids, tier L rows, token total (:162) ----
    double local = 0.0;
    for (int i = lane; i < NTW; i += kWave) {
        myidsT[i] = i < nmineT ? p.term_id[lo + nbT + i] : 0;
        mycntT[i] = i < nmineT ? (double)p.term_ct[lo + nbT + i] : 0.0;
    }
    for (int n = tid; n < N; n += NT) local += (double)p.term_ct[lo + n];
    local = wave_sum(local);
    wave_lds_exchange();
    for (int off = 0; off < NLW; off += 4) {
        const int i = off + g;
        if (i >= NLW) continue;
        const double2* src = table + (size_t)myidsT[i] * ldk2 + c;
        double2* dst = reinterpret_cast<double2*>(myrows + (size_t)i * ROW) + c;
#pragma unroll
        for (int jj = 0; jj < KRL / 2; ++jj) dst[16 * jj] = src[16 * jj];
    }
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

    // one tail word's partial normaliser / topic sums, from an LDS row or a table row
    auto tail_pass_a = [&](const double2* row, const double2* tq2) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int jj = 0; jj < KRL / 2; ++jj) {
            const double2 b2 = row[16 * jj];
            const double2 t2 = tq2[16 * jj];
            a0 = fma(b2.x, t2.x, a0);
            a1 = fma(b2.y, t2.y, a1);
        }
        return a0 + a1;
    };

    double r_mine = 0.0, nrm_mine = 1.0;
    int it = 0;
    int bad = 0;
    while (it < p.max_iter) {                                             // :174
        const int buf = it & 1;
        const double2* tq2 = reinterpret_cast<const double2*>(tt + buf * KT) + c;   // t is re-read from LDS where used

        // A(R). tier R partial normalisers -> LDS transpose -> sums, r
#pragma unroll
        for (int i = 0; i < RWL; ++i) {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int jj = 0; jj < KRL / 2; ++jj) {
                const double2 t2 = tq2[16 * jj];
                a0 = fma(B[i][2 * jj], t2.x, a0);
                a1 = fma(B[i][2 * jj + 1], t2.y, a1);
            }
            myred[(g * RWL + i) * 17 + c] = a0 + a1;
        }
        wave_lds_exchange();
        {
            const int part = lane % LPW;
            const double* src = myred + (lane / LPW) * 17 + part * PER;
            double s0 = src[0], s1 = src[1];
#pragma unroll
            for (int x = 2; x < PER; x += 2) {
                s0 += src[x];
                s1 += src[x + 1];
            }
            double s = s0 + s1;
#pragma unroll
            for (int m = 1; m < LPW; m <<= 1) s += __shfl_xor(s, m, kWave);
            nrm_mine = s;
            if (word_live && !(s > 1e-280 && s < 1e300)) bad = 1;
            r_mine = word_live ? my_cnt * rcp_newton(s) : 0.0;
            if (part == 0) myrr[lane / LPW] = r_mine;
        }
        wave_lds_exchange();

        // B. q[k] over tier R (registers), then the tail rows
        double q[KRL];
        {
            const double* rsrc = myrr + g * RWL;
            const double r0 = rsrc[0], r1 = rsrc[1];
#pragma unroll
            for (int j = 0; j < KRL; ++j) q[j] = fma(r1, B[1][j], r0 * B[0][j]);
#pragma unroll
            for (int i = 2; i < RWL; ++i) {
                const double ri = rsrc[i];
#pragma unroll
                for (int j = 0; j < KRL; ++j) q[j] = fma(ri, B[i][j], q[j]);
            }
        }
        // tail words, fused: row -> registers -> normaliser partial -> 16-lane sum -> r -> topic sums
        for (int off = 0; off < NTW; off += 4) {
            const int i = off + g;
            const double2* row = i < NLW ? reinterpret_cast<const double2*>(myrows + (size_t)i * ROW) + c
                                         : table + (size_t)myidsT[i] * ldk2 + c;
            double2 rowv[KRL / 2];
#pragma unroll
            for (int jj = 0; jj < KRL / 2; ++jj) rowv[jj] = row[16 * jj];
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int jj = 0; jj < KRL / 2; ++jj) {
                const double2 t2 = tq2[16 * jj];
                a0 = fma(rowv[jj].x, t2.x, a0);
                a1 = fma(rowv[jj].y, t2.y, a1);
            }
            myred[g * 17 + c] = a0 + a1;
            wave_lds_exchange();
            const double* src = myred + g * 17;              // the 16 partials of this lane group's word
            double s0 = src[0], s1 = src[1], s2 = src[2], s3 = src[3];
#pragma unroll
            for (int x = 4; x < 16; x += 4) {
                s0 += src[x];
                s1 += src[x + 1];
                s2 += src[x + 2];
                s3 += src[x + 3];
            }
            const double sn = (s0 + s1) + (s2 + s3);
            const bool live = i < nmineT;
            if (live && !(sn > 1e-280 && sn < 1e300)) bad = 1;
            const double rn = live ? mycntT[i] * rcp_newton(sn) : 0.0;
            if (c == 0) {
                mynrmT[i] = sn;
                myrrT[i] = rn;
            }
#pragma unroll
            for (int jj = 0; jj < KRL / 2; ++jj) {
                q[2 * jj] = fma(rn, rowv[jj].x, q[2 * jj]);
                q[2 * jj + 1] = fma(rn, rowv[jj].y, q[2 * jj + 1]);
            }
            __builtin_amdgcn_wave_barrier();                 // the next chunk reuses myred
        }
        double u[KRL / 2];
#pragma unroll
        for (int m = 0; m < KRL / 2; ++m) u[m] = swap32_add(q[m], q[m + KRL / 2]);
#pragma unroll
        for (int m = 0; m < QV; ++m) {
            const double v = swap16_add(u[m], u[m + QV]);
            const int slot = m + (g & 1) * QV + (g >> 1) * (KRL / 2);
            sp[wave * KT + 2 * c + (slot & 1) + 32 * (slot >> 1)] = v;
        }
        __syncthreads();

        // C. gamma update by the topic threads
        if (topic_thread) {
            double s0 = sp[tid], s1 = sp[KT + tid];
#pragma unroll
            for (int w = 2; w < W; w += 2) {
                s0 += sp[w * KT + tid];
                s1 += sp[(w + 1) * KT + tid];
            }
            const double gnew = fma(t_mine, s0 + s1, alpha_k);            // :185
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
    const int last = (it - 1) & 1;

    bad = __syncthreads_or(bad);
    if (bad) {
        if (!p.heldout) {
            for (int n = tid; n < N; n += NT) p.rfinal[lo + n] = 0.0;
            for (int k = tid; k < ldk; k += NT) p.tfinal[(size_t)doc * ldk + k] = 0.0;
        }
        if (tid == 0) p.status[doc] = 1;
        return;
    }

    // ---- document terms (:195-204) with the last phi = B t r (identities: estep_slab.h) ----
    double term1 = 0.0;
    if (p.heldout || p.want_doc_ll) {
        const double2* tq2 = reinterpret_cast<const double2*>(tt + last * KT) + c;
        const double2* gtable = reinterpret_cast<const double2*>(p.expElog_elog);
#pragma unroll
        for (int i = 0; i < RWL; ++i) {
            const int n = wbR + i;
            if (n < N)
                term1 = fma(myrr[g * RWL + i], tail_pass_a(gtable + (size_t)p.term_id[lo + n] * ldk2 + c, tq2), term1);
        }
        for (int off = 0; off < NTW; off += 4) {
            const int i = off + g;
            term1 = fma(myrrT[i], tail_pass_a(gtable + (size_t)myidsT[i] * ldk2 + c, tq2), term1);
        }
    }
    const bool word_owner = word_live && (lane % LPW) == 0;
    double term3 = word_owner ? my_cnt * log(nrm_mine) : 0.0;
    double shift_term = (word_owner && p.heldout) ? my_cnt * p.shift[p.term_id[lo + my_word]] : 0.0;
    if (word_owner && !p.heldout) p.rfinal[lo + my_word] = r_mine;
    for (int i = lane; i < nmineT; i += kWave) {
        const double cnt = mycntT[i];
        term3 = fma(cnt, log(mynrmT[i]), term3);
        if (p.heldout) shift_term = fma(cnt, p.shift[myidsT[i]], shift_term);
        if (!p.heldout) p.rfinal[lo + nbT + i] = myrrT[i];
    }
    double term2 = 0.0, lse_term = 0.0, lgam = 0.0, gsum = 0.0;
    if (topic_live) {
        const double t_last = tt[last * KT + tid];
        const double moved = gam - alpha_k;
        const double ltv = digamma(gam_prev) - psi_total;
        term2 = ltv * moved;
        if (p.heldout) lse_term = p.topic_lse[tid] * moved;
        lgam = lgamma_pos(gam);
        gsum = gam;
        p.gamma[(size_t)doc * K + tid] = gam;
        if (!p.heldout) p.tfinal[(size_t)doc * ldk + tid] = t_last;
    } else if (topic_thread && !p.heldout) {
        p.tfinal[(size_t)doc * ldk + tid] = 0.0;
    }
    term1 = wave_sum(term1);
    term2 = wave_sum(term2);
    lse_term = wave_sum(lse_term);
    lgam = wave_sum(lgam);
    gsum = wave_sum(gsum);
    term3 = wave_sum(term3);
    shift_term = wave_sum(shift_term);
    __syncthreads();
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
