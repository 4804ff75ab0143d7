// Streaming E-step kernel for tiles that do not fit on chip: 128 < K <= 512, or
// documents with more distinct terms than the register kernels hold.
//
// The N_d x K tile of B = exp(E_log_eta - shift) at K = 256, N_d ~ 200 is 400 KB -
// more than the vector registers plus LDS of a CU - so it is re-read from the
// table (L2 / Infinity Cache; the tables are tens of MB) twice per inner
// iteration, with the lane layout of the quilt kernel so that every read is a
// 16-lane x 256-byte contiguous piece and both cross-lane reductions stay cheap:
//
//   lane = 16*g + c :  word slot g (0..3) x topic lane c (0..15)
//   lane owns topics 2c + 32*jj + {0,1}, jj < KRL/2   (KRL = ldk/16 per lane)
//   a wavefront walks its words four at a time (one per slot g).
//
// Per inner iteration (variational_bayes.py:177-190, exp-hoisted):
//   A. for each 4-word chunk: load rows, p = sum_k B t (in-lane) -> LDS partial[word][c]
//      per wavefront: nrm[n] = sum_c partial (lane <-> word), r[n] = c_n / nrm[n] -> LDS
//   B. for each chunk: load rows again, q[k] += r[n] B[n][k]; reduce q over the 4
//      slots (two permlane-swap levels) -> LDS partial per wavefront; barrier
//   C. K topic threads: gamma'_k, convergence sum (fixed-point LDS atomic), next t; barrier
// Traffic: 2 * N_d * K * 8 B per iteration from L2 - this kernel is L2-bandwidth
// bound (about 2x the fp64 time at K = 256), which is why the register kernels
// exist for K <= 128.
#pragma once
#include "estep_common.h"
#include "special_device.h"

namespace pylda {

constexpr int kQsMaxWordsPerWave = 128;     // 8 wavefronts => documents up to 1024 distinct terms
constexpr int kQsSpan = 32;                 // words whose normalisers are finished together

template <int W, int KRL>
struct QstreamLds {
    static constexpr int kTopics = 16 * KRL;
    static constexpr size_t red = 0;                                                  // [W][kQsSpan][17]
    static constexpr size_t rr = red + (size_t)W * kQsSpan * 17 * 8;                  // [W][kQsMaxWordsPerWave]
    static constexpr size_t nrm = rr + (size_t)W * kQsMaxWordsPerWave * 8;            // [W][kQsMaxWordsPerWave]
    static constexpr size_t sp = nrm + (size_t)W * kQsMaxWordsPerWave * 8;            // [W][kTopics]
    static constexpr size_t tt = sp + (size_t)W * kTopics * 8;                        // [2][kTopics]
    static constexpr size_t ids = tt + (size_t)2 * kTopics * 8;                       // int [W][kQsMaxWordsPerWave]
    static constexpr size_t chg = ids + (size_t)W * kQsMaxWordsPerWave * 4;           // u64[2]
    static constexpr size_t misc = chg + 16;                                          // [8][W]
    static constexpr size_t total = (misc + (size_t)8 * W * 8 + 15) & ~(size_t)15;
};

// Up to K = 256 the kernel fits 128 VGPRs and ~75 KB of LDS: two documents per CU, which is
// what hides the L2 latency of the row stream.
template <int W, int KRL>
__global__ __launch_bounds__(kWave* W, (KRL <= 16 ? 4 : 2)) void estep_qstream_kernel(EstepParams p)
{
    using L = QstreamLds<W, KRL>;
    constexpr int NT = kWave * W;
    constexpr int KT = 16 * KRL;            // padded topic count (== ldk)
    constexpr int QV = KRL / 4;
    static_assert(KRL % 4 == 0 && KRL >= 4 && KRL <= 32, "ldk a multiple of 64, at most 512");
    static_assert(KT <= NT, "one thread per topic in the gamma phase");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem + L::red);
    double* rr = reinterpret_cast<double*>(smem + L::rr);
    double* nrmv = reinterpret_cast<double*>(smem + L::nrm);
    double* sp = reinterpret_cast<double*>(smem + L::sp);
    double* tt = reinterpret_cast<double*>(smem + L::tt);
    int* ids = reinterpret_cast<int*>(smem + L::ids);
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
    // words are dealt to wavefronts in contiguous blocks of NW (a multiple of 4)
    const int NW = ((N + W - 1) / W + 3) & ~3;
    const int nb = wave * NW;                             // first word of this wavefront
    const int nmine = max(0, min(NW, N - nb));            // its live words
    double* myred = red + (size_t)wave * kQsSpan * 17;
    double* myrr = rr + wave * kQsMaxWordsPerWave;
    double* mynrm = nrmv + wave * kQsMaxWordsPerWave;
    int* myids = ids + wave * kQsMaxWordsPerWave;

    // ---- stage ids, counts (as r numerators later), token total (:162) ----
    double local = 0.0;
    for (int i = lane; i < NW; i += kWave) {
        const bool live = i < nmine;
        myids[i] = live ? p.term_id[lo + nb + i] : 0;
        local += live ? (double)p.term_ct[lo + nb + i] : 0.0;
    }
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

    const double2* table = reinterpret_cast<const double2*>(p.expElog);
    const int ldk2 = ldk / 2;
    int it = 0;
    int bad = 0;
    while (it < p.max_iter) {                                             // :174
        const int buf = it & 1;
        double tq[KRL];
#pragma unroll
        for (int jj = 0; jj < KRL / 2; ++jj) {
            const double2 t2 = reinterpret_cast<const double2*>(tt + buf * KT)[c + 16 * jj];
            tq[2 * jj] = t2.x;
            tq[2 * jj + 1] = t2.y;
        }
        // A. normaliser partials, kQsSpan words at a time, then lane <-> word sums
        for (int base = 0; base < NW; base += kQsSpan) {
            const int span = min(kQsSpan, NW - base);
            for (int off = 0; off < span; off += 4) {
                const int i = base + off + g;
                const double2* row = table + (size_t)myids[i] * ldk2 + c;
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int jj = 0; jj < KRL / 2; ++jj) {
                    const double2 b2 = row[16 * jj];
                    a0 = fma(b2.x, tq[2 * jj], a0);
                    a1 = fma(b2.y, tq[2 * jj + 1], a1);
                }
                myred[(off + g) * 17 + c] = i < nmine ? a0 + a1 : 0.0;
            }
            wave_lds_exchange();
            if (lane < span) {
                const double* src = myred + lane * 17;
                double s0 = src[0], s1 = src[1];
#pragma unroll
                for (int x = 2; x < 16; x += 2) {
                    s0 += src[x];
                    s1 += src[x + 1];
                }
                const double s = s0 + s1;
                const int i = base + lane;
                const bool live = i < nmine;
                if (live && !(s > 1e-280 && s < 1e300)) bad = 1;
                const double cnt = live ? (double)p.term_ct[lo + nb + i] : 0.0;
                mynrm[i] = s;
                myrr[i] = live ? cnt * rcp_newton(s) : 0.0;
            }
            wave_lds_exchange();
        }
        // B. q[k] = sum over this lane's words (rows re-read), then over the 4 slots
        double q[KRL];
#pragma unroll
        for (int j = 0; j < KRL; ++j) q[j] = 0.0;
        for (int off = 0; off < NW; off += 4) {
            const int i = off + g;
            const double rn = myrr[i];
            const double2* row = table + (size_t)myids[i] * ldk2 + c;
#pragma unroll
            for (int jj = 0; jj < KRL / 2; ++jj) {
                const double2 b2 = row[16 * jj];
                q[2 * jj] = fma(rn, b2.x, q[2 * jj]);
                q[2 * jj + 1] = fma(rn, b2.y, q[2 * jj + 1]);
            }
        }
        double u[KRL / 2];
#pragma unroll
        for (int m = 0; m < KRL / 2; ++m) u[m] = swap32_add(q[m], q[m + KRL / 2]);
#pragma unroll
        for (int m = 0; m < QV; ++m) {
            const double v = swap16_add(u[m], u[m + QV]);
            const int slot = m + (g & 1) * QV + (g >> 1) * (KRL / 2);      // register index j of the topic
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
        double tq[KRL];
#pragma unroll
        for (int jj = 0; jj < KRL / 2; ++jj) {
            const double2 t2 = reinterpret_cast<const double2*>(tt + last * KT)[c + 16 * jj];
            tq[2 * jj] = t2.x;
            tq[2 * jj + 1] = t2.y;
        }
        const double2* gtable = reinterpret_cast<const double2*>(p.expElog_elog);
        for (int off = 0; off < NW; off += 4) {
            const int i = off + g;
            const double2* row = gtable + (size_t)myids[i] * ldk2 + c;
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int jj = 0; jj < KRL / 2; ++jj) {
                const double2 g2 = row[16 * jj];
                a0 = fma(g2.x, tq[2 * jj], a0);
                a1 = fma(g2.y, tq[2 * jj + 1], a1);
            }
            term1 = fma(myrr[i], a0 + a1, term1);          // r = 0 for padding words
        }
    }
    double term3 = 0.0, shift_term = 0.0;
    for (int i = lane; i < nmine; i += kWave) {
        const double cnt = (double)p.term_ct[lo + nb + i];
        term3 = fma(cnt, log(mynrm[i]), term3);
        if (p.heldout) shift_term = fma(cnt, p.shift[myids[i]], shift_term);
        if (!p.heldout) p.rfinal[lo + nb + i] = myrr[i];
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
