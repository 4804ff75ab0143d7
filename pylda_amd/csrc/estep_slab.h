// Register-resident E-step kernel ("slab"): the N_d x K tile of
// B = exp(E_log_eta - shift) lives in VGPRs for the whole inner loop.
//
// Why registers: on CDNA4 the vector register file (512 KiB per CU) is the
// largest on-chip memory, 3x the LDS, and a v_fma_f64 whose operands are all
// registers is the only form that can run at the fp64 FMA rate - an LDS
// operand costs 8 bytes of LDS bandwidth per FMA, twice what the CU has.
//
// One workgroup = one document = W wavefronts.
//   wavefront w   owns topics  [w*RK, (w+1)*RK)         (a "slab" of the tile)
//   lane l        owns words   l, l+64, ..., l+64*(RN-1)
//   => lane registers hold B[RN][RK]; t[k] of the wave's topics is broadcast
//      to every lane with v_readlane (wave-uniform operand of the FMA).
//
// One inner iteration (variational_bayes.py:177-190), exp-hoisted as in
// estep_generic.h, with t[k] = exp(psi(gamma_k) - psi(sum gamma)) so that no
// per-iteration max over topics is needed (sum gamma is invariant: sum alpha
// + #tokens):
//   A. p[n] = sum_{k in slab} B[n][k] t[k]      in-lane FMAs      -> LDS partial[w][n]
//      barrier
//   B. nrm[n] = sum_w partial[w][n];  r[n] = c[n] / nrm[n]
//      q[k] = sum_{n in lane} r[n] B[n][k]      in-lane FMAs
//      s[k] = sum_lanes q[k]                    2 permlane-swap levels + LDS transpose
//      gamma'_k = alpha_k + t[k] s[k]           every lane group of 64/RK lanes owns a topic
//      barrier (convergence decision, :187-190)
// Everything is summed in a fixed order: results are bitwise reproducible.
#pragma once
#include "estep_common.h"
#include "special_device.h"

namespace pylda {

template <int W, int RK, int RN>
struct SlabLds {
    static constexpr int kWords = kWave * RN;
    static constexpr size_t partial = 0;                                         // [2][W][kWords]
    static constexpr size_t red = partial + (size_t)2 * W * kWords * 8;          // [W][RK][17]
    static constexpr size_t chg = red + (size_t)W * RK * 17 * 8;                 // [2][W]
    static constexpr size_t misc = chg + (size_t)2 * W * 8 + 16;                 // [8][W]
    static constexpr size_t total = ((misc + (size_t)8 * W * 8) + 15) & ~(size_t)15;
};

// One document on the calling workgroup (W wavefronts); `smem` holds SlabLds<W, RK, RN>::total bytes.
template <int W, int RK, int RN>
__device__ __forceinline__ void slab_document(const EstepParams& p, const int doc, char* smem)
{
    using L = SlabLds<W, RK, RN>;
    constexpr int NT = kWave * W;
    constexpr int LP = kWave / RK;          // lanes that share one topic in the gamma update
    constexpr int Q = RK / 4;               // values per lane after the two swap levels
    static_assert(RK == 16 || RK == 32, "slab width");
    double* partial = reinterpret_cast<double*>(smem + L::partial);
    double* red = reinterpret_cast<double*>(smem + L::red);
    unsigned long long* chg = reinterpret_cast<unsigned long long*>(smem + L::chg);   // [2], 2^40 fixed point
    double* misc = reinterpret_cast<double*>(smem + L::misc);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid / kWave;
    const int K = p.K, ldk = p.ldk;
    const int64_t lo = p.doc_ptr[doc];
    const int N = (int)(p.doc_ptr[doc + 1] - lo);
    const int k0 = wave * RK;

    // ---- load the slab: this lane's words x this wave's topics ----
    double B[RN][RK];
    double cnt[RN];
    int wid[RN];
    double total = 0.0;
#pragma unroll
    for (int i = 0; i < RN; ++i) {
        const int n = lane + kWave * i;
        const bool live = n < N;
        wid[i] = live ? p.term_id[lo + n] : 0;
        cnt[i] = live ? (double)p.term_ct[lo + n] : 0.0;
        total += cnt[i];
        const double2* src = reinterpret_cast<const double2*>(p.expElog + (size_t)wid[i] * ldk + k0);
#pragma unroll
        for (int j = 0; j < RK / 2; ++j) {
            const double2 v = live ? src[j] : make_double2(0.0, 0.0);
            B[i][2 * j] = v.x;
            B[i][2 * j + 1] = v.y;
        }
    }
    total = wave_sum(total);                                              // :162 (every wave sees all words)

    // ---- per-topic state: lane group g = lane / LP owns topic k0 + g ----
    const int kt = k0 + lane / LP;
    const bool topic_live = kt < K;
    const double alpha_k = topic_live ? p.alpha[kt] : 1.0;
    double gam = alpha_k + total / K;                                     // :165
    // sum_k gamma_k is invariant under the update: sum alpha + #tokens
    double asum = 0.0;
    for (int k = lane; k < K; k += kWave) asum += p.alpha[k];
    asum = wave_sum(asum);
    const double psi_total = digamma(asum + total);

    double tv = 0.0, gam_prev = gam;
    if (tid == 0) chg[0] = chg[1] = 0ull;          // ordered before the first use by the loop's first barrier
    double r[RN], nrm[RN];
#pragma unroll
    for (int i = 0; i < RN; ++i) r[i] = nrm[i] = 0.0;
    int it = 0;
    int bad = 0;
    // moved * 2^-40 <= tol * K  <=>  moved <= floor(tol * K * 2^40): an integer compare on the fixed-point sum
    const double thresh_f = p.tol * K * kChangeScale;
    const long long thresh = !(thresh_f >= 0.0) ? -1ll : thresh_f >= 9.2e18 ? 0x7fffffffffffffffll : (long long)thresh_f;
    while (it < p.max_iter) {                                             // :174
        const int buf = it & 1;
        // t[k] = exp(psi(gamma_k) - psi(sum gamma))  (the constant cancels in :182)
        gam_prev = gam;
        tv = topic_live ? exp_digamma_minus(gam, psi_total) : 0.0;
        double ts[RK];
#pragma unroll
        for (int j = 0; j < RK; ++j) ts[j] = readlane_f64(tv, j * LP);

        // A. partial normalisers of this slab
        double* mine = partial + ((size_t)buf * W + wave) * L::kWords;
#pragma unroll
        for (int i = 0; i < RN; ++i) {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int j = 0; j < RK; j += 2) {
                a0 = fma(B[i][j], ts[j], a0);
                a1 = fma(B[i][j + 1], ts[j + 1], a1);
            }
            mine[lane + kWave * i] = a0 + a1;
        }
        __syncthreads();

        // B. full normalisers, r = count / nrm
        const double* all = partial + (size_t)buf * W * L::kWords;
#pragma unroll
        for (int i = 0; i < RN; ++i) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < W; ++w) s += all[(size_t)w * L::kWords + lane + kWave * i];
            const bool live = lane + kWave * i < N;
            if (live && !(s > 1e-280 && s < 1e300)) bad = 1;
            nrm[i] = s;
            r[i] = live ? cnt[i] * rcp_newton(s) : 0.0;
        }
        // q[k] = sum over this lane's words; s[k] = sum over lanes: two transposing
        // swap levels (topic m pairs with m + RK/2, then with m + RK/4) ...
        double u[RK / 2];
#pragma unroll
        for (int m = 0; m < RK / 2; ++m) {
            double a = r[0] * B[0][m], b = r[0] * B[0][m + RK / 2];
#pragma unroll
            for (int i = 1; i < RN; ++i) {
                a = fma(r[i], B[i][m], a);
                b = fma(r[i], B[i][m + RK / 2], b);
            }
            u[m] = swap32_add(a, b);
        }
        double v[Q];
#pragma unroll
        for (int m = 0; m < Q; ++m) v[m] = swap16_add(u[m], u[m + Q]);
        // ... lane (row r4 = lane/16, column c = lane%16) now holds, for m < Q, topic
        // m + (r4&1)*Q + (r4>>1)*RK/2 summed over the 4 lanes congruent to c mod 16;
        // finish through a (RK x 16, stride 17) LDS transpose.
        double* myred = red + (size_t)wave * RK * 17;
        {
            const int r4 = lane >> 4, c = lane & 15;
            const int tbase = (r4 & 1) * Q + (r4 >> 1) * (RK / 2);
#pragma unroll
            for (int m = 0; m < Q; ++m) myred[(tbase + m) * 17 + c] = v[m];
        }
        wave_lds_exchange();
        double s;
        {
            const int g = lane / LP, part = lane % LP;
            const double* src = myred + g * 17 + part * (16 / LP);
            s = src[0];
#pragma unroll
            for (int x = 1; x < 16 / LP; ++x) s += src[x];
            s = lane_group_sum<LP>(s);      // LP (2, 4 or 8) neighbouring lanes share a topic: DPP, no LDS
        }
        // gamma update, this lane group's topic
        const double gnew = fma(tv, s, alpha_k);                          // :185
        // sum_k |gamma' - gamma| as a 2^40 fixed-point LDS atomic: integer addition is associative,
        // so the stop decision is order-independent (and there is no 6-level wavefront reduction
        // on the serial path)
        if (topic_live && (lane % LP) == 0) atomicAdd(&chg[buf], change_fixed(fabs(gnew - gam)));   // :187
        gam = gnew;                                                       // :188
        if (tid == 0) chg[buf ^ 1] = 0ull;
        ++it;
        __syncthreads();
        if ((long long)chg[buf] <= thresh) break;                         // :189 (mean <= tol)
    }

    bad = __syncthreads_or(bad);
    if (bad) {
        if (!p.heldout) {      // contributes nothing to the gather pass; the log-space kernel adds it
            for (int n = tid; n < N; n += NT) p.rfinal[lo + n] = 0.0;
            for (int k = tid; k < ldk; k += NT) p.tfinal[(size_t)doc * ldk + k] = 0.0;
        }
        if (tid == 0) p.status[doc] = 1;
        return;
    }

    // ---- document terms (:195-204) with the last phi = B t r ----
    //   sum_n c_n sum_k phi log phi = sum_n r_n sum_k (B log B)[n][k] t_k
    //                               + sum_k log t_k (gamma_k - alpha_k) - sum_n c_n log nrm_n
    // (rows of phi sum to one; columns of phi*c sum to gamma - alpha).
    double ts[RK];
#pragma unroll
    for (int j = 0; j < RK; ++j) ts[j] = readlane_f64(tv, j * LP);
    double term1 = 0.0;
    const bool do_term1 = p.heldout || p.want_doc_ll;     // else: taken per corpus from the statistics
#pragma unroll
    for (int i = 0; i < RN; ++i) {
        if (lane + kWave * i < N && do_term1) {
            const double2* src = reinterpret_cast<const double2*>(p.expElog_elog + (size_t)wid[i] * ldk + k0);
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int j = 0; j < RK / 2; ++j) {
                const double2 g2 = src[j];
                a0 = fma(g2.x, ts[2 * j], a0);
                a1 = fma(g2.y, ts[2 * j + 1], a1);
            }
            term1 = fma(r[i], a0 + a1, term1);
        }
    }
    const bool owner = topic_live && (lane % LP) == 0;
    const double ltv = digamma(gam_prev) - psi_total;                     // log t of the last iteration
    double term2 = owner ? ltv * (gam - alpha_k) : 0.0;
    double lse_term = (owner && p.heldout) ? p.topic_lse[kt] * (gam - alpha_k) : 0.0;
    double lgam = owner ? lgamma_pos(gam) : 0.0;
    double gsum = owner ? gam : 0.0;
    double term3 = 0.0, shift_term = 0.0;
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < RN; ++i) {
            if (lane + kWave * i < N) {
                term3 = fma(cnt[i], log(nrm[i]), term3);
                if (p.heldout) shift_term = fma(cnt[i], p.shift[wid[i]], shift_term);
            }
        }
    }
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
    if (owner) p.gamma[(size_t)doc * K + kt] = gam;
    if (!p.heldout) {
        if ((lane % LP) == 0) p.tfinal[(size_t)doc * ldk + kt] = tv;     // padded topics: 0
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < RN; ++i)
                if (lane + kWave * i < N) p.rfinal[lo + lane + kWave * i] = r[i];
        }
        for (int k = W * RK + tid; k < ldk; k += NT) p.tfinal[(size_t)doc * ldk + k] = 0.0;
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

template <int W, int RK, int RN>
__global__ __launch_bounds__(kWave* W) void estep_slab_kernel(EstepParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    slab_document<W, RK, RN>(p, p.order[blockIdx.x], smem);
}

// Small corpora: every launch class of the slab family in ONE dispatch.  A corpus of a few thousand documents
// (associated-press: 2000 documents in five words-per-lane classes) cannot fill the chip; run as five kernels
// on five streams it pays the fork / join over the streams (2-3 x the slowest class, profiles/r01_ap_k10_summary.txt)
// for nothing.  The
// workgroup picks its instantiation from its position in the (longest first) schedule; the switch is uniform.
struct SlabUberClasses {
    int n;            // classes
    int first[7];     // class i holds workgroups first[i] .. first[i + 1] - 1
    int rn[6];        // ... and runs RN = rn[i] words per lane
};

// LONG: the 6-words-per-lane instantiation is part of the switch (16-topic slabs).  It needs more than 256 registers -
// one wavefront per SIMD for EVERY workgroup of the launch - so it is only compiled into the kernel used when a corpus
// has such documents (associated-press: 57 of 2000; in their own launch beside the others they cost a fork / join over
// two streams: document kernels 0.160 ms against 0.129 in one dispatch).
template <int W, int RK, bool LONG>
__global__ __launch_bounds__(kWave* W) void estep_slab_uber_kernel(EstepParams p, SlabUberClasses cls)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    int rn = cls.rn[0];
#pragma unroll
    for (int i = 1; i < 6; ++i)
        if (i < cls.n && b >= cls.first[i]) rn = cls.rn[i];
    const int doc = p.order[b];
    if constexpr (RK == 32) {
        if (rn == 1) slab_document<W, RK, 1>(p, doc, smem);
        else slab_document<W, RK, 2>(p, doc, smem);
    } else {
        switch (rn) {
        case 1: slab_document<W, RK, 1>(p, doc, smem); break;
        case 2: slab_document<W, RK, 2>(p, doc, smem); break;
        case 3: slab_document<W, RK, 3>(p, doc, smem); break;
        case 4: slab_document<W, RK, 4>(p, doc, smem); break;
        default:
            if constexpr (LONG) slab_document<W, RK, 6>(p, doc, smem);
            break;
        }
    }
}

}  // namespace pylda
