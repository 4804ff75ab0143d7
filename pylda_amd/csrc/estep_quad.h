// Register-resident E-step kernel with FOUR wavefronts per document, two documents per CU
// (K <= 128, N_d <= 16 * (RWL + TWL)).
//
// The 8-wavefront quilt kernel (estep_quilt.h) fills a CU's register file with ONE document, and
// an inner iteration is a serial chain (normalisers -> r -> topic sums -> cross-wavefront
// reduction -> gamma / digamma / exp -> t), so its LDS round trips, its two barriers and the
// gamma phase (2 of 8 wavefronts busy) leave the fp64 pipes idle more than half of the time
// (round-1 counters: 44 % of wave-cycles waiting).  Nothing of the same document can fill those
// gaps; another document can.  Here a document is a 256-thread workgroup whose wavefronts each
// sit on a different SIMD, at <= 256 VGPRs, so TWO documents are co-resident per CU and every SIMD
// alternates between them: one document's gamma phase and exchanges hide under the other's FMAs.
//
// Lane layout inside a wavefront is the quilt's 4 x 16 grid (lane = 16*g + c: word group g, topic
// lane c, topics 2c + 32*jj + {0,1}); a document has 16 word groups gg = 4*wave + g and word n
// belongs to group n % 16, slot n / 16.  Slots 0 .. RWL-1 of a group live in VGPRs (RWL = 10: 160
// words, 160 VGPRs), slots RWL .. RWL+TWL-1 as whole rows in LDS (16 * TWL KiB at K = 128: with
// TWL <= 3 two workgroups fit a CU's 160 KiB).  LDS rows are read twice per iteration (normaliser
// pass, topic-sum pass) as conflict-free ds_read_b128 (row stride 1 KiB = 0 mod 256 B: the 16 lanes
// an instruction services together read 16 different 16-byte slots).
//
// Normalisers: as in the quilt kernel, partial sums over a lane's 8 topics go through an LDS
// transpose of 8 rows per word group and a lane pair finishes each word - done twice per
// iteration (slots 0-7, then slots 8 .. RWL+TWL-1) through the same 5 KiB-per-wavefront buffer
// (the LDS executes a wavefront's DS instructions in order, so the second set of writes may be
// issued right behind the first set of reads).  r reaches the 16 lanes holding a word's tile entries
// as the DPP row-broadcast operand of the FMA itself (estep_common.h row_bcast_fmac).
#pragma once
#include "estep_common.h"
#include "special_device.h"

namespace pylda {

constexpr int kQuadWaves = 4;

template <int KRL, int RWL, int TWL>
struct QuadLds {
    static constexpr int W = kQuadWaves;
    static constexpr int kTopics = 16 * KRL;
    static constexpr int kRedStride = 20;                                          // see QuiltLds (8 rows per group)
    static constexpr size_t red = 0;                                               // [W][4][8][kRedStride]
    static constexpr size_t sp = red + (size_t)W * 4 * 8 * kRedStride * 8;         // [W][kTopics]
    static constexpr size_t tt = sp + (size_t)W * kTopics * 8;                     // [2][kTopics]
    static constexpr size_t chg = tt + (size_t)2 * kTopics * 8;                    // u64[2]
    static constexpr size_t misc = chg + 16;                                       // [8][W]
    static constexpr size_t alf = misc + (size_t)8 * W * 8;                        // [kTopics] alpha (0 beyond K)
    static constexpr size_t gpv = alf + (size_t)kTopics * 8;                       // [kTopics] gamma before the last update
    static constexpr size_t cnt = gpv + (size_t)kTopics * 8;                       // int32 [2][W * 64] counts of the words a lane finishes
    static constexpr size_t rows = (cnt + (size_t)2 * W * 64 * 4 + 255) & ~(size_t)255;   // [16][TWL][kTopics]
    static_assert(TWL < 3 || 2 * ((cnt + (size_t)2 * W * 64 * 4 + 255) / 256 * 256 + (size_t)16 * TWL * kTopics * 8) <= 160 * 1024, "two workgroups per CU");
    static constexpr size_t total = rows + (size_t)16 * TWL * kTopics * 8;
};

template <int KRL, int RWL, int TWL>
__global__ __launch_bounds__(kWave* kQuadWaves, 2) void estep_quad_kernel(EstepParams p)
{
    using L = QuadLds<KRL, RWL, TWL>;
    constexpr int W = kQuadWaves;
    constexpr int NT = kWave * W;
    constexpr int KT = 16 * KRL;            // padded topic count (== ldk)
    constexpr int WPG = RWL + TWL;          // word slots per group
    constexpr int C0 = WPG < 8 ? WPG : 8;   // slots finished in the first transpose
    constexpr int C1 = WPG - C0;            // ... in the second
    constexpr int R1 = RWL > 8 ? RWL - 8 : 0;   // register slots of the second chunk
    constexpr int RS = L::kRedStride;
    constexpr int QV = KRL / 4;
    static_assert(KRL == 8, "ldk 128");
    static_assert(RWL >= 2 && RWL <= 10 && TWL >= 0 && WPG <= 16, "word slots per group");
    static_assert(TWL == 0 || RWL >= 8, "LDS slots belong to the second chunk");
    static_assert(KT <= NT, "one thread per topic in the gamma phase");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem + L::red);
    double* sp = reinterpret_cast<double*>(smem + L::sp);
    double* tt = reinterpret_cast<double*>(smem + L::tt);
    unsigned long long* chg = reinterpret_cast<unsigned long long*>(smem + L::chg);
    double* misc = reinterpret_cast<double*>(smem + L::misc);
    // per-thread constants of the inner loop live in LDS, not in VGPRs (the tile takes 160 of 256):
    double* alf = reinterpret_cast<double*>(smem + L::alf);
    double* gpv = reinterpret_cast<double*>(smem + L::gpv);
    int* cntv = reinterpret_cast<int*>(smem + L::cnt);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int g = lane >> 4, c = lane & 15;
    const int gg = wave * 4 + g;            // word group of this lane: words gg, gg + 16, gg + 32, ...
    const int K = p.K, ldk = p.ldk;
    const int doc = p.order[blockIdx.x];
    const int64_t lo = p.doc_ptr[doc];
    const int N = (int)(p.doc_ptr[doc + 1] - lo);
    const double2* table = reinterpret_cast<const double2*>(p.expElog);
    const int ldk2 = ldk / 2;
    // this lane group's tail rows in LDS: [TWL][KT] doubles, the lane reads 16-byte pieces c + 16*jj
    double2* myrows = reinterpret_cast<double2*>(smem + L::rows) + (size_t)gg * TWL * (KT / 2) + c;

    // ---- small loads first: they must not queue behind the tile gather (vmcnt retires in order) ----
    int wid[WPG];
#pragma unroll
    for (int s = 0; s < WPG; ++s) wid[s] = s * 16 + gg < N ? p.term_id[lo + s * 16 + gg] : -1;
    // the words whose normalisers this lane finishes (with its pair lane c ^ 1): slots c/2 and 8 + c/2
    const int slot0 = c >> 1, slot1 = 8 + (c >> 1);
    const int word0 = slot0 * 16 + gg, word1 = slot1 * 16 + gg;
    const bool live0 = slot0 < C0 && word0 < N;
    const bool live1 = C1 > 0 && slot1 < WPG && word1 < N;
    cntv[tid] = live0 ? p.term_ct[lo + word0] : 0;
    cntv[NT + tid] = live1 ? p.term_ct[lo + word1] : 0;
    double local = 0.0;
    for (int n = tid; n < N; n += NT) local += (double)p.term_ct[lo + n];
    double asum = 0.0;
    for (int k = lane; k < K; k += kWave) asum += p.alpha[k];
    const bool topic_thread = tid < KT;
    const bool topic_live = tid < K;
    if (topic_thread) alf[tid] = topic_live ? p.alpha[tid] : 1.0;

    // ---- the tile gather: register slots, then the LDS slots (through registers) ----
    double B[RWL][KRL];
#pragma unroll
    for (int i = 0; i < RWL; ++i) {
        if (wid[i] >= 0) {
            const double2* row = table + (size_t)wid[i] * ldk2 + c;
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
#pragma unroll
    for (int t = 0; t < TWL; ++t) {
        double2 v2[KRL / 2];
        if (wid[RWL + t] >= 0) {
            const double2* row = table + (size_t)wid[RWL + t] * ldk2 + c;
#pragma unroll
            for (int jj = 0; jj < KRL / 2; ++jj) v2[jj] = row[16 * jj];
        } else {
#pragma unroll
            for (int jj = 0; jj < KRL / 2; ++jj) v2[jj] = double2{0.0, 0.0};
        }
#pragma unroll
        for (int jj = 0; jj < KRL / 2; ++jj) myrows[t * (KT / 2) + 16 * jj] = v2[jj];
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
    const double psi_total = uniform_f64(digamma(asum + total));

    // ---- gamma phase state: thread k < KT owns topic k ----
    double gam = 1.0;
    if (topic_thread) {
        gam = topic_live ? alf[tid] + total / K : 1.0;                    // :165 (padding topics never move)
        tt[tid] = topic_live ? exp_digamma_minus(gam, psi_total) : 0.0;
    }
    lds_only_barrier();

    double r0 = 0.0, r1 = 0.0;
    int it = 0;
    int bad = 0;
    double* myred = red + (size_t)wave * 4 * 8 * RS + (size_t)g * 8 * RS;         // this lane group's 8 rows
    const double2* mysrc = reinterpret_cast<const double2*>(myred + (c >> 1) * RS) + (c & 1);
    // stop test: integer compare on the fixed-point sum, evaluated behind the first half of the next
    // iteration (see estep_quilt.h)
    const double thresh_f = p.tol * K * kChangeScale;
    const long long thresh = __double_as_longlong(uniform_f64(__longlong_as_double(
        !(thresh_f >= 0.0) ? -1ll : thresh_f >= 9.2e18 ? 0x7fffffffffffffffll : (long long)thresh_f)));
    long long moved = 0x7fffffffffffffffll;
    int left = p.max_iter;
    double tq[KRL];
#pragma unroll
    for (int jj = 0; jj < KRL / 2; ++jj) {
        const double2 t2 = reinterpret_cast<const double2*>(tt)[c + 16 * jj];
        tq[2 * jj] = t2.x;
        tq[2 * jj + 1] = t2.y;
    }
#pragma unroll
    for (int j = 0; j < KRL; ++j) asm volatile("" : "+v"(tq[j]));
    ExpDigammaScalarCoef coef;
    // The loop body is ordered by hand (estep_common.h, "hand-ordered FMA blocks"): FMA blocks of eight
    // independent chains, and every LDS row requested one block of >= 16 FMAs before it is used, through
    // ONE 16-VGPR row buffer:   [32 FMAs] row 0 [32 FMAs] row 1 [16 FMAs] row 2   in both passes.
    LdsRow rowbuf;
    auto request_row = [&](int t) { lds_row_request(rowbuf, myrows + t * (KT / 2)); };
    if constexpr (TWL > 0) request_row(0);
    for (;;) {                                                            // :174
        const int buf = it & 1;

        // A. partial normalisers over this lane's topics -> LDS transpose -> sum over the 16 topic lanes
        double a[8];
        double pr[TWL > 0 ? TWL : 1];
        auto row_partial = [&](auto idx) {                   // LDS slot t: partial normaliser, next row requested
            constexpr int t = decltype(idx)::value;
            if constexpr (t < TWL) {
                lds_row_wait(rowbuf);
                double row[8];
                rowbuf.unpack(row);
                pr[t] = dot8(row, tq);
                if constexpr (t + 1 < TWL) request_row(t + 1);
            }
        };
        static_assert(C0 == 8 || TWL == 0, "LDS slots need the eight-slot first chunk");
        if constexpr (C0 == 8) {
            col_mul8(a, B[0][0], B[1][0], B[2][0], B[3][0], B[4][0], B[5][0], B[6][0], B[7][0], tq[0]);
            col_fmac8(a, B[0][1], B[1][1], B[2][1], B[3][1], B[4][1], B[5][1], B[6][1], B[7][1], tq[1]);
            col_fmac8(a, B[0][2], B[1][2], B[2][2], B[3][2], B[4][2], B[5][2], B[6][2], B[7][2], tq[2]);
            col_fmac8(a, B[0][3], B[1][3], B[2][3], B[3][3], B[4][3], B[5][3], B[6][3], B[7][3], tq[3]);
            row_partial(StaticIndex<0>());
            col_fmac8(a, B[0][4], B[1][4], B[2][4], B[3][4], B[4][4], B[5][4], B[6][4], B[7][4], tq[4]);
            col_fmac8(a, B[0][5], B[1][5], B[2][5], B[3][5], B[4][5], B[5][5], B[6][5], B[7][5], tq[5]);
            col_fmac8(a, B[0][6], B[1][6], B[2][6], B[3][6], B[4][6], B[5][6], B[6][6], B[7][6], tq[6]);
            col_fmac8(a, B[0][7], B[1][7], B[2][7], B[3][7], B[4][7], B[5][7], B[6][7], B[7][7], tq[7]);
            row_partial(StaticIndex<1>());
        } else {
#pragma unroll
            for (int i = 0; i < C0; ++i) a[i] = dot8(B[i], tq);
        }
#pragma unroll
        for (int i = 0; i < C0; ++i) myred[i * RS + c] = a[i];
        if (moved <= thresh || left <= 0) {                               // :189 (mean <= tol), :174
            if constexpr (TWL > 2) lds_row_wait(rowbuf);                  // no read may land after the loop
            break;
        }
        wave_lds_exchange();
        double2 h0[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) h0[x] = mysrc[2 * x];                 // this lane's half of its word's 16 partials
        const double cnt0 = (double)cntv[tid];
        double s0, s1 = 1.0, cnt1 = 0.0;
        if constexpr (C1 > 0) {
            double a1[R1 > 0 ? R1 : 1];
#pragma unroll
            for (int i = 0; i < R1; ++i) a1[i] = dot8(B[8 + i], tq);
            row_partial(StaticIndex<2>());
            if constexpr (TWL > 0) request_row(0);                        // for pass B
            s0 = ((h0[0].x + h0[1].x) + (h0[2].x + h0[3].x)) + ((h0[0].y + h0[1].y) + (h0[2].y + h0[3].y));
            asm volatile("" : "+v"(s0));                                  // h0 is dead from here on
            wave_lds_exchange();                                          // the writes below stay behind the reads above
#pragma unroll
            for (int i = 0; i < R1; ++i) myred[i * RS + c] = a1[i];
#pragma unroll
            for (int t = 0; t < TWL; ++t) myred[(R1 + t) * RS + c] = pr[t];
            wave_lds_exchange();
            double2 h1[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) h1[x] = mysrc[2 * x];
            cnt1 = (double)cntv[NT + tid];
            {   // the reciprocal chain of the first chunk runs while the second transpose is in flight
                const double s = lane_group_sum<2>(s0);
                if (live0 && !(s > 1e-280 && s < 1e300)) bad = 1;
                r0 = live0 ? cnt0 * rcp_newton(s) : 0.0;
            }
            s1 = ((h1[0].x + h1[1].x) + (h1[2].x + h1[3].x)) + ((h1[0].y + h1[1].y) + (h1[2].y + h1[3].y));
            const double s = lane_group_sum<2>(s1);
            if (live1 && !(s > 1e-280 && s < 1e300)) bad = 1;
            r1 = live1 ? cnt1 * rcp_newton(s) : 0.0;
        } else {
            s0 = ((h0[0].x + h0[1].x) + (h0[2].x + h0[3].x)) + ((h0[0].y + h0[1].y) + (h0[2].y + h0[3].y));
            const double s = lane_group_sum<2>(s0);
            if (live0 && !(s > 1e-280 && s < 1e300)) bad = 1;
            r0 = live0 ? cnt0 * rcp_newton(s) : 0.0;
        }

        // B. q[k] over this lane's words (registers and LDS rows interleaved), then over the 4 word groups
        double q[KRL];
        auto row_topic_sums = [&](auto idx) {                // LDS slot t: q += r * row, next row requested
            constexpr int t = decltype(idx)::value;
            if constexpr (t < TWL) {
                lds_row_wait(rowbuf);
                double row[8];
                rowbuf.unpack(row);
                row_bcast_fmac<2 * (R1 + t)>(q, r1, row);
                if constexpr (t + 1 < TWL) request_row(t + 1);
                else request_row(0);                                      // for pass A of the next iteration
            }
        };
        {
            const double rb = row_bcast<0>(r0);            // r of slot i sits in lane 2*i of this lane's row
#pragma unroll
            for (int j = 0; j < KRL; ++j) q[j] = rb * B[0][j];
        }
        static_for<(C0 < 4 ? C0 : 4) - 1>([&](auto idx) {
            constexpr int i = decltype(idx)::value + 1;
            row_bcast_fmac<2 * i>(q, r0, B[i]);
        });
        row_topic_sums(StaticIndex<0>());
        static_for<(C0 > 4 ? C0 - 4 : 0)>([&](auto idx) {
            constexpr int i = decltype(idx)::value + 4;
            row_bcast_fmac<2 * i>(q, r0, B[i]);
        });
        row_topic_sums(StaticIndex<1>());
        static_for<R1>([&](auto idx) {
            constexpr int i = decltype(idx)::value;
            row_bcast_fmac<2 * i>(q, r1, B[8 + i]);
        });
        row_topic_sums(StaticIndex<2>());
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
            double part[W];
#pragma unroll
            for (int w = 0; w < W; ++w) part[w] = sp[w * KT + tid];
            const double t_mine = tt[buf * KT + tid], alpha_k = alf[tid];
            keep_together(part);
            const double gnew = fma(t_mine, (part[0] + part[1]) + (part[2] + part[3]), alpha_k);   // :185
            const double diff = fabs(gnew - gam);                         // :187
            gpv[tid] = gam;
            gam = gnew;                                                   // :188
            atomicAdd(&chg[buf], change_fixed(diff));
            coef.load();
            tt[(buf ^ 1) * KT + tid] = topic_live ? exp_digamma_minus_with(gam, psi_total, coef) : 0.0;
            if (tid == 0) chg[buf ^ 1] = 0ull;
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

    // ---- document terms (:195-204) with the last phi = B t r (see estep_slab.h) ----
#pragma unroll
    for (int jj = 0; jj < KRL / 2; ++jj) {
        const double2 t2 = reinterpret_cast<const double2*>(tt + last * KT)[c + 16 * jj];
        tq[2 * jj] = t2.x;
        tq[2 * jj + 1] = t2.y;
    }
    double term1 = 0.0;
    if (p.heldout || p.want_doc_ll) {         // else: taken per corpus from the statistics
        double rl[16];
        row_bcast_all<C0, 2>(r0, rl);
        if constexpr (C1 > 0) row_bcast_all<C1, 2>(r1, rl + 8);
        const double2* gtable = reinterpret_cast<const double2*>(p.expElog_elog);
#pragma unroll
        for (int s = 0; s < WPG; ++s) {
            const int n = s * 16 + gg;
            if (n < N) {
                const double2* row = gtable + (size_t)p.term_id[lo + n] * ldk2 + c;
                double gsum2 = 0.0;
#pragma unroll
                for (int jj = 0; jj < KRL / 2; ++jj) {
                    const double2 g2 = row[16 * jj];
                    gsum2 = fma(g2.y, tq[2 * jj + 1], fma(g2.x, tq[2 * jj], gsum2));
                }
                term1 = fma(rl[s], gsum2, term1);
            }
        }
    }
    // c_n log(normaliser_n) from r_n = c_n / normaliser_n (the normalisers themselves were not kept)
    const bool owner0 = live0 && (c & 1) == 0, owner1 = live1 && (c & 1) == 0;
    const double cnt0 = (double)cntv[tid], cnt1 = (double)cntv[NT + tid];
    double term3 = (owner0 ? cnt0 * (log(cnt0) - log(r0)) : 0.0) + (owner1 ? cnt1 * (log(cnt1) - log(r1)) : 0.0);
    double shift_term = 0.0;
    if (p.heldout) {
        if (owner0) shift_term = cnt0 * p.shift[p.term_id[lo + word0]];
        if (owner1) shift_term = fma(cnt1, p.shift[p.term_id[lo + word1]], shift_term);
    } else {
        if (owner0) p.rfinal[lo + word0] = r0;
        if (owner1) p.rfinal[lo + word1] = r1;
    }
    double term2 = 0.0, lse_term = 0.0, lgam = 0.0, gsum = 0.0;
    if (topic_live) {
        const double t_last = tt[last * KT + tid], alpha_k = alf[tid];
        const double mass = gam - alpha_k;                                // = t_last * s
        const double ltv = digamma(gpv[tid]) - psi_total;                 // log t of the last iteration
        term2 = ltv * mass;
        if (p.heldout) lse_term = p.topic_lse[tid] * mass;
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
