// Tiered E-step kernel on a WIDE quilt: 2 word groups x 32 topic lanes per wavefront, for
// 128 < K <= 256 (cfg 4: K = 256, N_d ~ 200, tile 400 KB > a CU's registers) and for long
// documents at K <= 128.
//
//   lane = 32*g + c :  word group g (0..1)  x  topic lane c (0..31)
//   lane owns topics  2c + 64*jj + {0,1}   (jj < JJ = ldk/64; KRL = 2*JJ <= 8 values per row)
//   => a table row is read as 32-lane x 512-byte contiguous pieces (16-byte loads)
//
// With 32 topic lanes a K = 256 row costs 8 registers per lane (the 16-lane quilt of
// estep_qhybrid.h needs 16): t, the topic sums and a tail row each shrink by half, which is
// what leaves room for two row buffers that are re-filled two steps ahead.
//
// The tile is split in three tiers (as in estep_qhybrid.h):
//   tier R   the first 16*8 = 128 words        in VGPRs   B[8][KRL]
//   tier L   the next  NL words                in LDS     whole rows, loaded once per document
//   tier S   whatever is left                  streamed from L2 / Infinity Cache twice per iteration
//
// Per inner iteration:
//   A(R)  in-lane sum_k B t per word -> LDS transpose (32 partials per word, 16-byte aligned rows,
//         conflict-free ds_read_b128) -> four lanes per word finish the normaliser (DPP inside the
//         row, one permlane16 swap across the group's two rows) -> r
//   tail  (two words per step, one per group; a round is up to 8 steps = 16 words per wavefront,
//         handled like a second tier R whose rows come from memory)
//         pass 1: row -> in-lane partial -> the same transpose area; finish: normalisers and r of
//         the round's words side by side, r kept in LDS; B(R): q[k] = sum r B with r broadcast
//         inside the row by the FMA's DPP operand; pass 2: row again -> q[k] += r row
//   then one permlane32 swap level over the two groups, per-wavefront partials to LDS (in the
//   transpose area), barrier, gamma phase on ldk topic threads (estep_quilt.h; its state lives in
//   LDS and its coefficients in scalar registers: the K = 256 instance is at the 256-VGPR limit),
//   barrier.
#pragma once
#include "estep_common.h"
#include "special_device.h"
#include "estep_limits.h"

namespace pylda {


template <int W, int JJ>
struct QwideLds {
    static constexpr int kTopics = 64 * JJ;
    static constexpr int kRowDoubles = kTopics + 2;                                    // +16 B: the two groups' rows differ in bank
    static constexpr int kRedStride = 40;                                              // 32 partials + pad: conflict-free b128 reads
    static constexpr size_t red = 0;                                                   // [W][16][kRedStride]; reused for the W x kTopics partial sums
    static constexpr size_t red_wave = (size_t)16 * kRedStride * 8;
    static_assert(red_wave >= (size_t)kTopics * 8, "a wavefront's topic partials fit in its transpose area");
    static constexpr size_t rr = red + (size_t)W * red_wave;                           // [W][kQwMaxTail]  r of the tail words
    static constexpr size_t nrm = rr + (size_t)W * kQwMaxTail * 8;                     // [W][kQwMaxTail]
    static constexpr size_t cnt = nrm + (size_t)W * kQwMaxTail * 8;                    // [W][kQwMaxTail]
    static constexpr size_t tt = cnt + (size_t)W * kQwMaxTail * 8;                     // [2][kTopics]
    static constexpr size_t gst = tt + (size_t)2 * kTopics * 8;                        // [2][kTopics]: alpha, previous gamma
    static constexpr size_t ids = gst + (size_t)2 * kTopics * 8;                       // int [W][kQwMaxTail]
    static constexpr size_t chg = ids + (size_t)W * kQwMaxTail * 4;                    // u64[2]
    static constexpr size_t misc = chg + 16;                                           // [8][W]
    static constexpr size_t rows = (misc + (size_t)8 * W * 8 + 15) & ~(size_t)15;      // tier L rows start here
    static constexpr size_t fixed_total = rows;
    static constexpr int rows_that_fit(size_t lds_limit)
    {
        return lds_limit > fixed_total ? (int)((lds_limit - fixed_total) / ((size_t)kRowDoubles * 8)) : 0;
    }
};

// sum over the 32 lanes of a word group (two 16-lane rows); every lane gets it
__device__ __forceinline__ double group32_sum(double s)
{
    s += dpp_f64<0xB1>(s);      // quad_perm [1,0,3,2]
    s += dpp_f64<0x4E>(s);      // quad_perm [2,3,0,1]
    s += dpp_f64<0x141>(s);     // row_half_mirror
    s += dpp_f64<0x140>(s);     // row_mirror
    return swap16_add(s, s);    // rows 2g and 2g+1
}

// MULTI: documents with more than 8 tail steps per wavefront (N > 256) loop over rounds; the
// single-round instance lets t die after pass 1 (the kernel is at the register limit at K = 256).
template <int W, int JJ, bool MULTI>
__global__ __launch_bounds__(kWave* W) void estep_qwide_kernel(EstepParams p, int lds_rows_per_wave)
{
    using L = QwideLds<W, JJ>;
    constexpr int NT = kWave * W;
    constexpr int KT = 64 * JJ;             // padded topic count (== ldk)
    constexpr int KRL = 2 * JJ;             // values per lane and row
    constexpr int RWL = 8;                  // tier R words per lane
    constexpr int RNW = 2 * RWL;            // tier R words per wavefront
    constexpr int RS = L::kRedStride;
    constexpr int ROW = L::kRowDoubles;
    static_assert(JJ >= 2 && JJ <= 4, "ldk 128, 192 or 256");
    static_assert(KT <= NT, "one thread per topic in the gamma phase");
    static_assert(W * RNW == kQwRegWords, "tier R size");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem + L::red);
    double* rr = reinterpret_cast<double*>(smem + L::rr);
    double* nrmv = reinterpret_cast<double*>(smem + L::nrm);
    double* cntv = reinterpret_cast<double*>(smem + L::cnt);
    double* tt = reinterpret_cast<double*>(smem + L::tt);
    double* alphaS = reinterpret_cast<double*>(smem + L::gst);            // gamma-phase state kept in LDS, not in
    double* gprevS = alphaS + KT;                                         // registers (read where the partial sums are)
    int* ids = reinterpret_cast<int*>(smem + L::ids);
    unsigned long long* chg = reinterpret_cast<unsigned long long*>(smem + L::chg);
    double* misc = reinterpret_cast<double*>(smem + L::misc);
    double* rows = reinterpret_cast<double*>(smem + L::rows);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int g = lane >> 5, c = lane & 31;
    const int cl = lane & 15, half = (lane >> 4) & 1;       // position inside the 16-lane row, row of the group
    const int K = p.K, ldk = p.ldk;
    const int doc = p.order[blockIdx.x];
    const int64_t lo = p.doc_ptr[doc];
    const int N = (int)(p.doc_ptr[doc + 1] - lo);

    // ---- word assignment ----
    // tier R: wave w, group g owns words [w*16 + g*8, +8); tail (tiers L, S): words >= 128 are dealt
    // to the wavefronts in contiguous blocks of NTW (even); the first NLW (even) of a wavefront's
    // tail words live in LDS, the rest are streamed.
    const int wbR = wave * RNW + g * RWL;
    const int tail = max(0, N - W * RNW);
    const int NTW = ((tail + W - 1) / W + 1) & ~1;
    const int nbT = W * RNW + wave * NTW;
    const int nmineT = max(0, min(NTW, N - nbT));
    const int NLW = min(NTW, lds_rows_per_wave & ~1);
    double* myred = red + (size_t)wave * (L::red_wave / 8);
    double* myrrT = rr + wave * kQwMaxTail;
    double* mynrmT = nrmv + wave * kQwMaxTail;
    double* mycntT = cntv + wave * kQwMaxTail;
    int* myidsT = ids + wave * kQwMaxTail;
    double* myrows = rows + (size_t)wave * lds_rows_per_wave * ROW;
    const double2* table = reinterpret_cast<const double2*>(p.expElog);
    const int ldk2 = ldk / 2;

    // ---- small loads first (they must not queue behind the tile gather) ----
    int wid[RWL];
#pragma unroll
    for (int i = 0; i < RWL; ++i) wid[i] = wbR + i < N ? p.term_id[lo + wbR + i] : -1;
    // the tier R word whose normaliser this lane finishes: slot cl/2 of its group, one of four
    // lanes (two per row); r reaches the word's 32 tile lanes by a DPP row broadcast in each row
    const int my_slot = cl >> 1;
    const int my_part = (cl & 1) + 2 * half;
    const int my_word = wbR + my_slot;
    const bool word_live = my_word < N;
    const double my_cnt = word_live ? (double)p.term_ct[lo + my_word] : 0.0;
    double local = 0.0;
    for (int n = tid; n < N; n += NT) local += (double)p.term_ct[lo + n];
    double asum = 0.0;
    for (int k = lane; k < K; k += kWave) asum += p.alpha[k];
    const bool topic_thread = tid < KT;
    const bool topic_live = tid < K;
    const double alpha_k = topic_live ? p.alpha[tid] : 1.0;
    for (int i = lane; i < NTW; i += kWave) {
        myidsT[i] = i < nmineT ? p.term_id[lo + nbT + i] : 0;
        mycntT[i] = i < nmineT ? (double)p.term_ct[lo + nbT + i] : 0.0;
    }

    // ---- tier R gather (in flight during the set-up) ----
    double B[RWL][KRL];
#pragma unroll
    for (int i = 0; i < RWL; ++i) {
        if (wid[i] >= 0) {
            const double2* row = table + (size_t)wid[i] * ldk2 + c;
#pragma unroll
            for (int jj = 0; jj < JJ; ++jj) {
                const double2 v2 = row[32 * jj];
                B[i][2 * jj] = v2.x;
                B[i][2 * jj + 1] = v2.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < KRL; ++j) B[i][j] = 0.0;
        }
    }

    // ---- tier L rows: table -> LDS, once per document ----
    wave_lds_exchange();                    // myidsT written above, read below by other lanes of this wavefront
    for (int off = 0; off < NLW; off += 2) {
        const int i = off + g;
        const double2* src = table + (size_t)myidsT[i] * ldk2 + c;
        double2* dst = reinterpret_cast<double2*>(myrows + (size_t)i * ROW) + c;
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) dst[32 * jj] = src[32 * jj];
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
    const double psi_total = uniform_f64(digamma(asum + total));          // the same in every lane: scalar registers

    double gam = topic_live ? alpha_k + total / K : alpha_k;              // :165 (padding topics never move)
    if (topic_thread) {
        tt[tid] = topic_live ? exp_digamma_minus(gam, psi_total) : 0.0;
        alphaS[tid] = alpha_k;
        gprevS[tid] = gam;
    }
    __syncthreads();                        // also: the tier L rows are in place (their global loads drained)

    double r_mine = 0.0, nrm_mine = 1.0;
    int it = 0;
    int bad = 0;
    const double2* mysrc = reinterpret_cast<const double2*>(myred + (g * RWL + my_slot) * RS) + my_part;
    // the stop test of iteration i rides behind the first half of iteration i+1 (estep_quilt.h)
    const double thresh_f = p.tol * K * kChangeScale;
    const long long thresh = __double_as_longlong(uniform_f64(__longlong_as_double(
        !(thresh_f >= 0.0) ? -1ll : thresh_f >= 9.2e18 ? 0x7fffffffffffffffll : (long long)thresh_f)));
    long long moved = 0x7fffffffffffffffll;
    int left = p.max_iter;
    double tq[KRL];
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj) {
        const double2 t2 = reinterpret_cast<const double2*>(tt)[c + 32 * jj];
        tq[2 * jj] = t2.x;
        tq[2 * jj + 1] = t2.y;
    }
#pragma unroll
    for (int j = 0; j < KRL; ++j) asm volatile("" : "+v"(tq[j]));

    // Tail steps (two words each: one per group), streamed steps first.  A ROUND is up to eight
    // steps = 16 words per wavefront, handled like a second tier R whose rows come from memory:
    //   pass 1  row -> in-lane partial -> transpose area
    //   finish  four lanes per word: normaliser, r (kept in LDS for pass 2 and the epilogue)
    //   pass 2  row again -> q += r row
    // so the reduction and reciprocal chains of a round's words run side by side, and a step is
    // 8 + 8 FMAs and a handful of address instructions.  Rows go through two alternating register
    // buffers, each re-filled two steps ahead; the first two rows of an iteration (streamed ones,
    // when the document has any) are requested before the tier R work.  cfg 4 (~70 tail words) is
    // one round.
    typedef const f64x2 __attribute__((address_space(3)))* lds_row_ptr;
    const int nS = (NTW - NLW) / 2, nsteps = NTW / 2;
    // (a generic pointer into LDS carries the LDS byte offset in its low 32 bits)
    const unsigned myrows_lds = (unsigned)(uintptr_t)(myrows + 2 * c);
    auto tail_index = [&](int st) { return (st < nS ? NLW + 2 * st : 2 * (st - nS)) + g; };
    auto fetch_streamed = [&](int st, f64x2 (&rowv)[JJ]) {     // tier S: from the table (L2 / Infinity Cache)
        const f64x2* row = reinterpret_cast<const f64x2*>(table + (size_t)myidsT[NLW + 2 * st + g] * ldk2 + c);
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) rowv[jj] = row[32 * jj];
    };
    auto fetch_resident = [&](int st, f64x2 (&rowv)[JJ]) {     // tier L: explicit LDS address space (ds_read_b128)
        lds_row_ptr row = (lds_row_ptr)(uintptr_t)(myrows_lds + (unsigned)(2 * (st - nS) + g) * (ROW * 8));
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) rowv[jj] = row[32 * jj];
    };
    // steps [lo, hi) of ONE tier through the two alternating buffers; each buffer is re-filled two
    // steps ahead and the loop body issues a fixed number of loads (so the compiler's wait counts
    // are exact: a step waits for its own row only).  `primed`: r0 / r1 already hold steps lo, lo+1.
    auto run_steps = [&](int lo, int hi, bool streamed, bool primed, f64x2 (&r0)[JJ], f64x2 (&r1)[JJ], auto&& body) {
        if (lo >= hi) return;
        auto fetch = [&](int st, f64x2 (&rowv)[JJ]) {
            if (streamed) fetch_streamed(st, rowv);
            else fetch_resident(st, rowv);
        };
        if (!primed) {
            fetch(lo, r0);
            fetch(min(lo + 1, hi - 1), r1);
        }
        for (int st = lo;; st += 2) {
            body(st, r0);
            const bool more = st + 2 < hi;
            if (more) fetch(st + 2, r0);
            if (st + 1 < hi) body(st + 1, r1);
            if (!more) break;
            fetch(min(st + 3, hi - 1), r1);
        }
    };
    // four lanes per word finish a normaliser from the transpose area
    auto finish_sum = [&]() {
        double2 s2 = mysrc[0];
#pragma unroll
        for (int x = 1; x < 4; ++x) {
            const double2 v2 = mysrc[4 * x];
            s2.x += v2.x;
            s2.y += v2.y;
        }
        double s = s2.x + s2.y;
        s += dpp_f64<0xB1>(s);              // the word's other lane of this row
        return swap16_add(s, s);            // ... and of the group's other row
    };

    for (;;) {                                                            // :174
        const int buf = it & 1;
        f64x2 r0[JJ], r1[JJ];
        const int nS0 = min(nS, 8);                                       // streamed steps of the first round
        if (nS0 > 0) {                                                    // requested before the tier R work
            fetch_streamed(0, r0);
            fetch_streamed(min(1, nS0 - 1), r1);
        }

        // A(R). tier R partial normalisers -> LDS transpose
#pragma unroll
        for (int i = 0; i < RWL; ++i) {
            double a0 = B[i][0] * tq[0];
#pragma unroll
            for (int j = 1; j < KRL; ++j) a0 = fma(B[i][j], tq[j], a0);
            myred[(g * RWL + i) * RS + c] = a0;
        }
        if (moved <= thresh || left <= 0) break;                          // :189 (mean <= tol), :174
        wave_lds_exchange();
        {
            const double s = finish_sum();
            nrm_mine = s;
            if (word_live && !(s > 1e-280 && s < 1e300)) bad = 1;
            r_mine = word_live ? my_cnt * rcp_newton(s) : 0.0;
        }

        double q[KRL];
        for (int st0 = 0; st0 == 0 || (MULTI && st0 < nsteps); st0 += 8) {
            const int stn = min(st0 + 8, nsteps);
            const int sS = min(st0, nS), eS = min(stn, nS);               // the round's streamed steps [sS, eS)
            const int sL = max(st0, nS), eL = stn;                        // ... and LDS steps [sL, eL)
            // ---- pass 1 (the transposes of the previous finish have been read by this wavefront) ----
            wave_lds_exchange();
            auto dot_to_transpose = [&](int st, const f64x2 (&rowv)[JJ]) {
                double a0 = rowv[0].x * tq[0], a1 = rowv[0].y * tq[1];
#pragma unroll
                for (int jj = 1; jj < JJ; ++jj) {
                    a0 = fma(rowv[jj].x, tq[2 * jj], a0);
                    a1 = fma(rowv[jj].y, tq[2 * jj + 1], a1);
                }
                myred[(g * RWL + (st - st0)) * RS + c] = a0 + a1;
            };
            run_steps(sS, eS, true, st0 == 0, r0, r1, dot_to_transpose);
            run_steps(sL, eL, false, false, r0, r1, dot_to_transpose);
            // the first rows of pass 2 are requested before the normalisers are finished
            const bool s_first = sS < eS;
            if (s_first) {
                fetch_streamed(sS, r0);
                fetch_streamed(min(sS + 1, eS - 1), r1);
            }
            wave_lds_exchange();
            if (st0 < stn) {
                const int st = st0 + my_slot;                 // the step whose word (of group g) this lane finishes
                const int i = tail_index(min(st, nsteps - 1));
                const bool live = st < stn && i < nmineT;
                const double sn = finish_sum();
                if (live && !(sn > 1e-280 && sn < 1e300)) bad = 1;
                const double rn = live ? mycntT[i] * rcp_newton(sn) : 0.0;
                if (my_part == 0 && st < stn) {
                    mynrmT[i] = sn;
                    myrrT[i] = rn;
                }
            }
            // ---- B(R) once, then pass 2: topic sums of the round's words ----
            if (st0 == 0) row_bcast_matvec<RWL, 2>(q, r_mine, B);
            wave_lds_exchange();                              // r of the round's words is in LDS
            auto accumulate = [&](int st, const f64x2 (&rowv)[JJ]) {
                const double rn = myrrT[tail_index(st)];
#pragma unroll
                for (int jj = 0; jj < JJ; ++jj) {
                    q[2 * jj] = fma(rn, rowv[jj].x, q[2 * jj]);
                    q[2 * jj + 1] = fma(rn, rowv[jj].y, q[2 * jj + 1]);
                }
            };
            run_steps(sS, eS, true, s_first, r0, r1, accumulate);
            run_steps(sL, eL, false, false, r0, r1, accumulate);
        }

        // the two word groups (one permlane32 swap level); lane (g, c) keeps registers j = m + g*JJ
        wave_lds_exchange();                // the transposes of this wavefront have been read: the area is reused
#pragma unroll
        for (int m = 0; m < JJ; ++m) {
            const double v = swap32_add(q[m], q[m + JJ]);
            const int j = m + g * JJ;       // register index: topic 2c + 64*(j>>1) + (j&1)
            myred[2 * c + 64 * (j >> 1) + (j & 1)] = v;
        }
        __syncthreads();

        // C. gamma update by the topic threads
        if (topic_thread) {
            double part[W];
#pragma unroll
            for (int w = 0; w < W; ++w) part[w] = red[(size_t)w * (L::red_wave / 8) + tid];
            keep_together(part);
            const double t_mine = tt[buf * KT + tid], a_k = alphaS[tid];
            double s0 = part[0], s1 = part[1];
#pragma unroll
            for (int w = 2; w < W; w += 2) {
                s0 += part[w];
                s1 += part[w + 1];
            }
            const double gnew = fma(t_mine, s0 + s1, a_k);                // :185
            const double diff = fabs(gnew - gam);                         // :187
            gprevS[tid] = gam;
            gam = gnew;                                                   // :188
            atomicAdd(&chg[buf], change_fixed(diff));
            const double t_next = exp_digamma_minus_levels(gam, psi_total)   /* (tables fetched in place: no scalar registers to spare across pass B) */;
            tt[(buf ^ 1) * KT + tid] = topic_live ? t_next : 0.0;
            if (tid == 0) store_u64_hi(&chg[buf ^ 1], 0u);
        }
        ++it;
        --left;
        __syncthreads();
        moved = (long long)chg[buf];
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
            const double2 t2 = reinterpret_cast<const double2*>(tt + (buf ^ 1) * KT)[c + 32 * jj];
            tq[2 * jj] = t2.x;
            tq[2 * jj + 1] = t2.y;
        }
    }
    const int last = (it - 1) & 1;          // tt[last] holds t of the last executed iteration

    bad = __syncthreads_or(bad);
    if (bad) {
        if (!p.heldout) {
            for (int n = tid; n < N; n += NT) p.rfinal[lo + n] = 0.0;
            for (int k = tid; k < ldk; k += NT) p.tfinal[(size_t)doc * ldk + k] = 0.0;
        }
        if (tid == 0) p.status[doc] = 1;
        return;
    }

    // ---- training fast path: the document terms are left to doc_terms_kernel (doc_terms.h; see estep_quad.h) ----
    if (!p.heldout && !p.want_doc_ll) {
        if (word_live && my_part == 0) p.rfinal[lo + my_word] = r_mine;
        for (int i = lane; i < nmineT; i += kWave) p.rfinal[lo + nbT + i] = myrrT[i];
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

    // ---- document terms (:195-204) with the last phi = B t r (identities: estep_slab.h) ----
    double term1 = 0.0;
    if (p.heldout || p.want_doc_ll) {
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
            const double2 t2 = reinterpret_cast<const double2*>(tt + last * KT)[c + 32 * jj];
            tq[2 * jj] = t2.x;
            tq[2 * jj + 1] = t2.y;
        }
        const double2* gtable = reinterpret_cast<const double2*>(p.expElog_elog);
        auto g_row_dot = [&](int word_id) {
            const double2* row = gtable + (size_t)word_id * ldk2 + c;
            double acc = 0.0;
#pragma unroll
            for (int jj = 0; jj < JJ; ++jj) {
                const double2 g2 = row[32 * jj];
                acc = fma(g2.y, tq[2 * jj + 1], fma(g2.x, tq[2 * jj], acc));
            }
            return acc;
        };
        double rl[RWL];
        row_bcast_all<RWL, 2>(r_mine, rl);
#pragma unroll
        for (int i = 0; i < RWL; ++i)
            if (wbR + i < N) term1 = fma(rl[i], g_row_dot(p.term_id[lo + wbR + i]), term1);
        for (int off = 0; off < NTW; off += 2) {
            const int i = off + g;
            term1 = fma(myrrT[i], g_row_dot(myidsT[i]), term1);      // dead slots: r = 0, id 0
        }
    }
    const bool word_owner = word_live && my_part == 0;
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
        const double moved_k = gam - alphaS[tid];
        const double ltv = digamma(gprevS[tid]) - psi_total;
        term2 = ltv * moved_k;
        if (p.heldout) lse_term = p.topic_lse[tid] * moved_k;
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
