// Fused streaming E-step kernel on the quad kernel's lane grid, for the documents the quad kernel cannot hold:
// more than 256 distinct terms at table stride 64 / 128 / 256 (29 of cfg 4's million documents, 2 of cfg 3's 100 000;
// a third of nips.88-05 at K = 100), up to 1024.
//
//   lane = TL*g + c : word group g (64 / TL per wavefront), topic lane c; a lane holds 8 values of a table row,
//   topics 2c + 2*TL*jj + {0,1}, jj < 4 (estep_quad.h) - ALL topics of a word sit in one TL-lane group, so the scheme
//   of estep_qfuse.h applies per group:   row -> row . t -> sum over the group's lanes (DPP, + one permlane16 swap
//   at TL = 32) -> r = c / normaliser -> topic sums += r row.   Every row is read ONCE per inner iteration and a
//   wavefront instruction serves 64 / TL words (estep_qfuse.h at these strides would give a word a whole wavefront
//   for 2-4 values per lane).  8 wavefronts per document, 8 * 64 / TL groups; word n belongs to group n % NG, slot
//   n / NG; the first RWL slots of a group stay in VGPRs (TL = 16: 256 words, TL = 32: 128), the rest is streamed from
//   the table through two row buffers, each re-requested two slots ahead.  TL = 8 is table stride 64 (documents beyond
//   the quilt and slab kernels' 256 / 384 terms).
//
// It replaces three kernels of rounds 1-2 (the tiered estep_qwide.h, estep_qhybrid.h and the two-pass estep_qstream.h,
// 1200 lines) that re-read their streamed tier twice per iteration.
#pragma once
#include "estep_common.h"
#include "special_device.h"
#include "estep_epilogue.h"
#include "estep_limits.h"

namespace pylda {

template <int TL>
struct QgroupLds {
    static constexpr int W = 8;
    static constexpr int NG = W * (kWave / TL);                                    // word groups per document (32 or 16)
    static constexpr int kSlots = kQgMaxWords / NG;                                // word slots per group
    static constexpr int kTopics = 8 * TL;
    static constexpr size_t sp = 0;                                                // [W][kTopics] topic partials
    static constexpr size_t tt = sp + (size_t)W * kTopics * 8;                     // [2][kTopics]
    static constexpr size_t alf = tt + (size_t)2 * kTopics * 8;                    // [kTopics] alpha
    static constexpr size_t gpv = alf + (size_t)kTopics * 8;                       // [kTopics] gamma before the last update
    static constexpr size_t chg = gpv + (size_t)kTopics * 8;                       // u64[2]
    static constexpr size_t misc = chg + 16;                                       // [8][W]
    static constexpr size_t off = misc + (size_t)8 * W * 8;                        // unsigned [NG][kSlots] byte offset of the row
    static constexpr size_t cnt = off + (size_t)kQgMaxWords * 4;                   // double [NG][kSlots]
    static constexpr size_t rr = cnt + (size_t)kQgMaxWords * 8;                    // double [NG][kSlots]  r of the last iteration
    static constexpr size_t total = rr + (size_t)kQgMaxWords * 8;
    static_assert(total <= 64 * 1024, "no opt-in needed");
};

// sum over the TL lanes of a word group; every lane gets it
template <int TL>
__device__ __forceinline__ double group_sum(double v)
{
    v = lane_group_sum<8>(v);
    if constexpr (TL >= 16) v += dpp_f64<0x140>(v);         // row_mirror: the other half of the 16-lane row
    if constexpr (TL == 32) v = swap16_add(v, v);           // ... and the group's other row
    return v;
}

// the newest row requested may stay in flight (four loads)
__device__ __forceinline__ void table_row_wait_older(LdsRow& r)
{
    asm volatile("s_waitcnt vmcnt(4)" : "+v"(r.p[0]), "+v"(r.p[1]), "+v"(r.p[2]), "+v"(r.p[3]) : : "memory");
}

template <int TL, int RWL>
__global__ __launch_bounds__(512, 2) void estep_qgroup_kernel(EstepParams p)
{
    using L = QgroupLds<TL>;
    constexpr int W = 8, NT = 512, KT = 8 * TL, KRL = 8, G = kWave / TL, NG = L::NG, SL = L::kSlots;
    constexpr int QV = KRL / G;             // topic values per lane after the in-wavefront reduction (2 or 4)
    static_assert(TL == 8 || TL == 16 || TL == 32, "table stride 64, 128 or 256");
    static_assert(RWL % 2 == 0 && RWL <= SL, "register slots");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sp = reinterpret_cast<double*>(smem + L::sp);
    double* tt = reinterpret_cast<double*>(smem + L::tt);
    double* alf = reinterpret_cast<double*>(smem + L::alf);
    double* gpv = reinterpret_cast<double*>(smem + L::gpv);
    unsigned long long* chg = reinterpret_cast<unsigned long long*>(smem + L::chg);
    double* misc = reinterpret_cast<double*>(smem + L::misc);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int g = lane / TL, c = lane % TL;
    const int gg = wave * G + g;            // word group of this lane: words gg, gg + NG, gg + 2 NG, ...
    const int K = p.K, ldk = p.ldk;
    const int doc = p.order[blockIdx.x];
    const int64_t lo = p.doc_ptr[doc];
    const int N = (int)(p.doc_ptr[doc + 1] - lo);
    const int S = (N + NG - 1) / NG;                        // word slots per group
    const int NS = S > RWL ? (S - RWL + 1) & ~1 : 0;        // streamed slots, padded to whole trips of two
    const int Spad = RWL + NS;
    unsigned* myoff = reinterpret_cast<unsigned*>(smem + L::off) + gg * SL;
    double* mycnt = reinterpret_cast<double*>(smem + L::cnt) + gg * SL;
    double* myrr = reinterpret_cast<double*>(smem + L::rr) + gg * SL;
    const int ldk2 = ldk / 2;

    // ---- rows / counts of this group's slots (the group's lanes share the work), token total (:162) ----
    double local = 0.0;
    for (int s = c; s < Spad; s += TL) {
        const int n = s * NG + gg;
        const bool live = n < N;
        myoff[s] = (unsigned)(live ? p.term_id[lo + n] : 0) * (unsigned)(ldk2 * 16);   // dead slots: a valid row, count 0 => r = 0
        const double ct = live ? (double)p.term_ct[lo + n] : 0.0;
        mycnt[s] = ct;
        myrr[s] = 0.0;
        local += ct;
    }
    local = wave_sum(local);
    double asum = 0.0;
    for (int k = lane; k < K; k += kWave) asum += p.alpha[k];
    asum = wave_sum(asum);
    const bool topic_thread = tid < KT;
    const bool topic_live = tid < K;
    if (topic_thread) alf[tid] = topic_live ? p.alpha[tid] : 1.0;
    if (lane == 0) misc[wave] = local;
    if (tid == 0) chg[0] = chg[1] = 0ull;
    __syncthreads();                                        // also: myoff / mycnt are in place

    // ---- register tier ----
    const char* table = reinterpret_cast<const char*>(p.expElog);
    double B[RWL][KRL];
#pragma unroll
    for (int i = 0; i < RWL; ++i) {
        const double2* row = reinterpret_cast<const double2*>(table + myoff[i]) + c;
#pragma unroll
        for (int jj = 0; jj < KRL / 2; ++jj) {
            const double2 v2 = row[TL * jj];
            B[i][2 * jj] = v2.x;
            B[i][2 * jj + 1] = v2.y;
        }
    }
    double total = 0.0;
#pragma unroll
    for (int w = 0; w < W; ++w) total += misc[w];
    const double psi_total = uniform_f64(digamma(asum + total));
    double gam = 1.0;
    if (topic_thread) {
        gam = topic_live ? alf[tid] + total / K : 1.0;                    // :165 (padding topics never move)
        tt[tid] = topic_live ? exp_digamma_minus(gam, psi_total) : 0.0;
    }
    __syncthreads();

    int it = 0;
    int bad = 0;
    const unsigned lane_off = (unsigned)c * 16u;
    while (it < p.max_iter) {                                             // :174
        const int buf = it & 1;
        double tq[KRL];
#pragma unroll
        for (int jj = 0; jj < KRL / 2; ++jj) {
            const double2 t2 = reinterpret_cast<const double2*>(tt + buf * KT)[c + TL * jj];
            tq[2 * jj] = t2.x;
            tq[2 * jj + 1] = t2.y;
        }
        double q[KRL];
#pragma unroll
        for (int j = 0; j < KRL; ++j) q[j] = 0.0;
        // one word per group, fused: normaliser, r, topic sums
        auto word = [&](const double (&row)[KRL], int slot) {
            double nrm = dot8(row, tq);
            asm volatile("s_nop 1" : "+v"(nrm));                          // (a DPP read behind the asm block's last add)
            nrm = group_sum<TL>(nrm);
            const double cnt = mycnt[slot];
            const bool live = cnt > 0.0;
            if (live && !(nrm > 1e-280)) bad = 1;                         // (B, t <= 1: no overflow; NaN fails the compare)
            const double r = live ? cnt * rcp_newton(nrm) : 0.0;
            if (c == 0) myrr[slot] = r;
#pragma unroll
            for (int j = 0; j < KRL; ++j) q[j] = fma(r, row[j], q[j]);
        };
#pragma unroll
        for (int i = 0; i < RWL; ++i) word(B[i], i);
        if (NS > 0) {
            LdsRow g0, g1;
            table_row_request<TL * 16>(g0, table, myoff[RWL] + lane_off);
            table_row_request<TL * 16>(g1, table, myoff[RWL + 1] + lane_off);
            double row[KRL];
            int s = RWL;
            for (; s + 2 < Spad; s += 2) {          // full trips: a buffer is re-requested two slots ahead
                table_row_wait_older(g0); g0.unpack(row); word(row, s);
                table_row_request<TL * 16>(g0, table, myoff[s + 2] + lane_off);
                table_row_wait_older(g1); g1.unpack(row); word(row, s + 1);
                table_row_request<TL * 16>(g1, table, myoff[s + 3] + lane_off);
            }
            table_row_wait_older(g0); g0.unpack(row); word(row, s);       // last trip: the pipeline drains
            table_row_wait(g1); g1.unpack(row); word(row, s + 1);
        }
        // over the word groups of the wavefront (estep_quad.h): lane (g, c), register j <-> topic 2c + (j & 1) + 2 TL (j >> 1)
        double* mysp = sp + (size_t)wave * KT;
        if constexpr (TL == 8) {
            // 8 groups: both wavefront halves, both rows of a half, both halves of a row; lane (g, c) ends with register j = g
            double u[KRL / 2];
#pragma unroll
            for (int m = 0; m < KRL / 2; ++m) u[m] = swap32_add(q[m], q[m + KRL / 2]);
            double v0 = swap16_add(u[0], u[2]), v1 = swap16_add(u[1], u[3]);
            v0 += dpp_f64<0x128>(v0);                        // row_ror:8 - the other half of the 16-lane row
            v1 += dpp_f64<0x128>(v1);
            mysp[2 * c + (g & 1) + 2 * TL * (g >> 1)] = (g & 1) ? v1 : v0;
        } else if constexpr (TL == 16) {
            double u[KRL / 2];
#pragma unroll
            for (int m = 0; m < KRL / 2; ++m) u[m] = swap32_add(q[m], q[m + KRL / 2]);
#pragma unroll
            for (int m = 0; m < QV; ++m) {
                const double v = swap16_add(u[m], u[m + QV]);
                const int j = m + (g & 1) * QV + (g >> 1) * (KRL / 2);
                mysp[2 * c + (j & 1) + 2 * TL * (j >> 1)] = v;
            }
        } else {
#pragma unroll
            for (int m = 0; m < QV; ++m) {
                const double v = swap32_add(q[m], q[m + QV]);
                const int j = m + g * QV;
                mysp[2 * c + (j & 1) + 2 * TL * (j >> 1)] = v;
            }
        }
        __syncthreads();

        // C. gamma update: one thread per topic
        if (topic_thread) {
            double part[W];
#pragma unroll
            for (int w = 0; w < W; ++w) part[w] = sp[w * KT + tid];
            const double t_mine = tt[buf * KT + tid], alpha_k = alf[tid];
            keep_together(part);
            const double s0 = (part[0] + part[1]) + (part[4] + part[5]), s1 = (part[2] + part[3]) + (part[6] + part[7]);
            const double gnew = fma(t_mine, s0 + s1, alpha_k);            // :185
            const double diff = topic_live ? fabs(gnew - gam) : 0.0;      // :187
            gpv[tid] = gam;
            gam = gnew;                                                   // :188
            atomicAdd(&chg[buf], change_fixed(diff));
            const double t_next = exp_digamma_minus_levels(gam, psi_total);
            tt[(buf ^ 1) * KT + tid] = topic_live ? t_next : 0.0;
            if (tid == 0) store_u64_hi(&chg[buf ^ 1], 0u);
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
            if (topic_thread) p.tfinal[(size_t)doc * ldk + tid] = 0.0;
        }
        if (tid == 0) p.status[doc] = 1;
        return;
    }

    // ---- training fast path: the document terms are left to doc_terms_kernel (doc_terms.h; see estep_quad.h) ----
    if (!p.heldout && !p.want_doc_ll) {
        for (int s = c; s < S; s += TL)
            if (s * NG + gg < N) p.rfinal[lo + s * NG + gg] = myrr[s];
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
    {
        double tq[KRL];
#pragma unroll
        for (int jj = 0; jj < KRL / 2; ++jj) {
            const double2 t2 = reinterpret_cast<const double2*>(tt + last * KT)[c + TL * jj];
            tq[2 * jj] = t2.x;
            tq[2 * jj + 1] = t2.y;
        }
        const char* gtable = reinterpret_cast<const char*>(p.expElog_elog);
        for (int s = 0; s < S; ++s) {
            const double2* row = reinterpret_cast<const double2*>(gtable + myoff[s]) + c;
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int jj = 0; jj < KRL / 2; ++jj) {
                const double2 g2 = row[TL * jj];
                a0 = fma(g2.x, tq[2 * jj], a0);
                a1 = fma(g2.y, tq[2 * jj + 1], a1);
            }
            term1 = fma(myrr[s], a0 + a1, term1);          // r = 0 for dead slots; summed over the lanes below
        }
    }
    double term3 = 0.0, shift_term = 0.0;
    for (int s = c; s < S; s += TL) {
        const int n = s * NG + gg;
        if (n < N) {
            const double cnt = mycnt[s], r = myrr[s];
            term3 = fma(cnt, log(cnt) - log(r), term3);    // c_n log(normaliser_n), normaliser = c_n / r_n
            if (p.heldout) shift_term = fma(cnt, p.shift[p.term_id[lo + n]], shift_term);
            else p.rfinal[lo + n] = r;
        }
    }
    TopicShare share;
    if (topic_thread)
        topic_share(p, doc, tid, ldk, topic_live, true, gam, alf[tid], gpv[tid], tt[last * KT + tid], psi_total, share);
    finish_document<W>(p, doc, it, misc, lane, wave, tid, term1, term3, shift_term, share);
}

}  // namespace pylda
