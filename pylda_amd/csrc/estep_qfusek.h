// Fused streaming E-step kernel for 512 < K <= 1024 (table stride 128 * NP, NP = 5 .. 8).
//
// The reference takes any K (variational_bayes.py:132); round 2 sent everything above 512 topics to the generic
// kernels (one thread per word walking a row: a 10x cliff).  This is the single-pass scheme of estep_qfuse.h
// with every row streamed: a word's row (NP x 1 KiB) sits in ONE wavefront, 2 NP values per lane, so
//
//     row -> dot with t -> wavefront sum -> r = c / nrm -> q += r * row
//
// reads each row ONCE per inner iteration from L2 / Infinity Cache and needs no LDS transpose.  No word stays on
// chip (a 230-term document at K = 1000 is 1.8 MB); four row buffers and words in pairs up to stride 768, two buffers
// and single words above (2 x 2 NP VGPR pairs each) per wavefront, t and the topic sums live in registers.  Gamma phase: KT <= 1024 topics on the workgroup's 512
// threads, two topics per thread (tid, tid + 512).
//
// Layout: 8 wavefronts per document, word n belongs to wavefront n % 8, slot n / 8 (documents up to
// 8 * kQfMaxSlots = 1024 distinct terms); lane c holds topics 2c + 128 jj + {0,1}, jj < NP.
#pragma once
#include "estep_common.h"
#include "estep_qfuse.h"
#include "special_device.h"
#include "estep_epilogue.h"
#include "estep_limits.h"

namespace pylda {

template <int NP>
struct QfusekLds {
    static constexpr int W = 8;
    static constexpr int kTopics = 128 * NP;
    static constexpr size_t sp = 0;                                                // [W][kTopics] topic partials
    static constexpr size_t tt = sp + (size_t)W * kTopics * 8;                     // [2][kTopics]
    static constexpr size_t alf = tt + (size_t)2 * kTopics * 8;                    // [kTopics] alpha
    static constexpr size_t gpv = alf + (size_t)kTopics * 8;                       // [kTopics] gamma before the last update
    static constexpr size_t chg = gpv + (size_t)kTopics * 8;                       // u64[2]
    static constexpr size_t misc = chg + 16;                                       // [8][W]
    static constexpr size_t ids = misc + (size_t)8 * W * 8;                        // int [W][kQfMaxSlots]
    static constexpr size_t cnt = ids + (size_t)W * kQfMaxSlots * 4;               // double [W][kQfMaxSlots]
    static constexpr size_t rr = cnt + (size_t)W * kQfMaxSlots * 8;                // double [W][kQfMaxSlots]  r of the last iteration
    static constexpr size_t total = rr + (size_t)W * kQfMaxSlots * 8;
    static_assert(total <= 160 * 1024, "fits the LDS");
};

// One 16-byte piece of a streamed row, requested NOW (asm: the compiler would sink the load to its first use).
template <int OFF>
__device__ __forceinline__ void piece_request(f64x2& dst, const void* ptr)
{
    static_assert(OFF == 0 || OFF == 1024 || OFF == 2048 || OFF == 3072, "piece offsets inside the 12-bit immediate");
    if constexpr (OFF == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(ptr) : "memory");
    else if constexpr (OFF == 1024) asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=&v"(dst) : "v"(ptr) : "memory");
    else if constexpr (OFF == 2048) asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=&v"(dst) : "v"(ptr) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off offset:3072" : "=&v"(dst) : "v"(ptr) : "memory");
}

template <int NP>
struct StreamRow {
    f64x2 p[NP];
    __device__ __forceinline__ void request(const char* ptr)
    {
        const char* hi = ptr + 4096;
        piece_request<0>(p[0], ptr);
        piece_request<1024>(p[1], ptr);
        piece_request<2048>(p[2], ptr);
        piece_request<3072>(p[3], ptr);
        piece_request<0>(p[4], hi);
        if constexpr (NP > 5) piece_request<1024>(p[5], hi);
        if constexpr (NP > 6) piece_request<2048>(p[6], hi);
        if constexpr (NP > 7) piece_request<3072>(p[7], hi);
    }
    // ROWS_NEWER: rows requested after this one (their NP loads each may stay outstanding)
    template <int ROWS_NEWER>
    __device__ __forceinline__ void wait()
    {
        static_assert(NP >= 5 && NP <= 8 && ROWS_NEWER >= 0 && ROWS_NEWER <= 2 && NP * ROWS_NEWER <= 16, "vmcnt immediates below");
        constexpr int OUT = NP * ROWS_NEWER;
        if constexpr (OUT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (OUT == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if constexpr (OUT == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (OUT == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else if constexpr (OUT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (OUT == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if constexpr (OUT == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if constexpr (OUT == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#pragma unroll
        for (int jj = 0; jj < NP; ++jj) asm volatile("" : "+v"(p[jj]));      // the values exist from here on
    }
};

template <int NP>
__global__ __launch_bounds__(512, 2) void estep_qfusek_kernel(EstepParams p)
{
    using L = QfusekLds<NP>;
    constexpr int W = 8, NT = 512, KT = 128 * NP, KRL = 2 * NP;
    constexpr int TPT = (KT + NT - 1) / NT;          // topics per thread in the gamma phase (2)
    constexpr bool PAIRS = NP <= 6;                  // four row buffers fit the registers: words go in pairs
    static_assert(NP >= 5 && NP <= 8 && TPT == 2, "table stride 640 .. 1024");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sp = reinterpret_cast<double*>(smem + L::sp);
    double* tt = reinterpret_cast<double*>(smem + L::tt);
    double* alf = reinterpret_cast<double*>(smem + L::alf);
    double* gpv = reinterpret_cast<double*>(smem + L::gpv);
    unsigned long long* chg = reinterpret_cast<unsigned long long*>(smem + L::chg);
    double* misc = reinterpret_cast<double*>(smem + L::misc);

    const int tid = threadIdx.x;
    const int c = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int K = p.K, ldk = p.ldk;
    const int doc = p.order[blockIdx.x];
    const int64_t lo = p.doc_ptr[doc];
    const int N = (int)(p.doc_ptr[doc + 1] - lo);
    const int S = (N + W - 1) / W;                          // word slots per wavefront: word n = slot * 8 + wave
    const int Spad = PAIRS ? (S + 3) & ~3 : (S + 1) & ~1;   // whole trips of four (pairs) / two
    int* myids = reinterpret_cast<int*>(smem + L::ids) + wave * kQfMaxSlots;
    double* mycnt = reinterpret_cast<double*>(smem + L::cnt) + wave * kQfMaxSlots;
    double* myrr = reinterpret_cast<double*>(smem + L::rr) + wave * kQfMaxSlots;
    const char* table = reinterpret_cast<const char*>(p.expElog) + (size_t)c * 16;
    const size_t row_bytes = (size_t)ldk * 8;

    // ---- word ids / counts of this wavefront's slots, token total (:162) ----
    double local = 0.0;
    for (int s = c; s < Spad; s += kWave) {
        const int n = s * W + wave;
        const bool live = n < N;
        myids[s] = live ? p.term_id[lo + n] : 0;            // dead slots: a valid row, count 0 => r = 0
        const double ct = live ? (double)p.term_ct[lo + n] : 0.0;
        mycnt[s] = ct;
        myrr[s] = 0.0;
        local += ct;
    }
    local = wave_sum(local);
    double asum = 0.0;
    for (int k = c; k < K; k += kWave) asum += p.alpha[k];
    asum = wave_sum(asum);
    bool topic_thread[TPT], topic_live[TPT];
#pragma unroll
    for (int u = 0; u < TPT; ++u) {
        const int k = tid + u * NT;
        topic_thread[u] = k < KT;
        topic_live[u] = k < K;
        if (topic_thread[u]) alf[k] = topic_live[u] ? p.alpha[k] : 1.0;
    }
    if (c == 0) misc[wave] = local;
    if (tid == 0) chg[0] = chg[1] = 0ull;
    __syncthreads();                                        // also: myids / mycnt are in place

    double total = 0.0;
#pragma unroll
    for (int w = 0; w < W; ++w) total += misc[w];
    const double psi_total = uniform_f64(digamma(asum + total));
    double gam[TPT];
#pragma unroll
    for (int u = 0; u < TPT; ++u) {
        const int k = tid + u * NT;
        gam[u] = 1.0;
        if (topic_thread[u]) {
            gam[u] = topic_live[u] ? alf[k] + total / K : 1.0;                // :165 (padding topics never move)
            tt[k] = topic_live[u] ? exp_digamma_minus(gam[u], psi_total) : 0.0;
        }
    }
    __syncthreads();

    int it = 0;
    int bad = 0;
    auto row_address = [&](int slot) { return table + (size_t)myids[slot] * row_bytes; };
    while (it < p.max_iter) {                                             // :174
        const int buf = it & 1;
        double tq[KRL];
#pragma unroll
        for (int jj = 0; jj < NP; ++jj) {
            const double2 t2 = reinterpret_cast<const double2*>(tt + buf * KT)[c + 64 * jj];
            tq[2 * jj] = t2.x;
            tq[2 * jj + 1] = t2.y;
        }
        double q[KRL];
#pragma unroll
        for (int j = 0; j < KRL; ++j) q[j] = 0.0;
        // one word, fused: normaliser (wavefront sum of the lanes' dots), r, topic sums
        auto word = [&](const StreamRow<NP>& g, int slot) {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int jj = 0; jj < NP; ++jj) {
                a0 = fma(g.p[jj].x, tq[2 * jj], a0);
                a1 = fma(g.p[jj].y, tq[2 * jj + 1], a1);
            }
            const double cnt = mycnt[slot];
            const double nrm = wave_sum(a0 + a1);
            const bool live = cnt > 0.0;
            if (live && !(nrm > 1e-280 && nrm < 1e300)) bad = 1;
            const double r = live ? cnt * rcp_newton(nrm) : 0.0;
            if (c == 0) myrr[slot] = r;
#pragma unroll
            for (int jj = 0; jj < NP; ++jj) {
                q[2 * jj] = fma(r, g.p[jj].x, q[2 * jj]);
                q[2 * jj + 1] = fma(r, g.p[jj].y, q[2 * jj + 1]);
            }
        };
        if constexpr (PAIRS) {
            // words in pairs (estep_qfuse.h): one swap level folds the two words' per-lane dots into one register, the
            // remaining reduction levels, the range check and the reciprocal run once for both; four row buffers, the
            // next pair in flight while this one is reduced (up to stride 768: 4 x 12 VGPR pairs)
            auto word_pair = [&](const StreamRow<NP>& ga, const StreamRow<NP>& gb, int slot) {
                double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
                for (int jj = 0; jj < NP; ++jj) {
                    a0 = fma(ga.p[jj].x, tq[2 * jj], a0);
                    a1 = fma(ga.p[jj].y, tq[2 * jj + 1], a1);
                    b0 = fma(gb.p[jj].x, tq[2 * jj], b0);
                    b1 = fma(gb.p[jj].y, tq[2 * jj + 1], b1);
                }
                const double nrm = half_wave_sum(swap32_add(a0 + a1, b0 + b1));
                const double cnt = mycnt[slot + (c >> 5)];
                const bool live = cnt > 0.0;
                if (live && !(nrm > 1e-280 && nrm < 1e300)) bad = 1;
                const double r = live ? cnt * rcp_newton(nrm) : 0.0;
                if ((c & 31) == 0) myrr[slot + (c >> 5)] = r;
                const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(r), __double2loint(r), false, false);
                const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(r), __double2hiint(r), false, false);
                const double rA = __hiloint2double(hi[0], lo[0]), rB = __hiloint2double(hi[1], lo[1]);
#pragma unroll
                for (int jj = 0; jj < NP; ++jj) {
                    q[2 * jj] = fma(rA, ga.p[jj].x, q[2 * jj]);
                    q[2 * jj + 1] = fma(rA, ga.p[jj].y, q[2 * jj + 1]);
                }
#pragma unroll
                for (int jj = 0; jj < NP; ++jj) {
                    q[2 * jj] = fma(rB, gb.p[jj].x, q[2 * jj]);
                    q[2 * jj + 1] = fma(rB, gb.p[jj].y, q[2 * jj + 1]);
                }
            };
            if (Spad > 0) {
                StreamRow<NP> g0, g1, g2, g3;
                g0.request(row_address(0));
                g1.request(row_address(1));
                g2.request(row_address(2));
                g3.request(row_address(3));
                int s = 0;
                for (; s + 4 < Spad; s += 4) {      // full trips: a pair of buffers is re-requested four slots ahead
                    g0.template wait<2>(); g1.template wait<2>(); word_pair(g0, g1, s + 0);
                    g0.request(row_address(s + 4)); g1.request(row_address(s + 5)); __builtin_amdgcn_sched_barrier(0);
                    g2.template wait<2>(); g3.template wait<2>(); word_pair(g2, g3, s + 2);
                    g2.request(row_address(s + 6)); g3.request(row_address(s + 7)); __builtin_amdgcn_sched_barrier(0);
                }
                g0.template wait<2>(); g1.template wait<2>(); word_pair(g0, g1, s + 0);      // last trip: the pipeline drains
                __builtin_amdgcn_sched_barrier(0);
                g2.template wait<0>(); g3.template wait<0>(); word_pair(g2, g3, s + 2);
            }
        } else if (Spad > 0) {
            StreamRow<NP> g0, g1;
            g0.request(row_address(0));
            g1.request(row_address(1));
            int s = 0;
            for (; s + 2 < Spad; s += 2) {          // full trips: a buffer is re-requested two slots ahead
                g0.template wait<1>(); word(g0, s + 0); g0.request(row_address(s + 2)); __builtin_amdgcn_sched_barrier(0);
                g1.template wait<1>(); word(g1, s + 1); g1.request(row_address(s + 3)); __builtin_amdgcn_sched_barrier(0);
            }
            g0.template wait<1>(); word(g0, s + 0); __builtin_amdgcn_sched_barrier(0);      // last trip: the pipeline drains
            g1.template wait<0>(); word(g1, s + 1);
        }
        // per-wavefront topic partials: lane c, register j  <->  topic 2c + 128*(j>>1) + (j&1)
#pragma unroll
        for (int jj = 0; jj < NP; ++jj)
            reinterpret_cast<double2*>(sp + (size_t)wave * KT)[c + 64 * jj] = double2{q[2 * jj], q[2 * jj + 1]};
        __syncthreads();

        // C. gamma update: two topics per thread
#pragma unroll
        for (int u = 0; u < TPT; ++u) {
            const int k = tid + u * NT;
            if (topic_thread[u]) {
                double part[W];
#pragma unroll
                for (int w = 0; w < W; ++w) part[w] = sp[w * KT + k];
                const double t_mine = tt[buf * KT + k], alpha_k = alf[k];
                const double s0 = (part[0] + part[1]) + (part[4] + part[5]), s1 = (part[2] + part[3]) + (part[6] + part[7]);
                const double gnew = fma(t_mine, s0 + s1, alpha_k);            // :185
                const double diff = topic_live[u] ? fabs(gnew - gam[u]) : 0.0;   // :187
                gpv[k] = gam[u];
                gam[u] = gnew;                                                // :188
                atomicAdd(&chg[buf], change_fixed(diff));
                const double t_next = exp_digamma_minus_levels(gam[u], psi_total);
                tt[(buf ^ 1) * KT + k] = topic_live[u] ? t_next : 0.0;
            }
        }
        if (tid == 0) store_u64_hi(&chg[buf ^ 1], 0u);
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

    // ---- training fast path: the document terms are left to doc_terms_kernel (doc_terms.h; see estep_quad.h) ----
    if (!p.heldout && !p.want_doc_ll) {
        for (int s = c; s < S; s += kWave)
            if (s * W + wave < N) p.rfinal[lo + s * W + wave] = myrr[s];
#pragma unroll
        for (int u = 0; u < TPT; ++u) {
            const int k = tid + u * NT;
            if (topic_thread[u]) {
                if (topic_live[u]) p.gamma[(size_t)doc * K + k] = gam[u];
                p.tfinal[(size_t)doc * ldk + k] = topic_live[u] ? tt[last * KT + k] : 0.0;
            }
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
        for (int jj = 0; jj < NP; ++jj) {
            const double2 t2 = reinterpret_cast<const double2*>(tt + last * KT)[c + 64 * jj];
            tq[2 * jj] = t2.x;
            tq[2 * jj + 1] = t2.y;
        }
        const double2* gtable = reinterpret_cast<const double2*>(p.expElog_elog);
        const int ldk2 = ldk / 2;
        for (int s = 0; s < S; ++s) {
            const double2* row = gtable + (size_t)myids[s] * ldk2 + c;
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int jj = 0; jj < NP; ++jj) {
                const double2 g2 = row[64 * jj];
                a0 = fma(g2.x, tq[2 * jj], a0);
                a1 = fma(g2.y, tq[2 * jj + 1], a1);
            }
            term1 = fma(myrr[s], a0 + a1, term1);          // r = 0 for dead slots; summed over the lanes below
        }
    }
    double term3 = 0.0, shift_term = 0.0;
    for (int s = c; s < S; s += kWave) {
        const int n = s * W + wave;
        if (n < N) {
            const double cnt = mycnt[s], r = myrr[s];
            term3 = fma(cnt, log(cnt) - log(r), term3);    // c_n log(normaliser_n), normaliser = c_n / r_n
            if (p.heldout) shift_term = fma(cnt, p.shift[myids[s]], shift_term);
            else p.rfinal[lo + n] = r;
        }
    }
    TopicShare share;
#pragma unroll
    for (int u = 0; u < TPT; ++u) {
        const int k = tid + u * NT;
        if (topic_thread[u])
            topic_share(p, doc, k, ldk, topic_live[u], true, gam[u], alf[k], gpv[k], tt[last * KT + k], psi_total, share);
    }
    finish_document<W>(p, doc, it, misc, c, wave, tid, term1, term3, shift_term, share);
}

}  // namespace pylda
