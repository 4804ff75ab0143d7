// Fused streaming E-step kernel for 256 < K <= 512 (table stride 384 or 512 = 128 * NP; cfg 5:
// nips.88-05 at K = 500, N_d ~ 230, tile 230 x 4 KiB = 920 KB - more than a CU holds).
//
// Round 2's two-pass streaming kernel (retired) re-read the whole tile from L2 / Infinity Cache
// twice per inner iteration; at K = 500 every CU streamed 1.8 MB per document-iteration and the chip was
// bound by the fabric (6.8 TB/s in aggregate, 95 % of wave-cycles waiting:
// profiles/r02_nips_k500_qstream_rocprof_summary.txt).  Two observations cut that traffic ~3x:
//
//   * a word's normaliser needs only ITS row and t:  nrm_n = sum_k B[n][k] t[k],  r_n = c_n / nrm_n,
//     so with all 512 topics of a row inside ONE wavefront (64 lanes x 8 values) normaliser and
//     topic sums fuse:  row -> dot -> wavefront sum -> r -> q += r * row.  Each row is read ONCE per
//     iteration, and the normaliser reduction is a wavefront-local DPP / permlane sum (no LDS
//     transposes, no second pass).  The extra VALU work (one 64-lane reduction per word) is free
//     here: the kernel waits for memory;
//   * part of the tile stays on chip: the first RWL slots of every wavefront in VGPRs (8 x 8 = 64
//     words, 128 VGPRs), the next TWL slots as whole rows in LDS (8 x 2 = 16 words, 64 KiB).
//
// Layout: 8 wavefronts per document, word n belongs to wavefront n % 8, slot n / 8; lane c holds
// topics 2c + 128*jj + {0,1}, jj < 4, of a row (16-byte pieces 1 KiB apart).  Streamed rows go
// through FOUR 16-VGPR buffers per wavefront, each requested four slots ahead by an asm statement and
// handed over by its s_waitcnt vmcnt (the loop has no other vector-memory traffic), i.e. 128 KiB in
// flight per CU.
//
// Cross-wavefront: per-wavefront topic partials in LDS -> barrier -> 512 topic threads (all eight
// wavefronts): gamma update, exp(psi), fixed-point convergence sum (estep_quilt.h) -> barrier.
#pragma once
#include "estep_common.h"
#include "special_device.h"
#include "estep_epilogue.h"
#include "estep_limits.h"

namespace pylda {


template <int NP, int TWL>
struct QfuseLds {
    static constexpr int W = 8;
    static constexpr int kTopics = 128 * NP;
    static constexpr size_t sp = 0;                                                // [W][kTopics] topic partials
    static constexpr size_t tt = sp + (size_t)W * kTopics * 8;                     // [2][kTopics]
    static constexpr size_t alf = tt + (size_t)2 * kTopics * 8;                    // [kTopics] alpha
    static constexpr size_t gpv = alf + (size_t)kTopics * 8;                       // [kTopics] gamma before the last update
    static constexpr size_t chg = gpv + (size_t)kTopics * 8;                       // u64[2]
    static constexpr size_t misc = chg + 16;                                       // [8][W]
    static constexpr size_t ids = misc + (size_t)8 * W * 8;                        // u64 [W][kQfMaxSlots] byte offset of the slot's row in the table
    static constexpr size_t cnt = ids + (size_t)W * kQfMaxSlots * 8;               // double [W][kQfMaxSlots]
    static constexpr size_t rr = cnt + (size_t)W * kQfMaxSlots * 8;                // double [W][kQfMaxSlots]  r of the last iteration
    static constexpr size_t rows = (rr + (size_t)W * kQfMaxSlots * 8 + 255) & ~(size_t)255;   // [W][TWL][kTopics]
    static constexpr size_t total = rows + (size_t)W * TWL * kTopics * 8;
    static_assert(total <= 160 * 1024, "fits the LDS");
};

template <int NP>
struct GlobalRow {
    f64x2 p[NP];
    __device__ __forceinline__ void unpack(double (&row)[2 * NP]) const
    {
#pragma unroll
        for (int jj = 0; jj < NP; ++jj) {
            row[2 * jj] = p[jj].x;
            row[2 * jj + 1] = p[jj].y;
        }
    }
};
// the NP 16-byte pieces (1 KiB apart) of a table row, requested now, usable after global_row_wait<ROWS_NEWER>
__device__ __forceinline__ void global_row_request(GlobalRow<4>& r, const void* ptr)
{
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"
                 "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072"
                 : "=&v"(r.p[0]), "=&v"(r.p[1]), "=&v"(r.p[2]), "=&v"(r.p[3])
                 : "v"(ptr)
                 : "memory");
}
__device__ __forceinline__ void global_row_request(GlobalRow<3>& r, const void* ptr)
{
    asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:1024\n\t"
                 "global_load_dwordx4 %2, %3, off offset:2048"
                 : "=&v"(r.p[0]), "=&v"(r.p[1]), "=&v"(r.p[2])
                 : "v"(ptr)
                 : "memory");
}
// ROWS_NEWER: rows requested after this one (their loads may stay outstanding)
template <int ROWS_NEWER>
__device__ __forceinline__ void global_row_wait(GlobalRow<4>& r)
{
    static_assert(ROWS_NEWER >= 0 && ROWS_NEWER <= 3, "four buffers");
    if constexpr (ROWS_NEWER == 3) asm volatile("s_waitcnt vmcnt(12)" : "+v"(r.p[0]), "+v"(r.p[1]), "+v"(r.p[2]), "+v"(r.p[3]) : : "memory");
    else if constexpr (ROWS_NEWER == 2) asm volatile("s_waitcnt vmcnt(8)" : "+v"(r.p[0]), "+v"(r.p[1]), "+v"(r.p[2]), "+v"(r.p[3]) : : "memory");
    else if constexpr (ROWS_NEWER == 1) asm volatile("s_waitcnt vmcnt(4)" : "+v"(r.p[0]), "+v"(r.p[1]), "+v"(r.p[2]), "+v"(r.p[3]) : : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.p[0]), "+v"(r.p[1]), "+v"(r.p[2]), "+v"(r.p[3]) : : "memory");
}
template <int ROWS_NEWER>
__device__ __forceinline__ void global_row_wait(GlobalRow<3>& r)
{
    static_assert(ROWS_NEWER >= 0 && ROWS_NEWER <= 3, "four buffers");
    if constexpr (ROWS_NEWER == 3) asm volatile("s_waitcnt vmcnt(9)" : "+v"(r.p[0]), "+v"(r.p[1]), "+v"(r.p[2]) : : "memory");
    else if constexpr (ROWS_NEWER == 2) asm volatile("s_waitcnt vmcnt(6)" : "+v"(r.p[0]), "+v"(r.p[1]), "+v"(r.p[2]) : : "memory");
    else if constexpr (ROWS_NEWER == 1) asm volatile("s_waitcnt vmcnt(3)" : "+v"(r.p[0]), "+v"(r.p[1]), "+v"(r.p[2]) : : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.p[0]), "+v"(r.p[1]), "+v"(r.p[2]) : : "memory");
}

// sum_j row[j] * t[j] over a lane's values of one word
__device__ __forceinline__ double lane_dot(const double (&row)[8], const double (&t)[8]) { return dot8(row, t); }
__device__ __forceinline__ double lane_dot(const double (&row)[6], const double (&t)[6])
{
    const double a = fma(row[4], t[4], fma(row[2], t[2], row[0] * t[0]));
    const double b = fma(row[5], t[5], fma(row[3], t[3], row[1] * t[1]));
    return a + b;
}

// Sum over each 32-lane half of the wavefront, result in every lane of the half: wave_sum (estep_common.h) without
// its last level - four DPP levels inside the 16-lane rows, one permlane16 swap across the two rows of a half.
__device__ __forceinline__ double half_wave_sum(double v)
{
#define PYLDA_DPP_ADD(CTRL)                                                                     \
    v += __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false),   \
                          __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false))
    PYLDA_DPP_ADD(0xB1);        // quad_perm [1,0,3,2]
    PYLDA_DPP_ADD(0x4E);        // quad_perm [2,3,0,1]
    PYLDA_DPP_ADD(0x141);       // row_half_mirror
    PYLDA_DPP_ADD(0x140);       // row_mirror
#undef PYLDA_DPP_ADD
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);       // rows 0+1 | 0+1 | 2+3 | 2+3
}

template <int NP, int RWL, int TWL>
__global__ __launch_bounds__(512, 2) void estep_qfuse_kernel(EstepParams p)
{
    using L = QfuseLds<NP, TWL>;
    constexpr int W = 8, NT = 512, KT = 128 * NP, KRL = 2 * NP;
    static_assert(NP == 3 || NP == 4, "table stride 384 or 512");
    static_assert(RWL % 2 == 0 && TWL % 2 == 0, "words are processed in pairs");
    static_assert((RWL + TWL) % 4 == 0, "the on-chip words go four to a reduction");
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
    constexpr int kOnChip = RWL + TWL;
    const int NS = S > kOnChip ? (S - kOnChip + 3) & ~3 : 0;   // streamed slots, padded to whole trips of four
    const int Spad = kOnChip + NS;
    unsigned long long* myoff = reinterpret_cast<unsigned long long*>(smem + L::ids) + wave * kQfMaxSlots;
    double* mycnt = reinterpret_cast<double*>(smem + L::cnt) + wave * kQfMaxSlots;
    double* myrr = reinterpret_cast<double*>(smem + L::rr) + wave * kQfMaxSlots;
    double2* myrows = reinterpret_cast<double2*>(smem + L::rows) + (size_t)wave * TWL * (KT / 2) + c;
    // this lane's first piece of row 0: a row's address is one 64-bit add of the slot's byte offset (kept in LDS)
    const char* table = reinterpret_cast<const char*>(reinterpret_cast<const double2*>(p.expElog) + c);
    const unsigned long long row_bytes = (unsigned long long)ldk * 8;

    // ---- word ids / counts of this wavefront's slots, token total (:162) ----
    double local = 0.0;
    for (int s = c; s < Spad; s += kWave) {
        const int n = s * W + wave;
        const bool live = n < N;
        // dead slots: the document's own first term with count 0 (r = 0 without a select, and a normaliser that can only
        // leave the fp64 range when a real term's does - row 0 of the table is not a word of the document, and ITS
        // normaliser may underflow on a healthy document, which would send the document to the log-space kernel)
        myoff[s] = (N > 0 ? (unsigned long long)p.term_id[lo + (live ? n : 0)] : 0ull) * row_bytes;     // (an empty document: row 0)
        const double ct = live ? (double)p.term_ct[lo + n] : 0.0;
        mycnt[s] = ct;
        myrr[s] = 0.0;
        local += ct;
    }
    local = wave_sum(local);
    double asum = 0.0;
    for (int k = c; k < K; k += kWave) asum += p.alpha[k];
    asum = wave_sum(asum);
    const bool topic_thread = tid < KT;
    const bool topic_live = tid < K;
    if (topic_thread) alf[tid] = topic_live ? p.alpha_sgn[tid] : 1.0;      // (alpha; sign bit: the topic never counts as dead, kMortalT)
    if (c == 0) misc[wave] = local;
    // live topics after an iteration (gamma_k != alpha_k bitwise), for the hand-over to the live-topic kernel; and the
    // per-wavefront counts of that exit - both in the idle part of misc
    unsigned* livec = reinterpret_cast<unsigned*>(misc + 2 * W);
    unsigned* wcount = reinterpret_cast<unsigned*>(misc + 3 * W);
    if (tid == 0) {
        chg[0] = chg[1] = 0ull;
        livec[0] = livec[1] = 0u;
    }
    const int handoff_at = handoff_threshold(p, N);
    __syncthreads();                                        // also: myoff / mycnt are in place

    // ---- on-chip tiers: registers, LDS rows ----
    double B[RWL][KRL];
#pragma unroll
    for (int i = 0; i < RWL; ++i) {
        const double2* row = reinterpret_cast<const double2*>(table + myoff[i]);
#pragma unroll
        for (int jj = 0; jj < NP; ++jj) {
            const double2 v2 = row[64 * jj];
            B[i][2 * jj] = v2.x;
            B[i][2 * jj + 1] = v2.y;
        }
    }
#pragma unroll
    for (int t = 0; t < TWL; ++t) {
        const double2* row = reinterpret_cast<const double2*>(table + myoff[RWL + t]);
#pragma unroll
        for (int jj = 0; jj < NP; ++jj) myrows[t * (KT / 2) + 64 * jj] = row[64 * jj];
    }
    double total = 0.0;
#pragma unroll
    for (int w = 0; w < W; ++w) total += misc[w];
    const double psi_total = uniform_f64(digamma(asum + total));
    double gam = 1.0;
    if (topic_thread) {
        gam = topic_live ? fabs(alf[tid]) + total / K : 1.0;              // :165 (padding topics never move)
        tt[tid] = topic_live ? exp_digamma_minus(gam, psi_total) : 0.0;
    }
    __syncthreads();                                        // t is published; the LDS rows are written (same wavefront reads them)

    int it = 0;
    int bad = 0;
    auto row_address = [&](int slot) { return (const void*)(table + myoff[slot]); };
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
        // TWO words (slots `slot`, `slot + 1`), fused: normalisers, r, topic sums.  A word costs ~20 instructions of
        // wavefront reduction and reciprocal around its 2 x 8 FMAs, and this kernel is bound by exactly that issue
        // count (on-chip capacity barely matters: 24 or 56 words on chip, 5.15 vs 4.97 ms at K = 500).  In pairs: one
        // swap level folds the two words' per-lane dots into ONE register (word A's partial sums in lanes 0-31, B's in
        // 32-63), so the five remaining reduction levels, the range check and the reciprocal run once for both.
        auto word_pair = [&](const double (&rowA)[KRL], const double (&rowB)[KRL], int slot) {
            const double nrm = half_wave_sum(swap32_add(lane_dot(rowA, tq), lane_dot(rowB, tq)));
            // (B, t <= 1: a normaliser cannot overflow; NaN fails the compare; an empty slot reads the document's first term
            //  with count 0: the normaliser of a real word, r = 0 without a select)
            const double cnt = mycnt[slot + (c >> 5)];
            if (!(nrm > 1e-280)) bad = 1;
            const double r = cnt * rcp_newton(nrm);
            if ((c & 31) == 0) myrr[slot + (c >> 5)] = r;
            const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(r), __double2loint(r), false, false);
            const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(r), __double2hiint(r), false, false);
            const double rA = __hiloint2double(hi[0], lo[0]), rB = __hiloint2double(hi[1], lo[1]);   // every lane: r of A, of B
#pragma unroll
            for (int j = 0; j < KRL; ++j) q[j] = fma(rA, rowA[j], q[j]);
#pragma unroll
            for (int j = 0; j < KRL; ++j) q[j] = fma(rB, rowB[j], q[j]);
        };
        // FOUR on-chip words (slots slot .. slot + 3): two swap levels fold the four per-lane dots into one register -
        // 16-lane row 0 holds word A's partial sums, row 1 C's, row 2 B's, row 3 D's - so the four remaining reduction
        // levels, the range check and the reciprocal run once for all four, and three swaps hand every r to every lane:
        // 21 + 12 + 6 instructions around the 4 x (11 + 8) of the dots and the topic sums, where two pairs take 36 + 24 + 4.
        // (The streamed words stay in pairs: four rows in use and none in flight would expose every L2 round trip.)
        auto word_quad = [&](const double (&rowA)[KRL], const double (&rowB)[KRL], const double (&rowC)[KRL],
                             const double (&rowD)[KRL], int slot) {
            double nrm = swap16_add(swap32_add(lane_dot(rowA, tq), lane_dot(rowB, tq)),
                                    swap32_add(lane_dot(rowC, tq), lane_dot(rowD, tq)));
            nrm = lane_group_sum<8>(nrm);
            nrm += dpp_f64<0x140>(nrm);                       // row_mirror: the other half of the 16-lane row
            const int mine = ((c >> 4) & 1) * 2 + (c >> 5);    // the word whose normaliser this lane's row holds: A, C, B, D
            const double cnt = mycnt[slot + mine];
            if (!(nrm > 1e-280)) bad = 1;
            const double r = cnt * rcp_newton(nrm);
            if ((c & 15) == 0) myrr[slot + mine] = r;
            const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(r), __double2loint(r), false, false);
            const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(r), __double2hiint(r), false, false);
            // [0]: rows A C A C, [1]: rows B D B D; a 16-lane swap of each with itself leaves one word's r in every lane
            const auto aclo = __builtin_amdgcn_permlane16_swap(lo[0], lo[0], false, false);
            const auto achi = __builtin_amdgcn_permlane16_swap(hi[0], hi[0], false, false);
            const auto bdlo = __builtin_amdgcn_permlane16_swap(lo[1], lo[1], false, false);
            const auto bdhi = __builtin_amdgcn_permlane16_swap(hi[1], hi[1], false, false);
            const double rA = __hiloint2double(achi[0], aclo[0]), rC = __hiloint2double(achi[1], aclo[1]);
            const double rB = __hiloint2double(bdhi[0], bdlo[0]), rD = __hiloint2double(bdhi[1], bdlo[1]);
#pragma unroll
            for (int j = 0; j < KRL; ++j) q[j] = fma(rA, rowA[j], q[j]);
#pragma unroll
            for (int j = 0; j < KRL; ++j) q[j] = fma(rB, rowB[j], q[j]);
#pragma unroll
            for (int j = 0; j < KRL; ++j) q[j] = fma(rC, rowC[j], q[j]);
#pragma unroll
            for (int j = 0; j < KRL; ++j) q[j] = fma(rD, rowD[j], q[j]);
        };
        auto lds_row = [&](int t, double (&row)[KRL]) {
#pragma unroll
            for (int jj = 0; jj < NP; ++jj) {
                const double2 v2 = myrows[t * (KT / 2) + 64 * jj];
                row[2 * jj] = v2.x;
                row[2 * jj + 1] = v2.y;
            }
        };
        static_for<(RWL + TWL) / 4>([&](auto idx) {
            constexpr int i = 4 * decltype(idx)::value;       // slots i .. i + 3: registers while i + x < RWL, LDS rows beyond
            if constexpr (i + 4 <= RWL) {
                word_quad(B[i], B[i + 1], B[i + 2], B[i + 3], i);
            } else if constexpr (i >= RWL) {
                double r0[KRL], r1[KRL], r2[KRL], r3[KRL];
                lds_row(i - RWL, r0); lds_row(i + 1 - RWL, r1); lds_row(i + 2 - RWL, r2); lds_row(i + 3 - RWL, r3);
                word_quad(r0, r1, r2, r3, i);
            } else {
                static_assert(i + 4 <= RWL || i >= RWL || i + 2 == RWL, "two register words + two LDS rows");
                double r2[KRL], r3[KRL];
                lds_row(0, r2); lds_row(1, r3);
                word_quad(B[i], B[i + 1], r2, r3, i);
            }
            __builtin_amdgcn_sched_barrier(0);                // one group's chains at a time (registers)
        });
        if (NS > 0) {
            // (requested here, not before the on-chip words: with the buffers live across those the kernel
            //  spills - and a spilled in-flight buffer would be stored before its data lands)
            GlobalRow<NP> g0, g1, g2, g3;
            global_row_request(g0, row_address(kOnChip + 0));
            global_row_request(g1, row_address(kOnChip + 1));
            global_row_request(g2, row_address(kOnChip + 2));
            global_row_request(g3, row_address(kOnChip + 3));
            int s = kOnChip;
            double rowA[KRL], rowB[KRL];
            for (; s + 4 < Spad; s += 4) {          // full trips: a pair of buffers is re-requested four slots ahead
                global_row_wait<2>(g0); global_row_wait<2>(g1); g0.unpack(rowA); g1.unpack(rowB); word_pair(rowA, rowB, s + 0);
                global_row_request(g0, row_address(s + 4)); global_row_request(g1, row_address(s + 5)); __builtin_amdgcn_sched_barrier(0);
                global_row_wait<2>(g2); global_row_wait<2>(g3); g2.unpack(rowA); g3.unpack(rowB); word_pair(rowA, rowB, s + 2);
                global_row_request(g2, row_address(s + 6)); global_row_request(g3, row_address(s + 7)); __builtin_amdgcn_sched_barrier(0);
            }
            global_row_wait<2>(g0); global_row_wait<2>(g1); g0.unpack(rowA); g1.unpack(rowB); word_pair(rowA, rowB, s + 0);   // last trip: the pipeline drains
            __builtin_amdgcn_sched_barrier(0);
            global_row_wait<0>(g2); global_row_wait<0>(g3); g2.unpack(rowA); g3.unpack(rowB); word_pair(rowA, rowB, s + 2);
        }
        // per-wavefront topic partials: lane c, register j  <->  topic 2c + 128*(j>>1) + (j&1)
#pragma unroll
        for (int jj = 0; jj < NP; ++jj)
            reinterpret_cast<double2*>(sp + (size_t)wave * KT)[c + 64 * jj] = double2{q[2 * jj], q[2 * jj + 1]};
        __syncthreads();

        // C. gamma update: one thread per topic
        if (topic_thread) {
            double part[W];
#pragma unroll
            for (int w = 0; w < W; ++w) part[w] = sp[w * KT + tid];
            const double t_mine = tt[buf * KT + tid], alpha_k = alf[tid];
            keep_together(part);
            const double s0 = (part[0] + part[1]) + (part[4] + part[5]), s1 = (part[2] + part[3]) + (part[6] + part[7]);
            const double gnew = fma(t_mine, s0 + s1, fabs(alpha_k));      // :185 (the sign bit: a topic that never counts as dead)
            const double diff = topic_live ? fabs(gnew - gam) : 0.0;      // :187
            gpv[tid] = gam;
            gam = gnew;                                                   // :188
            atomicAdd(&chg[buf], change_fixed(diff));
            {
                const unsigned long long moving = __ballot(topic_live && gnew != alpha_k);
                if (c == 0) atomicAdd(&livec[buf], (unsigned)__builtin_popcountll(moving));
            }
            const double t_next = exp_digamma_minus_levels(gam, psi_total)   /* (tables fetched in place: no scalar registers to spare across pass B) */;
            tt[(buf ^ 1) * KT + tid] = topic_live ? t_next : 0.0;
            if (tid == 0) {
                store_u64_hi(&chg[buf ^ 1], 0u);
                livec[buf ^ 1] = 0u;
            }
        }
        ++it;
        __syncthreads();
        const double change = (double)chg[buf] * (1.0 / kChangeScale);
        if (change <= p.tol * K) break;                                   // :189 (mean <= tol)
        // few enough topics still move: the live-topic kernel (estep_compact.h) runs the remaining iterations on them
        if (it < p.max_iter && __builtin_amdgcn_readfirstlane((int)livec[buf]) <= handoff_at && !__syncthreads_or(bad)) {
            stream_hand_over(p, doc, tid, topic_live, gam, alf[tid < KT ? tid : 0], wcount, W, it);
            return;
        }
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
        for (int s = c; s < S; s += kWave)
            if (s * W + wave < N) p.rfinal[lo + s * W + wave] = myrr[s];
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
        double tq[KRL];
#pragma unroll
        for (int jj = 0; jj < NP; ++jj) {
            const double2 t2 = reinterpret_cast<const double2*>(tt + last * KT)[c + 64 * jj];
            tq[2 * jj] = t2.x;
            tq[2 * jj + 1] = t2.y;
        }
        const char* gtable = reinterpret_cast<const char*>(reinterpret_cast<const double2*>(p.expElog_elog) + c);
        for (int s = 0; s < S; ++s) {
            const double2* row = reinterpret_cast<const double2*>(gtable + myoff[s]);
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
            if (p.heldout) shift_term = fma(cnt, p.shift[p.term_id[lo + n]], shift_term);
            else p.rfinal[lo + n] = r;
        }
    }
    TopicShare share;
    if (topic_thread)
        topic_share(p, doc, tid, ldk, topic_live, true, gam, fabs(alf[tid]), gpv[tid], tt[last * KT + tid], psi_total, share);
    finish_document<W>(p, doc, it, misc, c, wave, tid, term1, term3, shift_term, share);
}

}  // namespace pylda
