// Register + LDS tile E-step kernel with the tile cut into TOPIC BANDS: wavefront w of a document owns topics
// [32 w, 32 w + 32) of EVERY word (table stride 128: 4 wavefronts, two documents per CU; stride 256: 8, one per CU).
//
// Why.  The quad kernel (estep_quad.h) cuts the tile by WORDS: every wavefront holds all K topics of a sixteenth
// of the words, so the topic sums q_k = sum_n r_n B[n][k] cross the wavefronts (partials through LDS, barrier 1),
// the gamma update and exp(psi(gamma)) run on the K / 64 wavefronts that own a topic each while the others wait,
// and the new t crosses back (LDS, barrier 2): an inner iteration is a chain of five LDS round trips and two
// workgroup barriers, and the kernel's time is that chain, not its instructions (a lone document: ~4200 ticks
// per iteration for ~1400 ticks of FMA issue).  Cut by topics, the chain is short:
//
//   pass A   a[n] = sum_{k in band} B[n][k] t[k]          in-lane dot products (16 topics per lane)
//            + ONE add across the two lanes of a word     (DPP), W partial normalisers per word -> LDS
//   -------- the only workgroup barrier of the iteration --------
//   finish   normaliser = sum of the W partials, r = count / normaliser      (every wavefront, for its own lanes' words)
//   pass B   q[k] += r_n B[n][k]                          plain FMAs, 16 accumulators per lane
//   reduce   q over the wavefront's 32 word groups        two permlane-swap levels + three DPP levels, no LDS
//   gamma    gamma_k = alpha_k + t_k q_k, t_k = exp(psi(gamma_k) - psi(sum gamma))     ALL wavefronts, one topic per lane
//   spread   t back to the lanes of the same wavefront    1 KiB of wavefront-private LDS, no barrier
//
// Every wavefront does the same work (nobody waits for a "gamma wavefront"), q, gamma and t never leave the
// wavefront, and the stop test's fixed-point sum is read one barrier later, where the quad kernel read it too
// ("behind the first half of the next iteration").
//
// Lane layout.  lane = 2 g + h: word group g (0..31), topic half h: the lane holds topics 32 w + 16 h + j, j < 16
// (128 contiguous bytes of a table row), of the words n = 32 s + g, s < NS = RW + TW word slots: RW slots in VGPRs
// (5: 160 registers), TW as 128-byte pieces in LDS.  The two lanes of a pair hold the SAME words in CROSSED register
// sets - set 2i is slot 2i + h, set 2i + 1 is slot 2i + 1 - h - so that
//   * one DPP add per PAIR of slots, S_i = a[2i] + (partner's a[2i+1]), leaves each lane with the band's partial
//     normaliser of "its" word of the pair (a reduce-scatter, not an all-reduce),
//   * the lane that finishes a word (R_i) uses it for its set 2i, the partner receives it by one DPP move (Q_i) and
//     uses it for ITS set 2i + 1: no selects anywhere.
// An odd last slot is held by both lanes in the same set and finished by both.
//
// After the reduction lane l holds topic j = (l >> 2) & 15 of its half (lanes l and l ^ 2 hold the same one; only
// the first writes).  Results are bitwise reproducible (fixed summation orders, integer stop test).
#pragma once
#include "estep_common.h"
#include "special_device.h"

namespace pylda {

template <int W, int RW, int TW>
struct BandLds {
    static constexpr int NS = RW + TW;
    static constexpr int kTopics = 32 * W;
    static constexpr size_t p_word = (size_t)W * 8;                          // a word's W partial normalisers
    // 32 word groups per slot + a shift that puts the two lanes of a pair (slots 2i / 2i + 1) on disjoint banks
    static constexpr size_t p_slot = 32 * p_word + (W == 4 ? 16 : 32);
    static constexpr size_t p_buf = ((size_t)NS * p_slot + 63) & ~(size_t)63;
    static constexpr size_t part = 0;                                        // [2][NS][32][W], by iteration parity
    static constexpr size_t tband = 2 * p_buf;                               // [W][32] t of the wavefront's band (private to it)
    static constexpr size_t chg = tband + (size_t)kTopics * 8;               // u64[4], by iteration mod 4
    static constexpr size_t misc = chg + 32;                                 // [8][W]
    static constexpr size_t rows = (misc + (size_t)8 * W * 8 + 255) & ~(size_t)255;   // [TW][W][8 pieces][64 lanes] x 16 B
    static constexpr size_t row_slot = (size_t)W * 8 * 64 * 16;
    static constexpr size_t total = rows + (size_t)TW * row_slot;
    static_assert(W != 4 || 2 * total <= 160 * 1024, "stride 128: two workgroups per CU");
    static_assert(total <= 160 * 1024, "fits the LDS");
};

// ---- hand-ordered FMA blocks of this layout (see estep_common.h: an asm statement is scheduled as a unit) ----
// a[e] = b_e * t  /  a[e] += b_e * t over the register sets at one topic column
template <int J, int RW>
__device__ __forceinline__ void band_col(double* a, const double (&B)[RW][16], double t)
{
    static_assert(RW == 4 || RW == 5, "register slots");
    if constexpr (RW == 5) {
        if constexpr (J == 0)
            asm("v_mul_f64 %0, %5, %10\n\tv_mul_f64 %1, %6, %10\n\tv_mul_f64 %2, %7, %10\n\tv_mul_f64 %3, %8, %10\n\t"
                "v_mul_f64 %4, %9, %10"
                : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4])
                : "v"(B[0][J]), "v"(B[1][J]), "v"(B[2][J]), "v"(B[3][J]), "v"(B[4][J]), "v"(t));
        else
            asm("v_fmac_f64_e32 %0, %5, %10\n\tv_fmac_f64_e32 %1, %6, %10\n\tv_fmac_f64_e32 %2, %7, %10\n\t"
                "v_fmac_f64_e32 %3, %8, %10\n\tv_fmac_f64_e32 %4, %9, %10"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4])
                : "v"(B[0][J]), "v"(B[1][J]), "v"(B[2][J]), "v"(B[3][J]), "v"(B[4][J]), "v"(t));
    } else {
        if constexpr (J == 0)
            asm("v_mul_f64 %0, %4, %8\n\tv_mul_f64 %1, %5, %8\n\tv_mul_f64 %2, %6, %8\n\tv_mul_f64 %3, %7, %8"
                : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3])
                : "v"(B[0][J]), "v"(B[1][J]), "v"(B[2][J]), "v"(B[3][J]), "v"(t));
        else
            asm("v_fmac_f64_e32 %0, %4, %8\n\tv_fmac_f64_e32 %1, %5, %8\n\tv_fmac_f64_e32 %2, %6, %8\n\t"
                "v_fmac_f64_e32 %3, %7, %8"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
                : "v"(B[0][J]), "v"(B[1][J]), "v"(B[2][J]), "v"(B[3][J]), "v"(t));
    }
}

// q[j] = r * b[j]  /  q[j] += r * b[j], eight topics at a time (an asm statement takes at most 30 operands)
template <bool FIRST>
__device__ __forceinline__ void band_row8(double* q, double r, const double* b)
{
    if constexpr (FIRST)
        asm("v_mul_f64 %0, %8, %9\n\tv_mul_f64 %1, %8, %10\n\tv_mul_f64 %2, %8, %11\n\tv_mul_f64 %3, %8, %12\n\t"
            "v_mul_f64 %4, %8, %13\n\tv_mul_f64 %5, %8, %14\n\tv_mul_f64 %6, %8, %15\n\tv_mul_f64 %7, %8, %16"
            : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7])
            : "v"(r), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
    else
        asm("v_fmac_f64_e32 %0, %8, %9\n\tv_fmac_f64_e32 %1, %8, %10\n\tv_fmac_f64_e32 %2, %8, %11\n\t"
            "v_fmac_f64_e32 %3, %8, %12\n\tv_fmac_f64_e32 %4, %8, %13\n\tv_fmac_f64_e32 %5, %8, %14\n\t"
            "v_fmac_f64_e32 %6, %8, %15\n\tv_fmac_f64_e32 %7, %8, %16"
            : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])
            : "v"(r), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
}

// Half of an LDS word (4 of a lane's 8 sixteen-byte pieces, 1 KiB apart) requested now, waited for later
// (estep_common.h lds_row_request / lds_row_wait).
__device__ __forceinline__ void band_half_request(LdsRow& r, unsigned addr)
{
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                 "ds_read_b128 %3, %4 offset:3072"
                 : "=&v"(r.p[0]), "=&v"(r.p[1]), "=&v"(r.p[2]), "=&v"(r.p[3])
                 : "v"(addr)
                 : "memory");
}

// One double in LDS requested now, waited for later (the compiler sinks a plain read to its use: a round trip on the chain)
__device__ __forceinline__ void lds_f64_request(double& v, const double* lds_ptr)
{
    typedef __attribute__((address_space(3))) const char* lds_cptr;
    const unsigned addr = (unsigned)(uintptr_t)(lds_cptr)lds_ptr;
    asm volatile("ds_read_b64 %0, %1" : "=&v"(v) : "v"(addr) : "memory");
}
__device__ __forceinline__ void lds_f64_wait(double& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory"); }

// One halving level of a reduce-scatter inside the 16-lane rows: lanes of the banks LOW keep a, the others keep b, and
// each adds its partner's value of the same name.  CTRL_LO / CTRL_HI: the DPP controls that fetch the partner for
// the low / high lanes (a bank = four lanes; disabled lanes of an update_dpp keep `old`).
template <int CTRL_LO, int CTRL_HI, int LOW>
__device__ __forceinline__ double band_halve(double a, double b)
{
    constexpr int HIGH = 0xf & ~LOW;
    const int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    int rlo = __builtin_amdgcn_update_dpp(0, alo, CTRL_LO, 0xf, LOW, false);
    int rhi = __builtin_amdgcn_update_dpp(0, ahi, CTRL_LO, 0xf, LOW, false);
    rlo = __builtin_amdgcn_update_dpp(rlo, blo, CTRL_HI, 0xf, HIGH, false);
    rhi = __builtin_amdgcn_update_dpp(rhi, bhi, CTRL_HI, 0xf, HIGH, false);
    const int klo = __builtin_amdgcn_update_dpp(alo, blo, 0xE4, 0xf, HIGH, false);      // quad_perm [0,1,2,3]: the lane's own b
    const int khi = __builtin_amdgcn_update_dpp(ahi, bhi, 0xE4, 0xf, HIGH, false);
    return __hiloint2double(khi, klo) + __hiloint2double(rhi, rlo);
}

template <int W, int RW, int TW>
__global__ __launch_bounds__(kWave* W, 2) void estep_band_kernel(EstepParams p)
{
    using L = BandLds<W, RW, TW>;
    constexpr int NT = kWave * W;
    constexpr int NS = RW + TW;             // word slots = register sets of a lane
    constexpr int NP = NS / 2;              // crossed pairs of sets
    constexpr bool ODD = (NS & 1) != 0;     // a last slot held (and finished) by both lanes of a pair
    constexpr int NF = NP + (ODD ? 1 : 0);  // words a lane finishes
    constexpr int NH = 2 * TW;              // LDS half words per pass
    static_assert(W == 4 || W == 8, "table stride 128 or 256");
    static_assert((RW == 4 || RW == 5) && TW >= 0 && TW <= 2 && (TW == 0 || RW == 5), "word slots");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tband = reinterpret_cast<double*>(smem + L::tband);
    unsigned long long* chg = reinterpret_cast<unsigned long long*>(smem + L::chg);
    double* misc = reinterpret_cast<double*>(smem + L::misc);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    if constexpr (W == 8) {
        if (wave < 4) __builtin_amdgcn_s_setprio(2);      // (estep_quad.h: wavefronts w and w + 4 share a SIMD)
    }
    const int h = lane & 1, g = lane >> 1;
    const int K = p.K, ldk = p.ldk;
    const int doc = p.order[blockIdx.x];
    const int64_t lo = p.doc_ptr[doc];
    const int N = (int)(p.doc_ptr[doc + 1] - lo);
    const int ldk2 = ldk / 2;
    const double2* table = reinterpret_cast<const double2*>(p.expElog) + 16 * wave + 8 * h;   // this lane's 16 topics of a row
    // word slot of register set e (crossed inside the pairs)
    auto slot_of = [&](int e) { return e < 2 * NP ? (e & ~1) + ((e & 1) ^ h) : e; };

    // ---- small loads first: they must not queue behind the tile gather (vmcnt retires in order) ----
    int wid[NS];
#pragma unroll
    for (int e = 0; e < NS; ++e) {
        const int n = slot_of(e) * 32 + g;
        wid[e] = n < N ? p.term_id[lo + n] : -1;
    }
    // the words this lane finishes: those of its even sets (and the odd last one)
    // (counts as integers, converted where they are used: the loop sits at the 256-register limit)
    auto fw = [&](int i) { return slot_of(2 * i) * 32 + g; };
    int cnt[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) cnt[i] = fw(i) < N ? p.term_ct[lo + fw(i)] : 0;
    double local = 0.0;
    for (int n = tid; n < N; n += NT) local += (double)p.term_ct[lo + n];
    double asum = 0.0;
    for (int k = lane; k < K; k += kWave) asum += p.alpha[k];
    // the topic this lane owns in the gamma phase
    const int jtop = (lane >> 2) & 15;
    const int ktid = 32 * wave + 16 * h + jtop;
    const bool primary = (lane & 2) == 0;                  // lanes l and l ^ 2 hold the same topic
    const bool topic_live = ktid < K;
    const double alpha_k = topic_live ? p.alpha[ktid] : 1.0;

    // ---- the tile gather: register sets, then the LDS sets (through registers) ----
    double B[RW][16];
#pragma unroll
    for (int e = 0; e < RW; ++e) {
        if (wid[e] >= 0) {
            const double2* row = table + (size_t)wid[e] * ldk2;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const double2 v2 = row[jj];
                B[e][2 * jj] = v2.x;
                B[e][2 * jj + 1] = v2.y;
            }
        } else {
            // a word slot beyond the document: a row of ones, count 0 (estep_quad.h): its normaliser is sum_k t_k > 0,
            // r = 0 and it adds 0 * 1 to every topic sum - no select per iteration
#pragma unroll
            for (int j = 0; j < 16; ++j) B[e][j] = 1.0;
        }
    }
    double2* myrows = reinterpret_cast<double2*>(smem + L::rows) + (size_t)wave * 8 * 64 + lane;   // + (t * W * 8 + piece) * 64
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        double2 v2[8];
        if (wid[RW + t] >= 0) {
            const double2* row = table + (size_t)wid[RW + t] * ldk2;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) v2[jj] = row[jj];
        } else {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) v2[jj] = double2{1.0, 1.0};
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) myrows[(size_t)(t * W * 8 + jj) * 64] = v2[jj];
    }

    // ---- total token count (:162) and the invariant sum_k gamma_k ----
    local = wave_sum(local);
    asum = wave_sum(asum);
    if (lane == 0) misc[wave] = local;
    if (tid == 0) {
        chg[0] = chg[1] = chg[2] = 0ull;
        chg[3] = 0x7fffffffffffffffull;                    // "the iteration before the first" has not converged
    }
    lds_only_barrier();
    double total = 0.0;
#pragma unroll
    for (int w = 0; w < W; ++w) total += misc[w];
    const double psi_total = uniform_f64(digamma(asum + total));

    // ---- gamma phase state: this lane's topic ----
    double gam = topic_live ? alpha_k + total / K : 1.0;                      // :165 (padding topics never move)
    // (gamma_k is the only per-topic value the loop keeps in registers: the t of the iteration in flight is re-read from
    // the band's LDS copy, and the document's row of tfinal is rewritten every iteration with it - fire and forget - so
    // that the t of the LAST EXECUTED iteration, which the statistics pass and the document terms need, is in memory
    // when the loop ends instead of in two more registers of a loop that sits at the 256-register limit)
    // (alpha_k likewise: re-read every iteration, an L1 hit requested a whole pass ahead.)  One 32-bit byte offset
    // serves the three rows - scalar base + offset addressing, no 64-bit pointers in registers
    const unsigned koff = (unsigned)ktid * 8u;
    char* const tfinal_row = reinterpret_cast<char*>(p.tfinal + (size_t)doc * ldk);
    const char* const alpha_row = reinterpret_cast<const char*>(p.alpha);
    // (the empty asm keeps the compiler from hoisting the 64-bit address - or, below, a count converted to a double -
    // out of the loop and into registers it does not have)
    auto launder = [](unsigned v) { asm volatile("" : "+v"(v)); return v; };
    auto my_tfinal = [&]() { return reinterpret_cast<double*>(tfinal_row + launder(koff)); };
    double* myband = tband + 32 * wave;
    double tq[16];
    // t of the band to the lanes: wavefront-private LDS, in-order DS execution, no barrier (estep_common.h wave_lds_exchange)
    auto spread = [&](double t_of_topic) {
        wave_lds_exchange();                               // (the previous reads of the band stay in front)
        if (primary) myband[16 * h + jtop] = t_of_topic;
        wave_lds_exchange();
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const double2 t2 = reinterpret_cast<const double2*>(myband + 16 * h)[jj];
            tq[2 * jj] = t2.x;
            tq[2 * jj + 1] = t2.y;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(tq[j]));
    };
    spread(topic_live ? exp_digamma_minus(gam, psi_total) : 0.0);

    // partial normalisers in LDS: this lane writes / reads the words of its even sets (slot 2i + h) and the odd last one
    // (the write address is the read address + 8 * wave, added where it is used: registers)
    char* const pr_pair = smem + L::part + (size_t)h * L::p_slot + (size_t)(g * W) * 8;
    char* const pr_last = smem + L::part + (size_t)(NS - 1) * L::p_slot + (size_t)(g * W) * 8;

    int bad = 0;
    double R[NF], Q[NP > 0 ? NP : 1];
    // normalisers of the words this lane finishes from the W partials of buffer `pb`, r = count / normaliser (:182-185)
    // two words at a time (four interleaved reciprocal chains would take 30 more registers than the loop has): the second
    // pair's partials are requested before the first pair's chains run
    // (requested in asm statements, as the LDS rows are: the compiler would sink the reads behind the stop test's
    // branch and pay a second round trip; values may only be used through finish_wait)
    struct Partials { f64x2 d[2][W / 2]; };
    auto finish_read = [&](unsigned pb, int i0, Partials& P) {
        typedef __attribute__((address_space(3))) const char* lds_cptr;
#pragma unroll
        for (int i = i0; i < i0 + 2 && i < NF; ++i) {
            const char* at = (ODD && i == NF - 1 ? pr_last : pr_pair + (size_t)(2 * i) * L::p_slot) + pb;
            const unsigned addr = (unsigned)(uintptr_t)(lds_cptr)at;
            if constexpr (W == 4)
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16"
                             : "=&v"(P.d[i - i0][0]), "=&v"(P.d[i - i0][1]) : "v"(addr) : "memory");
            else
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
                             "ds_read_b128 %3, %4 offset:48"
                             : "=&v"(P.d[i - i0][0]), "=&v"(P.d[i - i0][1]), "=&v"(P.d[i - i0][2]), "=&v"(P.d[i - i0][3]) : "v"(addr) : "memory");
        }
    };
    auto finish_wait = [&](Partials& P) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int x = 0; x < W / 2; ++x) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(P.d[i][x]) : : "memory");
    };
    auto finish_pair = [&](int i0, const Partials& P) {
#pragma unroll
        for (int i = i0; i < i0 + 2 && i < NF; ++i) {
            double s = (P.d[i - i0][0].x + P.d[i - i0][0].y) + (P.d[i - i0][1].x + P.d[i - i0][1].y);
            if constexpr (W == 8) s += (P.d[i - i0][2].x + P.d[i - i0][2].y) + (P.d[i - i0][3].x + P.d[i - i0][3].y);
            if (!(s > 1e-280)) bad = 1;                    // (B, t <= 1: a normaliser cannot overflow; NaN fails the compare)
            R[i] = (double)(int)launder((unsigned)cnt[i]) * rcp_newton(s);
        }
    };
    // part0: the first pair's partials, already requested
    auto finish = [&](unsigned pb, Partials& part0) {
        Partials part1;
        if constexpr (NF > 2) finish_read(pb, 2, part1);
        finish_wait(part0);
        finish_pair(0, part0);
        if constexpr (NF > 2) {
            finish_wait(part1);
            finish_pair(2, part1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NP; ++i) Q[i] = dpp_f64<0xB1>(R[i]);             // the partner's word of the pair
    };
    auto r_of_set = [&](int e) -> double { return e < 2 * NP ? ((e & 1) ? Q[e / 2] : R[e / 2]) : R[NF - 1]; };

    const double thresh_f = p.tol * K * kChangeScale;
    const long long thresh = __double_as_longlong(uniform_f64(__longlong_as_double(
        !(thresh_f >= 0.0) ? -1ll : thresh_f >= 9.2e18 ? 0x7fffffffffffffffll : (long long)thresh_f)));
    int it = 0;
    int left = p.max_iter;
    // LDS half words: half H = 2 t + hh is pieces 4 hh .. 4 hh + 3 of LDS set t.  Pass A walks H = 0 .. NH-1, pass B
    // NH-1 .. 0, so the half in the buffer at the end of a pass is the first one of the next: 2 (NH - 1) requests of
    // four ds_read_b128 per iteration instead of 2 NH.
    LdsRow rowbuf;
    typedef __attribute__((address_space(3))) const char* lds_cptr;
    const unsigned rows_addr = (unsigned)(uintptr_t)(lds_cptr)(smem + L::rows) + (unsigned)(wave * 8 * 64 + lane) * 16u;   // LDS byte offset
    auto request_half = [&](int H) { band_half_request(rowbuf, rows_addr + (unsigned)(H >> 1) * (unsigned)L::row_slot + (unsigned)(H & 1) * 4096u); };
    if constexpr (NH > 0) request_half(0);
    unsigned pb = 0;                                       // byte offset of this iteration's partial buffer
    for (;;) {                                             // :174
        if (left <= 0) break;

        // A. partial normalisers of this band: in-lane dot products over the lane's 16 topics
        double a[NS];
        auto half_partial = [&](auto idx) {
            constexpr int H = decltype(idx)::value;
            if constexpr (H < NH) {
                lds_row_wait(rowbuf);
                double row[8];
                rowbuf.unpack(row);
                const double part = dot8_two_chains(row, *reinterpret_cast<const double(*)[8]>(&tq[8 * (H & 1)]));
                if constexpr ((H & 1) == 0) a[RW + H / 2] = part;
                else a[RW + H / 2] += part;
                if constexpr (H + 1 < NH) request_half(H + 1);
            }
        };
        static_for<4>([&](auto cidx) {
            constexpr int cc = decltype(cidx)::value;
            static_for<4>([&](auto jidx) {
                constexpr int j = 4 * cc + decltype(jidx)::value;
                band_col<j, RW>(a, B, tq[j]);
            });
            // the LDS halves ride between the four column chunks (each requested a chunk ahead)
            if constexpr (NH == 4) half_partial(StaticIndex<cc>());
            if constexpr (NH == 2 && (cc & 1)) half_partial(StaticIndex<cc / 2>());
        });
        // one add across the pair per PAIR of slots; a DPP read needs two wait states behind the VALU write of its
        // source and the compiler's hazard recogniser does not look inside the asm blocks that produced a[]
#pragma unroll
        for (int e = 1; e < NS; e += 2) asm volatile("s_nop 1" : "+v"(a[e]));
        if constexpr (ODD) asm volatile("s_nop 1" : "+v"(a[NS - 1]));
#pragma unroll
        for (int i = 0; i < NP; ++i) reinterpret_cast<double*>(pr_pair + (size_t)(2 * i) * L::p_slot + pb)[wave] = a[2 * i] + dpp_f64<0xB1>(a[2 * i + 1]);
        if constexpr (ODD) reinterpret_cast<double*>(pr_last + pb)[wave] = a[NS - 1] + dpp_f64<0xB1>(a[NS - 1]);
        lds_only_barrier();                                // THE barrier of the iteration (LDS traffic only: no global access is in flight)
        // stop test of the iteration before (:189, mean <= tol): an integer compare on its fixed-point sum; every
        // wavefront reads the same complete value (all its atomics precede this barrier, the next reset follows the next one)
        // (the partials are requested ahead of it: one LDS round trip for both)
        Partials part0;
        finish_read(pb, 0, part0);
        const long long moved = (long long)chg[(it + 3) & 3];
        if (moved <= thresh) break;
        // alpha_k of this lane's topic (1 beyond K: padding topics never move, their t is 0)
        double alpha_it = 1.0;
        if (topic_live) alpha_it = *reinterpret_cast<const double*>(alpha_row + launder(koff));
        // both coefficient tables of exp_digamma_minus_levels, requested here: the scalar-cache round trips ride on pass B
        ExpDigammaLevelsA coef_a;
        ExpDigammaLevelsB coef_b;
        coef_a.load();
        coef_b.load();
        finish(pb, part0);

        // B. q[k] over this lane's words (plain FMAs, sixteen accumulators), LDS halves in reverse order
        double q[16];
        auto half_topic_sums = [&](auto idx) {
            constexpr int H = decltype(idx)::value;
            if constexpr (H >= 0 && H < NH) {
                lds_row_wait(rowbuf);
                double row[8];
                rowbuf.unpack(row);
                band_row8<false>(&q[8 * (H & 1)], r_of_set(RW + H / 2), row);
                if constexpr (H > 0) request_half(H - 1);
            }
        };
        {
            const double r0 = r_of_set(0);
            band_row8<true>(&q[0], r0, &B[0][0]);
            band_row8<true>(&q[8], r0, &B[0][8]);
        }
        half_topic_sums(StaticIndex<NH - 1>());
        static_for<RW - 1>([&](auto idx) {
            constexpr int e = decltype(idx)::value + 1;
            const double re = r_of_set(e);
            band_row8<false>(&q[0], re, &B[e][0]);
            band_row8<false>(&q[8], re, &B[e][8]);
            if constexpr (NH == 4 && e < 4) half_topic_sums(StaticIndex<NH - 1 - e>());
            if constexpr (NH == 2 && e == 2) half_topic_sums(StaticIndex<0>());
        });

        double t_mine;                                     // this lane's topic: t of the iteration in flight, from the band's LDS copy
        lds_f64_request(t_mine, myband + 16 * h + jtop);
        // over the 32 word groups of the wavefront: two swap levels across the rows, two halving levels and one
        // all-reduce level inside them; lane l ends up with topic (l >> 2) & 15 of its half
        double u[8], v[4], x[2];
#pragma unroll
        for (int m = 0; m < 8; ++m) u[m] = swap32_add(q[m], q[m + 8]);
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = swap16_add(u[m], u[m + 4]);
#pragma unroll
        for (int m = 0; m < 2; ++m) x[m] = band_halve<0x128, 0x128, 0x3>(v[m], v[m + 2]);       // row_ror:8, lanes 0-7 | 8-15
        double qk = band_halve<0x104, 0x114, 0x5>(x[0], x[1]);                                   // row_shl:4 | row_shr:4, lanes ^ 4
        qk += dpp_f64<0x4E>(qk);                                                                 // quad_perm [2,3,0,1], lanes ^ 2

        // C. gamma update (:185-188) and the next t, every lane for its topic
        lds_f64_wait(t_mine);
        const double gnew = fma(t_mine, qk, alpha_it);
        const double diff = fabs(gnew - gam);
        gam = gnew;
        if (primary) atomicAdd(&chg[it & 3], change_fixed(diff));
        if (tid == 0) chg[(it + 2) & 3] = 0ull;            // read last behind the barrier before this one, added to behind the one after next
        if (primary) *my_tfinal() = t_mine;                // (behind the use of alpha_it: the wait for that load must not cover this store)
        const double t_next = exp_digamma_minus_levels<true>(gam, psi_total, coef_a, &coef_b);
        spread(topic_live ? t_next : 0.0);
        ++it;
        --left;
        pb = (unsigned)L::p_buf - pb;
    }
    // R / Q of the last EXECUTED iteration: its partials are still in the other buffer (the one just written, if any,
    // belongs to the half iteration behind the last update)
    if constexpr (NH > 0) lds_row_wait(rowbuf);            // no read may land after the loop
    {
        Partials part0;
        finish_read((unsigned)L::p_buf - pb, 0, part0);
        finish((unsigned)L::p_buf - pb, part0);
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

    // ---- training fast path: the document terms are left to doc_terms_kernel (doc_terms.h) ----
    if (!p.heldout && !p.want_doc_ll) {
#pragma unroll
        for (int i = 0; i < NF; ++i)       // (every wavefront finishes every word: wavefront i % W writes word i)
            if (wave == i % W && fw(i) < N && (h == 0 || !(ODD && i == NF - 1))) p.rfinal[lo + fw(i)] = R[i];
        if (primary && topic_live) p.gamma[(size_t)doc * K + ktid] = gam;      // (tfinal: written by the loop)
        if (tid == 0) {
            p.iters[doc] = it;
            p.status[doc] = 3;
        }
        return;
    }

    // ---- document terms (:195-204) with the last phi = B t r (see estep_slab.h) ----
    const double t_prev = primary ? *my_tfinal() : 0.0;      // this thread's own store of the last executed iteration
    spread(t_prev);
    double term1 = 0.0;
    {
        const double2* gtable = reinterpret_cast<const double2*>(p.expElog_elog) + 16 * wave + 8 * h;
#pragma unroll
        for (int e = 0; e < NS; ++e) {
            const int n = slot_of(e) * 32 + g;
            if (n < N) {
                const double2* row = gtable + (size_t)p.term_id[lo + n] * ldk2;
                double gs = 0.0;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const double2 g2 = row[jj];
                    gs = fma(g2.y, tq[2 * jj + 1], fma(g2.x, tq[2 * jj], gs));
                }
                term1 = fma(r_of_set(e), gs, term1);
            }
        }
    }
    // c_n log(normaliser_n) from r_n = c_n / normaliser_n (the normalisers themselves were not kept)
    double term3 = 0.0, shift_term = 0.0;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const bool owner = wave == i % W && fw(i) < N && (h == 0 || !(ODD && i == NF - 1));   // one of the W wavefronts that hold R[i]
        if (owner) {
            const double c = (double)cnt[i];
            term3 = fma(c, log(c) - log(R[i]), term3);
            if (p.heldout) shift_term = fma(c, p.shift[p.term_id[lo + fw(i)]], shift_term);
            else p.rfinal[lo + fw(i)] = R[i];
        }
    }
    double term2 = 0.0, lse_term = 0.0, lgam = 0.0, gsum = 0.0;
    if (primary) {
        if (topic_live) {
            const double mass = gam - p.alpha[ktid];                          // = t_last * q
            if (mass != 0.0) term2 = log(t_prev) * mass;                      // (t_k may have underflowed where the mass did)
            if (p.heldout) lse_term = p.topic_lse[ktid] * mass;
            lgam = lgamma_pos(gam);
            gsum = gam;
            p.gamma[(size_t)doc * K + ktid] = gam;
        }
    }
    term1 = wave_sum(term1);
    term2 = wave_sum(term2);
    lse_term = wave_sum(lse_term);
    lgam = wave_sum(lgam);
    gsum = wave_sum(gsum);
    term3 = wave_sum(term3);
    shift_term = wave_sum(shift_term);
    if (lane == 0) {                                       // (the barrier of the `bad` vote separates this from the prologue's use of misc)
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
