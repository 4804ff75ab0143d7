// Register + LDS tile E-step kernel on the quilt lane grid, 16 word groups per document:
//
//   TL = 16 (64 < K <= 128):  FOUR wavefronts per document, 4 x 16 lanes each, TWO documents per CU
//   TL = 32 (128 < K <= 256): EIGHT wavefronts per document, 2 x 32 lanes each, one document per CU
//
// Why four wavefronts at K <= 128.  The 8-wavefront quilt kernel (estep_quilt.h) fills a CU's
// register file with ONE document, and an inner iteration is a serial chain (normalisers -> r ->
// topic sums -> cross-wavefront reduction -> gamma / digamma / exp -> t), so its LDS round trips,
// its two barriers and the gamma phase leave the fp64 pipes idle more than half of the time
// (round-1 counters: 44 % of wave-cycles waiting).  Nothing of the same document can fill those
// gaps; another document can.  A document here is a 256-thread workgroup whose wavefronts each sit
// on a different SIMD, at <= 256 VGPRs, so two documents are co-resident per CU and every SIMD
// alternates between them (measured: 205 ns per document with two per CU, 338 ns with one).
// At K = 256 the tile (N_d x 2 KiB = 400 KiB) leaves room for one document per CU; the same body
// runs with eight wavefronts, and - unlike the tiered kernels of rounds 1-2 - keeps EVERY word
// of a document of up to 224 terms on chip: nothing is re-read from L2 inside the loop.
//
// Layout.  lane = TL*g + c: word group g, topic lane c; a lane holds 8 values of a table row:
// topics 2c + 2*TL*jj + {0,1}, jj < 4 (16-byte pieces, TL*16 bytes apart).  A document has 16 word
// groups gg = (64/TL)*wave + g and word n belongs to group n % 16, slot n / 16.  Slots 0 .. RWL-1
// of a group live in VGPRs (RWL = 10: 160 words, 160 VGPRs), slots RWL .. RWL+TWL-1 as whole rows
// in LDS, read twice per iteration (normaliser pass, topic-sum pass) as conflict-free ds_read_b128
// (row stride = 0 mod 256 B: the 16 lanes an instruction services together read 16 different
// 16-byte slots).  Per-thread loop constants (alpha, previous gamma, word counts) also live in LDS.
//
// Normalisers: partial sums over a lane's 8 topics go through an LDS transpose of 8 rows per word
// group and TL/8 lanes finish each word (DPP inside the 16-lane row, one permlane16 swap between
// the two rows of a 32-lane group) - done twice per iteration (slots 0-7, then slots 8 ..
// RWL+TWL-1) through the same 5 KiB-per-wavefront buffer (the LDS executes a wavefront's DS
// instructions in order, so the second set of writes may be issued right behind the first set of
// reads).  r reaches the lanes holding a word's tile entries as the DPP row-broadcast operand of
// the FMA itself (estep_common.h row_bcast_fmac).
//
// The loop body is ordered by hand (estep_common.h, "hand-ordered FMA blocks"): FMA blocks of eight
// independent chains, and every LDS row requested at least 16 FMAs before it is used, through ONE
// 16-VGPR row buffer:   [32 FMAs] row 0 [32 FMAs] row 1 [16 FMAs] row 2   in both passes.
#pragma once
#include "estep_common.h"
#include "special_device.h"
#include "estep_epilogue.h"

namespace pylda {

template <int TL, int RWL, int TWL>
struct QuadLds {
    static constexpr int W = TL / 4;                                               // wavefronts per document
    static constexpr int G = kWave / TL;                                           // word groups per wavefront
    static constexpr int kTopics = 8 * TL;
    // kPre (four LDS slots): neighbouring lanes add their partial normalisers before the transpose (one
    // DPP level, 3 instructions per word) so that it takes half the LDS - which, with the word counts
    // re-read from global memory, is what lets 64 rows fit beside it (stride 256: 128 KiB in one
    // workgroup's 160 KiB; stride 128: 64 KiB in each of two workgroups' 80 KiB).
    static constexpr bool kPre = TWL >= 4;
    static constexpr int kPartials = kPre ? TL / 2 : TL;                           // partial sums per word in the transpose
    static constexpr int kRedStride = kPartials == 8 ? 10 : kPartials == 16 ? 20 : 40;   // 16-byte aligned rows, b128 read-back (QuiltLds / QwideLds)
    static constexpr size_t red_wave = (size_t)G * 8 * kRedStride * 8;             // 5120 B (kPre: 2560 B)
    static_assert(red_wave >= (size_t)kTopics * 8, "a wavefront's topic partials fit in its transpose area");
    static constexpr size_t red = 0;                                               // [W][G][8][kRedStride]; reused for the W x kTopics partial sums
    static constexpr size_t tt = red + (size_t)W * red_wave;                       // [2][kTopics]
    static constexpr size_t chg = tt + (size_t)2 * kTopics * 8;                    // u64[2]
    static constexpr size_t livec = chg + 16;                                      // u32[2] (+ 8 bytes of padding): topics with gamma_k != alpha_k
    static constexpr size_t misc = livec + 16;                                     // [8][W]
    static constexpr size_t alf = misc + (size_t)8 * W * 8;                        // [kTopics] alpha (1 beyond K)
    static constexpr size_t gpv = alf + (size_t)kTopics * 8;                       // [kTopics] gamma before the last update
    static constexpr size_t cnt = gpv + (size_t)kTopics * 8;                       // double [2][W * 64] counts of the words a lane finishes
    static constexpr bool kGlobalCounts = TWL >= 4;
    static constexpr size_t rows = (cnt + (kGlobalCounts ? 0 : (size_t)2 * W * 64 * 8) + 255) & ~(size_t)255;   // [16][TWL][kTopics]
    static constexpr size_t total = rows + (size_t)16 * TWL * kTopics * 8;
    static_assert(TL != 16 || 2 * total <= 160 * 1024, "K <= 128: two workgroups per CU");
    static_assert(total <= 160 * 1024, "fits the LDS");
};

// Which streamed slot a pass handles at its stop P (0: start ... 3: end), or -1: the SWL slots are spread over the pass.
constexpr int quad_stream_stop(int swl, int stop)
{
    return swl == 4 ? stop : swl == 3 ? (stop == 0 ? 0 : stop == 2 ? 1 : stop == 3 ? 2 : -1)
         : swl == 2 ? (stop == 0 ? 0 : stop == 3 ? 1 : -1) : swl == 1 ? (stop == 0 ? 0 : -1) : -1;
}

// Hand-over of a document to the live-topic kernel (estep_compact.h), which runs the remaining iterations on the
// document's N x L tile - one wavefront, eight documents per CU.  What it needs: gamma after `it` updates (a dead topic
// stays at alpha_k), the live topics in ascending order, and their columns of the tile, term-minor (its lanes own terms).
// Everything lane-shaped is formed again from the thread index: nothing of the prologue is kept alive for this exit.
template <int TL, int RWL, int TWL, int SWL>
__device__ __forceinline__ void quad_hand_over(const EstepParams& p, char* smem, const double (&B)[RWL][8], double gam, int doc, int64_t lo, int N,
                                               int it)
{
    using L = QuadLds<TL, RWL, TWL>;
    constexpr int G = L::G, KT = 8 * TL, KRL = 8, WPR = RWL + TWL, WPG = WPR + SWL;
    const int K = p.K;
    double* tt = reinterpret_cast<double*>(smem + L::tt);
    double* misc = reinterpret_cast<double*>(smem + L::misc);
    const double* alf = reinterpret_cast<const double*>(smem + L::alf);
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int c = lane % TL, gg = wave * G + lane / TL;
    const int trank = wave < KT / kWave ? wave : -1;
    const int ktid = trank >= 0 ? trank * kWave + lane : 0;
    int* pos = reinterpret_cast<int*>(tt);                                // [KT]: column of topic k in the compact tile, or -1
    unsigned* wcount = reinterpret_cast<unsigned*>(misc);                 // live topics per topic wavefront
    const bool alive = trank >= 0 && ktid < K && gam != alf[ktid];
    const unsigned long long mask = __ballot(alive);
    if (trank >= 0 && lane == 0) wcount[trank] = (unsigned)__builtin_popcountll(mask);
    __syncthreads();                                                      // (every read of tt[] by the loop is behind us: __syncthreads_or)
    if (trank >= 0) {
        int at = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        for (int w = 0; w < trank; ++w) at += (int)wcount[w];
        pos[ktid] = alive ? at : -1;
        if (alive) *live_idx_at(live_list_of(p.live_list, doc), at) = (uint16_t)ktid;
        if (ktid < K) p.gamma[(size_t)doc * K + ktid] = gam;
    }
    __syncthreads();
    // uniform base + 32-bit byte offset: one address register per store instead of a pointer pair (a document's tile is
    // at most 256 x 32 x 8 bytes)
    double* tile = p.live_tile + p.tile_ptr[doc];
    {
        unsigned long long bits = reinterpret_cast<unsigned long long>(tile);
        bits = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bits >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)bits);
        tile = reinterpret_cast<double*>(bits);
    }
    const double2* rows = reinterpret_cast<const double2*>(smem + L::rows) + (size_t)gg * TWL * (KT / 2) + c;
    // topic by topic of this lane's eight (2 c + 2 TL jj + {0, 1}): its column, then the lane's terms
#ifndef PYLDA_EXPERIMENT_NO_TILE      // (timing probe: what the tile stores of the hand-over cost - the results are wrong without them)
    static_for<KRL>([&](auto idx) {
        constexpr int j = decltype(idx)::value;
        const int at = pos[2 * (c + TL * (j / 2)) + (j & 1)];
        if (at >= 0) {
            const unsigned base = (unsigned)(at * N) * 8u;
#pragma unroll
            for (int s = 0; s < WPG; ++s) {
                const int n = s * 16 + (s < WPR ? gg : 15 - gg);
                if (n < N) {
                    double v;
                    if (s < RWL) v = B[s < RWL ? s : 0][j];
                    else if (s < WPR) v = reinterpret_cast<const double*>(rows + (s - RWL) * (KT / 2) + TL * (j / 2))[j & 1];
                    else v = p.expElog[(size_t)p.term_id[lo + n] * p.ldk + 2 * (c + TL * (j / 2)) + (j & 1)];
                    store_f64_uniform_base(tile, base + (unsigned)n * 8u, v);
                }
            }
        }
    });
#endif
    if (tid == 0) {
        unsigned total_live = 0;
        for (int w = 0; w < KT / kWave; ++w) total_live += wcount[w];
        p.live_n[doc] = (int)total_live;
        p.handoff_it[doc] = it;
        p.iters[doc] = it;
        p.status[doc] = 4;
    }
}

// SWL > 0: word slots beyond the register and LDS capacity - documents of 225-256 terms at stride 256, the 3 % of cfg 4
// that the two-pass tiered kernel (estep_qwide.h, retired) ran at 950 ns per document.  A streamed slot is a row of
// the table (an L2 hit: the same CU read it an iteration earlier) read through ONE more 16-VGPR buffer per wavefront
// and handled exactly like an LDS slot (partial normaliser into the second transpose, r from the row broadcast) - and,
// like the LDS slots, walked forwards by the normaliser pass and backwards by the topic-sum pass, so the row one pass
// ends on is the row the next pass starts with: 2 (SWL - 1) table rows per iteration.  The register slots go down to
// eight to make room (with nine the tile itself spills).  The streamed slots deal their words to the word groups in
// REVERSE order: a document that fills its last slot only half (three quarters of the class) leaves the empty half to
// the topic wavefronts - the critical path - which skip the slot, requests included.  Measured, cfg 4 documents of
// 225-240 terms alone on the chip: 667 ns per document against 944.  What is left is exposed latency: the FMA work
// of a whole pass is ~0.3 us, less than an L2 round trip, and the one long window of an iteration - the gamma phase -
// can hide rows only if they stay in registers across it (tried: two or three rows in flight across the gamma phase,
// consumed at the top of the next iteration in the fused form of estep_qfuse.h; the kernel then spills 150-280 bytes
// per lane and every reload's vmcnt(0) waits for the rows in flight: 1300 ns per document).
// HANDOFF false: the kernel of the corpora that hand nothing over (option compact = 0; alpha grown past the mortality bound,
// host_internal.h alpha_allows_live; hand-over buffers that did not fit) - no live count, no exit from the loop: the
// loop of round 5, whose register allocation the counting perturbs (cfg 4, dense kernels only: 447 -> 4xx ms per E-step).
template <int TL, int RWL, int TWL, int SWL = 0, bool HANDOFF = true>
__global__ __launch_bounds__(kWave*(TL / 4), 2) void estep_quad_kernel(EstepParams p)
{
    using L = QuadLds<TL, RWL, TWL>;
    constexpr int W = L::W, G = L::G;
    constexpr int NT = kWave * W;
    constexpr int KT = 8 * TL;              // padded topic count (== ldk)
    constexpr int KRL = 8;                  // values of a row per lane
    constexpr int WPG = RWL + TWL + SWL;    // word slots per group
    constexpr int C0 = WPG < 8 ? WPG : 8;   // slots finished in the first transpose
    constexpr int C1 = WPG - C0;            // ... in the second
    constexpr int R1 = RWL > 8 ? RWL - 8 : 0;   // register slots of the second chunk
    constexpr int RS = L::kRedStride;
    constexpr int FL = TL / 8;              // lanes that finish one normaliser (2 or 4)
    constexpr bool PRE = L::kPre;
    constexpr int NPIECE = L::kPartials / 2 / FL;   // 16-byte pieces of the transpose each of them reads (4; kPre: 2)
    constexpr int QV = KRL / G;             // topic values per lane after the in-wavefront reduction (2 or 4)
    static_assert(TL == 16 || TL == 32, "ldk 128 or 256");
    constexpr int WPR = RWL + TWL;          // ... of them on chip
    static_assert(RWL >= 2 && RWL <= 10 && TWL >= 0 && TWL <= 4 && SWL >= 0 && SWL <= 4 && WPG <= 16, "word slots per group");
    static_assert(SWL == 0 || TWL > 0, "streamed slots come behind the LDS slots of the second chunk");
    static_assert(TWL == 0 || RWL >= 8, "LDS slots belong to the second chunk");
    static_assert(C0 == 8 || TWL == 0, "LDS slots need the eight-slot first chunk");
    static_assert(KT <= NT, "one thread per topic in the gamma phase");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem + L::red);
    double* tt = reinterpret_cast<double*>(smem + L::tt);
    unsigned long long* chg = reinterpret_cast<unsigned long long*>(smem + L::chg);
    unsigned* livec = reinterpret_cast<unsigned*>(smem + L::livec);
    double* misc = reinterpret_cast<double*>(smem + L::misc);
    double* alf = reinterpret_cast<double*>(smem + L::alf);
    double* gpv = reinterpret_cast<double*>(smem + L::gpv);
    double* cntv = reinterpret_cast<double*>(smem + L::cnt);    // (as doubles: no int -> fp64 conversion per iteration)

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    // Stride 256: wavefronts w and w + 4 of the document share a SIMD and, being in the same phase, would
    // fight for issue slots in the FMA bursts and then wait for the LDS together.  With the first four at a
    // higher priority a SIMD runs one wavefront's burst at full rate while the other one's LDS round trips
    // are in flight (measured on the 193-208-term class of cfg 4: 466 -> 440 ns per document; priority 1, 2
    // or 3, or raising the other four instead, all within 1.5 % of each other).  At stride 128 the two
    // wavefronts of a SIMD belong to different documents, already out of phase: no gain there.
    if constexpr (TL == 32) {
        if (wave < 4) __builtin_amdgcn_s_setprio(2);
    }
    const int g = lane / TL, c = lane % TL;
    const int cl = lane & 15;               // position inside the 16-lane row
    const int half = (lane >> 4) & (TL / 16 - 1);   // row of a 32-lane group
    const int gg = wave * G + g;            // word group of this lane: words gg, gg + 16, gg + 32, ...
    const int K = p.K, ldk = p.ldk;
    const int doc = p.order[blockIdx.x];
    const int64_t lo = p.doc_ptr[doc];
    const int N = (int)(p.doc_ptr[doc + 1] - lo);
    const double2* table = reinterpret_cast<const double2*>(p.expElog);
    const int ldk2 = ldk / 2;
    // this lane group's rows in LDS: [TWL][KT] doubles, the lane reads 16-byte pieces c + TL*jj
    double2* myrows = reinterpret_cast<double2*>(smem + L::rows) + (size_t)gg * TWL * (KT / 2) + c;

    // ---- small loads first: they must not queue behind the tile gather (vmcnt retires in order) ----
    int wid[WPG];
#pragma unroll
    for (int s = 0; s < WPG; ++s) {
        const int n = s * 16 + (s < WPR ? gg : 15 - gg);      // (streamed slots: groups in reverse order, see above)
        wid[s] = n < N ? p.term_id[lo + n] : -1;
    }
    // the words whose normalisers this lane finishes (one of FL lanes): slots cl/2 and 8 + cl/2 of its group
    const int slot0 = cl >> 1, slot1 = 8 + (cl >> 1);
    const int part = (cl & 1) + 2 * half;
    const int word0 = slot0 * 16 + gg, word1 = slot1 * 16 + (slot1 < WPR ? gg : 15 - gg);
    const bool live0 = slot0 < C0 && word0 < N;
    const bool live1 = C1 > 0 && slot1 < WPG && word1 < N;
    // lanes whose slot exists in this launch class (the transpose rows of the others are never written)
    const bool exists0 = slot0 < C0, exists1 = C1 > 0 && slot1 < WPG;
    constexpr bool GCNT = L::kGlobalCounts;
    if constexpr (!GCNT) {
        cntv[tid] = live0 ? (double)p.term_ct[lo + word0] : 0.0;
        cntv[NT + tid] = live1 ? (double)p.term_ct[lo + word1] : 0.0;
    }
    // kPre: no LDS left for the counts - re-read from global memory every iteration (an L1 / L2 hit requested
    // a whole pass before it is used); the empty asm keeps the compiler from hoisting the load into a VGPR
    auto count_of = [&](int which) -> double {
        if constexpr (GCNT) {
            unsigned at = which ? (live1 ? word1 : 0) : (live0 ? word0 : 0);   // (uniform base + 32-bit index: one VGPR)
            asm volatile("" : "+v"(at));
            const int ct = (p.term_ct + lo)[at];
            return (which ? live1 : live0) ? (double)ct : 0.0;
        } else {
            return cntv[which * NT + tid];
        }
    };
    double local = 0.0;
    for (int n = tid; n < N; n += NT) local += (double)p.term_ct[lo + n];
    double asum = 0.0;
    for (int k = lane; k < K; k += kWave) asum += p.alpha[k];
    if (tid < KT) alf[tid] = tid < K ? p.alpha_sgn[tid] : 1.0;      // (alpha; sign bit: the topic never counts as dead, kMortalT)

    // ---- the tile gather: register slots, then the LDS slots (through registers) ----
    double B[RWL][KRL];
#pragma unroll
    for (int i = 0; i < RWL; ++i) {
        if (wid[i] >= 0) {
            const double2* row = table + (size_t)wid[i] * ldk2 + c;
#pragma unroll
            for (int jj = 0; jj < KRL / 2; ++jj) {
                const double2 v2 = row[TL * jj];
                B[i][2 * jj] = v2.x;
                B[i][2 * jj + 1] = v2.y;
            }
        } else {
            // a word slot beyond the document: a row of ones, count 0.  Its normaliser is sum_k t_k > 0 (finite
            // reciprocal), r = 0 * that = 0 and it adds 0 * 1 to every topic sum - without a select per
            // iteration on r (a zero row would give 0 * (1 / 0))
#pragma unroll
            for (int j = 0; j < KRL; ++j) B[i][j] = 1.0;
        }
    }
#pragma unroll
    for (int t = 0; t < TWL; ++t) {
        double2 v2[KRL / 2];
        if (wid[RWL + t] >= 0) {
            const double2* row = table + (size_t)wid[RWL + t] * ldk2 + c;
#pragma unroll
            for (int jj = 0; jj < KRL / 2; ++jj) v2[jj] = row[TL * jj];
        } else {
#pragma unroll
            for (int jj = 0; jj < KRL / 2; ++jj) v2[jj] = double2{1.0, 1.0};
        }
#pragma unroll
        for (int jj = 0; jj < KRL / 2; ++jj) myrows[t * (KT / 2) + TL * jj] = v2[jj];
    }

    // streamed slots: byte offset of this lane's piece of the row (the table is below 4 GiB: plan.hip); a slot beyond
    // the document reads row 0 and its partial normaliser is replaced; swave: any live word in this WAVEFRONT (uniform)
    unsigned srow[SWL > 0 ? SWL : 1];
    bool slive[SWL > 0 ? SWL : 1], swave[SWL > 0 ? SWL : 1];
#pragma unroll
    for (int s = 0; s < SWL; ++s) {
        slive[s] = wid[WPR + s] >= 0;
        swave[s] = (WPR + s) * 16 + 15 - (wave * G + G - 1) < N;
        srow[s] = ((unsigned)(slive[s] ? wid[WPR + s] : 0) * (unsigned)ldk2 + (unsigned)c) * 16u;
    }

    // ---- total token count (:162) and the invariant sum_k gamma_k ----
    local = wave_sum(local);
    asum = wave_sum(asum);
    if (lane == 0) misc[wave] = local;
    if (tid == 0) {
        chg[0] = chg[1] = 0ull;
        livec[0] = livec[1] = 0u;
    }
    lds_only_barrier();
    // the gamma phase: one thread per topic on the first KT / 64 wavefronts.  (Measured and not adopted: taking
    // the two documents' gamma wavefronts on disjoint SIMD pairs - HW_ID / LDS_ALLOC tell a workgroup where it
    // sits - and s_setprio around the phase: both within noise, the phase is bound by its dependent chain.)
    const int trank = wave < KT / kWave ? wave : -1;
    const bool topic_thread = trank >= 0;
    const int ktid = topic_thread ? trank * kWave + lane : 0;      // the topic this thread owns in the gamma phase
    const bool topic_live = topic_thread && ktid < K;
    double total = 0.0;
#pragma unroll
    for (int w = 0; w < W; ++w) total += misc[w];
    const double psi_total = uniform_f64(digamma(asum + total));

    // ---- gamma phase state: thread k < KT owns topic k ----
    double gam = 1.0;
    if (topic_thread) {
        gam = topic_live ? fabs(alf[ktid]) + total / K : 1.0;             // :165 (padding topics never move)
        tt[ktid] = topic_live ? exp_digamma_minus(gam, psi_total) : 0.0;
    }
    lds_only_barrier();

    double r0 = 0.0, r1 = 0.0;
    int it = 0;
    int bad = 0;
    double* myred = red + (size_t)wave * (L::red_wave / 8) + (size_t)g * 8 * RS;  // this lane group's 8 rows
    const double2* mysrc = reinterpret_cast<const double2*>(myred + (cl >> 1) * RS) + part;
    const int cw = PRE ? c >> 1 : c;        // where this lane's partial goes in a transpose row
    // this lane's share of its word's partials (NPIECE pieces of 16 bytes), then over the word's FL lanes
    auto finish_sum = [&](const double2 (&h)[NPIECE]) {
        double s;
        if constexpr (NPIECE == 4) s = ((h[0].x + h[1].x) + (h[2].x + h[3].x)) + ((h[0].y + h[1].y) + (h[2].y + h[3].y));
        else s = (h[0].x + h[1].x) + (h[0].y + h[1].y);
        s = lane_group_sum<2>(s);                          // the word's other lane of this 16-lane row
        if constexpr (TL == 32) s = swap16_add(s, s);      // ... and the two lanes of the group's other row
        return s;
    };
    // stop test: integer compare on the fixed-point sum, evaluated behind the first half of the next
    // iteration (see estep_quilt.h)
    const double thresh_f = p.tol * K * kChangeScale;
    const long long thresh = __double_as_longlong(uniform_f64(__longlong_as_double(
        !(thresh_f >= 0.0) ? -1ll : thresh_f >= 9.2e18 ? 0x7fffffffffffffffll : (long long)thresh_f)));
    long long moved = 0x7fffffffffffffffll;
    int left = p.max_iter;
    // topics whose gamma differs from alpha (bitwise) after the last update, and the count at which the document
    // leaves this kernel (-1: never)
    int nlive = KT;
    const int handoff_at = p.handoff_live > 0 ? p.handoff_live : -1;      // (one scalar: these kernels have no register to spare for a look-up)
    double tq[KRL];
#pragma unroll
    for (int jj = 0; jj < KRL / 2; ++jj) {
        const double2 t2 = reinterpret_cast<const double2*>(tt)[c + TL * jj];
        tq[2 * jj] = t2.x;
        tq[2 * jj + 1] = t2.y;
    }
#pragma unroll
    for (int j = 0; j < KRL; ++j) asm volatile("" : "+v"(tq[j]));
    LdsRow rowbuf;
    auto request_row = [&](int t) { lds_row_request<TL * 16>(rowbuf, myrows + t * (KT / 2)); };
    if constexpr (TWL > 0) request_row(0);
    LdsRow sbuf;
    auto request_srow = [&](int s) { table_row_request<TL * 16>(sbuf, p.expElog, srow[s]); };
    if constexpr (SWL > 0) request_srow(0);                               // (slot 0 is never empty: N > 16 * WPR)
    for (;;) {                                                            // :174
        const int buf = it & 1;

        // A. partial normalisers over this lane's topics -> LDS transpose -> sum over the TL topic lanes
        double a[8];
        double pr[TWL > 0 ? TWL : 1];
        double ps[SWL > 0 ? SWL : 1];
        auto stream_partial = [&](auto stop) {               // streamed slot s: partial normaliser, next row requested
            constexpr int s = quad_stream_stop(SWL, decltype(stop)::value);
            if constexpr (s >= 0) {
                ps[s] = 1.0;                                 // (an empty slot: any positive normaliser, its count is 0)
                if (swave[s]) {
                    table_row_wait(sbuf);
                    double row[8];
                    sbuf.unpack(row);
                    const double d = TL == 16 ? dot8_two_chains(row, tq) : dot8(row, tq);
                    if (slive[s]) ps[s] = d;
                    if constexpr (s + 1 < SWL) {
                        if (swave[s + 1]) request_srow(s + 1);
                    }
                }
            }
        };
        auto row_partial = [&](auto idx) {                   // LDS slot t: partial normaliser, next row requested
            constexpr int t = decltype(idx)::value;
            if constexpr (t < TWL) {
                lds_row_wait(rowbuf);
                double row[8];
                rowbuf.unpack(row);
                pr[t] = TL == 16 ? dot8_two_chains(row, tq) : dot8(row, tq);
                if constexpr (t + 1 < TWL) request_row(t + 1);
            }
        };
        stream_partial(StaticIndex<0>());
        if constexpr (C0 == 8) {
            col_mul8(a, B[0][0], B[1][0], B[2][0], B[3][0], B[4][0], B[5][0], B[6][0], B[7][0], tq[0]);
            col_fmac8(a, B[0][1], B[1][1], B[2][1], B[3][1], B[4][1], B[5][1], B[6][1], B[7][1], tq[1]);
            col_fmac8(a, B[0][2], B[1][2], B[2][2], B[3][2], B[4][2], B[5][2], B[6][2], B[7][2], tq[2]);
            col_fmac8(a, B[0][3], B[1][3], B[2][3], B[3][3], B[4][3], B[5][3], B[6][3], B[7][3], tq[3]);
            row_partial(StaticIndex<0>());
            stream_partial(StaticIndex<1>());
            col_fmac8(a, B[0][4], B[1][4], B[2][4], B[3][4], B[4][4], B[5][4], B[6][4], B[7][4], tq[4]);
            col_fmac8(a, B[0][5], B[1][5], B[2][5], B[3][5], B[4][5], B[5][5], B[6][5], B[7][5], tq[5]);
            col_fmac8(a, B[0][6], B[1][6], B[2][6], B[3][6], B[4][6], B[5][6], B[6][6], B[7][6], tq[6]);
            col_fmac8(a, B[0][7], B[1][7], B[2][7], B[3][7], B[4][7], B[5][7], B[6][7], B[7][7], tq[7]);
            row_partial(StaticIndex<1>());
            stream_partial(StaticIndex<2>());
        } else {
#pragma unroll
            for (int i = 0; i < C0; ++i) a[i] = dot8(B[i], tq);
        }
        if constexpr (PRE) {
            // a DPP read needs two wait states behind the VALU write of its source, and the compiler's hazard
            // recogniser does not look inside the asm blocks that produced a[]: the last two chains settle here
            asm volatile("s_nop 1" : "+v"(a[C0 - 2]), "+v"(a[C0 - 1]));
        }
#pragma unroll
        for (int i = 0; i < C0; ++i) myred[i * RS + cw] = PRE ? lane_group_sum<2>(a[i]) : a[i];
        // ... or the document is handed to the live-topic kernel (estep_compact.h): few enough topics still move
        const bool done = moved <= thresh || left <= 0;                   // :189 (mean <= tol), :174
        if (done || (HANDOFF && nlive <= handoff_at)) {
            if constexpr (TWL > 2) lds_row_wait(rowbuf);                  // no read may land after the loop (row 2 is in flight)
            if constexpr (SWL > 1) table_row_wait(sbuf);
            // (the hand-over sits INSIDE the loop, where the tile is alive anyway: behind the loop it would stretch the
            //  tile's live range over the exit paths and the allocator answers by spilling two tile rows in the loop)
            if constexpr (HANDOFF) {
                if (!done && !__syncthreads_or(bad)) {
                    quad_hand_over<TL, RWL, TWL, SWL>(p, smem, B, gam, doc, lo, N, it);
                    return;
                }
            }
            break;
        }
        wave_lds_exchange();
        double2 h0[NPIECE];
#pragma unroll
        for (int x = 0; x < NPIECE; ++x) h0[x] = mysrc[FL * x];
        const double cnt0 = count_of(0);
        double s1 = 1.0, cnt1h = 0.0;
        if constexpr (C1 > 0) {
            double a1[R1 > 0 ? R1 : 1];
#pragma unroll
            for (int i = 0; i < R1; ++i) a1[i] = TL == 16 ? dot8_two_chains(B[8 + i], tq) : dot8(B[8 + i], tq);
            row_partial(StaticIndex<2>());
            row_partial(StaticIndex<3>());                                // (the last row stays in the buffer: pass B starts with it)
            stream_partial(StaticIndex<3>());                             // (the last streamed row likewise)
            double s0 = finish_sum(h0);
            asm volatile("" : "+v"(s0));                                  // h0 is dead from here on
            wave_lds_exchange();                                          // the writes below stay behind the reads above
            if constexpr (PRE && R1 > 0) asm volatile("s_nop 1" : "+v"(a1[R1 - 1]));       // (as above: dot8's last add)
            if constexpr (PRE && TWL > 0) asm volatile("s_nop 1" : "+v"(pr[TWL - 1]));
#pragma unroll
            for (int i = 0; i < R1; ++i) myred[i * RS + cw] = PRE ? lane_group_sum<2>(a1[i]) : a1[i];
#pragma unroll
            for (int t = 0; t < TWL; ++t) myred[(R1 + t) * RS + cw] = PRE ? lane_group_sum<2>(pr[t]) : pr[t];
#pragma unroll
            for (int s = 0; s < SWL; ++s) myred[(R1 + TWL + s) * RS + cw] = PRE ? lane_group_sum<2>(ps[s]) : ps[s];
            wave_lds_exchange();
            double2 h1[NPIECE];
#pragma unroll
            for (int x = 0; x < NPIECE; ++x) h1[x] = mysrc[FL * x];
            const double cnt1 = count_of(1);
            // the reciprocal chain of the first chunk runs while the second transpose is in flight; the second
            // chunk's chain is placed behind the first 32 FMAs of pass B (which need r0 only)
            if (exists0 && !(s0 > 1e-280)) bad = 1;   // (B, t <= 1: a normaliser cannot overflow; NaN fails the compare;
            r0 = cnt0 * rcp_newton(s0);               //  an empty slot's sum_k t_k is not below any real word's normaliser)
            s1 = finish_sum(h1);
            cnt1h = cnt1;
        } else {
            const double s0 = finish_sum(h0);
            if (exists0 && !(s0 > 1e-280)) bad = 1;
            r0 = cnt0 * rcp_newton(s0);
        }

        dpp_source_ready(r0);

        // B. q[k] over this lane's words (registers and LDS rows interleaved), then over the word groups
        double q[KRL];
        // Pass B walks the LDS slots BACKWARDS: the row pass A used last is still in the buffer, and the one this pass
        // uses last (slot 0) stays there for pass A of the next iteration - 2 (TWL - 1) row requests per iteration
        // instead of 2 TWL (with one LDS slot the row simply lives in the buffer)
        auto row_topic_sums = [&](auto idx) {                // step u: LDS slot t = TWL - 1 - u: q += r * row, next row requested
            constexpr int t = TWL - 1 - decltype(idx)::value;
            if constexpr (t >= 0) {
                lds_row_wait(rowbuf);
                double row[8];
                rowbuf.unpack(row);
                row_bcast_fmac<2 * (R1 + t)>(q, r1, row);
                if constexpr (t > 0) request_row(t - 1);
            }
        };
        auto stream_topic_sums = [&](auto stop) {            // the streamed slots backwards
            constexpr int u = quad_stream_stop(SWL, decltype(stop)::value), s = SWL - 1 - u;
            if constexpr (u >= 0) {
                if (swave[s]) {                              // (the first live one is still in the buffer)
                    table_row_wait(sbuf);
                    double row[8];
                    sbuf.unpack(row);
                    row_bcast_fmac<2 * (R1 + TWL + s)>(q, r1, row);
                    if constexpr (s > 0) request_srow(s - 1);
                }
            }
        };
        {
            const double rb = row_bcast<0>(r0);            // r of slot i sits in lane 2*i of every 16-lane row of the group
#pragma unroll
            for (int j = 0; j < KRL; ++j) q[j] = rb * B[0][j];
        }
        static_for<(C0 < 4 ? C0 : 4) - 1>([&](auto idx) {
            constexpr int i = decltype(idx)::value + 1;
            row_bcast_fmac<2 * i>(q, r0, B[i]);
        });
        if constexpr (C1 > 0) {
            asm volatile("" : "+v"(s1));                                  // (keeps the chain below behind the FMAs above)
            if (exists1 && !(s1 > 1e-280)) bad = 1;
            r1 = cnt1h * rcp_newton(s1);
            dpp_source_ready(r1);
        }
        stream_topic_sums(StaticIndex<0>());
        row_topic_sums(StaticIndex<0>());
        stream_topic_sums(StaticIndex<1>());
        static_for<(C0 > 4 ? C0 - 4 : 0)>([&](auto idx) {
            constexpr int i = decltype(idx)::value + 4;
            row_bcast_fmac<2 * i>(q, r0, B[i]);
        });
        row_topic_sums(StaticIndex<1>());
        stream_topic_sums(StaticIndex<2>());
        static_for<R1>([&](auto idx) {
            constexpr int i = decltype(idx)::value;
            row_bcast_fmac<2 * i>(q, r1, B[8 + i]);
        });
        row_topic_sums(StaticIndex<2>());
        row_topic_sums(StaticIndex<3>());
        stream_topic_sums(StaticIndex<3>());
        // over the word groups of the wavefront; the per-wavefront partials go to the (now idle) transpose area
        wave_lds_exchange();
        double* mysp = red + (size_t)wave * (L::red_wave / 8);
        if constexpr (TL == 16) {
            double u[KRL / 2];
#pragma unroll
            for (int m = 0; m < KRL / 2; ++m) u[m] = swap32_add(q[m], q[m + KRL / 2]);
#pragma unroll
            for (int m = 0; m < QV; ++m) {
                const double v = swap16_add(u[m], u[m + QV]);
                const int j = m + (g & 1) * QV + (g >> 1) * (KRL / 2);     // register index of the topic
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
        // both coefficient tables of exp_digamma_minus_levels, requested ahead of the barrier: the scalar-cache
        // round trips (150-200 ticks each when taken inside the phase) ride on the barrier wait
        ExpDigammaLevelsA coef_a;
        ExpDigammaLevelsB coef_b;
        if (topic_thread) {
            coef_a.load();
            coef_b.load();
        }
        __syncthreads();

        // C. gamma update by the topic threads
        if (topic_thread) {
            double part_sum[W];
#pragma unroll
            for (int w = 0; w < W; ++w) part_sum[w] = red[(size_t)w * (L::red_wave / 8) + ktid];
            const double t_mine = tt[buf * KT + ktid], alpha_k = alf[ktid];
            keep_together(part_sum);
            double s0 = part_sum[0] + part_sum[1], s1 = part_sum[2] + part_sum[3];
            if constexpr (W == 8) {
                s0 += part_sum[4] + part_sum[5];
                s1 += part_sum[6] + part_sum[7];
            }
            const double gnew = fma(t_mine, s0 + s1, fabs(alpha_k));      // :185 (the sign bit: a topic that never counts as dead)
            const double diff = fabs(gnew - gam);                         // :187
            gpv[ktid] = gam;
            gam = gnew;                                                   // :188
            atomicAdd(&chg[buf], change_fixed(diff));
            if constexpr (HANDOFF) {   // (a padding topic has alpha = gamma = 1 and t = 0: never counted)
                const unsigned long long moving = __ballot(gnew != alpha_k);
                if (lane == 0) atomicAdd(&livec[buf], (unsigned)__builtin_popcountll(moving));
            }
            const double t_next = exp_digamma_minus_levels<true>(gam, psi_total, coef_a, &coef_b);
            tt[(buf ^ 1) * KT + ktid] = topic_live ? t_next : 0.0;
            if (ktid == 0) {
                store_u64_hi(&chg[buf ^ 1], 0u);
                if constexpr (HANDOFF) livec[buf ^ 1] = 0u;
            }
        }
        ++it;
        --left;
        __syncthreads();
        // (both uniform.  The live count always sits in a scalar register; the stop sum where that frees a vector register
        //  pair the allocator can use - these kernels sit AT the 256-register limit and the allocation is chaotic around
        //  it: tools/kernel_resources.py "HOT BLOCK" lines, checked by tests/test_kernel_resources.py)
        if constexpr (TL == 32) {
            const unsigned long long m = chg[buf];
            moved = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)m));
        } else {
            moved = (long long)chg[buf];
        }
        if constexpr (HANDOFF) nlive = __builtin_amdgcn_readfirstlane((int)livec[buf]);
#pragma unroll
        for (int jj = 0; jj < KRL / 2; ++jj) {
            const double2 t2 = reinterpret_cast<const double2*>(tt + (buf ^ 1) * KT)[c + TL * jj];
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

    // ---- training fast path: the document terms are left to doc_terms_kernel (doc_terms.h), which recomputes
    //      them from gamma, t and r at full occupancy instead of on this workgroup's handful of wavefronts ----
    if (!p.heldout && !p.want_doc_ll) {
        if (live0 && part == 0) p.rfinal[lo + word0] = r0;
        if (live1 && part == 0) p.rfinal[lo + word1] = r1;
        if (topic_thread) {
            if (topic_live) p.gamma[(size_t)doc * K + ktid] = gam;
            p.tfinal[(size_t)doc * ldk + ktid] = topic_live ? tt[last * KT + ktid] : 0.0;
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
        const double2 t2 = reinterpret_cast<const double2*>(tt + last * KT)[c + TL * jj];
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
            const int n = s * 16 + (s < WPR ? gg : 15 - gg);
            if (n < N) {
                const double2* row = gtable + (size_t)p.term_id[lo + n] * ldk2 + c;
                double gsum2 = 0.0;
#pragma unroll
                for (int jj = 0; jj < KRL / 2; ++jj) {
                    const double2 g2 = row[TL * jj];
                    gsum2 = fma(g2.y, tq[2 * jj + 1], fma(g2.x, tq[2 * jj], gsum2));
                }
                term1 = fma(rl[s], gsum2, term1);
            }
        }
    }
    // c_n log(normaliser_n) from r_n = c_n / normaliser_n (the normalisers themselves were not kept)
    const bool owner0 = live0 && part == 0, owner1 = live1 && part == 0;
    const double cnt0 = count_of(0), cnt1 = count_of(1);
    double term3 = (owner0 ? cnt0 * (log(cnt0) - log(r0)) : 0.0) + (owner1 ? cnt1 * (log(cnt1) - log(r1)) : 0.0);
    double shift_term = 0.0;
    if (p.heldout) {
        if (owner0) shift_term = cnt0 * p.shift[p.term_id[lo + word0]];
        if (owner1) shift_term = fma(cnt1, p.shift[p.term_id[lo + word1]], shift_term);
    } else {
        if (owner0) p.rfinal[lo + word0] = r0;
        if (owner1) p.rfinal[lo + word1] = r1;
    }
    TopicShare share;
    if (topic_thread)
        topic_share(p, doc, ktid, ldk, topic_live, true, gam, fabs(alf[ktid]), gpv[ktid], tt[last * KT + ktid], psi_total, share);
    finish_document<W>(p, doc, it, misc, lane, wave, tid, term1, term3, shift_term, share);
}

}  // namespace pylda
