// Sufficient statistics (variational_bayes.py:207) as ONE persistent sweep, no partial rows.
//
// sstats_kernels.h blocks the gather by documents so that a block's t rows are L2 hits, and paces the workgroups by
// dispatch order; the price is a partial row per (term, document block) pair - written once, read once by the finalize
// pass: 1.2 + 1.3 GB of the 28 GB an E-step moves at cfg 3, 45 + 46 GB of 843 GB at cfg 4, and 4 GiB of rows in rounds.
// Here every wavefront OWNS a handful of terms and keeps their K accumulators in registers while the whole chip sweeps
// the document blocks together:
//
//     for pass (the terms that fit the register file at once: all 50 k of cfg 3, a third of cfg 4's 100 k)
//         for block b = 0 .. NB-1            every wavefront: the postings of ITS terms that fall into block b
//             rendezvous of all workgroups   (pacing only - see below)
//         sstats[w][:] = B[w][:] * acc,  entropy term            for the owned terms: the finalize pass folded in
//
// Measured (tools/gather_ab.py): cfg 4 (3 passes x 240 blocks) 47.8 ms against 50.2 for the dispatch-paced gather in 12
// rounds, without pacing (spin limit 0) 59.9; cfg 3 (1 pass x 24 blocks) 1.84 against 1.63 - a rendezvous costs ~25 us of
// arrival skew (letting workgroups run one or two blocks ahead: 56-60 ms at cfg 4, the locality goes), so the sweep is
// the default only where the partial rows would not fit their budget (option gather_sweep).  The rendezvous itself is
// cheap (tools/rendezvous_bench.hip: 3.7 us for 256 workgroups on one counter, 2.4 with per-XCD counters, idle chip);
// what a block costs beside its rows is the skew of the arrivals and the latency chain bounds -> postings -> r in
// front of the rows.  A three-deep software pipeline of those stages over (block, group) items was built and measured:
// its state no longer fits 128 registers beside the accumulators (scratch traffic inside the loop) - cfg 4 57.9 ms,
// cfg 3 2.15 ms - so the groups of four stay.  L2-sized blocks (480 at cfg 4) lose to 240 for the same reason: the
// per-block cost, not the fabric, is the larger term.
//
// The rendezvous is what keeps a block's rows (one L2's worth) hot in every XCD's L2 while all 256 CUs read them: each
// row comes over the fabric once per XCD and pass instead of once per posting.  It carries NO data - everything the
// sweep reads was written by earlier kernels - so it needs no release / acquire, and a rendezvous that times out costs
// locality, never correctness: the spin is bounded and the kernel then simply goes on.  One workgroup per CU, all
// resident (the host checks the occupancy), a monotonic counter polled by one lane with s_sleep.
//
// Per block a wavefront's terms are handled four at a time, stage by stage - segment bounds of the four, then their
// postings (document, CSR position), then r, then the rows - so that a block costs T dependent memory round trips, not 4 T
// (segments are cut at 64 postings for the sweep: one chunk each; a term with more than 64 postings in a block, or a
// second segment there, takes the loop behind the batch).  Terms are dealt to the wavefronts by posting count (largest
// first, boustrophedon), so a wavefront's work per block is within a few per cent of every other's.  A term's postings
// are walked in document order with one accumulator chain per (lane, piece): a fixed summation order, bitwise
// reproducible.
#pragma once
#include "estep_common.h"

namespace pylda {

constexpr int kSweepSegment = 64;
constexpr int kSweepCounters = 8;        // rendezvous counters: one per XCD (workgroups are dealt to the XCDs round-robin)

struct SweepParams {
    const int64_t* seg_begin;       // segments of the postings, cut at document-block boundaries, in term order
    const int64_t* seg_end;
    const int32_t* seg_block;       // document block of a segment
    const int64_t* word_seg_ptr;    // V + 1
    const int32_t* post_doc;
    const void* post_pos;           // int32 / int64 CSR positions
    const double* tfinal;
    const double* rfinal;
    const double* expElog;
    const double* expElog_elog;
    double* sstats;
    double* entropy_partial;        // [passes][wavefronts]
    const int32_t* term_of;         // [passes][wavefronts][T]: owned terms, -1 = none
    int passes;
    int NB;
    unsigned* rendezvous;           // kSweepCounters counters, 32 words apart, zeroed before the launch
    int per_xcd;                    // 1: the workgroups of an XCD meet among themselves (the L2 they share is what the pacing is for)
    unsigned spin_limit;
    int per_block;                  // documents per document block
    int sub;                        // sub-blocks a block's document range is walked in (see sweep_rows)
};

// All workgroups of the grid meet (pacing only): bounded, no memory ordering implied.
__device__ __forceinline__ void sweep_rendezvous(unsigned* counter, unsigned target, unsigned spin_limit)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (unsigned spin = 0; spin < spin_limit; ++spin) {
            if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();
}

// acc += sum over the postings held one per lane (document d, factor r) in lanes [from, to), U rows in flight
template <int NP, int U>
__device__ __forceinline__ void sweep_rows(f64x2 (&acc)[NP], const double* __restrict__ tfinal, int ldk, int lane, int d, double r, int from,
                                           int to)
{
    for (int q = from; q < to; q += U) {
        f64x2 row[U][NP];
        double rr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                           // (past the range: the range's first row again, factor 0)
            const int at = q + u < to ? q + u : from;
            const int doc = __builtin_amdgcn_readlane(d, at & (kWave - 1));
            rr[u] = q + u < to ? readlane_f64(r, at & (kWave - 1)) : 0.0;
            const f64x2* src = reinterpret_cast<const f64x2*>(tfinal + (size_t)doc * ldk) + lane;
#pragma unroll
            for (int j = 0; j < NP; ++j) row[u][j] = src[64 * j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)                             // posting order: one chain per (lane, piece)
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                acc[j].x = fma(rr[u], row[u][j].x, acc[j].x);
                acc[j].y = fma(rr[u], row[u][j].y, acc[j].y);
            }
    }
}

// The same for TWO terms side by side, two rows each per trip: a sub-step of a block leaves a term ~2 postings, and four
// rows in flight per wavefront are what hides the round trip.  Each accumulator still sees its postings in lane order.
template <int NP>
__device__ __forceinline__ void sweep_rows_pair(f64x2 (&accA)[NP], f64x2 (&accB)[NP], const double* __restrict__ tfinal, int ldk, int lane,
                                                int dA, double rA, int fromA, int toA, int dB, double rB, int fromB, int toB)
{
    for (int qa = fromA, qb = fromB; qa < toA || qb < toB; qa += 2, qb += 2) {
        f64x2 row[4][NP];
        double rr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                           // (past a range: document 0, factor 0)
            const int at = u < 2 ? qa + u : qb + u - 2;
            const bool valid = u < 2 ? at < toA : at < toB;
            const int doc = valid ? __builtin_amdgcn_readlane(u < 2 ? dA : dB, at & (kWave - 1)) : 0;
            rr[u] = valid ? readlane_f64(u < 2 ? rA : rB, at & (kWave - 1)) : 0.0;
            const f64x2* src = reinterpret_cast<const f64x2*>(tfinal + (size_t)doc * ldk) + lane;
#pragma unroll
            for (int j = 0; j < NP; ++j) row[u][j] = src[64 * j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                f64x2& a = u < 2 ? accA[j] : accB[j];
                a.x = fma(rr[u], row[u][j].x, a.x);
                a.y = fma(rr[u], row[u][j].y, a.y);
            }
    }
}

// NCH = ldk / 64 (2 or 4), T = terms per wavefront, WPB = wavefronts per workgroup, P = type of a CSR position
template <int NCH, int T, int WPB, typename P>
__global__ __launch_bounds__(kWave* WPB) void sstats_sweep_kernel(SweepParams p)
{
    static_assert(NCH == 2 || NCH == 4, "table stride 128 or 256");
    static_assert(T % 4 == 0, "terms are staged four at a time");
    constexpr int ldk = 64 * NCH, NP = NCH / 2, U = 4, GS = 4;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int nwaves = gridDim.x * WPB;
    const int gw = blockIdx.x * WPB + wave;
    const P* post_pos = static_cast<const P*>(p.post_pos);
    unsigned meet = 0;
    for (int pass = 0; pass < p.passes; ++pass) {
        int term[T], cs[T], hi[T], nb[T];
        f64x2 acc[T][NP];
#pragma unroll
        for (int i = 0; i < T; ++i) {
            term[i] = p.term_of[((size_t)pass * nwaves + gw) * T + i];
            cs[i] = term[i] >= 0 ? (int)p.word_seg_ptr[term[i]] : 0;
            hi[i] = term[i] >= 0 ? (int)p.word_seg_ptr[term[i] + 1] : 0;
            nb[i] = cs[i] < hi[i] ? p.seg_block[cs[i]] : 0x7fffffff;
#pragma unroll
            for (int j = 0; j < NP; ++j) acc[i][j] = f64x2{0.0, 0.0};
        }
        for (int b = 0; b < p.NB; ++b) {
            // ---- the first segment of every owned term in this block, stage by stage, GS terms at a time (all T at
            //      once needs 7 staging registers per term beside the accumulators: spills) ----
            static_for<T / GS>([&](auto group) {
                constexpr int i0 = decltype(group)::value * GS;
                int64_t sb[GS];
                int n[GS], d[GS];
                double r[GS];
                P pos[GS];
#pragma unroll
                for (int x = 0; x < GS; ++x) {                          // stage 1: bounds
                    const bool here = nb[i0 + x] == b;
                    sb[x] = here ? p.seg_begin[cs[i0 + x]] : 0;
                    n[x] = here ? (int)(p.seg_end[cs[i0 + x]] - sb[x]) : 0;
                }
#pragma unroll
                for (int x = 0; x < GS; ++x) {                          // stage 2: postings (one per lane)
                    const bool mine = lane < n[x];
                    d[x] = mine ? p.post_doc[sb[x] + lane] : 0;
                    pos[x] = mine ? post_pos[sb[x] + lane] : (P)0;
                }
#pragma unroll
                for (int x = 0; x < GS; ++x) r[x] = lane < n[x] ? p.rfinal[pos[x]] : 0.0;      // stage 3: r
                // stage 4: rows.  A block's t rows (cfg 4: 8.5 MB) do not fit an XCD's L2 (4 MB) - 63 % of the rows came
                // over the fabric, at its rate - so the block's document range is walked in `sub` steps: a term's postings are
                // in document order, the lanes below the step's bound are a prefix, and all wavefronts of the XCD move
                // through the block roughly together with a working set of 1 / sub of it.  Same postings in the same order
                // per accumulator: bitwise the one-step result.
                int done[GS];
#pragma unroll
                for (int x = 0; x < GS; ++x) done[x] = 0;
                for (int step = 1; step <= p.sub; ++step) {
                    const int bound = step == p.sub ? 0x7fffffff : b * p.per_block + (int)((int64_t)p.per_block * step / p.sub);
                    int upto[GS];
#pragma unroll
                    for (int x = 0; x < GS; ++x)
                        upto[x] = n[x] > done[x] ? __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < n[x] && d[x] < bound)) : done[x];
#pragma unroll
                    for (int x = 0; x < GS; x += 2) {
                        if (upto[x] > done[x] || upto[x + 1] > done[x + 1])
                            sweep_rows_pair<NP>(acc[i0 + x], acc[i0 + x + 1], p.tfinal, ldk, lane, d[x], r[x], done[x], upto[x], d[x + 1],
                                                r[x + 1], done[x + 1], upto[x + 1]);
                        done[x] = upto[x];
                        done[x + 1] = upto[x + 1];
                    }
                }
#pragma unroll
                for (int x = 0; x < GS; ++x) {
                    if (n[x] > 0) {
                        ++cs[i0 + x];
                        nb[i0 + x] = cs[i0 + x] < hi[i0 + x] ? p.seg_block[cs[i0 + x]] : 0x7fffffff;
                    }
                }
            });
            // ---- further segments of a term in this block (more than 64 postings there): one at a time ----
#pragma unroll
            for (int i = 0; i < T; ++i) {
                while (nb[i] == b) {                                    // (wavefront-uniform)
                    const int64_t s0 = p.seg_begin[cs[i]];
                    const int m = (int)(p.seg_end[cs[i]] - s0);
                    const bool mine = lane < m;
                    const int dd = mine ? p.post_doc[s0 + lane] : 0;
                    const double rr = mine ? p.rfinal[post_pos[s0 + lane]] : 0.0;
                    sweep_rows<NP, U>(acc[i], p.tfinal, ldk, lane, dd, rr, 0, m);
                    ++cs[i];
                    nb[i] = cs[i] < hi[i] ? p.seg_block[cs[i]] : 0x7fffffff;
                }
            }
            if (b + 1 < p.NB || pass + 1 < p.passes) {
                ++meet;
                if (p.per_xcd) {
                    // workgroup g runs on XCD g % 8: only those 32 share an L2, so only they need to walk the blocks
                    // together - a rendezvous of 32 has less arrival skew than one of 256, and the XCDs may drift
                    const unsigned xcd = blockIdx.x % kSweepCounters;
                    const unsigned members = (gridDim.x + kSweepCounters - 1 - xcd) / kSweepCounters;
                    sweep_rendezvous(p.rendezvous + 32 * xcd, meet * members, p.spin_limit);
                } else {
                    sweep_rendezvous(p.rendezvous, meet * gridDim.x, p.spin_limit);
                }
            }
        }
        // the finalize pass for the owned terms: sstats = B * acc, and the corpus entropy term the document kernels
        // skip on the training fast path (sstats_kernels.h, sstats_finalize_kernel)
        double ent = 0.0;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            if (term[i] >= 0) {
                const size_t base = (size_t)term[i] * ldk;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const f64x2 bv = reinterpret_cast<const f64x2*>(p.expElog + base)[lane + 64 * j];
                    const f64x2 gv = reinterpret_cast<const f64x2*>(p.expElog_elog + base)[lane + 64 * j];
                    reinterpret_cast<f64x2*>(p.sstats + base)[lane + 64 * j] = f64x2{bv.x * acc[i][j].x, bv.y * acc[i][j].y};
                    ent = fma(gv.x, acc[i][j].x, ent);
                    ent = fma(gv.y, acc[i][j].y, ent);
                }
            }
        }
        ent = wave_sum(ent);
        if (lane == 0) p.entropy_partial[(size_t)pass * nwaves + gw] = ent;
    }
}

}  // namespace pylda
