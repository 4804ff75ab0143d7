// Sufficient-statistics accumulation without atomics (variational_bayes.py:207).
//
// The reference adds phi[n][k] * count[n] of every document into
// sstats[k][w_n].  In the exp-hoisted form that contribution factorises:
//
//     phi[n][k] * c_n = B[w_n][k] * t_d[k] * r_dn ,   r_dn = c_n / nrm_dn
//
// so   sstats[w][k] = B[w][k] * sum_{d containing w} r_dw * t_d[k] .
//
// The E-step kernels therefore only write t_d (K doubles per document) and
// r_dn (one double per term); this file does the word-major sum as a GATHER
// over the corpus' postings (CSC index built once when the corpus is
// uploaded): read-only traffic that runs at L2 / Infinity-Cache speed,
// versus 2.5e9 fp64 atomics per outer iteration at cfg 3 that measured
// 13-54 ms on MI355X depending on the word skew (tools/atomic_bench.hip).
// It is also bitwise reproducible: every sum has a fixed order.
//
// Work decomposition: a posting list is cut into segments of <= kSegment
// entries; one wavefront accumulates one (segment, 64-topic chunk) into a
// partial row, and a second small kernel adds a word's partial rows in order
// and applies the B[w][k] factor.
#pragma once
#include "estep_common.h"

namespace pylda {

constexpr int kSegment = 256;

// LR = lanes along topics (16, 32 or 64); 64 / LR postings are handled side by side.
// P: type of a CSR position (int32_t; int64_t for corpora of 2^31 or more (document, term) pairs).
template <int LR, typename P>
__global__ __launch_bounds__(256) void sstats_gather_kernel(
    const int64_t* __restrict__ seg_begin, const int64_t* __restrict__ seg_end, int64_t nseg,
    const int32_t* __restrict__ post_doc, const P* __restrict__ post_pos,
    const double* __restrict__ tfinal, const double* __restrict__ rfinal, int ldk,
    double* __restrict__ partial)
{
    constexpr int EP = kWave / LR;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t seg = (int64_t)blockIdx.x * 4 + threadIdx.x / kWave;
    if (seg >= nseg) return;
    const int kl = lane & (LR - 1), sub = lane / LR;
    const int k = blockIdx.y * 64 + kl;
    const int64_t b = seg_begin[seg], e = seg_end[seg];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int64_t i = b + sub;
    for (; i + 3 * EP < e; i += 4 * EP) {
        const int d0 = post_doc[i], d1 = post_doc[i + EP], d2 = post_doc[i + 2 * EP], d3 = post_doc[i + 3 * EP];
        const double r0 = rfinal[post_pos[i]], r1 = rfinal[post_pos[i + EP]];
        const double r2 = rfinal[post_pos[i + 2 * EP]], r3 = rfinal[post_pos[i + 3 * EP]];
        a0 = fma(r0, tfinal[(size_t)d0 * ldk + k], a0);
        a1 = fma(r1, tfinal[(size_t)d1 * ldk + k], a1);
        a2 = fma(r2, tfinal[(size_t)d2 * ldk + k], a2);
        a3 = fma(r3, tfinal[(size_t)d3 * ldk + k], a3);
    }
    for (; i < e; i += EP) a0 = fma(rfinal[post_pos[i]], tfinal[(size_t)post_doc[i] * ldk + k], a0);
    double acc = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int m = LR; m < kWave; m <<= 1) acc += __shfl_xor(acc, m, kWave);
    if (sub == 0) partial[(size_t)seg * ldk + k] = acc;
}

// Whole-row variant for ldk = 64*NCH (NCH = 1, 2, 4): one wavefront accumulates all topics of a
// segment, so each posting's t_d row is one contiguous ldk*8-byte read and the posting index is
// loaded once instead of once per 64-topic chunk.
//
// Document-blocked mode (exec_order != nullptr; sstats_gather.hip build_postings).  A term's postings are in document
// order, so cutting its segments at the boundaries of NB contiguous document blocks costs nothing - and a block's
// t rows (<= 3.3 MB) fit one XCD's 4 MB L2.  Workgroups are dispatched round-robin over the 8 XCDs, so workgroup g
// takes its four segments from the list of XCD g % 8, which holds the segments of blocks g % 8, g % 8 + 8, ...
// block after block: the t rows a CU gathers are then hits in ITS L2 instead of reads over the fabric (the
// unblocked gather runs at the fabric's ~6.7 TB/s for L2 misses, wherever the rows live).  The price is one
// partial row per (term, block): worth it while a pair holds >= 8 postings (cfg 3: 12; cfg 4: 3 - unblocked).
template <int NCH, typename P>
__global__ __launch_bounds__(256) void sstats_gather_rows_kernel(
    const int64_t* __restrict__ seg_begin, const int64_t* __restrict__ seg_end, int64_t nseg,
    const int32_t* __restrict__ post_doc, const P* __restrict__ post_pos,
    const double* __restrict__ tfinal, const double* __restrict__ rfinal, double* __restrict__ partial,
    const int32_t* __restrict__ exec_order, int64_t seg_lo)
{
    constexpr int ldk = 64 * NCH;
    const int lane = threadIdx.x & (kWave - 1);
    int64_t seg = (int64_t)blockIdx.x * 4 + threadIdx.x / kWave;
    if (exec_order) seg = exec_order[seg];          // (the grid covers the padded list exactly; -1: no segment)
    if (seg < 0 || seg >= nseg) return;
    partial += (size_t)(seg - seg_lo) * ldk;        // seg_lo: first segment of this round (rounds share the partial rows)
    const int64_t b = seg_begin[seg], e = seg_end[seg];
    double acc0[NCH], acc1[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) acc0[j] = acc1[j] = 0.0;
    int64_t i = b;
    for (; i + 1 < e; i += 2) {
        const double* t0 = tfinal + (size_t)post_doc[i] * ldk + lane;
        const double* t1 = tfinal + (size_t)post_doc[i + 1] * ldk + lane;
        const double r0 = rfinal[post_pos[i]], r1 = rfinal[post_pos[i + 1]];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            acc0[j] = fma(r0, t0[64 * j], acc0[j]);
            acc1[j] = fma(r1, t1[64 * j], acc1[j]);
        }
    }
    if (i < e) {
        const double* t0 = tfinal + (size_t)post_doc[i] * ldk + lane;
        const double r0 = rfinal[post_pos[i]];
#pragma unroll
        for (int j = 0; j < NCH; ++j) acc0[j] = fma(r0, t0[64 * j], acc0[j]);
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) partial[lane + 64 * j] = acc0[j] + acc1[j];
}

// The same pass with the postings' metadata fetched in bulk.  The kernel above walks a segment two postings at a
// time, each step a chain of three dependent fetches (posting -> r_dn at a random CSR position -> the t row):
// with 32 wavefronts per CU and two 1-KiB rows in flight each that is ~6.7 TB/s at ~2.5 us per step - the
// "ceiling" rounds 1 and 2 measured was this chain, not the fabric.  Here a wavefront loads up to 64 postings of
// its segment at once (lane l: document, position, then r), then walks them with the document and r of posting p
// read from lane p (v_readlane: scalar row base + lane offset, scalar multiplier), U rows in flight per wavefront,
// 16 bytes per lane and load.  Topic of (lane, piece j, half c): 128 j + 2 lane + c.  Fixed summation order.
template <int NCH, int U, typename P>
__global__ __launch_bounds__(256) void sstats_gather_bulk_kernel(
    const int64_t* __restrict__ seg_begin, const int64_t* __restrict__ seg_end, int64_t nseg,
    const int32_t* __restrict__ post_doc, const P* __restrict__ post_pos,
    const double* __restrict__ tfinal, const double* __restrict__ rfinal, double* __restrict__ partial,
    const int32_t* __restrict__ exec_order, int64_t seg_lo)
{
    static_assert(NCH == 2 || NCH == 4, "table stride 128 or 256");
    static_assert(64 % U == 0, "whole trips over a 64-posting chunk");
    constexpr int ldk = 64 * NCH, NP = NCH / 2;
    const int lane = threadIdx.x & (kWave - 1);
    int64_t seg = (int64_t)blockIdx.x * 4 + threadIdx.x / kWave;
    if (exec_order) seg = exec_order[seg];
    if (seg < 0 || seg >= nseg) return;
    const int64_t b = seg_begin[seg], e = seg_end[seg];
    f64x2 acc[U][NP];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < NP; ++j) acc[u][j] = f64x2{0.0, 0.0};
    for (int64_t chunk = b; chunk < e; chunk += kWave) {
        const int n = (int)(e - chunk < kWave ? e - chunk : kWave);
        const bool mine = lane < n;
        const int d = mine ? post_doc[chunk + lane] : 0;                  // lanes beyond the segment: row 0 times r = 0
        const double r = mine ? rfinal[post_pos[chunk + lane]] : 0.0;
        for (int p = 0; p < n; p += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int doc = __builtin_amdgcn_readlane(d, p + u);
                const double rr = readlane_f64(r, p + u);
                const f64x2* row = reinterpret_cast<const f64x2*>(tfinal + (size_t)doc * ldk) + lane;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const f64x2 t2 = row[64 * j];
                    acc[u][j].x = fma(rr, t2.x, acc[u][j].x);
                    acc[u][j].y = fma(rr, t2.y, acc[u][j].y);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        f64x2 s = acc[0][j];
#pragma unroll
        for (int u = 1; u < U; ++u) {
            s.x += acc[u][j].x;
            s.y += acc[u][j].y;
        }
        reinterpret_cast<f64x2*>(partial + (size_t)(seg - seg_lo) * ldk)[lane + 64 * j] = s;
    }
}

// sstats[w][k] = B[w][k] * sum over the word's segments (in order).
// Also the corpus-level entropy term the document kernels skip on the training
// fast path:  sum_d sum_n c_n sum_k phi_nk log B[w_n][k] = sum_{w,k} sstats[w][k] * (E_log_eta - shift)[w][k]
// = sum_{w,k} (B log B)[w][k] * acc[w][k]; one partial per workgroup, summed in order afterwards.
__global__ __launch_bounds__(256) void sstats_finalize_kernel(
    const int64_t* __restrict__ word_seg_ptr, const double* __restrict__ partial,
    const double* __restrict__ expElog, const double* __restrict__ expElog_elog, int w_first, int n_words, int ldk,
    int64_t seg_lo, double* __restrict__ sstats, double* __restrict__ entropy_partial)
{
    // words w_first .. w_first + n_words - 1 (one ROUND of the gather: the partial rows hold the segments from seg_lo on)
    __shared__ double scratch[4];
    const int64_t local = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double ent = 0.0;
    if (local < (int64_t)n_words * ldk) {
        const int w = w_first + (int)(local / ldk), k = (int)(local % ldk);
        const int64_t idx = (int64_t)w * ldk + k;
        double s = 0.0;
        for (int64_t sg = word_seg_ptr[w]; sg < word_seg_ptr[w + 1]; ++sg) s += partial[(size_t)(sg - seg_lo) * ldk + k];
        sstats[idx] = expElog[idx] * s;
        ent = expElog_elog[idx] * s;
    }
    ent = block_sum<256>(ent, scratch);
    if (threadIdx.x == 0) entropy_partial[blockIdx.x] = ent;
}

}  // namespace pylda
