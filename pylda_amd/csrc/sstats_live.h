// Sufficient statistics (variational_bayes.py:207) from the documents' LISTS of live topics.
//
// sstats_kernels.h / sstats_sweep.h gather, per posting (term w, document d), the document's whole row t_d[0 .. ldk):
// 2 KiB at K = 256 of which - once the live-topic kernel (estep_compact.h) has finished the document - all but a
// dozen entries are exactly the dead topics' 1e-114, i.e. nothing a statistic can see (eta = statistics + beta).  Those
// documents now leave a list: up to 60 (topic, t) pairs, twelve to a 128-byte line (estep_common.h), and this pass adds
//
//     acc[w][k_j] += r_dw t_dj        for the entries j of the list of d
//
// One wavefront per posting segment (the postings of a term, <= 256 each, in document order), the K accumulators of
// the segment in LDS (ldk doubles per wavefront): a posting is ONE ds_add_f64 with a lane per list entry - the lanes
// hit different topics, the LDS executes a wavefront's instructions in order, so every accumulator sees its postings
// in document order: bitwise reproducible, no global atomics.  A document that finished on a dense kernel (live_n = -1:
// short prefix, another kernel family, the safety net) adds its row tfinal[d] instead.  The finalize pass
// (sstats_kernels.h) sums a term's segment rows in order and applies B[w][k], as for the dispatch-paced gather.
//
// Per posting the pass moves ~200 bytes (a line of the list; r_dw; 12 bytes of posting) instead of 2 KiB, and the lists' first
// lines of the whole corpus (cfg 4: 1M x 128 B) fit the Infinity Cache: no document blocking, no rendezvous, no sweep.
#pragma once
#include "estep_common.h"

namespace pylda {

constexpr int kLiveSegment = 256;       // == kSegment: the plain segment cut of the postings

// U: postings whose lists are in flight together
template <int U, typename P>
__global__ __launch_bounds__(256) void sstats_gather_live_kernel(
    const int64_t* __restrict__ seg_begin, const int64_t* __restrict__ seg_end, int64_t nseg,
    const int32_t* __restrict__ post_doc, const P* __restrict__ post_pos,
    const double* __restrict__ tfinal, const double* __restrict__ rfinal,
    const int32_t* __restrict__ live_n, const char* __restrict__ live_list, int ldk, double* __restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) double acc_all[];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int64_t seg = (int64_t)blockIdx.x * 4 + wave;
    if (seg >= nseg) return;                                    // (no workgroup barrier below: a wavefront is on its own)
    double* acc = acc_all + (size_t)wave * ldk;
    for (int k = lane; k < ldk; k += kWave) acc[k] = 0.0;
    wave_lds_exchange();
    const int64_t b = seg_begin[seg], e = seg_end[seg];
    for (int64_t chunk = b; chunk < e; chunk += kWave) {
        const int n = (int)(e - chunk < kWave ? e - chunk : kWave);
        const bool mine = lane < n;
        const int d = mine ? post_doc[chunk + lane] : 0;
        const double r = mine ? rfinal[post_pos[chunk + lane]] : 0.0;
        const int listed = mine ? live_n[d] : 0;
        // are all of the chunk's documents listed?  (wavefront-uniform; the rule in the timed window)
        const bool all_listed = __builtin_amdgcn_ballot_w64(mine && listed < 0) == 0ull;
        if (all_listed) {
            for (int p = 0; p < n; p += U) {
                int k[U];
                double t[U], rr[U];
                bool on[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {                   // (past the chunk: document 0, no entries)
                    const int at = p + u < n ? p + u : 0;
                    const int doc = __builtin_amdgcn_readlane(d, at);
                    const int entries = p + u < n ? __builtin_amdgcn_readlane(listed, at) : 0;
                    rr[u] = readlane_f64(r, at);
                    on[u] = lane < entries;
                    char* list = live_list_of(const_cast<char*>(live_list), doc);
                    k[u] = on[u] ? *live_idx_at(list, lane) : 0;
                    t[u] = on[u] ? *live_t_at(list, lane) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)                     // posting order: the LDS keeps a wavefront's instructions in order
                    if (on[u]) __hip_atomic_fetch_add(acc + k[u], rr[u] * t[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            for (int p = 0; p < n; ++p) {
                const int doc = __builtin_amdgcn_readlane(d, p);
                const int entries = __builtin_amdgcn_readlane(listed, p);
                const double rr = readlane_f64(r, p);
                if (entries >= 0) {
                    if (lane < entries) {
                        char* list = live_list_of(const_cast<char*>(live_list), doc);
                        const int k = *live_idx_at(list, lane);
                        const double t = *live_t_at(list, lane);
                        __hip_atomic_fetch_add(acc + k, rr * t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
                    const double* row = tfinal + (size_t)doc * ldk;
                    wave_lds_exchange();                        // (plain read-modify-write of this lane's own accumulators)
                    for (int k = lane; k < ldk; k += kWave) acc[k] = fma(rr, row[k], acc[k]);
                    wave_lds_exchange();
                }
            }
        }
    }
    wave_lds_exchange();
    for (int k = lane; k < ldk; k += kWave) partial[(size_t)seg * ldk + k] = acc[k];
}

}  // namespace pylda
