// The host-side planner of libpylda_hip.so as pure functions over host arrays - no HIP types, no device calls:
//   * launch classes of the document kernels: kernel variant and geometry per distinct-term count;
//   * the statistics pass: document blocks, segment cut, XCD execution order, rounds under a byte budget, the
//     persistent sweep's term dealing.
// plan.hip and sstats_gather.hip fill the small configuration structs from the context and move the results to the
// device.  This translation unit is also compiled on its own with g++ -fsanitize=address,undefined and fuzzed in the
// CPU suite (tests/test_planner_sanitizers.py, tests/native/planner_fuzz.cpp): the index arithmetic the kernels rely on
// - segments that partition the postings, partial-row indices inside the rows allocated, every document in exactly
// one launch class whose geometry admits its length - is checked there, without a GPU.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <thread>
#include <utility>
#include <vector>

#include "estep_limits.h"

namespace pylda_plan {

// ---- launch classes ----------------------------------------------------------------------------------------------

enum Variant : int {
    kGeneric64 = 0,    // 1 wavefront / document, tile in LDS
    kGeneric256 = 1,   // 4 wavefronts / document, tile in LDS
    kGeneric512 = 2,   // 8 wavefronts / document, tile in LDS (up to the whole 160 KiB)
    kGenericGlobal = 3, // tile larger than LDS: rows re-read from the table
    kSlab = 4,          // tile in registers, word-major lanes (estep_slab.h)
    kRetired5 = 5,      // (the topic-major column kernel of round 1: measured 2x slower than the quilt layout, removed)
    kQuilt = 6,         // tile in registers, 4 x 16 word-group x topic lanes (estep_quilt.h)
    kRetired7 = 7,      // (rounds 1-2: two-pass streaming, hybrid and wide tiered kernels - their streamed tier was read
    kRetired8 = 8,      //  twice per iteration; replaced by the quad kernel's streamed slots and estep_qgroup.h)
    kQgroup = 9,        // table stride 64 / 128 / 256, more than 256 terms: rows streamed once per iteration, fused per word group (estep_qgroup.h)
    kQuad = 10,         // 16 word groups / document, tile in registers + LDS rows (+ streamed slots) (estep_quad.h)
    kQfuse = 11,        // table stride 384 / 512: rows streamed ONCE per iteration, normaliser and topic sums fused (estep_qfuse.h)
    kGenericHuge = 12,  // a document too long even for its per-term scalars in LDS: those in global memory too (estep_generic.h MODE 2)
    kQfusek = 13,       // table stride 640 .. 1024: every row streamed once per iteration, fused (estep_qfusek.h)
    kVariantLast = kQfusek
};

struct Launch {
    int variant;
    int64_t first;   // offset into the sorted order
    int64_t count;   // documents (= workgroups)
    int n_cap;       // largest distinct-term count in the launch
    int tile_stride;
    size_t lds_bytes;
    int rn;          // geometry code (slab: words per lane; quilt: W * 100 + RWL; quad: SWL * 1000000 + TL * 10000 + RWL * 100 + TWL)
    int rk;          // slab kernels: topics per wavefront
};

// what the launch plan depends on (from the context and its options)
struct PlanConfig {
    int K = 0, V = 0, ldk = 0, num_cu = 256;
    size_t lds_limit = 64 * 1024;
    int force_variant = -1;
    bool exact_stop = false;        // this E-step's threshold is outside the fixed-point stop test's range: generic kernels
    int quad = 1, quad_stream = 1, quilt12 = 0, quilt_odd = 1, slab_uber = 1;
};

// table stride: K rounded up to 16 / 32 / 64 / 128 / 256 (the strides the register kernels are built for: 129-192 topics
// run the stride-256 kernels), from 257 to 1024 to a multiple of 128 (the fused streaming kernels' rows are 64 lanes x
// 16-byte pieces), beyond to a multiple of 64
inline int table_stride_for(int K)
{
    return K <= 16 ? 16 : K <= 32 ? 32 : K <= 64 ? 64 : K <= 128 ? 128 : K <= 256 ? 256 : K <= 1024 ? (K + 127) / 128 * 128 : (K + 63) / 64 * 64;
}
inline int tile_stride_for(int K) { return K | 1; }   // LDS tile row stride of the generic kernels: odd => conflict-free ds_read_b64 along words

int choose_variant(const PlanConfig& cfg, int n, size_t* lds_bytes);
// geometry code and slab width of variant `v` for a document of n distinct terms
int geometry_for(const PlanConfig& cfg, int variant, int n, int* rk);
// largest distinct-term count the (variant, geometry) pair can hold; the fuzz harness holds every document against it
int64_t capacity_of(const PlanConfig& cfg, int variant, int rn, int rk, size_t lds_bytes);
// is there a kernel instantiation behind the geometry code (launch_small.hip / launch_quad.hip switch on the same codes)
bool geometry_is_instantiated(const PlanConfig& cfg, int variant, int rn, int rk);
// terms_sorted: distinct-term counts of the documents in schedule order (descending)
std::vector<Launch> build_launch_classes(const PlanConfig& cfg, const int32_t* terms_sorted, int64_t D);
// first class of the one-dispatch slab group (estep_slab.h, estep_slab_uber_kernel), or -1
int slab_uber_from(const PlanConfig& cfg, const std::vector<Launch>& plan);

// ---- statistics pass ------------------------------------------------------------------------------------------------

constexpr int kGatherSegment = 256;      // postings per segment of the dispatch-paced gather (sstats_kernels.h kSegment)
constexpr int kSweepSegmentCap = 64;     // ... of the persistent sweep (sstats_sweep.h kSweepSegment)
constexpr int kXcd = 8;

struct GatherConfig {
    int V = 0, ldk = 0, num_cu = 256;
    int64_t D = 0, nnz = 0;
    int gather_rows = 2, gather_blocks = -1, gather_sweep = 1, gather_round_mb = 0;
};

// document blocks of the gather (1: unblocked)
int document_blocks(const GatherConfig& g);
// the byte budget the DECISIONS are taken against (sweep or gather): the option, else 4 GiB - never the free memory, so
// that the statistics path (and with it the bits of the result) does not depend on what else occupies the device
double decision_budget(const GatherConfig& g);
// the byte budget the ROUNDS are sized with: the decision budget, capped by a quarter of the free device memory
double round_budget(const GatherConfig& g, size_t free_device_bytes);
struct SweepGeom { int T, WPB, passes; };
SweepGeom sweep_geometry(const GatherConfig& g);
// the persistent sweep instead of partial rows? (resident: its geometry fits one workgroup per CU)
bool sweep_wanted(const GatherConfig& g, int NB, bool resident);
// document blocks of the SWEEP given the gather's: two thirds - it walks a block's document range in sub-steps (sstats_sweep.h),
// so larger blocks cost no locality and a third of the rendezvous go (cfg 4: 240 blocks 38.6 ms, 160 36.2, 120 37.0)
int sweep_blocks(const GatherConfig& g, int NB);

// One host thread's share of the segment cut: the segments of a contiguous range of terms.
struct CutPiece {
    std::vector<int64_t> begin, end, per_word;
    std::vector<int32_t> block, per_block;      // per_block[b]: this piece's segments in document block b
    int v0 = 0;
    int64_t base = 0;                           // index of its first segment in the whole list
};
struct SegmentCut {
    std::vector<int64_t> seg_begin, seg_end, word_seg_ptr;     // segments in term order; V + 1 offsets
    std::vector<CutPiece> pieces;                              // (blocked cut only)
};
// col_ptr: V + 1 posting offsets; post_doc: the postings' documents (document order within a term).
// Blocked cut: a segment never crosses a document-block boundary nor exceeds `cap` postings; pieces start on multiples
// of 16 terms (the finalize pass' blocks of 256 statistics then never straddle a round).  Returns nullptr or what failed.
const char* cut_segments_blocked(const int64_t* col_ptr, const int32_t* post_doc, int V, int64_t D, int64_t nnz, int NB,
                                 int64_t cap, int nthreads, SegmentCut* out);
void cut_segments_plain(const int64_t* col_ptr, int V, SegmentCut* out);

// [passes][wavefronts][T]: the terms a wavefront of the sweep owns (-1: none), dealt by posting count
std::vector<int32_t> deal_terms(const int64_t* col_ptr, int V, int64_t nwaves, int T, int passes);

struct Round { int64_t seg_lo, seg_hi; int w_first, n_words; int64_t slot_lo, slot_count; int64_t ent_first, ent_blocks; };
struct RoundPlan {
    std::vector<Round> rounds;
    std::vector<int32_t> order;      // segment of every (workgroup, wavefront) slot of the XCD execution order, or -1
    int64_t partial_rows = 0, ent_blocks = 0;
};
inline int64_t finalize_blocks(int64_t n_words, int ldk) { return (n_words * ldk + 255) / 256; }
// rounds of the blocked gather under max_rows partial rows, each with its XCD-ordered execution list
RoundPlan plan_rounds(const SegmentCut& cut, int NB, int64_t max_rows, int ldk);
RoundPlan single_round(int64_t nseg, int V, int ldk);

// fn(0) .. fn(nthreads - 1), side by side
template <typename F>
void run_on_threads(int nthreads, F&& fn)
{
    std::vector<std::thread> workers;
    for (int t = 1; t < nthreads; ++t) workers.emplace_back(fn, t);
    fn(0);
    for (auto& w : workers) w.join();
}

}  // namespace pylda_plan
