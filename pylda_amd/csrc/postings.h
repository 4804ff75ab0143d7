// Postings (CSC) of a device-resident corpus, built on the device (see postings.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pylda __attribute__((visibility("hidden"))) {

// For every word the (document, CSR position) pairs of its occurrences, in document order:
//   post_pos[i]  position in the corpus' CSR arrays, grouped by term id (stable: document order inside a term);
//                int32 entries, or int64 (wide_positions) for corpora of 2^31 or more (document, term) pairs
//   post_doc[i]  document of that position
//   col_ptr[v]   (host, V + 1 entries) first posting of term v
// Returns hipSuccess or the failing HIP status; *what names the failing step.
hipError_t build_postings_device(hipStream_t stream, int V, int64_t D, int64_t nnz, const int64_t* d_doc_ptr,
                                 const int32_t* d_term_id, int32_t* d_post_doc, void* d_post_pos, bool wide_positions,
                                 int64_t* h_col_ptr, const char** what);

}  // namespace pylda
