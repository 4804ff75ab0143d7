// Postings (CSC index) of a corpus for the sufficient-statistics gather pass, built on the device.
//
// Round 1 counting-sorted the 2e8 postings of cfg 4 on one host thread with random access (about two
// E-steps' worth of time) and kept a second host copy of the term ids for it.  Here: a stable device
// radix sort of (term id, CSR position) pairs - rocPRIM's, a library primitive used as such - then one
// binary search per posting for its document and one per term for its first posting.
#include "postings.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

namespace pylda {

namespace {

template <typename P>
__global__ __launch_bounds__(256) void iota_kernel(P* out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (P)i;
}

// post_doc[i] = the document whose CSR range holds position post_pos[i]
template <typename P>
__global__ __launch_bounds__(256) void doc_of_position_kernel(const P* __restrict__ post_pos, int64_t n,
                                                              const int64_t* __restrict__ doc_ptr, int64_t D,
                                                              int32_t* __restrict__ post_doc)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t pos = post_pos[i];
    int64_t lo = 0, hi = D;             // largest d with doc_ptr[d] <= pos (empty documents share offsets)
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (doc_ptr[mid] <= pos) lo = mid;
        else hi = mid;
    }
    post_doc[i] = (int32_t)lo;
}

// col_ptr[v] = first index i with sorted_term[i] >= v   (v = 0 .. V)
__global__ __launch_bounds__(256) void first_posting_kernel(const int32_t* __restrict__ sorted_term, int64_t n, int V,
                                                            int64_t* __restrict__ col_ptr)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v > V) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sorted_term[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    col_ptr[v] = lo;
}

template <typename P>
hipError_t build_postings_typed(hipStream_t stream, int V, int64_t D, int64_t nnz, const int64_t* d_doc_ptr,
                                const int32_t* d_term_id, int32_t* d_post_doc, P* d_post_pos,
                                int64_t* h_col_ptr, const char** what)
{
    int32_t* d_sorted = nullptr;
    P* d_iota = nullptr;
    int64_t* d_col = nullptr;
    void* d_temp = nullptr;
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r, const char* name) {
        if (e == hipSuccess && r != hipSuccess) {
            e = r;
            *what = name;
        }
        return e == hipSuccess;
    };
    const size_t n = (size_t)(nnz > 0 ? nnz : 1);
    step(hipMalloc((void**)&d_sorted, n * sizeof(int32_t)), "postings: hipMalloc");
    step(hipMalloc((void**)&d_iota, n * sizeof(P)), "postings: hipMalloc");
    step(hipMalloc((void**)&d_col, ((size_t)V + 1) * sizeof(int64_t)), "postings: hipMalloc");
    if (e == hipSuccess && nnz > 0) {
        const unsigned blocks = (unsigned)((nnz + 255) / 256);
        hipLaunchKernelGGL(iota_kernel<P>, dim3(blocks), dim3(256), 0, stream, d_iota, nnz);
        int end_bit = 1;
        while (end_bit < 31 && (1ll << end_bit) < (int64_t)V) ++end_bit;
        size_t temp_bytes = 0;
        step(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_term_id, d_sorted, d_iota, d_post_pos, (size_t)nnz, 0,
                                       (unsigned)end_bit, stream), "postings: radix sort (sizing)");
        step(hipMalloc(&d_temp, temp_bytes > 0 ? temp_bytes : 1), "postings: hipMalloc (sort scratch)");
        if (e == hipSuccess)
            step(rocprim::radix_sort_pairs(d_temp, temp_bytes, d_term_id, d_sorted, d_iota, d_post_pos, (size_t)nnz, 0,
                                           (unsigned)end_bit, stream), "postings: radix sort");
        if (e == hipSuccess) {
            hipLaunchKernelGGL(doc_of_position_kernel<P>, dim3(blocks), dim3(256), 0, stream, d_post_pos, nnz, d_doc_ptr, D,
                               d_post_doc);
            step(hipGetLastError(), "postings: kernel launch");
        }
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(first_posting_kernel, dim3((unsigned)((V + 256) / 256)), dim3(256), 0, stream, d_sorted, nnz, V, d_col);
        step(hipGetLastError(), "postings: kernel launch");
        step(hipMemcpyAsync(h_col_ptr, d_col, ((size_t)V + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, stream), "postings: D2H");
        step(hipStreamSynchronize(stream), "postings: synchronize");
    }
    if (d_sorted) (void)hipFree(d_sorted);
    if (d_iota) (void)hipFree(d_iota);
    if (d_col) (void)hipFree(d_col);
    if (d_temp) (void)hipFree(d_temp);
    return e;
}

}  // namespace

hipError_t build_postings_device(hipStream_t stream, int V, int64_t D, int64_t nnz, const int64_t* d_doc_ptr,
                                 const int32_t* d_term_id, int32_t* d_post_doc, void* d_post_pos, bool wide_positions,
                                 int64_t* h_col_ptr, const char** what)
{
    if (wide_positions)
        return build_postings_typed<int64_t>(stream, V, D, nnz, d_doc_ptr, d_term_id, d_post_doc,
                                             static_cast<int64_t*>(d_post_pos), h_col_ptr, what);
    return build_postings_typed<int32_t>(stream, V, D, nnz, d_doc_ptr, d_term_id, d_post_doc,
                                         static_cast<int32_t*>(d_post_pos), h_col_ptr, what);
}

}  // namespace pylda
