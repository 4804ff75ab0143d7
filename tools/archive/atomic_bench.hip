// Micro-benchmark (not part of the product): throughput of the sstats
// scatter-add pattern on MI355X.  One wavefront adds a contiguous K-double
// row into a (V x K) table at a Zipf-distributed row; compares agent-scope
// and workgroup-scope f64 atomics, plain (racy) stores and the row gather.
#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void scatter(const int* __restrict__ rows, long nrows, int K, double* table, double* sink)
{
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    double acc = 0.0;
    for (long i = wave; i < nrows; i += nwaves) {
        double* row = table + (size_t)rows[i] * K;
        for (int k = lane; k < K; k += 64) {
            if (MODE == 0) unsafeAtomicAdd(&row[k], 1.0);                                                // agent scope (default)
            else if (MODE == 1) __hip_atomic_fetch_add(&row[k], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 2) row[k] = 1.0;                                                            // plain store
            else if (MODE == 3) acc += row[k];                                                           // gather
            else if (MODE == 4) __hip_atomic_fetch_add(&row[k], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (MODE == 3 && acc == -1.0) sink[0] = acc;
}

int main(int argc, char** argv)
{
    const int V = argc > 1 ? atoi(argv[1]) : 50000, K = argc > 2 ? atoi(argv[2]) : 128;
    const long nrows = argc > 3 ? atol(argv[3]) : 19600000;
    const double zipf = argc > 4 ? atof(argv[4]) : 1.0;
    std::vector<double> cdf(V);
    double s = 0; for (int v = 0; v < V; ++v) { s += 1.0 / pow(v + 1.0, zipf); cdf[v] = s; }
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(0, s);
    std::vector<int> rows(nrows);
    std::vector<int> perm(V); for (int v = 0; v < V; ++v) perm[v] = v;
    std::shuffle(perm.begin(), perm.end(), rng);
    for (long i = 0; i < nrows; ++i) rows[i] = perm[std::lower_bound(cdf.begin(), cdf.end(), U(rng)) - cdf.begin()];
    int* d_rows; double *d_table, *d_sink;
    CK(hipMalloc(&d_rows, nrows * sizeof(int)));
    CK(hipMalloc(&d_table, (size_t)V * K * 8 * 8));   // room for 8 private copies
    CK(hipMalloc(&d_sink, 8));
    CK(hipMemcpy(d_rows, rows.data(), nrows * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemset(d_table, 0, (size_t)V * K * 8 * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char* names[] = {"unsafeAtomicAdd(agent)", "atomic workgroup-scope", "plain store", "gather read", "hip_atomic agent-scope"};
    for (int grid : {2048, 8192}) {
        for (int mode = 0; mode < 5; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(a));
                switch (mode) {
                case 0: hipLaunchKernelGGL(scatter<0>, dim3(grid), dim3(256), 0, 0, d_rows, nrows, K, d_table, d_sink); break;
                case 1: hipLaunchKernelGGL(scatter<1>, dim3(grid), dim3(256), 0, 0, d_rows, nrows, K, d_table, d_sink); break;
                case 2: hipLaunchKernelGGL(scatter<2>, dim3(grid), dim3(256), 0, 0, d_rows, nrows, K, d_table, d_sink); break;
                case 3: hipLaunchKernelGGL(scatter<3>, dim3(grid), dim3(256), 0, 0, d_rows, nrows, K, d_table, d_sink); break;
                case 4: hipLaunchKernelGGL(scatter<4>, dim3(grid), dim3(256), 0, 0, d_rows, nrows, K, d_table, d_sink); break;
                }
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (rep == 1)
                    printf("grid %5d  %-26s %8.3f ms  %8.1f GB/s (row bytes)\n", grid, names[mode], ms,
                           (double)nrows * K * 8 / ms / 1e6);
            }
        }
    }
    return 0;
}
