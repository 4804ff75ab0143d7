// (rows x cols) -> (cols x rows) copies between numpy's (K, V) layout and the device's word-major (V, ldk) tables.
#pragma once
#include <hip/hip_runtime.h>

namespace pylda {

// Transposes of plain fp64 matrices (sstats export / import).
// in: rows x cols with leading dimension in_ld; out: cols x rows with leading dimension out_ld.
__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ in, int rows,
                                                        int cols, int in_ld, int out_ld,
                                                        double* __restrict__ out)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + j * 8, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + j * 8][tx] = in[(size_t)r * in_ld + c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + j * 8, r = r0 + tx;
        if (r < rows && c < cols) out[(size_t)c * out_ld + r] = tile[tx][ty + j * 8];
    }
}

}  // namespace pylda
