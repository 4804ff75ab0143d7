// Cost of the pacing rendezvous of sstats_sweep.h: 256 workgroups (one per CU) of 1024 / 768 threads meet N times
// back to back.  hipcc --offload-arch=gfx950 -O3 -o tools/rendezvous_bench tools/rendezvous_bench.hip
//   variant 0: one counter, every workgroup's lane 0 polls it (what the sweep does)
//   variant 1: XCD-hierarchical - a counter per XCD (workgroup b sits on XCD b % 8), the last arriver of an XCD bumps the
//              top counter, pollers watch their XCD's generation word
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(1024) void flat_kernel(unsigned* counter, int rounds, int sleep)
{
    for (int r = 1; r <= rounds; ++r) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r * gridDim.x)
                if (sleep) __builtin_amdgcn_s_sleep(4);
        }
        __syncthreads();
    }
}

// words: [0..7] per-XCD arrival counters (64 bytes apart: index * 16), [8] top counter, [9..16] per-XCD generation
__global__ __launch_bounds__(1024) void xcd_kernel(unsigned* words, int rounds, int sleep)
{
    const int xcd = blockIdx.x & 7;
    const unsigned per_xcd = (gridDim.x + 7 - xcd) / 8;
    unsigned* mine = words + 16 * xcd;
    unsigned* top = words + 16 * 8;
    unsigned* gen = words + 16 * (9 + xcd);
    for (int r = 1; r <= rounds; ++r) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned before = __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before + 1 == (unsigned)r * per_xcd) {          // last of this XCD: report upstairs, wait for all XCDs, release mine
                __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r * 8)
                    if (sleep) __builtin_amdgcn_s_sleep(2);
                __hip_atomic_store(gen, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r)
                    if (sleep) __builtin_amdgcn_s_sleep(4);
            }
        }
        __syncthreads();
    }
}

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 1000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned* d = nullptr;
    hipMalloc((void**)&d, 4096);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int threads : {1024, 768, 256})
        for (int variant = 0; variant < 2; ++variant)
            for (int sleep = 0; sleep < 2; ++sleep) {
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipMemset(d, 0, 4096);
                    hipEventRecord(a);
                    if (variant == 0) hipLaunchKernelGGL(flat_kernel, dim3(cus), dim3(threads), 0, 0, d, rounds, sleep);
                    else hipLaunchKernelGGL(xcd_kernel, dim3(cus), dim3(threads), 0, 0, d, rounds, sleep);
                    hipEventRecord(b);
                    hipEventSynchronize(b);
                    float ms = 0;
                    hipEventElapsedTime(&ms, a, b);
                    if (ms < best) best = ms;
                }
                printf("%4d workgroups x %4d threads, %s, %s: %.2f us per rendezvous\n", cus, threads,
                       variant ? "per-XCD counters" : "one counter     ", sleep ? "s_sleep" : "spin   ", best * 1e3f / rounds);
            }
    return 0;
}
