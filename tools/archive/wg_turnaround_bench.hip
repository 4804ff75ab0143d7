// What a workgroup costs before its first and after its last instruction: the dense quad kernels hold one document per
// CU (8 wavefronts x 256 VGPRs, > 80 KB of LDS), so nothing overlaps a workgroup's launch with its predecessor's work.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wgbench tools/archive/wg_turnaround_bench.hip && /tmp/wgbench
// Workgroups of 512 threads that pin the whole register file of a CU (256 VGPRs: amdgpu_num_vgpr) and 100 KB of LDS and
// then do NOTHING but wait `hold` ticks of s_memtime: (time of N workgroups / (N / CUs)) - hold = the turnaround.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(256))) void hold_kernel(long long hold, int* sink)
{
    extern __shared__ char smem[];
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < hold) __builtin_amdgcn_s_sleep(2);
    if (hold < 0) sink[threadIdx.x] = smem[threadIdx.x];
}

int main()
{
    int* sink;
    hipMalloc(&sink, 4096);
    hipFuncSetAttribute(reinterpret_cast<const void*>(hold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int n = cus * 2000;
    for (long long hold : {0ll, 2000ll, 10000ll, 50000ll}) {       // ticks of the 100 MHz counter: 10 ns each
        hipLaunchKernelGGL(hold_kernel, dim3(cus * 10), dim3(512), 100 * 1024, 0, hold, sink);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(hold_kernel, dim3(n), dim3(512), 100 * 1024, 0, hold, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double per_wg_us = ms * 1e3 / (n / cus);
        printf("hold %6lld ticks: %d workgroups on %d CUs in %.3f ms = %.3f us per workgroup and CU\n", hold, n, cus, ms, per_wg_us);
    }
    return 0;
}
