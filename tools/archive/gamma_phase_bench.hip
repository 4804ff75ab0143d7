// Micro-benchmark (not part of the product): ticks per evaluation of t = exp(psi(x) - c) as the gamma phase
// of the register kernels runs it - ONE dependent evaluation per thread at a time, one wavefront per SIMD
// (the document's other wavefronts wait at a barrier) - for the forms in csrc/special_device.h.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Ipylda_amd/csrc -o tools/gamma_phase_bench tools/gamma_phase_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "special_device.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using namespace pylda;

constexpr int kReps = 2000;

template <int FORM>
__global__ __launch_bounds__(256) void bench(double* out, long long* ticks, double x0, double c0)
{
    double x = x0 + 1e-3 * threadIdx.x;
    const double c = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(c0)), __builtin_amdgcn_readfirstlane(__double2loint(c0)));
    double acc = 0.0;
    ExpDigammaLevelsA ka;
    ExpDigammaLevelsB kb;
    if constexpr (FORM == 3) {          // both coefficient tables resident (the kernels request them ahead of a barrier)
        ka.load();
        kb.load();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < kReps; ++it) {
        double t;
        if constexpr (FORM == 0) {
            ExpDigammaScalarCoef k;
            k.load();
            t = exp_digamma_minus_with(x, c, k);
        } else if constexpr (FORM == 1) {
            t = exp_digamma_minus(x, c);
        } else if constexpr (FORM == 2) {
            ExpDigammaLevelsA k;
            k.load();
            t = exp_digamma_minus_levels(x, c, k);
        } else {
            t = exp_digamma_minus_levels<true>(x, c, ka, &kb);
        }
        acc += t;
        x = fma(t, 3.0, 0.25);      // the next argument depends on the result, as gamma' = alpha + t * s does
    }
    asm volatile("s_nop 0" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) ticks[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int FORM>
void run(const char* name, int blocks_per_cu)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * blocks_per_cu, threads = 256;
    double* out;
    long long* ticks;
    CK(hipMalloc(&out, (size_t)blocks * threads * 8));
    CK(hipMalloc(&ticks, (size_t)blocks * threads / 64 * 8));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((bench<FORM>), dim3(blocks), dim3(threads), 0, 0, out, ticks, 0.3, 5.0);
        CK(hipDeviceSynchronize());
    }
    const int nw = blocks * threads / 64;
    long long* h = (long long*)malloc((size_t)nw * 8);
    CK(hipMemcpy(h, ticks, (size_t)nw * 8, hipMemcpyDeviceToHost));
    double mean = 0;
    for (int i = 0; i < nw; ++i) mean += (double)h[i];
    printf("%-44s wavefronts/SIMD %d : %7.1f ticks per evaluation\n", name, blocks_per_cu, mean / nw / kReps);
    free(h);
    CK(hipFree(out));
    CK(hipFree(ticks));
}

int main()
{
    run<0>("exp_digamma_minus_with (scalar table)", 1);
    run<1>("exp_digamma_minus (literals)", 1);
    run<2>("exp_digamma_minus_levels (tables fetched in place)", 1);
    run<3>("exp_digamma_minus_levels (tables resident)", 1);
    run<0>("exp_digamma_minus_with (scalar table)", 2);
    run<2>("exp_digamma_minus_levels (tables fetched in place)", 2);
    run<3>("exp_digamma_minus_levels (tables resident)", 2);
    return 0;
}
