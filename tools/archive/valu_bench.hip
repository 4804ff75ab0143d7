// Micro-benchmark (not part of the product): issue rate of fp64 FMAs on one SIMD of MI355X as a
// function of (a) the number of independent accumulator chains in the instruction stream,
// (b) the instruction form (plain v_fmac_f64 vs v_fmac_f64_dpp with a row_newbcast operand),
// (c) wavefronts per SIMD (1, 2, 4).  Timed with s_memtime inside the kernel (ticks ~ core cycles)
// and with HIP events outside.  These numbers decide how the E-step kernels order their FMAs.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_bench tools/valu_bench.hip && ./valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#ifndef KREPS
#define KREPS 512
#endif
constexpr int kReps = KREPS;

// CHAINS independent accumulators, each instruction an FMA on the next accumulator (round robin)
template <int CHAINS, int MODE>
__global__ __launch_bounds__(256) void fma_chains(double* out, long long* ticks, double x0)
{
    double acc[CHAINS];
    double b[8];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) acc[i] = x0 + i + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = x0 * (i + 1) * 1e-3;
    double r = x0 * 0.999;
    asm volatile("" : "+v"(r));
    const long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < kReps; ++it) {
#pragma unroll
        for (int u = 0; u < 64 / CHAINS; ++u) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (MODE == 0) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[i]) : "v"(r), "v"(b[(i + u) & 7]));
                else if (MODE == 1)
                    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(r), "v"(b[(i + u) & 7]));
                else if (MODE == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[i]) : "v"(r));
                else if (MODE == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[i]) : "v"(r));
                else if (MODE == 4) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(r), "v"(b[(i + u) & 7]));
                else if (MODE == 5) asm volatile("v_rcp_f64_e32 %0, %0" : "+v"(acc[i]));
                else if (MODE == 6) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(((int*)&acc[i])[0]) : "v"(((int*)&r)[0]));
                else if (MODE == 7) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(((float*)&acc[i])[0]) : "v"(((float*)&r)[0]), "v"(((float*)&b[(i + u) & 7])[0]));
            }
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += acc[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int CHAINS, int MODE>
void run(const char* name, int blocks_per_cu, int threads)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const int blocks = ncu * blocks_per_cu;
    double* out;
    long long* ticks;
    CK(hipMalloc(&out, (size_t)blocks * threads * 8));
    CK(hipMalloc(&ticks, (size_t)blocks * threads / 64 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((fma_chains<CHAINS, MODE>), dim3(blocks), dim3(threads), 0, 0, out, ticks, 1.0000001);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((fma_chains<CHAINS, MODE>), dim3(blocks), dim3(threads), 0, 0, out, ticks, 1.0000001);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int nw = blocks * threads / 64;
    long long* h = (long long*)malloc((size_t)nw * 8);
    CK(hipMemcpy(h, ticks, (size_t)nw * 8, hipMemcpyDeviceToHost));
    double mean = 0;
    for (int i = 0; i < nw; ++i) mean += (double)h[i];
    mean /= nw;
    const double instr = (double)kReps * (64 / CHAINS) * CHAINS;
    const int waves_per_simd = blocks_per_cu * threads / 64 / 4;
    printf("%-14s chains %2d  waves/SIMD %d : %6.2f ticks/instr/wave  => %5.2f ticks per instr per SIMD;  wall %.3f ms (%.2f ns per instr per SIMD)\n",
           name, CHAINS, waves_per_simd, mean / instr, mean / instr / waves_per_simd, ms,
           ms * 1e6 / (instr * waves_per_simd));
    free(h);
    CK(hipFree(out));
    CK(hipFree(ticks));
}

int main(int argc, char** argv)
{
    if (argc > 1) {     // calibration: long kernels, ticks against wall time
        run<8, 0>("v_fmac_f64", 1, 256); run<8, 0>("v_fmac_f64", 2, 256); run<8, 0>("v_fmac_f64", 3, 256);
        run<8, 0>("v_fmac_f64", 4, 256); run<8, 0>("v_fmac_f64", 6, 256); run<8, 0>("v_fmac_f64", 8, 256);
        run<2, 0>("v_fmac_f64", 3, 256); run<2, 0>("v_fmac_f64", 8, 256);
        run<8, 6>("v_mov_b32_dpp", 4, 256); run<8, 6>("v_mov_b32_dpp", 8, 256);
        return 0;
    }
#define SWEEP(MODE, NAME)                                                     \
    run<1, MODE>(NAME, 1, 256); run<2, MODE>(NAME, 1, 256); run<3, MODE>(NAME, 1, 256); run<4, MODE>(NAME, 1, 256); \
    run<8, MODE>(NAME, 1, 256); run<16, MODE>(NAME, 1, 256);                  \
    run<1, MODE>(NAME, 2, 256); run<2, MODE>(NAME, 2, 256); run<4, MODE>(NAME, 2, 256); run<8, MODE>(NAME, 2, 256); \
    run<2, MODE>(NAME, 4, 256); run<8, MODE>(NAME, 4, 256);
    SWEEP(0, "v_fmac_f64")
    SWEEP(1, "v_fmac_f64_dpp")
    SWEEP(4, "v_fma_f64")
    SWEEP(2, "v_mul_f64")
    SWEEP(3, "v_add_f64")
    run<8, 5>("v_rcp_f64", 1, 256); run<8, 5>("v_rcp_f64", 2, 256);
    run<8, 6>("v_mov_b32_dpp", 1, 256); run<8, 6>("v_mov_b32_dpp", 2, 256);
    run<8, 7>("v_fmac_f32", 1, 256); run<8, 7>("v_fmac_f32", 2, 256); run<1, 7>("v_fmac_f32", 1, 256);
    return 0;
}
