// Sizes and small parameter blocks that BOTH the kernels and the host-side planner need (the kernel headers define
// __global__ functions and can be included by one translation unit each; this header by all of them).
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#elif !defined(__host__)        // the planner alone under g++ (host_plan.cpp, sanitizer build)
#define __host__
#define __device__
#endif
#include <stddef.h>
#include <stdint.h>

namespace pylda {

// ---- estep_generic.h: LDS carve (all offsets multiples of 16 bytes, G17) ----
struct GenericLds {
    size_t tile, t, lt, gam, r, lognrm, cts, ids, red, scratch, total;
};

__host__ __device__ inline GenericLds generic_lds_layout(int K, int n_cap, int tile_stride,
                                                          int nthreads, bool tile_global)
{
    auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    GenericLds L;
    size_t off = 0;
    L.tile = off;   off = a16(off + (tile_global ? 0 : (size_t)n_cap * tile_stride * 8));
    L.t = off;      off = a16(off + (size_t)K * 8);
    L.lt = off;     off = a16(off + (size_t)K * 8);
    L.gam = off;    off = a16(off + (size_t)K * 8);
    L.r = off;      off = a16(off + (size_t)n_cap * 8);
    L.lognrm = off; off = a16(off + (size_t)n_cap * 8);
    L.cts = off;    off = a16(off + (size_t)n_cap * 8);
    L.ids = off;    off = a16(off + (size_t)n_cap * 4);
    // cross-group partials of pass 2: G x K doubles, G = nthreads / KL <= nthreads / min(K', nthreads)
    int kl = 1;
    while (kl < K && kl < nthreads) kl <<= 1;
    int groups = nthreads / kl;
    L.red = off;    off = a16(off + (size_t)groups * K * 8);
    L.scratch = off; off = a16(off + (size_t)(nthreads / 64) * 8);
    L.total = off;
    return L;
}

// ---- estep_logspace.h ----
__host__ __device__ inline size_t logspace_lds_bytes(int K)
{
    // psi[K], gam[K], gacc[4][K], scratch[4]
    return (size_t)(6 * K + 4) * 8 + 64;
}

// ---- estep_qfuse.h / estep_qfusek.h ----
#ifndef PYLDA_QF_SLOTS
#define PYLDA_QF_SLOTS 128
#endif
constexpr int kQfMaxSlots = PYLDA_QF_SLOTS;   // word slots per wavefront: documents up to 1024 distinct terms

// ---- estep_qgroup.h ----
constexpr int kQgMaxWords = 1024;           // distinct terms per document

// ---- mstep_kernels.h: parameters of alpha_newton_kernel (variational_bayes.py:277-324) ----
struct NewtonParams {
    int iterations;             // hyper_parameter_iteration (100)
    int maximum_decay;          // hyper_parameter_maximum_decay (10)
    double threshold;           // hyper_parameter_converge_threshold (1e-6)
    double decay_power[17];     // numpy.power(hyper_parameter_decay_factor, d), d = 0 .. maximum_decay (computed by the host's pow)
};

}  // namespace pylda
