// libpylda_hip.so - the register + LDS tile document kernel (estep_quad.h): instantiations and launcher.
// (host side of the C ABI declared in include/pylda_hip.h; see host_internal.h for the map of the translation units)
#include "host_internal.h"
#include "estep_quad.h"

namespace pylda_host {

#ifndef PYLDA_QUAD_HANDOFF
#define PYLDA_QUAD_HANDOFF true           // (launch_quad_dense.hip: this file again with false - the kernels of corpora that hand nothing over)
#define PYLDA_QUAD_LAUNCHER launch_quad_any
#endif

namespace {     // (internal linkage: this template exists twice in the library, once per value of PYLDA_QUAD_HANDOFF)

template <int TL, int RWL, int TWL, int SWL = 0>
int launch_quad(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    auto kern = estep_quad_kernel<TL, RWL, TWL, SWL, PYLDA_QUAD_HANDOFF>;
    const size_t lds = QuadLds<TL, RWL, TWL>::total + (size_t)ctx->lds_pad;
    if (lds > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(kWave * (TL / 4)), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

}  // namespace

int PYLDA_QUAD_LAUNCHER(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    if (PYLDA_QUAD_HANDOFF && p.handoff_live <= 0) return launch_quad_dense_any(ctx, p, L);
    switch (L.rn) {
    case 160800: return launch_quad<16, 8, 0>(ctx, p, L);
    case 161000: return launch_quad<16, 10, 0>(ctx, p, L);
    case 161001: return launch_quad<16, 10, 1>(ctx, p, L);
    case 161002: return launch_quad<16, 10, 2>(ctx, p, L);
    case 161003: return launch_quad<16, 10, 3>(ctx, p, L);
    case 161004: return launch_quad<16, 10, 4>(ctx, p, L);
    case 320800: return launch_quad<32, 8, 0>(ctx, p, L);
    case 321000: return launch_quad<32, 10, 0>(ctx, p, L);
    case 321001: return launch_quad<32, 10, 1>(ctx, p, L);
    case 321002: return launch_quad<32, 10, 2>(ctx, p, L);
    case 321003: return launch_quad<32, 10, 3>(ctx, p, L);
    case 321004: return launch_quad<32, 10, 4>(ctx, p, L);
    // ... + streamed slots (code + 1000000 * SWL): 225-240 and 241-256 terms
    case 2160904: return launch_quad<16, 9, 4, 2>(ctx, p, L);
    case 3160904: return launch_quad<16, 9, 4, 3>(ctx, p, L);
    case 3320804: return launch_quad<32, 8, 4, 3>(ctx, p, L);
    case 4320804: return launch_quad<32, 8, 4, 4>(ctx, p, L);
    }
    return fail(ctx, PYLDA_ERR_STATE, "no quad kernel for geometry %d", L.rn);
}

}  // namespace pylda_host

