// libpylda_hip.so - document kernels of the streaming families (qfuse, qfusek, qstream, qhybrid, qwide): instantiations and launchers.
// (host side of the C ABI declared in include/pylda_hip.h; see host_internal.h for the map of the translation units)
#include "host_internal.h"
#include "estep_qfuse.h"
#include "estep_qfusek.h"
#include "estep_qstream.h"
#include "estep_qhybrid.h"
#include "estep_qwide.h"

namespace pylda_host {

template <int NP, int RWL, int TWL>
int launch_qfuse_np(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    auto kern = estep_qfuse_kernel<NP, RWL, TWL>;
    const size_t lds = QfuseLds<NP, TWL>::total;
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(512), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

int launch_qfuse(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
#ifndef PYLDA_QF4_RWL
#define PYLDA_QF4_RWL 6
#endif
#ifndef PYLDA_QF4_TWL
#define PYLDA_QF4_TWL 2
#endif
    return ctx->ldk == 512 ? launch_qfuse_np<4, PYLDA_QF4_RWL, PYLDA_QF4_TWL>(ctx, p, L) : launch_qfuse_np<3, 8, 2>(ctx, p, L);
}

template <int NP>
int launch_qfusek_np(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    auto kern = estep_qfusek_kernel<NP>;
    const size_t lds = QfusekLds<NP>::total;
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(512), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

int launch_qfusek(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    switch (ctx->ldk / 128) {
    case 5: return launch_qfusek_np<5>(ctx, p, L);
    case 6: return launch_qfusek_np<6>(ctx, p, L);
    case 7: return launch_qfusek_np<7>(ctx, p, L);
    case 8: return launch_qfusek_np<8>(ctx, p, L);
    }
    return fail(ctx, PYLDA_ERR_STATE, "no fused streaming kernel for table stride %d", ctx->ldk);
}

template <int KRL>
int launch_qstream(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    auto kern = estep_qstream_kernel<8, KRL>;
    const size_t lds = QstreamLds<8, KRL>::total;
    if (lds > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(kWave * 8), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

template <int KRL>
int launch_qhybrid(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    using Lds = QhybridLds<8, KRL, 4>;
    auto kern = estep_qhybrid_kernel<8, KRL, 4>;
    const size_t limit = 160 * 1024;
    const int rows_per_wave = std::min(kQhMaxTail, Lds::rows_that_fit(limit) / 8);
    const size_t lds = Lds::fixed_total + (size_t)8 * rows_per_wave * Lds::kRowDoubles * 8;
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(kWave * 8), lds, ctx->stream, p, rows_per_wave);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

template <int JJ, bool MULTI>
int launch_qwide(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    using Lds = QwideLds<8, JJ>;
    auto kern = estep_qwide_kernel<8, JJ, MULTI>;
    const size_t limit = 160 * 1024;
    const int rows_per_wave = std::min(kQwMaxTail, Lds::rows_that_fit(limit) / 8) & ~1;
    const size_t lds = Lds::fixed_total + (size_t)8 * rows_per_wave * Lds::kRowDoubles * 8;
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(kWave * 8), lds, ctx->stream, p, rows_per_wave);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

int launch_qwide_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    const bool multi = L.rn != 1;       // more than one round of tail steps per wavefront
    switch (ctx->ldk / 64) {
    case 2: return multi ? launch_qwide<2, true>(ctx, p, L) : launch_qwide<2, false>(ctx, p, L);
    case 3: return multi ? launch_qwide<3, true>(ctx, p, L) : launch_qwide<3, false>(ctx, p, L);
    case 4: return multi ? launch_qwide<4, true>(ctx, p, L) : launch_qwide<4, false>(ctx, p, L);
    }
    return fail(ctx, PYLDA_ERR_STATE, "no wide tiered kernel for table stride %d", ctx->ldk);
}

int launch_qhybrid_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    switch (ctx->ldk / 16) {
    case 4: return launch_qhybrid<4>(ctx, p, L);
    case 8: return launch_qhybrid<8>(ctx, p, L);
    case 12: return launch_qhybrid<12>(ctx, p, L);
    case 16: return launch_qhybrid<16>(ctx, p, L);
    }
    return fail(ctx, PYLDA_ERR_STATE, "no hybrid kernel for table stride %d", ctx->ldk);
}

int launch_qstream_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    switch (ctx->ldk / 16) {
    case 4: return launch_qstream<4>(ctx, p, L);
    case 8: return launch_qstream<8>(ctx, p, L);
    case 12: return launch_qstream<12>(ctx, p, L);
    case 16: return launch_qstream<16>(ctx, p, L);
    case 20: return launch_qstream<20>(ctx, p, L);
    case 24: return launch_qstream<24>(ctx, p, L);
    case 28: return launch_qstream<28>(ctx, p, L);
    case 32: return launch_qstream<32>(ctx, p, L);
    }
    return fail(ctx, PYLDA_ERR_STATE, "no streaming kernel for table stride %d", ctx->ldk);
}

}  // namespace pylda_host

