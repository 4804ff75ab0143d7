// libpylda_hip.so - document kernels of the fused streaming families (qfuse, qfusek, qgroup): instantiations and launchers.
// (host side of the C ABI declared in include/pylda_hip.h; see host_internal.h for the map of the translation units)
#include "host_internal.h"
#include "estep_qfuse.h"
#include "estep_qfusek.h"
#include "estep_qgroup.h"

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
    return ctx->ldk == 512 ? launch_qfuse_np<4, PYLDA_QF4_RWL, PYLDA_QF4_TWL>(ctx, p, L) : launch_qfuse_np<3, 8, 4>(ctx, p, L);
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

template <int TL>
int launch_qgroup_tl(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    auto kern = estep_qgroup_kernel<TL, 8>;
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(512), QgroupLds<TL>::total, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

int launch_qgroup(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    if (ctx->ldk == 64) return launch_qgroup_tl<8>(ctx, p, L);
    if (ctx->ldk == 128) return launch_qgroup_tl<16>(ctx, p, L);
    if (ctx->ldk == 256) return launch_qgroup_tl<32>(ctx, p, L);
    return fail(ctx, PYLDA_ERR_STATE, "no group-fused streaming kernel for table stride %d", ctx->ldk);
}

}  // namespace pylda_host

