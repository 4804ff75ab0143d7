// libpylda_hip.so - document kernels of the generic, slab and quilt families: instantiations and launchers.
// (host side of the C ABI declared in include/pylda_hip.h; see host_internal.h for the map of the translation units)
#include "host_internal.h"
#include "estep_generic.h"
#include "estep_slab.h"
#include "estep_quilt.h"

namespace pylda_host {

template <int NT, int MODE>
int launch_generic(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    auto kern = estep_generic_kernel<NT, MODE>;
    if (L.lds_bytes > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)L.lds_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(NT), L.lds_bytes, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

int launch_generic_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    switch (L.variant) {
    case kGeneric64: return launch_generic<64, 0>(ctx, p, L);
    case kGeneric256: return launch_generic<256, 0>(ctx, p, L);
    case kGeneric512: return launch_generic<512, 0>(ctx, p, L);
    case kGenericHuge: return launch_generic<256, 2>(ctx, p, L);
    default: return launch_generic<256, 1>(ctx, p, L);
    }
}

template <int W, int RK, int RN>
int launch_slab(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    auto kern = estep_slab_kernel<W, RK, RN>;
    const size_t lds = SlabLds<W, RK, RN>::total;
    if (lds > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(kWave * W), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

int launch_slab_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    const int W = ctx->ldk / L.rk;
#define SLAB_CASE(w_, rk_, rn_) \
    if (W == w_ && L.rk == rk_ && L.rn == rn_) return launch_slab<w_, rk_, rn_>(ctx, p, L);
#define SLAB_RN32(w) SLAB_CASE(w, 32, 1) SLAB_CASE(w, 32, 2)
#define SLAB_RN16(w) SLAB_CASE(w, 16, 1) SLAB_CASE(w, 16, 2) SLAB_CASE(w, 16, 3) SLAB_CASE(w, 16, 4)
    SLAB_RN32(1) SLAB_RN32(2) SLAB_RN32(4)
    SLAB_RN16(1) SLAB_RN16(2) SLAB_RN16(4) SLAB_RN16(8)
    SLAB_CASE(1, 16, 6) SLAB_CASE(2, 16, 6) SLAB_CASE(4, 16, 6)
#undef SLAB_RN16
#undef SLAB_RN32
#undef SLAB_CASE
    return fail(ctx, PYLDA_ERR_STATE, "no slab kernel for W=%d RK=%d RN=%d", W, L.rk, L.rn);
}

template <int W, int RK>
int launch_slab_uber(pylda_ctx* ctx, const EstepParams& p, const pylda_corpus* c, int from)
{
    SlabUberClasses cls;
    memset(&cls, 0, sizeof cls);
    cls.n = (int)c->plan.size() - from;
    size_t lds = 0;
    int64_t docs = 0;
    for (int i = 0; i < cls.n; ++i) {
        const Launch& L = c->plan[(size_t)(from + i)];
        cls.first[i] = (int)(L.first - c->plan[(size_t)from].first);
        cls.rn[i] = L.rn;
        docs += L.count;
        // (the largest class comes first: documents are scheduled longest first)
        const size_t need = RK == 32 ? (L.rn == 1 ? SlabLds<W, RK, 1>::total : SlabLds<W, RK, 2>::total)
                          : L.rn == 1 ? SlabLds<W, RK, 1>::total : L.rn == 2 ? SlabLds<W, RK, 2>::total
                          : L.rn == 3 ? SlabLds<W, RK, 3>::total : L.rn == 4 ? SlabLds<W, RK, 4>::total : SlabLds<W, RK, 6>::total;
        lds = std::max(lds, need);
    }
    cls.first[cls.n] = (int)docs;
    bool has_long = false;
    for (int i = 0; i < cls.n; ++i) has_long = has_long || cls.rn[i] > 4;
    auto kern = (RK == 16 && has_long) ? estep_slab_uber_kernel<W, RK, true> : estep_slab_uber_kernel<W, RK, false>;
    if (lds > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)docs), dim3(kWave * W), lds, ctx->stream, p, cls);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

int launch_slab_uber_any(pylda_ctx* ctx, const EstepParams& p, const pylda_corpus* c, int from)
{
    const int rk = c->plan[(size_t)from].rk, W = ctx->ldk / rk;
#define UBER_CASE(w_, rk_) if (W == w_ && rk == rk_) return launch_slab_uber<w_, rk_>(ctx, p, c, from);
    UBER_CASE(1, 32) UBER_CASE(2, 32) UBER_CASE(4, 32)
    UBER_CASE(1, 16) UBER_CASE(2, 16) UBER_CASE(4, 16)
#undef UBER_CASE
    return fail(ctx, PYLDA_ERR_STATE, "no slab uber kernel for W=%d RK=%d", W, rk);
}

template <int W, int KRL, int RWL>
int launch_quilt(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    auto kern = estep_quilt_kernel<W, KRL, RWL>;
    const size_t lds = QuiltLds<W, KRL, RWL>::total;
    if (lds > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)L.count), dim3(kWave * W), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PYLDA_OK;
}

int launch_quilt_any(pylda_ctx* ctx, const EstepParams& p, const Launch& L)
{
    const int KRL = ctx->ldk / 16;
#define QUILT_CASE(w_, krl_, rwl_) \
    if (KRL == krl_ && L.rn == w_ * 100 + rwl_) return launch_quilt<w_, krl_, rwl_>(ctx, p, L);
    QUILT_CASE(8, 4, 2) QUILT_CASE(8, 4, 4) QUILT_CASE(8, 4, 8) QUILT_CASE(8, 8, 2) QUILT_CASE(8, 8, 4) QUILT_CASE(8, 8, 8)
    QUILT_CASE(8, 4, 6) QUILT_CASE(8, 4, 7) QUILT_CASE(8, 8, 6) QUILT_CASE(8, 8, 7)
    QUILT_CASE(12, 4, 4) QUILT_CASE(12, 8, 4)
#undef QUILT_CASE
    return fail(ctx, PYLDA_ERR_STATE, "no quilt kernel for KRL=%d RWL=%d", KRL, L.rn);
}

}  // namespace pylda_host

