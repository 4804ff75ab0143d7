// libpylda_hip.so - contexts: handles, options, the model tables in and out, host memory, stream marks, RCCL glue, test hooks.
// (host side of the C ABI declared in include/pylda_hip.h; see host_internal.h for the map of the translation units)
#include "host_internal.h"
#include "transpose_kernel.h"
#include "special_device.h"

namespace pylda_host {

std::string g_create_error;

int fail(pylda_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

hipEvent_t take_event(pylda_ctx* ctx)
{
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void drain_events(pylda_ctx* ctx)
{
    for (auto& br : ctx->pending_events) {
        float ms = 0.f;
        if (hipEventSynchronize(br.b) == hipSuccess && hipEventElapsedTime(&ms, br.a, br.b) == hipSuccess) {
            if (br.slot == -1) ctx->doc_kernel_ms += ms;
            else if (br.slot == -2) ctx->sstats_kernel_ms += ms;
            else if ((size_t)br.slot < ctx->class_ms.size()) ctx->class_ms[(size_t)br.slot] += ms;
        }
        ctx->event_pool.push_back(br.a);
        ctx->event_pool.push_back(br.b);
    }
    ctx->pending_events.clear();
}

}  // namespace pylda_host

extern "C" {

const char* pylda_version(void) { return "pylda_hip 0.3 (gfx950, abi 3)"; }
int pylda_abi_version(void) { return PYLDA_ABI_VERSION; }

int pylda_device_count(int* count)
{
    if (!count) return PYLDA_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return PYLDA_OK;
}

const char* pylda_last_error(const pylda_ctx* ctx)
{
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int pylda_create(int device, int K, int V, pylda_ctx** out)
{
    if (!out) return fail(nullptr, PYLDA_ERR_INVALID, "pylda_create: out is NULL");
    *out = nullptr;
    if (K < 1 || V < 1) return fail(nullptr, PYLDA_ERR_INVALID, "pylda_create: K=%d V=%d", K, V);
    if ((int64_t)K * V > ((int64_t)1 << 40))
        return fail(nullptr, PYLDA_ERR_INVALID, "pylda_create: K*V too large");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, PYLDA_ERR_HIP,
                    "pylda_create: no HIP device visible; this library has no CPU fallback");
    if (device < 0 || device >= ndev)
        return fail(nullptr, PYLDA_ERR_INVALID, "pylda_create: device %d of %d", device, ndev);
    pylda_ctx* ctx = new (std::nothrow) pylda_ctx;
    if (!ctx) return fail(nullptr, PYLDA_ERR_OOM, "pylda_create: host allocation failed");
    ctx->device = device;
    ctx->K = K;
    ctx->V = V;
    ctx->ldk = table_stride_for(K);
    auto bail = [&](int code) {
        g_create_error = ctx->err;
        pylda_destroy(ctx);
        return code;
    };
#define CREATE_TRY(expr)                     \
    do {                                     \
        int rc_ = (expr);                    \
        if (rc_ != PYLDA_OK) return bail(rc_); \
    } while (0)
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return (int)PYLDA_OK;
        return fail(ctx, e == hipErrorOutOfMemory ? PYLDA_ERR_OOM : PYLDA_ERR_HIP, "%s: %s", what,
                    hipGetErrorString(e));
    };
    CREATE_TRY(hip_ok(hipSetDevice(device), "hipSetDevice"));
    hipDeviceProp_t prop;
    CREATE_TRY(hip_ok(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties"));
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    size_t lds = prop.maxSharedMemoryPerMultiProcessor;
    if (lds < 64 * 1024) lds = 64 * 1024;
    if (lds > 160 * 1024) lds = 160 * 1024;
    ctx->lds_limit = lds;
    CREATE_TRY(hip_ok(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking),
                      "hipStreamCreate"));
    ctx->stream = ctx->own_stream;
    for (int i = 0; i < pylda_ctx::kAux; ++i) {
        CREATE_TRY(hip_ok(hipStreamCreateWithFlags(&ctx->aux_stream[i], hipStreamNonBlocking), "hipStreamCreate"));
        CREATE_TRY(hip_ok(hipEventCreateWithFlags(&ctx->join_event[i], hipEventDisableTiming), "hipEventCreate"));
    }
    CREATE_TRY(hip_ok(hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming), "hipEventCreate"));
    const size_t kv = (size_t)K * V, wk = (size_t)V * ctx->ldk;
    CREATE_TRY(dev_alloc(ctx, &ctx->d_eta, kv));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_elog, wk));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_expElog, wk));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_expElog_elog, wk));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_sstats, wk));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_kv_scratch, kv));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_shift, (size_t)V));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_beta, (size_t)V));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_psi_rowsum, (size_t)K));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_topic_lse, (size_t)K));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_alpha, (size_t)K));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_alpha_sgn, (size_t)K));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_small, (size_t)(4 * K + 16)));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_partial, (size_t)1024 * K));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_outer, (size_t)(3 * K + 8)));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_newton_work, (size_t)4 * K));
    CREATE_TRY(dev_alloc(ctx, &ctx->d_work, (size_t)6));
    CREATE_TRY(hip_ok(hipMemsetAsync(ctx->d_work, 0, 6 * sizeof(double), ctx->stream), "hipMemsetAsync"));
    CREATE_TRY(hip_ok(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pin), (size_t)(5 * K + 8) * sizeof(double), hipHostMallocDefault),
                      "hipHostMalloc"));
    for (int i = 0; i < 2; ++i)
        CREATE_TRY(hip_ok(hipEventCreateWithFlags(&ctx->alpha_event[i], hipEventDisableTiming), "hipEventCreate"));
    CREATE_TRY(hip_ok(hipMemsetAsync(ctx->d_sstats, 0, wk * sizeof(double), ctx->stream),
                      "hipMemsetAsync"));
#undef CREATE_TRY
    *out = ctx;
    return PYLDA_OK;
}

void pylda_destroy(pylda_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->comm) pylda::comm_destroy(ctx->comm);
    dev_free(ctx->d_comm_small);
    drain_events(ctx);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    dev_free(ctx->d_eta); dev_free(ctx->d_elog); dev_free(ctx->d_expElog); dev_free(ctx->d_expElog_elog); dev_free(ctx->d_sstats);
    dev_free(ctx->d_kv_scratch); dev_free(ctx->d_shift); dev_free(ctx->d_beta);
    dev_free(ctx->d_psi_rowsum); dev_free(ctx->d_topic_lse); dev_free(ctx->d_alpha); dev_free(ctx->d_alpha_sgn);
    dev_free(ctx->d_small); dev_free(ctx->d_partial); dev_free(ctx->d_outer); dev_free(ctx->d_newton_work); dev_free(ctx->d_work); dev_free(ctx->d_eta_ckpt);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    for (int i = 0; i < 2; ++i)
        if (ctx->alpha_event[i]) (void)hipEventDestroy(ctx->alpha_event[i]);
    for (hipEvent_t e : ctx->mark_event)
        if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < pylda_ctx::kAux; ++i) {
        if (ctx->aux_stream[i]) { (void)hipStreamSynchronize(ctx->aux_stream[i]); (void)hipStreamDestroy(ctx->aux_stream[i]); }
        if (ctx->join_event[i]) (void)hipEventDestroy(ctx->join_event[i]);
    }
    if (ctx->fork_event) (void)hipEventDestroy(ctx->fork_event);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int pylda_set_stream(pylda_ctx* ctx, void* hip_stream)
{
    // As everywhere in HIP, a NULL handle is the device's default ("null") stream - which is
    // what torch.cuda.current_stream().cuda_stream reports for torch's default stream.
    if (!ctx) return PYLDA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return PYLDA_OK;
}

int pylda_use_own_stream(pylda_ctx* ctx)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = ctx->own_stream;
    return PYLDA_OK;
}

int pylda_synchronize(pylda_ctx* ctx)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PYLDA_OK;
}

int pylda_set_option(pylda_ctx* ctx, const char* name, int64_t value)
{
    if (!ctx || !name) return PYLDA_ERR_INVALID;
    if (!strcmp(name, "force_logspace")) ctx->force_logspace = value != 0;
    else if (!strcmp(name, "force_variant")) {
        if (value < -1 || value > kVariantLast || value == kRetired5 || value == kRetired7 || value == kRetired8)
            return fail(ctx, PYLDA_ERR_INVALID, "force_variant %lld is not a kernel variant", (long long)value);
        ctx->force_variant = (int)value;
        ctx->plan_epoch += 1;
    } else if (!strcmp(name, "gather_rows")) {
        ctx->gather_rows = (int)value;               // 0: 64-topic chunks, 1: whole rows, 2: whole rows, postings in bulk
    } else if (!strcmp(name, "gather_blocks")) {     // (takes effect for corpora created afterwards)
        if (value > 1 && value % 8) return fail(ctx, PYLDA_ERR_INVALID, "gather_blocks=%lld: a multiple of 8, or -1 / 0 / 1", (long long)value);
        ctx->gather_blocks = (int)value;
    } else if (!strcmp(name, "quilt_odd")) {
        ctx->quilt_odd = value != 0;
        ctx->plan_epoch += 1;
    } else if (!strcmp(name, "sweep_xcd")) {
        ctx->sweep_xcd = value != 0;
    } else if (!strcmp(name, "sweep_sub")) {
        ctx->sweep_sub = (int)std::min<int64_t>(64, std::max<int64_t>(1, value));
    } else if (!strcmp(name, "sweep_spin")) {
        ctx->sweep_spin = (int)std::max<int64_t>(0, value);
    } else if (!strcmp(name, "gather_sweep")) {      // (takes effect for corpora whose postings are built afterwards)
        ctx->gather_sweep = (int)std::min<int64_t>(2, std::max<int64_t>(0, value));
    } else if (!strcmp(name, "gather_round_mb")) {   // (takes effect for corpora whose postings are built afterwards)
        ctx->gather_round_mb = (int)std::max<int64_t>(0, value);
    } else if (!strcmp(name, "launch_order")) {
        ctx->launch_order = (int)value;
    } else if (!strcmp(name, "terms_overlap")) {
        ctx->terms_overlap = value != 0;
    } else if (!strcmp(name, "slab_uber")) {
        ctx->slab_uber = value != 0;
    } else if (!strcmp(name, "wide_postings")) {     // (takes effect for corpora whose postings are built afterwards)
        ctx->wide_postings = value != 0;
    } else if (!strcmp(name, "lds_pad")) {
        ctx->lds_pad = (int)value;
    } else if (!strcmp(name, "quad")) {
        ctx->quad = value != 0;
        ctx->plan_epoch += 1;
    } else if (!strcmp(name, "quad_stream")) {
        ctx->quad_stream = value != 0;
        ctx->plan_epoch += 1;
    } else if (!strcmp(name, "quilt12")) {
        ctx->quilt12 = value != 0;
        ctx->plan_epoch += 1;
    } else if (!strcmp(name, "doc_values")) {
        ctx->doc_values = value != 0;
    } else if (!strcmp(name, "gather_live")) {       // (takes effect for corpora whose postings are built afterwards)
        ctx->gather_live = value != 0;
    } else if (!strcmp(name, "compact")) {
        ctx->compact = value != 0;
    } else if (!strcmp(name, "compact_pair")) {
        ctx->compact_pair = value < 0 ? -1 : value != 0;
    } else if (!strcmp(name, "compact_stream")) {
        ctx->compact_stream = value != 0;
    } else if (!strcmp(name, "compact_phase")) {
        ctx->compact_phase = value != 0;
    } else if (!strcmp(name, "compact_cap")) {
        if (value < 0 || value > kLiveStride) return fail(ctx, PYLDA_ERR_INVALID, "compact_cap=%lld: 0 .. %d", (long long)value, kLiveStride);
        ctx->compact_cap = (int)value;
    } else if (!strcmp(name, "compact_guard_fail")) {
        ctx->compact_guard_fail = value != 0;
    } else
        return fail(ctx, PYLDA_ERR_INVALID, "unknown option '%s'", name);
    return PYLDA_OK;
}

int pylda_set_eta(pylda_ctx* ctx, const double* eta_kv)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!eta_kv) return fail(ctx, PYLDA_ERR_INVALID, "set_eta: NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_eta, eta_kv, (size_t)ctx->K * ctx->V * sizeof(double),
                                hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // host buffer is not retained
    ctx->have_eta = true;
    return PYLDA_OK;
}

int pylda_get_eta(pylda_ctx* ctx, double* eta_kv)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!eta_kv) return fail(ctx, PYLDA_ERR_INVALID, "get_eta: NULL");
    if (!ctx->have_eta) return fail(ctx, PYLDA_ERR_STATE, "get_eta: eta was never set");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(eta_kv, ctx->d_eta, (size_t)ctx->K * ctx->V * sizeof(double),
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PYLDA_OK;
}

int pylda_set_alpha(pylda_ctx* ctx, const double* alpha_k)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!alpha_k) return fail(ctx, PYLDA_ERR_INVALID, "set_alpha: NULL");
    for (int k = 0; k < ctx->K; ++k)
        if (!(alpha_k[k] > 0.0) || !std::isfinite(alpha_k[k]))
            return fail(ctx, PYLDA_ERR_INVALID, "set_alpha: alpha[%d]=%g is not positive", k, alpha_k[k]);
    // (the device already holds exactly these values: after an alpha update on the device - pylda_outer_fetch - the
    //  host hands back what it was handed)
    if (ctx->have_alpha && ctx->h_alpha.size() == (size_t)ctx->K &&
        memcmp(ctx->h_alpha.data(), alpha_k, (size_t)ctx->K * sizeof(double)) == 0)
        return PYLDA_OK;
    ctx->h_alpha.assign(alpha_k, alpha_k + ctx->K);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // no stream wait: the values go through one of two pinned slots (the copy is ordered on the stream behind the
    // kernels still reading the previous alpha); a slot is reused only when its last copy has left it
    const int slot = ctx->alpha_slot;
    ctx->alpha_slot ^= 1;
    if (ctx->alpha_event_used[slot]) HIP_TRY(ctx, hipEventSynchronize(ctx->alpha_event[slot]));
    double* pin = ctx->h_pin + (size_t)slot * ctx->K;
    memcpy(pin, alpha_k, (size_t)ctx->K * sizeof(double));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_alpha, pin, (size_t)ctx->K * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->alpha_event[slot], ctx->stream));
    ctx->alpha_event_used[slot] = true;
    ctx->have_alpha = true;
    return PYLDA_OK;
}

int pylda_get_sstats(pylda_ctx* ctx, double* sstats_kv)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!sstats_kv) return fail(ctx, PYLDA_ERR_INVALID, "get_sstats: NULL");
    if (!ctx->have_sstats) return fail(ctx, PYLDA_ERR_STATE, "get_sstats: no training-mode E-step has run");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int K = ctx->K, V = ctx->V;
    // device layout is (V, K); hand back numpy's (K, V)
    hipLaunchKernelGGL(transpose_kernel, dim3((K + 31) / 32, (V + 31) / 32), dim3(256), 0, ctx->stream,
                       ctx->d_sstats, V, K, ctx->ldk, V, ctx->d_kv_scratch);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(sstats_kv, ctx->d_kv_scratch, (size_t)K * V * sizeof(double),
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PYLDA_OK;
}

int pylda_set_sstats(pylda_ctx* ctx, const double* sstats_kv)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!sstats_kv) return fail(ctx, PYLDA_ERR_INVALID, "set_sstats: NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int K = ctx->K, V = ctx->V;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_kv_scratch, sstats_kv, (size_t)K * V * sizeof(double),
                                hipMemcpyHostToDevice, ctx->stream));
    // numpy's (K, V) -> device layout (V, K)
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_sstats, 0, (size_t)V * ctx->ldk * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(transpose_kernel, dim3((V + 31) / 32, (K + 31) / 32), dim3(256), 0, ctx->stream,
                       ctx->d_kv_scratch, K, V, V, ctx->ldk, ctx->d_sstats);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_sstats = true;
    return PYLDA_OK;
}

int pylda_table_stride(const pylda_ctx* ctx) { return ctx ? ctx->ldk : 0; }
void* pylda_sstats_device(pylda_ctx* ctx) { return ctx ? ctx->d_sstats : nullptr; }
void* pylda_eta_device(pylda_ctx* ctx) { return ctx ? ctx->d_eta : nullptr; }
void* pylda_gamma_device(pylda_corpus* c) { return c ? c->d_gamma : nullptr; }

int pylda_mark_device_state(pylda_ctx* ctx, int have_eta, int have_sstats)
{
    // the caller wrote eta / sstats through the device pointers above
    if (!ctx) return PYLDA_ERR_INVALID;
    if (have_eta >= 0) ctx->have_eta = have_eta != 0;
    if (have_sstats >= 0) ctx->have_sstats = have_sstats != 0;
    return PYLDA_OK;
}

int pylda_host_alloc(int64_t bytes, void** out)
{
    if (!out || bytes < 0) return PYLDA_ERR_INVALID;
    *out = nullptr;
    const hipError_t e = hipHostMalloc(out, (size_t)std::max<int64_t>(bytes, 1), hipHostMallocDefault);
    if (e != hipSuccess) {
        *out = nullptr;
        return fail(nullptr, e == hipErrorOutOfMemory ? PYLDA_ERR_OOM : PYLDA_ERR_HIP, "host_alloc: %s", hipGetErrorString(e));
    }
    return PYLDA_OK;
}

int pylda_host_free(void* p)
{
    if (p && hipHostFree(p) != hipSuccess) return fail(nullptr, PYLDA_ERR_HIP, "host_free: not a pylda_host_alloc pointer");
    return PYLDA_OK;
}

int pylda_model_checkpoint(pylda_ctx* ctx, int restore)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)ctx->K * ctx->V * sizeof(double);
    if (!restore) {
        if (!ctx->have_eta) return fail(ctx, PYLDA_ERR_STATE, "model_checkpoint: eta was never set");
        if (!ctx->d_eta_ckpt) {
            const int rc = dev_alloc(ctx, &ctx->d_eta_ckpt, (size_t)ctx->K * ctx->V);
            if (rc != PYLDA_OK) return rc;
        }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_eta_ckpt, ctx->d_eta, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        if (!ctx->d_eta_ckpt) return fail(ctx, PYLDA_ERR_STATE, "model_checkpoint: nothing was saved");
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_eta, ctx->d_eta_ckpt, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        ctx->have_eta = true;
    }
    return PYLDA_OK;
}

int pylda_mark_time(pylda_ctx* ctx, int slot)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (slot < 0 || slot >= 4) return fail(ctx, PYLDA_ERR_INVALID, "mark_time: slot %d", slot);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->mark_event[slot]) HIP_TRY(ctx, hipEventCreate(&ctx->mark_event[slot]));
    HIP_TRY(ctx, hipEventRecord(ctx->mark_event[slot], ctx->stream));
    return PYLDA_OK;
}

int pylda_elapsed_ms(pylda_ctx* ctx, int slot_from, int slot_to, double* ms)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (slot_from < 0 || slot_from >= 4 || slot_to < 0 || slot_to >= 4 || !ms || !ctx->mark_event[slot_from] || !ctx->mark_event[slot_to])
        return fail(ctx, PYLDA_ERR_INVALID, "elapsed_ms: slots %d, %d", slot_from, slot_to);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->mark_event[slot_to]));
    float f = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&f, ctx->mark_event[slot_from], ctx->mark_event[slot_to]));
    *ms = f;
    return PYLDA_OK;
}

int pylda_work_counters(pylda_ctx* ctx, double* inner_iterations, double* inner_iteration_terms)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    // ONE read (and reset) of the device counters; pylda_executed_work / pylda_clock_counters report the rest of it
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->work_cache, ctx->d_work, sizeof ctx->work_cache, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_work, 0, sizeof ctx->work_cache, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (inner_iterations) *inner_iterations = ctx->work_cache[0];
    if (inner_iteration_terms) *inner_iteration_terms = ctx->work_cache[1];
    return PYLDA_OK;
}

int pylda_executed_work(pylda_ctx* ctx, double* tile_entries, double* handed_over)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (tile_entries) *tile_entries = ctx->work_cache[2];
    if (handed_over) *handed_over = ctx->work_cache[3];
    return PYLDA_OK;
}

int pylda_clock_counters(pylda_ctx* ctx, double* shader_ticks, double* wall_ticks, double* wall_hz)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (shader_ticks) *shader_ticks = ctx->work_cache[4];
    if (wall_ticks) *wall_ticks = ctx->work_cache[5];
    if (wall_hz) {
        int khz = 0;
        HIP_TRY(ctx, hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
        *wall_hz = 1e3 * khz;
    }
    return PYLDA_OK;
}

int pylda_comm_unique_id(void* id_out)
{
    if (!id_out) return fail(nullptr, PYLDA_ERR_INVALID, "comm_unique_id: NULL");
    std::string err;
    const int rc = pylda::comm_unique_id(id_out, &err);
    return rc == PYLDA_OK ? rc : fail(nullptr, rc, "comm_unique_id: %s", err.c_str());
}

int pylda_comm_init(pylda_ctx* ctx, const void* id, int rank, int world_size)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!id || world_size < 1 || rank < 0 || rank >= world_size)
        return fail(ctx, PYLDA_ERR_INVALID, "comm_init: rank %d of %d", rank, world_size);
    if (ctx->comm) return fail(ctx, PYLDA_ERR_STATE, "comm_init: the context already has a communicator");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    const int rc = pylda::comm_init(&ctx->comm, id, rank, world_size, &err);
    if (rc != PYLDA_OK) return fail(ctx, rc, "comm_init: %s", err.c_str());
    ctx->comm_world = world_size;
    return PYLDA_OK;
}

int pylda_comm_destroy(pylda_ctx* ctx)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->comm) pylda::comm_destroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 1;
    return PYLDA_OK;
}

int pylda_allreduce_sstats(pylda_ctx* ctx)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!ctx->comm) return fail(ctx, PYLDA_ERR_STATE, "allreduce_sstats: pylda_comm_init has not been called");
    if (!ctx->have_sstats) return fail(ctx, PYLDA_ERR_STATE, "allreduce_sstats: no training-mode E-step has run");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    // on the context's stream: ordered behind the E-step's kernels and before the M-step's
    const int rc = pylda::comm_allreduce_sum_f64(ctx->comm, ctx->d_sstats, (size_t)ctx->V * ctx->ldk, ctx->stream, &err);
    return rc == PYLDA_OK ? rc : fail(ctx, rc, "allreduce_sstats: %s", err.c_str());
}

int pylda_allreduce_doubles(pylda_ctx* ctx, double* values, int64_t n)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (!ctx->comm) return fail(ctx, PYLDA_ERR_STATE, "allreduce_doubles: pylda_comm_init has not been called");
    if (n < 0 || (n > 0 && !values)) return fail(ctx, PYLDA_ERR_INVALID, "allreduce_doubles: bad argument");
    if (n == 0) return PYLDA_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->comm_small_cap < (size_t)n) {
        dev_free(ctx->d_comm_small);
        ctx->comm_small_cap = 0;
        const int rc = dev_alloc(ctx, &ctx->d_comm_small, (size_t)n);
        if (rc != PYLDA_OK) return rc;
        ctx->comm_small_cap = (size_t)n;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_comm_small, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    std::string err;
    const int rc = pylda::comm_allreduce_sum_f64(ctx->comm, ctx->d_comm_small, (size_t)n, ctx->stream, &err);
    if (rc != PYLDA_OK) return fail(ctx, rc, "allreduce_doubles: %s", err.c_str());
    HIP_TRY(ctx, hipMemcpyAsync(values, ctx->d_comm_small, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PYLDA_OK;
}

int pylda_set_profiling(pylda_ctx* ctx, int enabled)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    ctx->profiling = enabled != 0;
    return PYLDA_OK;
}

int pylda_kernel_time(pylda_ctx* ctx, double* doc_kernel_ms, double* sstats_kernel_ms, int64_t* estep_calls)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    drain_events(ctx);
    if (doc_kernel_ms) *doc_kernel_ms = ctx->doc_kernel_ms;
    if (sstats_kernel_ms) *sstats_kernel_ms = ctx->sstats_kernel_ms;
    if (estep_calls) *estep_calls = ctx->estep_calls;
    ctx->doc_kernel_ms = ctx->sstats_kernel_ms = 0.0;
    ctx->estep_calls = 0;
    return PYLDA_OK;
}

namespace {
__global__ void special_test_kernel(const double* x, int64_t n, double* dg, double* lg)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        dg[i] = pylda::digamma(x[i]);
        lg[i] = pylda::lgamma_pos(x[i]);
    }
}
__global__ void expdigamma_test_kernel(const double* x, int64_t n, double c, double* out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = c > 1e3 ? pylda::exp_digamma_minus_levels(x[i], c - 2e3) : pylda::exp_digamma_minus(x[i], c);
}
}  // namespace

int pylda_test_special(pylda_ctx* ctx, int64_t n, const double* x, double* digamma_out, double* lgamma_out)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (n < 0 || !x || !digamma_out || !lgamma_out) return fail(ctx, PYLDA_ERR_INVALID, "test_special: bad argument");
    if (n == 0) return PYLDA_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    double *dx = nullptr, *dd = nullptr, *dl = nullptr;
    int rc = dev_alloc(ctx, &dx, (size_t)n);
    if (rc == PYLDA_OK) rc = dev_alloc(ctx, &dd, (size_t)n);
    if (rc == PYLDA_OK) rc = dev_alloc(ctx, &dl, (size_t)n);
    if (rc == PYLDA_OK) {
        hipError_t e = hipMemcpy(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(special_test_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dx, n, dd, dl);
            e = hipStreamSynchronize(ctx->stream);
        }
        if (e == hipSuccess) e = hipMemcpy(digamma_out, dd, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(lgamma_out, dl, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(ctx, PYLDA_ERR_HIP, "test_special: %s", hipGetErrorString(e));
    }
    dev_free(dx); dev_free(dd); dev_free(dl);
    return rc;
}

int pylda_test_expdigamma(pylda_ctx* ctx, int64_t n, const double* x, double c, double* out)
{
    if (!ctx) return PYLDA_ERR_INVALID;
    if (n < 0 || !x || !out) return fail(ctx, PYLDA_ERR_INVALID, "test_expdigamma: bad argument");
    if (n == 0) return PYLDA_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    double *dx = nullptr, *dout = nullptr;
    int rc = dev_alloc(ctx, &dx, (size_t)n);
    if (rc == PYLDA_OK) rc = dev_alloc(ctx, &dout, (size_t)n);
    if (rc == PYLDA_OK) {
        hipError_t e = hipMemcpy(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(expdigamma_test_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dx, n, c, dout);
            e = hipStreamSynchronize(ctx->stream);
        }
        if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(ctx, PYLDA_ERR_HIP, "test_expdigamma: %s", hipGetErrorString(e));
    }
    dev_free(dx); dev_free(dout);
    return rc;
}

}  // extern "C"

