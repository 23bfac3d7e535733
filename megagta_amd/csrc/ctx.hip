// ctx.hip — context life-cycle and error reporting of libmegagta_hip.so
#include "common.hpp"

namespace mgta {
static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
void ctx_release(mgta_ctx *ctx) {
    if (__atomic_sub_fetch(&ctx->refs, 1, __ATOMIC_ACQ_REL) > 0) return;   // objects created from it are still alive
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    delete ctx;
}
}  // namespace mgta

extern "C" {

const char *mgta_last_error(void) { return mgta::g_err; }
const char *mgta_version(void) { return "megagta_amd 0.1 (gfx950)"; }

mgta_ctx *mgta_ctx_create(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        mgta::set_error("no HIP device available (%s): libmegagta_hip has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return nullptr;
    }
    if (device_id < 0 || device_id >= n) { mgta::set_error("device %d out of range (%d devices)", device_id, n); return nullptr; }
    auto *ctx = new mgta_ctx;
    ctx->device = device_id;
    try {
        MGTA_HIP_CHECK(hipSetDevice(device_id));
        MGTA_HIP_CHECK(hipGetDeviceProperties(&ctx->prop, device_id));
        ctx->num_cus = ctx->prop.multiProcessorCount;
        MGTA_HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    } catch (const mgta::HipError &) {
        delete ctx;
        return nullptr;
    }
    return ctx;
}

void mgta_ctx_destroy(mgta_ctx *ctx) {
    if (ctx) mgta::ctx_release(ctx);
}

int mgta_ctx_set_full_lsd(mgta_ctx *ctx, int on) {
    if (!ctx) return MGTA_EINVAL;
    ctx->force_full_lsd = on & 1;
    ctx->force_lsd_tiles = on >> 1;
    return MGTA_OK;
}

int mgta_ctx_set_search_cost_rate(mgta_ctx *ctx, int expansions_per_seed) {
    if (!ctx || expansions_per_seed < -64) return MGTA_EINVAL;
    ctx->search_cost_rate = expansions_per_seed;
    ctx->search_cost_knee = 0; ctx->search_cost_rate2 = 0;
    return MGTA_OK;
}

int mgta_ctx_set_search_cost_curve(mgta_ctx *ctx, int expansions_per_seed, uint64_t knee_expansions, int expansions_per_seed_beyond) {
    if (!ctx || expansions_per_seed < 1 || (knee_expansions != 0 && expansions_per_seed_beyond < expansions_per_seed)) {
        mgta::set_error("mgta_ctx_set_search_cost_curve: rate >= 1, and beyond the knee a rate >= the first one (the delay is concave in the expansions)");
        return MGTA_EINVAL;
    }
    ctx->search_cost_rate = expansions_per_seed;
    ctx->search_cost_knee = knee_expansions;
    ctx->search_cost_rate2 = knee_expansions ? expansions_per_seed_beyond : 0;
    return MGTA_OK;
}

int mgta_ctx_keep_stream(mgta_ctx *ctx, int on) {
    if (!ctx) return MGTA_EINVAL;
    ctx->keep_stream = on == 2 ? 2 : on ? 1 : 0;
    if (!on) {
        // after a keep-stream build last_rec / last_tips point INTO the stream buffers: they go with them
        if (ctx->acc_valid) { ctx->last_rec = nullptr; ctx->last_tips = nullptr; ctx->last_first = nullptr; ctx->last_n_rec = 0; ctx->last_n_tips = 0; ctx->last_k = 0; }
        ctx->acc_rec.release(); ctx->acc_tips.release(); ctx->acc_valid = false;
    }
    return MGTA_OK;
}

int mgta_ctx_release_scratch(mgta_ctx *ctx) {
    if (!ctx) return MGTA_EINVAL;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    const bool in_pool = ctx->last_rec && !ctx->acc_valid;      // the last pass's stream lives in the pool: it goes with it
    ctx->pool.clear();
    ctx->astar.pool.release(); ctx->astar.meta.release();
    if (in_pool) { ctx->last_rec = nullptr; ctx->last_tips = nullptr; ctx->last_first = nullptr; ctx->last_n_rec = 0; ctx->last_n_tips = 0; ctx->last_k = 0; }
    return MGTA_OK;
}

int mgta_ctx_device_memory(mgta_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes) {
    if (!ctx) return MGTA_EINVAL;
    size_t f = 0, t = 0;
    if (hipSetDevice(ctx->device) != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) { mgta::set_error("hipMemGetInfo failed"); return MGTA_EHIP; }
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return MGTA_OK;
}

int mgta_ctx_set_mem_limit(mgta_ctx *ctx, uint64_t bytes) {
    if (!ctx) return MGTA_EINVAL;
    ctx->mem_limit = bytes;
    return MGTA_OK;
}

}  // extern "C"
