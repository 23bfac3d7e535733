// common.hpp — shared host-side plumbing of libmegagta_hip.so (context, error reporting, device buffers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/megagta_hip.h"

namespace mgta {

void set_error(const char *fmt, ...);

struct HipError {
    int code;
};

#define MGTA_HIP_CHECK(expr)                                                                        \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            ::mgta::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            throw ::mgta::HipError{e_ == hipErrorOutOfMemory ? MGTA_ENOMEM : MGTA_EHIP};           \
        }                                                                                           \
    } while (0)

// RAII device allocation; tracks the context's running / peak byte count.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    uint64_t *live = nullptr, *peak = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept { *this = std::move(o); }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; live = o.live; peak = o.peak; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t n, uint64_t *live_ = nullptr, uint64_t *peak_ = nullptr) {
        release();
        if (n == 0) n = 16;
        MGTA_HIP_CHECK(hipMalloc(&p, n));
        bytes = n; live = live_; peak = peak_;
        if (live) { *live += n; if (peak && *live > *peak) *peak = *live; }
    }
    void release() {
        if (p) { (void)hipFree(p); if (live) *live -= bytes; }
        p = nullptr; bytes = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

}  // namespace mgta

namespace mgta {
struct AstarArenas {           // A* work memory kept between mgta_astar_batch calls: the chunk pool (base arenas of the search slots +
    DevBuf pool, meta;         // the region the searches grow into) and the allocator's bookkeeping
};
}  // namespace mgta

struct mgta_ctx {
    int refs = 1;                 // the handle itself + every reads / graph / hmm object created from it
    int device = 0;
    hipStream_t stream = nullptr;
    uint64_t mem_limit = 0;       // 0 = auto
    int force_full_lsd = 0;
    int lsd_skip_left = 0;       // sorts left that go straight to LSD passes in LDS (the last look found mostly long runs)
    int astar_log_b0 = 0;        // base arena of a search slot = 1 << astar_log_b0 nodes (0 = default 12); searches grow beyond it in place
    uint64_t astar_pool_bytes = 0;   // device memory the searches may grow into (0 = auto)
    int search_share_num = 1, search_share_den = 1;   // share of the CUs a search batch of this context takes (mgta_ctx_set_search_share)
    int search_cost_rate = 0;    // shared-cache searches: a path found with c expansions becomes visible c / rate seeds later (0 = no cost term)
    uint64_t search_cost_knee = 0;   // ... up to this many expansions, and 1 / search_cost_rate2 seeds per expansion beyond (0 = one rate)
    int search_cost_rate2 = 0;
    int force_lsd_tiles = 0;     // segment-local sort: LSD passes over every digit, no finish by comparison
    uint64_t live_bytes = 0, peak_bytes = 0;
    int num_cus = 256;
    hipDeviceProp_t prop;
    std::vector<mgta::DevBuf> pool;   // grow-only scratch kept between calls
    mgta::AstarArenas astar;
    std::vector<int64_t> edge_counting;   // (k+1)-mer multiplicity histogram of the last stage-1 run (.counting)
    const void *last_rec = nullptr;   // records of the last build pass, still resident in the pool
    uint64_t last_n_rec = 0;
    uint32_t last_bucket_lo = 0, last_bucket_hi = 0;
    const void *last_tips = nullptr;  // tip labels of that pass (words_per_tip words each), and where every bucket starts ([nb][3], -1 = empty)
    const void *last_first = nullptr;
    uint64_t last_n_tips = 0;
    int last_k = 0, last_words_per_tip = 0;
    // mgta_ctx_keep_stream: a build of several memory-bound passes also leaves its WHOLE edge stream on the device (records and tip
    // labels of every pass appended here, records per bucket on the host), so that mgta_sdbg_load_resident works at any size
    int keep_stream = 0;
    bool acc_valid = false;
    mgta::DevBuf acc_rec, acc_tips;
    uint64_t acc_n_rec = 0, acc_n_tips = 0;
    std::vector<int64_t> acc_items;
};

namespace mgta {
inline void ctx_retain(mgta_ctx *c) { __atomic_add_fetch(&c->refs, 1, __ATOMIC_RELAXED); }
void ctx_release(mgta_ctx *c);    // frees the context when the last reference goes (ctx.hip)
}  // namespace mgta

struct mgta_reads {
    mgta_ctx *ctx = nullptr;
    const uint32_t *d_packed = nullptr;   // padded with >= 16 zero words past n_words
    const uint64_t *d_start = nullptr;    // [n_reads+1]
    uint64_t n_words = 0, n_reads = 0;
    mgta::DevBuf own_packed, own_start;   // empty when adopted
};
