// probe.hip — what the device sustains on random 128-byte lines (gfx950): the ceiling of the graph walks and of the A* kernel.
//
// The A* expansion (hmm_graph_search.h:191-343) and the walks under it (OutgoingEdges succinct_dbg.cpp:78-97, rank / select
// rank_and_select.h:153-280) are chains of random 128-byte line reads: one GLine, one heap block, one hash line, one node at a time.
// What bounds them is neither the 8 TB/s stream rate nor an MFMA peak but (a) how many independent random lines per second the memory
// system delivers at a given number of lines in flight and (b) the latency of one dependent line.  This file measures both with the same
// access shape the kernels use: a GROUP OF 8 LANES reads one aligned 128-byte line (16 B per lane, one request), a wavefront carries
// `groups` (1..8) such groups, every group keeps `unroll` (1..8) independent lines in flight, `waves_per_cu` (1..32) waves per CU.
// Lines in flight per CU = waves_per_cu * groups * unroll.  dependent = 1: the next line's index comes out of the line just read
// (pointer chase: unroll chains per group), so time / steps is the loaded latency of one dependent line; dependent = 2: the chase with a
// 16-byte store to another random line in every step (what a store in front of a dependent fetch costs: vmcnt counts both).
#include "common.hpp"
#include "device_utils.hpp"

namespace mgta {
namespace {

__device__ __forceinline__ uint64_t pmix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ uint64_t line_of(uint64_t h, uint64_t n_lines) { return __umul64hi(h, n_lines); }   // h uniform in 2^64 -> [0, n_lines)

// every 16-byte element of line i holds pmix(i) in its low 8 bytes (the chase reads the next index from whichever element a lane holds)
__global__ __launch_bounds__(256) void probe_fill_kernel(uint4 *tab, uint64_t n_lines) {
    const uint64_t n = n_lines * 8, stride = (uint64_t)gridDim.x * 256;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {
        const uint64_t h = pmix((e >> 3) + 0x9E3779B97F4A7C15ULL);
        tab[e] = make_uint4((uint32_t)h, (uint32_t)(h >> 32), (uint32_t)e, 0u);
    }
}

// DEP: 0 = independent lines, 1 = pointer chase, 2 = pointer chase in which every step also STORES 16 bytes per lane (the group: a whole
// line) to another random line (3: ONE lane of the group stores its 16 bytes: a partial line) before it
// reads the next one (vmcnt counts stores and loads together, in order: the load's wait is also the wait for the store's acknowledgement --
// the shape of the A* commit, where a heap / hash / node store is followed by the next dependent fetch)
template <int U, int DEP>
__global__ __launch_bounds__(64) void probe_lines_kernel(uint4 *tab, uint64_t n_lines, int groups, uint64_t steps, uint64_t salt,
                                                         unsigned long long *sink) {
    const int lane = threadIdx.x & 63, grp = lane >> 3, sub = lane & 7;
    if (grp >= groups) return;                                         // (whole groups leave: the rest of the wave runs on)
    const uint64_t gid = (uint64_t)blockIdx.x * 8 + (uint64_t)grp;
    uint64_t acc = 0;
    uint64_t idx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) idx[u] = line_of(pmix(salt + gid * 0x100000001B3ULL + (uint64_t)u * 0x9E3779B97F4A7C15ULL), n_lines);
    for (uint64_t s = 0; s < steps; ++s) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = tab[idx[u] * 8 + (uint64_t)sub];     // U independent lines in flight per group
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t got = (uint64_t)v[u].x | ((uint64_t)v[u].y << 32);
            acc += got + v[u].z;
            if (DEP == 2 || (DEP == 3 && sub == 0)) {                   // (the stored element keeps its line's words: the chase reads the same values afterwards)
                const uint64_t wl = line_of(pmix(got + 0x632BE59BD9B4E019ULL + (uint64_t)u), n_lines), e = wl * 8 + (uint64_t)sub;
                const uint64_t h = pmix(wl + 0x9E3779B97F4A7C15ULL);
                tab[e] = make_uint4((uint32_t)h, (uint32_t)(h >> 32), (uint32_t)e, (uint32_t)s);
            }
            if (DEP) idx[u] = line_of(pmix(got ^ (s + (uint64_t)u)), n_lines);   // the next line is named by the one just read
            else idx[u] = line_of(pmix(salt + gid * 0x100000001B3ULL + (s + 1) * 0xD6E8FEB86659FD93ULL + (uint64_t)u * 0x9E3779B97F4A7C15ULL), n_lines);
        }
    }
    if (acc == 0x0123456789ABCDEFULL) atomicAdd(sink, 1ull);           // (keeps the loads alive)
}

template <int U>
void launch_probe(int dep, int blocks, uint4 *tab, uint64_t n_lines, int groups, uint64_t steps, uint64_t salt, unsigned long long *sink, hipStream_t st) {
    if (dep == 3) hipLaunchKernelGGL((probe_lines_kernel<U, 3>), dim3(blocks), dim3(64), 0, st, tab, n_lines, groups, steps, salt, sink);
    else if (dep == 2) hipLaunchKernelGGL((probe_lines_kernel<U, 2>), dim3(blocks), dim3(64), 0, st, tab, n_lines, groups, steps, salt, sink);
    else if (dep) hipLaunchKernelGGL((probe_lines_kernel<U, 1>), dim3(blocks), dim3(64), 0, st, tab, n_lines, groups, steps, salt, sink);
    else hipLaunchKernelGGL((probe_lines_kernel<U, 0>), dim3(blocks), dim3(64), 0, st, tab, n_lines, groups, steps, salt, sink);
}

}  // namespace
}  // namespace mgta

extern "C" int mgta_probe_random_lines(mgta_ctx *ctx, uint64_t table_bytes, mgta_line_probe *cfg, int n_cfg) {
    using namespace mgta;
    if (!ctx || !cfg || n_cfg < 0 || table_bytes < (1ull << 20)) { set_error("mgta_probe_random_lines: bad argument"); return MGTA_EINVAL; }
    for (int i = 0; i < n_cfg; ++i) {
        const mgta_line_probe &c = cfg[i];
        const bool u_ok = c.unroll == 1 || c.unroll == 2 || c.unroll == 4 || c.unroll == 8;
        if (c.waves_per_cu < 1 || c.waves_per_cu > 32 || c.groups < 1 || c.groups > 8 || !u_ok || c.steps < 1 || c.dependent < 0 || c.dependent > 3) {
            set_error("mgta_probe_random_lines: configuration %d: waves_per_cu 1..32, groups 1..8, unroll 1|2|4|8, steps >= 1", i);
            return MGTA_EINVAL;
        }
    }
    try {
        MGTA_HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const uint64_t n_lines = table_bytes / 128;
        DevBuf tab, sink;
        tab.alloc(n_lines * 128, &ctx->live_bytes, &ctx->peak_bytes);
        sink.alloc(8);
        MGTA_HIP_CHECK(hipMemsetAsync(sink.p, 0, 8, st));
        hipLaunchKernelGGL(probe_fill_kernel, dim3(ctx->num_cus * 8), dim3(256), 0, st, tab.as<uint4>(), n_lines);
        MGTA_HIP_CHECK(hipGetLastError());
        hipEvent_t e0, e1;
        MGTA_HIP_CHECK(hipEventCreate(&e0));
        MGTA_HIP_CHECK(hipEventCreate(&e1));
        int rc = MGTA_OK;
        for (int i = 0; i < n_cfg && rc == MGTA_OK; ++i) {
            mgta_line_probe &c = cfg[i];
            const int blocks = ctx->num_cus * c.waves_per_cu;           // one wave per workgroup: the dispatcher spreads them over the CUs evenly
            const uint64_t salt = 0x5851F42D4C957F2DULL * (uint64_t)(i + 1);
            hipError_t e = hipEventRecord(e0, st);
            switch (c.unroll) {
                case 1: launch_probe<1>(c.dependent, blocks, tab.as<uint4>(), n_lines, c.groups, c.steps, salt, sink.as<unsigned long long>(), st); break;
                case 2: launch_probe<2>(c.dependent, blocks, tab.as<uint4>(), n_lines, c.groups, c.steps, salt, sink.as<unsigned long long>(), st); break;
                case 4: launch_probe<4>(c.dependent, blocks, tab.as<uint4>(), n_lines, c.groups, c.steps, salt, sink.as<unsigned long long>(), st); break;
                default: launch_probe<8>(c.dependent, blocks, tab.as<uint4>(), n_lines, c.groups, c.steps, salt, sink.as<unsigned long long>(), st); break;
            }
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipEventRecord(e1, st);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e != hipSuccess) { set_error("mgta_probe_random_lines: %s", hipGetErrorString(e)); rc = MGTA_EHIP; break; }
            c.ms = ms;
            c.lines = (uint64_t)blocks * (uint64_t)c.groups * (uint64_t)c.unroll * c.steps;
            c.lines_in_flight_per_cu = c.waves_per_cu * c.groups * c.unroll;
            c.gb_per_s = ms > 0 ? (double)c.lines * 128.0 / (ms * 1e6) : 0.0;
            c.ns_per_step = ms > 0 ? (double)ms * 1e6 / (double)c.steps : 0.0;
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        return rc;
    } catch (const HipError &e) { return e.code; }
}
