// scan.hpp — device-wide exclusive prefix sums (uint32 counts -> uint64 offsets), three launches:
// chunk sums, single-workgroup scan of the sums, chunk-local scan + base.  Used for per-block item
// bases of the read scan and for the compaction steps of the edge emitter.
#pragma once
#include "common.hpp"
#include "device_utils.hpp"

namespace mgta {

constexpr int kScanThreads = 256;
constexpr int kScanPerThread = 16;
constexpr int kScanChunk = kScanThreads * kScanPerThread;   // 4096 elements per workgroup

static __global__ __launch_bounds__(kScanThreads) void scan_chunk_sums(const uint32_t *in, uint64_t n, uint64_t *chunk_sum) {
    __shared__ uint64_t scratch[kScanThreads / 64 + 1];
    uint64_t base = (uint64_t)blockIdx.x * kScanChunk;
    uint64_t s = 0;
    for (int i = 0; i < kScanPerThread; ++i) {
        uint64_t idx = base + (uint64_t)i * kScanThreads + threadIdx.x;
        if (idx < n) s += in[idx];
    }
    s = wave_sum64(s);
    if (lane_id() == 0) scratch[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < kScanThreads / 64; ++w) t += scratch[w];
        chunk_sum[blockIdx.x] = t;
    }
}

// one workgroup of 1024 threads, sequential over tiles of 1024: exclusive scan in place; total -> *total
static __global__ __launch_bounds__(1024) void scan_sums_inplace(uint64_t *v, uint64_t n, uint64_t *total) {
    __shared__ uint64_t scratch[1024 / 64 + 1];
    __shared__ uint64_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n; base += 1024) {
        uint64_t idx = base + threadIdx.x;
        uint64_t x = idx < n ? v[idx] : 0;
        uint64_t tot;
        uint64_t ex = block_excl_scan64<1024>(x, scratch, &tot);
        uint64_t carry = carry_s;
        if (idx < n) v[idx] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = carry_s;
}

static __global__ __launch_bounds__(kScanThreads) void scan_chunks(const uint32_t *in, uint64_t n, const uint64_t *chunk_base,
                                                            uint64_t *out) {
    __shared__ uint64_t scratch[kScanThreads / 64 + 1];
    // blocked arrangement: thread t owns elements [t*16, t*16+16) of the chunk
    uint64_t base = (uint64_t)blockIdx.x * kScanChunk + (uint64_t)threadIdx.x * kScanPerThread;
    uint32_t loc[kScanPerThread];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < kScanPerThread; ++i) {
        uint64_t idx = base + i;
        loc[i] = idx < n ? in[idx] : 0;
        s += loc[i];
    }
    uint64_t ex = block_excl_scan64<kScanThreads>(s, scratch, nullptr) + chunk_base[blockIdx.x];
#pragma unroll
    for (int i = 0; i < kScanPerThread; ++i) {
        uint64_t idx = base + i;
        if (idx < n) out[idx] = ex;
        ex += loc[i];
    }
}

// out[i] = sum_{j<i} in[j]; *d_total = sum of all.  tmp must hold ceil(n/4096) uint64.
inline void exclusive_scan_u32(hipStream_t st, const uint32_t *in, uint64_t n, uint64_t *out, uint64_t *tmp, uint64_t *d_total) {
    if (n == 0) { MGTA_HIP_CHECK(hipMemsetAsync(d_total, 0, 8, st)); return; }
    uint64_t chunks = (n + kScanChunk - 1) / kScanChunk;
    hipLaunchKernelGGL(scan_chunk_sums, dim3((unsigned)chunks), dim3(kScanThreads), 0, st, in, n, tmp);
    hipLaunchKernelGGL(scan_sums_inplace, dim3(1), dim3(1024), 0, st, tmp, chunks, d_total);
    hipLaunchKernelGGL(scan_chunks, dim3((unsigned)chunks), dim3(kScanThreads), 0, st, in, n, tmp, out);
}
inline uint64_t scan_tmp_elems(uint64_t n) { return (n + kScanChunk - 1) / kScanChunk + 1; }

}  // namespace mgta
