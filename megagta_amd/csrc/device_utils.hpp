// device_utils.hpp — wave64 / workgroup primitives shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mgta {

constexpr int kWave = 64;   // CDNA wavefront

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// inclusive wave scan (sum) of a 32-bit value via DPP-free shuffles
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (l >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t v) {
    int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t t = __shfl_up(v, d, 64);
        if (l >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Block-wide exclusive scan of one uint32 per thread (blockDim.x = NT, multiple of 64, <= 1024).
// `scratch` needs NT/64 + 1 words of LDS.  Returns the exclusive prefix; *total = block sum.
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *scratch, uint32_t *total) {
    constexpr int NW = NT / 64;
    uint32_t inc = wave_incl_scan(v);
    int w = wave_id(), l = lane_id();
    __syncthreads();
    if (l == 63) scratch[w] = inc;
    __syncthreads();
    if (w == 0) {
        uint32_t x = l < NW ? scratch[l] : 0;
        uint32_t xi = wave_incl_scan(x);
        if (l < NW) scratch[l] = xi - x;
        if (l == NW - 1) scratch[NW] = xi;
    }
    __syncthreads();
    uint32_t res = scratch[w] + inc - v;
    if (total) *total = scratch[NW];
    return res;
}
template <int NT>
__device__ __forceinline__ uint64_t block_excl_scan64(uint64_t v, uint64_t *scratch, uint64_t *total) {
    constexpr int NW = NT / 64;
    uint64_t inc = wave_incl_scan64(v);
    int w = wave_id(), l = lane_id();
    __syncthreads();
    if (l == 63) scratch[w] = inc;
    __syncthreads();
    if (w == 0) {
        uint64_t x = l < NW ? scratch[l] : 0;
        uint64_t xi = wave_incl_scan64(x);
        if (l < NW) scratch[l] = xi - x;
        if (l == NW - 1) scratch[NW] = xi;
    }
    __syncthreads();
    uint64_t res = scratch[w] + inc - v;
    if (total) *total = scratch[NW];
    return res;
}

}  // namespace mgta
