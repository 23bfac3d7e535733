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
__device__ __forceinline__ long long wave_min_ll(long long v) {   // every lane gets the minimum over the wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int lo = __shfl_xor((int)(v & 0xFFFFFFFFll), off, 64), hi = __shfl_xor((int)(v >> 32), off, 64);
        const long long o = ((long long)hi << 32) | (unsigned int)lo;
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { uint32_t o = __shfl_xor(v, d, 64); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { uint32_t o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Lanes of the wave holding the same digit value (gfx950 has no match-any instruction: one ballot per digit bit).
// rank = peers in lower lanes, cnt = all peers (self included); lanes with !valid are nobody's peer.
// Per bit: sign-extend the bit, ballot it, peers &= xnor(ballot, sign) — 4 VALU ops.
__device__ __forceinline__ void wave_match(uint32_t dg, int nbits, bool valid, uint32_t &rank, uint32_t &cnt) {
    const uint64_t v = __ballot(valid);
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    auto step = [&](int b) {
        const int sx = (int)(dg << (31 - b)) >> 31;
        const uint64_t bal = __ballot(sx < 0);
        lo &= ~((uint32_t)bal ^ (uint32_t)sx);
        hi &= ~((uint32_t)(bal >> 32) ^ (uint32_t)sx);
    };
    step(0); step(1); step(2); step(3);
    if (nbits > 4) { step(4); step(5); step(6); step(7); }          // wave-uniform
    rank = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
    cnt = (uint32_t)__popc(lo) + (uint32_t)__popc(hi);
}

// a value every lane of the wave holds identically (read from LDS, say) moved to scalar registers
__device__ __forceinline__ uint32_t wave_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t wave_uniform(uint64_t v) {
    return ((uint64_t)wave_uniform((uint32_t)(v >> 32)) << 32) | wave_uniform((uint32_t)v);
}

// orders the LDS accesses of one wave as written (lane-to-lane hand-over through LDS inside a wave: the hardware
// executes a wave's LDS instructions in order, the compiler must not cache or reorder them)
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

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

// ---------------------------------------------------------------------------------------------------------------------
// Chained scan across workgroups (decoupled look-back): tile t publishes its aggregate, then walks back over its predecessors
// until it meets one whose inclusive prefix is known.  state[t] = flag << 62 | value (flag 0 empty, 1 aggregate, 2 inclusive), zeroed
// by the host before the launch; tiles are numbered by an atomic ticket so every predecessor of a running tile has started.
// All loads/stores are agent-scope atomics: the L2s of the eight XCDs are not coherent with each other for plain accesses.
// The walk is bounded: after kChainSpinLimit fruitless polls the tile raises *error and returns (the host fails the call).
// ---------------------------------------------------------------------------------------------------------------------
constexpr unsigned long long kChainAggregate = 1ull << 62, kChainInclusive = 2ull << 62, kChainValue = (1ull << 62) - 1ull;
constexpr uint32_t kChainSpinLimit = 1u << 19;   // ~0.5 s of polling

__device__ __forceinline__ uint32_t chain_ticket(uint32_t *ticket, uint32_t *s_slot) {   // all threads call; one barrier
    if (threadIdx.x == 0) *s_slot = atomicAdd(ticket, 1u);
    __syncthreads();
    return wave_uniform(*s_slot);
}

// one full wave calls with the same (tile, agg); returns the sum of the aggregates of the tiles before `tile`
__device__ __forceinline__ uint64_t chain_exclusive(unsigned long long *state, uint32_t tile, uint64_t agg, uint32_t *error) {
    const int lane = lane_id();
    if (tile == 0) {
        if (lane == 0) __hip_atomic_store(&state[0], kChainInclusive | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0;
    }
    if (lane == 0) __hip_atomic_store(&state[tile], kChainAggregate | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t excl = 0;
    long long base = (long long)tile - 1;                             // the nearest predecessor not summed yet
    uint32_t spins = 0;
    for (;;) {
        const long long t = base - lane;
        const unsigned long long v = t >= 0 ? __hip_atomic_load(&state[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kChainInclusive;
        const uint32_t flag = (uint32_t)(v >> 62);
        const uint64_t empty = __ballot(flag == 0), incl = __ballot(flag == 2);
        const int first_incl = incl ? __ffsll((long long)incl) - 1 : 64;
        const uint64_t need = first_incl >= 63 ? ~0ull : ((2ull << first_incl) - 1ull);   // lanes up to the first inclusive one
        if (empty & need) {
            ++spins;
            if (spins > kChainSpinLimit || ((spins & 255u) == 0 && __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                if (lane == 0) atomicExch(error, 1u);                 // give up; every tile still waiting follows within 256 polls
                break;
            }
            __builtin_amdgcn_s_sleep(16);
            continue;
        }
        excl += wave_sum64(lane <= first_incl ? (uint64_t)(v & kChainValue) : 0ull);
        if (first_incl < 64) break;
        base -= 64;
    }
    if (lane == 0) __hip_atomic_store(&state[tile], kChainInclusive | ((excl + agg) & kChainValue), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

}  // namespace mgta
