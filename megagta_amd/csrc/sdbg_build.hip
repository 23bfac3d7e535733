// sdbg_build.hip — reads -> succinct-de-Bruijn-graph edge stream on gfx950 (MI355X).
//
// Replaces, behind the C ABI of include/megagta_hip.h, the reference's CX1 stage-2 pipeline
//   s2_lv0_calc_bucket_size   cx1_read2sdbg_s2.cpp:252-315   (bucket census)
//   s2_lv1_fill_offset        cx1_read2sdbg_s2.cpp:475-584   (per-bucket item lists)
//   s2_lv2_extract_substr_    cx1_read2sdbg_s2.cpp:586-677   (key extraction, CopySubstring[RC] packed_reads.h:44-176)
//   lv2_cpu_radix_sort_st     lv2_cpu_sort.h:133-150         (ascending lexicographic sort of the W-word keys)
//   output_ + SdbgWriter::write  cx1_read2sdbg_s2.cpp:742-835, sdbg_multi_io.h:83-112 (edge records)
// with a device-resident design (no differential offset lists, no per-bucket CPU threads):
//
//   pass over a bucket range [b_lo,b_hi) that fits the memory budget (the analogue of CX1's lv1 loop)
//     1. items per workgroup: closed form when k+1 is odd and every bucket is wanted (2 (len - k) + 4 per read), else
//        item_scan<count>: one wave per read, one lane per (k+1)-mer position: funnel-shift the edge out of the 2-bit read
//        array, reverse-complement it in registers, count the <= 6 sort items of the position whose first 8 characters fall
//        into the range (one scan counts every range that is still ahead)
//     2. prefix sum of the per-workgroup counts
//     3. item_scan<write>   same scan, keys written (array-of-structs, W words) with wave-aggregated offsets, and next to every key
//        the byte the first global sort pass sorts on (the SIDE array, W <= 7)
//     4. sort: P <= 4 global LSD passes on the P leading key bytes (digit census per tile FROM THE SIDE BYTES, row scan, stable
//        LDS-staged scatter with wave-level match ranking and coalesced run writes, which leaves the next pass's side bytes behind:
//        the keys cross HBM once per pass for the scatter, not twice), then every segment of equal prefix is finished inside LDS:
//        one counting pass on (segment, next <= 8 bits) with LDS atomics, then every key ranks itself inside its short run of
//        equal leading bits by comparison (local_sort_kernel); tiles with long runs and segments that did not fit take LSD
//        passes in LDS (local_lsd_kernel), segments longer than a tile global passes over their own range (segment_sort_kernel)
//     5. edge emission: run descriptors (a, b, group head, bucket head) compacted in one read of the keys by a chained scan
//        across workgroups -> per run decision (W, last, tip, multiplicity, $-suppression) -> order-preserving compaction of
//        records, large multiplicities and tip labels + per-bucket boundaries
//
// Everything is integer / byte work bound by HBM traffic; no MFMA.  Parity: bit-exact edge stream vs
// the oracle (tests/test_sdbg_build_gpu.py).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <type_traits>

#include "common.hpp"
#include "device_utils.hpp"
#include "scan.hpp"

namespace mgta {

constexpr int kDollar = 4;   // kSentinelValue, cx1_read2sdbg.h:72

template <int W>
struct Key {
    uint32_t w[W];
};

// ---------------------------------------------------------------------------------------------
// multi-word helpers on big-endian 2-bit strings (x[0] holds the first 16 characters)
// ---------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void shl_bits(uint32_t (&x)[W], int s) {   // 0 <= s <= 32
    if (s == 0) return;
    if (s == 32) {
#pragma unroll
        for (int j = 0; j < W - 1; ++j) x[j] = x[j + 1];
        x[W - 1] = 0;
        return;
    }
#pragma unroll
    for (int j = 0; j < W; ++j) x[j] = (x[j] << s) | (j + 1 < W ? (x[j + 1] >> (32 - s)) : 0u);
}

template <int W>
__device__ __forceinline__ void keep_chars(uint32_t (&x)[W], int n) {   // keep the first n characters
#pragma unroll
    for (int j = 0; j < W; ++j) {
        int lo = j * 16;
        if (n <= lo) x[j] = 0;
        else if (n < lo + 16) x[j] &= ~0u << (32 - 2 * (n - lo));
    }
}

// key = characters [from, from+n) of `e` (n = k or k-1), zero padded, flags in the low 4 bits of the
// last word: (n == k) << 3 | prev     [cx1_read2sdbg_s2.cpp:613-671]
template <int W>
__device__ __forceinline__ Key<W> make_key(const uint32_t (&e)[W], int from, int n, int k, int prev) {
    uint32_t t[W];
#pragma unroll
    for (int j = 0; j < W; ++j) t[j] = e[j];
    shl_bits<W>(t, 2 * from);
    keep_chars<W>(t, n);
    t[W - 1] |= (uint32_t)(n == k) << 3;
    t[W - 1] |= (uint32_t)prev;
    Key<W> key;
#pragma unroll
    for (int j = 0; j < W; ++j) key.w[j] = t[j];
    return key;
}

__device__ __forceinline__ uint32_t rev_chars(uint32_t x) {   // reverse the 16 characters of a word
    x = __brev(x);
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}

// ---------------------------------------------------------------------------------------------
// 1./3. read scan
// ---------------------------------------------------------------------------------------------
constexpr int kScanBlock = 256;            // 4 waves
constexpr int kReadsPerBlock = 64;

struct ScanArgs {
    const uint32_t *packed;
    uint64_t n_words;
    const uint64_t *start;
    uint64_t n_reads;
    int k;
    uint32_t b_lo, b_hi;                   // bucket range [b_lo, b_hi)
    uint32_t *block_count;                 // count mode: items per workgroup
    const uint64_t *block_base;            // write mode
    void *out;                             // Key<W>*
    unsigned long long *n_kmers;           // count mode, sum over reads of max(0, len - k)
    const unsigned long long *is_solid;    // stage-1 verdicts (bit num_k1_per_read*read + position), nullptr = every position solid
    int num_k1_per_read;
    uint64_t n_short;                      // reads >= n_short (assist sequences) are always solid (s2.cpp:276)
    unsigned long long *n_sentinel;        // closed-form writers, k+1 even: number of sentinel keys written (see item_write_closed_kernel)
    uint32_t multi_width, multi_n;         // count mode: multi_n > 0 counts multi_n consecutive bucket ranges of multi_width buckets from
                                           // b_lo in one scan: block_count[range * gridDim.x + workgroup]
    uint64_t multi_magic;                  // ceil(2^32 / multi_width) (the range of a bucket without a division per item)
    uint8_t *side;                         // write mode, optional: ((word 0 - side_bias) >> side_shift) & 255 of every key next to it (the
    int side_shift;                        // first global sort pass then counts these bytes instead of reading the keys back)
    uint32_t side_bias;
};
constexpr int kMaxCountRanges = 64;

// masks of the first n characters of a W-word string (keep_chars as W ANDs with wave-uniform operands)
template <int W>
__device__ __forceinline__ void keep_masks(int n, uint32_t (&m)[W]) {
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int lo = j * 16;
        m[j] = n <= lo ? 0u : (n < lo + 16 ? ~0u << (32 - 2 * (n - lo)) : ~0u);
    }
}

// The kernel is bound by VALU issue in write mode (SQ counters at 100 M reads, k = 44: 227 vector instructions per 64 positions, a wave64
// instruction takes four cycles of a 16-lane SIMD) and by load latency in count mode (117), so the write path keeps everything
// wave-uniform out of the vector unit: character masks and the workgroup's slice of the output as scalars, no bounds checks on the read
// words where the workgroup's reads end well inside the array, no staging array for the items of a position.
template <int W, bool WRITE>
__global__ __launch_bounds__(kScanBlock) void item_scan_kernel(ScanArgs a) {
    __shared__ uint32_t s_cursor;
    __shared__ uint32_t s_wave_cnt[kScanBlock / 64];
    __shared__ uint32_t s_range_cnt[kMaxCountRanges];
    const int k = a.k;
    const int lane = lane_id(), wv = wave_id();
    __shared__ uint64_t s_start[kReadsPerBlock + 1];                  // one coalesced load instead of two dependent ones per read
    const bool multi = !WRITE && a.multi_n > 0;
    const uint64_t multi_magic = a.multi_magic;                      // ceil(2^32 / multi_width): exact quotients for operands of at most 2^16
    const bool magic24 = multi_magic < (1u << 24);                   // (multi_width > 256: the product of a 16-bit and a 24-bit operand)
    uint64_t r0 = (uint64_t)blockIdx.x * kReadsPerBlock;
    uint64_t r1 = r0 + kReadsPerBlock < a.n_reads ? r0 + kReadsPerBlock : a.n_reads;
    if (threadIdx.x == 0) s_cursor = 0;
    if (!WRITE && threadIdx.x < kMaxCountRanges) s_range_cnt[threadIdx.x] = 0;
    if (threadIdx.x <= r1 - r0) s_start[threadIdx.x] = a.start[r0 + threadIdx.x];
    __syncthreads();
    Key<W> *out = reinterpret_cast<Key<W> *>(a.out);
    uint64_t base = WRITE ? a.block_base[blockIdx.x] : 0;
    uint32_t my_count = 0;
    unsigned long long kmers = 0;
    const int pad_bits = 2 * (16 * W - (k + 1));   // 2..32
    const bool even = !((k + 1) & 1);
    // up to 8 ranges counted at once: per lane in registers (an LDS atomic per item from 64 lanes on 3 addresses serialises)
    const bool few = multi && a.multi_n <= 8;
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t m_edge[W], m_k[W], m_k1[W];                             // the first k+1 / k / k-1 characters
    keep_masks<W>(k + 1, m_edge);
    keep_masks<W>(k, m_k);
    keep_masks<W>(k - 1, m_k1);
    // every word a lane of this workgroup can ask for lies inside the array (all but the workgroups of the array's last reads)
    const bool in_bounds = wave_uniform((s_start[r1 - r0] >> 4) + (uint64_t)(W + 1)) < a.n_words;

    // one round of 64 positions: lane = position p of read r (s0 = its first character, npos = its positions); `active` lanes have one
    auto positions = [&](bool active, uint64_t r, uint64_t s0, int p, int npos) {
        bool run_first = p == 0, run_last = p == npos - 1;
        if (active && a.is_solid && r < a.n_short) {               // solid runs inside the read (s2.cpp:276,280,288)
            auto sol = [&](int pp) {
                uint64_t bit = (uint64_t)a.num_k1_per_read * r + (uint64_t)pp;
                return (bool)((a.is_solid[bit >> 6] >> (bit & 63)) & 1);
            };
            active = sol(p);
            if (active) {
                run_first = p == 0 || !sol(p - 1);
                run_last = p == npos - 1 || !sol(p + 1);
            }
        }
        int cnt = 0;
        uint32_t fields = 0;                                       // count mode, few ranges: items of this position per range, 4 bits each
        uint32_t mask = 0;                                         // write mode: bit t = item type t of the position is wanted
        uint32_t e[W], rc[W];
        if (active) {
            // edge = characters [p, p+k] of the read
            uint64_t q = s0 + (uint64_t)p;
            uint64_t wi = q >> 4;
            int sh = (int)(q & 15) * 2;
            uint32_t raw[W + 1];
            if (in_bounds) {
#pragma unroll
                for (int j = 0; j <= W; ++j) raw[j] = a.packed[wi + j];
            } else {
#pragma unroll
                for (int j = 0; j <= W; ++j) raw[j] = (wi + j < a.n_words) ? a.packed[wi + j] : 0u;
            }
#pragma unroll
            for (int j = 0; j < W; ++j) e[j] = (uint32_t)(((((uint64_t)raw[j]) << 32) | (uint64_t)raw[j + 1]) >> (32 - sh)) & m_edge[j];
            // reverse complement (MegahitKmer::ReverseComplement, megahit_kmer.h:115-174)
#pragma unroll
            for (int j = 0; j < W; ++j) rc[j] = rev_chars(~e[W - 1 - j]);
            shl_bits<W>(rc, pad_bits);
            bool pal = even;                                       // s2.cpp:278 (an edge of odd length is never its own reverse complement)
            if (even) {
#pragma unroll
                for (int j = 0; j < W; ++j) pal = pal && (e[j] == rc[j]);
            }
            // the bucket (first 8 characters of the key, s2.cpp:832) is known before the key is built: most keys of a
            // narrow bucket range (memory-bound passes, multi-GPU shares) are dropped after three instructions
            auto push = [&](int t, const uint32_t (&src)[W], int from) {
                const uint32_t b = ((src[0] << (2 * from)) >> 16);   // characters [from, from + 8), from <= 2
                if (b < a.b_lo || b >= a.b_hi) return;
                if (WRITE) mask |= 1u << t;
                else if (multi) {
                    const uint32_t x = b - a.b_lo;                  // (b - b_lo) / multi_width: both at most 2^16
                    uint32_t range;
                    if (magic24) asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(range) : "v"(x), "v"((uint32_t)multi_magic));   // full rate (v_mul_hi_u32: a quarter)
                    else range = (uint32_t)(((uint64_t)x * multi_magic) >> 32);
                    if (few) fields += 1u << (4 * range);
                    else atomicAdd(&s_range_cnt[range], 1u);
                }
                ++cnt;
            };
            if (run_first) {                                       // left $  (s2.cpp:531-540)
                push(0, e, 0);
                if (!pal) push(1, rc, 2);
            }
            push(2, e, 1);                                         // solid   (s2.cpp:543-550)
            if (!pal) push(3, rc, 1);
            if (run_last) {                                        // right $ (s2.cpp:553-562)
                push(4, e, 2);
                if (!pal) push(5, rc, 0);
            }
        }
        if (!WRITE) {
            my_count += (uint32_t)cnt;
            if (few) {
#pragma unroll
                for (int g = 0; g < 8; ++g)
                    if (g < (int)a.multi_n) acc[g] += (fields >> (4 * g)) & 15u;
            }
        } else {
            const uint32_t inc = wave_incl_scan((uint32_t)cnt);
            const uint32_t tot = __shfl(inc, 63, 64);
            if (tot) {
                uint32_t wbase = 0;
                if (lane == 0) wbase = atomicAdd(&s_cursor, tot);
                const uint64_t first = base + (uint64_t)wave_uniform(wbase);   // the wave's slots [first, first + tot): a scalar
                char *const wave_out = reinterpret_cast<char *>(out + first);
                uint8_t *const wave_side = a.side ? a.side + first : nullptr;
                uint32_t slot = inc - (uint32_t)cnt;                    // < 6 * 64
                // every wanted item straight to its slot (no staging array: a lane-varying index into one costs a waterfall loop per store).
                // key = characters [from, from + n) of src, flags in the low 4 bits: (n == k) << 3 | prev   [cx1_read2sdbg_s2.cpp:613-671]
                auto put = [&](int t, const uint32_t (&src)[W], int from, const uint32_t (&keep)[W], uint32_t flags) {
                    if (!((mask >> t) & 1u)) return;
                    Key<W> key;
#pragma unroll
                    for (int j = 0; j < W; ++j) {
                        const uint32_t nx = j + 1 < W ? src[j + 1 < W ? j + 1 : j] : 0u;
                        key.w[j] = (from ? ((src[j] << (2 * from)) | (nx >> (32 - 2 * from))) : src[j]) & keep[j];
                    }
                    key.w[W - 1] |= flags;
                    *reinterpret_cast<Key<W> *>(wave_out + slot * (uint32_t)sizeof(Key<W>)) = key;
                    if (wave_side) wave_side[slot] = (uint8_t)((key.w[0] - a.side_bias) >> a.side_shift);
                    ++slot;
                };
                const uint32_t e0 = e[0] >> 30, e1 = (e[0] >> 28) & 3u, r0c = rc[0] >> 30, r1c = (rc[0] >> 28) & 3u;
                put(0, e, 0, m_k, 8u | (uint32_t)kDollar);
                put(1, rc, 2, m_k1, r1c);
                put(2, e, 1, m_k, 8u | e0);
                put(3, rc, 1, m_k, 8u | r0c);
                put(4, e, 2, m_k1, e1);
                put(5, rc, 0, m_k, 8u | (uint32_t)kDollar);
            }
        }
    };
    // The reads of a wave all of one length (the common case) and no solid bits: their positions are numbered through, 64 per round --
    // reads of 150 bp at k = 44 have 106 positions: 26.5 rounds for the wave's 16 reads instead of 32 (the second round of a read of its
    // own has 42 of 64 lanes at work).  Else: a read at a time.
    constexpr int kWaves = kScanBlock / 64, kReadsPerWave = kReadsPerBlock / kWaves;
    const uint32_t n_mine = r0 + wv < r1 ? (uint32_t)((r1 - r0 - wv + kWaves - 1) / kWaves) : 0u;   // reads r0 + wv + kWaves i, i < n_mine
    uint32_t my_len = 0;
    if ((uint32_t)lane < n_mine) my_len = (uint32_t)(s_start[wv + kWaves * lane + 1] - s_start[wv + kWaves * lane]);
    const uint32_t len0 = wave_uniform(my_len);
    const bool uniform = !a.is_solid && n_mine > 0 && len0 >= (uint32_t)(k + 1) && len0 - (uint32_t)k <= 4096u &&
                         __ballot((uint32_t)lane < n_mine && my_len != len0) == 0;
    static_assert(kReadsPerWave <= 64, "one lane per read of the wave");
    if (uniform) {
        const uint32_t npos = len0 - (uint32_t)k, total = npos * n_mine;     // <= 4096 * 16
        const uint32_t magic = (uint32_t)(((1ull << 32) + npos - 1) / npos);  // idx / npos for idx < 2^16 (npos = 1: magic wraps, handled)
        if (lane == 0) kmers += (unsigned long long)total;
        for (uint32_t v = 0; v < total; v += 64) {
            const uint32_t idx = v + (uint32_t)lane;
            const bool active = idx < total;
            const uint32_t i = npos == 1 ? idx : __umulhi(idx, magic), p = idx - i * npos;
            const uint32_t slot = active ? wv + kWaves * i : wv;           // read r0 + slot
            positions(active, r0 + slot, s_start[slot], (int)p, (int)npos);
        }
    } else {
        for (uint64_t r = r0 + wv; r < r1; r += kWaves) {
            const uint64_t s0 = s_start[r - r0];
            const int len = (int)(s_start[r - r0 + 1] - s0);
            if (len < k + 1) continue;                                     // s2.cpp:262-264
            const int npos = len - k;
            if (lane == 0) kmers += (unsigned long long)npos;
            for (int c0 = 0; c0 < npos; c0 += 64) positions(c0 + lane < npos, r, s0, c0 + lane, npos);
        }
    }
    if (!WRITE) {
        my_count = wave_sum(my_count);
        if (few) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g < (int)a.multi_n) {
                    const uint32_t t = wave_sum(acc[g]);
                    if (lane == 0 && t) atomicAdd(&s_range_cnt[g], t);
                }
            }
        }
        if (lane == 0) s_wave_cnt[wv] = my_count;
        __syncthreads();
        if (threadIdx.x == 0 && !multi) {
            uint32_t t = 0;
            for (int w = 0; w < kScanBlock / 64; ++w) t += s_wave_cnt[w];
            a.block_count[blockIdx.x] = t;
        }
        if (multi && threadIdx.x < a.multi_n) a.block_count[(uint64_t)threadIdx.x * gridDim.x + blockIdx.x] = s_range_cnt[threadIdx.x];
        if (lane == 0 && kmers && a.n_kmers) atomicAdd(a.n_kmers, kmers);
    }
}

// Key generation when the item layout is known in closed form (every position solid; k+1 odd: no palindromes, k+1 even: see below; every bucket
// wanted): read r owns 2 npos + 4 consecutive keys [left $ of e, left $ of rc, then (e, rc) of every position, right $ of e,
// right $ of rc], so every lane stores its two keys at a fixed place: no staging, no prefix sums, fully coalesced 24-byte pairs.
// With k+1 even a (k+1)-mer can be its own reverse complement; the reference then emits the forward items only (s2.cpp:278).  The
// layout stays closed: the rc slot of such a position takes a SENTINEL key (all ones: larger than any real key, whose low three bits
// are a character code <= 4), the sentinels are counted, end up behind every real key in the sorted array and the emitter stops
// before them.
template <int W>
__global__ __launch_bounds__(kScanBlock) void item_write_closed_kernel(ScanArgs a) {
    __shared__ uint32_t s_read_base[kReadsPerBlock];
    __shared__ uint64_t s_start[kReadsPerBlock + 1];                  // one coalesced load instead of two dependent ones per read
    const int k = a.k;
    const int lane = lane_id(), wv = wave_id();
    const uint64_t r0 = (uint64_t)blockIdx.x * kReadsPerBlock;
    const uint64_t r1 = r0 + kReadsPerBlock < a.n_reads ? r0 + kReadsPerBlock : a.n_reads;
    if (wv == 0) {                                                    // first key of every read of the workgroup
        const uint64_t st = r0 + lane <= r1 ? a.start[r0 + lane] : 0, st_next = r0 + lane < r1 ? a.start[r0 + lane + 1] : st;
        s_start[lane] = st;
        if (r0 + lane + 1 == r1) s_start[lane + 1] = st_next;
        uint32_t items = 0;
        if (r0 + lane < r1) {
            const int len = (int)(st_next - st);
            if (len >= k + 1) items = 2u * (uint32_t)(len - k) + 4u;
        }
        s_read_base[lane] = wave_incl_scan(items) - items;
    }
    __syncthreads();
    Key<W> *out = reinterpret_cast<Key<W> *>(a.out) + a.block_base[blockIdx.x];
    const int pad_bits = 2 * (16 * W - (k + 1));
    for (uint64_t r = r0 + wv; r < r1; r += kScanBlock / 64) {
        const uint64_t s0 = s_start[r - r0];
        const int len = (int)(s_start[r - r0 + 1] - s0);
        if (len < k + 1) continue;
        const int npos = len - k;
        Key<W> *ro = out + s_read_base[r - r0];
        for (int c0 = 0; c0 < npos; c0 += 64) {
            const int p = c0 + lane;
            if (p >= npos) continue;
            const uint64_t q = s0 + (uint64_t)p, wi = q >> 4;
            const int sh = (int)(q & 15) * 2;
            uint32_t raw[W + 1], e[W], rc[W];
#pragma unroll
            for (int j = 0; j <= W; ++j) raw[j] = (wi + j < a.n_words) ? a.packed[wi + j] : 0u;
#pragma unroll
            for (int j = 0; j < W; ++j) e[j] = sh ? ((raw[j] << sh) | (raw[j + 1] >> (32 - sh))) : raw[j];
            keep_chars<W>(e, k + 1);
#pragma unroll
            for (int j = 0; j < W; ++j) rc[j] = rev_chars(~e[W - 1 - j]);
            shl_bits<W>(rc, pad_bits);
            const int e0 = e[0] >> 30, e1 = (e[0] >> 28) & 3, r0c = rc[0] >> 30, r1c = (rc[0] >> 28) & 3;
            bool pal = false;                                           // s2.cpp:278 (only possible when k+1 is even)
            if (!((k + 1) & 1)) {
                pal = true;
#pragma unroll
                for (int j = 0; j < W; ++j) pal = pal && e[j] == rc[j];
            }
            Key<W> sentinel;
#pragma unroll
            for (int j = 0; j < W; ++j) sentinel.w[j] = ~0u;
            ro[2 + 2 * p] = make_key<W>(e, 1, k, k, e0);                 // solid   (s2.cpp:543-550)
            ro[3 + 2 * p] = pal ? sentinel : make_key<W>(rc, 1, k, k, r0c);
            if (p == 0) {                                               // left $  (s2.cpp:531-540)
                ro[0] = make_key<W>(e, 0, k, k, kDollar);
                ro[1] = pal ? sentinel : make_key<W>(rc, 2, k - 1, k, r1c);
            }
            if (p == npos - 1) {                                        // right $ (s2.cpp:553-562)
                ro[2 + 2 * npos] = make_key<W>(e, 2, k - 1, k, e1);
                ro[3 + 2 * npos] = pal ? sentinel : make_key<W>(rc, 0, k, k, kDollar);
            }
            const unsigned long long pm = __ballot(pal);                 // rare: one atomic per wave that saw any
            if (pm && lane == __ffsll((long long)pm) - 1) {
                unsigned long long c = 0;
                for (unsigned long long m = pm; m; m &= m - 1) {
                    const int l = __ffsll((long long)m) - 1, pp = c0 + l;
                    c += 1ull + (pp == 0) + (pp == npos - 1);
                }
                atomicAdd(a.n_sentinel, c);
            }
        }
    }
}

// Items per workgroup of item_scan_kernel in closed form.  With k+1 odd no (k+1)-mer equals its reverse complement (with k+1 even the
// slot of the missing item takes a sentinel key, item_write_closed_kernel), so when
// every position is solid and every bucket is wanted a read with npos = len - k >= 1 positions yields exactly
// 2 npos + 4 items (two per position, two more at each end of the read): no edge has to be built to count them.
__global__ __launch_bounds__(256) void item_count_closed_kernel(const uint64_t *start, uint64_t n_reads, uint64_t n_blocks, int k,
                                                                uint32_t *block_count, unsigned long long *n_kmers) {
    // a wave per 16 consecutive workgroups of item_scan_kernel (64 reads each), one lane per read
    unsigned long long kmers = 0;
    for (int q = 0; q < 16; ++q) {
        const uint64_t blk = ((uint64_t)blockIdx.x * 4 + wave_id()) * 16 + q;
        if (blk >= n_blocks) break;
        const uint64_t r = blk * kReadsPerBlock + lane_id();
        uint32_t items = 0, npos = 0;
        if (r < n_reads) {
            const int len = (int)(start[r + 1] - start[r]);
            if (len >= k + 1) { npos = (uint32_t)(len - k); items = 2 * npos + 4; }
        }
        items = wave_sum(items);
        kmers += wave_sum(npos);
        if (lane_id() == 0 && block_count) block_count[blk] = items;   // (nullptr: only the k-mer total is wanted)
    }
    if (lane_id() == 0 && n_kmers && kmers) atomicAdd(n_kmers, kmers);
}

// ---------------------------------------------------------------------------------------------
// 4. LSD radix sort (8-bit digits)
// ---------------------------------------------------------------------------------------------
constexpr int kSortThreads = 1024;                   // 16 waves
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kItemsPerThread = 4;
constexpr int kSubTile = kSortThreads * kItemsPerThread;   // 4096 keys staged in LDS at a time
constexpr int kSubTilesPerBlock = 8;
constexpr int kBlockTile = kSubTile * kSubTilesPerBlock;   // 32768 keys per workgroup

struct Digit {
    int pos;     // bit position counted from the least significant bit of the whole key
    int bits;    // <= 8
    uint32_t bias;   // subtracted from key word 0 before the bits are taken (0 except for the global passes of a bucket sub-range build,
                     // where word 0 - (first bucket << 16) has leading zero bits that the digits skip)
};

template <int W, bool BIASED = false>          // BIASED: honour d.bias (only the global passes of a sub-range build set it)
__device__ __forceinline__ uint32_t get_digit(const Key<W> &key, Digit d) {
    // the word is picked by a wave-uniform branch on compile-time indices: a run-time index into key.w[] would push every
    // register-resident key of the caller into scratch memory, and a chain of selects costs 2W VALU ops per key
    // (the empty asm keeps the compiler from turning the branches back into selects)
    const int wi = W - 1 - (d.pos >> 5), off = d.pos & 31;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        if (j == wi) {
            lo = key.w[j] - (BIASED && j == 0 ? d.bias : 0u);
            hi = j > 0 ? key.w[j - 1] - (BIASED && j == 1 ? d.bias : 0u) : 0u;
            asm volatile("" : "+v"(lo), "+v"(hi));
        }
    }
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)off) & ((1u << d.bits) - 1u);
}

// the same digit of N register-resident keys: one wave-uniform branch chain for all of them
template <int W, int N, bool BIASED = false>
__device__ __forceinline__ void get_digits(const Key<W> (&key)[N], Digit d, uint32_t (&dg)[N]) {
    const int wi = W - 1 - (d.pos >> 5), off = d.pos & 31;
    uint32_t lo[N], hi[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { lo[i] = 0; hi[i] = 0; }
#pragma unroll
    for (int j = 0; j < W; ++j) {
        if (j == wi) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                lo[i] = key[i].w[j] - (BIASED && j == 0 ? d.bias : 0u);
                hi[i] = j > 0 ? key[i].w[j - 1] - (BIASED && j == 1 ? d.bias : 0u) : 0u;
                asm volatile("" : "+v"(lo[i]), "+v"(hi[i]));
            }
        }
    }
    const uint32_t mask = (1u << d.bits) - 1u;
#pragma unroll
    for (int i = 0; i < N; ++i) dg[i] = __builtin_amdgcn_alignbit(hi[i], lo[i], (uint32_t)off) & mask;
}

// tile of workgroup b of n when workgroup b runs on XCD b % 8 (MGTA_XCD_TILES: 1 = contiguous eighths per XCD, 0 = tile b)
#ifndef MGTA_XCD_TILES
#define MGTA_XCD_TILES 1
#endif
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t n) {
    if (!MGTA_XCD_TILES || n < 64) return b;
    const uint32_t x = b & 7u, idx = b >> 3, q = n >> 3, r = n & 7u;
    return x * q + (x < r ? x : r) + idx;
}

// census: hist[digit * n_tiles + tile]
template <int W>
__global__ __launch_bounds__(kSortThreads) void radix_census_kernel(const Key<W> *keys, uint64_t n, Digit d, uint64_t n_tiles,
                                                                     uint64_t *hist) {
    __shared__ uint32_t h[256];
    for (int i = threadIdx.x; i < 256; i += kSortThreads) h[i] = 0;
    __syncthreads();
    const uint32_t tile = xcd_tile(blockIdx.x, (uint32_t)n_tiles);     // (neighbouring tiles' counters share lines: one L2 writes them)
    uint64_t base = (uint64_t)tile * kBlockTile;
    int wi = W - 1 - (d.pos >> 5), off = d.pos & 31;
    bool straddle = off + d.bits > 32 && wi > 0;
    uint32_t mask = (1u << d.bits) - 1u;
    for (int i = 0; i < kBlockTile / kSortThreads; ++i) {
        uint64_t idx = base + (uint64_t)i * kSortThreads + threadIdx.x;
        if (idx < n) {
            uint32_t v = (keys[idx].w[wi] - (wi == 0 ? d.bias : 0u)) >> off;
            if (straddle) v |= (keys[idx].w[wi - 1] - (wi == 1 ? d.bias : 0u)) << (32 - off);
            atomicAdd(&h[v & mask], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += kSortThreads) hist[(uint64_t)i * n_tiles + tile] = h[i];
}

// The same census from the SIDE array of the previous scatter (one byte per key: the digit this pass sorts on, written next to the
// key at its destination): 1 byte read per key instead of the whole key (W = 3: 7.2 GB instead of 86 GB per launch at 100 M reads).
__global__ __launch_bounds__(kSortThreads) void radix_census_side_kernel(const uint8_t *side, uint64_t n, uint64_t n_tiles, uint64_t *hist) {
    __shared__ uint32_t h[256];
    for (int i = threadIdx.x; i < 256; i += kSortThreads) h[i] = 0;
    __syncthreads();
    const uint32_t tile = xcd_tile(blockIdx.x, (uint32_t)n_tiles);
    const uint64_t base = (uint64_t)tile * kBlockTile;                       // a multiple of 32768: 16-byte loads are aligned
    constexpr int kPerThread = kBlockTile / kSortThreads;                    // 32 bytes = two 16-byte loads, consecutive threads consecutive
#pragma unroll
    for (int half = 0; half < kPerThread / 16; ++half) {
        const uint64_t idx = base + (uint64_t)half * (kSortThreads * 16) + (uint64_t)threadIdx.x * 16;
        if (idx + 16 <= n) {
            const uint4 v = *reinterpret_cast<const uint4 *>(side + idx);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                atomicAdd(&h[w[j] & 255u], 1u);
                atomicAdd(&h[(w[j] >> 8) & 255u], 1u);
                atomicAdd(&h[(w[j] >> 16) & 255u], 1u);
                atomicAdd(&h[w[j] >> 24], 1u);
            }
        } else {
            for (uint64_t i = idx; i < n && i < idx + 16; ++i) atomicAdd(&h[side[i]], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += kSortThreads) hist[(uint64_t)i * n_tiles + tile] = h[i];
}

// Closed-form key generation (see item_write_closed_kernel) cut along the census tiles of the first global sort pass: workgroup t
// writes exactly the keys [t kBlockTile, (t+1) kBlockTile) and counts their first-pass digit in LDS on the way, so the keys are not
// read back for that census (one of the 11 HBM crossings of the key array).  A read owns npos + 2 PAIRS of keys (left $ pair, one pair
// per position, right $ pair); read bases and tile bounds are even, so a pair never straddles a tile: lane = pair, 24 contiguous
// bytes per lane, consecutive lanes consecutive pairs.  The reads of a tile are found from block_base (first key of every 64 reads).
template <int W>
__global__ __launch_bounds__(kScanBlock) void item_write_tiled_kernel(ScanArgs a, uint64_t n_blocks, uint64_t n_items, int digit_shift,
                                                                      uint64_t n_tiles, uint64_t *hist) {
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_read_base[kReadsPerBlock];
    __shared__ uint64_t s_start[kReadsPerBlock + 1];
    const int k = a.k;
    const int lane = lane_id(), wv = wave_id();
    for (int i = threadIdx.x; i < 256; i += kScanBlock) s_hist[i] = 0;
    const uint64_t lo = (uint64_t)blockIdx.x * kBlockTile;
    const uint64_t hi = lo + kBlockTile < n_items ? lo + kBlockTile : n_items;
    // last group of 64 reads whose first key is at or before `lo`: 64-ary search, every wave on its own (block_base[0] = 0 <= lo)
    uint64_t left = 0, right = n_blocks;
    while (right - left > 1) {
        const uint64_t step = (right - left + 63) / 64, idx = left + (uint64_t)lane * step;
        const bool ok = idx < right && a.block_base[idx] <= lo;
        const unsigned long long m = __ballot(ok);                     // a prefix of lanes: block_base is non-decreasing
        const uint64_t nl = left + (uint64_t)(__popcll(m) - 1) * step;
        right = nl + step < right ? nl + step : right;
        left = nl;
    }
    Key<W> *out = reinterpret_cast<Key<W> *>(a.out);
    const int pad_bits = 2 * (16 * W - (k + 1));
    for (uint64_t b = left; b < n_blocks; ++b) {
        const uint64_t bb = a.block_base[b];
        if (bb >= hi) break;
        const uint64_t r0 = b * kReadsPerBlock;
        const uint64_t r1 = r0 + kReadsPerBlock < a.n_reads ? r0 + kReadsPerBlock : a.n_reads;
        __syncthreads();                                               // the previous group's table is no longer read
        if (wv == 0) {
            const uint64_t st = r0 + lane <= r1 ? a.start[r0 + lane] : 0, st_next = r0 + lane < r1 ? a.start[r0 + lane + 1] : st;
            s_start[lane] = st;
            if (r0 + lane + 1 == r1) s_start[lane + 1] = st_next;
            uint32_t items = 0;
            if (r0 + lane < r1) {
                const int len = (int)(st_next - st);
                if (len >= k + 1) items = 2u * (uint32_t)(len - k) + 4u;
            }
            s_read_base[lane] = wave_incl_scan(items) - items;
        }
        __syncthreads();
        for (uint64_t r = r0 + wv; r < r1; r += kScanBlock / 64) {
            const uint64_t s0 = s_start[r - r0];
            const int len = (int)(s_start[r - r0 + 1] - s0);
            if (len < k + 1) continue;
            const int npos = len - k;
            const uint64_t rb = bb + s_read_base[r - r0], re = rb + 2ull * (uint64_t)npos + 4ull;
            if (re <= lo || rb >= hi) continue;
            const int j_lo = rb >= lo ? 0 : (int)((lo - rb) >> 1), j_hi = re <= hi ? npos + 2 : (int)((hi - rb) >> 1);
            Key<W> *ro = out + rb;
            for (int c0 = j_lo; c0 < j_hi; c0 += 64) {
                const int j = c0 + lane;
                if (j >= j_hi) continue;
                const int p = j == 0 ? 0 : (j == npos + 1 ? npos - 1 : j - 1);
                const uint64_t q = s0 + (uint64_t)p, wi = q >> 4;
                const int sh = (int)(q & 15) * 2;
                uint32_t raw[W + 1], e[W], rc[W];
#pragma unroll
                for (int i = 0; i <= W; ++i) raw[i] = (wi + i < a.n_words) ? a.packed[wi + i] : 0u;
#pragma unroll
                for (int i = 0; i < W; ++i) e[i] = sh ? ((raw[i] << sh) | (raw[i + 1] >> (32 - sh))) : raw[i];
                keep_chars<W>(e, k + 1);
#pragma unroll
                for (int i = 0; i < W; ++i) rc[i] = rev_chars(~e[W - 1 - i]);
                shl_bits<W>(rc, pad_bits);
                const int e0 = e[0] >> 30, e1 = (e[0] >> 28) & 3, r0c = rc[0] >> 30, r1c = (rc[0] >> 28) & 3;
                bool pal = false;                                       // s2.cpp:278: the rc slot takes a sentinel (item_write_closed_kernel)
                if (!((k + 1) & 1)) {
                    pal = true;
#pragma unroll
                    for (int i = 0; i < W; ++i) pal = pal && e[i] == rc[i];
                }
                Key<W> ka, kb;
                if (j == 0) {                                           // left $  (s2.cpp:531-540)
                    ka = make_key<W>(e, 0, k, k, kDollar);
                    kb = make_key<W>(rc, 2, k - 1, k, r1c);
                } else if (j == npos + 1) {                             // right $ (s2.cpp:553-562)
                    ka = make_key<W>(e, 2, k - 1, k, e1);
                    kb = make_key<W>(rc, 0, k, k, kDollar);
                } else {                                                // solid   (s2.cpp:543-550)
                    ka = make_key<W>(e, 1, k, k, e0);
                    kb = make_key<W>(rc, 1, k, k, r0c);
                }
                if (pal) {
#pragma unroll
                    for (int i = 0; i < W; ++i) kb.w[i] = ~0u;
                }
                const unsigned long long pm = __ballot(pal);
                if (pm && lane == __ffsll((long long)pm) - 1) atomicAdd(a.n_sentinel, (unsigned long long)__popcll(pm));
                ro[2 * j] = ka;
                ro[2 * j + 1] = kb;
                atomicAdd(&s_hist[(ka.w[0] >> digit_shift) & 255u], 1u);
                atomicAdd(&s_hist[(kb.w[0] >> digit_shift) & 255u], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += kScanBlock) hist[(uint64_t)i * n_tiles + blockIdx.x] = s_hist[i];
}

// one workgroup per digit value: exclusive scan of its row of tile counts, in place; row total -> totals[digit]
__global__ __launch_bounds__(1024) void radix_rowscan_kernel(uint64_t *hist, uint64_t n_tiles, uint64_t *totals) {
    __shared__ uint64_t scratch[1024 / 64 + 1];
    __shared__ uint64_t carry_s;
    uint64_t *row = hist + (uint64_t)blockIdx.x * n_tiles;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint64_t b = 0; b < n_tiles; b += 1024) {
        uint64_t idx = b + threadIdx.x;
        uint64_t x = idx < n_tiles ? row[idx] : 0, tot;
        uint64_t ex = block_excl_scan64<1024>(x, scratch, &tot);
        uint64_t carry = carry_s;
        if (idx < n_tiles) row[idx] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

// keys staged in LDS per round of the scatter: 4096, or 2048 for the 10- and 11-word records of stage 1 at k > 110 (a 4096-key stage
// of those would not fit the 160 KB); a tile stays 32768 keys either way (census and scatter agree on that)
#ifndef MGTA_SCATTER_IPT_NARROW
#define MGTA_SCATTER_IPT_NARROW kItemsPerThread                  // (experiment builds: keys per thread of the staged sub-tile for W <= 3)
#endif
template <int W> struct ScatterCfg {
    static constexpr int kIpt = W >= 10 ? 2 : (W <= 3 ? MGTA_SCATTER_IPT_NARROW : kItemsPerThread);
    static constexpr int kSub = kSortThreads * kIpt;
    static constexpr int kChunk = kSub / kSortWaves;
};

// shared state of one scatter workgroup
template <int W>
struct ScatterShared {
    Key<W> keys[ScatterCfg<W>::kSub];
    uint16_t whist[kSortWaves][256];   // per wave: running count, then position of the wave's first key of the digit value in the sorted sub-tile
    uint64_t gbase[256];               // global destination of the next key of each digit value
    uint64_t gdelta[256];              // global destination minus position in the sorted sub-tile
    uint32_t scratch[kSortThreads / 64 + 1];
};

// stable scatter of in[0..n) by digit d to out[gbase[digit]++...], sub-tile by sub-tile (gbase must be set and visible; all threads
// call).  Four workgroup barriers per sub-tile; the next sub-tile's keys are on their way while the current one is placed.
// SIDE: the digit the NEXT pass sorts on (d_next) of every key, one byte per key, for that pass's census.  The bytes of a 32768-key
// tile are collected in LDS in the order the tile's keys land in (per digit value the eight sub-tiles append to ONE run of ~128 keys)
// and written run by run when the tile is done: byte stores straight from the sub-tiles (runs of ~16 bytes, four per wave
// instruction) cost the scatter 21 % (100 M reads: 43.9 -> 53.4 ms per launch).
struct SideShared {
    uint8_t bytes[kBlockTile];
    uint64_t gtile[256];               // global destination of the tile's first key of a digit value minus that key's place in `bytes`
    uint32_t toff[257];                // place in `bytes` of the tile's first key of a digit value
};

// STABLE = false: keys of one digit value may leave in any order -- all the FIRST pass of an LSD sort needs (nothing is ordered yet).
// The rank inside the wave's chunk then comes from an LDS atomic on the wave's counter instead of the eight ballots of wave_match,
// which are 45 % of the stable kernel's vector instructions (54 of ~120 per key).
template <int W, bool BIASED = false, bool SIDE = false, bool STABLE = true>
__device__ __forceinline__ void scatter_subtiles(ScatterShared<W> &sh, const Key<W> *in, Key<W> *out, uint64_t n, Digit d, SideShared *ss = nullptr,
                                                 Digit d_next = Digit{0, 0, 0}) {
    constexpr int kItemsPerThread = ScatterCfg<W>::kIpt, kSubTile = ScatterCfg<W>::kSub, kWaveChunk = ScatterCfg<W>::kChunk;   // (shadow the 4096-key constants)
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    uint16_t *whist = sh.whist[wv];               // wave-private row, updated lane-to-lane inside the wave
    for (int i = lane; i < 256; i += 64) whist[i] = 0;
    constexpr bool kPrefetch = W * kItemsPerThread <= 16;   // wider keys: the registers are better spent on the keys in flight
    Key<W> key[kItemsPerThread], nxt[kItemsPerThread];
    auto load = [&](Key<W> (&dst)[kItemsPerThread], uint64_t sub_base) {   // wave w owns keys [w*512, w*512+512) of the sub-tile, 64 at a time
#pragma unroll
        for (int it = 0; it < kItemsPerThread; ++it) {
            uint64_t idx = sub_base + (uint32_t)wv * kWaveChunk + (uint32_t)it * 64 + (uint32_t)lane;
            if (idx < n) dst[it] = in[idx];
        }
    };
    load(key, 0);
    for (uint64_t sub_base = 0; sub_base < n; sub_base += kSubTile) {
        uint32_t n_valid = (uint32_t)((n - sub_base) < (uint64_t)kSubTile ? (n - sub_base) : (uint64_t)kSubTile);
        // phase 1: rank inside the wave chunk: digit + peers of every key, then the wave's running counts round by round
        uint32_t dr[kItemsPerThread], cnt[kItemsPerThread];   // digit | rank-in-wave-chunk << 8 | valid << 31
        get_digits<W, kItemsPerThread, BIASED>(key, d, dr);
        if constexpr (STABLE) {
#pragma unroll
            for (int it = 0; it < kItemsPerThread; ++it) {
                uint32_t j = (uint32_t)wv * kWaveChunk + (uint32_t)it * 64 + (uint32_t)lane;
                bool valid = j < n_valid;
                uint32_t dg = valid ? dr[it] : 0u, rank;
                wave_match(dg, d.bits, valid, rank, cnt[it]);
                dr[it] = dg | (rank << 8) | ((uint32_t)valid << 31);
            }
#pragma unroll
            for (int it = 0; it < kItemsPerThread; ++it) {
                if (dr[it] >> 31) {
                    const uint32_t dg = dr[it] & 255u, rank = (dr[it] >> 8) & 0xFFu, prev = whist[dg];
                    if (rank == cnt[it] - 1) whist[dg] = (uint16_t)(prev + cnt[it]);   // highest peer lane publishes the new count
                    dr[it] += prev << 8;
                }
                wave_lds_fence();
            }
        } else {
            uint32_t *const wh32 = reinterpret_cast<uint32_t *>(whist);      // two 16-bit counters per word: a wave's chunk holds <= 512 keys
#pragma unroll
            for (int it = 0; it < kItemsPerThread; ++it) {
                const uint32_t j = (uint32_t)wv * kWaveChunk + (uint32_t)it * 64 + (uint32_t)lane;
                const bool valid = j < n_valid;
                const uint32_t dg = valid ? dr[it] : 0u;
                uint32_t taken = 0;
                if (valid) taken = (atomicAdd(&wh32[dg >> 1], 1u << (16u * (dg & 1u))) >> (16u * (dg & 1u))) & 0xFFFFu;   // keys of the value before this one, in any order
                dr[it] = dg | (taken << 8) | ((uint32_t)valid << 31);
            }
            wave_lds_fence();
        }
        if (kPrefetch && sub_base + kSubTile < n) load(nxt, sub_base + kSubTile);
        __syncthreads();
        // phase 2: per digit value (threads 0..255 = waves 0..3): total over the waves, scan over the values, wave bases, global bases
        uint32_t tot = 0, inc = 0;
        if (wv < 4) {
#pragma unroll
            for (int w = 0; w < kSortWaves; ++w) tot += sh.whist[w][tid];
            inc = wave_incl_scan(tot);
            if (lane == 63) sh.scratch[wv] = inc;
        }
        __syncthreads();
        if (wv < 4) {
            uint32_t running = inc - tot;
            for (int w = 0; w < wv; ++w) running += sh.scratch[w];
            const uint64_t g = sh.gbase[tid];
            sh.gdelta[tid] = g - running;
            sh.gbase[tid] = g + tot;
#pragma unroll
            for (int w = 0; w < kSortWaves; ++w) { uint32_t c = sh.whist[w][tid]; sh.whist[w][tid] = (uint16_t)running; running += c; }
        }
        __syncthreads();
        // phase 3: place the keys in sorted order in LDS
#pragma unroll
        for (int it = 0; it < kItemsPerThread; ++it) {
            if (dr[it] >> 31) {
                const uint32_t pos = whist[dr[it] & 255u] + ((dr[it] >> 8) & 0x7FFFFFu);
#pragma unroll
                for (int w = 0; w < W; ++w) sh.keys[pos].w[w] = key[it].w[w];
            }
        }
        __syncthreads();
        for (int i = lane; i < 256; i += 64) whist[i] = 0;                // own row, read by this wave only since the last barrier
        // phase 4: stream the sorted sub-tile out; consecutive threads hit consecutive addresses inside a run
        if (kPrefetch) {
            Key<W> kk[kItemsPerThread];
            uint32_t dg[kItemsPerThread];
#pragma unroll
            for (int it = 0; it < kItemsPerThread; ++it) {
                uint32_t j = (uint32_t)it * kSortThreads + (uint32_t)tid;
                if (j < n_valid) kk[it] = sh.keys[j];
            }
            get_digits<W, kItemsPerThread, BIASED>(kk, d, dg);
            if constexpr (SIDE) {
                uint32_t dn[kItemsPerThread];
                get_digits<W, kItemsPerThread, BIASED>(kk, d_next, dn);
#pragma unroll
                for (int it = 0; it < kItemsPerThread; ++it) {
                    uint32_t j = (uint32_t)it * kSortThreads + (uint32_t)tid;
                    if (j < n_valid) {
                        const uint64_t to = sh.gdelta[dg[it]] + j;
                        out[to] = kk[it];
                        ss->bytes[(uint32_t)(to - ss->gtile[dg[it]])] = (uint8_t)dn[it];
                    }
                }
            } else {
#pragma unroll
                for (int it = 0; it < kItemsPerThread; ++it) {
                    uint32_t j = (uint32_t)it * kSortThreads + (uint32_t)tid;
                    if (j < n_valid) out[sh.gdelta[dg[it]] + j] = kk[it];
                }
            }
#pragma unroll
            for (int it = 0; it < kItemsPerThread; ++it) key[it] = nxt[it];
        } else {
#pragma unroll
            for (int it = 0; it < kItemsPerThread; ++it) {
                uint32_t j = (uint32_t)it * kSortThreads + (uint32_t)tid;
                if (j < n_valid) {
                    Key<W> kk = sh.keys[j];
                    const uint32_t dg = get_digit<W, BIASED>(kk, d);
                    const uint64_t to = sh.gdelta[dg] + j;
                    out[to] = kk;
                    if constexpr (SIDE) ss->bytes[(uint32_t)(to - ss->gtile[dg])] = (uint8_t)get_digit<W, BIASED>(kk, d_next);
                }
            }
            if (sub_base + kSubTile < n) load(key, sub_base + kSubTile);
        }
    }
}

// stable scatter of one 32768-key tile by the current digit
// side digits only where the tile's bytes fit the LDS next to the staged keys (W <= 7 key words)
template <int W> constexpr bool kSideFits = sizeof(ScatterShared<W>) + sizeof(SideShared) + 1024 <= 160 * 1024;

template <int W, bool BIASED, bool SIDE = false, bool STABLE = true>
__global__ __launch_bounds__(kSortThreads, 4) void radix_scatter_kernel(const Key<W> *in, Key<W> *out, uint64_t n, Digit d,
                                                                      uint64_t n_tiles, const uint64_t *rowoff,
                                                                      const uint64_t *totals, uint8_t *side = nullptr,
                                                                      Digit d_next = Digit{0, 0, 0}) {
    __shared__ ScatterShared<W> sh;
    __shared__ uint64_t s64[kSortThreads / 64 + 1];
    const int tid = threadIdx.x;
    // workgroups go to the 8 XCDs in turn: XCD x takes the x-th eighth of the tiles, so that the tiles that run side by side on one
    // L2 are neighbours (their runs of a digit value are adjacent in the output: lines shared by two tiles are completed in one L2)
    const uint32_t tile = xcd_tile(blockIdx.x, (uint32_t)n_tiles);
    // global base of every digit value for this tile = scan(totals)[digit] + rowoff[digit][tile]
    uint64_t t = tid < 256 ? totals[tid] : 0;
    uint64_t ex = block_excl_scan64<kSortThreads>(t, s64, nullptr);
    const uint64_t my_off = tid < 256 ? rowoff[(uint64_t)tid * n_tiles + tile] : 0;
    if (tid < 256) sh.gbase[tid] = ex + my_off;
    const uint64_t tile_base = (uint64_t)tile * kBlockTile;
    if constexpr (SIDE) {
        __shared__ SideShared ss;
        // keys of the tile per digit value: the next tile's row offset (the row total behind the last tile) minus this tile's
        uint64_t c = 0;
        if (tid < 256) c = (tile + 1 < n_tiles ? rowoff[(uint64_t)tid * n_tiles + tile + 1] : t) - my_off;
        __syncthreads();                                                 // s64 is free again
        const uint64_t place = block_excl_scan64<kSortThreads>(c, s64, nullptr);
        if (tid < 256) {
            ss.toff[tid] = (uint32_t)place;
            ss.gtile[tid] = ex + my_off - place;
            if (tid == 255) ss.toff[256] = (uint32_t)(place + c);
        }
        __syncthreads();
        if (tile_base >= n) return;
        const uint64_t cnt = (n - tile_base) < (uint64_t)kBlockTile ? (n - tile_base) : (uint64_t)kBlockTile;
        scatter_subtiles<W, BIASED, true, STABLE>(sh, in + tile_base, out, cnt, d, &ss, d_next);
        __syncthreads();
        // the tile's bytes, run by run: a wave per 16 digit values, consecutive lanes consecutive bytes
        const int lane = lane_id(), wv = wave_id();
        for (int v = wv * (256 / kSortWaves); v < (wv + 1) * (256 / kSortWaves); ++v) {
            const uint32_t from = ss.toff[v], len = ss.toff[v + 1] - from;
            uint8_t *to = side + ss.gtile[v] + from;
            for (uint32_t i = lane; i < len; i += 64) to[i] = ss.bytes[from + i];
        }
    } else {
        __syncthreads();
        if (tile_base >= n) return;
        const uint64_t cnt = (n - tile_base) < (uint64_t)kBlockTile ? (n - tile_base) : (uint64_t)kBlockTile;
        scatter_subtiles<W, BIASED, false, STABLE>(sh, in + tile_base, out, cnt, d);
    }
}

// One workgroup sorts ONE oversized segment [big[b], big_end[b]) on all the low digits: census, scan and scatter of
// every pass inside the same launch, ping-ponging between the two key buffers (the range is private to the workgroup).
template <int W>
__global__ __launch_bounds__(kSortThreads) void segment_sort_kernel(Key<W> *buf_a, Key<W> *buf_b, const uint64_t *big, const uint64_t *big_end,
                                                                     const Digit *low_plan, int n_low, uint32_t lds_cap) {
    __shared__ ScatterShared<W> sh;
    __shared__ uint64_t s64[kSortThreads / 64 + 1];
    __shared__ uint32_t s_cnt[256];
    const int tid = threadIdx.x;
    const uint64_t s0 = big[blockIdx.x], cnt = big_end[blockIdx.x] - s0;
    if (cnt <= (uint64_t)lds_cap) return;                           // sorted in LDS by local_deferred_kernel
    Key<W> *x = buf_a + s0, *y = buf_b + s0;
    for (int pass = 0; pass < n_low; ++pass) {
        const Digit d = low_plan[pass];
        if (tid < 256) s_cnt[tid] = 0;
        __syncthreads();
        for (uint64_t i = tid; i < cnt; i += kSortThreads) atomicAdd(&s_cnt[get_digit<W>(x[i], d)], 1u);
        __syncthreads();
        uint64_t c = tid < 256 ? s_cnt[tid] : 0;
        uint64_t ex = block_excl_scan64<kSortThreads>(c, s64, nullptr);
        if (tid < 256) sh.gbase[tid] = ex;
        __syncthreads();
        scatter_subtiles<W>(sh, x, y, cnt, d);
        __threadfence();                         // the next pass re-reads what this one wrote (other waves' stores, L1 lines of an older pass)
        __syncthreads();
        Key<W> *t = x; x = y; y = t;
    }
    if (x != buf_a + s0)                         // odd number of passes: bring the result back to the main buffer
        for (uint64_t i = tid; i < cnt; i += kSortThreads) y[i] = x[i];
}

// ---------------------------------------------------------------------------------------------
// 4b. segment-local finish.  After P global passes on the P most significant key bytes the array is
// partitioned into segments of equal 8P-bit prefix (in arbitrary inner order).  Each workgroup takes
// the whole segments that START in its stride of the array (<= LocalCfg<W>::kTile keys) and sorts
// them in LDS, so these keys cross HBM once more instead of once per remaining digit:
//   local_sort_kernel  two or three stable LSD passes in LDS (the 8 key bits below the prefix, then the rank of the
//                      segment inside the tile) leave runs of equal leading 8P+8 bits — a handful of keys unless the
//                      input is highly redundant; every key then finds its rank inside its run by comparing itself
//                      with the run's other keys and goes straight to its final place.
//   local_lsd_kernel   tiles whose runs are too long for that (reported by the first kernel), and single segments
//                      that did not fit a tile next to their neighbours: LSD passes in LDS over every remaining digit.
//   segment_sort_kernel (above)  segments longer than a tile: global passes over just that range.
// ---------------------------------------------------------------------------------------------
template <int W> struct LocalCfg {
    static constexpr int kTile = W <= 4 ? 4096 : 2048;           // keys sorted in LDS by one workgroup (LDS budget: kTile * 4W bytes)
    static constexpr int kIpt = kTile / kSortThreads;
    static constexpr int kChunk = kTile / kSortWaves;
};

// segments just too long for that tile still fit the LDS of a workgroup that has it to itself: twice the tile (one segment, no
// neighbours: segment_lds_kernel); 0 = no such tile for this key width
template <int W> struct BigCfg {
    static constexpr int kTile = W <= 4 ? 8192 : (W <= 8 ? 4096 : 0);
};

template <int W, int TILE = LocalCfg<W>::kTile>
struct LocalShared {
    Key<W> keys[TILE];
    uint16_t seg[TILE];
    uint16_t whist[kSortWaves][256];
    uint32_t start[256];
    uint32_t scratch[kSortThreads / 64 + 1];
    uint32_t wheads[kSortWaves + 1];
};

// LDS of local_sort_kernel: the tile, one 16-bit counter per (segment, upper digit) bin, head masks
template <int W>
struct CompareShared {
    // two workgroups per CU (80 KB each) where the keys leave room for >= 2048 counters, else one
    static constexpr int kRoom = (80 * 1024 - 1024 - LocalCfg<W>::kTile * 4 * W) / 2;
    static constexpr int kBins = kRoom >= 2048 ? (kRoom / 2048 * 2048 < 16384 ? kRoom / 2048 * 2048 : 16384) : 16384;
    static_assert(kBins >= LocalCfg<W>::kTile || kBins == 16384, "a bin per segment at least");
    Key<W> keys[LocalCfg<W>::kTile];
    uint32_t hist[kBins / 2];                                        // bin b: bits [16 (b & 1), +16) of word b >> 1
    uint32_t scratch[kSortThreads / 64 + 1];
    uint32_t flags[2];                                              // runs in the tile, "a run is too long"
    uint64_t masks[LocalCfg<W>::kTile / 64];                         // segment / run heads, one bit per key
};

template <int W>
__device__ __forceinline__ uint32_t key_prefix(const Key<W> &key, int T) { return T == 0 ? 0u : (key.w[0] >> (32 - T)); }   // leading T <= 32 bits

// What the segment-local kernels need besides the keys (host-built, passed by value)
struct LocalPlan {
    const Digit *low_plan;   // every significant digit below the 8P-bit prefix, least significant first
    int n_low;
    int T;                   // bits of the prefix the global passes sorted on (8 per pass, + the leading zero bits a bucket sub-range skips)
    Digit upper;             // the (<= 8) key bits right below the prefix (inside the first two key words)
    uint32_t mask_last2;     // significant bits of key word W-2 / W-1 (stage 1 carries a payload there that must not be compared);
    uint32_t mask_last;      // the digits of low_plan never look at a masked-out bit
    uint32_t stride;         // keys of the array per workgroup of local_sort_kernel (multiple of 64, < kTile): a workgroup owns the segments
                             // that START in its stride; the rest of the tile is room for the last one to hang over
    uint64_t *lsd_list;      // [2 * lsd_cap]: (first, end) of the tiles left to local_lsd_kernel
    uint32_t *lsd_count;
    uint32_t lsd_cap;
    int debug;               // timing experiments only
};

constexpr uint32_t kMaxAvgRun = 40;    // finish by comparison when the runs of equal (prefix, upper digit) are short on average ...
constexpr uint32_t kMaxRun = 256;      // ... and none of them is longer than this

// a < b on words FIRST..W-1, two words per comparison; MASKED: only the bits m2 / m1 of words W-2 / W-1 count
template <int W, int FIRST, bool MASKED>
__device__ __forceinline__ bool key_less(const Key<W> &a, const Key<W> &b, uint32_t m2, uint32_t m1) {
    bool lt = false;
    auto word = [&](const Key<W> &k, int j) {
        uint32_t x = k.w[j];
        if (MASKED && j == W - 1) x &= m1;
        if (MASKED && j == W - 2) x &= m2;
        return x;
    };
#pragma unroll
    for (int j = W - 1; j >= FIRST; j -= 2) {                          // least significant pair first
        if (j - 1 >= FIRST) {
            const uint64_t x = ((uint64_t)word(a, j - 1) << 32) | word(a, j), y = ((uint64_t)word(b, j - 1) << 32) | word(b, j);
            lt = x < y || (x == y && lt);
        } else {
            const uint32_t x = word(a, j), y = word(b, j);
            lt = x < y || (x == y && lt);
        }
    }
    return lt;
}

// loads the tile keys[first, first + nt) in (wave chunk, round, lane) order and ranks the segments inside the tile;
// returns the number of LSD passes the segment rank needs (0: one segment).  All threads call.
template <int W>
__device__ __forceinline__ int tile_load(LocalShared<W> &sh, const Key<W> *keys, uint64_t first, uint32_t nt, int T, Key<W> (&key)[LocalCfg<W>::kIpt],
                                         uint32_t (&seg)[LocalCfg<W>::kIpt]) {
    constexpr int kIpt = LocalCfg<W>::kIpt, kChunk = LocalCfg<W>::kChunk;
    const int lane = lane_id(), wv = wave_id();
    const uint64_t le_mask = lanemask_lt() | (1ull << lane);
    uint32_t running = 0;
#pragma unroll
    for (int it = 0; it < kIpt; ++it) {
        uint32_t j = (uint32_t)wv * kChunk + (uint32_t)it * 64 + (uint32_t)lane;
        bool valid = j < nt, head = false;
        if (valid) {
            key[it] = keys[first + j];
            head = j == 0 || key_prefix<W>(key[it], T) != key_prefix<W>(keys[first + j - 1], T);
        }
        uint64_t bal = __ballot(head);
        seg[it] = running + (uint32_t)__popcll(bal & le_mask);         // heads at positions <= j inside this wave chunk
        running += (uint32_t)__popcll(bal);
    }
    if (lane == 0) sh.wheads[wv] = running;
    __syncthreads();
    uint32_t before = 0, nseg = 0;
    for (int w = 0; w < kSortWaves; ++w) { uint32_t c = sh.wheads[w]; if (w < wv) before += c; nseg += c; }
    before = wave_uniform(before);
    nseg = wave_uniform(nseg);
#pragma unroll
    for (int it = 0; it < kIpt; ++it) seg[it] = seg[it] + before - 1;
    return nseg <= 1 ? 0 : (nseg <= 256 ? 1 : 2);
}

// one stable LSD pass over the tile: keys (+ segment ranks) go from registers (wave-chunk order) to their sorted LDS positions
// [0, nt); with `reload` they come back into registers in position order.  All threads call; every wave's row of sh.whist
// must be zero on entry (tile_zero_counts) and is zero again on return.  Four workgroup barriers.
template <int W, int TILE = LocalCfg<W>::kTile>
__device__ __forceinline__ void tile_zero_counts(LocalShared<W, TILE> &sh) {
    uint16_t *whist = sh.whist[wave_id()];
    for (int i = lane_id(); i < 256; i += 64) whist[i] = 0;             // wave-private: no barrier needed before the wave counts
}

template <int W, int TILE = LocalCfg<W>::kTile>
__device__ __forceinline__ void lds_pass(LocalShared<W, TILE> &sh, Key<W> (&key)[TILE / kSortThreads], uint32_t (&seg)[TILE / kSortThreads], uint32_t nt,
                                         Digit d, bool by_seg, int seg_shift, bool reload, uint32_t off = 0) {
    constexpr int kIpt = TILE / kSortThreads, kChunk = TILE / kSortWaves;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    uint16_t *whist = sh.whist[wv];
    uint32_t dr[kIpt], cnt[kIpt];
    // digit + peers of every key first (independent work the compiler can interleave) ...
    if (by_seg) {
#pragma unroll
        for (int it = 0; it < kIpt; ++it) dr[it] = (seg[it] >> seg_shift) & 255u;
    } else {
        get_digits<W, kIpt>(key, d, dr);
    }
#pragma unroll
    for (int it = 0; it < kIpt; ++it) {
        uint32_t j = (uint32_t)wv * kChunk + (uint32_t)it * 64 + (uint32_t)lane;
        bool valid = j - off < nt;                                        // the registers hold positions [off, off + nt) of what was loaded
        uint32_t dg = valid ? dr[it] : 0u, rank;
        wave_match(dg, by_seg ? 8 : d.bits, valid, rank, cnt[it]);
        dr[it] = dg | (rank << 8) | ((uint32_t)valid << 31);
    }
    // ... then the wave's running counts, round by round (lane-to-lane hand-over through LDS)
#pragma unroll
    for (int it = 0; it < kIpt; ++it) {
        if (dr[it] >> 31) {
            const uint32_t dg = dr[it] & 255u, rank = (dr[it] >> 8) & 0xFFu, prev = whist[dg];
            if (rank == cnt[it] - 1) whist[dg] = (uint16_t)(prev + cnt[it]);
            dr[it] += prev << 8;
        }
        wave_lds_fence();
    }
    __syncthreads();
    // per digit value (threads 0..255 = waves 0..3): total over the waves, scan over the values, base of every wave
    uint32_t tot = 0, inc = 0;
    if (wv < 4) {
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) tot += sh.whist[w][tid];
        inc = wave_incl_scan(tot);
        if (lane == 63) sh.scratch[wv] = inc;
    }
    __syncthreads();
    if (wv < 4) {
        uint32_t running = inc - tot;
        for (int w = 0; w < wv; ++w) running += sh.scratch[w];
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) { uint32_t c = sh.whist[w][tid]; sh.whist[w][tid] = (uint16_t)running; running += c; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kIpt; ++it) {
        if (dr[it] >> 31) {
            uint32_t pos = whist[dr[it] & 255u] + ((dr[it] >> 8) & 0x7FFFFFu);
            sh.keys[pos] = key[it];
            sh.seg[pos] = (uint16_t)seg[it];
        }
    }
    __syncthreads();
    tile_zero_counts<W, TILE>(sh);                                        // own row, read by this wave only since the last barrier
    if (reload) {
#pragma unroll
        for (int it = 0; it < kIpt; ++it) {
            uint32_t j = (uint32_t)wv * kChunk + (uint32_t)it * 64 + (uint32_t)lane;
            if (j < nt) { key[it] = sh.keys[j]; seg[it] = sh.seg[j]; }
        }
    }
}

// Last step of local_sort_kernel.  The tile lies in sh.keys sorted by (segment, upper digit): runs of equal leading
// `32 - run_shift` bits.  Run heads are balloted per 64-key window and the masks shared through LDS, so every key reads its
// run [rs, re) off a few masks; it then finds its rank among the run's keys by comparing words FIRST..W-1 (equal keys keep
// their order) and goes to its final place in global memory.  Returns false (nothing written) when the runs are longer than
// kMaxAvgRun on average or one of them is longer than kMaxRun.
// the leading PW words of a key as one integer (PW = 2: the run prefix reaches into the second word)
template <int W, int PW>
__device__ __forceinline__ uint64_t key_top(const Key<W> &k) {
    return PW == 1 ? (uint64_t)k.w[0] : (((uint64_t)k.w[0] << 32) | (uint64_t)k.w[W > 1 ? 1 : 0]);
}

template <int W, int FIRST, bool MASKED, int PW>
__device__ __forceinline__ bool finish_by_comparison(CompareShared<W> &sh, Key<W> *keys, uint64_t first, uint32_t nt, int run_bits, uint32_t m2,
                                                     uint32_t m1, bool skip_compare) {
    const int run_shift = 32 * PW - run_bits;                          // runs = equal leading run_bits bits
    constexpr int kIpt = LocalCfg<W>::kIpt, kWindows = LocalCfg<W>::kTile / 64;
    const int lane = lane_id(), wv = wave_id();
    const uint64_t le_mask = lanemask_lt() | (1ull << lane);
    uint32_t heads = 0;
#pragma unroll
    for (int it = 0; it < kIpt; ++it) {
        const uint32_t win = (uint32_t)it * kSortWaves + (uint32_t)wv, j = win * 64 + (uint32_t)lane;
        const bool head = j < nt && (j == 0 || ((key_top<W, PW>(sh.keys[j - 1]) ^ key_top<W, PW>(sh.keys[j])) >> run_shift) != 0);
        const uint64_t hm = __ballot(head);
        if (lane == 0) sh.masks[win] = hm;
        heads += (uint32_t)__popcll(hm);
    }
    if (lane == 0 && heads) atomicAdd(&sh.flags[0], heads);
    __syncthreads();
    if (nt > wave_uniform(sh.flags[0]) * kMaxAvgRun) return false;
    uint32_t run[kIpt];                                               // rs | (re - rs) << 16
    bool over = false;
#pragma unroll
    for (int it = 0; it < kIpt; ++it) {
        const uint32_t win = (uint32_t)it * kSortWaves + (uint32_t)wv, j = win * 64 + (uint32_t)lane;
        run[it] = 0;
        if (j < nt) {
            const uint64_t hm = sh.masks[win];
            // last head at or before j: window 0 holds the head of key 0, so the walk back ends there at the latest
            uint64_t m = hm & le_mask;
            uint32_t w = win;
            while (m == 0 && win - w <= kMaxRun / 64) m = sh.masks[--w];
            const uint32_t rs = m ? w * 64 + 63u - (uint32_t)__clzll((long long)m) : j;
            over = over || m == 0;                                                 // no head within kMaxRun keys
            // first head behind j (masks beyond nt are 0: the last run ends with the tile)
            m = hm & ~le_mask;
            w = win;
            while (m == 0 && w + 1 < (uint32_t)kWindows && w - win <= kMaxRun / 64) m = sh.masks[++w];
            const uint32_t re = m ? w * 64 + (uint32_t)__ffsll((long long)m) - 1u : nt;
            over = over || re - rs > kMaxRun;
            run[it] = rs | ((re - rs) << 16);
        }
    }
    if (__ballot(over) && lane == 0) sh.flags[1] = 1;
    __syncthreads();
    if (wave_uniform(sh.flags[1])) return false;
#pragma unroll
    for (int it = 0; it < kIpt; ++it) {
        const uint32_t j = ((uint32_t)it * kSortWaves + (uint32_t)wv) * 64 + (uint32_t)lane;
        if (j < nt) {
            const Key<W> mine = sh.keys[j];
            const uint32_t rs = run[it] & 0xFFFFu, re = rs + (run[it] >> 16);
            uint32_t r = j - rs;
            if (!skip_compare) {
                r = 0;
#pragma unroll 2
                for (uint32_t t = rs; t < j; ++t) r += key_less<W, FIRST, MASKED>(mine, sh.keys[t], m2, m1) ? 0u : 1u;
#pragma unroll 2
                for (uint32_t t = j + 1; t < re; ++t) r += key_less<W, FIRST, MASKED>(sh.keys[t], mine, m2, m1) ? 1u : 0u;
            }
            keys[first + rs + r] = mine;
        }
    }
    return true;
}

template <int W, bool WANT_KEYS>   // !WANT_KEYS: only the tile bounds are wanted (every tile goes to local_lsd_kernel)
__global__ __launch_bounds__(kSortThreads, 8) void local_sort_kernel(Key<W> *keys, uint64_t n, LocalPlan lp, uint64_t *big, uint32_t *big_count,
                                                                     uint32_t big_cap) {
    constexpr int kTile = LocalCfg<W>::kTile, kIpt = LocalCfg<W>::kIpt, kChunk = LocalCfg<W>::kChunk, kWindows = kTile / 64;
    constexpr uint32_t NONE = ~0u;
    static_assert(kWindows <= 64, "tile bounds are read off one 64-bit head mask per lane");
    const uint32_t kLocalStride = lp.stride;
    __shared__ CompareShared<W> sh;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id(), T = lp.T;
    const uint64_t lo = (uint64_t)blockIdx.x * kLocalStride;
    const uint32_t loaded = (uint32_t)(n - lo < (uint64_t)kTile ? n - lo : (uint64_t)kTile);   // the window [lo, lo + loaded)
    constexpr bool want_keys = WANT_KEYS;
    if (tid == 0) { sh.flags[0] = 0; sh.flags[1] = 0; }
    // 1. one read of the window, in (wave chunk, round, lane) order; segment heads balloted per 64 keys
    Key<W> key[kIpt];
    uint32_t seg[kIpt];
    uint32_t carry = 0;
#pragma unroll
    for (int it = 0; it < kIpt; ++it) {
        const uint32_t j = (uint32_t)wv * kChunk + (uint32_t)it * 64 + (uint32_t)lane;
        const bool inw = j < loaded;
        uint32_t p = 0;
        if (inw) {
            if (want_keys) key[it] = keys[lo + j];
            else key[it].w[0] = keys[lo + j].w[0];
            p = key_prefix<W>(key[it], T);
        }
        uint32_t pp = __shfl_up(p, 1, 64);
        if (lane == 0) pp = it == 0 ? ((inw && lo + j > 0) ? key_prefix<W>(keys[lo + j - 1], T) : 0u) : carry;
        carry = __shfl(p, 63, 64);
        const uint64_t hm = __ballot(inw && (lo + j == 0 || p != pp));
        if (lane == 0) sh.masks[wv * kIpt + it] = hm;                 // window m = positions [64m, 64m + 64)
    }
    __syncthreads();
    // 2. the tile: from the first head in the stride to the first head behind it (every wave computes the same scalars)
    const uint64_t hmask = lane < kWindows ? sh.masks[lane] : 0ull;
    const bool in_stride = (uint32_t)lane < kLocalStride / 64;
    const uint32_t first_off = wave_uniform(wave_min(in_stride && hmask ? (uint32_t)lane * 64 + (uint32_t)__ffsll((long long)hmask) - 1u : NONE));
    if (first_off == NONE) return;                                    // the stride lies inside one long segment
    uint32_t end_off = wave_uniform(wave_min(!in_stride && hmask ? (uint32_t)lane * 64 + (uint32_t)__ffsll((long long)hmask) - 1u : NONE));
    if (end_off == NONE) {
        if (lo + loaded == n) end_off = loaded;
        else {                                                        // the last segment starting here does not fit: deferred
            end_off = wave_uniform(wave_max(in_stride && hmask ? (uint32_t)lane * 64 + 63u - (uint32_t)__clzll((long long)hmask) : 0u));
            if (tid == 0) {
                uint32_t q = atomicAdd(big_count, 1u);
                if (q < big_cap) big[q] = lo + end_off;
            }
        }
    }
    const uint32_t nt = end_off - first_off;
    if (nt == 0) return;
    const uint64_t first = lo + first_off;
    auto leave_to_lsd = [&]() {
        if (tid == 0) {
            uint32_t q = atomicAdd(lp.lsd_count, 1u);
            if (q < lp.lsd_cap) { lp.lsd_list[2 * q] = first; lp.lsd_list[2 * q + 1] = lo + end_off; }
        }
    };
    if (!want_keys) { leave_to_lsd(); return; }
    // rank of every key's segment inside the tile: heads in [first_off, its position]
    uint32_t nseg;
    {
        uint64_t in_tile = hmask;                                     // heads of window `lane` inside [first_off, end_off)
        const uint32_t w0 = (uint32_t)lane * 64;
        if (w0 + 64 <= first_off || w0 >= end_off) in_tile = 0;
        else {
            if (first_off > w0) in_tile &= ~0ull << (first_off - w0);
            if (end_off < w0 + 64) in_tile &= ~0ull >> (w0 + 64 - end_off);
        }
        const uint32_t c = (uint32_t)__popcll(in_tile), inc = wave_incl_scan(c);
        nseg = wave_uniform((uint32_t)__shfl(inc, 63, 64));
        const uint64_t le_mask = lanemask_lt() | (1ull << lane);
#pragma unroll
        for (int it = 0; it < kIpt; ++it) {
            const int m = wv * kIpt + it;
            const uint32_t before = (uint32_t)__shfl(inc - c, m, 64);
            const uint64_t mm = (uint64_t)__shfl((unsigned long long)in_tile, m, 64);
            seg[it] = before + (uint32_t)__popcll(mm & le_mask) - 1u;       // only read for positions inside the tile
        }
    }
    if (lp.debug & 2) return;
    // 3. one counting pass on (segment rank, upper digit): the order inside a bin is left to step 4, so plain LDS atomics rank the keys.
    //    The digit shrinks when the tile holds many (then short) segments: nseg << ub bins fit the counter array.
    constexpr int kBins = CompareShared<W>::kBins, kWordsPerThread = kBins / 2 / kSortThreads;
    int ub = 31 - __clz((int)((uint32_t)kBins / nseg));
    if (ub > lp.upper.bits) ub = lp.upper.bits;
    const uint32_t n_words = ((nseg << ub) + 1u) >> 1;
    for (uint32_t i = tid; i < n_words; i += kSortThreads) sh.hist[i] = 0;
    __syncthreads();
    const int run_bits = T + ub;                                       // <= 40: the digit may reach into the second key word
    uint32_t br[kIpt];                                                // bin | rank inside the bin << 16
#pragma unroll
    for (int it = 0; it < kIpt; ++it) {
        const uint32_t j = (uint32_t)wv * kChunk + (uint32_t)it * 64 + (uint32_t)lane;
        br[it] = ~0u;
        if (j - first_off < nt) {
            const uint32_t dg = ub ? (uint32_t)((key_top<W, 2>(key[it]) << T) >> (64 - ub)) : 0u;
            const uint32_t bin = (seg[it] << ub) | dg, sh16 = (bin & 1u) * 16u;
            const uint32_t old = atomicAdd(&sh.hist[bin >> 1], 1u << sh16);
            br[it] = bin | (((old >> sh16) & 0xFFFFu) << 16);
        }
    }
    __syncthreads();
    {   // exclusive scan of the counters, in place (a tile holds <= 4096 keys: 16 bits never overflow)
        uint32_t wds[kWordsPerThread], run = 0;
#pragma unroll
        for (int i = 0; i < kWordsPerThread; ++i) {
            const uint32_t wi = (uint32_t)tid * kWordsPerThread + i;
            const uint32_t x = wi < n_words ? sh.hist[wi] : 0u;
            wds[i] = run | ((run + (x & 0xFFFFu)) << 16);
            run += (x & 0xFFFFu) + (x >> 16);
        }
        const uint32_t base = block_excl_scan<kSortThreads>(run, sh.scratch, nullptr);
#pragma unroll
        for (int i = 0; i < kWordsPerThread; ++i) {
            const uint32_t wi = (uint32_t)tid * kWordsPerThread + i;
            if (wi < n_words) sh.hist[wi] = wds[i] + base * 0x00010001u;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kIpt; ++it) {
        if (br[it] != ~0u) {
            const uint32_t bin = br[it] & 0xFFFFu;
            const uint32_t pos = ((sh.hist[bin >> 1] >> ((bin & 1u) * 16u)) & 0xFFFFu) + (br[it] >> 16);
#pragma unroll
            for (int w = 0; w < W; ++w) sh.keys[pos].w[w] = key[it].w[w];
        }
    }
    __syncthreads();
    if (lp.debug & 4) return;
    // 4. runs of equal (prefix, upper digit): every key ranks itself inside its run
    const bool skip = (lp.debug & 1) != 0;
    const bool masked = (lp.mask_last2 & lp.mask_last) != ~0u;
    bool done;
    if (run_bits > 32 && W > 1)
        done = masked ? finish_by_comparison<W, (W > 1 ? 1 : 0), true, 2>(sh, keys, first, nt, run_bits, lp.mask_last2, lp.mask_last, skip)
                      : finish_by_comparison<W, (W > 1 ? 1 : 0), false, 2>(sh, keys, first, nt, run_bits, lp.mask_last2, lp.mask_last, skip);
    else if (run_bits == 32 && W > 1)
        done = masked ? finish_by_comparison<W, (W > 1 ? 1 : 0), true, 1>(sh, keys, first, nt, run_bits, lp.mask_last2, lp.mask_last, skip)
                      : finish_by_comparison<W, (W > 1 ? 1 : 0), false, 1>(sh, keys, first, nt, run_bits, lp.mask_last2, lp.mask_last, skip);
    else
        done = masked ? finish_by_comparison<W, 0, true, 1>(sh, keys, first, nt, run_bits, lp.mask_last2, lp.mask_last, skip)
                      : finish_by_comparison<W, 0, false, 1>(sh, keys, first, nt, run_bits, lp.mask_last2, lp.mask_last, skip);
    if (!done) leave_to_lsd();                                        // global memory still holds the tile as it was
}

// one workgroup per listed range [start[b], end[b]) of whole segments (<= kTile keys, longer ones are skipped: segment_sort_kernel):
// stable LSD passes in LDS over every significant digit below the prefix, then the segment rank
template <int W>
__global__ __launch_bounds__(kSortThreads, 8) void local_lsd_kernel(Key<W> *keys, const uint64_t *start, const uint64_t *end, int stride,
                                                                    LocalPlan lp) {
    constexpr int kIpt = LocalCfg<W>::kIpt;
    __shared__ LocalShared<W> sh;
    const int tid = threadIdx.x;
    const uint64_t first = wave_uniform(start[(uint64_t)blockIdx.x * stride]), cnt = wave_uniform(end[(uint64_t)blockIdx.x * stride]) - first;
    if (cnt > (uint64_t)LocalCfg<W>::kTile || cnt == 0) return;
    const uint32_t nt = (uint32_t)cnt;
    Key<W> key[kIpt];
    uint32_t seg[kIpt];
    const int n_seg_pass = tile_load<W>(sh, keys, first, nt, lp.T, key, seg);
    const int n_pass = lp.n_low + n_seg_pass;
    tile_zero_counts<W>(sh);
    for (int pass = 0; pass < n_pass; ++pass) {
        const bool by_seg = pass >= lp.n_low;
        Digit d;
        d.pos = 0; d.bits = 8; d.bias = 0;
        if (!by_seg) d = lp.low_plan[pass];
        lds_pass<W>(sh, key, seg, nt, d, by_seg, by_seg ? 8 * (pass - lp.n_low) : 0, pass + 1 < n_pass);
    }
    // write back (coalesced); n_low >= 1, so LDS holds the result
#pragma unroll
    for (int it = 0; it < kIpt; ++it) {
        uint32_t j = (uint32_t)it * kSortThreads + (uint32_t)tid;
        if (j < nt) keys[first + j] = sh.keys[j];
    }
}

// one workgroup per listed segment of LocalCfg::kTile < keys <= BigCfg::kTile: the same LSD passes with the LDS of a whole CU (a
// hot 16-mer prefix of 5-8 thousand keys took eight global census + scatter round trips of ~100 us each in segment_sort_kernel)
template <int W>
__global__ __launch_bounds__(kSortThreads) void segment_lds_kernel(Key<W> *keys, const uint64_t *start, const uint64_t *end, LocalPlan lp) {
    constexpr int TILE = BigCfg<W>::kTile > 0 ? BigCfg<W>::kTile : LocalCfg<W>::kTile, kIpt = TILE / kSortThreads, kChunk = TILE / kSortWaves;
    __shared__ LocalShared<W, TILE> sh;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const uint64_t first = wave_uniform(start[blockIdx.x]), cnt = wave_uniform(end[blockIdx.x]) - first;
    if (cnt <= (uint64_t)LocalCfg<W>::kTile || cnt > (uint64_t)BigCfg<W>::kTile) return;
    const uint32_t nt = (uint32_t)cnt;
    Key<W> key[kIpt];
    uint32_t seg[kIpt];
#pragma unroll
    for (int it = 0; it < kIpt; ++it) {
        const uint32_t j = (uint32_t)wv * kChunk + (uint32_t)it * 64 + (uint32_t)lane;
        seg[it] = 0;
        if (j < nt) key[it] = keys[first + j];
    }
    tile_zero_counts<W, TILE>(sh);
    for (int pass = 0; pass < lp.n_low; ++pass) lds_pass<W, TILE>(sh, key, seg, nt, lp.low_plan[pass], false, 0, pass + 1 < lp.n_low);
#pragma unroll
    for (int it = 0; it < kIpt; ++it) {                                  // n_low >= 1: LDS holds the result
        const uint32_t j = (uint32_t)it * kSortThreads + (uint32_t)tid;
        if (j < nt) keys[first + j] = sh.keys[j];
    }
}

// first segment head after `start` (end of a big segment): one workgroup per big segment
template <int W>
__global__ __launch_bounds__(256) void segment_end_kernel(const Key<W> *keys, uint64_t n, int T, const uint64_t *big, uint64_t *big_end) {
    __shared__ unsigned long long s_end;
    const uint64_t start = big[blockIdx.x];
    const uint32_t pre = key_prefix<W>(keys[start], T);
    if (threadIdx.x == 0) s_end = ~0ull;
    __syncthreads();
    for (uint64_t base = start + 1; base < n; base += 256 * 16) {
        unsigned long long found = ~0ull;
        for (int i = 0; i < 16; ++i) {
            uint64_t idx = base + (uint64_t)i * 256 + threadIdx.x;
            if (idx < n && key_prefix<W>(keys[idx], T) != pre && (unsigned long long)idx < found) found = idx;
        }
        if (found != ~0ull) atomicMin(&s_end, found);
        __syncthreads();
        if (s_end != ~0ull) break;
    }
    __syncthreads();
    if (threadIdx.x == 0) big_end[blockIdx.x] = s_end == ~0ull ? n : s_end;
}

// ---------------------------------------------------------------------------------------------
// 5. edge emission
// ---------------------------------------------------------------------------------------------
constexpr int kEmitThreads = 256;
constexpr int kEmitPerThread = 16;
constexpr int kEmitTile = kEmitThreads * kEmitPerThread;   // 4096 sorted keys per workgroup

template <int W>
__device__ __forceinline__ bool keys_equal(const Key<W> &x, const Key<W> &y) {
    bool eq = true;
#pragma unroll
    for (int j = 0; j < W; ++j) eq = eq && (x.w[j] == y.w[j]);
    return eq;
}
template <int W>
__device__ __forceinline__ bool same_km1(const Key<W> &x, const Key<W> &y, int k) {   // IsDiffKMinusOneMer, s2.cpp:54-75
    int full = (k - 1) >> 4, rem = (k - 1) & 15;
    bool eq = true;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        if (j < full) eq = eq && (x.w[j] == y.w[j]);
        else if (j == full && rem > 0) eq = eq && ((x.w[j] >> (16 - rem) * 2) == (y.w[j] >> (16 - rem) * 2));
    }
    return eq;
}
template <int W>
__device__ __forceinline__ int key_a(const Key<W> &x, int k) {   // Extract_a, s2.cpp:83-94
    if ((x.w[W - 1] >> 3) & 1u) {
        int wi = (k - 1) >> 4, ci = (k - 1) & 15;
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < W; ++j) if (j == wi) word = x.w[j];
        return (word >> (15 - ci) * 2) & 3;
    }
    return kDollar;
}
template <int W>
__device__ __forceinline__ int key_b(const Key<W> &x) { return x.w[W - 1] & 7u; }   // Extract_b, s2.cpp:96-98

// Where the runs start: the low 32 bits per run, the full 64 bits for every kRunBaseStep-th run.  Runs are consecutive, so a
// length is a 32-bit difference (a run never holds 2^32 keys) and a full start is rebuilt from the nearest base.
constexpr int kRunBaseStepLog = 10;
struct RunStarts {
    uint32_t *lo;                // [m]
    uint64_t *base;              // [m >> kRunBaseStepLog + 1]
};
__device__ __forceinline__ uint64_t run_full_start(const RunStarts &r, uint64_t s) {
    const uint64_t b = r.base[s >> kRunBaseStepLog];
    return b + (uint32_t)(r.lo[s] - (uint32_t)b);
}
__device__ __forceinline__ uint64_t run_length(const RunStarts &r, uint64_t s, uint64_t m, uint64_t n_items) {
    return s + 1 < m ? (uint64_t)(uint32_t)(r.lo[s + 1] - r.lo[s]) : n_items - run_full_start(r, s);
}

// E1+E2: one descriptor per run (distinct key): start index + (a | b<<3 | group_head<<6 | bucket_head<<7), compacted in key order.
// The keys are read once: the number of runs before a tile comes from a chained scan across the workgroups (device_utils.hpp).
struct EmitChain {
    unsigned long long *state;   // [n_tiles], zeroed
    uint32_t *ticket;            // zeroed
    uint32_t *error;
    unsigned long long *total;   // number of runs, written by the last tile
};

// TICKET = false numbers the tiles by blockIdx (workgroups are dispatched in that order on this hardware, which nothing guarantees):
// if a tile ever waits in vain the bounded walk raises chain.error and the host repeats the launch with TICKET = true.
template <int W, bool TICKET>
__global__ __launch_bounds__(kEmitThreads) void emit_compact_kernel(const Key<W> *keys, uint64_t n, int k, EmitChain chain, uint32_t n_tiles,
                                                                     RunStarts sub_start, uint8_t *sub_info) {
    __shared__ uint32_t s_cnt[kEmitPerThread * (kEmitThreads / 64)];
    __shared__ uint32_t s_scr[kEmitThreads / 64 + 1];
    __shared__ uint32_t s_tile;
    __shared__ unsigned long long s_base;
    const uint32_t tile = TICKET ? chain_ticket(chain.ticket, &s_tile) : blockIdx.x;
    uint64_t base = (uint64_t)tile * kEmitTile;
    const int lane = lane_id(), wv = wave_id();
    uint32_t headbits = 0;
    uint32_t rank_in_wave[kEmitPerThread];
    uint8_t info[kEmitPerThread];
#pragma unroll
    for (int it = 0; it < kEmitPerThread; ++it) {
        uint64_t idx = base + (uint64_t)it * kEmitThreads + threadIdx.x;
        bool head = false;
        info[it] = 0;
        if (idx < n) {
            Key<W> cur = keys[idx];
            bool ghead = true, bhead = true;
            if (idx == 0) head = true;
            else {
                Key<W> prv = keys[idx - 1];                              // (taking it from the neighbouring lane by a shuffle was slower: 9.3 -> 11.2 ms)
                head = !keys_equal<W>(cur, prv);
                ghead = !same_km1<W>(cur, prv, k);
                bhead = (cur.w[0] >> 16) != (prv.w[0] >> 16);             // first key of its bucket
            }
            info[it] = (uint8_t)(key_a<W>(cur, k) | (key_b<W>(cur) << 3) | ((int)ghead << 6) | ((int)bhead << 7));
        }
        uint64_t bal = __ballot(head);
        rank_in_wave[it] = (uint32_t)__popcll(bal & lanemask_lt());
        headbits |= (uint32_t)head << it;
        if (lane == 0) s_cnt[it * (kEmitThreads / 64) + wv] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    // exclusive scan of the 64 (iteration, wave) counts: order of keys = iteration-major, wave, lane
    uint32_t v = threadIdx.x < kEmitPerThread * (kEmitThreads / 64) ? s_cnt[threadIdx.x] : 0, tile_total = 0;
    uint32_t ex = block_excl_scan<kEmitThreads>(v, s_scr, &tile_total);
    __syncthreads();
    if (threadIdx.x < kEmitPerThread * (kEmitThreads / 64)) s_cnt[threadIdx.x] = ex;
    if (wv == 0) {
        const uint64_t before = chain_exclusive(chain.state, tile, tile_total, chain.error);
        if (lane == 0) {
            s_base = before;
            if (tile + 1 == n_tiles) *chain.total = before + tile_total;
        }
    }
    __syncthreads();
    uint64_t tb = s_base;
#pragma unroll
    for (int it = 0; it < kEmitPerThread; ++it) {
        if ((headbits >> it) & 1u) {
            uint64_t idx = base + (uint64_t)it * kEmitThreads + threadIdx.x;
            uint64_t s = tb + s_cnt[it * (kEmitThreads / 64) + wv] + rank_in_wave[it];
            sub_start.lo[s] = (uint32_t)idx;
            if ((s & ((1u << kRunBaseStepLog) - 1)) == 0) sub_start.base[s >> kRunBaseStepLog] = idx;
            sub_info[s] = info[it];
        }
    }
}

// E3: decide every run (sub-group): the rules of output_(), cx1_read2sdbg_s2.cpp:763-834.
// rec = w | last<<4 | tip<<5 | min(mult,255)<<8, or 0xFFFF when the run is suppressed.
constexpr int kDecideThreads = 256;
constexpr int kDecidePerThread = 4;
constexpr int kDecideTile = kDecideThreads * kDecidePerThread;   // 1024 runs per workgroup
constexpr int kDecideHalo = 32;

__device__ __forceinline__ bool run_suppressed(int a, int b, int has_a, int has_b) {
    return (a == kDollar && ((has_b >> b) & 1)) || (b == kDollar && ((has_a >> a) & 1));
}

__global__ __launch_bounds__(kDecideThreads) void emit_decide_kernel(RunStarts sub_start, const uint8_t *sub_info, uint64_t m,
                                                                      uint64_t n_items, uint16_t *rec, uint32_t *cnt_e,
                                                                      uint32_t *cnt_l, uint32_t *cnt_t) {
    __shared__ uint32_t s_e[kDecideThreads / 64], s_l[kDecideThreads / 64], s_t[kDecideThreads / 64];
    // the tile's descriptors and a halo of one group on either side, staged once: the group walks below are chains of dependent
    // byte loads (a walk that leaves the window, which no real group does, falls back to global memory)
    __shared__ uint8_t s_info[kDecideTile + 2 * kDecideHalo];
    const uint64_t t0 = (uint64_t)blockIdx.x * kDecideTile;
    const uint64_t w_lo = t0 >= (uint64_t)kDecideHalo ? t0 - kDecideHalo : 0;
    const uint64_t w_hi = t0 + kDecideTile + kDecideHalo < m ? t0 + kDecideTile + kDecideHalo : m;
    for (uint64_t x = w_lo + threadIdx.x; x < w_hi; x += kDecideThreads) s_info[x - w_lo] = sub_info[x];
    __syncthreads();
    auto info_at = [&](uint64_t x) -> int { return (x >= w_lo && x < w_hi) ? (int)s_info[x - w_lo] : (int)sub_info[x]; };
    uint32_t ne = 0, nl = 0, nt = 0;
    for (int q = 0; q < kDecidePerThread; ++q) {
        uint64_t s = t0 + (uint64_t)q * kDecideThreads + threadIdx.x;      // consecutive lanes, consecutive runs
        if (s >= m) break;
        int inf = info_at(s);
        int a = inf & 7, b = (inf >> 3) & 7;
        // group extent [gs, ge): at most 24 runs share a (k-1)-mer; the characters its solid runs carry are collected on the way (the
        // kernel is bound by vector issue: one walk over the group here instead of two, and the walk back also answers outputed_b: 16.3 ->
        // 11.35 ms at 100 M reads; folding the walk over the runs behind in as well costs more than it saves: 15.1 ms)
        int has_a = 0, has_b = 0;
        auto collect = [&](int xi) {
            const int xa = xi & 7, xb = (xi >> 3) & 7;
            if (xa != kDollar && xb != kDollar) { has_a |= 1 << xa; has_b |= 1 << xb; }
        };
        collect(inf);
        // (outputed_b, s2.cpp:822-824, from the same walk back: a run in front with the same b is not suppressed if its a != $, or if its
        // a == $ and no solid run of the group carries b)
        bool front_b_solid = false, front_b_dollar = false;
        uint64_t gs = s;
        for (int gi = inf; !(gi & 64);) {
            --gs; gi = info_at(gs); collect(gi);
            if (((gi >> 3) & 7) == b) { if ((gi & 7) != kDollar) front_b_solid = true; else front_b_dollar = true; }
        }
        uint64_t ge = s + 1;
        while (ge < m) {
            const int xi = info_at(ge);
            if (xi & 64) break;
            collect(xi);
            ++ge;
        }
        uint16_t r = 0xFFFF;
        if (!run_suppressed(a, b, has_a, has_b)) {
            const bool seen_b = front_b_solid || (front_b_dollar && !((has_b >> b) & 1));
            int w = (b == kDollar) ? 0 : (seen_b ? b + 5 : b + 1);
            int last = 0;
            if (a != kDollar) {                                    // last_a[], s2.cpp:776-779,823
                bool later = false;
                for (uint64_t x = s + 1; x < ge; ++x) {
                    int xi = info_at(x), xa = xi & 7, xb = (xi >> 3) & 7;
                    if (xa == a && (xb != kDollar || !((has_a >> a) & 1))) later = true;
                }
                last = later ? 0 : 1;
            }
            uint64_t run = run_length(sub_start, s, m, n_items);
            uint32_t count = run > 65535 ? 65535u : (uint32_t)run;  // kMaxMulti_t
            int tip = a == kDollar;
            r = (uint16_t)(w | (last << 4) | (tip << 5) | ((count > 255 ? 255u : count) << 8));
            ne++;
            nl += count > 254;                                      // kMaxMulti2_t, sdbg_multi_io.h:99
            nt += tip;
        }
        rec[s] = r;
    }
    ne = wave_sum(ne); nl = wave_sum(nl); nt = wave_sum(nt);
    if (lane_id() == 0) { s_e[wave_id()] = ne; s_l[wave_id()] = nl; s_t[wave_id()] = nt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t e = 0, l = 0, t = 0;
        for (int w = 0; w < kDecideThreads / 64; ++w) { e += s_e[w]; l += s_l[w]; t += s_t[w]; }
        cnt_e[blockIdx.x] = e; cnt_l[blockIdx.x] = l; cnt_t[blockIdx.x] = t;
    }
}

// E5: order-preserving compaction into the output stream + per-bucket boundaries
template <int W>
__global__ __launch_bounds__(kDecideThreads) void emit_write_kernel(const Key<W> *keys, RunStarts sub_start, const uint8_t *sub_info, const uint16_t *rec,
                                                                     uint64_t m, uint64_t n_items, const uint64_t *base_e,
                                                                     const uint64_t *base_l, const uint64_t *base_t, int words_per_tip,
                                                                     uint32_t b_lo, uint16_t *out_rec, uint16_t *out_large,
                                                                     uint32_t *out_tips, int64_t *bucket_first /* [nb][3] */) {
    __shared__ uint64_t s_scr[kDecideThreads / 64 + 1];
    uint64_t s0 = (uint64_t)blockIdx.x * kDecideTile + (uint64_t)threadIdx.x * kDecidePerThread;
    uint16_t r[kDecidePerThread];
    uint64_t packed = 0;   // emitted | large << 20 | tips << 40
    uint32_t cnts[kDecidePerThread];
    for (int q = 0; q < kDecidePerThread; ++q) {
        uint64_t s = s0 + q;
        r[q] = 0xFFFF; cnts[q] = 0;
        if (s < m) {
            r[q] = rec[s];
            if (r[q] != 0xFFFF) {
                uint64_t run = run_length(sub_start, s, m, n_items);
                cnts[q] = run > 65535 ? 65535u : (uint32_t)run;
                packed += 1ull + ((uint64_t)(cnts[q] > 254) << 20) + ((uint64_t)((r[q] >> 5) & 1) << 40);
            }
        }
    }
    uint64_t ex = block_excl_scan64<kDecideThreads>(packed, s_scr, nullptr);
    uint64_t ie = base_e[blockIdx.x] + (ex & 0xFFFFF), il = base_l[blockIdx.x] + ((ex >> 20) & 0xFFFFF),
             it = base_t[blockIdx.x] + (ex >> 40);
    for (int q = 0; q < kDecidePerThread; ++q) {
        uint64_t s = s0 + q;
        if (s >= m) break;
        const bool is_tip = r[q] != 0xFFFF && ((r[q] >> 5) & 1);
        const bool bucket_head = (sub_info[s] & 128) != 0;
        const uint64_t first_item = (is_tip || bucket_head) ? run_full_start(sub_start, s) : 0;   // only these two look at the key
        if (bucket_head) {                                         // number of records/large/tips before this bucket
            uint32_t bucket = keys[first_item].w[0] >> 16;
            int64_t *bf = bucket_first + (uint64_t)(bucket - b_lo) * 3;
            bf[0] = (int64_t)ie; bf[1] = (int64_t)il; bf[2] = (int64_t)it;
        }
        if (r[q] != 0xFFFF) {
            out_rec[ie++] = r[q];
            if (cnts[q] > 254) out_large[il++] = (uint16_t)cnts[q];
            if ((r[q] >> 5) & 1) {
                for (int j = 0; j < words_per_tip; ++j) out_tips[it * words_per_tip + j] = keys[first_item].w[j];   // s2.cpp:826-830
                it++;
            }
        }
    }
}

}  // namespace mgta
#include "sdbg_solid.hpp"
namespace mgta {

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------
struct Timer {
    hipEvent_t a, b;
    hipStream_t st;
    explicit Timer(hipStream_t s) : st(s) { MGTA_HIP_CHECK(hipEventCreate(&a)); MGTA_HIP_CHECK(hipEventCreate(&b)); }
    ~Timer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
    void start() { MGTA_HIP_CHECK(hipEventRecord(a, st)); }
    double stop() {   // milliseconds, synchronises; a launch the runtime rejected inside the phase (grid or LDS limits) surfaces here
        MGTA_HIP_CHECK(hipGetLastError());
        MGTA_HIP_CHECK(hipEventRecord(b, st));
        MGTA_HIP_CHECK(hipEventSynchronize(b));
        MGTA_HIP_CHECK(hipGetLastError());
        float ms = 0;
        MGTA_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        return ms;
    }
};

// most significant digits first: digit i = key bits [32W - 8(i+1), 32W - 8i)
static Digit top_digit(int W, int i) { return Digit{32 * W - 8 * (i + 1), 8}; }
// digits of the bits below 32W - T (T = prefix bits the global passes sorted on), least significant first (flags, then characters; the zero pad is skipped)
static std::vector<Digit> low_digit_plan(int k, int W, int T) {
    std::vector<Digit> plan;
    int pad = 32 * W - 2 * k - 4, top = 32 * W - T;
    if (top > 0) plan.push_back(Digit{0, std::min(4, top)});
    for (int pos = 4 + pad; pos < top; pos += 8) plan.push_back(Digit{pos, std::min(8, top - pos)});
    return plan;
}

// grow-only device buffers kept in the context between calls (multi-k builds, repeated steps):
// hipMalloc/hipFree of multi-GB buffers costs far more than the kernels that use them.
enum Slot { S_BLOCK_COUNT, S_BLOCK_BASE, S_SCAN_TMP, S_SMALL, S_KEYS_A, S_KEYS_B, S_HIST, S_TILE_HEADS, S_TILE_BASE, S_CNT, S_BASE,
            S_FIRST, S_OUT_REC, S_OUT_LARGE, S_OUT_TIPS, S_PLAN, S_BIG, S_LSD, S_MULTI_COUNT, S_POS2ID, S_SOLID, S_MERCY, S_EDGE_COUNT, S_SIDE, S_NUM };

template <class T>
static T *pool_get(mgta_ctx *ctx, int slot, uint64_t bytes) {
    if ((int)ctx->pool.size() < S_NUM) ctx->pool.resize(S_NUM);
    DevBuf &b = ctx->pool[slot];
    if (b.bytes < bytes || !b.p) {
        b.release();
        // an eighth of headroom (the pass planner leaves that much): the builds of a multi-k run need a few per cent more from one k to
        // the next (contigs join the reads), and obtaining the buffers again costs more than a build (20 M reads: 1.7 s for 9 % more)
        b.alloc(bytes + bytes / 8, &ctx->live_bytes, &ctx->peak_bytes);
    }
    return b.as<T>();
}
static uint64_t pool_bytes(const mgta_ctx *ctx) {
    uint64_t t = 0;
    for (const DevBuf &b : ctx->pool) t += b.bytes;
    return t;
}

// Sort n keys of WT words ascending on the digits that matter: `max_top` leading bytes may be used for global passes;
// low_plan_for(P) lists the remaining significant digits (least significant first).  Returns the buffer (a or b) that
// holds the result, nullptr on an unsupported input (error set).
// P: smallest number of leading bytes that leaves segments of <= 256 keys on average — up to 1024 rather than a fourth
// global pass: a 4096-key tile still holds several such segments, and the comparison route needs key bits left in word 0.
// prefix_frac: share of the leading-byte values the keys can take (a pass over a bucket sub-range only holds that share of
// the prefixes, so its segments are as long as those of the whole key set)
static double avg_segment_len(uint64_t n_items, int p, double prefix_frac) { return (double)n_items / std::max(1.0, std::pow(256.0, p) * prefix_frac); }
// The global passes of a build over the buckets [b_lo, b_hi) need not spend digit values on prefixes no key has: word 0 minus
// (b_lo << 16) has `skip` leading zero bits, and P digits of 8 bits taken right below them order the keys by their leading
// T = 8P + skip bits (a third of the buckets: skip = 1, so three passes leave the 645-key segments that otherwise take four).
// T >= 16 is required: then the bias is a multiple of 2^(32 - T) and equal biased prefixes are equal key prefixes.
struct TopPlan {
    int P = 0, skip = 0;
    uint32_t bias = 0;
    double frac = 1.0;           // share of the 2^T prefix values the keys can take
    int T() const { return 8 * P + skip; }
};
static TopPlan choose_top_plan(const mgta_ctx *ctx, uint64_t n_items, int max_top, double prefix_frac, uint32_t b_lo = 0, uint32_t b_hi = 0) {
    TopPlan plain;
    plain.frac = prefix_frac;
    if (ctx && ctx->force_full_lsd) return plain;
    auto passes = [&](double frac_of, int p_min) {
        int P = p_min;
        while (P < max_top && avg_segment_len(n_items, P, frac_of) > (P >= 3 ? 700.0 : 256.0)) ++P;
        return P;
    };
    plain.P = passes(prefix_frac, 0);
    // 0 never, 1 (default) when it saves a pass and leaves short segments, 2 whenever valid (tests); read per call: scripts flip it between builds
    const char *mode_env = getenv("MGTA_SORT_BIAS");
    const int mode = mode_env ? atoi(mode_env) : 1;
    if (mode == 0 || b_hi <= b_lo || (b_lo == 0 && b_hi >= (uint32_t)MGTA_NUM_BUCKETS)) return plain;
    const uint32_t span = ((b_hi - b_lo) << 16) - 1u;                  // largest biased word 0
    const int skip_full = __builtin_clz(span | 1u);
    for (int P = 1; P <= max_top; ++P) {
        const int skip = std::min(skip_full, 32 - 8 * P);
        if (skip <= 0 || 8 * P + skip < 16) continue;
        const double frac = std::min(1.0, prefix_frac * std::pow(2.0, skip));
        // measured at 100 M reads (7.2 G keys per pass, a third of the buckets): three biased passes leave 645-key segments and the LDS
        // finish then costs 170 ms instead of 100 ms — exactly what the fourth scatter + census (45 + 17 ms) would have cost.  The pass is
        // only worth skipping when the segments stay short (<= 400 keys).
        const double limit = mode >= 2 ? (P >= 3 ? 700.0 : 256.0) : (P >= 3 ? 400.0 : 256.0);
        if (P < max_top && avg_segment_len(n_items, P, frac) > limit) continue;
        if (P < plain.P || (mode >= 2 && P <= plain.P)) {
            TopPlan b;
            b.P = P; b.skip = skip; b.bias = b_lo << 16; b.frac = frac;
            return b;
        }
        break;
    }
    return plain;
}

// first_census_done: the census of the first global pass (digit top_digit(WT, P-1), tiles of kBlockTile keys of `a`) is already in
// the S_HIST buffer (item_write_tiled_kernel counted while it wrote the keys)
template <int WT, class LowPlanFn>
static Key<WT> *device_sort(mgta_ctx *ctx, hipStream_t stream, Key<WT> *a, Key<WT> *b, uint64_t n_items, int max_top, LowPlanFn low_plan_for,
                            std::vector<std::pair<hipEvent_t, hipEvent_t>> *scatter_ev, mgta_build_stats *S, double prefix_frac = 1.0,
                            uint32_t mask_last2 = ~0u, uint32_t mask_last = ~0u, bool first_census_done = false, uint32_t b_lo = 0, uint32_t b_hi = 0,
                            bool first_side_done = false) {
    const uint64_t n_tiles = (n_items + kBlockTile - 1) / kBlockTile;
    uint64_t *d_hist = pool_get<uint64_t>(ctx, S_HIST, std::max<uint64_t>(1, n_tiles) * 256 * 8);
    uint64_t *d_totals = pool_get<uint64_t>(ctx, S_SMALL, 4096) + 8;
    Key<WT> *src = a, *dst = b;
    bool side_failed = false;
    // side digits (MGTA_SORT_SIDE=0 switches them off; 2 checks every side census against the census of the keys): the scatter of
    // a pass leaves the NEXT pass's digit of every key in a byte array, and that pass's census reads the bytes instead of the keys
    const char *side_env = getenv("MGTA_SORT_SIDE");
    const int side_mode = side_env ? atoi(side_env) : 1;
    const TopPlan tp = choose_top_plan(ctx, n_items, max_top, prefix_frac, b_lo, b_hi);
    const int P = tp.P, T = tp.T();
    uint8_t *d_side = side_mode > 0 && P >= 2 && kSideFits<WT> ? pool_get<uint8_t>(ctx, S_SIDE, n_items + 64) : nullptr;
    bool side_valid = first_side_done && d_side;                        // d_side holds the digits of the pass about to run
    static const bool unstable_first = !(getenv("MGTA_SORT_UNSTABLE_FIRST") && atoi(getenv("MGTA_SORT_UNSTABLE_FIRST")) == 0);   // (measurements)
    auto global_pass = [&](Key<WT> *from, Key<WT> *to, uint64_t cnt, const Digit &dg, bool have_census, const Digit *dg_next, bool unstable_ok) {
        uint64_t tiles = (cnt + kBlockTile - 1) / kBlockTile;
        if (!have_census) {
            if (side_valid) {
                hipLaunchKernelGGL(radix_census_side_kernel, dim3((unsigned)tiles), dim3(kSortThreads), 0, stream, d_side, cnt, tiles, d_hist);
                if (side_mode >= 2) {                                   // (test aid) the same census from the keys must agree
                    std::vector<uint64_t> h_side(tiles * 256), h_keys(tiles * 256);
                    MGTA_HIP_CHECK(hipStreamSynchronize(stream));
                    MGTA_HIP_CHECK(hipMemcpy(h_side.data(), d_hist, tiles * 256 * 8, hipMemcpyDeviceToHost));
                    hipLaunchKernelGGL((radix_census_kernel<WT>), dim3((unsigned)tiles), dim3(kSortThreads), 0, stream, from, cnt, dg, tiles, d_hist);
                    MGTA_HIP_CHECK(hipStreamSynchronize(stream));
                    MGTA_HIP_CHECK(hipMemcpy(h_keys.data(), d_hist, tiles * 256 * 8, hipMemcpyDeviceToHost));
                    if (h_side != h_keys) { set_error("internal: side-digit census differs from the census of the keys"); side_failed = true; }
                }
            } else
                hipLaunchKernelGGL((radix_census_kernel<WT>), dim3((unsigned)tiles), dim3(kSortThreads), 0, stream, from, cnt, dg, tiles, d_hist);
        }
        hipLaunchKernelGGL(radix_rowscan_kernel, dim3(256), dim3(1024), 0, stream, d_hist, tiles, d_totals);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (scatter_ev) {
            MGTA_HIP_CHECK(hipEventCreate(&e0));
            MGTA_HIP_CHECK(hipEventCreate(&e1));
            MGTA_HIP_CHECK(hipEventRecord(e0, stream));
        }
        const bool write_side = d_side && dg_next;
        const Digit dn = dg_next ? *dg_next : Digit{0, 0, 0};
        const dim3 grid((unsigned)tiles), block(kSortThreads);
        // (the first pass of the sort orders nothing that was ordered before: its ranks need not be stable)
        auto launch = [&](auto biased, auto side, auto stable) {
            hipLaunchKernelGGL((radix_scatter_kernel<WT, decltype(biased)::value, decltype(side)::value, decltype(stable)::value>), grid, block, 0, stream, from, to, cnt,
                               dg, tiles, d_hist, d_totals, decltype(side)::value ? d_side : nullptr, dn);
        };
        auto pick_stable = [&](auto biased, auto side) {
            if (unstable_ok) launch(biased, side, std::false_type{});
            else launch(biased, side, std::true_type{});
        };
        auto pick_side = [&](auto biased) {
            if constexpr (kSideFits<WT>) {
                if (write_side) { pick_stable(biased, std::true_type{}); return; }
            }
            pick_stable(biased, std::false_type{});
        };
        if (dg.bias) pick_side(std::true_type{});
        else pick_side(std::false_type{});
        side_valid = write_side;
        if (scatter_ev) {
            MGTA_HIP_CHECK(hipEventRecord(e1, stream));
            scatter_ev->emplace_back(e0, e1);
            if (S) S->n_sort_launches++;
        }
        return 0;
    };
    auto pass_digit = [&](int i) {
        Digit dg = top_digit(WT, i);
        dg.pos -= tp.skip;
        dg.bias = tp.bias;
        return dg;
    };
    for (int i = P - 1; i >= 0; --i) {
        const Digit dg = pass_digit(i), dn = i > 0 ? pass_digit(i - 1) : Digit{0, 0, 0};
        if (global_pass(src, dst, n_items, dg, first_census_done && i == P - 1, i > 0 ? &dn : nullptr, unstable_first && i == P - 1)) return nullptr;
        if (side_failed) return nullptr;
        std::swap(src, dst);
    }
    const std::vector<Digit> low = low_plan_for(T);
    if (low.size() > 64) { set_error("too many sort digits"); return nullptr; }
    Digit *d_plan = pool_get<Digit>(ctx, S_PLAN, 64 * sizeof(Digit));
    MGTA_HIP_CHECK(hipMemcpyAsync(d_plan, low.data(), low.size() * sizeof(Digit), hipMemcpyHostToDevice, stream));
    // room behind the stride for the last segment of a tile: ~2.5 average segments, an eighth of the tile at least, half at most
    const double avg_seg = avg_segment_len(n_items, P, tp.frac);
    static const double margin_factor = getenv("MGTA_SORT_MARGIN") ? atof(getenv("MGTA_SORT_MARGIN")) : 2.5;      // (measurement knob)
    uint32_t margin = (uint32_t)std::min<double>(LocalCfg<WT>::kTile / 2, std::max<double>(LocalCfg<WT>::kTile / 8, margin_factor * avg_seg));
    margin = (margin + 63u) & ~63u;
    const uint32_t stride = (uint32_t)LocalCfg<WT>::kTile - margin;
    const uint64_t l_blocks = (n_items + stride - 1) / stride;
    if (l_blocks > 0x7FFFFFFFull) { set_error("too many sort tiles"); return nullptr; }
    const uint32_t big_cap = (uint32_t)l_blocks + 1;                  // every workgroup defers one segment at most
    uint64_t *d_big = pool_get<uint64_t>(ctx, S_BIG, (2 * (uint64_t)big_cap + 2) * 8);
    uint32_t *d_big_count = reinterpret_cast<uint32_t *>(d_big + 2 * (uint64_t)big_cap);
    MGTA_HIP_CHECK(hipMemsetAsync(d_big_count, 0, 8, stream));
    uint64_t *d_lsd = pool_get<uint64_t>(ctx, S_LSD, 2 * l_blocks * 8);
    hipEvent_t le0, le1;
    MGTA_HIP_CHECK(hipEventCreate(&le0));
    MGTA_HIP_CHECK(hipEventCreate(&le1));
    MGTA_HIP_CHECK(hipEventRecord(le0, stream));
    LocalPlan lp;
    lp.low_plan = d_plan;
    lp.n_low = (int)low.size();
    lp.T = T;
    // the comparison route is skipped for a few sorts after one that found mostly long runs (highly redundant input)
    const bool skip_a = ctx->lsd_skip_left > 0;
    if (skip_a) --ctx->lsd_skip_left;
    const int ub = ((ctx->force_lsd_tiles & 1) || skip_a) ? 0 : std::max(0, std::min(8, (WT > 1 ? 40 : 32) - T));
    lp.upper = Digit{32 * WT - T - ub, ub};
    lp.mask_last2 = mask_last2;
    lp.mask_last = mask_last;
    lp.stride = stride;
    lp.lsd_list = d_lsd;
    lp.lsd_count = d_big_count + 1;
    lp.lsd_cap = (uint32_t)l_blocks;
    lp.debug = ctx->force_lsd_tiles >> 1;
    if (ub > 0)
        hipLaunchKernelGGL((local_sort_kernel<WT, true>), dim3((unsigned)l_blocks), dim3(kSortThreads), 0, stream, src, n_items, lp, d_big, d_big_count,
                           big_cap);
    else
        hipLaunchKernelGGL((local_sort_kernel<WT, false>), dim3((unsigned)l_blocks), dim3(kSortThreads), 0, stream, src, n_items, lp, d_big, d_big_count,
                           big_cap);
    MGTA_HIP_CHECK(hipEventRecord(le1, stream));
    uint32_t n_big = 0, n_lsd_tiles = 0;
    MGTA_HIP_CHECK(hipMemcpyAsync(&n_lsd_tiles, d_big_count + 1, 4, hipMemcpyDeviceToHost, stream));
    MGTA_HIP_CHECK(hipMemcpyAsync(&n_big, d_big_count, 4, hipMemcpyDeviceToHost, stream));
    MGTA_HIP_CHECK(hipStreamSynchronize(stream));
    float ms_local = 0;
    MGTA_HIP_CHECK(hipEventElapsedTime(&ms_local, le0, le1));
    if (n_big > big_cap) { set_error("more than %u oversized key segments in one pass", big_cap); return nullptr; }
    if (n_lsd_tiles > lp.lsd_cap) { set_error("internal: tile list overflow"); return nullptr; }
    if (ub > 0 && l_blocks >= 64 && 2 * (uint64_t)n_lsd_tiles > l_blocks) ctx->lsd_skip_left = 7;   // then look again
    MGTA_HIP_CHECK(hipEventRecord(le0, stream));
    if (n_lsd_tiles > 0 && !(lp.debug & 2))
        hipLaunchKernelGGL((local_lsd_kernel<WT>), dim3(n_lsd_tiles), dim3(kSortThreads), 0, stream, src, d_lsd, d_lsd + 1, 2, lp);
    if (n_big > 0) {
        // segments that did not fit a tile next to their neighbours: alone in LDS if they fit, else (hot k-mers, or the whole array
        // when it is tiny) one workgroup each with global ping-pong passes
        hipLaunchKernelGGL((segment_end_kernel<WT>), dim3(n_big), dim3(256), 0, stream, src, n_items, T, d_big, d_big + big_cap);
        if (getenv("MGTA_SORT_DIAG")) {                               // (measurement aid) sizes of the deferred segments of this sort
            std::vector<uint64_t> s(n_big), e(n_big);
            MGTA_HIP_CHECK(hipStreamSynchronize(stream));
            MGTA_HIP_CHECK(hipMemcpy(s.data(), d_big, n_big * 8ull, hipMemcpyDeviceToHost));
            MGTA_HIP_CHECK(hipMemcpy(e.data(), d_big + big_cap, n_big * 8ull, hipMemcpyDeviceToHost));
            uint64_t hist[40] = {}, keys[40] = {}, mx = 0;
            for (uint32_t i = 0; i < n_big; ++i) {
                const uint64_t c = e[i] - s[i];
                const int b = c ? 64 - __builtin_clzll(c) : 0;
                hist[b]++; keys[b] += c; mx = std::max(mx, c);
            }
            fprintf(stderr, "[sort diag] %u deferred segments of %llu keys, longest %llu:", n_big, (unsigned long long)n_items, (unsigned long long)mx);
            for (int b = 0; b < 40; ++b)
                if (hist[b]) fprintf(stderr, " <2^%d: %llu (%llu keys)", b, (unsigned long long)hist[b], (unsigned long long)keys[b]);
            fprintf(stderr, "\n");
        }
        hipLaunchKernelGGL((local_lsd_kernel<WT>), dim3(n_big), dim3(kSortThreads), 0, stream, src, d_big, d_big + big_cap, 1, lp);
        static const bool big_tile = !getenv("MGTA_SORT_NO_BIG_TILE");  // (measurement knob)
        const bool use_big = BigCfg<WT>::kTile > 0 && big_tile && !low.empty();
        if (use_big) hipLaunchKernelGGL((segment_lds_kernel<WT>), dim3(n_big), dim3(kSortThreads), 0, stream, src, d_big, d_big + big_cap, lp);
        hipLaunchKernelGGL((segment_sort_kernel<WT>), dim3(n_big), dim3(kSortThreads), 0, stream, src, dst, d_big, d_big + big_cap, d_plan,
                           (int)low.size(), (uint32_t)(use_big ? BigCfg<WT>::kTile : LocalCfg<WT>::kTile));
    }
    MGTA_HIP_CHECK(hipEventRecord(le1, stream));
    MGTA_HIP_CHECK(hipEventSynchronize(le1));
    {
        float ms = 0;
        MGTA_HIP_CHECK(hipEventElapsedTime(&ms, le0, le1));
        if (S) { S->ms_local_sort += ms_local + ms; S->n_lsd_tiles += n_lsd_tiles; S->n_big_segments += n_big; }
        (void)hipEventDestroy(le0); (void)hipEventDestroy(le1);
    }
    return src;
}

// ---- stage 1 (min_count >= 2): fills the is_solid bit-vector (+ mercy edges) that stage 2 then honours -------------
template <int W>
static int run_stage1(mgta_ctx *ctx, const mgta_reads *rd, uint64_t n_short, int k, int min_count, int need_mercy, uint64_t budget,
                      unsigned long long **is_solid_out, int *num_k1_out, mgta_build_stats *S) {
    constexpr int WT = W + 2;                                          // key words (= words_per_substring of s1, s1.cpp:248) + 64-bit payload
    hipStream_t stream = ctx->stream;
    const uint64_t n_reads = rd->n_reads;
    const uint64_t n_blocks = (n_reads + kReadsPerBlock - 1) / kReadsPerBlock;
    uint64_t *d_small = pool_get<uint64_t>(ctx, S_SMALL, 4096);
    uint64_t *d_total = d_small;
    unsigned int *d_maxlen = reinterpret_cast<unsigned int *>(d_small + 300);
    unsigned long long *d_mercy_count = reinterpret_cast<unsigned long long *>(d_small + 302), *d_num_mercy = d_mercy_count + 1;
    MGTA_HIP_CHECK(hipMemsetAsync(d_small + 300, 0, 64, stream));
    if (n_short > n_reads) n_short = n_reads;
    if (n_short) hipLaunchKernelGGL(max_len_kernel, dim3((unsigned)((n_short + 255) / 256)), dim3(256), 0, stream, rd->d_start, n_short, d_maxlen);
    unsigned int max_len = 0;
    MGTA_HIP_CHECK(hipMemcpyAsync(&max_len, d_maxlen, 4, hipMemcpyDeviceToHost, stream));
    MGTA_HIP_CHECK(hipStreamSynchronize(stream));
    // base index -> read id table (one entry per 1024 bases, one past the end) and the left shift that puts the highest bit a mercy
    // candidate (base index << 2 | code) can have at bit 63
    if (n_reads >= 0xFFFFFFFFull) { set_error("stage 1: more than 2^32 reads"); return MGTA_EUNSUPPORTED; }
    const uint64_t bases_bound = rd->n_words * 16 + 16;
    const uint64_t n_pos_entries = (bases_bound >> kPosStepLog) + 2;
    uint32_t *d_pos2id = pool_get<uint32_t>(ctx, S_POS2ID, n_pos_entries * 4);
    hipLaunchKernelGGL(pos_to_id_kernel, dim3((unsigned)((n_pos_entries + 255) / 256)), dim3(256), 0, stream, rd->d_start, n_reads, n_pos_entries, d_pos2id);
    int cand_shift = 0;
    while (cand_shift < 40 && (((bases_bound << 2) | 3ull) << (cand_shift + 1)) >> (cand_shift + 1) == ((bases_bound << 2) | 3ull)) ++cand_shift;
    const int num_k1 = std::max<int>(0, (int)max_len - k);                 // num_k1_per_read, s1.cpp:152
    const uint64_t n_bits = (uint64_t)num_k1 * n_short;
    unsigned long long *d_solid = pool_get<unsigned long long>(ctx, S_SOLID, (n_bits / 64 + 2) * 8);
    MGTA_HIP_CHECK(hipMemsetAsync(d_solid, 0, (n_bits / 64 + 2) * 8, stream));
    unsigned long long *d_edge_count = pool_get<unsigned long long>(ctx, S_EDGE_COUNT, 65536 * 8);
    MGTA_HIP_CHECK(hipMemsetAsync(d_edge_count, 0, 65536 * 8, stream));
    uint32_t *d_block_count = pool_get<uint32_t>(ctx, S_BLOCK_COUNT, std::max<uint64_t>(1, n_blocks) * 4);
    uint64_t *d_block_base = pool_get<uint64_t>(ctx, S_BLOCK_BASE, std::max<uint64_t>(1, n_blocks) * 8);

    S1Args sa;
    sa.packed = rd->d_packed; sa.n_words = rd->n_words; sa.start = rd->d_start; sa.n_reads = n_reads; sa.k = k;
    sa.block_count = d_block_count; sa.block_base = d_block_base; sa.out = nullptr;
    Key<2> *d_mercy = nullptr;
    uint64_t mercy_cap = 0;
    int n_pass = 1;
    uint32_t b_lo = 0;
    while (b_lo < MGTA_NUM_BUCKETS) {
        uint32_t width = (MGTA_NUM_BUCKETS + n_pass - 1) / n_pass;
        uint32_t b_hi = std::min<uint32_t>(MGTA_NUM_BUCKETS, b_lo + width);
        sa.b_lo = b_lo; sa.b_hi = b_hi;
        uint64_t *d_scan_tmp = pool_get<uint64_t>(ctx, S_SCAN_TMP, scan_tmp_elems(std::max<uint64_t>(n_blocks, 1024)) * 8);
        if (n_blocks) hipLaunchKernelGGL((s1_scan_kernel<W, false>), dim3((unsigned)n_blocks), dim3(kScanBlock), 0, stream, sa);
        exclusive_scan_u32(stream, d_block_count, n_blocks, d_block_base, d_scan_tmp, d_total);
        uint64_t n_items = 0;
        MGTA_HIP_CHECK(hipMemcpyAsync(&n_items, d_total, 8, hipMemcpyDeviceToHost, stream));
        MGTA_HIP_CHECK(hipStreamSynchronize(stream));
        uint64_t n_tiles = (n_items + kBlockTile - 1) / kBlockTile;
        uint64_t need = 2 * n_items * sizeof(Key<WT>) + n_tiles * 256 * 8 + (need_mercy ? n_items * 8 : 0) + (8u << 20);
        uint64_t other = ctx->live_bytes - pool_bytes(ctx);
        if (need + need / 8 > budget - std::min<uint64_t>(budget, other) && width > 1) { n_pass *= 2; continue; }
        if (n_items > 0) {
            Key<WT> *d_a = pool_get<Key<WT>>(ctx, S_KEYS_A, n_items * sizeof(Key<WT>));
            Key<WT> *d_b = pool_get<Key<WT>>(ctx, S_KEYS_B, n_items * sizeof(Key<WT>));
            if (need_mercy && (!d_mercy || mercy_cap < 2 * n_items)) {
                // a candidate list that only grows: keep what earlier passes appended
                uint64_t have = 0;
                MGTA_HIP_CHECK(hipMemcpy(&have, d_mercy_count, 8, hipMemcpyDeviceToHost));
                uint64_t new_cap = have + 2 * n_items;
                DevBuf keep;
                if (have) { keep.alloc(have * 8); MGTA_HIP_CHECK(hipMemcpy(keep.p, d_mercy, have * 8, hipMemcpyDeviceToDevice)); }
                d_mercy = pool_get<Key<2>>(ctx, S_MERCY, new_cap * 8);
                if (have) MGTA_HIP_CHECK(hipMemcpy(d_mercy, keep.p, have * 8, hipMemcpyDeviceToDevice));
                mercy_cap = new_cap;
            }
            sa.out = d_a;
            hipLaunchKernelGGL((s1_scan_kernel<W, true>), dim3((unsigned)n_blocks), dim3(kScanBlock), 0, stream, sa);
            // sort on: key characters + head/tail (key words), then prev/next (low 6 bits of the payload); positions are ignored
            const int pad1 = 32 * W - 2 * (k - 1) - 6;
            auto low_plan = [&](int T) {
                std::vector<Digit> plan;
                int top = 32 * WT - T;
                plan.push_back(Digit{0, 6});
                if (top > 64) plan.push_back(Digit{64, std::min(6, top - 64)});
                for (int pos = 64 + 6 + pad1; pos < top; pos += 8) plan.push_back(Digit{pos, std::min(8, top - pos)});
                return plan;
            };
            const int max_top = std::min(4, std::max(0, (2 * (k - 1)) / 8));
            Key<WT> *sorted = device_sort<WT>(ctx, stream, d_a, d_b, n_items, max_top, low_plan, nullptr, nullptr, (double)(b_hi - b_lo) / MGTA_NUM_BUCKETS, 0u, 0x3Fu);
            if (!sorted) return MGTA_EUNSUPPORTED;
            char *scratch = reinterpret_cast<char *>(sorted == d_a ? d_b : d_a);
            uint64_t e_tiles = (n_items + kEmitTile - 1) / kEmitTile;
            uint32_t *d_tile_heads = pool_get<uint32_t>(ctx, S_TILE_HEADS, e_tiles * 4);
            uint64_t *d_tile_base = pool_get<uint64_t>(ctx, S_TILE_BASE, e_tiles * 8);
            d_scan_tmp = pool_get<uint64_t>(ctx, S_SCAN_TMP, scan_tmp_elems(std::max<uint64_t>(n_blocks, e_tiles)) * 8);
            hipLaunchKernelGGL((s1_mark_kernel<W>), dim3((unsigned)e_tiles), dim3(kEmitThreads), 0, stream, sorted, n_items, d_tile_heads);
            exclusive_scan_u32(stream, d_tile_heads, e_tiles, d_tile_base, d_scan_tmp, d_total);
            uint64_t m = 0;
            MGTA_HIP_CHECK(hipMemcpyAsync(&m, d_total, 8, hipMemcpyDeviceToHost, stream));
            MGTA_HIP_CHECK(hipStreamSynchronize(stream));
            uint64_t *run_start = reinterpret_cast<uint64_t *>(scratch);               // 12 bytes per run <= sizeof(Key<WT>) per key
            uint16_t *run_info = reinterpret_cast<uint16_t *>(scratch + m * 8);
            uint16_t *group_mask = reinterpret_cast<uint16_t *>(scratch + m * 10);
            hipLaunchKernelGGL((s1_compact_kernel<W>), dim3((unsigned)e_tiles), dim3(kEmitThreads), 0, stream, sorted, n_items, k, d_tile_base,
                               run_start, run_info);
            unsigned rb = (unsigned)((m + 255) / 256);
            hipLaunchKernelGGL(s1_group_kernel, dim3(rb), dim3(256), 0, stream, run_start, run_info, m, n_items, min_count, group_mask);
            hipLaunchKernelGGL((s1_apply_kernel<W>), dim3(std::min(rb, 8192u)), dim3(256), 0, stream, sorted, run_start, run_info, group_mask, m, n_items, min_count,
                               k, rd->d_start, n_reads, n_short, num_k1, d_solid, d_edge_count, d_mercy, d_mercy_count, mercy_cap, need_mercy,
                               d_pos2id, cand_shift);
        }
        b_lo = b_hi;
    }
    // .counting histogram (s1_post_proc, s1.cpp:905-930)
    ctx->edge_counting.assign(65536, 0);
    MGTA_HIP_CHECK(hipMemcpyAsync(ctx->edge_counting.data(), d_edge_count, 65536 * 8, hipMemcpyDeviceToHost, stream));
    MGTA_HIP_CHECK(hipStreamSynchronize(stream));
    if (need_mercy) {
        uint64_t n_cand = 0;
        MGTA_HIP_CHECK(hipMemcpy(&n_cand, d_mercy_count, 8, hipMemcpyDeviceToHost));
        if (n_cand > mercy_cap) { set_error("mercy candidate list overflow"); return MGTA_ENOMEM; }
        if (n_cand > 0) {
            Key<2> *d_tmp = pool_get<Key<2>>(ctx, S_KEYS_A, n_cand * 8);
            auto low64 = [&](int T) {
                std::vector<Digit> plan;
                for (int pos = 0; pos < 64 - T; pos += 8) plan.push_back(Digit{pos, std::min(8, 64 - T - pos)});
                return plan;
            };
            Key<2> *cs = device_sort<2>(ctx, stream, d_mercy, d_tmp, n_cand, 4, low64, nullptr, nullptr);
            if (!cs) return MGTA_EUNSUPPORTED;
            if ((int)max_len <= kMercyMaxLen) {
                hipLaunchKernelGGL(mercy_kernel<false>, dim3((unsigned)((n_cand + 255) / 256)), dim3(256), 0, stream, cs, n_cand, rd->d_start, n_reads, k,
                                   num_k1, d_solid, d_num_mercy, d_pos2id, cand_shift, (uint8_t *)nullptr, (uint64_t)0);
                MGTA_HIP_CHECK(hipGetLastError());
                MGTA_HIP_CHECK(hipStreamSynchronize(stream));
            } else {     // reads longer than the LDS flags hold: the flags of every wave go to device memory, a fixed grid walks the groups
                const uint64_t stride = ((uint64_t)max_len + 64 + 63) & ~63ull;
                const unsigned grid = (unsigned)std::min<uint64_t>((n_cand + 255) / 256, (uint64_t)ctx->num_cus * 8);
                DevBuf d_flags;
                d_flags.alloc((uint64_t)grid * 4 * 3 * stride, &ctx->live_bytes, &ctx->peak_bytes);
                hipLaunchKernelGGL(mercy_kernel<true>, dim3(grid), dim3(256), 0, stream, cs, n_cand, rd->d_start, n_reads, k, num_k1, d_solid,
                                   d_num_mercy, d_pos2id, cand_shift, d_flags.as<uint8_t>(), stride);
                MGTA_HIP_CHECK(hipGetLastError());
                MGTA_HIP_CHECK(hipStreamSynchronize(stream));
            }
        }
    }
    (void)S;
    *is_solid_out = d_solid;
    *num_k1_out = num_k1;
    return MGTA_OK;
}

template <int W>
static int build_impl(mgta_ctx *ctx, const mgta_reads *rd, uint64_t n_short, int k, int min_count, int need_mercy, uint32_t bucket_begin,
                      uint32_t bucket_end, mgta_edge_sink sink, void *user, mgta_build_stats *st) {
    hipStream_t stream = ctx->stream;
    MGTA_HIP_CHECK(hipSetDevice(ctx->device));
    const int words_per_tip = (2 * k + 31) / 32;                       // sdbg_multi_io.h:63
    const uint64_t n_reads = rd->n_reads;
    const uint64_t n_blocks = (n_reads + kReadsPerBlock - 1) / kReadsPerBlock;
    // the scan kernels run one workgroup of kScanBlock threads per kReadsPerBlock reads: a dispatch holds fewer than 2^32 work-items
    if (n_blocks * (uint64_t)kScanBlock >= (1ull << 32)) { set_error("too many reads for one launch (%llu)", (unsigned long long)n_reads); return MGTA_EUNSUPPORTED; }
    mgta_build_stats S;
    memset(&S, 0, sizeof(S));
    S.k = k; S.words_per_key = W; S.words_per_tip = words_per_tip; S.n_reads = (int64_t)n_reads;
    ctx->peak_bytes = ctx->live_bytes;

    size_t free_b = 0, total_b = 0;
    MGTA_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    // what the build may hold: the explicit limit, else 90 % of (free + what our pool already holds)
    uint64_t avail = (uint64_t)free_b + pool_bytes(ctx);
    uint64_t budget = ctx->mem_limit ? std::min<uint64_t>(ctx->mem_limit, avail) : (uint64_t)(avail * 0.9);

    // whole-stream hand-off of a multi-pass build (mgta_ctx_keep_stream)
    const bool acc = ctx->keep_stream != 0;           // (of a bucket sub-range too: the shard a rank hands to the all-gather)
    const double range_frac = (double)(bucket_end - bucket_begin) / (double)MGTA_NUM_BUCKETS;
    ctx->acc_valid = false;
    ctx->acc_n_rec = 0; ctx->acc_n_tips = 0;
    if (acc) ctx->acc_items.assign(MGTA_NUM_BUCKETS, 0);
    std::vector<int64_t> acc_first;

    Timer t_all(stream), t_ph(stream);
    std::vector<std::pair<hipEvent_t, hipEvent_t>> scatter_ev;
    uint32_t *d_block_count = pool_get<uint32_t>(ctx, S_BLOCK_COUNT, std::max<uint64_t>(1, n_blocks) * 4);
    uint64_t *d_block_base = pool_get<uint64_t>(ctx, S_BLOCK_BASE, std::max<uint64_t>(1, n_blocks) * 8);
    uint64_t *d_small = pool_get<uint64_t>(ctx, S_SMALL, 4096);       // [0] total, [1] kmers, [2..4] emit totals, [5] sentinels, [8..263] digit totals
    uint64_t *d_total = d_small, *d_kmers = d_small + 1, *d_tot3 = d_small + 2, *d_sentinel = d_small + 5;

    ScanArgs sa;
    sa.packed = rd->d_packed; sa.n_words = rd->n_words; sa.start = rd->d_start; sa.n_reads = n_reads; sa.k = k;
    sa.block_count = d_block_count; sa.block_base = d_block_base; sa.out = nullptr;
    sa.n_kmers = (unsigned long long *)d_kmers;
    sa.n_sentinel = (unsigned long long *)d_sentinel;
    sa.is_solid = nullptr; sa.num_k1_per_read = 0; sa.n_short = n_short;
    sa.side = nullptr; sa.side_shift = 0; sa.side_bias = 0;
    if (min_count > 1) {
        unsigned long long *sol = nullptr;
        int nk1 = 0;
        int rc1 = MGTA_EUNSUPPORTED;
        Timer t_s1(stream);
        t_s1.start();
        if constexpr (W <= 9) rc1 = run_stage1<W>(ctx, rd, n_short, k, min_count, need_mercy, budget, &sol, &nk1, &S);
        else set_error("min_count > 1: a key of %d words is not supported", W);
        if (rc1 != MGTA_OK) return rc1;
        S.ms_stage1 = t_s1.stop();
        sa.is_solid = sol; sa.num_k1_per_read = nk1;
        // run_stage1 may have re-grown pool slots: refresh the small pointers
        d_block_count = pool_get<uint32_t>(ctx, S_BLOCK_COUNT, std::max<uint64_t>(1, n_blocks) * 4);
        d_block_base = pool_get<uint64_t>(ctx, S_BLOCK_BASE, std::max<uint64_t>(1, n_blocks) * 8);
        sa.block_count = d_block_count; sa.block_base = d_block_base;
    }

    t_all.start();
    int n_pass = 1;
    uint32_t b_lo = bucket_begin;
    const uint32_t span = bucket_end - bucket_begin;
    bool first_pass = true;
    std::vector<int64_t> h_first;
    std::vector<uint16_t> h_rec, h_large;
    std::vector<uint32_t> h_tips;
    std::vector<int64_t> h_items;

    uint32_t multi_n = 0, multi_width = 0, multi_lo = 0;             // ranges counted ahead by one scan (d_multi_count rows)
    uint32_t *d_multi_count = nullptr;
    while (b_lo < bucket_end) {
        uint32_t width = (span + n_pass - 1) / n_pass;
        uint32_t b_hi = std::min<uint32_t>(bucket_end, b_lo + width);
        // ---- 1. count
        t_ph.start();
        sa.b_lo = b_lo; sa.b_hi = b_hi;
        sa.n_kmers = first_pass ? (unsigned long long *)d_kmers : nullptr;
        if (first_pass) MGTA_HIP_CHECK(hipMemsetAsync(d_kmers, 0, 8, stream));
        uint64_t *d_scan_tmp = pool_get<uint64_t>(ctx, S_SCAN_TMP, scan_tmp_elems(std::max<uint64_t>(n_blocks, 1024)) * 8);
        static_assert(kReadsPerBlock == 64, "item_count_closed_kernel: one lane per read of a workgroup");
        static const bool closed_even = !(getenv("MGTA_CLOSED_EVEN") && atoi(getenv("MGTA_CLOSED_EVEN")) == 0);   // 0: k+1 even takes the scans
        const bool closed_form = (((k + 1) & 1) || closed_even) && !sa.is_solid && b_lo == 0 && b_hi == (uint32_t)MGTA_NUM_BUCKETS && !ctx->force_full_lsd;
        const uint32_t *counts = d_block_count;
        sa.multi_n = 0; sa.multi_width = 0; sa.multi_magic = 0;
        if (n_blocks && closed_form)
            hipLaunchKernelGGL(item_count_closed_kernel, dim3((unsigned)((n_blocks + 63) / 64)), dim3(256), 0, stream, sa.start, sa.n_reads, n_blocks, k,
                               sa.block_count, sa.n_kmers);
        else if (n_blocks) {
            // the k-mer total of the first pass from the read lengths alone: one atomic per wave of the scan on ONE address (6.25 M of
            // them at 100 M reads) kept the count scan at 80 ms where the write scan, which does more, takes 52
            if (sa.n_kmers) {
                hipLaunchKernelGGL(item_count_closed_kernel, dim3((unsigned)((n_blocks + 63) / 64)), dim3(256), 0, stream, sa.start, sa.n_reads, n_blocks, k,
                                   (uint32_t *)nullptr, sa.n_kmers);
                sa.n_kmers = nullptr;
            }
            // several equally wide ranges ahead (memory-bound passes): one scan counts them all
            const uint32_t ranges_left = (bucket_end - b_lo + width - 1) / width;
            if (multi_n && (width != multi_width || b_lo < multi_lo || (b_lo - multi_lo) % width != 0 || (b_lo - multi_lo) / width >= multi_n)) multi_n = 0;
            if (!multi_n && ranges_left > 1 && ranges_left <= (uint32_t)kMaxCountRanges) {
                d_multi_count = pool_get<uint32_t>(ctx, S_MULTI_COUNT, (uint64_t)ranges_left * n_blocks * 4);
                ScanArgs sm = sa;
                sm.block_count = d_multi_count; sm.b_hi = bucket_end; sm.multi_width = width; sm.multi_n = ranges_left; sm.multi_magic = ((1ull << 32) + width - 1) / width;
                hipLaunchKernelGGL((item_scan_kernel<W, false>), dim3((unsigned)n_blocks), dim3(kScanBlock), 0, stream, sm);
                multi_n = ranges_left; multi_width = width; multi_lo = b_lo;
            }
            if (multi_n) counts = d_multi_count + (uint64_t)((b_lo - multi_lo) / width) * n_blocks;
            else hipLaunchKernelGGL((item_scan_kernel<W, false>), dim3((unsigned)n_blocks), dim3(kScanBlock), 0, stream, sa);
        }
        exclusive_scan_u32(stream, counts, n_blocks, d_block_base, d_scan_tmp, d_total);
        uint64_t n_items = 0;
        MGTA_HIP_CHECK(hipMemcpyAsync(&n_items, d_total, 8, hipMemcpyDeviceToHost, stream));
        S.ms_count += t_ph.stop();
        if (first_pass) {
            unsigned long long km = 0;
            MGTA_HIP_CHECK(hipMemcpy(&km, d_kmers, 8, hipMemcpyDeviceToHost));
            S.n_kmers = (int64_t)km;
        }
        // does the pass fit?  two key buffers (the second doubles as emit scratch) + census + outputs (estimate)
        uint64_t n_tiles = (n_items + kBlockTile - 1) / kBlockTile;
        // either key buffer may end up as the emitter's scratch (11 bytes per key: run start u64, record u16, info u8), whichever
        // the last sort pass leaves idle: both hold >= 12 bytes per key
        uint64_t key_b = std::max<uint64_t>(n_items * sizeof(Key<W>), n_items * 12) + 4096;
        uint64_t need = 2 * key_b + n_tiles * 256 * 8 + n_items * 2 + n_items /* side digits */ + (8u << 20);
        uint64_t other = ctx->live_bytes - pool_bytes(ctx);
        if (acc) {     // room for the stream the passes leave behind: ~0.6 edges of 2 bytes per (k+1)-mer, tips, slack
            const uint64_t est = (uint64_t)((double)S.n_kmers * 1.5 * range_frac) + (64ull << 20);
            const uint64_t have = ctx->acc_rec.bytes + ctx->acc_tips.bytes;
            other += est > have ? est - have : 0;
        }
        const uint64_t avail = budget - std::min<uint64_t>(budget, other);
        if (need + need / 8 > avail && width > 1) {                   // narrower bucket ranges (CX1's lv1 loop, cx1.h:494)
            const double ratio = (double)(need + need / 8) / (double)std::max<uint64_t>(avail, 1) * 1.03;
            n_pass = std::max(n_pass + 1, (int)std::ceil((double)n_pass * ratio));
            continue;
        }
        first_pass = false;
        S.n_items += (int64_t)n_items;
        S.n_passes++;
        const uint32_t nb = b_hi - b_lo;
        h_items.assign((size_t)nb * 3, 0);
        uint64_t n_edges = 0, n_large = 0, n_tips = 0;
        if (n_items > 0) {
            Key<W> *d_a = pool_get<Key<W>>(ctx, S_KEYS_A, key_b);
            Key<W> *d_b = pool_get<Key<W>>(ctx, S_KEYS_B, key_b);
            // ---- 3. write keys
            t_ph.start();
            sa.out = d_a;
            const int max_top = (2 * k + 4 + 7) / 8 > 1 ? std::min(4, (32 * W - 8) / 8) : 0;
            const double prefix_frac = (double)(b_hi - b_lo) / MGTA_NUM_BUCKETS;
            // closed form + at least one global sort pass: the key writer works tile by tile of that pass and leaves its census behind
            const int P_top = choose_top_plan(ctx, n_items, max_top, prefix_frac).P;
            static const bool tiled_keygen = !(getenv("MGTA_KEYGEN_TILED") && atoi(getenv("MGTA_KEYGEN_TILED")) == 0);
            const bool fused_census = closed_form && P_top >= 1 && tiled_keygen && n_tiles <= 0x7FFFFFFFull;
            bool first_side = false;
            if (closed_form) MGTA_HIP_CHECK(hipMemsetAsync(d_sentinel, 0, 8, stream));
            if (fused_census) {
                uint64_t *d_hist = pool_get<uint64_t>(ctx, S_HIST, std::max<uint64_t>(1, n_tiles) * 256 * 8);   // the buffer device_sort uses
                hipLaunchKernelGGL((item_write_tiled_kernel<W>), dim3((unsigned)n_tiles), dim3(kScanBlock), 0, stream, sa, n_blocks, n_items,
                                   32 - 8 * P_top, n_tiles, d_hist);
            } else if (closed_form)
                hipLaunchKernelGGL((item_write_closed_kernel<W>), dim3((unsigned)n_blocks), dim3(kScanBlock), 0, stream, sa);
            else {
                // the general writer leaves the first global pass's digit of every key in the side array (see device_sort): that pass's
                // census then reads one byte per key.  The plan is the one device_sort is about to choose (same arguments).
                const TopPlan tp = choose_top_plan(ctx, n_items, max_top, prefix_frac, b_lo, b_hi);
                const char *side_env = getenv("MGTA_SORT_SIDE");
                if (tp.P >= 2 && kSideFits<W> && (!side_env || atoi(side_env) > 0)) {
                    sa.side = pool_get<uint8_t>(ctx, S_SIDE, n_items + 64);
                    sa.side_shift = 32 - 8 * tp.P - tp.skip;
                    sa.side_bias = tp.bias;
                    first_side = true;
                }
                hipLaunchKernelGGL((item_scan_kernel<W, true>), dim3((unsigned)n_blocks), dim3(kScanBlock), 0, stream, sa);
                sa.side = nullptr;
            }
            S.ms_gen += t_ph.stop();
            // ---- 4. sort: P global passes on the most significant bytes, then the segment-local finish in LDS
            t_ph.start();
            Key<W> *src = device_sort<W>(ctx, stream, d_a, d_b, n_items, max_top, [&](int T) { return low_digit_plan(k, W, T); }, &scatter_ev, &S,
                                           prefix_frac, ~0u, ~0u, fused_census, b_lo, b_hi, first_side);
            if (!src) return MGTA_EUNSUPPORTED;
            Key<W> *dst = src == d_a ? d_b : d_a;
            S.ms_sort += t_ph.stop();
            if (closed_form && !((k + 1) & 1)) {
                // sentinel keys (rc slots of palindromic (k+1)-mers) sorted behind every real key: the emitter stops before them
                unsigned long long n_sent = 0;
                MGTA_HIP_CHECK(hipMemcpy(&n_sent, d_sentinel, 8, hipMemcpyDeviceToHost));
                if (n_sent >= n_items) { set_error("internal: %llu sentinel keys among %llu items", n_sent, (unsigned long long)n_items); return MGTA_EINTERNAL; }
                n_items -= n_sent;
                S.n_items -= (int64_t)n_sent;
            }
            // ---- 5. emit.  `src` holds the sorted keys; the other buffer is scratch.
            t_ph.start();
            const Key<W> *sorted = src;
            char *scratch = reinterpret_cast<char *>(dst);
            uint64_t e_tiles = (n_items + kEmitTile - 1) / kEmitTile;
            d_scan_tmp = pool_get<uint64_t>(ctx, S_SCAN_TMP, scan_tmp_elems(std::max<uint64_t>(std::max(n_blocks, e_tiles), n_items / kDecideTile + 1)) * 8);
            // run descriptors, compacted in key order in one read of the keys (chained scan over the tiles).  Scratch layout for up to
            // n_items runs (<= 7.01 bytes per key of a >= 12-byte-per-key buffer): start u32 | rec u16 | info u8 | full start u64 per 1024 runs
            if (e_tiles > 0xFFFFFFFFull) { set_error("too many emit tiles"); return MGTA_EUNSUPPORTED; }
            unsigned long long *d_chain = pool_get<unsigned long long>(ctx, S_TILE_BASE, (e_tiles + 4) * 8);
            EmitChain chain;
            chain.state = d_chain;
            chain.total = d_chain + e_tiles;
            chain.ticket = reinterpret_cast<uint32_t *>(d_chain + e_tiles + 1);
            chain.error = reinterpret_cast<uint32_t *>(d_chain + e_tiles + 2);
            RunStarts sub_start;
            sub_start.lo = reinterpret_cast<uint32_t *>(scratch);
            uint16_t *rec = reinterpret_cast<uint16_t *>(scratch + n_items * 4);
            uint8_t *info = reinterpret_cast<uint8_t *>(scratch + n_items * 6);
            sub_start.base = reinterpret_cast<uint64_t *>(scratch + ((n_items * 7 + 7) & ~7ull));
            unsigned long long chain_out[3] = {0, 0, 0};                   // total, ticket, error
            for (int attempt = 0; attempt < 2; ++attempt) {
                MGTA_HIP_CHECK(hipMemsetAsync(d_chain, 0, (e_tiles + 4) * 8, stream));
                if (attempt == 0 && !ctx->force_full_lsd)
                    hipLaunchKernelGGL((emit_compact_kernel<W, false>), dim3((unsigned)e_tiles), dim3(kEmitThreads), 0, stream, sorted, n_items, k, chain,
                                       (uint32_t)e_tiles, sub_start, info);
                else
                    hipLaunchKernelGGL((emit_compact_kernel<W, true>), dim3((unsigned)e_tiles), dim3(kEmitThreads), 0, stream, sorted, n_items, k, chain,
                                       (uint32_t)e_tiles, sub_start, info);
                MGTA_HIP_CHECK(hipMemcpyAsync(chain_out, d_chain + e_tiles, 24, hipMemcpyDeviceToHost, stream));
                MGTA_HIP_CHECK(hipStreamSynchronize(stream));
                if ((uint32_t)chain_out[2] == 0) break;
            }
            if ((uint32_t)chain_out[2] != 0) { set_error("internal: chained scan of the emitter timed out"); return MGTA_EINTERNAL; }
            const uint64_t m = chain_out[0];
            uint64_t d_tiles = (m + kDecideTile - 1) / kDecideTile;
            uint32_t *ce = pool_get<uint32_t>(ctx, S_CNT, d_tiles * 4 * 3), *cl = ce + d_tiles, *ct = cl + d_tiles;
            uint64_t *be = pool_get<uint64_t>(ctx, S_BASE, d_tiles * 8 * 3), *bl = be + d_tiles, *bt = bl + d_tiles;
            int64_t *d_first = pool_get<int64_t>(ctx, S_FIRST, (uint64_t)MGTA_NUM_BUCKETS * 3 * 8);
            MGTA_HIP_CHECK(hipMemsetAsync(d_first, 0xFF, (uint64_t)nb * 3 * 8, stream));
            hipLaunchKernelGGL(emit_decide_kernel, dim3((unsigned)d_tiles), dim3(kDecideThreads), 0, stream, sub_start, info, m, n_items,
                               rec, ce, cl, ct);
            exclusive_scan_u32(stream, ce, d_tiles, be, d_scan_tmp, d_tot3);
            exclusive_scan_u32(stream, cl, d_tiles, bl, d_scan_tmp, d_tot3 + 1);
            exclusive_scan_u32(stream, ct, d_tiles, bt, d_scan_tmp, d_tot3 + 2);
            uint64_t tot3[3];
            MGTA_HIP_CHECK(hipMemcpyAsync(tot3, d_tot3, 24, hipMemcpyDeviceToHost, stream));
            MGTA_HIP_CHECK(hipStreamSynchronize(stream));
            n_edges = tot3[0]; n_large = tot3[1]; n_tips = tot3[2];
            uint16_t *d_out_rec = pool_get<uint16_t>(ctx, S_OUT_REC, n_edges * 2);
            uint16_t *d_out_large = pool_get<uint16_t>(ctx, S_OUT_LARGE, n_large * 2);
            uint32_t *d_out_tips = pool_get<uint32_t>(ctx, S_OUT_TIPS, n_tips * words_per_tip * 4);
            hipLaunchKernelGGL((emit_write_kernel<W>), dim3((unsigned)d_tiles), dim3(kDecideThreads), 0, stream, sorted, sub_start, info, rec, m,
                               n_items, be, bl, bt, words_per_tip, b_lo, d_out_rec, d_out_large, d_out_tips, d_first);
            S.ms_emit += t_ph.stop();
            ctx->last_rec = d_out_rec; ctx->last_n_rec = n_edges; ctx->last_bucket_lo = b_lo; ctx->last_bucket_hi = b_hi;
            ctx->last_tips = d_out_tips; ctx->last_n_tips = n_tips; ctx->last_first = d_first; ctx->last_k = k; ctx->last_words_per_tip = words_per_tip;
            if (acc) {
                // append this pass to the whole-stream buffers (device to device); capacity from the share of the buckets done so far
                auto ensure = [&](DevBuf &buf, uint64_t used, uint64_t add) {
                    if (used + add <= buf.bytes) return;
                    const double done = (double)(b_hi - bucket_begin) / (double)(bucket_end - bucket_begin);
                    uint64_t want = (uint64_t)((double)(used + add) / std::max(done, 1e-3) * 1.1) + (1u << 20);
                    want = std::max<uint64_t>(want, used + add);
                    DevBuf bigger;
                    bigger.alloc(want, &ctx->live_bytes, &ctx->peak_bytes);
                    if (used) MGTA_HIP_CHECK(hipMemcpyAsync(bigger.p, buf.p, used, hipMemcpyDeviceToDevice, stream));
                    MGTA_HIP_CHECK(hipStreamSynchronize(stream));
                    buf = std::move(bigger);
                };
                ensure(ctx->acc_rec, ctx->acc_n_rec * 2, n_edges * 2);
                ensure(ctx->acc_tips, ctx->acc_n_tips * words_per_tip * 4, n_tips * words_per_tip * 4);
                if (n_edges) MGTA_HIP_CHECK(hipMemcpyAsync(ctx->acc_rec.as<char>() + ctx->acc_n_rec * 2, d_out_rec, n_edges * 2, hipMemcpyDeviceToDevice, stream));
                if (n_tips) MGTA_HIP_CHECK(hipMemcpyAsync(ctx->acc_tips.as<char>() + ctx->acc_n_tips * words_per_tip * 4, d_out_tips,
                                                          n_tips * words_per_tip * 4, hipMemcpyDeviceToDevice, stream));
                acc_first.resize((size_t)nb * 3);
                MGTA_HIP_CHECK(hipMemcpyAsync(acc_first.data(), d_first, (size_t)nb * 3 * 8, hipMemcpyDeviceToHost, stream));
                MGTA_HIP_CHECK(hipStreamSynchronize(stream));
                int64_t nxt = (int64_t)n_edges;
                for (int64_t b = (int64_t)nb - 1; b >= 0; --b) {
                    int64_t f = acc_first[(size_t)b * 3];
                    if (f < 0) f = nxt;
                    ctx->acc_items[(size_t)b_lo + (size_t)b] = nxt - f;
                    nxt = f;
                }
                ctx->acc_n_rec += n_edges; ctx->acc_n_tips += n_tips;
            }
            // ---- device -> host
            if (sink) {
                t_ph.start();
                // keep_stream 2: records and tip labels stay on the device only (the caller takes the whole stream afterwards,
                // mgta_sdbg_stream_detach / mgta_stream_download); the sink still gets the counts and the large multiplicities
                const bool to_host = ctx->keep_stream != 2;
                h_rec.resize(to_host ? n_edges : 0); h_large.resize(n_large); h_tips.resize(to_host ? n_tips * words_per_tip : 0);
                h_first.resize((size_t)nb * 3);
                if (n_edges && to_host) MGTA_HIP_CHECK(hipMemcpyAsync(h_rec.data(), d_out_rec, n_edges * 2, hipMemcpyDeviceToHost, stream));
                if (n_large) MGTA_HIP_CHECK(hipMemcpyAsync(h_large.data(), d_out_large, n_large * 2, hipMemcpyDeviceToHost, stream));
                if (n_tips && to_host) MGTA_HIP_CHECK(hipMemcpyAsync(h_tips.data(), d_out_tips, n_tips * words_per_tip * 4, hipMemcpyDeviceToHost, stream));
                MGTA_HIP_CHECK(hipMemcpyAsync(h_first.data(), d_first, (size_t)nb * 3 * 8, hipMemcpyDeviceToHost, stream));
                S.ms_d2h += t_ph.stop();
                // bucket boundaries -> counts; untouched entries (-1) are empty buckets
                int64_t nxt[3] = {(int64_t)n_edges, (int64_t)n_large, (int64_t)n_tips};
                for (int64_t b = (int64_t)nb - 1; b >= 0; --b)
                    for (int c = 0; c < 3; ++c) {
                        int64_t f = h_first[(size_t)b * 3 + c];
                        if (f < 0) f = nxt[c];
                        h_items[(size_t)b * 3 + c] = nxt[c] - f;
                        nxt[c] = f;
                    }
            }
        } else {
            h_rec.clear(); h_large.clear(); h_tips.clear();
            ctx->last_rec = nullptr; ctx->last_n_rec = 0; ctx->last_bucket_lo = b_lo; ctx->last_bucket_hi = b_hi;
            ctx->last_tips = nullptr; ctx->last_n_tips = 0; ctx->last_first = nullptr; ctx->last_k = k; ctx->last_words_per_tip = words_per_tip;
        }
        S.n_edges += (int64_t)n_edges; S.n_large += (int64_t)n_large; S.n_tips += (int64_t)n_tips;
        if (sink) {
            const bool to_host = ctx->keep_stream != 2;
            int rc = sink(user, (int32_t)b_lo, (int32_t)b_hi, h_items.data(), to_host ? h_rec.data() : nullptr, (int64_t)n_edges, h_large.data(),
                          (int64_t)n_large, to_host ? h_tips.data() : nullptr, (int64_t)(n_tips * words_per_tip));
            if (rc != 0) { set_error("edge sink returned %d", rc); return MGTA_ESINK; }
        }
        b_lo = b_hi;
    }
    if (acc) {
        ctx->last_rec = ctx->acc_rec.p; ctx->last_n_rec = ctx->acc_n_rec; ctx->last_bucket_lo = bucket_begin; ctx->last_bucket_hi = bucket_end;
        ctx->last_tips = ctx->acc_tips.p; ctx->last_n_tips = ctx->acc_n_tips; ctx->last_first = nullptr; ctx->last_k = k;
        ctx->last_words_per_tip = words_per_tip;
        ctx->acc_valid = true;
    }
    S.ms_total = t_all.stop();
    for (auto &ev : scatter_ev) {
        float ms = 0;
        MGTA_HIP_CHECK(hipEventElapsedTime(&ms, ev.first, ev.second));
        S.ms_sort_scatter += ms;
        (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second);
    }
    S.bytes_peak = ctx->peak_bytes;
    if (st) *st = S;
    return MGTA_OK;
}

}  // namespace mgta

using namespace mgta;

extern "C" {

int mgta_reads_upload(mgta_ctx *ctx, const uint32_t *packed, uint64_t n_words, const uint64_t *start_idx, uint64_t n_reads,
                      mgta_reads **out) {
    if (!ctx || !packed || !start_idx || !out) { set_error("mgta_reads_upload: null argument"); return MGTA_EINVAL; }
    try {
        MGTA_HIP_CHECK(hipSetDevice(ctx->device));
        auto r = std::make_unique<mgta_reads>();
        r->ctx = ctx; r->n_words = n_words; r->n_reads = n_reads;
        r->own_packed.alloc((n_words + 16) * 4, &ctx->live_bytes, &ctx->peak_bytes);
        r->own_start.alloc((n_reads + 1) * 8, &ctx->live_bytes, &ctx->peak_bytes);
        MGTA_HIP_CHECK(hipMemsetAsync((char *)r->own_packed.p + n_words * 4, 0, 64, ctx->stream));
        MGTA_HIP_CHECK(hipMemcpyAsync(r->own_packed.p, packed, n_words * 4, hipMemcpyHostToDevice, ctx->stream));
        MGTA_HIP_CHECK(hipMemcpyAsync(r->own_start.p, start_idx, (n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        MGTA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        r->d_packed = r->own_packed.as<uint32_t>();
        r->d_start = r->own_start.as<uint64_t>();
        ctx_retain(ctx);
        *out = r.release();
        return MGTA_OK;
    } catch (const HipError &e) { return e.code; }
}

int mgta_reads_adopt_device(mgta_ctx *ctx, const uint32_t *d_packed, uint64_t n_words, const uint64_t *d_start, uint64_t n_reads,
                            mgta_reads **out) {
    if (!ctx || !d_packed || !d_start || !out) { set_error("mgta_reads_adopt_device: null argument"); return MGTA_EINVAL; }
    auto *r = new mgta_reads;
    r->ctx = ctx; r->d_packed = d_packed; r->d_start = d_start; r->n_words = n_words; r->n_reads = n_reads;
    ctx_retain(ctx);
    *out = r;
    return MGTA_OK;
}

void mgta_reads_free(mgta_reads *r) {
    if (!r) return;
    mgta_ctx *c = r->ctx;
    delete r;
    ctx_release(c);
}

int mgta_sdbg_build_resident(mgta_ctx *ctx, const mgta_reads *rd, uint64_t n_short_reads, int k, int min_count, int need_mercy,
                             int32_t bucket_begin, int32_t bucket_end, mgta_edge_sink sink, void *user, mgta_build_stats *stats) {
    if (!ctx || !rd) { set_error("mgta_sdbg_build: null argument"); return MGTA_EINVAL; }
    if (k < 9 || k > 127) { set_error("k=%d out of range [9,127] (kMaxK, definitions.h:56)", k); return MGTA_EINVAL; }
    if (bucket_begin < 0 || bucket_end > MGTA_NUM_BUCKETS || bucket_begin >= bucket_end) {
        set_error("bucket range [%d,%d) invalid", bucket_begin, bucket_end);
        return MGTA_EINVAL;
    }
    if (min_count < 1) { set_error("min_count must be >= 1"); return MGTA_EINVAL; }
    if (min_count > 1 && (bucket_begin != 0 || bucket_end != MGTA_NUM_BUCKETS) ) {
        // stage 1 always covers every bucket (its verdicts feed every stage-2 bucket); a stage-2 shard is fine
    }
    try {
        int W = (2 * k + 4 + 31) / 32;                                 // words_per_substring, s2.cpp:331
        switch (W) {
        case 1: return build_impl<1>(ctx, rd, n_short_reads, k, min_count, need_mercy, (uint32_t)bucket_begin, (uint32_t)bucket_end, sink, user, stats);
        case 2: return build_impl<2>(ctx, rd, n_short_reads, k, min_count, need_mercy, (uint32_t)bucket_begin, (uint32_t)bucket_end, sink, user, stats);
        case 3: return build_impl<3>(ctx, rd, n_short_reads, k, min_count, need_mercy, (uint32_t)bucket_begin, (uint32_t)bucket_end, sink, user, stats);
        case 4: return build_impl<4>(ctx, rd, n_short_reads, k, min_count, need_mercy, (uint32_t)bucket_begin, (uint32_t)bucket_end, sink, user, stats);
        case 5: return build_impl<5>(ctx, rd, n_short_reads, k, min_count, need_mercy, (uint32_t)bucket_begin, (uint32_t)bucket_end, sink, user, stats);
        case 6: return build_impl<6>(ctx, rd, n_short_reads, k, min_count, need_mercy, (uint32_t)bucket_begin, (uint32_t)bucket_end, sink, user, stats);
        case 7: return build_impl<7>(ctx, rd, n_short_reads, k, min_count, need_mercy, (uint32_t)bucket_begin, (uint32_t)bucket_end, sink, user, stats);
        case 8: return build_impl<8>(ctx, rd, n_short_reads, k, min_count, need_mercy, (uint32_t)bucket_begin, (uint32_t)bucket_end, sink, user, stats);
        default: return build_impl<9>(ctx, rd, n_short_reads, k, min_count, need_mercy, (uint32_t)bucket_begin, (uint32_t)bucket_end, sink, user, stats);
        }
    } catch (const HipError &e) { return e.code; }
}

int mgta_sdbg_last_counting(mgta_ctx *ctx, int64_t *hist) {
    if (!ctx || !hist) { set_error("mgta_sdbg_last_counting: null argument"); return MGTA_EINVAL; }
    if (ctx->edge_counting.size() != 65536) { set_error("no stage-1 run (min_count > 1) on this context yet"); return MGTA_EINVAL; }
    std::copy(ctx->edge_counting.begin(), ctx->edge_counting.end(), hist);
    return MGTA_OK;
}

int mgta_sort_plan(uint64_t n_items, int words_per_key, uint32_t bucket_begin, uint32_t bucket_end, int *n_passes, int *skip_bits) {
    if (!n_passes || !skip_bits || words_per_key < 2 || bucket_end > (uint32_t)MGTA_NUM_BUCKETS || bucket_begin >= bucket_end) {
        set_error("mgta_sort_plan: bad argument");
        return MGTA_EINVAL;
    }
    const TopPlan tp = choose_top_plan(nullptr, n_items, std::min(4, (32 * words_per_key - 8) / 8), (double)(bucket_end - bucket_begin) / MGTA_NUM_BUCKETS,
                                       bucket_begin, bucket_end);
    *n_passes = tp.P;
    *skip_bits = tp.skip;
    return MGTA_OK;
}

int mgta_sdbg_export_records_device(mgta_ctx *ctx, void *d_dst, uint64_t capacity_bytes, uint64_t *n_records) {
    if (!ctx || !n_records) { set_error("mgta_sdbg_export_records_device: null argument"); return MGTA_EINVAL; }
    *n_records = ctx->last_n_rec;
    if (ctx->last_k == 0 || (!ctx->last_rec && ctx->last_n_rec)) { set_error("no device-resident build output"); return MGTA_EINVAL; }
    if (!d_dst) return MGTA_OK;                                    // size query
    if (capacity_bytes < ctx->last_n_rec * 2) { set_error("destination too small"); return MGTA_EINVAL; }
    try {
        MGTA_HIP_CHECK(hipSetDevice(ctx->device));
        if (ctx->last_n_rec) MGTA_HIP_CHECK(hipMemcpyAsync(d_dst, ctx->last_rec, ctx->last_n_rec * 2, hipMemcpyDeviceToDevice, ctx->stream));
        MGTA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return MGTA_OK;
    } catch (const HipError &e) { return e.code; }
}

int mgta_sdbg_build(mgta_ctx *ctx, const uint32_t *packed, uint64_t n_words, const uint64_t *start_idx, uint64_t n_reads,
                    uint64_t n_short_reads, int k, int min_count, int need_mercy, mgta_edge_sink sink, void *user,
                    mgta_build_stats *stats) {
    mgta_reads *rd = nullptr;
    int rc = mgta_reads_upload(ctx, packed, n_words, start_idx, n_reads, &rd);
    if (rc != MGTA_OK) return rc;
    rc = mgta_sdbg_build_resident(ctx, rd, n_short_reads, k, min_count, need_mercy, 0, MGTA_NUM_BUCKETS, sink, user, stats);
    mgta_reads_free(rd);
    return rc;
}

}  // extern "C"
