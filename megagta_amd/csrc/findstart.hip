// findstart.hip — seed finder on gfx950 (MI355X): which windows of k nucleotides of the reads translate into a k/3-residue
// word of the gene's reference alignment.   SURVEY.md §8(f) row 2.
//
// Replaces the scan of `megagta findstart` (fast_kmer_filter.cpp:108-176: both strands of every read, ProcessSequenceMulti
// :193-215: three frames, seq::AASequence::translate, HashSetST<ProtKmer>::find) behind mgta_findstart().  The reference set
// itself (a few thousand words, prot_kmer_generator.h) is built on the host and handed over as packed codes.
//
//   one wave per read, one lane per window start: the k bases are funnel-shifted out of the 2-bit read array, both strands are
//   translated codon by codon through a 64-entry table in LDS (frames fall out of the window start modulo 3), residues are packed
//   5 bits each; a 32768-bit filter in LDS on the first three residues drops most windows before the probe of the (L2-resident)
//   open-addressing table; hits are appended through an atomic cursor.
// Integer work, ~0.25 byte of HBM traffic per window: bound by instruction issue, not memory.
#include <vector>

#include "common.hpp"
#include "device_utils.hpp"

namespace mgta {

constexpr int kFsWords = 5;                 // 72 nucleotides (24 residues, Kmer::MAX_PROT_KMER_SIZE) = 144 bits
constexpr int kFsBlock = 256;
constexpr int kFsReadsPerBlock = 64;

// standard genetic code, codon = 16 b0 + 4 b1 + b2 (A0 C1 G2 T3) -> residue code of prot_kmer.h:31-43 (ARNDCQEGHILKMFPSTWYV = 0..19, '*' = 20)
__constant__ uint8_t kCodonCode[64] = {11, 2, 11, 2, 16, 16, 16, 16, 1, 15, 1, 15, 9, 9, 12, 9, 5, 8, 5, 8, 14, 14, 14, 14, 1, 1, 1, 1, 10, 10, 10, 10,
                                       6, 3, 6, 3, 0, 0, 0, 0, 7, 7, 7, 7, 19, 19, 19, 19, 20, 18, 20, 18, 15, 15, 15, 15, 20, 4, 17, 4, 10, 13, 10, 13};

struct FsArgs {
    const uint32_t *packed;
    uint64_t n_words;
    const uint64_t *start;
    uint64_t n_reads;
    int k, kaa, reversed;
    const unsigned long long *tab;       // [tab_mask + 1][2]; first word ~0 = empty
    const int32_t *tab_ref;              // index of the reference word
    uint32_t tab_mask;
    const uint32_t *filter;              // 1024 words: bit (c0 << 10 | c1 << 5 | c2) of the first three residues
    mgta_seed_hit *hits;
    unsigned long long *n_hits;
    uint64_t cap;
};

__device__ __forceinline__ uint32_t fs_rev_chars(uint32_t x) {   // reverse the 16 characters of a word
    x = __brev(x);
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}
__device__ __forceinline__ uint64_t fs_mix(uint64_t a, uint64_t b) {
    uint64_t x = a ^ (b * 0x9E3779B97F4A7C15ull);
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// residues of a left-aligned k-character string -> (w0: first <= 12 residues, w1: the rest), 5 bits each, first residue highest
__device__ __forceinline__ void fs_translate(const uint32_t (&s)[kFsWords], int kaa, const uint8_t *lut, uint64_t &w0, uint64_t &w1) {
    w0 = 0; w1 = 0;
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        if (i < kaa) {                                                   // wave-uniform
            constexpr int dummy = 0; (void)dummy;
            const int bit = 6 * i, j = bit >> 5, off = bit & 31;         // compile-time after unrolling
            const uint64_t two = ((uint64_t)s[j] << 32) | (uint64_t)(j + 1 < kFsWords ? s[j + 1] : 0u);
            const uint32_t codon = (uint32_t)(two >> (58 - off)) & 63u;
            const uint64_t aa = lut[codon];
            if (i < 12) w0 = (w0 << 5) | aa; else w1 = (w1 << 5) | aa;
        }
    }
}

__global__ __launch_bounds__(kFsBlock) void findstart_kernel(FsArgs a) {
    __shared__ uint8_t s_lut[64];
    __shared__ uint32_t s_filter[1024];
    for (int i = threadIdx.x; i < 64; i += kFsBlock) s_lut[i] = kCodonCode[i];
    for (int i = threadIdx.x; i < 1024; i += kFsBlock) s_filter[i] = a.filter[i];
    __syncthreads();
    const int k = a.k, kaa = a.kaa;
    const int lane = lane_id(), wv = wave_id();
    const int pad_bits = 32 * kFsWords - 2 * k;
    const uint64_t r0 = (uint64_t)blockIdx.x * kFsReadsPerBlock;
    const uint64_t r1 = r0 + kFsReadsPerBlock < a.n_reads ? r0 + kFsReadsPerBlock : a.n_reads;
    for (uint64_t r = r0 + wv; r < r1; r += kFsBlock / 64) {
        const uint64_t s0 = a.start[r];
        const int len = (int)(a.start[r + 1] - s0);
        if (len < k) continue;                                           // fast_kmer_filter.cpp:120
        const int npos = len - k + 1;
        for (int c0 = 0; c0 < npos; c0 += 64) {
            const int p = c0 + lane;
            const bool active = p < npos;                                // (no early exit: the hits are appended wave by wave)
            const uint64_t q = s0 + (uint64_t)(active ? p : 0), wi = q >> 4;
            const int sh = (int)(q & 15) * 2;
            uint32_t raw[kFsWords + 1], e[kFsWords], v[kFsWords];
#pragma unroll
            for (int j = 0; j <= kFsWords; ++j) raw[j] = (wi + j < a.n_words) ? a.packed[wi + j] : 0u;
#pragma unroll
            for (int j = 0; j < kFsWords; ++j) e[j] = sh ? ((raw[j] << sh) | (raw[j + 1] >> (32 - sh))) : raw[j];
            // keep the first k characters
#pragma unroll
            for (int j = 0; j < kFsWords; ++j) {
                const int lo = j * 16;
                if (k <= lo) e[j] = 0;
                else if (k < lo + 16) e[j] &= ~0u << (32 - 2 * (k - lo));
            }
            // the window read backwards (characters reversed, left-aligned again)
#pragma unroll
            for (int j = 0; j < kFsWords; ++j) v[j] = fs_rev_chars(e[kFsWords - 1 - j]);
            {   // shift left by pad_bits (0 <= pad_bits < 32 * kFsWords)
                const int ws = pad_bits >> 5, bs = pad_bits & 31;
                uint32_t t[kFsWords];
#pragma unroll
                for (int j = 0; j < kFsWords; ++j) {
                    uint32_t lo_w = 0, hi_w = 0;
#pragma unroll
                    for (int x = 0; x < kFsWords; ++x) {
                        if (x == j + ws) hi_w = v[x];
                        if (x == j + ws + 1) lo_w = v[x];
                    }
                    t[j] = bs ? ((hi_w << bs) | (lo_w >> (32 - bs))) : hi_w;
                }
#pragma unroll
                for (int j = 0; j < kFsWords; ++j) v[j] = t[j];
            }
            // strand 0 = the read as sequenced, strand 1 = its reverse complement (fast_kmer_filter.cpp:124-133).
            // forward storage: window e at p is strand 0 at p, comp(rev(e)) is strand 1 at len-k-p;
            // reversed storage (what buildgraph uploads): rev(e) is strand 0 at len-k-p, comp(e) is strand 1 at p.
            uint32_t sA[kFsWords], sB[kFsWords];
#pragma unroll
            for (int j = 0; j < kFsWords; ++j) {
                const int lo = j * 16;
                uint32_t m = k <= lo ? 0u : (k < lo + 16 ? (~0u << (32 - 2 * (k - lo))) : ~0u);
                sA[j] = a.reversed ? v[j] : e[j];
                sB[j] = (a.reversed ? ~e[j] : ~v[j]) & m;
            }
            const uint32_t posA = a.reversed ? (uint32_t)(len - k - p) : (uint32_t)p;
            const uint32_t posB = a.reversed ? (uint32_t)p : (uint32_t)(len - k - p);
#pragma unroll
            for (int strand = 0; strand < 2; ++strand) {
                uint64_t w0, w1;
                if (strand == 0) fs_translate(sA, kaa, s_lut, w0, w1);
                else fs_translate(sB, kaa, s_lut, w0, w1);
                // first three residues: the top 15 bits of the first word's used part
                const int n0 = kaa < 12 ? kaa : 12;
                const uint32_t pre = (uint32_t)(w0 >> (5 * (n0 - 3))) & 0x7FFFu;
                int32_t ref = -1;
                if (active && ((s_filter[pre >> 5] >> (pre & 31)) & 1u)) {
                    uint32_t h = (uint32_t)fs_mix(w0, w1) & a.tab_mask;
                    for (;;) {
                        const unsigned long long k0 = a.tab[2 * (uint64_t)h];
                        if (k0 == ~0ull) break;
                        if (k0 == w0 && a.tab[2 * (uint64_t)h + 1] == w1) { ref = a.tab_ref[h]; break; }
                        h = (h + 1) & a.tab_mask;
                    }
                }
                const uint64_t found = __ballot(ref >= 0);                // one cursor update per wave
                if (found) {
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(a.n_hits, (unsigned long long)__popcll(found));
                    base = __shfl(base, 0, 64);
                    if (ref >= 0) {
                        const unsigned long long slot = base + (unsigned long long)__popcll(found & lanemask_lt());
                        if (slot < a.cap) {
                            mgta_seed_hit hit;
                            hit.read = r;
                            hit.pos_strand = ((strand == 0 ? posA : posB) << 1) | (uint32_t)strand;
                            hit.ref = ref;
                            a.hits[slot] = hit;
                        }
                    }
                }
            }
        }
    }
}

}  // namespace mgta

using namespace mgta;

extern "C" int mgta_findstart(mgta_ctx *ctx, const mgta_reads *reads, int reads_reversed, int k, const uint64_t *ref_words, int64_t n_ref,
                              mgta_seed_hit *hits, int64_t cap, int64_t *n_hits, double *ms_kernel) {
    if (!ctx || !reads || !n_hits || n_ref < 0 || cap < 0 || (cap > 0 && !hits) || (n_ref > 0 && !ref_words)) {
        set_error("mgta_findstart: bad argument");
        return MGTA_EINVAL;
    }
    if (k < 9 || k % 3 != 0 || k / 3 > 24) {                            // Kmer::MAX_PROT_KMER_SIZE, prot_kmer_generator.h:33-35
        set_error("mgta_findstart: k = %d (a multiple of 3 in [9, 72] is required: k/3 residues, at most 24)", k);
        return MGTA_EINVAL;
    }
    try {
        MGTA_HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const int kaa = k / 3, n0 = kaa < 12 ? kaa : 12;
        // open-addressing table + three-residue filter, built on the host (a few thousand words)
        uint64_t tcap = 64;
        while (tcap < (uint64_t)n_ref * 2 + 2) tcap <<= 1;
        std::vector<unsigned long long> tab(tcap * 2, ~0ull);
        std::vector<int32_t> tab_ref(tcap, -1);
        std::vector<uint32_t> filter(1024, 0u);
        auto mix = [](uint64_t a, uint64_t b) {
            uint64_t x = a ^ (b * 0x9E3779B97F4A7C15ull);
            x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
            return x;
        };
        for (int64_t i = 0; i < n_ref; ++i) {
            const uint64_t w0 = ref_words[2 * i], w1 = ref_words[2 * i + 1];
            uint64_t h = mix(w0, w1) & (tcap - 1);
            bool dup = false;
            while (tab[2 * h] != ~0ull) {
                if (tab[2 * h] == w0 && tab[2 * h + 1] == w1) { dup = true; break; }   // first insertion wins (insert_unique)
                h = (h + 1) & (tcap - 1);
            }
            if (dup) continue;
            tab[2 * h] = w0; tab[2 * h + 1] = w1; tab_ref[h] = (int32_t)i;
            const uint32_t pre = (uint32_t)(w0 >> (5 * (n0 - 3))) & 0x7FFFu;
            filter[pre >> 5] |= 1u << (pre & 31);
        }
        DevBuf d_tab, d_ref, d_filter, d_hits, d_cnt;
        d_tab.alloc(tab.size() * 8, &ctx->live_bytes, &ctx->peak_bytes);
        d_ref.alloc(tab_ref.size() * 4, &ctx->live_bytes, &ctx->peak_bytes);
        d_filter.alloc(4096, &ctx->live_bytes, &ctx->peak_bytes);
        d_hits.alloc(std::max<uint64_t>(1, (uint64_t)cap) * sizeof(mgta_seed_hit), &ctx->live_bytes, &ctx->peak_bytes);
        d_cnt.alloc(8, &ctx->live_bytes, &ctx->peak_bytes);
        MGTA_HIP_CHECK(hipMemcpyAsync(d_tab.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, st));
        MGTA_HIP_CHECK(hipMemcpyAsync(d_ref.p, tab_ref.data(), tab_ref.size() * 4, hipMemcpyHostToDevice, st));
        MGTA_HIP_CHECK(hipMemcpyAsync(d_filter.p, filter.data(), 4096, hipMemcpyHostToDevice, st));
        MGTA_HIP_CHECK(hipMemsetAsync(d_cnt.p, 0, 8, st));
        FsArgs a;
        a.packed = reads->d_packed; a.n_words = reads->n_words; a.start = reads->d_start; a.n_reads = reads->n_reads;
        a.k = k; a.kaa = kaa; a.reversed = reads_reversed ? 1 : 0;
        a.tab = d_tab.as<unsigned long long>(); a.tab_ref = d_ref.as<int32_t>(); a.tab_mask = (uint32_t)(tcap - 1);
        a.filter = d_filter.as<uint32_t>();
        a.hits = d_hits.as<mgta_seed_hit>(); a.n_hits = d_cnt.as<unsigned long long>(); a.cap = (uint64_t)cap;
        hipEvent_t e0, e1;
        MGTA_HIP_CHECK(hipEventCreate(&e0));
        MGTA_HIP_CHECK(hipEventCreate(&e1));
        MGTA_HIP_CHECK(hipEventRecord(e0, st));
        const uint64_t n_blocks = (reads->n_reads + kFsReadsPerBlock - 1) / kFsReadsPerBlock;
        if (n_blocks) hipLaunchKernelGGL(findstart_kernel, dim3((unsigned)n_blocks), dim3(kFsBlock), 0, st, a);
        MGTA_HIP_CHECK(hipEventRecord(e1, st));
        unsigned long long n = 0;
        MGTA_HIP_CHECK(hipMemcpyAsync(&n, d_cnt.p, 8, hipMemcpyDeviceToHost, st));
        MGTA_HIP_CHECK(hipStreamSynchronize(st));
        float ms = 0;
        MGTA_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (ms_kernel) *ms_kernel = ms;
        *n_hits = (int64_t)n;
        const uint64_t got = std::min<uint64_t>(n, (uint64_t)cap);
        if (got) MGTA_HIP_CHECK(hipMemcpy(hits, d_hits.p, got * sizeof(mgta_seed_hit), hipMemcpyDeviceToHost));
        return MGTA_OK;
    } catch (const HipError &e) { return e.code; }
}
