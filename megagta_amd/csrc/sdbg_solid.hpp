// sdbg_solid.hpp — stage 1 of read -> SdBG for `-m >= 2`: count (k+1)-mers, mark the solid ones per read
// position, collect mercy-edge candidates and add mercy edges.  Included by sdbg_build.hip only.
//
// Replaces s1_lv0_calc_bucket_size / s1_lv1_fill_offset / s1_extract_subtstr_ / s1_lv2_output_ / s1_post_proc
// (cx1_read2sdbg_s1.cpp:177-229,408-513,515-596,671-830,905-951) and s2_read_mercy_prepare
// (cx1_read2sdbg_s2.cpp:106-250).  Same idea as the reference: sort every canonical (k-1)-mer S of every read together
// with its two neighbours on each side (a b S c d): the multiplicity of the (k+1)-mer bSc says whether the edge is solid,
// those of abS / Scd whether it has a solid predecessor / successor.
//
// Device design: one sort record = [key: (k-1)-mer | head<<3|tail] [payload: (abs offset<<1|strand)<<6 | prev<<3|next],
// sorted with the same radix machinery as stage 2 (payload position bits ignored, prev/next bits included so that equal
// (S,head,tail,prev,next) form runs); run descriptors -> per (k-1)-mer group masks -> per run verdict -> per item:
// atomic OR into the is_solid bit-vector + mercy candidates appended through a wave-aggregated cursor.
#pragma once

namespace mgta {

template <int W>
__device__ __forceinline__ void shl_bits_any(uint32_t (&x)[W], int s) {   // s >= 0, any size
    while (s >= 32) { shl_bits<W>(x, 32); s -= 32; }
    shl_bits<W>(x, s);
}

struct S1Args {
    const uint32_t *packed;
    uint64_t n_words;
    const uint64_t *start;
    uint64_t n_reads;
    int k;
    uint32_t b_lo, b_hi;
    uint32_t *block_count;
    const uint64_t *block_base;
    void *out;                  // Key<W1 + 2>*
};

// one wave per read chunk, one lane per (k-1)-mer offset o in [0, len-k+1]
template <int W1, bool WRITE>
__global__ __launch_bounds__(kScanBlock) void s1_scan_kernel(S1Args a) {
    constexpr int WT = W1 + 2;
    __shared__ uint32_t s_cursor;
    __shared__ uint32_t s_wave_cnt[kScanBlock / 64];
    const int k = a.k, km1 = a.k - 1;
    const int lane = lane_id(), wv = wave_id();
    if (threadIdx.x == 0) s_cursor = 0;
    __syncthreads();
    uint64_t r0 = (uint64_t)blockIdx.x * kReadsPerBlock;
    uint64_t r1 = r0 + kReadsPerBlock < a.n_reads ? r0 + kReadsPerBlock : a.n_reads;
    Key<WT> *out = reinterpret_cast<Key<WT> *>(a.out);
    uint64_t base = WRITE ? a.block_base[blockIdx.x] : 0;
    uint32_t my_count = 0;
    const int pad_bits = 2 * (16 * W1 - km1);
    auto base_at = [&](uint64_t pos) -> int { return (int)((a.packed[pos >> 4] >> (30 - 2 * (pos & 15))) & 3); };

    for (uint64_t r = r0 + wv; r < r1; r += kScanBlock / 64) {
        uint64_t s0 = a.start[r];
        int len = (int)(a.start[r + 1] - s0);
        if (len < k + 1) continue;                                     // s1.cpp:187-189
        int n_off = len - k + 2;
        for (int c0 = 0; c0 < n_off; c0 += 64) {
            int o = c0 + lane;
            Key<WT> items[2];
            int cnt = 0;
            if (o < n_off) {
                uint64_t q = s0 + (uint64_t)o;
                uint64_t wi = q >> 4;
                int sh = (int)(q & 15) * 2;
                uint32_t raw[W1 + 1], f[W1], rc[W1];
#pragma unroll
                for (int j = 0; j <= W1; ++j) raw[j] = (wi + j < a.n_words) ? a.packed[wi + j] : 0u;
#pragma unroll
                for (int j = 0; j < W1; ++j) f[j] = sh ? ((raw[j] << sh) | (raw[j + 1] >> (32 - sh))) : raw[j];
                keep_chars<W1>(f, km1);
#pragma unroll
                for (int j = 0; j < W1; ++j) rc[j] = rev_chars(~f[W1 - 1 - j]);
                shl_bits_any<W1>(rc, pad_bits);
                // neighbours: (k+1)-mer = a b S c d  ->  prev=a head=b tail=c next=d   (s1.cpp:534-563)
                int head = o > 0 ? base_at(q - 1) : kDollar, prev = o > 1 ? base_at(q - 2) : kDollar;
                int tail = o + km1 < len ? base_at(q + km1) : kDollar, next = o + k < len ? base_at(q + k) : kDollar;
                bool emit0, emit1;
                if (o == 0 || o == n_off - 1) { emit0 = emit1 = true; }            // first / last (k-1)-mer: both strands (s1.cpp:470-472,503-506)
                else {
                    int cmp = 0;
#pragma unroll
                    for (int j = 0; j < W1; ++j) if (cmp == 0 && f[j] != rc[j]) cmp = f[j] < rc[j] ? -1 : 1;
                    if (cmp < 0) { emit0 = true; emit1 = false; }
                    else if (cmp > 0) { emit0 = false; emit1 = true; }
                    else { emit0 = head <= 3 - tail; emit1 = !emit0; }             // palindrome rule (s1.cpp:488-497): prev = base(o-1), next = base(o+k-1)
                }
                auto push = [&](const uint32_t (&kw)[W1], int hd, int tl, int pv, int nx, int strand) {
                    uint32_t b = kw[0] >> 16;
                    if (b < a.b_lo || b >= a.b_hi) return;
                    Key<WT> it;
#pragma unroll
                    for (int j = 0; j < W1; ++j) it.w[j] = kw[j];
                    it.w[W1 - 1] |= (uint32_t)((hd << 3) | tl);
                    uint64_t info = ((((q << 1) | (uint64_t)strand)) << 6) | (uint64_t)((pv << 3) | nx);
                    it.w[W1] = (uint32_t)(info >> 32);
                    it.w[W1 + 1] = (uint32_t)info;
                    items[cnt++] = it;
                };
                auto comp = [](int c) { return c == kDollar ? kDollar : 3 - c; };
                if (emit0) push(f, head, tail, prev, next, 0);
                if (emit1) push(rc, comp(tail), comp(head), comp(next), comp(prev), 1);   // s1.cpp:575-582
            }
            if (!WRITE) my_count += (uint32_t)cnt;
            else {
                uint32_t inc = wave_incl_scan((uint32_t)cnt);
                uint32_t tot = __shfl(inc, 63, 64);
                uint32_t wbase = 0;
                if (lane == 0 && tot) wbase = atomicAdd(&s_cursor, tot);
                wbase = __shfl(wbase, 0, 64);
                uint64_t dst = base + wbase + (inc - (uint32_t)cnt);
                for (int i = 0; i < cnt; ++i) out[dst + i] = items[i];
            }
        }
    }
    if (!WRITE) {
        my_count = wave_sum(my_count);
        if (lane == 0) s_wave_cnt[wv] = my_count;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int w = 0; w < kScanBlock / 64; ++w) t += s_wave_cnt[w];
            a.block_count[blockIdx.x] = t;
        }
    }
}

// ---- runs of equal (S, head, tail, prev, next) ---------------------------------------------------
template <int W1>
__device__ __forceinline__ bool s1_same_run(const Key<W1 + 2> &x, const Key<W1 + 2> &y) {
    bool eq = true;
#pragma unroll
    for (int j = 0; j < W1; ++j) eq = eq && (x.w[j] == y.w[j]);
    return eq && ((x.w[W1 + 1] & 63u) == (y.w[W1 + 1] & 63u));
}
template <int W1>
__device__ __forceinline__ bool s1_same_group(const Key<W1 + 2> &x, const Key<W1 + 2> &y, int k) {   // same (k-1)-mer: IsDiffKMinusOneMer, s1.cpp:57-78
    int full = (k - 1) >> 4, rem = (k - 1) & 15;
    bool eq = true;
#pragma unroll
    for (int j = 0; j < W1; ++j) {
        if (j < full) eq = eq && (x.w[j] == y.w[j]);
        else if (j == full && rem > 0) eq = eq && ((x.w[j] >> (16 - rem) * 2) == (y.w[j] >> (16 - rem) * 2));
    }
    return eq;
}

template <int W1>
__global__ __launch_bounds__(kEmitThreads) void s1_mark_kernel(const Key<W1 + 2> *keys, uint64_t n, uint32_t *tile_heads) {
    __shared__ uint32_t s_cnt[kEmitThreads / 64];
    uint64_t base = (uint64_t)blockIdx.x * kEmitTile;
    uint32_t c = 0;
    for (int it = 0; it < kEmitPerThread; ++it) {
        uint64_t idx = base + (uint64_t)it * kEmitThreads + threadIdx.x;
        bool head = false;
        if (idx < n) head = idx == 0 || !s1_same_run<W1>(keys[idx], keys[idx - 1]);
        c += (uint32_t)__popcll(__ballot(head));
    }
    if (lane_id() == 0) s_cnt[wave_id()] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kEmitThreads / 64; ++w) t += s_cnt[w];
        tile_heads[blockIdx.x] = t;
    }
}

// run descriptor: start index + info = head | tail<<3 | prev<<6 | next<<9 | group_head<<12
template <int W1>
__global__ __launch_bounds__(kEmitThreads) void s1_compact_kernel(const Key<W1 + 2> *keys, uint64_t n, int k, const uint64_t *tile_base,
                                                                   uint64_t *run_start, uint16_t *run_info) {
    __shared__ uint32_t s_cnt[kEmitPerThread * (kEmitThreads / 64)];
    __shared__ uint32_t s_scr[kEmitThreads / 64 + 1];
    uint64_t base = (uint64_t)blockIdx.x * kEmitTile;
    const int lane = lane_id(), wv = wave_id();
    uint32_t headbits = 0;
    uint32_t rank_in_wave[kEmitPerThread];
    uint16_t info[kEmitPerThread];
#pragma unroll
    for (int it = 0; it < kEmitPerThread; ++it) {
        uint64_t idx = base + (uint64_t)it * kEmitThreads + threadIdx.x;
        bool head = false;
        info[it] = 0;
        if (idx < n) {
            Key<W1 + 2> cur = keys[idx];
            bool ghead = true;
            if (idx == 0) head = true;
            else {
                Key<W1 + 2> prv = keys[idx - 1];
                head = !s1_same_run<W1>(cur, prv);
                ghead = !s1_same_group<W1>(cur, prv, k);
            }
            uint32_t ht = cur.w[W1 - 1] & 63u, pn = cur.w[W1 + 1] & 63u;
            info[it] = (uint16_t)((ht >> 3) | ((ht & 7u) << 3) | ((pn >> 3) << 6) | ((pn & 7u) << 9) | ((uint32_t)ghead << 12));
        }
        uint64_t bal = __ballot(head);
        rank_in_wave[it] = (uint32_t)__popcll(bal & lanemask_lt());
        headbits |= (uint32_t)head << it;
        if (lane == 0) s_cnt[it * (kEmitThreads / 64) + wv] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    uint32_t v = threadIdx.x < kEmitPerThread * (kEmitThreads / 64) ? s_cnt[threadIdx.x] : 0;
    uint32_t ex = block_excl_scan<kEmitThreads>(v, s_scr, nullptr);
    __syncthreads();
    if (threadIdx.x < kEmitPerThread * (kEmitThreads / 64)) s_cnt[threadIdx.x] = ex;
    __syncthreads();
    uint64_t tb = tile_base[blockIdx.x];
#pragma unroll
    for (int it = 0; it < kEmitPerThread; ++it) {
        if ((headbits >> it) & 1u) {
            uint64_t idx = base + (uint64_t)it * kEmitThreads + threadIdx.x;
            uint64_t s = tb + s_cnt[it * (kEmitThreads / 64) + wv] + rank_in_wave[it];
            run_start[s] = idx;
            run_info[s] = info[it];
        }
    }
}

// per (k-1)-mer group (thread of its first run): has_in | has_out<<4 | l_has_out<<8 | r_has_in<<12   (s1.cpp:716-748)
__global__ __launch_bounds__(256) void s1_group_kernel(const uint64_t *run_start, const uint16_t *run_info, uint64_t m, uint64_t n_items,
                                                       int threshold, uint16_t *group_mask) {
    uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= m || !((run_info[s] >> 12) & 1)) return;
    // counts over the group: (prev,head), (tail,next), (head,tail); '$' (4) never contributes to a mask
    uint32_t cph[16], ctn[16], cht[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) cph[i] = ctn[i] = cht[i] = 0;
    uint64_t x = s;
    do {
        uint32_t inf = run_info[x];
        int hd = inf & 7, tl = (inf >> 3) & 7, pv = (inf >> 6) & 7, nx = (inf >> 9) & 7;
        uint64_t end = x + 1 < m ? run_start[x + 1] : n_items;
        uint32_t c = (uint32_t)(end - run_start[x]);
        // register arrays indexed at run time would go to scratch: this kernel is off the hot path, clarity wins
        if (pv < 4 && hd < 4) cph[pv * 4 + hd] += c;
        if (tl < 4 && nx < 4) ctn[tl * 4 + nx] += c;
        if (hd < 4 && tl < 4) cht[hd * 4 + tl] += c;
        ++x;
    } while (x < m && !((run_info[x] >> 12) & 1));
    uint32_t T = (uint32_t)threshold;
    int has_in = 0, has_out = 0, l_has_out = 0, r_has_in = 0;
    for (int j = 0; j < 4; ++j)
        for (int q = 0; q < 4; ++q) {
            if (cph[q * 4 + j] >= T) has_in |= 1 << j;
            if (ctn[j * 4 + q] >= T) has_out |= 1 << j;
            if (cht[j * 4 + q] >= T) { l_has_out |= 1 << j; r_has_in |= 1 << q; }
        }
    group_mask[s] = (uint16_t)(has_in | (has_out << 4) | (l_has_out << 8) | (r_has_in << 12));
}

// read that holds absolute base index `abs` (SequencePackage::get_id, sequence_package.h:164-188): like the reference's pos_to_id_
// table, a coarse table (one entry per 1024 bases = the last read starting at or before that base) narrows the search to a few reads
constexpr int kPosStepLog = 10;
__global__ __launch_bounds__(256) void pos_to_id_kernel(const uint64_t *start_idx, uint64_t n_reads, uint64_t n_entries, uint32_t *table) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_entries) return;
    const uint64_t abs = i << kPosStepLog;
    uint64_t lo = 0, hi = n_reads;
    while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (start_idx[mid] <= abs) lo = mid; else hi = mid; }
    table[i] = (uint32_t)lo;
}
__device__ __forceinline__ uint64_t read_of_base(const uint64_t *start_idx, uint64_t n_reads, const uint32_t *table, uint64_t abs) {
    uint64_t lo = table[abs >> kPosStepLog], hi = (uint64_t)table[(abs >> kPosStepLog) + 1] + 1;   // the table has one entry past the last base
    if (hi > n_reads) hi = n_reads;
    while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (start_idx[mid] <= abs) lo = mid; else hi = mid; }
    return lo;
}

// per run: apply the verdict to every item of the run (s1.cpp:750-828)
template <int W1>
__global__ __launch_bounds__(256) void s1_apply_kernel(const Key<W1 + 2> *keys, const uint64_t *run_start, const uint16_t *run_info,
                                                       const uint16_t *group_mask, uint64_t m, uint64_t n_items, int threshold, int k,
                                                       const uint64_t *start_idx, uint64_t n_reads, uint64_t n_short, int num_k1_per_read,
                                                       unsigned long long *is_solid, unsigned long long *edge_count /* [65536] */,
                                                       Key<2> *mercy, unsigned long long *mercy_count, uint64_t mercy_cap,
                                                       int need_mercy, const uint32_t *pos_to_id, int cand_shift) {
    // multiplicities are few and small: counted in LDS, flushed once per workgroup (a global atomic per (k+1)-mer would
    // hammer a handful of addresses a billion times)
    __shared__ unsigned int s_count[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) s_count[i] = 0;
    __syncthreads();
    for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < m; s += (uint64_t)gridDim.x * 256) {
    const uint32_t inf = run_info[s];
    const int hd = inf & 7, tl = (inf >> 3) & 7;
    // group head + masks
    uint64_t gs = s;
    while (!((run_info[gs] >> 12) & 1)) --gs;
    const uint32_t gm = group_mask[gs];
    const int has_in = gm & 15, has_out = (gm >> 4) & 15, l_has_out = (gm >> 8) & 15, r_has_in = (gm >> 12) & 15;
    // count_head_tail of this (head,tail): the adjacent runs of the group that share it
    uint64_t a = s, b = s + 1;
    while (!((run_info[a] >> 12) & 1) && (run_info[a - 1] & 63u) == (inf & 63u)) --a;
    while (b < m && !((run_info[b] >> 12) & 1) && (run_info[b] & 63u) == (inf & 63u)) ++b;
    const uint64_t cnt_ht = (b < m ? run_start[b] : n_items) - run_start[a];
    const bool real = hd != kDollar && tl != kDollar;
    if (real && a == s) {                                                                  // one count per (k+1)-mer (s1.cpp:756-758)
        if (cnt_ht < 1024) atomicAdd(&s_count[cnt_ht], 1u);
        else atomicAdd(&edge_count[cnt_ht > 65535 ? 65535 : cnt_ht], 1ull);
    }
    const bool solid = real && cnt_ht >= (uint64_t)threshold;
    const uint64_t end = s + 1 < m ? run_start[s + 1] : n_items;
    auto cand = [&](uint64_t v) {
        if (!need_mercy) return;
        // the lanes that are here together share one cursor update
        const uint64_t act = __ballot(1);
        const int leader = __ffsll((long long)act) - 1;
        unsigned long long q = 0;
        if (lane_id() == leader) q = atomicAdd(mercy_count, (unsigned long long)__popcll(act));
        q = __shfl(q, leader, 64) + (unsigned long long)__popcll(act & lanemask_lt());
        v <<= cand_shift;                                              // left-aligned: the sort's leading bytes are not all zero
        if (q < mercy_cap) { mercy[q].w[0] = (uint32_t)(v >> 32); mercy[q].w[1] = (uint32_t)v; }
    };
    for (uint64_t i = run_start[s]; i < end; ++i) {
        const Key<W1 + 2> &it = keys[i];
        uint64_t info = ((uint64_t)it.w[W1] << 32) | it.w[W1 + 1];
        uint64_t fo = info >> 6;
        int strand = (int)(fo & 1);
        uint64_t abs = fo >> 1;
        const uint64_t read_id = read_of_base(start_idx, n_reads, pos_to_id, abs);
        if (read_id >= n_short) continue;                                                      // assist sequences are always solid
        const int64_t offset = (int64_t)(abs - start_idx[read_id]) - 1;
        const int64_t l_off = strand == 0 ? offset : offset + 1, r_off = strand == 0 ? offset + 1 : offset;
        const uint64_t st = start_idx[read_id];
        if (solid) {
            uint64_t bit = (uint64_t)num_k1_per_read * read_id + (uint64_t)offset;
            atomicOr(&is_solid[bit >> 6], 1ull << (bit & 63));
            if (!((has_in >> hd) & 1)) cand(((st + l_off) << 2) | (uint64_t)(1 + strand));
            if (!((has_out >> tl) & 1)) cand(((st + r_off) << 2) | (uint64_t)(2 - strand));
        } else {
            // not solid: still tell whether the k-mers left / right of it touch solid edges (s1.cpp:786-826).
            // hd / tl == '$' select no mask bit (masks have 4 bits), exactly like `1 << 4` in the reference.
            if (hd < 4) {
                if ((l_has_out >> hd) & 1) cand(((st + l_off) << 2) | (uint64_t)(((has_in >> hd) & 1) ? 0 : 1 + strand));
                else if ((has_in >> hd) & 1) cand(((st + l_off) << 2) | (uint64_t)(2 - strand));
            }
            if (tl < 4) {
                if ((r_has_in >> tl) & 1) cand(((st + r_off) << 2) | (uint64_t)(((has_out >> tl) & 1) ? 0 : 2 - strand));
                else if ((has_out >> tl) & 1) cand(((st + r_off) << 2) | (uint64_t)(1 + strand));
            }
        }
    }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256)
        if (s_count[i]) atomicAdd(&edge_count[i], (unsigned long long)s_count[i]);
}

// ---- mercy edges (s2_read_mercy_prepare, cx1_read2sdbg_s2.cpp:106-250) ---------------------------------
constexpr int kMercyMaxLen = 1024;   // longest read whose flags live in LDS; longer reads take the variant with its flags in device memory

// sorted candidates -> one wave per read: flag arrays, then the serial gap-filling scan.  GLOBAL: the three flag arrays of a wave
// are `stride` bytes each in gflags (reads of any length), a fixed grid of waves walks the candidate groups.
template <bool GLOBAL>
__global__ __launch_bounds__(256) void mercy_kernel(const Key<2> *cands, uint64_t n_cand, const uint64_t *start_idx, uint64_t n_reads, int k,
                                                    int num_k1_per_read, unsigned long long *is_solid, unsigned long long *num_mercy,
                                                    const uint32_t *pos_to_id, int cand_shift, uint8_t *gflags, uint64_t stride) {
    __shared__ uint8_t s_flags[GLOBAL ? 1 : 4][3][GLOBAL ? 64 : kMercyMaxLen + 64];
    const int lane = lane_id(), wv = wave_id();
    uint8_t *const fl = GLOBAL ? gflags + ((uint64_t)blockIdx.x * 4 + wv) * 3 * stride : &s_flags[GLOBAL ? 0 : wv][0][0];
    const uint64_t fs = GLOBAL ? stride : (uint64_t)(kMercyMaxLen + 64);
    uint8_t *no_in = fl, *no_out = fl + fs, *has_k = fl + 2 * fs;
    auto sync_flags = [&]() {                                             // lanes hand the flags to each other: LDS is in order inside a wave,
        if (GLOBAL) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // device memory needs the stores drained first
    };
    auto val = [&](uint64_t i) { return (((uint64_t)cands[i].w[0] << 32) | cands[i].w[1]) >> cand_shift; };
    auto read_of = [&](uint64_t abs) { return read_of_base(start_idx, n_reads, pos_to_id, abs); };
    // every wave takes the candidate ranges whose first candidate index is a multiple-of-stride hit: simple static split by
    // candidate index: wave g handles the reads whose FIRST candidate lies in [g*64, g*64+64)
    const uint64_t n_groups = (n_cand + 63) / 64, n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t g = (uint64_t)blockIdx.x * 4 + wv; g < n_groups; g += n_waves) {
    const uint64_t c_lo = g * 64, c_hi = c_lo + 64 < n_cand ? c_lo + 64 : n_cand;
    for (uint64_t c = c_lo; c < c_hi; ++c) {
        const uint64_t read_id = read_of(val(c) >> 2);
        if (c > 0 && read_of(val(c - 1) >> 2) == read_id) continue;             // not the first candidate of its read
        const uint64_t st = start_idx[read_id];
        const int len = (int)(start_idx[read_id + 1] - st);
        if (!GLOBAL && len > kMercyMaxLen) continue;                             // (the host launches the other variant then)
        for (int i = lane; i < len + 2; i += 64) { no_in[i] = 0; no_out[i] = 0; has_k[i] = 0; }
        sync_flags();
        int first_0_out = 1 << 30, last_0_in = -1;
        uint64_t e = c;
        // candidates of this read are contiguous
        while (true) {
            uint64_t idx = e + lane;
            bool mine = idx < n_cand && read_of(val(idx) >> 2) == read_id;
            if (mine) {
                uint64_t v = val(idx);
                int off = (int)((v >> 2) - st);
                int code = (int)(v & 3);
                if (code == 2) { no_out[off] = 1; first_0_out = off < first_0_out ? off : first_0_out; }
                else if (code == 1) { no_in[off] = 1; last_0_in = off > last_0_in ? off : last_0_in; }
                has_k[off] = 1;
            }
            uint64_t bal = __ballot(mine);
            if (bal != ~0ull) break;
            e += 64;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            int o1 = __shfl_xor(first_0_out, d, 64), o2 = __shfl_xor(last_0_in, d, 64);
            first_0_out = o1 < first_0_out ? o1 : first_0_out;
            last_0_in = o2 > last_0_in ? o2 : last_0_in;
        }
        if (last_0_in < first_0_out) continue;
        sync_flags();
        auto solid = [&](int i) {
            uint64_t bit = (uint64_t)num_k1_per_read * read_id + (uint64_t)i;
            return (int)((is_solid[bit >> 6] >> (bit & 63)) & 1);
        };
        for (int i = lane; i + k < len; i += 64)
            if (solid(i)) { has_k[i] = 1; has_k[i + 1] = 1; }
        sync_flags();
        if (lane == 0) {
            int last_no_out = -1;
            unsigned long long added = 0;
            for (int i = 0; i + k <= len; ++i) {
                if (no_in[i] && last_no_out != -1) {
                    for (int j = last_no_out; j < i; ++j) {
                        uint64_t bit = (uint64_t)num_k1_per_read * read_id + (uint64_t)j;
                        atomicOr(&is_solid[bit >> 6], 1ull << (bit & 63));
                    }
                    added += (unsigned long long)(i - last_no_out);
                }
                if (has_k[i]) last_no_out = -1;
                if (no_out[i]) last_no_out = i;
            }
            if (added) atomicAdd(num_mercy, added);
        }
        sync_flags();
    }
    }
}


__global__ __launch_bounds__(256) void max_len_kernel(const uint64_t *start, uint64_t n_short, unsigned int *out) {
    uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned int len = r < n_short ? (unsigned int)(start[r + 1] - start[r]) : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { unsigned int o = __shfl_xor(len, d, 64); len = o > len ? o : len; }
    if (lane_id() == 0 && len) atomicMax(out, len);
}

}  // namespace mgta
