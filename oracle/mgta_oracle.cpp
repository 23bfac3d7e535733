// mgta_oracle.cpp — CPU ORACLE.   *** TEST INFRASTRUCTURE, NOT PRODUCT CODE ***
//
// Plain C++17 restatement of the MegaGTA hot path (read -> SdBG edges -> succinct graph navigation
// -> profile-HMM A*), written for clarity, not speed.  Each block cites the reference file:line
// whose behaviour it follows (paths relative to /root/reference/src).  Pinned against golden
// vectors generated from the compiled reference (tests/golden/, tests/test_oracle_golden.py).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
#include "mgta_oracle.h"

#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <queue>
#include <sstream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int kBuckets = ORC_NUM_BUCKETS;
constexpr int kMaxMulti = 65535;    // definitions.h:33
constexpr int kMulti2Sp = 255;      // definitions.h:37
constexpr int kMaxMulti2 = 254;     // definitions.h:36
constexpr int kDollar = 4;          // kSentinelValue, cx1_read2sdbg.h:72

// =================================================================================================
// 1. logical SdBG edge stream
// =================================================================================================
struct Stream {
    int k = 0, words_per_tip = 0;
    int64_t n_items_sorted = 0;
    std::vector<int64_t> bucket_items = std::vector<int64_t>(kBuckets, 0);
    std::vector<uint16_t> records, large;
    std::vector<uint32_t> tips;
};

struct BucketOut {
    std::vector<uint16_t> records, large;
    std::vector<uint32_t> tips;
};

inline int base_at(const uint32_t *packed, uint64_t pos) {   // sequence_package.h:126-129
    return (packed[pos >> 4] >> (30 - 2 * (pos & 15))) & 3;
}

// Key of one sort item: `nchars` characters (k or k-1), 2 bits each, left aligned, zero padded;
// low 4 bits of the last word = (is_full_k<<3)|prev   [cx1_read2sdbg_s2.cpp:586-677, 639-641]
template <int W>
inline std::array<uint32_t, W> make_key(const uint8_t *chars, int nchars, int k, int prev) {
    std::array<uint32_t, W> key{};
    for (int j = 0; j < nchars; ++j) key[j >> 4] |= uint32_t(chars[j]) << (30 - 2 * (j & 15));
    key[W - 1] |= uint32_t(nchars == k) << 3;
    key[W - 1] |= uint32_t(prev);
    return key;
}

template <int W>
inline bool same_km1(const std::array<uint32_t, W> &a, const std::array<uint32_t, W> &b, int k) {
    // IsDiffKMinusOneMer, cx1_read2sdbg_s2.cpp:54-75
    int full = (k - 1) / 16, rem = (k - 1) % 16;
    for (int i = 0; i < full; ++i)
        if (a[i] != b[i]) return false;
    if (rem > 0 && (a[full] >> (16 - rem) * 2) != (b[full] >> (16 - rem) * 2)) return false;
    return true;
}

template <int W>
inline int key_a(const std::array<uint32_t, W> &it, int k) {   // Extract_a, s2.cpp:83-94
    if ((it[W - 1] >> 3) & 1) return (it[(k - 1) / 16] >> (15 - (k - 1) % 16) * 2) & 3;
    return kDollar;
}
template <int W>
inline int key_b(const std::array<uint32_t, W> &it) { return it[W - 1] & 7; }   // Extract_b, s2.cpp:96-98

// output_(), cx1_read2sdbg_s2.cpp:742-835 + SdbgWriter::write, sdbg_multi_io.h:83-112
template <int W>
void emit_bucket(const std::array<uint32_t, W> *it, int64_t n, int k, int words_per_tip, BucketOut &out) {
    int64_t start = 0;
    while (start < n) {
        int64_t end = start + 1;
        while (end < n && same_km1<W>(it[start], it[end], k)) ++end;
        int has_solid_a = 0, has_solid_b = 0, outputed_b = 0;
        int64_t last_a[4] = {-1, -1, -1, -1};
        for (int64_t i = start; i < end; ++i) {
            int a = key_a<W>(it[i], k), b = key_b<W>(it[i]);
            if (a != kDollar && b != kDollar) { has_solid_a |= 1 << a; has_solid_b |= 1 << b; }
            if (a != kDollar && (b != kDollar || !(has_solid_a & (1 << a)))) last_a[a] = i;
        }
        for (int64_t i = start, j; i < end; i = j) {
            int a = key_a<W>(it[i], k), b = key_b<W>(it[i]);
            j = i + 1;
            while (j < end && key_a<W>(it[j], k) == a && key_b<W>(it[j]) == b) ++j;
            int count = (int)std::min<int64_t>(j - i, kMaxMulti);
            int is_dollar = 0;
            if (a == kDollar) {
                if (has_solid_b & (1 << b)) continue;
                is_dollar = 1;
            }
            if (b == kDollar) {
                if (has_solid_a & (1 << a)) continue;
            }
            int w = (b == kDollar) ? 0 : ((outputed_b & (1 << b)) ? b + 5 : b + 1);
            int last = (a == kDollar) ? 0 : (last_a[a] == j - 1 ? 1 : 0);
            outputed_b |= 1 << b;
            out.records.push_back(uint16_t(w | (last << 4) | (is_dollar << 5) | (std::min(count, kMulti2Sp) << 8)));
            if (count > kMaxMulti2) out.large.push_back(uint16_t(count));
            if (is_dollar)
                for (int t = 0; t < words_per_tip; ++t) out.tips.push_back(it[i][t]);
        }
        start = end;
    }
}

// Enumerate the sort items of one read (stage 2, every position solid).
// s2_lv0_calc_bucket_size / s2_lv1_fill_offset / s2_lv2_extract_substr_, cx1_read2sdbg_s2.cpp:252-315,475-677
// `solid` (may be null = every position solid, the -m 1 case): one flag per (k+1)-mer position of the read; items are generated for the
// solid positions only and the `$` items at the ends of every RUN of solid positions (s2.cpp:276-297,528-565).
template <int W, class F>
inline void for_each_item(const uint8_t *r, int len, int k, F &&f, const uint8_t *solid = nullptr) {
    if (len < k + 1) return;                                     // s2.cpp:262-264
    std::vector<uint8_t> rc(k + 1);
    const int npos = len - k;
    for (int p = 0; p + k < len; ++p) {
        if (solid && !solid[p]) continue;
        const uint8_t *e = r + p;                                // edge = (k+1)-mer
        for (int i = 0; i <= k; ++i) rc[i] = 3 - e[k - i];
        bool pal = std::equal(e, e + k + 1, rc.begin());         // s2.cpp:278
        bool first = (p == 0) || (solid && !solid[p - 1]), last = (p == npos - 1) || (solid && !solid[p + 1]);
        if (first) {                                             // left $   (s2.cpp:531-540)
            f(make_key<W>(e, k, k, kDollar));
            if (!pal) f(make_key<W>(rc.data() + 2, k - 1, k, rc[1]));
        }
        f(make_key<W>(e + 1, k, k, e[0]));                       // solid    (s2.cpp:543-550)
        if (!pal) f(make_key<W>(rc.data() + 1, k, k, rc[0]));
        if (last) {                                              // right $  (s2.cpp:553-562)
            f(make_key<W>(e + 2, k - 1, k, e[1]));
            if (!pal) f(make_key<W>(rc.data(), k, k, kDollar));
        }
    }
}

using SolidFlags = std::vector<std::vector<uint8_t>>;           // [read][(k+1)-mer position]; an empty row = every position solid

template <int W>
Stream *build_stream(const uint32_t *packed, const uint64_t *start_idx, uint64_t n_reads, int k, int n_threads, const SolidFlags *solid = nullptr) {
    using Key = std::array<uint32_t, W>;
    auto *s = new Stream;
    s->k = k;
    s->words_per_tip = (2 * k + 31) / 32;                        // sdbg_multi_io.h:63
    std::vector<int64_t> cnt(kBuckets + 1, 0);
    int maxlen = 0;
    for (uint64_t r = 0; r < n_reads; ++r) maxlen = std::max<int>(maxlen, int(start_idx[r + 1] - start_idx[r]));
    std::vector<uint8_t> buf(maxlen + 1);
    auto decode = [&](uint64_t r) {
        int len = int(start_idx[r + 1] - start_idx[r]);
        for (int i = 0; i < len; ++i) buf[i] = (uint8_t)base_at(packed, start_idx[r] + i);
        return len;
    };
    auto solid_of = [&](uint64_t r) -> const uint8_t * { return (solid && !(*solid)[r].empty()) ? (*solid)[r].data() : nullptr; };
    // pass 1: bucket sizes (bucket = first 8 characters of the key, s2.cpp:832 `key[0] >> 16`)
    for (uint64_t r = 0; r < n_reads; ++r) {
        int len = decode(r);
        for_each_item<W>(buf.data(), len, k, [&](const Key &key) { ++cnt[(key[0] >> 16) + 1]; }, solid_of(r));
    }
    for (int b = 0; b < kBuckets; ++b) cnt[b + 1] += cnt[b];
    int64_t total = cnt[kBuckets];
    s->n_items_sorted = total;
    std::vector<Key> items((size_t)total);
    std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
    // pass 2: materialise the keys bucket by bucket
    for (uint64_t r = 0; r < n_reads; ++r) {
        int len = decode(r);
        for_each_item<W>(buf.data(), len, k, [&](const Key &key) { items[(size_t)fill[key[0] >> 16]++] = key; }, solid_of(r));
    }
    // per bucket: sort ascending as W big-endian words (lv2_cpu_sort.h:87-150) and emit
    std::vector<BucketOut> outs(kBuckets);
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads > 0 ? n_threads : 1)
    for (int b = 0; b < kBuckets; ++b) {
        int64_t lo = cnt[b], hi = cnt[b + 1];
        if (lo == hi) continue;
        std::sort(items.begin() + lo, items.begin() + hi);
        emit_bucket<W>(items.data() + lo, hi - lo, k, s->words_per_tip, outs[b]);
    }
    for (int b = 0; b < kBuckets; ++b) {
        s->bucket_items[b] = (int64_t)outs[b].records.size();
        s->records.insert(s->records.end(), outs[b].records.begin(), outs[b].records.end());
        s->large.insert(s->large.end(), outs[b].large.begin(), outs[b].large.end());
        s->tips.insert(s->tips.end(), outs[b].tips.begin(), outs[b].tips.end());
    }
    return s;
}

// =================================================================================================
// 1b. stage 1 (-m >= 2): which (k+1)-mer positions of which read are solid, mercy edges, the .counting histogram
//     cx1_read2sdbg_s1.cpp:177-229,408-596 (items), :671-830 (s1_lv2_output_), :905-951 (s1_post_proc);
//     cx1_read2sdbg_s2.cpp:106-250 (s2_read_mercy_prepare)
// Plain restatement: every (k-1)-mer S of every read is listed once, in its smaller orientation, with the characters around
// it:  prev head [S] tail next.  The multiplicity of (head S tail) is the multiplicity of that (k+1)-mer.
// =================================================================================================
struct S1Item {
    std::string S;                 // k-1 characters 0..3
    uint8_t head, tail, prev, next;   // 0..3 or kDollar
    uint64_t abs;                  // absolute base index of S[0] in the concatenated reads (start_idx[read] + offset)
    uint8_t strand;
};

struct Stage1Out {
    SolidFlags solid;
    std::vector<int64_t> counting = std::vector<int64_t>(65536, 0);   // [m] = number of distinct (k+1)-mers seen m times (m capped at 65535)
    int64_t n_mercy = 0;
};

static Stage1Out stage1(const uint32_t *packed, const uint64_t *start_idx, uint64_t n_reads, uint64_t n_short, int k, int threshold,
                        bool need_mercy) {
    Stage1Out out;
    out.solid.resize(n_reads);
    const int km1 = k - 1;
    std::vector<S1Item> items;
    auto comp = [](int c) { return c == kDollar ? kDollar : 3 - c; };
    std::vector<uint8_t> buf;
    for (uint64_t r = 0; r < n_reads; ++r) {
        const int len = int(start_idx[r + 1] - start_idx[r]);
        if (len < k + 1) continue;                                                    // s1.cpp:187-189
        if (r < n_short) out.solid[r].assign(len - k, 0);                             // assist sequences stay "all solid" (empty row), s2.cpp:276
        buf.resize(len);
        for (int i = 0; i < len; ++i) buf[i] = (uint8_t)base_at(packed, start_idx[r] + i);
        const int n_off = len - k + 2;                                                // (k-1)-mers of the read
        for (int o = 0; o < n_off; ++o) {
            std::string f(buf.begin() + o, buf.begin() + o + km1), rc(km1, 0);
            for (int i = 0; i < km1; ++i) rc[i] = (char)(3 - f[km1 - 1 - i]);
            const int head = o > 0 ? buf[o - 1] : kDollar, prev = o > 1 ? buf[o - 2] : kDollar;            // s1.cpp:534-563
            const int tail = o + km1 < len ? buf[o + km1] : kDollar, next = o + k < len ? buf[o + k] : kDollar;
            bool fwd, rev;
            if (o == 0 || o == n_off - 1) fwd = rev = true;                           // the first / last (k-1)-mer goes in both strands (s1.cpp:470-472,503-506)
            else if (f != rc) { fwd = f < rc; rev = !fwd; }
            else { fwd = head <= 3 - tail; rev = !fwd; }                              // palindrome (s1.cpp:488-497)
            const uint64_t abs = start_idx[r] + (uint64_t)o;
            if (fwd) items.push_back(S1Item{f, (uint8_t)head, (uint8_t)tail, (uint8_t)prev, (uint8_t)next, abs, 0});
            if (rev)                                                                  // s1.cpp:575-582
                items.push_back(S1Item{rc, (uint8_t)comp(tail), (uint8_t)comp(head), (uint8_t)comp(next), (uint8_t)comp(prev), abs, 1});
        }
    }
    std::sort(items.begin(), items.end(), [](const S1Item &a, const S1Item &b) {
        if (a.S != b.S) return a.S < b.S;
        if (a.head != b.head) return a.head < b.head;
        return a.tail < b.tail;
    });
    std::vector<uint64_t> cand;                                                      // mercy candidates: (absolute k-mer index << 2) | code
    auto read_of = [&](uint64_t abs) {                                               // SequencePackage::get_id, sequence_package.h:164-188
        return (uint64_t)(std::upper_bound(start_idx, start_idx + n_reads, abs) - start_idx) - 1;
    };
    for (size_t g0 = 0; g0 < items.size();) {
        size_t g1 = g0;
        while (g1 < items.size() && items[g1].S == items[g0].S) ++g1;
        // 5 x 5 count tables of the group (s1.cpp:716-748); '$' never counts towards a mask
        int64_t cph[4][4] = {}, ctn[4][4] = {}, cht[4][4] = {};
        for (size_t i = g0; i < g1; ++i) {
            const S1Item &it = items[i];
            if (it.prev < 4 && it.head < 4) ++cph[it.prev][it.head];
            if (it.tail < 4 && it.next < 4) ++ctn[it.tail][it.next];
            if (it.head < 4 && it.tail < 4) ++cht[it.head][it.tail];
        }
        int has_in = 0, has_out = 0, l_has_out = 0, r_has_in = 0;
        for (int j = 0; j < 4; ++j)
            for (int q = 0; q < 4; ++q) {
                if (cph[q][j] >= threshold) has_in |= 1 << j;
                if (ctn[j][q] >= threshold) has_out |= 1 << j;
                if (cht[j][q] >= threshold) { l_has_out |= 1 << j; r_has_in |= 1 << q; }
            }
        for (size_t h0 = g0; h0 < g1;) {                                             // sub-groups of equal (head, tail): one (k+1)-mer each
            size_t h1 = h0;
            while (h1 < g1 && items[h1].head == items[h0].head && items[h1].tail == items[h0].tail) ++h1;
            const int hd = items[h0].head, tl = items[h0].tail;
            const bool real = hd != kDollar && tl != kDollar;
            const int64_t cnt = (int64_t)(h1 - h0);
            if (real) ++out.counting[(size_t)std::min<int64_t>(cnt, 65535)];           // s1.cpp:756-758
            const bool solid = real && cnt >= threshold;
            for (size_t i = h0; i < h1; ++i) {                                         // s1.cpp:750-828
                const S1Item &it = items[i];
                const uint64_t read_id = read_of(it.abs);
                if (read_id >= n_short) continue;
                const uint64_t st = start_idx[read_id];
                const int64_t offset = (int64_t)(it.abs - st) - 1;                     // position of the (k+1)-mer head S tail in the read
                const int64_t l_off = it.strand == 0 ? offset : offset + 1, r_off = it.strand == 0 ? offset + 1 : offset;
                auto add = [&](int64_t off, int code) { if (need_mercy) cand.push_back(((st + (uint64_t)off) << 2) | (uint64_t)code); };
                if (solid) {
                    out.solid[read_id][(size_t)offset] = 1;
                    if (!((has_in >> hd) & 1)) add(l_off, 1 + it.strand);
                    if (!((has_out >> tl) & 1)) add(r_off, 2 - it.strand);
                } else {
                    if (hd < 4) {
                        if ((l_has_out >> hd) & 1) add(l_off, ((has_in >> hd) & 1) ? 0 : 1 + it.strand);
                        else if ((has_in >> hd) & 1) add(l_off, 2 - it.strand);
                    }
                    if (tl < 4) {
                        if ((r_has_in >> tl) & 1) add(r_off, ((has_out >> tl) & 1) ? 0 : 2 - it.strand);
                        else if ((has_out >> tl) & 1) add(r_off, 1 + it.strand);
                    }
                }
            }
            h0 = h1;
        }
        g0 = g1;
    }
    // mercy edges: a stretch of non-solid positions between a k-mer with no solid out-edge and a later one with no solid in-edge
    // of the same read is made solid (s2.cpp:164-231)
    std::sort(cand.begin(), cand.end());
    for (size_t i = 0; i < cand.size();) {
        const uint64_t read_id = read_of(cand[i] >> 2), st = start_idx[read_id];
        const int len = int(start_idx[read_id + 1] - st);
        std::vector<uint8_t> no_in(len + 2, 0), no_out(len + 2, 0), has_solid_kmer(len + 2, 0);
        int first_0_out = 1 << 30, last_0_in = -1;
        for (; i < cand.size() && read_of(cand[i] >> 2) == read_id; ++i) {
            const int off = int((cand[i] >> 2) - st), code = int(cand[i] & 3);
            if (code == 2) { no_out[off] = 1; first_0_out = std::min(first_0_out, off); }
            else if (code == 1) { no_in[off] = 1; last_0_in = std::max(last_0_in, off); }
            has_solid_kmer[off] = 1;
        }
        if (last_0_in < first_0_out) continue;
        std::vector<uint8_t> &sol = out.solid[read_id];
        for (int p = 0; p + k < len; ++p)
            if (sol[p]) has_solid_kmer[p] = has_solid_kmer[p + 1] = 1;
        int last_no_out = -1;
        for (int p = 0; p + k <= len; ++p) {
            if (no_in[p] && last_no_out != -1) {
                for (int j = last_no_out; j < p; ++j) sol[j] = 1;
                out.n_mercy += p - last_no_out;
            }
            if (has_solid_kmer[p]) last_no_out = -1;
            if (no_out[p]) last_no_out = p;
        }
    }
    return out;
}

template <int W>
static Stream *build_dispatch(const uint32_t *packed, const uint64_t *start_idx, uint64_t n_reads, int k, int n_threads, const SolidFlags *solid) {
    return build_stream<W>(packed, start_idx, n_reads, k, n_threads, solid);
}
static Stream *build_any(const uint32_t *packed, const uint64_t *start_idx, uint64_t n_reads, int k, int n_threads, const SolidFlags *solid) {
    const int W = (2 * k + 4 + 31) / 32;                          // words_per_substring, s2.cpp:331
    switch (W) {
    case 1: return build_dispatch<1>(packed, start_idx, n_reads, k, n_threads, solid);
    case 2: return build_dispatch<2>(packed, start_idx, n_reads, k, n_threads, solid);
    case 3: return build_dispatch<3>(packed, start_idx, n_reads, k, n_threads, solid);
    case 4: return build_dispatch<4>(packed, start_idx, n_reads, k, n_threads, solid);
    case 5: return build_dispatch<5>(packed, start_idx, n_reads, k, n_threads, solid);
    case 6: return build_dispatch<6>(packed, start_idx, n_reads, k, n_threads, solid);
    case 7: return build_dispatch<7>(packed, start_idx, n_reads, k, n_threads, solid);
    case 8: return build_dispatch<8>(packed, start_idx, n_reads, k, n_threads, solid);
    default: return build_dispatch<9>(packed, start_idx, n_reads, k, n_threads, solid);
    }
}

// =================================================================================================
// 2. succinct de Bruijn graph
// =================================================================================================
struct Graph {
    int64_t size = 0;
    int k = 0, words_per_tip = 0;
    int64_t f[6] = {0, 0, 0, 0, 0, 0};
    int64_t rank_f[6] = {0, 0, 0, 0, 0, 0};
    int64_t num_tips = 0;
    std::vector<uint64_t> w, last, tip, invalid, multi1;
    std::vector<uint32_t> tip_labels;
    // simple sampled rank directories (layout is ours; only the ANSWERS follow rank_and_select.h)
    std::vector<int64_t> last_cum;            // ones before 64-bit word i
    std::vector<int64_t> tip_cum;
    std::vector<int64_t> w_cum[9];            // occurrences of c before 64-bit word i (16 chars)
    int64_t w_total[9] = {0};
    int64_t last_total = 0;

    int W(int64_t x) const { return (w[x >> 4] >> ((x & 15) * 4)) & 15; }                 // succinct_dbg.h:88
    bool bit(const std::vector<uint64_t> &v, int64_t x) const { return (v[x >> 6] >> (x & 63)) & 1; }
    bool is_last(int64_t x) const { return bit(last, x); }
    bool is_tip(int64_t x) const { return bit(tip, x); }
    bool last_or_tip(int64_t x) const { return ((last[x >> 6] | tip[x >> 6]) >> (x & 63)) & 1; }   // .h:101
    bool valid(int64_t x) const { return !bit(invalid, x); }
    bool is_multi1(int64_t x) const { return bit(multi1, x); }
    int out_label(int64_t x) const { int c = W(x); return c > 4 ? c - 4 : c; }              // .h:92-95
};

inline int count_sym(uint64_t word, int c) {     // number of 4-bit fields equal to c
    int n = 0;
    for (int i = 0; i < 16; ++i) n += int(((word >> (4 * i)) & 15) == (uint64_t)c);
    return n;
}

int64_t rank_bits(const std::vector<uint64_t> &v, const std::vector<int64_t> &cum, int64_t total, int64_t size, int64_t pos) {
    // number of ones in [0..pos]; pos<0 -> 0; pos>=size-1 -> total   (rank_and_select.h:492-495)
    if (pos < 0) return 0;
    if (pos >= size - 1) return total;
    int64_t wi = pos >> 6;
    int bits = int(pos & 63) + 1;
    uint64_t mask = bits == 64 ? ~0ULL : ((1ULL << bits) - 1);
    return cum[wi] + __builtin_popcountll(v[wi] & mask);
}

int64_t select_bits(const std::vector<uint64_t> &v, const std::vector<int64_t> &cum, int64_t total, int64_t size, int64_t r) {
    // position of the r-th (0-based) one; `size` if r >= total; -1 if r < 0  (rank_and_select.h:560-568)
    if (r >= total) return size;
    if (r < 0) return -1;
    // last word whose cum <= r
    int64_t lo = 0, hi = (int64_t)v.size() - 1;
    while (lo < hi) {
        int64_t mid = (lo + hi + 1) >> 1;
        if (cum[mid] <= r) lo = mid; else hi = mid - 1;
    }
    uint64_t word = v[lo];
    int64_t rem = r - cum[lo];
    for (int64_t i = 0; i < rem; ++i) word &= word - 1;
    return lo * 64 + __builtin_ctzll(word);
}

int64_t g_rank_last(const Graph &g, int64_t pos) { return rank_bits(g.last, g.last_cum, g.last_total, g.size, pos); }
int64_t g_select_last(const Graph &g, int64_t r) { return select_bits(g.last, g.last_cum, g.last_total, g.size, r); }

int64_t g_rank_w(const Graph &g, int c, int64_t pos) {   // rank_and_select.h:153-157
    if (pos < 0) return 0;
    if (pos >= g.size - 1) return g.w_total[c];
    int64_t wi = pos >> 4;
    int n = int(pos & 15) + 1;
    int64_t r = g.w_cum[c][wi];
    uint64_t word = g.w[wi];
    for (int i = 0; i < n; ++i) r += int(((word >> (4 * i)) & 15) == (uint64_t)c);
    return r;
}

int64_t g_select_w(const Graph &g, int c, int64_t r) {   // rank_and_select.h:220-227
    if (r >= g.w_total[c]) return g.size;
    if (r < 0) return -1;
    const auto &cum = g.w_cum[c];
    int64_t lo = 0, hi = (int64_t)g.w.size() - 1;
    while (lo < hi) {
        int64_t mid = (lo + hi + 1) >> 1;
        if (cum[mid] <= r) lo = mid; else hi = mid - 1;
    }
    int64_t rem = r - cum[lo];
    uint64_t word = g.w[lo];
    for (int i = 0; i < 16; ++i)
        if (((word >> (4 * i)) & 15) == (uint64_t)c) {
            if (rem == 0) return lo * 16 + i;
            --rem;
        }
    return g.size;
}

int node_last_char(const Graph &g, int64_t x) {   // GetNodeLastChar, succinct_dbg.h:109-115
    for (int i = 1;; ++i)
        if (g.f[i] > x) return i - 1;
}

int64_t g_forward(const Graph &g, int64_t e) {    // succinct_dbg.h:155-164
    int a = g.out_label(e);
    int64_t count_a = g_rank_w(g, a, e);
    return g_select_last(g, g.rank_f[a] + count_a - 1);
}

int64_t g_backward(const Graph &g, int64_t e) {   // succinct_dbg.h:166-170
    int a = node_last_char(g, e);
    int64_t count_a = g_rank_last(g, e - 1) - g.rank_f[a];
    return g_select_w(g, a, count_a);
}

int g_outgoing(const Graph &g, int64_t e, int64_t *out) {   // succinct_dbg.cpp:78-97
    if (!g.valid(e)) return -1;
    int od = 0;
    int64_t nx = g_forward(g, e);
    do {
        if (g.valid(nx)) out[od++] = nx;
        --nx;
    } while (nx >= 0 && !g.last_or_tip(nx));
    return od;
}

int g_incoming(const Graph &g, int64_t e, int64_t *in) {    // succinct_dbg.cpp:99-127
    if (!g.valid(e)) return -1;
    int64_t first = g_backward(g, e);
    int c = g.W(first);
    int ones = g.last_or_tip(first);
    int id = g.valid(first);
    if (id > 0) in[0] = first;
    for (int64_t y = first + 1; ones < 5 && y < g.size; ++y) {
        ones += g.last_or_tip(y);
        int cur = g.W(y);
        if (cur == c) break;
        if (cur == c + 4 && g.valid(y)) in[id++] = y;
    }
    return id;
}

inline int tip_char(const Graph &g, int64_t tip_rank, int j) {
    const uint32_t *t = g.tip_labels.data() + (size_t)g.words_per_tip * tip_rank;
    return (t[j >> 4] >> (15 - (j & 15)) * 2) & 3;
}

int g_label(const Graph &g, int64_t e, uint8_t *seq) {      // succinct_dbg.cpp:503-528
    int64_t x = e;
    for (int i = g.k - 1; i >= 0; --i) {
        if (g.is_tip(x)) {
            int64_t tr = rank_bits(g.tip, g.tip_cum, g.num_tips, g.size, x) - 1;
            for (int j = 0; j <= i; ++j) seq[i - j] = uint8_t(tip_char(g, tr, j) + 1);
            break;
        }
        x = g_backward(g, x);
        int c = g.W(x);
        seq[i] = uint8_t(c > 4 ? c - 4 : c);
    }
    return g.k;
}

int64_t g_index_node(const Graph &g, const uint8_t *seq) {  // IndexBinarySearch, succinct_dbg.cpp:427-501
    int k = g.k;
    int64_t l = g.f[seq[k - 1]], r = g.f[seq[k - 1] + 1] - 1;
    while (l <= r) {
        int cmp = 0;
        int64_t mid = (l + r) / 2, y = mid;
        for (int i = k - 1; i >= 0; --i) {
            if (g.is_tip(y)) {
                int64_t tr = rank_bits(g.tip, g.tip_cum, g.num_tips, g.size, y) - 1;
                for (int j = 0; j < i; ++j) {
                    int c = tip_char(g, tr, j) + 1;
                    if (c < seq[i - j]) { cmp = -1; break; }
                    if (c > seq[i - j]) { cmp = 1; break; }
                }
                if (cmp == 0) {
                    if (g.is_tip(mid)) cmp = -1;                       // :455-457 (a tip never matches)
                    else {
                        int c = tip_char(g, tr, i) + 1;
                        if (c < seq[0]) cmp = -1;
                        else if (c > seq[0]) cmp = 1;
                    }
                }
                break;
            }
            y = g_backward(g, y);
            int c = g.W(y);
            if (c < seq[i]) { cmp = -1; break; }
            if (c > seq[i]) { cmp = 1; break; }
        }
        if (cmp == 0) {                                                // GetLastIndex = rs_last_.Succ(mid)
            int64_t p = mid;
            while (p < g.size && !g.is_last(p)) ++p;
            return p;
        }
        if (cmp > 0) r = mid - 1; else l = mid + 1;
    }
    return -1;
}

int64_t g_index_edge(const Graph &g, const uint8_t *seq) {  // IndexBinarySearchEdge, succinct_dbg.cpp:530-549
    int64_t node = g_index_node(g, seq);
    if (node == -1) return -1;
    do {
        int lab = g.W(node);
        if (lab == seq[g.k] || lab - 4 == seq[g.k]) return node;
        --node;
    } while (node >= 0 && !g.last_or_tip(node));
    return -1;
}

Graph *graph_from_stream(const Stream &s) {                   // LoadFromMultiFile(prefix,false), succinct_dbg.cpp:595-723
    auto *g = new Graph;
    g->k = s.k;
    g->words_per_tip = s.words_per_tip;
    int64_t n = (int64_t)s.records.size();
    g->size = n;
    // f_: sdbg_multi_io.h:254-268
    g->f[0] = -1; g->f[1] = 0;
    int64_t acc = 0;
    for (int b = 0; b < kBuckets; ++b) { acc += s.bucket_items[b]; g->f[b / (kBuckets / 4) + 2] = acc; }
    size_t nw4 = (size_t)((n + 15) / 16), nw1 = (size_t)((n + 63) / 64);
    g->w.assign(nw4 + 1, 0); g->last.assign(nw1 + 1, 0); g->tip.assign(nw1 + 1, 0); g->multi1.assign(nw1 + 1, 0);
    for (int64_t i = 0; i < n; ++i) {
        uint16_t it = s.records[(size_t)i];
        g->w[i >> 4] |= uint64_t(it & 15) << ((i & 15) * 4);
        g->last[i >> 6] |= uint64_t((it >> 4) & 1) << (i & 63);
        g->tip[i >> 6] |= uint64_t((it >> 5) & 1) << (i & 63);
        g->multi1[i >> 6] |= uint64_t((it >> 8) <= 1) << (i & 63);         // :680
    }
    g->tip_labels = s.tips;
    g->num_tips = s.words_per_tip ? (int64_t)s.tips.size() / s.words_per_tip : 0;
    g->invalid = g->tip;                                                     // :717
    for (int64_t i = 0; i < n; ++i)
        if (g->W(i) == 0) g->invalid[i >> 6] |= 1ULL << (i & 63);           // init(), succinct_dbg.h:81-85
    // rank directories
    g->last_cum.assign(g->last.size() + 1, 0); g->tip_cum.assign(g->tip.size() + 1, 0);
    for (size_t i = 0; i < g->last.size(); ++i) {
        g->last_cum[i + 1] = g->last_cum[i] + __builtin_popcountll(g->last[i]);
        g->tip_cum[i + 1] = g->tip_cum[i] + __builtin_popcountll(g->tip[i]);
    }
    g->last_total = g->last_cum[g->last.size()];
    for (int c = 0; c < 9; ++c) {
        g->w_cum[c].assign(g->w.size() + 1, 0);
        for (size_t i = 0; i < g->w.size(); ++i) {
            int cnt = count_sym(g->w[i], c);
            if (c == 0) {   // padding fields beyond `size` read as 0: do not count them
                int64_t base = (int64_t)i * 16;
                if (base + 16 > n) cnt -= int(std::min<int64_t>(16, base + 16 - std::max<int64_t>(n, base)));
            }
            g->w_cum[c][i + 1] = g->w_cum[c][i] + cnt;
        }
        g->w_total[c] = g->w_cum[c][g->w.size()];
    }
    for (int i = 0; i < 6; ++i) g->rank_f[i] = g_rank_last(*g, g->f[i] - 1);   // succinct_dbg.h:74-76
    return g;
}

// =================================================================================================
// 3. profile HMM (HMMER3 text) + A* heuristic
// =================================================================================================
constexpr double NEG_INF = -std::numeric_limits<double>::infinity();
enum { MM = 0, MI = 1, MD = 2, IM = 3, II = 4, DM = 5, DD = 6 };   // profile_hmm.h:25

struct Hmm {
    int M = 0, A = 0;
    std::vector<double> msc, isc, tsc, maxm, h, compo;
    int alpha[127];
    double m(int k, int j) const { return k == 0 ? NEG_INF : msc[(size_t)k * A + j]; }   // profile_hmm.h:58-64
    double t(int k, int tr) const { return tsc[(size_t)tr * (M + 1) + k]; }
    double hc(int st, int k) const { return h[(size_t)st * (M + 1) + k]; }
};

double parse_p(const std::string &tok) { return tok == "*" ? 0.0 : std::exp(-1 * std::stod(tok)); }  // hmmer3b_parser.h:111-116

double heuristic_from(const Hmm &hm, char pre, int state_no) {    // computeCostInternal, most_probable_path.h:48-118
    double h = 0;
    for (int i = state_no + 1; i <= hm.M; i++) {
        double mt, it, dt;
        switch (pre) {
        case 'm': mt = hm.t(i - 1, MM); it = hm.t(i - 1, MI); dt = hm.t(i - 1, MD); break;
        case 'd': mt = hm.t(i - 1, DM); it = NEG_INF; dt = hm.t(i - 1, DD); break;
        default:  mt = hm.t(i - 1, IM); it = hm.t(i - 1, II); dt = NEG_INF; break;
        }
        double best_m = NEG_INF, best_i = NEG_INF;
        for (int j = 0; j < hm.A; ++j) {
            best_m = std::max(best_m, hm.m(i, j));
            best_i = std::max(best_i, hm.isc[(size_t)i * hm.A + j]);
        }
        mt += best_m - hm.maxm[i];
        dt -= hm.maxm[i];
        it += best_i;
        it = NEG_INF;                                               // :100 (insert branch disabled)
        if (it > mt && it > dt) { h += it; pre = 'i'; i--; }
        else if (dt > mt && dt > it) { h += dt; pre = 'd'; }
        else { h += mt; pre = 'm'; }
    }
    return h;
}

Hmm *parse_hmm(const char *path) {                                   // Parser::readHMM, hmmer3b_parser.h:19-177
    std::ifstream f(path);
    if (!f.is_open()) return nullptr;
    auto *hm = new Hmm;
    std::fill(hm->alpha, hm->alpha + 127, -1);
    std::string line, w1, w2;
    std::getline(f, line);                                            // version line
    while (std::getline(f, line)) {
        std::istringstream iss(line);
        w1.clear(); w2.clear();
        iss >> w1 >> w2;
        if (w1 == "LENG") hm->M = std::stoi(w2);
        else if (w1 == "HMM") {                                       // parseAlpha, :179-201
            std::istringstream a(line);
            std::string tok;
            a >> tok;
            int count = 0;
            while (a >> tok) {
                hm->alpha[toupper(tok[0])] = count;
                hm->alpha[tolower(tok[0])] = count;
                ++count;
            }
            hm->A = count;
            break;
        }
    }
    std::getline(f, line);                                            // transition labels
    std::getline(f, line);                                            // COMPO
    {
        std::istringstream iss(line);
        std::string tag, tok;
        iss >> tag;
        if (tag == "COMPO")
            for (int j = 0; j < hm->A; ++j) { iss >> tok; hm->compo.push_back(std::exp(-1 * std::stod(tok))); }
    }
    int M = hm->M, A = hm->A;
    hm->msc.assign((size_t)(M + 1) * A, 0.0);
    hm->isc.assign((size_t)(M + 1) * A, 0.0);
    hm->tsc.assign((size_t)7 * (M + 1), 0.0);
    hm->maxm.assign(M + 1, NEG_INF);
    for (int i = 0; i <= M; ++i) {
        std::string tok;
        if (i > 0) {
            std::getline(f, line);
            std::istringstream iss(line);
            iss >> tok;                                               // node number
            for (int j = 0; j < A; ++j) {
                iss >> tok;
                double v = std::log(parse_p(tok) / hm->compo[j]);     // normalized, :122-124
                hm->msc[(size_t)i * A + j] = v;
                if (v > hm->maxm[i]) hm->maxm[i] = v;                 // profile_hmm.h:72-78
            }
        }
        std::getline(f, line);                                        // insert emissions: 0 when normalized (:145-147)
        std::getline(f, line);                                        // transitions
        std::istringstream iss(line);
        for (int t = 0; t < 7; ++t) { iss >> tok; hm->tsc[(size_t)t * (M + 1) + i] = std::log(parse_p(tok)); }
    }
    for (int j = 0; j < A; ++j) hm->isc[(size_t)M * A + j] = NEG_INF; // :170-172
    hm->h.assign((size_t)3 * (M + 1), 0.0);
    for (int i = 0; i <= M; ++i) {
        hm->h[i] = heuristic_from(*hm, 'm', i);
        hm->h[(size_t)(M + 1) + i] = heuristic_from(*hm, 'i', i);
        hm->h[(size_t)2 * (M + 1) + i] = heuristic_from(*hm, 'd', i);
    }
    return hm;
}

// =================================================================================================
// 4. HMM-guided A*
// =================================================================================================
const char kCodon[65] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";   // codon.h:9-106

inline char codon_fwd(int c1, int c2, int c3) { return kCodon[c1 * 16 + c2 * 4 + c3]; }
inline char codon_rc(int c1, int c2, int c3) { return kCodon[(3 - c3) * 16 + (3 - c2) * 4 + (3 - c1)]; }  // codon.h:108-209

struct Node {                               // AStarNode, a_star_node.h:9-33
    Node *from = nullptr;
    int nucl_emission = 0;
    double score = 0, real_score = 0, max_score = 0;
    int state_no = 0;
    char state = 'm';
    int fval = 0, indels = 0, length = 0, negative_count = 0;
    int64_t node_id = -1;
};

inline int srank(char s) { return s == 'm' ? 3 : s == 'd' ? 2 : s == 'i' ? 1 : 0; }
inline bool node_less(const Node &a, const Node &b) {   // AStarNode::operator<, a_star_node.h:34-82
    if (a.fval != b.fval) return a.fval < b.fval;
    if (a.state_no != b.state_no) return a.state_no > b.state_no;
    return srank(a.state) < srank(b.state);
}
struct PtrLess { bool operator()(const Node *a, const Node *b) const { return node_less(*a, *b); } };

struct Key {
    int64_t node_id; int state_no; char state;
    bool operator==(const Key &o) const { return node_id == o.node_id && state_no == o.state_no && state == o.state; }
};
struct KeyHash {
    size_t operator()(const Key &k) const {
        uint64_t h = (uint64_t)k.node_id * 0x9E3779B97F4A7C15ULL ^ ((uint64_t)k.state_no << 8) ^ (uint64_t)k.state;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
        return (size_t)h;
    }
};
inline Key key_of(const Node &n) { return Key{n.node_id, n.state_no, n.state}; }

inline int to_fval(double x) {     // (int) cast of a double; x86 cvttsd2si yields INT_MIN when out of range / NaN
    if (!(x > -2147483649.0 && x < 2147483648.0)) return std::numeric_limits<int>::min();
    return (int)x;
}

using Cache = std::unordered_map<Key, Key, KeyHash>;   // term_nodes: parent key -> child key, first insert wins

struct Searcher {
    const Graph *g;
    const Hmm *hm[2];     // 0 = forward model, 1 = reverse model
    int prune;
    double low_cov_penalty;                                 // -log(low_cov_pen), node_enumerator.h:42
    double exit_prob[3000];                                 // hmm_graph_search.h:48-52
    Cache cache[2];
    // windowed warm mode: the path seed j found with c_j expansions is seen by the seeds >= j + window + c_j / cost_rate (cost_rate 0:
    // no cost term; window 1 then == the reference's sequential sharing)
    int window = 1, cost_rate = 0;
    int64_t cost_knee = 0; int cost_rate2 = 0;                 // concave cost term: c / rate up to the knee, knee / rate + (c - knee) / rate2 beyond (knee 0: one rate)
    int64_t seed_counter = 0;
    struct Pending { int64_t seed; int dir; Key parent, child; int em; int64_t visible_from; };
    std::vector<Pending> pending;
    int cur_dir = 0;
    std::vector<Node *> pool;
    Node *alloc() { pool.push_back(new Node); return pool.back(); }
    void release() { for (Node *n : pool) delete n; pool.clear(); }
};

// NodeEnumerator::enumerateNodes, node_enumerator.h:65-246
void enumerate(const Searcher &S, const Hmm &hm, Node &curr, bool forward, const Key *child, std::vector<Node> &ret) {
    ret.clear();
    int next_state = curr.state_no + 1;
    double mt, it, dt;
    switch (curr.state) {
    case 'm': mt = hm.t(curr.state_no, MM); it = hm.t(curr.state_no, MI); dt = hm.t(curr.state_no, MD); break;
    case 'd': mt = hm.t(curr.state_no, DM); it = NEG_INF; dt = hm.t(curr.state_no, DD); break;
    default:  mt = hm.t(curr.state_no, IM); it = hm.t(curr.state_no, II); dt = NEG_INF; break;
    }
    double max_match = hm.maxm[next_state];
    if (curr.node_id == -1) return;
    const Graph &g = *S.g;
    int64_t n1[4], n2[4], n3[4];
    std::vector<int64_t> packed;
    int od1 = g_outgoing(g, curr.node_id, n1);
    for (int i = 0; i < od1; ++i) {
        int od2 = g_outgoing(g, n1[i], n2);
        for (int j = 0; j < od2; ++j) {
            int od3 = g_outgoing(g, n2[j], n3);
            for (int k = 0; k < od3; ++k) {
                int64_t p = (n3[k] << 16) | ((g.out_label(n1[i]) - 1) << 6) | ((g.out_label(n2[j]) - 1) << 3) |
                            (g.out_label(n3[k]) - 1);
                p |= int64_t(g.is_multi1(n1[i]) && g.is_multi1(n2[j]) && g.is_multi1(n3[k])) << 9;
                packed.push_back(p);        // the low-coverage-sibling flag (bit 10) is set after the push: never used (:119-125)
            }
        }
    }
    for (int64_t p : packed) {
        int c1 = (p >> 6) & 7, c2 = (p >> 3) & 7, c3 = p & 7;
        char em = forward ? codon_fwd(c1, c2, c3) : codon_rc(c1, c2, c3);
        if (em == '*') continue;
        if (child && child->node_id != (p >> 16)) continue;
        double pen = (p & (1 << 9)) ? S.low_cov_penalty : 0;
        int aa = hm.alpha[(int)em];
        Node nx;
        nx.from = &curr; nx.state_no = next_state; nx.state = 'm';
        double e = mt + hm.msc[(size_t)next_state * hm.A + aa];
        nx.real_score = curr.real_score + e - pen;
        if (nx.real_score >= curr.max_score) { nx.max_score = nx.real_score; nx.negative_count = 0; }
        else { nx.max_score = curr.max_score; nx.negative_count = curr.negative_count + 1; }
        nx.nucl_emission = int(p & 511);
        double self = e - pen - max_match;
        nx.length = curr.length + 1;
        nx.score = curr.score + self;
        nx.fval = to_fval(10000 * (nx.score + 2.0 * hm.hc(0, next_state)));
        nx.indels = curr.indels;
        nx.node_id = p >> 16;
        ret.push_back(nx);
        if (child && *child == key_of(nx)) return;
        if (curr.state != 'd') {
            Node ni;
            ni.from = &curr; ni.state_no = curr.state_no; ni.state = 'i';
            double ei = it + hm.isc[(size_t)next_state * hm.A + aa];
            ni.real_score = curr.real_score + ei - pen;
            ni.max_score = curr.max_score;
            ni.negative_count = curr.negative_count + 1;
            ni.nucl_emission = int(p & 511);
            ni.length = curr.length + 1;
            ni.score = curr.score + (ei - pen);
            ni.fval = to_fval(10000 * (ni.score + 2.0 * hm.hc(1, curr.state_no)));
            ni.indels = curr.indels + 1;
            ni.node_id = p >> 16;
            ret.push_back(ni);
            if (child && *child == key_of(ni)) return;
        }
    }
    if (curr.state != 'i') {
        Node nd;
        nd.from = &curr; nd.state_no = next_state; nd.state = 'd';
        nd.real_score = curr.real_score + dt;
        nd.max_score = curr.max_score;
        nd.negative_count = curr.negative_count + 1;
        nd.nucl_emission = (4 << 6) | (4 << 3) | 4;
        nd.length = curr.length;
        nd.score = curr.score + (dt - max_match);
        nd.fval = to_fval(10000 * (nd.score + 2.0 * hm.hc(2, next_state)));
        nd.indels = curr.indels + 1;
        nd.node_id = curr.node_id;
        ret.push_back(nd);
    }
}

struct AstarOut { bool ok = false; Node goal; bool has_goal = false; int partial = 0; int64_t closed = 0, expanded = 0, opened = 0; };

// getHighestScoreNode, hmm_graph_search.h:345-356.  Returns the pool node chosen.
Node *highest_score(Node *inter) {
    Node *best = inter;
    for (Node *p = inter->from; p; p = p->from)
        if (p->real_score > best->real_score) best = p;
    return best;
}

// astarSearch (core loop), hmm_graph_search.h:191-343
// `lookup` = term_nodes.find (:212,279): the child recorded for a key, or null
template <class Lookup>
Node *astar_with(Searcher &S, const Hmm &hm, Node *start, bool forward, const Lookup &lookup, AstarOut &out) {
    out = AstarOut();
    if (start->state_no >= hm.M) { out.ok = true; return start; }           // :193-197
    static const double log2v = std::log(2);
    std::priority_queue<Node *, std::vector<Node *>, PtrLess> open;
    std::unordered_set<Key, KeyHash> closed;
    std::unordered_map<Key, Node *, KeyHash> open_hash;
    std::vector<Node> kids;
    auto cached_child = [&](const Node &n) -> const Key * { return lookup(key_of(n)); };
    enumerate(S, hm, *start, forward, cached_child(*start), kids);         // :212-233: no pruning / dedup here
    if (start->node_id != -1) out.expanded++;
    for (Node &nx : kids) { Node *p = S.alloc(); *p = nx; open.push(p); }
    out.opened = 1;
    if (open.empty()) { out.ok = false; return nullptr; }                  // :235-237
    Node *inter = start;
    auto better = [&](const Node &a, const Node &b) {
        return (a.real_score + S.exit_prob[a.length]) / log2v > (b.real_score + S.exit_prob[b.length]) / log2v;
    };
    while (!open.empty()) {
        Node *curr = open.top();
        open.pop();
        if (closed.count(key_of(*curr))) continue;                         // :254
        if (curr->state_no >= hm.M) {                                      // :259-270
            if (better(*curr, *inter)) inter = curr;
            out.ok = true;
            return highest_score(inter);
        }
        closed.insert(key_of(*curr));
        out.closed++;
        if (better(*curr, *inter)) inter = curr;                           // :274-277
        enumerate(S, hm, *curr, forward, cached_child(*curr), kids);
        out.expanded++;
        for (Node &nx : kids) {
            bool open_node = false;
            bool admissible = S.prune > 0 ? ((nx.length < 5 || nx.negative_count <= S.prune) && nx.real_score > 0.0) : true;  // :292-293
            if (admissible) {
                auto got = open_hash.find(key_of(nx));
                if (got != open_hash.end()) { if (node_less(*got->second, nx)) open_node = true; }   // :299-302
                else open_node = true;
            }
            if (open_node) {
                Node *p = S.alloc();
                *p = nx;
                open_hash[key_of(nx)] = p;                                 // :331
                out.opened++;
                open.push(p);
            }
        }
    }
    out.partial = 1;                                                       // :339
    out.ok = true;
    return highest_score(inter);
}

Node *astar(Searcher &S, const Hmm &hm, Node *start, bool forward, Cache &cache, AstarOut &out) {
    return astar_with(S, hm, start, forward, [&](const Key &k) -> const Key * {
        auto it = cache.find(k);
        return it == cache.end() ? nullptr : &it->second;
    }, out);
}

// partialResultFromGoal, hmm_graph_search.h:83-110
std::string path_string(Searcher &S, int dir, Node *goal) {
    std::string s;
    for (Node *p = goal; p && p->from; p = p->from) {
        if (p->state != 'd')
            for (int i = 0; i < 3; ++i) s.push_back("acgt-"[(p->nucl_emission >> (3 * i)) & 7]);
        // term_nodes.insert keeps the first value of a key (hash_table_st.h:309-331); applied when the window allows
        S.pending.push_back(Searcher::Pending{S.seed_counter, dir, key_of(*p->from), key_of(*p),
                                              p->nucl_emission | ((p->state == 'm' ? 0 : p->state == 'i' ? 1 : 2) << 9), -1});
    }
    std::reverse(s.begin(), s.end());
    return s;
}

inline char comp(char c) {   // hmm_graph_search.h:362-389
    switch (c) {
    case 'a': case 'A': return 't';
    case 'c': case 'C': return 'g';
    case 'g': case 'G': return 'c';
    case 't': case 'T': return 'a';
    case 'n': case 'N': return 'n';
    default: return '-';
    }
}
std::string revcomp(std::string s) {
    std::reverse(s.begin(), s.end());
    for (auto &c : s) c = comp(c);
    return s;
}
inline int dna_sym(char c) {  // dna_map, hmm_graph_search.h:54-58  (N -> 3 i.e. G)
    switch (c) {
    case 'A': case 'a': return 1;
    case 'C': case 'c': return 2;
    case 'G': case 'g': case 'N': case 'n': return 3;
    case 'T': case 't': return 4;
    default: return -1;
    }
}

// astarSearch (start-node set-up), hmm_graph_search.h:132-189
Node *start_from_kmer(Searcher &S, int dir, int starting_state, const std::string &kmer) {
    const Hmm &hm = *S.hm[dir];
    bool forward = dir == 0;
    int k = S.g->k;
    // protein of the k-mer (libseq translation == standard table), reversed for the left search
    std::string prot;
    for (size_t i = 0; i + 2 < kmer.size() && i / 3 < kmer.size() / 3; i += 3)
        prot.push_back(codon_fwd(dna_sym(kmer[i]) - 1, dna_sym(kmer[i + 1]) - 1, dna_sym(kmer[i + 2]) - 1));
    if (!forward) std::reverse(prot.begin(), prot.end());
    std::string word = forward ? kmer : revcomp(kmer);
    std::vector<uint8_t> seq(k + 1);
    for (int i = 0; i < k + 1; ++i) seq[i] = (uint8_t)dna_sym(word[i]);
    Node *st = S.alloc();
    st->from = nullptr;
    st->state_no = starting_state + int(kmer.size() / 3);
    st->state = 'm';
    st->length = int(kmer.size() / 3);
    st->fval = 0;
    double sc = 0, rs = 0;                                                   // scoreStart / realScoreStart, :112-130
    for (int i = 1; i <= (int)prot.size(); ++i) {
        int aa = hm.alpha[(int)prot[i - 1]];
        sc += hm.msc[(size_t)(starting_state + i) * hm.A + aa] + hm.t(starting_state + i - 1, MM) - hm.maxm[starting_state + i];
        rs += hm.msc[(size_t)(starting_state + i) * hm.A + aa] + hm.t(starting_state + i - 1, MM);
    }
    st->score = sc;
    st->real_score = rs;
    st->node_id = g_index_edge(*S.g, seq.data());
    return st;
}
Node *astar_from_kmer(Searcher &S, int dir, int starting_state, const std::string &kmer, AstarOut &out) {
    return astar(S, *S.hm[dir], start_from_kmer(S, dir, starting_state, kmer), dir == 0, S.cache[dir], out);
}

void fill_result(orc_astar_result *r, const AstarOut &o, const Node *goal) {
    if (!r) return;
    std::memset(r, 0, sizeof(*r));
    r->ok = o.ok; r->partial = o.partial;
    r->n_closed = o.closed; r->n_expanded = o.expanded; r->n_opened = o.opened;
    r->state = '-'; r->state_no = -1; r->node_id = -1;
    if (goal) {
        r->fval = goal->fval; r->length = goal->length; r->state_no = goal->state_no; r->state = goal->state;
        r->node_id = goal->node_id; r->real_score = goal->real_score; r->score = goal->score;
    }
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
// =================================================================================================
// 6. denovo: tips, bubbles, unitigs (the reference run with ONE thread: its loops are order dependent)
// =================================================================================================
namespace denovo {

struct Bits {
    std::vector<uint64_t> v;
    void reset(int64_t n) { v.assign((size_t)((n + 63) / 64) + 1, 0); }
    bool get(int64_t i) const { return (v[i >> 6] >> (i & 63)) & 1; }
    void set(int64_t i) { v[i >> 6] |= 1ULL << (i & 63); }
    void unset(int64_t i) { v[i >> 6] &= ~(1ULL << (i & 63)); }
    bool try_lock(int64_t i) { if (get(i)) return false; set(i); return true; }   // atomic_bit_vector.h try_lock, one thread
};

inline void set_invalid(Graph &g, int64_t e) { g.invalid[e >> 6] |= 1ULL << (e & 63); }
inline void set_valid(Graph &g, int64_t e) { g.invalid[e >> 6] &= ~(1ULL << (e & 63)); }
inline int multiplicity(const Graph &g, int64_t e) { return 2 - (int)g.is_multi1(e); }   // succinct_dbg.h:133-135 (need_multiplicity = false)

int64_t last_index(const Graph &g, int64_t x) {   // GetLastIndex = rs_last_.Succ, succinct_dbg.h:105
    while (x < g.size && !g.is_last(x)) ++x;
    return x;
}

// the edges that point to the node / edge x (shared scan of IncomingEdges, UniquePrevEdge, UniquePrevNode, NodeIndegreeZero,
// DeleteAllEdges: succinct_dbg.cpp:99-127,196-225,268-315,348-371); `all` also reports invalid ones
int incoming_scan(const Graph &g, int64_t x, int64_t *in, bool all) {
    int64_t first = g_backward(g, x);
    int c = g.W(first), ones = g.last_or_tip(first), n = 0;
    if (all || g.valid(first)) in[n++] = first;
    for (int64_t y = first + 1; ones < 5 && y < g.size; ++y) {
        ones += g.last_or_tip(y);
        int cur = g.W(y);
        if (cur == c) break;
        if (cur == c + 4 && (all || g.valid(y))) in[n++] = y;
    }
    return n;
}
int node_edges(const Graph &g, int64_t node, int64_t *out, bool all) {   // the node's own edges, from its last one downwards
    int64_t e = last_index(g, node);
    int n = 0;
    do {
        if (all || g.valid(e)) out[n++] = e;
        --e;
    } while (e >= 0 && !g.last_or_tip(e));
    return n;
}
bool node_outdegree_zero(const Graph &g, int64_t node) { int64_t t[8]; return node_edges(g, node, t, false) == 0; }     // :227-240
bool node_indegree_zero(const Graph &g, int64_t node) { int64_t t[8]; return incoming_scan(g, node, t, false) == 0; }   // :242-266
int64_t unique_prev_node(const Graph &g, int64_t node) {                                                                // :268-292
    int64_t t[8];
    return incoming_scan(g, node, t, false) == 1 ? last_index(g, t[0]) : -1;
}
int64_t unique_next_node(const Graph &g, int64_t node) {                                                                // :294-315
    int64_t t[8];
    return node_edges(g, node, t, false) == 1 ? last_index(g, g_forward(g, t[0])) : -1;
}
void delete_all_edges(Graph &g, int64_t node) {                                                                         // :317-346
    int64_t t[8];
    int n = node_edges(g, node, t, true);
    for (int i = 0; i < n; ++i) set_invalid(g, t[i]);
    n = incoming_scan(g, node, t, true);
    for (int i = 0; i < n; ++i) set_invalid(g, t[i]);
}
int64_t unique_next_edge(const Graph &g, int64_t e) {                                                                   // :173-194
    int64_t o[8];
    if (!g.valid(e)) return -1;
    int n = 0;
    int64_t nx = g_forward(g, e);
    do {
        if (g.valid(nx)) o[n++] = nx;
        --nx;
    } while (nx >= 0 && !g.last_or_tip(nx));
    return n == 1 ? o[0] : -1;
}
int64_t unique_prev_edge(const Graph &g, int64_t e) {                                                                   // :196-225
    int64_t t[8];
    if (!g.valid(e)) return -1;
    return incoming_scan(g, e, t, false) == 1 ? t[0] : -1;
}
int64_t prev_simple(const Graph &g, int64_t e) { int64_t p = unique_prev_edge(g, e); return p != -1 && unique_next_edge(g, p) != -1 ? p : -1; }
int64_t next_simple(const Graph &g, int64_t e) { int64_t n = unique_next_edge(g, e); return n != -1 && unique_prev_edge(g, n) != -1 ? n : -1; }
int edge_outdegree(const Graph &g, int64_t e) { int64_t o[8]; return g_outgoing(g, e, o); }
int edge_indegree(const Graph &g, int64_t e) { int64_t o[8]; return g_incoming(g, e, o); }

int64_t edge_reverse_complement(const Graph &g, int64_t e) {   // succinct_dbg.cpp:551-593
    if (!g.valid(e)) return -1;
    std::vector<uint8_t> seq((size_t)g.k + 1);
    g_label(g, e, seq.data());
    seq[g.k] = (uint8_t)g.out_label(e);
    std::reverse(seq.begin(), seq.end());
    for (auto &c : seq) c = uint8_t(5 - c);
    return g_index_edge(g, seq.data());
}

// Trim, assembly_algorithms.cpp:76-159
int64_t trim(Graph &g, Bits &removed, int len) {
    int64_t n_tips = 0;
    for (int64_t x = 0; x < g.size; ++x) {
        if (!(g.is_last(x) && !removed.get(x) && node_outdegree_zero(g, x))) continue;
        std::vector<int64_t> path{x};
        int64_t cur = x;
        bool is_tip = false;
        for (int i = 1; i < len; ++i) {
            int64_t prev = unique_prev_node(g, cur);
            if (prev == -1) { is_tip = node_indegree_zero(g, cur); break; }
            if (unique_next_node(g, prev) == -1) { is_tip = true; break; }
            path.push_back(prev);
            cur = prev;
        }
        if (is_tip) { for (int64_t p : path) removed.set(p); ++n_tips; }
    }
    for (int64_t x = 0; x < g.size; ++x) {
        if (!(g.is_last(x) && !removed.get(x) && node_indegree_zero(g, x))) continue;
        std::vector<int64_t> path{x};
        int64_t cur = x;
        bool is_tip = false;
        for (int i = 1; i < len; ++i) {
            int64_t next = unique_next_node(g, cur);
            if (next == -1) { is_tip = node_outdegree_zero(g, cur); break; }
            if (unique_prev_node(g, next) == -1) { is_tip = true; break; }   // the reference keeps looping here; the outcome is the same
            path.push_back(next);
            cur = next;
        }
        if (is_tip) { for (int64_t p : path) removed.set(p); ++n_tips; }
    }
    for (int64_t x = 0; x < g.size; ++x)
        if (removed.get(x)) delete_all_edges(g, x);
    return n_tips;
}

int64_t remove_tips(Graph &g, int max_tip_len) {   // :161-183
    Bits removed;
    removed.reset(g.size);
    int64_t n = 0;
    for (int len = 2; len < max_tip_len; len *= 2) n += trim(g, removed, len);
    n += trim(g, removed, max_tip_len);
    return n;
}

struct BranchGroup {   // branch_group.cpp:22-141
    Graph &g;
    int64_t begin, end = -1;
    int max_branches, max_length;
    std::vector<std::vector<int64_t>> branches;
    std::vector<int> mult;
    bool search() {
        if (!g.valid(begin)) return false;
        int outd = edge_outdegree(g, begin);
        if (outd <= 1 || outd > max_branches) return false;
        branches.push_back({begin});
        mult.push_back(0);
        bool converged = false;
        for (int j = 1; j < max_length; ++j) {
            int nb = (int)branches.size();
            for (int i = 0; i < nb; ++i) {
                int64_t out[8];
                int od = g_outgoing(g, branches[i].back(), out);
                if (od < 1) return false;       // a dead branch can never converge (the reference runs on and fails later)
                branches[i].push_back(out[0]);
                mult[i] += multiplicity(g, out[0]);
                if ((int)branches.size() + od - 1 > max_branches) return false;
                std::vector<int64_t> copy = branches[i];
                int base = mult[i] - multiplicity(g, out[0]);
                for (int x = 1; x < od; ++x) {
                    copy.back() = out[x];
                    branches.push_back(copy);
                    mult.push_back(base + multiplicity(g, out[x]));
                }
            }
            for (auto &b : branches) {
                int64_t in[8];
                int id = g_incoming(g, b.back(), in);
                if (id == 1) continue;
                for (int x = 0; x < id; ++x) {
                    bool found = false;
                    for (auto &o : branches)
                        if (o[j - 1] == in[x]) { found = true; break; }
                    if (!found) return false;
                }
            }
            end = branches[0].back();
            if (edge_outdegree(g, end) == 1) {
                converged = true;
                for (auto &b : branches)
                    if (b.back() != end) { converged = false; break; }
                if (converged) break;
            }
        }
        return converged && begin != end;
    }
    bool pop(Bits &marked) {
        int best = 0, best_m = mult[0];
        for (size_t i = 1; i < branches.size(); ++i)
            if (mult[i] >= best_m) { best = (int)i; best_m = mult[i]; }
        std::vector<int64_t> locked;
        for (auto &b : branches)
            for (size_t j = 1; j + 1 < b.size(); ++j) {
                if (!marked.try_lock(b[j])) {
                    for (int64_t e : locked) { marked.unset(e); set_valid(g, e); }
                    return false;
                }
                locked.push_back(b[j]);
                set_invalid(g, b[j]);
            }
        auto &w = branches[best];
        for (size_t j = 1; j + 1 < w.size(); ++j) { set_valid(g, w[j]); marked.unset(w[j]); }
        return true;
    }
};

int64_t pop_bubbles(Graph &g) {   // assembly_algorithms.cpp:245-301
    const int max_len = g.k * 2 + 4;
    std::vector<int64_t> cand, again;
    Bits marked;
    marked.reset(g.size);
    for (int64_t e = 0; e < g.size; ++e) {
        if (!g.valid(e)) continue;
        BranchGroup b{g, e, -1, 16, max_len};
        if (b.search()) cand.push_back(e);
    }
    int64_t popped = 0;
    for (int64_t e : cand) {
        BranchGroup b{g, e, -1, 16, max_len};
        if (b.search()) { if (b.pop(marked)) ++popped; else again.push_back(e); }
    }
    for (int64_t e : again) {
        BranchGroup b{g, e, -1, 16, max_len};
        if (b.search() && b.pop(marked)) ++popped;
    }
    return popped;
}

struct Contig { int flag; double multi; std::string seq; };

std::string path_label(const Graph &g, int64_t start, int64_t end, int length) {   // VertexToDNAString, unitig_graph.cpp:80-112
    std::string s;
    int64_t cur = end;
    for (int i = 1; i < length; ++i) {
        s.push_back("ACGT"[g.out_label(cur) - 1]);
        cur = prev_simple(g, cur);
    }
    s.push_back("ACGT"[g.out_label(cur) - 1]);
    std::vector<uint8_t> lab((size_t)g.k);
    g_label(g, start, lab.data());
    for (int i = g.k - 1; i >= 0; --i) s.push_back("ACGT"[lab[i] - 1]);
    std::reverse(s.begin(), s.end());
    return s;
}

std::vector<Contig> unitigs(const Graph &g, int min_contig, bool verbose = false) {   // UnitigGraph::InitFromSdBG with a file, unitig_graph.cpp:208-303
    std::vector<Contig> out;
    Bits marked;
    marked.reset(g.size);
    for (int64_t e = 0; e < g.size; ++e) {
        if (!(g.valid(e) && next_simple(g, e) == -1 && marked.try_lock(e))) continue;
        bool add = true;
        int64_t cur = e, prev;
        int64_t depth = multiplicity(g, e);
        int64_t length = 1;
        while ((prev = prev_simple(g, cur)) != -1) {
            cur = prev;
            if (!marked.try_lock(cur)) { add = false; break; }
            depth += multiplicity(g, cur);
            ++length;
        }
        if (!add) continue;
        int64_t rc_start = edge_reverse_complement(g, e), rc_end = -1;
        if (!marked.try_lock(rc_start)) {
            rc_end = edge_reverse_complement(g, cur);
            if (std::max(e, cur) < std::max(rc_start, rc_end)) add = false;
        } else {
            int64_t rc = rc_start;
            while ((rc = next_simple(g, rc)) != -1)
                if (!marked.try_lock(rc)) break;
        }
        if (!add) continue;
        std::string label = path_label(g, cur, e, (int)length);
        if ((int)label.size() < min_contig) continue;
        if (verbose) fprintf(stderr, "seq end=%lld start=%lld len=%lld rc=%lld\n", (long long)e, (long long)cur, (long long)length, (long long)rc_start);
        int flag = (edge_indegree(g, cur) == 0 && edge_outdegree(g, e) == 0) ? 1 : 0;     // contig_flag::kIsolated
        double multi = std::min(65535.0, (double)depth / (double)length);
        std::string rc(label.rbegin(), label.rend());
        for (auto &c : rc) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
        out.push_back(Contig{flag, multi, label < rc ? label : rc});
    }
    return out;
}

// The data-parallel formulation the device uses for the unitig step (denovo.hip), restated on the host so that it can be checked against
// the sequential loop above without a GPU: tests only.
std::vector<Contig> unitigs_claim_model(const Graph &g, int min_contig, bool verbose) {
    struct Rec { int64_t end, start, rc_start, depth; int64_t length; int target; int64_t dist; };
    std::vector<int64_t> ends;
    for (int64_t e = 0; e < g.size; ++e)
        if (g.valid(e) && next_simple(g, e) == -1) ends.push_back(e);
    int n = (int)ends.size();
    std::vector<Rec> rec((size_t)n);
    std::vector<std::vector<int>> claims((size_t)n);
    for (int p = 0; p < n; ++p) {
        Rec &r = rec[(size_t)p];
        r.end = ends[(size_t)p];
        int64_t cur = r.end, prev;
        r.depth = multiplicity(g, r.end); r.length = 1;
        while ((prev = prev_simple(g, cur)) != -1) { cur = prev; r.depth += multiplicity(g, cur); ++r.length; }
        r.start = cur;
        r.rc_start = edge_reverse_complement(g, r.end);
        r.target = -1; r.dist = 0;
        if (r.rc_start >= 0 && g.valid(r.rc_start)) {
            int64_t x = r.rc_start, nx, d = 0;
            bool cyc = false;
            while ((nx = next_simple(g, x)) != -1) { x = nx; ++d; if (x == r.rc_start) { cyc = true; break; } }
            if (!cyc) {
                auto it = std::lower_bound(ends.begin(), ends.end(), x);
                if (it != ends.end() && *it == x) { r.target = int(it - ends.begin()); r.dist = d; }
            }
        }
        if (r.target >= 0) claims[(size_t)r.target].push_back(p);
    }
    if (verbose) {
        fprintf(stderr, "model: %d paths\n", n);
        for (int p = 0; p < n; ++p) fprintf(stderr, "mrec p=%d end=%lld start=%lld len=%lld rc=%lld target=%d dist=%lld\n", p, (long long)rec[(size_t)p].end,
                                            (long long)rec[(size_t)p].start, (long long)rec[(size_t)p].length, (long long)rec[(size_t)p].rc_start, rec[(size_t)p].target, (long long)rec[(size_t)p].dist);
    }
    std::vector<int> state((size_t)n, 0);
    for (bool again = true; again;) {
        again = false;
        for (int p = 0; p < n; ++p) {
            if (state[(size_t)p]) continue;
            bool pending = false, skipped = false;
            for (int q : claims[(size_t)p]) {
                if (q == p || rec[(size_t)q].end >= rec[(size_t)p].end) continue;
                if (state[(size_t)q] == 1) { skipped = true; break; }
                if (state[(size_t)q] == 0) pending = true;
            }
            if (skipped) state[(size_t)p] = 2; else if (!pending) state[(size_t)p] = 1; else again = true;
        }
    }
    std::vector<Contig> out;
    for (int p = 0; p < n; ++p) {
        if (state[(size_t)p] != 1) continue;
        const Rec &r = rec[(size_t)p];
        bool add = true;
        if (r.target >= 0) {
            int t = r.target;
            bool locked = t == p || (state[(size_t)t] == 1 && rec[(size_t)t].end < r.end);
            for (int q : claims[(size_t)t])
                if (q != p && state[(size_t)q] == 1 && rec[(size_t)q].end < r.end && rec[(size_t)q].dist > r.dist) locked = true;
            if (locked) {
                int64_t rc_end = edge_reverse_complement(g, r.start);
                if (std::max(r.end, r.start) < std::max(r.rc_start, rc_end)) add = false;
            }
        }
        if (!add) continue;
        std::string label = path_label(g, r.start, r.end, (int)r.length);
        if ((int)label.size() < min_contig) continue;
        int flag = (edge_indegree(g, r.start) == 0 && edge_outdegree(g, r.end) == 0) ? 1 : 0;
        std::string rc(label.rbegin(), label.rend());
        for (auto &c : rc) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
        if (verbose) fprintf(stderr, "model p=%d end=%lld start=%lld len=%lld rc=%lld target=%d dist=%lld\n", p, (long long)r.end, (long long)r.start,
                             (long long)r.length, (long long)r.rc_start, r.target, (long long)r.dist);
        out.push_back(Contig{flag, std::min(65535.0, (double)r.depth / (double)r.length), label < rc ? label : rc});
    }
    return out;
}

}  // namespace denovo


struct orc_stream : Stream {};
struct orc_graph : Graph {};
struct orc_hmm : Hmm {};
struct orc_searcher : Searcher {};

extern "C" {

orc_stream *orc_sdbg_build(const uint32_t *packed, uint64_t n_words, const uint64_t *start_idx, uint64_t n_reads, int k,
                           int n_threads) {
    (void)n_words;
    if (k < 9 || k > 127) return nullptr;
    int W = (2 * k + 4 + 31) / 32;                                 // words_per_substring, s2.cpp:331
    Stream *s = nullptr;
    switch (W) {
    case 1: s = build_stream<1>(packed, start_idx, n_reads, k, n_threads); break;
    case 2: s = build_stream<2>(packed, start_idx, n_reads, k, n_threads); break;
    case 3: s = build_stream<3>(packed, start_idx, n_reads, k, n_threads); break;
    case 4: s = build_stream<4>(packed, start_idx, n_reads, k, n_threads); break;
    case 5: s = build_stream<5>(packed, start_idx, n_reads, k, n_threads); break;
    case 6: s = build_stream<6>(packed, start_idx, n_reads, k, n_threads); break;
    case 7: s = build_stream<7>(packed, start_idx, n_reads, k, n_threads); break;
    case 8: s = build_stream<8>(packed, start_idx, n_reads, k, n_threads); break;
    default: s = build_stream<9>(packed, start_idx, n_reads, k, n_threads); break;
    }
    return static_cast<orc_stream *>(s);
}

orc_stream *orc_sdbg_build_solid(const uint32_t *packed, uint64_t n_words, const uint64_t *start_idx, uint64_t n_reads, uint64_t n_short,
                                 int k, int min_count, int need_mercy, int n_threads, int64_t *counting, int64_t *n_mercy) {
    (void)n_words;
    if (k < 9 || k > 127 || min_count < 1) return nullptr;
    if (n_short > n_reads) n_short = n_reads;
    if (min_count == 1) {                                           // build_graph.cpp:96-118: stage 1 is skipped
        if (counting) std::fill(counting, counting + 65536, 0);
        if (n_mercy) *n_mercy = 0;
        return static_cast<orc_stream *>(build_any(packed, start_idx, n_reads, k, n_threads, nullptr));
    }
    Stage1Out s1 = stage1(packed, start_idx, n_reads, n_short, k, min_count, need_mercy != 0);
    if (counting) std::copy(s1.counting.begin(), s1.counting.end(), counting);
    if (n_mercy) *n_mercy = s1.n_mercy;
    return static_cast<orc_stream *>(build_any(packed, start_idx, n_reads, k, n_threads, &s1.solid));
}

orc_stream *orc_sdbg_read(const char *prefix) {                    // SdbgReader, sdbg_multi_io.h:240-382
    std::string p(prefix);
    FILE *info = fopen((p + ".sdbg_info").c_str(), "r");
    if (!info) return nullptr;
    auto *s = new orc_stream;
    int nb = 0, nf = 0;
    long long total = 0, ntips = 0, nlarge = 0;
    bool ok = fscanf(info, "k %d\n", &s->k) == 1 && fscanf(info, "words_per_tip_label %d\n", &s->words_per_tip) == 1 &&
              fscanf(info, "num_buckets %d\n", &nb) == 1 && fscanf(info, "num_threads %d\n", &nf) == 1 &&
              fscanf(info, "total_size %lld\n", &total) == 1 && fscanf(info, "num_tips %lld\n", &ntips) == 1 &&
              fscanf(info, "large_multi %lld\n", &nlarge) == 1 && nb == kBuckets;
    if (!ok) { fclose(info); delete s; return nullptr; }
    struct Rec { int tid; long long off, items, tips, large; };
    std::vector<Rec> recs(nb);
    for (int b = 0; b < nb; ++b) {
        int dummy;
        if (fscanf(info, "%d %d %lld %lld %lld %lld\n", &dummy, &recs[b].tid, &recs[b].off, &recs[b].items, &recs[b].tips,
                   &recs[b].large) != 6) { fclose(info); delete s; return nullptr; }
    }
    fclose(info);
    std::vector<std::vector<unsigned char>> files(nf);
    for (int t = 0; t < nf; ++t) {
        std::ifstream f(p + ".sdbg." + std::to_string(t), std::ios::binary);
        if (!f.is_open()) { delete s; return nullptr; }
        files[t].assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    }
    for (int b = 0; b < nb; ++b) {
        s->bucket_items[b] = recs[b].items;
        if (recs[b].tid < 0) continue;
        const unsigned char *ptr = files[recs[b].tid].data() + recs[b].off;
        for (long long i = 0; i < recs[b].items; ++i) {
            uint16_t it;
            memcpy(&it, ptr, 2); ptr += 2;
            s->records.push_back(it);
            if ((it >> 8) == kMulti2Sp) { uint16_t m; memcpy(&m, ptr, 2); ptr += 2; s->large.push_back(m); }   // :367-371
            if ((it >> 5) & 1) {
                for (int t = 0; t < s->words_per_tip; ++t) { uint32_t w; memcpy(&w, ptr, 4); ptr += 4; s->tips.push_back(w); }
            }
        }
    }
    if ((long long)s->records.size() != total) { delete s; return nullptr; }
    return s;
}

void orc_stream_free(orc_stream *s) { delete s; }
int orc_stream_k(const orc_stream *s) { return s->k; }
int orc_stream_words_per_tip(const orc_stream *s) { return s->words_per_tip; }
int64_t orc_stream_num_edges(const orc_stream *s) { return (int64_t)s->records.size(); }
int64_t orc_stream_num_tips(const orc_stream *s) { return s->words_per_tip ? (int64_t)s->tips.size() / s->words_per_tip : 0; }
int64_t orc_stream_num_large(const orc_stream *s) { return (int64_t)s->large.size(); }
int64_t orc_stream_num_items_sorted(const orc_stream *s) { return s->n_items_sorted; }
const int64_t *orc_stream_bucket_items(const orc_stream *s) { return s->bucket_items.data(); }
const uint16_t *orc_stream_records(const orc_stream *s) { return s->records.data(); }
const uint16_t *orc_stream_large(const orc_stream *s) { return s->large.data(); }
const uint32_t *orc_stream_tips(const orc_stream *s) { return s->tips.data(); }

orc_graph *orc_graph_from_stream(const orc_stream *s) { return static_cast<orc_graph *>(graph_from_stream(*s)); }
void orc_graph_free(orc_graph *g) { delete g; }
int64_t orc_graph_size(const orc_graph *g) { return g->size; }
int orc_graph_k(const orc_graph *g) { return g->k; }
const int64_t *orc_graph_f(const orc_graph *g) { return g->f; }
const uint64_t *orc_graph_w(const orc_graph *g) { return g->w.data(); }
const uint64_t *orc_graph_last(const orc_graph *g) { return g->last.data(); }
const uint64_t *orc_graph_tip(const orc_graph *g) { return g->tip.data(); }
const uint64_t *orc_graph_invalid(const orc_graph *g) { return g->invalid.data(); }
const uint64_t *orc_graph_multi1(const orc_graph *g) { return g->multi1.data(); }
const uint32_t *orc_graph_tip_labels(const orc_graph *g) { return g->tip_labels.data(); }
int64_t orc_graph_num_tips(const orc_graph *g) { return g->num_tips; }
int64_t orc_rank_last(const orc_graph *g, int64_t pos) { return g_rank_last(*g, pos); }
int64_t orc_select_last(const orc_graph *g, int64_t r) { return g_select_last(*g, r); }
int64_t orc_rank_w(const orc_graph *g, int c, int64_t pos) { return g_rank_w(*g, c, pos); }
int64_t orc_select_w(const orc_graph *g, int c, int64_t r) { return g_select_w(*g, c, r); }
int64_t orc_forward(const orc_graph *g, int64_t e) { return g_forward(*g, e); }
int64_t orc_backward(const orc_graph *g, int64_t e) { return g_backward(*g, e); }
int orc_outgoing(const orc_graph *g, int64_t e, int64_t out[4]) { return g_outgoing(*g, e, out); }
int orc_incoming(const orc_graph *g, int64_t e, int64_t in[4]) { return g_incoming(*g, e, in); }
int orc_label(const orc_graph *g, int64_t e, uint8_t *seq) { return g_label(*g, e, seq); }
int64_t orc_index_edge(const orc_graph *g, const uint8_t *seq) { return g_index_edge(*g, seq); }

orc_hmm *orc_hmm_parse(const char *path) { return static_cast<orc_hmm *>(parse_hmm(path)); }
void orc_hmm_free(orc_hmm *h) { delete h; }
int orc_hmm_M(const orc_hmm *h) { return h->M; }
int orc_hmm_A(const orc_hmm *h) { return h->A; }
const double *orc_hmm_msc(const orc_hmm *h) { return h->msc.data(); }
const double *orc_hmm_isc(const orc_hmm *h) { return h->isc.data(); }
const double *orc_hmm_tsc(const orc_hmm *h) { return h->tsc.data(); }
const double *orc_hmm_maxm(const orc_hmm *h) { return h->maxm.data(); }
const double *orc_hmm_h(const orc_hmm *h) { return h->h.data(); }
const int *orc_hmm_alpha(const orc_hmm *h) { return h->alpha; }

orc_searcher *orc_searcher_new(const orc_graph *g, const orc_hmm *fwd, const orc_hmm *rev, int prune_len, double low_cov_pen) {
    auto *s = new orc_searcher;
    s->g = g; s->hm[0] = fwd; s->hm[1] = rev;
    s->prune = prune_len;
    s->low_cov_penalty = -std::log(low_cov_pen);
    for (int i = 0; i < 3000; ++i) s->exit_prob[i] = std::log(2.0 / (i + 2)) * 2;
    return s;
}
void orc_searcher_free(orc_searcher *s) { s->release(); delete s; }
void orc_searcher_clear_cache(orc_searcher *s) { s->cache[0].clear(); s->cache[1].clear(); s->pending.clear(); s->seed_counter = 0; }
void orc_searcher_set_window(orc_searcher *s, int window) { s->window = window < 1 ? 1 : window; }
void orc_searcher_set_cost_rate(orc_searcher *s, int rate) { s->cost_rate = rate; s->cost_knee = 0; s->cost_rate2 = 0; }   // < 0: the cost term is c * |rate| (see the header)
void orc_searcher_set_cost_curve(orc_searcher *s, int rate, int64_t knee, int rate2) { s->cost_rate = rate; s->cost_knee = knee; s->cost_rate2 = knee ? rate2 : 0; }

int64_t orc_search_seed(orc_searcher *s, const char *kmer_c, int start_state, orc_astar_result *right, orc_astar_result *left,
                        char *contig, int64_t cap) {
    std::string kmer(kmer_c);
    for (auto &c : kmer) c = (char)tolower(c);                              // search.cpp:156
    if ((int)kmer.size() < s->g->k + 1) return -1;
    {   // make visible what this seed may see; of two entries for one key the one that became visible first stays (ties: the smaller
        // child descriptor, as the device's single atomicMax decides it)
        std::vector<Searcher::Pending> now;
        size_t keep = 0;
        for (size_t i = 0; i < s->pending.size(); ++i) {
            const auto &p = s->pending[i];
            if (p.visible_from <= s->seed_counter) now.push_back(p);
            else s->pending[keep++] = p;
        }
        s->pending.resize(keep);
        std::stable_sort(now.begin(), now.end(), [](const Searcher::Pending &a, const Searcher::Pending &b) {
            return a.visible_from != b.visible_from ? a.visible_from < b.visible_from : a.em < b.em; });
        for (const auto &p : now) s->cache[p.dir].emplace(p.parent, p.child);
    }
    AstarOut o1, o2;
    Node *g1 = astar_from_kmer(*s, 0, start_state, kmer, o1);               // hmm_graph_search.h:67
    std::string rs = g1 ? path_string(*s, 0, g1) : std::string();
    fill_result(right, o1, g1);
    int lstate = s->hm[1]->M - start_state - int(kmer.size() / 3);          // :73
    Node *g2 = astar_from_kmer(*s, 1, lstate, kmer, o2);
    std::string ls = g2 ? path_string(*s, 1, g2) : std::string();
    fill_result(left, o2, g2);
    for (auto &p : s->pending)
        if (p.visible_from < 0) {
            const int64_t c = p.dir == 0 ? o1.expanded : o2.expanded;
            const int64_t cost = s->cost_rate > 0 ? (s->cost_knee > 0 && c > s->cost_knee ? s->cost_knee / s->cost_rate + (c - s->cost_knee) / s->cost_rate2 : c / s->cost_rate)
                                                  : s->cost_rate < 0 ? c * (int64_t)(-s->cost_rate) : 0;
            p.visible_from = p.seed + s->window + cost;
        }
    s->release();
    s->seed_counter++;
    std::string out = revcomp(ls) + kmer + rs;                              // :77-79
    if ((int64_t)out.size() + 1 > cap) return -2;
    memcpy(contig, out.c_str(), out.size() + 1);
    return (int64_t)out.size();
}


// MODEL of speculative execution of the SEQUENTIAL sharing rule (window 1 = `search ... 1`): search i runs against the paths committed by
// the seeds <= i - lag (what a device with `lag` searches in flight would show it), logs the keys it looked up and MISSED, and is valid
// iff none of those keys has been committed by a seed in (i - lag, i) -- then its run is, look-up by look-up, the sequential one.  An
// invalid search runs again against everything below it.  Counts what the scheme costs; the committed paths are always the sequential ones.
// stats[0] searches, [1] without a missed look-up (never invalid), [2] invalid, [3] expansions of the speculative runs, [4] of the sequential
// runs, [5] of the re-runs (= sequential runs of the invalid ones), [6] largest re-run, [7] speculative result differs from the sequential one,
// [8] expansions of the invalid speculative runs (thrown away)
int64_t orc_spec_model(orc_searcher *s, const char *kmers, const int32_t *start_state, int64_t n, int64_t lag, int64_t *stats) {
    const int klen = s->g->k + 1;
    struct Ent { Key child; int64_t owner; };
    std::unordered_map<Key, Ent, KeyHash> tab[2];
    for (int q = 0; q < 9; ++q) stats[q] = 0;
    for (int64_t i = 0; i < n; ++i) {
        std::string kmer(kmers + i * klen, kmers + (i + 1) * klen);
        for (auto &c : kmer) c = (char)tolower(c);
        for (int dir = 0; dir < 2; ++dir) {
            const int st0 = dir == 0 ? start_state[i] : s->hm[1]->M - start_state[i] - klen / 3;
            std::vector<Key> missed;
            AstarOut o;
            Node *g = astar_with(*s, *s->hm[dir], start_from_kmer(*s, dir, st0, kmer), dir == 0, [&](const Key &k) -> const Key * {
                auto it = tab[dir].find(k);
                if (it != tab[dir].end() && it->second.owner <= i - lag) return &it->second.child;
                missed.push_back(k);
                return nullptr;
            }, o);
            bool invalid = false;
            for (const Key &k : missed) if (tab[dir].count(k)) { invalid = true; break; }
            std::vector<std::pair<Key, Key>> path;
            for (Node *p = g; p && p->from; p = p->from) path.push_back({key_of(*p->from), key_of(*p)});
            stats[0]++; stats[3] += o.expanded;
            if (missed.empty()) stats[1]++;
            if (invalid) {
                stats[2]++; stats[8] += o.expanded;
                s->release();
                AstarOut o2;
                Node *g2 = astar_with(*s, *s->hm[dir], start_from_kmer(*s, dir, st0, kmer), dir == 0, [&](const Key &k) -> const Key * {
                    auto it = tab[dir].find(k);
                    return it == tab[dir].end() ? nullptr : &it->second.child;
                }, o2);
                std::vector<std::pair<Key, Key>> path2;
                for (Node *p = g2; p && p->from; p = p->from) path2.push_back({key_of(*p->from), key_of(*p)});
                if (!(path2.size() == path.size() && std::equal(path.begin(), path.end(), path2.begin(), [](const std::pair<Key, Key> &a, const std::pair<Key, Key> &b) {
                        return a.first == b.first && a.second == b.second; }))) stats[7]++;
                path.swap(path2);
                stats[4] += o2.expanded; stats[5] += o2.expanded;
                stats[6] = std::max<int64_t>(stats[6], o2.expanded);
            } else {
                stats[4] += o.expanded;
            }
            for (auto &pc : path) tab[dir].emplace(pc.first, Ent{pc.second, i});
            s->release();
        }
    }
    return 0;
}

// main_assemble (assembler.cpp:98-167) on a loaded graph; the graph's validity bits are consumed.  Returns the FASTA text
// (">k{K}_{id} flag={f} multi={%.4lf} len={L}\n{seq}\n", unitig_graph.cpp:134-150), malloc'd.
char *orc_denovo(orc_graph *g, int max_tip_len, int no_bubble, int min_contig, int64_t *n_contigs, int64_t *total_len, int64_t *n_tips,
                 int64_t *n_bubbles) {
    if (max_tip_len == -1) max_tip_len = g->k * 2;
    int64_t tips = 0, bub = 0;
    if (max_tip_len > 0) tips = denovo::remove_tips(*g, max_tip_len);
    if (!no_bubble) bub = denovo::pop_bubbles(*g);
    auto cs = denovo::unitigs(*g, min_contig);
    std::string text;
    int64_t total = 0, id = 0;
    char head[128];
    for (auto &c : cs) {
        ++id;
        snprintf(head, sizeof head, ">k%d_%lld flag=%d multi=%.4lf len=%d\n", g->k, (long long)id, c.flag, c.multi, (int)c.seq.size());
        text += head;
        text += c.seq;
        text += '\n';
        total += (int64_t)c.seq.size();
    }
    if (n_contigs) *n_contigs = (int64_t)cs.size();
    if (total_len) *total_len = total;
    if (n_tips) *n_tips = tips;
    if (n_bubbles) *n_bubbles = bub;
    char *r = (char *)malloc(text.size() + 1);
    memcpy(r, text.c_str(), text.size() + 1);
    return r;
}
// tests: the device's claim formulation of the unitig step vs the sequential loop on the graph as it is now; returns the number of
// differing contigs (0 = identical lists)
int64_t orc_denovo_unitig_model_check(orc_graph *g, int min_contig, int verbose) {
    auto a = denovo::unitigs(*g, min_contig, verbose != 0), b = denovo::unitigs_claim_model(*g, min_contig, verbose != 0);
    int64_t diff = (int64_t)a.size() > (int64_t)b.size() ? (int64_t)(a.size() - b.size()) : (int64_t)(b.size() - a.size());
    for (size_t i = 0; i < std::min(a.size(), b.size()); ++i) diff += a[i].seq != b[i].seq || a[i].flag != b[i].flag || a[i].multi != b[i].multi;
    return diff;
}
int64_t orc_denovo_remove_tips(orc_graph *g, int max_tip_len) { return denovo::remove_tips(*g, max_tip_len == -1 ? g->k * 2 : max_tip_len); }
int64_t orc_denovo_pop_bubbles(orc_graph *g) { return denovo::pop_bubbles(*g); }
void orc_free(void *p) { free(p); }
const uint64_t *orc_graph_invalid_now(const orc_graph *g) { return g->invalid.data(); }

}  // extern "C"
