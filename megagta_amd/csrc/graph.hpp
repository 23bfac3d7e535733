// graph.hpp — device-resident succinct de Bruijn graph.
//
// The reference keeps five bit-vectors plus three separate two-level rank/select directories
// (rank_and_select.h:80-124,430-500) and answers one OutgoingEdges with ~10 dependent cache misses.
// Here everything about 64 consecutive edges lives in ONE 128-byte line (= one gfx950 L2 line):
// W symbols, last / tip / invalid / multi1 bits and the absolute rank counters needed by Forward /
// Backward, so Rank is a single line fetch and Select is a sample lookup + (usually) one line.
// Only the ANSWERS follow the reference: Rank(c,pos) = #c in [0..pos]  (rank_and_select.h:153),
// Select(r) = position of the r-th one, 0-based (:560), Forward / Backward / OutgoingEdges
// (succinct_dbg.h:155-170, succinct_dbg.cpp:78-97), IndexBinarySearch[Edge] (:427-549).
#pragma once
#include "common.hpp"

namespace mgta {

struct alignas(128) GLine {
    uint64_t w[4];        // 64 x 4-bit W (edge j: bits 4*(j&15) of w[j>>4]); 0=$ 1-4=ACGT 5-8=ACGT-minus
    uint64_t last;        // succinct_dbg.h:97
    uint64_t tip;         // is_tip_
    uint64_t invalid;     // tip | (W == 0)    succinct_dbg.cpp:717-721, .h:81-85
    uint64_t multi1;      // stored multiplicity <= 1   succinct_dbg.cpp:680
    uint64_t rank_last;   // ones of `last` before this line
    uint64_t rank_tip;    // tips before this line
    uint64_t rank_w[4];   // occurrences of W == a (a = 1..4, plain symbols only) before this line
    uint32_t fwd_hint[4]; // line where Forward of the last W == a edge BEFORE this line lands (a = 1..4): Forward(e) of any edge of
                          // this line (also an a + 4 edge that precedes the line's first plain a) needs no select-sample
                          // lookup, it starts from this hint and advances at most a line or two
};
static_assert(sizeof(GLine) == 128, "one line per 64 edges");

struct GraphDev {
    const GLine *lines;
    uint64_t n_lines;
    int64_t size;
    int k, words_per_tip;
    int64_t f[6];          // sdbg_multi_io.h:254-268
    int64_t rank_f[6];     // rs_last_.Rank(f[i]-1), succinct_dbg.h:74-76
    int64_t total_last;
    int64_t total_w[5];    // [1..4]
    const uint32_t *sel_last;     // line holding the (64*j)-th one of `last`
    const uint32_t *sel_w[5];     // [1..4]: line holding the (64*j)-th occurrence of symbol a
    const uint32_t *tip_labels;
};

// ---- device-side navigation -------------------------------------------------------------------
__device__ __forceinline__ int g_W(const GraphDev &g, int64_t x) {
    return (int)((g.lines[x >> 6].w[(x >> 4) & 3] >> ((x & 15) * 4)) & 15);
}
__device__ __forceinline__ int g_bit(uint64_t word, int64_t x) { return (int)((word >> (x & 63)) & 1); }
__device__ __forceinline__ bool g_valid(const GraphDev &g, int64_t x) { return !g_bit(g.lines[x >> 6].invalid, x); }
__device__ __forceinline__ bool g_last(const GraphDev &g, int64_t x) { return g_bit(g.lines[x >> 6].last, x); }
__device__ __forceinline__ bool g_tip(const GraphDev &g, int64_t x) { return g_bit(g.lines[x >> 6].tip, x); }
__device__ __forceinline__ bool g_last_or_tip(const GraphDev &g, int64_t x) {
    const GLine &L = g.lines[x >> 6];
    return g_bit(L.last | L.tip, x);
}
__device__ __forceinline__ bool g_multi1(const GraphDev &g, int64_t x) { return g_bit(g.lines[x >> 6].multi1, x); }

// nibbles of `word` equal to c -> one bit (lowest of the nibble) each
__device__ __forceinline__ uint64_t nib_eq(uint64_t word, int c) {
    uint64_t t = word ^ (0x1111111111111111ull * (uint64_t)c);
    t |= t >> 1;
    t |= t >> 2;
    return ~t & 0x1111111111111111ull;
}

// number of W == c (c in 1..4) in [0..pos]   (RankAndSelect4Bits::Rank semantics)
__device__ __forceinline__ int64_t g_rank_w(const GraphDev &g, int c, int64_t pos) {
    if (pos < 0) return 0;
    if (pos >= g.size - 1) return g.total_w[c];
    const GLine &L = g.lines[pos >> 6];
    int j = (int)(pos & 63);
    int64_t r = (int64_t)L.rank_w[c - 1];
    int fw = j >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < fw) r += __popcll(nib_eq(L.w[q], c));
        else if (q == fw) {
            int nb = (j & 15) + 1;
            uint64_t m = nb == 16 ? ~0ull : ((1ull << (4 * nb)) - 1);
            r += __popcll(nib_eq(L.w[q], c) & m);
        }
    }
    return r;
}

__device__ __forceinline__ int64_t g_rank_last(const GraphDev &g, int64_t pos) {   // ones in [0..pos]
    if (pos < 0) return 0;
    if (pos >= g.size - 1) return g.total_last;
    const GLine &L = g.lines[pos >> 6];
    int j = (int)(pos & 63);
    uint64_t m = j == 63 ? ~0ull : ((1ull << (j + 1)) - 1);
    return (int64_t)L.rank_last + __popcll(L.last & m);
}

__device__ __forceinline__ int select64(uint64_t x, int n) {   // position of the n-th (0-based) set bit
    int pos = 0;
    int c = __popc((uint32_t)x);
    if (n >= c) { n -= c; x >>= 32; pos += 32; }
    c = __popc((uint32_t)x & 0xFFFFu);
    if (n >= c) { n -= c; x >>= 16; pos += 16; }
    c = __popc((uint32_t)x & 0xFFu);
    if (n >= c) { n -= c; x >>= 8; pos += 8; }
    c = __popc((uint32_t)x & 0xFu);
    if (n >= c) { n -= c; x >>= 4; pos += 4; }
    c = __popc((uint32_t)x & 0x3u);
    if (n >= c) { n -= c; x >>= 2; pos += 2; }
    c = (int)(x & 1);
    if (n >= c) pos += 1;
    return pos;
}

__device__ __forceinline__ int64_t g_select_last(const GraphDev &g, int64_t r) {   // RankAndSelect1Bit::Select
    if (r >= g.total_last) return g.size;
    if (r < 0) return -1;
    uint64_t li = g.sel_last[r >> 6];
    // (the ones before the next line = the ones before this line + the ones in it: the next line is only touched when the target is there)
    while (li + 1 < g.n_lines && (int64_t)(g.lines[li].rank_last + (uint64_t)__popcll(g.lines[li].last)) <= r) ++li;
    const GLine &L = g.lines[li];
    return (int64_t)(li << 6) + select64(L.last, (int)(r - (int64_t)L.rank_last));
}

__device__ __forceinline__ int64_t g_select_w(const GraphDev &g, int c, int64_t r) {   // RankAndSelect4Bits::Select, c in 1..4
    if (r >= g.total_w[c]) return g.size;
    if (r < 0) return -1;
    uint64_t li = g.sel_w[c][r >> 6];
    for (; li + 1 < g.n_lines; ++li) {                                   // (the next line is only touched when the target is there)
        const GLine &C = g.lines[li];
        const int here = __popcll(nib_eq(C.w[0], c)) + __popcll(nib_eq(C.w[1], c)) + __popcll(nib_eq(C.w[2], c)) + __popcll(nib_eq(C.w[3], c));
        if ((int64_t)C.rank_w[c - 1] + here > r) break;
    }
    const GLine &L = g.lines[li];
    int rem = (int)(r - (int64_t)L.rank_w[c - 1]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint64_t e = nib_eq(L.w[q], c);
        int pc = __popcll(e);
        if (rem < pc) return (int64_t)(li << 6) + q * 16 + (select64(e, rem) >> 2);
        rem -= pc;
    }
    return g.size;
}

__device__ __forceinline__ int64_t g_forward(const GraphDev &g, int64_t e) {   // succinct_dbg.h:155-164
    int a = g_W(g, e);
    if (a > 4) a -= 4;
    int64_t cnt = g_rank_w(g, a, e);
    int64_t r = g.rank_f[a] + cnt - 1;
    if (r >= g.total_last) return g.size;
    if (r < 0) return -1;
    uint64_t li = g.lines[e >> 6].fwd_hint[a - 1];
    while (li + 1 < g.n_lines && (int64_t)(g.lines[li].rank_last + (uint64_t)__popcll(g.lines[li].last)) <= r) ++li;
    const GLine &L = g.lines[li];
    return (int64_t)(li << 6) + select64(L.last, (int)(r - (int64_t)L.rank_last));
}

__device__ __forceinline__ int g_node_last_char(const GraphDev &g, int64_t x) {   // succinct_dbg.h:109-115
    int i = 1;
    while (i < 5 && !(g.f[i] > x)) ++i;
    return i - 1;
}

__device__ __forceinline__ int64_t g_backward(const GraphDev &g, int64_t e) {   // succinct_dbg.h:166-170
    int a = g_node_last_char(g, e);
    int64_t cnt = g_rank_last(g, e - 1) - g.rank_f[a];
    return g_select_w(g, a, cnt);
}

// OutgoingEdges (succinct_dbg.cpp:78-97): valid edges of the node Forward(e) points to, in
// descending id order.  Per edge also its out-label (1..4) and multi1 bit, packed: id<<4 | multi1<<3 | label
__device__ __forceinline__ int g_outgoing(const GraphDev &g, int64_t e, int64_t out[4]) {
    if (!g_valid(g, e)) return -1;
    int od = 0;
    int64_t x = g_forward(g, e);
    do {
        const GLine &L = g.lines[x >> 6];
        if (!g_bit(L.invalid, x)) {
            int w = (int)((L.w[(x >> 4) & 3] >> ((x & 15) * 4)) & 15);
            if (od < 4) out[od] = (x << 4) | ((int64_t)g_bit(L.multi1, x) << 3) | (int64_t)(w > 4 ? w - 4 : w);
            ++od;
        }
        --x;
    } while (x >= 0 && !g_last_or_tip(g, x));
    return od > 4 ? 4 : od;
}

// ---- line-at-a-time navigation for the A* kernel -------------------------------------------------
// A whole 128-byte line is pulled into registers with one burst of loads; OutgoingEdges then costs one more
// burst (the hinted target line) instead of a chain of dependent 8-byte loads.  The target line is handed
// back: the returned child edges live in it, so the next level starts without a fetch.  Everything is kept
// in named scalars (no indexed register arrays: those would be demoted to scratch memory).
struct LineR {
    uint64_t w0, w1, w2, w3, last, tip, invalid, multi1, rank_last, rw0, rw1, rw2, rw3, h01, h23;
};
__device__ __forceinline__ LineR g_load_line(const GraphDev &g, uint64_t li) {
    const uint4 *q = reinterpret_cast<const uint4 *>(g.lines + li);
    uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4], v5 = q[5], v6 = q[6], v7 = q[7];
    auto u64 = [](uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)hi << 32); };
    LineR L;
    L.w0 = u64(v0.x, v0.y); L.w1 = u64(v0.z, v0.w); L.w2 = u64(v1.x, v1.y); L.w3 = u64(v1.z, v1.w);
    L.last = u64(v2.x, v2.y); L.tip = u64(v2.z, v2.w); L.invalid = u64(v3.x, v3.y); L.multi1 = u64(v3.z, v3.w);
    L.rank_last = u64(v4.x, v4.y);                      // (v4.z, v4.w) = rank_tip, not needed here
    L.rw0 = u64(v5.x, v5.y); L.rw1 = u64(v5.z, v5.w); L.rw2 = u64(v6.x, v6.y); L.rw3 = u64(v6.z, v6.w);
    L.h01 = u64(v7.x, v7.y); L.h23 = u64(v7.z, v7.w);
    return L;
}
__device__ __forceinline__ uint64_t sel4(uint64_t a0, uint64_t a1, uint64_t a2, uint64_t a3, int i) {
    uint64_t lo = (i & 1) ? a1 : a0, hi = (i & 1) ? a3 : a2;
    return (i & 2) ? hi : lo;
}
__device__ __forceinline__ int l_W(const LineR &L, int64_t x) {
    return (int)((sel4(L.w0, L.w1, L.w2, L.w3, (int)(x >> 4) & 3) >> ((x & 15) * 4)) & 15);
}

// `Le` must be line e>>6.  Outgoing edges (packed id<<4 | multi1<<3 | label) in o0..o3; Lt / lt_idx = line of o0.
__device__ __forceinline__ int g_outgoing_line(const GraphDev &g, const LineR &Le, int64_t e, int64_t &o0, int64_t &o1, int64_t &o2,
                                               int64_t &o3, LineR &Lt, uint64_t &lt_idx) {
    if (g_bit(Le.invalid, e)) return -1;
    int a = l_W(Le, e);
    if (a > 4) a -= 4;
    // Rank(a, e) from the registers (RankAndSelect4Bits::Rank, rank_and_select.h:153)
    int64_t cnt;
    if (e >= g.size - 1) cnt = a == 1 ? g.total_w[1] : a == 2 ? g.total_w[2] : a == 3 ? g.total_w[3] : g.total_w[4];
    else {
        int j = (int)(e & 63), fw = j >> 4;
        int nb = (j & 15) + 1;
        uint64_t m = nb == 16 ? ~0ull : ((1ull << (4 * nb)) - 1);
        cnt = (int64_t)sel4(Le.rw0, Le.rw1, Le.rw2, Le.rw3, a - 1);
        uint64_t e0 = nib_eq(Le.w0, a), e1 = nib_eq(Le.w1, a), e2 = nib_eq(Le.w2, a), e3 = nib_eq(Le.w3, a);
        cnt += __popcll(fw > 0 ? e0 : (e0 & m));
        if (fw >= 1) cnt += __popcll(fw > 1 ? e1 : (e1 & m));
        if (fw >= 2) cnt += __popcll(fw > 2 ? e2 : (e2 & m));
        if (fw >= 3) cnt += __popcll(e3 & m);
    }
    const int64_t rf = a == 1 ? g.rank_f[1] : a == 2 ? g.rank_f[2] : a == 3 ? g.rank_f[3] : g.rank_f[4];
    int64_t r = rf + cnt - 1;                                             // Forward: Select(rank_f[a] + count - 1), succinct_dbg.h:155-164
    if (r >= g.total_last || r < 0) return 0;
    uint64_t hh = (a <= 2) ? Le.h01 : Le.h23;
    uint64_t li = (a & 1) ? (hh & 0xFFFFFFFFull) : (hh >> 32);           // fwd_hint[a-1]
    LineR A = g_load_line(g, li);
    // (ones of `last` before the NEXT line = before this one + in this one: no need to touch the next line to know that the target is here)
    const uint64_t next_rank = A.rank_last + (uint64_t)__popcll(A.last);
    if (li + 1 < g.n_lines && (int64_t)next_rank <= r) {                  // rare: the target is a line or two further
        do { ++li; } while (li + 1 < g.n_lines && (int64_t)g.lines[li + 1].rank_last <= r);
        A = g_load_line(g, li);
    }
    int64_t x = (int64_t)(li << 6) + select64(A.last, (int)(r - (int64_t)A.rank_last));
    Lt = A; lt_idx = li;
    int od = 0;
    do {                                                                  // succinct_dbg.cpp:85-94
        const bool in = (uint64_t)(x >> 6) == li;                         // almost always: the node's edges sit in the target line
        bool inval = in ? g_bit(A.invalid, x) : !g_valid(g, x);
        if (!inval) {
            int w = in ? l_W(A, x) : g_W(g, x);
            int m1 = in ? g_bit(A.multi1, x) : (int)g_multi1(g, x);
            int64_t v = (x << 4) | ((int64_t)m1 << 3) | (int64_t)(w > 4 ? w - 4 : w);
            if (od == 0) o0 = v; else if (od == 1) o1 = v; else if (od == 2) o2 = v; else if (od == 3) o3 = v;
            ++od;
        }
        --x;
    } while (x >= 0 && !(((uint64_t)(x >> 6) == li) ? g_bit(A.last | A.tip, x) : (int)g_last_or_tip(g, x)));
    return od > 4 ? 4 : od;
}

// ---- OutgoingEdges without a loop: the edges of the node Forward(e) points to are the positions (y, x] where x is the node's `last`
// edge and y the nearest lower edge with last | tip set; inside one line that is a mask, the valid ones are mask & ~invalid, and the
// n-th edge in the reference's (descending) order is the n-th highest set bit.  Only a node whose edges straddle a line boundary
// (about one in thirty) needs the words of the line before (vm_lo, fetched on demand).
struct OutSet {
    uint64_t vm_hi, vm_lo;        // valid out-edges in line li / in line li - 1
    uint64_t li;
    int od;                       // min(4, number of edges), -1 = `e` is not a valid edge
};
// Aout = line o.li (the edges' own line), written when od > 0
__device__ __forceinline__ OutSet g_outset_line(const GraphDev &g, const LineR &Le, int64_t e, uint64_t &aw0, uint64_t &aw1, uint64_t &aw2,
                                                uint64_t &aw3, uint64_t &am1) {
    OutSet o;
    o.vm_hi = 0; o.vm_lo = 0; o.li = 0; o.od = 0;
    if (g_bit(Le.invalid, e)) { o.od = -1; return o; }
    int a = l_W(Le, e);
    if (a > 4) a -= 4;
    int64_t cnt;                                                          // Rank(a, e) from the registers (rank_and_select.h:153)
    if (e >= g.size - 1) cnt = a == 1 ? g.total_w[1] : a == 2 ? g.total_w[2] : a == 3 ? g.total_w[3] : g.total_w[4];
    else {
        const int j = (int)(e & 63), fw = j >> 4;
        const int nb = (j & 15) + 1;
        const uint64_t m = nb == 16 ? ~0ull : ((1ull << (4 * nb)) - 1);
        cnt = (int64_t)sel4(Le.rw0, Le.rw1, Le.rw2, Le.rw3, a - 1);
        const uint64_t e0 = nib_eq(Le.w0, a), e1 = nib_eq(Le.w1, a), e2 = nib_eq(Le.w2, a), e3 = nib_eq(Le.w3, a);
        cnt += __popcll(fw > 0 ? e0 : (e0 & m));
        if (fw >= 1) cnt += __popcll(fw > 1 ? e1 : (e1 & m));
        if (fw >= 2) cnt += __popcll(fw > 2 ? e2 : (e2 & m));
        if (fw >= 3) cnt += __popcll(e3 & m);
    }
    const int64_t rf = a == 1 ? g.rank_f[1] : a == 2 ? g.rank_f[2] : a == 3 ? g.rank_f[3] : g.rank_f[4];
    const int64_t r = rf + cnt - 1;                                       // Forward: Select(rank_f[a] + count - 1), succinct_dbg.h:155-164
    if (r >= g.total_last || r < 0) return o;
    const uint64_t hh = (a <= 2) ? Le.h01 : Le.h23;
    uint64_t li = (a & 1) ? (hh & 0xFFFFFFFFull) : (hh >> 32);           // fwd_hint[a-1]
    LineR A = g_load_line(g, li);
    // (ones of `last` before the NEXT line = before this one + in this one: no need to touch the next line to know that the target is here)
    uint64_t next_rank = A.rank_last + (uint64_t)__popcll(A.last);
    // the target is a line further (now and then two): that line is fetched WHOLE at once -- its own counts say whether the target is in it --
    // instead of first asking the line behind it for its rank and then fetching the line (two dependent trips)
    while (li + 1 < g.n_lines && (int64_t)next_rank <= r) {
        ++li;
        A = g_load_line(g, li);
        next_rank = A.rank_last + (uint64_t)__popcll(A.last);
    }
    const int xj = select64(A.last, (int)(r - (int64_t)A.rank_last));
    const uint64_t upto = xj == 63 ? ~0ull : ((2ull << xj) - 1ull);      // bits 0 .. xj
    const uint64_t stop = (A.last | A.tip) & (upto >> 1);                 // last | tip strictly below xj
    o.li = li;
    if (stop) {
        const int y = 63 - __builtin_clzll(stop);
        o.vm_hi = upto & ~((2ull << y) - 1ull) & ~A.invalid;
    } else {
        o.vm_hi = upto & ~A.invalid;
        if (li > 0) {                                                     // the node began in the line before
            const GLine &P = g.lines[li - 1];
            const uint64_t pstop = P.last | P.tip;
            const uint64_t keep = pstop ? ~((2ull << (63 - __builtin_clzll(pstop))) - 1ull) : ~0ull;
            o.vm_lo = keep & ~P.invalid;
            if (pstop >> 63) o.vm_lo = 0;
        }
    }
    const int n = __popcll(o.vm_hi) + __popcll(o.vm_lo);
    o.od = n > 4 ? 4 : n;
    // (field by field: a whole-struct copy is written to scratch memory even when nothing reads it back; only W and multi1 are needed)
    aw0 = A.w0; aw1 = A.w1; aw2 = A.w2; aw3 = A.w3; am1 = A.multi1;
    return o;
}
// n-th out-edge (n < od) packed id << 4 | multi1 << 3 | label; A = line o.li
__device__ __forceinline__ int64_t g_outset_get(const GraphDev &g, const OutSet &o, uint64_t aw0, uint64_t aw1, uint64_t aw2, uint64_t aw3,
                                                uint64_t am1, int n) {
    const int nh = __popcll(o.vm_hi);
    const bool hi = n < nh;
    uint64_t m = hi ? o.vm_hi : o.vm_lo;
    int skip = hi ? n : n - nh;
    if (skip >= 1) m &= ~(1ull << (63 - __builtin_clzll(m)));
    if (skip >= 2) m &= ~(1ull << (63 - __builtin_clzll(m)));
    if (skip >= 3) m &= ~(1ull << (63 - __builtin_clzll(m)));
    const int pos = 63 - __builtin_clzll(m | 1ull);
    int w, m1;
    int64_t x;
    if (hi) {
        x = (int64_t)(o.li << 6) + pos;
        w = (int)((sel4(aw0, aw1, aw2, aw3, (pos >> 4) & 3) >> ((pos & 15) * 4)) & 15); m1 = g_bit(am1, x);
    } else {
        x = (int64_t)((o.li - 1) << 6) + pos;
        w = g_W(g, x); m1 = (int)g_multi1(g, x);
    }
    return (x << 4) | ((int64_t)m1 << 3) | (int64_t)(w > 4 ? w - 4 : w);
}
// OutgoingEdges(e)[n] and the out-degree; the lines never leave the function (a 120-byte struct that flows from call to call ends
// up in scratch memory)
__device__ __forceinline__ int g_out_nth(const GraphDev &g, int64_t e, int n, int64_t &edge) {
    uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, m1 = 0;
    const OutSet o = g_outset_line(g, g_load_line(g, (uint64_t)e >> 6), e, w0, w1, w2, w3, m1);
    if (n < o.od) edge = g_outset_get(g, o, w0, w1, w2, w3, m1, n);
    return o.od;
}
__device__ __forceinline__ int g_out_all(const GraphDev &g, int64_t e, int64_t &o0, int64_t &o1, int64_t &o2, int64_t &o3) {
    uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, m1 = 0;
    const OutSet o = g_outset_line(g, g_load_line(g, (uint64_t)e >> 6), e, w0, w1, w2, w3, m1);
    if (o.od > 0) o0 = g_outset_get(g, o, w0, w1, w2, w3, m1, 0);
    if (o.od > 1) o1 = g_outset_get(g, o, w0, w1, w2, w3, m1, 1);
    if (o.od > 2) o2 = g_outset_get(g, o, w0, w1, w2, w3, m1, 2);
    if (o.od > 3) o3 = g_outset_get(g, o, w0, w1, w2, w3, m1, 3);
    return o.od;
}

// ---- forward descriptors --------------------------------------------------------------------------------------------------------
// Everything OutgoingEdges(e) needs from e's own line is the rank r of the `last` bit Forward(e) selects and the hint line where the
// look-up starts (succinct_dbg.h:155-164).  Both can be computed when e is FOUND -- e is then an out-edge in a line that sits in
// registers -- so a walk edge -> out-edges -> their out-edges ... fetches ONE line per step (the target) instead of two (the edge's own
// line again, then the target).  FwdDesc travels with the edge: in the A* kernel also inside the search node, so that an expansion
// starts at the target line of its node.  hint == kFdNone: not known (an edge of the line before the target line, the seed's start
// edge): the two-line path above.
constexpr uint32_t kFdNone = 0xFFFFFFFFu;
struct FwdDesc {
    int64_t r;            // Select argument of Forward: rank_f[a] + Rank(a, e) - 1
    uint32_t hint;        // fwd_hint[a - 1] of e's line, or kFdNone
};
// descriptor of the edge at bit `pos` of the line whose words are given (the edge is valid and carries a symbol 1..8)
__device__ __forceinline__ FwdDesc g_fd_of(const GraphDev &g, int64_t x, int pos, uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3, uint64_t rw0, uint64_t rw1,
                                           uint64_t rw2, uint64_t rw3, uint64_t h01, uint64_t h23) {
    FwdDesc f;
    int a = (int)((sel4(w0, w1, w2, w3, (pos >> 4) & 3) >> ((pos & 15) * 4)) & 15);
    if (a > 4) a -= 4;
    int64_t cnt;
    if (x >= g.size - 1) cnt = a == 1 ? g.total_w[1] : a == 2 ? g.total_w[2] : a == 3 ? g.total_w[3] : g.total_w[4];
    else {
        const int fw = pos >> 4, nb = (pos & 15) + 1;
        const uint64_t m = nb == 16 ? ~0ull : ((1ull << (4 * nb)) - 1);
        cnt = (int64_t)sel4(rw0, rw1, rw2, rw3, a - 1);
        const uint64_t e0 = nib_eq(w0, a), e1 = nib_eq(w1, a), e2 = nib_eq(w2, a), e3 = nib_eq(w3, a);
        cnt += __popcll(fw > 0 ? e0 : (e0 & m));
        if (fw >= 1) cnt += __popcll(fw > 1 ? e1 : (e1 & m));
        if (fw >= 2) cnt += __popcll(fw > 2 ? e2 : (e2 & m));
        if (fw >= 3) cnt += __popcll(e3 & m);
    }
    const int64_t rf = a == 1 ? g.rank_f[1] : a == 2 ? g.rank_f[2] : a == 3 ? g.rank_f[3] : g.rank_f[4];
    f.r = rf + cnt - 1;
    const uint64_t hh = (a <= 2) ? h01 : h23;
    f.hint = (uint32_t)((a & 1) ? (hh & 0xFFFFFFFFull) : (hh >> 32));
    return f;
}
// the out-set of the node Forward selects with rank `r`, starting at line `li` (g_outset_line from its second half on), with the whole
// target line handed back in named scalars
struct LineOut { uint64_t w0, w1, w2, w3, m1, rw0, rw1, rw2, rw3, h01, h23; };
__device__ __forceinline__ OutSet g_outset_from(const GraphDev &g, int64_t r, uint64_t li, LineOut &L) {
    OutSet o;
    o.vm_hi = 0; o.vm_lo = 0; o.li = 0; o.od = 0;
    if (r >= g.total_last || r < 0) return o;
    LineR A = g_load_line(g, li);
    // (ones of `last` before the NEXT line = before this one + in this one: no need to touch the next line to know that the target is here)
    const uint64_t next_rank = A.rank_last + (uint64_t)__popcll(A.last);
    if (li + 1 < g.n_lines && (int64_t)next_rank <= r) {                  // rare: the target is a line or two further
        do { ++li; } while (li + 1 < g.n_lines && (int64_t)g.lines[li + 1].rank_last <= r);
        A = g_load_line(g, li);
    }
    const int xj = select64(A.last, (int)(r - (int64_t)A.rank_last));
    const uint64_t upto = xj == 63 ? ~0ull : ((2ull << xj) - 1ull);      // bits 0 .. xj
    const uint64_t stop = (A.last | A.tip) & (upto >> 1);                 // last | tip strictly below xj
    o.li = li;
    if (stop) {
        const int y = 63 - __builtin_clzll(stop);
        o.vm_hi = upto & ~((2ull << y) - 1ull) & ~A.invalid;
    } else {
        o.vm_hi = upto & ~A.invalid;
        if (li > 0) {                                                     // the node began in the line before
            const GLine &P = g.lines[li - 1];
            const uint64_t pstop = P.last | P.tip;
            const uint64_t keep = pstop ? ~((2ull << (63 - __builtin_clzll(pstop))) - 1ull) : ~0ull;
            o.vm_lo = keep & ~P.invalid;
            if (pstop >> 63) o.vm_lo = 0;
        }
    }
    const int n = __popcll(o.vm_hi) + __popcll(o.vm_lo);
    o.od = n > 4 ? 4 : n;
    L.w0 = A.w0; L.w1 = A.w1; L.w2 = A.w2; L.w3 = A.w3; L.m1 = A.multi1;
    L.rw0 = A.rw0; L.rw1 = A.rw1; L.rw2 = A.rw2; L.rw3 = A.rw3; L.h01 = A.h01; L.h23 = A.h23;
    return o;
}
// descriptor of edge e from its own line (the slow start of a chain: one extra line); od < 0 when e is not a valid edge
__device__ __forceinline__ FwdDesc g_fd_load(const GraphDev &g, int64_t e, bool &valid_edge) {
    const LineR Le = g_load_line(g, (uint64_t)e >> 6);
    valid_edge = !g_bit(Le.invalid, e);
    return g_fd_of(g, e, (int)(e & 63), Le.w0, Le.w1, Le.w2, Le.w3, Le.rw0, Le.rw1, Le.rw2, Le.rw3, Le.h01, Le.h23);
}
// n-th out-edge of an out-set with its own descriptor (kFdNone when the edge lies in the line before the target line)
__device__ __forceinline__ int64_t g_outset_get_fd(const GraphDev &g, const OutSet &o, const LineOut &L, int n, FwdDesc &fd) {
    const int nh = __popcll(o.vm_hi);
    const bool hi = n < nh;
    uint64_t m = hi ? o.vm_hi : o.vm_lo;
    int skip = hi ? n : n - nh;
    if (skip >= 1) m &= ~(1ull << (63 - __builtin_clzll(m)));
    if (skip >= 2) m &= ~(1ull << (63 - __builtin_clzll(m)));
    if (skip >= 3) m &= ~(1ull << (63 - __builtin_clzll(m)));
    const int pos = 63 - __builtin_clzll(m | 1ull);
    int w, m1;
    int64_t x;
    if (hi) {
        x = (int64_t)(o.li << 6) + pos;
        w = (int)((sel4(L.w0, L.w1, L.w2, L.w3, (pos >> 4) & 3) >> ((pos & 15) * 4)) & 15); m1 = g_bit(L.m1, x);
        fd = g_fd_of(g, x, pos, L.w0, L.w1, L.w2, L.w3, L.rw0, L.rw1, L.rw2, L.rw3, L.h01, L.h23);
    } else {
        x = (int64_t)((o.li - 1) << 6) + pos;
        w = g_W(g, x); m1 = (int)g_multi1(g, x);
        fd.r = 0; fd.hint = kFdNone;
    }
    return (x << 4) | ((int64_t)m1 << 3) | (int64_t)(w > 4 ? w - 4 : w);
}
// OutgoingEdges of the edge described by `in` (or of `e` itself when in.hint == kFdNone): the n-th one + its descriptor; returns the out-degree
__device__ __forceinline__ int g_out_nth_fd(const GraphDev &g, int64_t e, FwdDesc in, int n, int64_t &edge, FwdDesc &out) {
    if (in.hint == kFdNone) {
        bool ok;
        in = g_fd_load(g, e, ok);
        if (!ok) return -1;
    }
    LineOut L;
    const OutSet o = g_outset_from(g, in.r, in.hint, L);
    if (n < o.od) edge = g_outset_get_fd(g, o, L, n, out);
    return o.od;
}
__device__ __forceinline__ int g_out_all_fd(const GraphDev &g, int64_t e, FwdDesc in, int64_t &o0, int64_t &o1, int64_t &o2, int64_t &o3, FwdDesc &f0, FwdDesc &f1,
                                            FwdDesc &f2, FwdDesc &f3) {
    if (in.hint == kFdNone) {
        bool ok;
        in = g_fd_load(g, e, ok);
        if (!ok) return -1;
    }
    LineOut L;
    const OutSet o = g_outset_from(g, in.r, in.hint, L);
    if (o.od > 0) o0 = g_outset_get_fd(g, o, L, 0, f0);
    if (o.od > 1) o1 = g_outset_get_fd(g, o, L, 1, f1);
    if (o.od > 2) o2 = g_outset_get_fd(g, o, L, 2, f2);
    if (o.od > 3) o3 = g_outset_get_fd(g, o, L, 3, f3);
    return o.od;
}

__device__ __forceinline__ int g_tip_char(const GraphDev &g, int64_t tip_rank, int j) {
    const uint32_t *t = g.tip_labels + (size_t)g.words_per_tip * tip_rank;
    return (t[j >> 4] >> (15 - (j & 15)) * 2) & 3;
}
__device__ __forceinline__ int64_t g_rank_tip(const GraphDev &g, int64_t pos) {   // tips in [0..pos]
    const GLine &L = g.lines[pos >> 6];
    int j = (int)(pos & 63);
    uint64_t m = j == 63 ? ~0ull : ((1ull << (j + 1)) - 1);
    return (int64_t)L.rank_tip + __popcll(L.tip & m);
}

// IndexBinarySearch (succinct_dbg.cpp:427-501): node whose label is seq[0..k-1] (symbols 1..4), or -1
__device__ inline int64_t g_index_node(const GraphDev &g, const uint8_t *seq) {
    const int k = g.k;
    int64_t l = g.f[seq[k - 1]], r = g.f[seq[k - 1] + 1] - 1;
    while (l <= r) {
        int cmp = 0;
        int64_t mid = (l + r) / 2, y = mid;
        for (int i = k - 1; i >= 0; --i) {
            if (g_tip(g, y)) {
                int64_t tr = g_rank_tip(g, y) - 1;
                for (int j = 0; j < i; ++j) {
                    int c = g_tip_char(g, tr, j) + 1;
                    if (c < seq[i - j]) { cmp = -1; break; }
                    if (c > seq[i - j]) { cmp = 1; break; }
                }
                if (cmp == 0) {
                    if (g_tip(g, mid)) cmp = -1;
                    else {
                        int c = g_tip_char(g, tr, i) + 1;
                        if (c < seq[0]) cmp = -1;
                        else if (c > seq[0]) cmp = 1;
                    }
                }
                break;
            }
            y = g_backward(g, y);
            int c = g_W(g, y);
            if (c < seq[i]) { cmp = -1; break; }
            if (c > seq[i]) { cmp = 1; break; }
        }
        if (cmp == 0) {
            int64_t p = mid;                       // GetLastIndex = rs_last_.Succ(mid)
            while (p < g.size && !g_last(g, p)) ++p;
            return p;
        }
        if (cmp > 0) r = mid - 1; else l = mid + 1;
    }
    return -1;
}

// IndexBinarySearchEdge (succinct_dbg.cpp:530-549): seq has k+1 symbols
__device__ inline int64_t g_index_edge(const GraphDev &g, const uint8_t *seq) {
    int64_t node = g_index_node(g, seq);
    if (node == -1) return -1;
    do {
        int lab = g_W(g, node);
        if (lab == seq[g.k] || lab - 4 == seq[g.k]) return node;
        --node;
    } while (node >= 0 && !g_last_or_tip(g, node));
    return -1;
}

}  // namespace mgta

struct mgta_sdbg {
    mgta_ctx *ctx = nullptr;
    mgta::GraphDev dev;                       // host copy of the descriptor (pointers are device pointers)
    mgta::DevBuf lines, sel_last, sel_w[5], tips;
};
