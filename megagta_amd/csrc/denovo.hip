// denovo.hip — `megagta denovo` (tips, bubbles, unitigs) on the device-resident succinct de Bruijn graph.
//
// Replaces main_assemble (assembler.cpp:98-167): RemoveTips / Trim (assembly_algorithms.cpp:76-183), PopBubbles (:245-301) with
// BranchGroup::Search / Pop (branch_group.cpp:22-141), and UnitigGraph::InitFromSdBG writing contigs (unitig_graph.cpp:80-150,208-303).
//
// The reference's OpenMP loops race (its 8-thread output differs from its 1-thread output on the same graph); the result reproduced
// here is the ONE-thread run, byte for byte, and it is computed in parallel:
//  * Trim marks on a snapshot and deletes afterwards, so every start node is an independent thread.
//  * PopBubbles is order dependent (a pop changes what the next search sees).  Pending candidates are taken in windows of ascending edge
//    id.  Every candidate of the window stamps, by 64-bit atomicMax of (round, ~rank) in a hash table keyed by edge id (a few GB whatever
//    the size of the graph: only the edges a window reaches carry a stamp), every edge its search could EVER read or write
//    while the graph only loses edges: all edges within max_bubble_len forward steps of its begin edge plus the edges into them (its
//    "reach").  Then it runs its search on the current graph and commits (pops) iff every still-valid edge the search read carries its
//    own stamp: no lower pending candidate can reach what it read or writes, now or after any later change, and its own writes stay
//    outside every lower candidate's reach - so each committed search saw exactly what the sequential loop would have shown it.  The
//    rest stay pending, in order; the lowest pending candidate always commits.  No clearing: a newer round outranks old stamps.
//  * Unitigs: the reference walks every maximal simple path back from its end edge in ascending order, locking edges in a bit
//    vector, then locks the path of the reverse complement forwards from RC(end).  With one thread the locks held inside a path are
//    always a suffix of it that reaches its end edge, so the whole protocol collapses to: path P (end e) is skipped iff an earlier
//    processed (q < e), not skipped path Q has RC(q) inside P.  One walk per path back from its end edge (start, length, depth), one
//    walk forwards from RC(end) to the end edge of the path it lies on (the reference's own locking walk: the claim = that path and
//    the distance); the recursion over ascending ids is resolved by a few data-parallel sweeps; the rarely taken "RC(end) already
//    locked" branch (:243-252) is evaluated from the same claims; the labels are written by a second walk of the paths that are
//    emitted.  Nothing is kept per edge (round 2 kept 8 bytes per edge twice: stamps and a path table), path ids are 64-bit.
//    (PopBubbles keeps the LAST branch on equal multiplicities, which is a different allele on the two strands: the graph is not
//    strand-symmetric afterwards, so none of this may assume that RC(path) is a path.)
#include <algorithm>
#include <chrono>
#include <memory>
#include <thread>

#include "graph.hpp"
#include "scan.hpp"

namespace mgta {
namespace {

constexpr int kMaxBranches = 16;      // kMaxBranchesPerGroup, assembly_algorithms.cpp:248
constexpr uint32_t kBubbleWindowMax = 1u << 18;
constexpr int kReachHash = 32768, kReachMax = 16384, kReachFrontier = 2048;   // per-candidate reach search: hash slots, edges, edges per level
constexpr int kMaxK = 255;

struct Dn {
    GraphDev g;
    GLine *rw;     // the same lines, writable: validity bits
};

// ---- bit vectors (AtomicBitVector, one bit per edge) ---------------------------------------------------------------------------
__device__ __forceinline__ bool bit_get(const unsigned long long *b, int64_t i) { return (b[i >> 6] >> (i & 63)) & 1; }
__device__ __forceinline__ void bit_set(unsigned long long *b, int64_t i) { atomicOr(&b[i >> 6], 1ull << (i & 63)); }
__device__ __forceinline__ void bit_unset(unsigned long long *b, int64_t i) { atomicAnd(&b[i >> 6], ~(1ull << (i & 63))); }
__device__ __forceinline__ bool bit_try_lock(unsigned long long *b, int64_t i) { return !((atomicOr(&b[i >> 6], 1ull << (i & 63)) >> (i & 63)) & 1); }

__device__ __forceinline__ void set_invalid(const Dn &d, int64_t e) { atomicOr((unsigned long long *)&d.rw[e >> 6].invalid, 1ull << (e & 63)); }
__device__ __forceinline__ void set_valid(const Dn &d, int64_t e) { atomicAnd((unsigned long long *)&d.rw[e >> 6].invalid, ~(1ull << (e & 63))); }
__device__ __forceinline__ int multiplicity(const GraphDev &g, int64_t e) { return 2 - (int)g_multi1(g, e); }   // succinct_dbg.h:133-135, need_multiplicity = false

// ---- navigation the A* path does not need (succinct_dbg.cpp:99-371) -----------------------------------------------------------
__device__ inline int64_t d_last_index(const GraphDev &g, int64_t x) {   // GetLastIndex = rs_last_.Succ, succinct_dbg.h:105
    uint64_t li = (uint64_t)x >> 6;
    uint64_t m = g.lines[li].last >> (x & 63);
    if (m) return x + __ffsll((long long)m) - 1;
    while (++li < g.n_lines) {
        uint64_t w = g.lines[li].last;
        if (w) return (int64_t)(li << 6) + __ffsll((long long)w) - 1;
    }
    return g.size;
}

// edges that point to the node of x (IncomingEdges / UniquePrev* / NodeIndegreeZero / DeleteAllEdges share this scan)
template <bool ALL>
__device__ inline int d_incoming_scan(const GraphDev &g, int64_t x, int64_t *in) {
    int64_t first = g_backward(g, x);
    const int c = g_W(g, first);
    int ones = g_last_or_tip(g, first), n = 0;
    if (ALL || g_valid(g, first)) in[n++] = first;
    for (int64_t y = first + 1; ones < 5 && y < g.size; ++y) {
        ones += g_last_or_tip(g, y);
        const int cur = g_W(g, y);
        if (cur == c) break;
        if (cur == c + 4 && (ALL || g_valid(g, y)) && n < 8) in[n++] = y;
    }
    return n;
}
template <bool ALL>
__device__ inline int d_node_edges(const GraphDev &g, int64_t node, int64_t *out) {   // the node's own edges, from its last one downwards
    int64_t e = d_last_index(g, node);
    int n = 0;
    do {
        if ((ALL || g_valid(g, e)) && n < 8) out[n++] = e;
        --e;
    } while (e >= 0 && !g_last_or_tip(g, e));
    return n;
}
__device__ inline int d_outgoing(const GraphDev &g, int64_t e, int64_t *out) {   // OutgoingEdges, valid edge ids in descending order
    if (!g_valid(g, e)) return -1;
    int n = 0;
    int64_t x = g_forward(g, e);
    do {
        if (g_valid(g, x) && n < 8) out[n++] = x;
        --x;
    } while (x >= 0 && !g_last_or_tip(g, x));
    return n;
}
__device__ inline int d_incoming(const GraphDev &g, int64_t e, int64_t *in) {
    if (!g_valid(g, e)) return -1;
    return d_incoming_scan<false>(g, e, in);
}
__device__ inline bool node_outdegree_zero(const GraphDev &g, int64_t node) { int64_t t[8]; return d_node_edges<false>(g, node, t) == 0; }
__device__ inline bool node_indegree_zero(const GraphDev &g, int64_t node) { int64_t t[8]; return d_incoming_scan<false>(g, node, t) == 0; }
__device__ inline int64_t unique_prev_node(const GraphDev &g, int64_t node) {
    int64_t t[8];
    return d_incoming_scan<false>(g, node, t) == 1 ? d_last_index(g, t[0]) : -1;
}
__device__ inline int64_t unique_next_node(const GraphDev &g, int64_t node) {
    int64_t t[8];
    return d_node_edges<false>(g, node, t) == 1 ? d_last_index(g, g_forward(g, t[0])) : -1;
}
__device__ inline int64_t unique_next_edge(const GraphDev &g, int64_t e) {
    int64_t t[8];
    return d_outgoing(g, e, t) == 1 ? t[0] : -1;
}
__device__ inline int64_t unique_prev_edge(const GraphDev &g, int64_t e) {
    int64_t t[8];
    return d_incoming(g, e, t) == 1 ? t[0] : -1;
}
__device__ inline int64_t prev_simple(const GraphDev &g, int64_t e) {   // PrevSimplePathEdge
    int64_t p = unique_prev_edge(g, e);
    return p != -1 && unique_next_edge(g, p) != -1 ? p : -1;
}
__device__ inline int64_t next_simple(const GraphDev &g, int64_t e) {   // NextSimplePathEdge
    int64_t n = unique_next_edge(g, e);
    return n != -1 && unique_prev_edge(g, n) != -1 ? n : -1;
}

__device__ inline void d_label(const GraphDev &g, int64_t e, uint8_t *seq) {   // Label, succinct_dbg.cpp:503-528 (symbols 1..4)
    int64_t x = e;
    for (int i = g.k - 1; i >= 0; --i) {
        if (g_tip(g, x)) {
            int64_t tr = g_rank_tip(g, x) - 1;
            for (int j = 0; j <= i; ++j) seq[i - j] = (uint8_t)(g_tip_char(g, tr, j) + 1);
            break;
        }
        x = g_backward(g, x);
        int c = g_W(g, x);
        seq[i] = (uint8_t)(c > 4 ? c - 4 : c);
    }
}
__device__ inline int64_t edge_reverse_complement(const GraphDev &g, int64_t e) {   // succinct_dbg.cpp:551-593
    if (!g_valid(g, e)) return -1;
    uint8_t seq[kMaxK + 2];
    d_label(g, e, seq);
    int w = g_W(g, e);
    seq[g.k] = (uint8_t)(w > 4 ? w - 4 : w);
    for (int i = 0, j = g.k; i <= j; ++i, --j) {
        uint8_t a = (uint8_t)(5 - seq[i]), b = (uint8_t)(5 - seq[j]);
        seq[i] = b; seq[j] = a;
    }
    return g_index_edge(g, seq);
}

// ---- ordered compaction: one 64-bit mask per line of edges -> ascending list of edge ids ------------------------------------------
struct PredTipStartOut {   // Trim, first loop: assembly_algorithms.cpp:82
    const unsigned long long *removed;
    __device__ bool operator()(const GraphDev &g, int64_t x) const { return g_last(g, x) && !bit_get(removed, x) && node_outdegree_zero(g, x); }
};
struct PredTipStartIn {    // second loop, :116
    const unsigned long long *removed;
    __device__ bool operator()(const GraphDev &g, int64_t x) const { return g_last(g, x) && !bit_get(removed, x) && node_indegree_zero(g, x); }
};
struct PredBranching {     // BranchGroup::Search can only succeed from an edge whose out-degree is 2..16 (branch_group.cpp:23-27)
    __device__ bool operator()(const GraphDev &g, int64_t x) const {
        int64_t t[8];
        return d_outgoing(g, x, t) >= 2;
    }
};
struct PredPathEnd {       // unitig_graph.cpp:223
    __device__ bool operator()(const GraphDev &g, int64_t x) const { return g_valid(g, x) && next_simple(g, x) == -1; }
};

template <class Pred>
__global__ __launch_bounds__(256) void edge_mask_kernel(GraphDev g, Pred pred, unsigned long long *mask, uint32_t *count) {
    // one wave = one line of 64 edges; capped grid (a dispatch holds < 2^32 work-items, a graph can hold more edges)
    for (uint64_t li = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); li < g.n_lines; li += (uint64_t)gridDim.x * 4) {
        const int64_t x = (int64_t)(li << 6) + (threadIdx.x & 63);
        const bool p = x < g.size && pred(g, x);
        const unsigned long long m = __ballot(p);
        if ((threadIdx.x & 63) == 0) {
            mask[li] = m;
            count[li] = (uint32_t)__popcll(m);
        }
    }
}
__global__ __launch_bounds__(256) void mask_expand_kernel(const unsigned long long *mask, const uint64_t *base, uint64_t n_lines, int64_t *list) {
    const uint64_t li = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (li >= n_lines) return;
    unsigned long long m = mask[li];
    uint64_t o = base[li];
    while (m) {
        int b = __ffsll((long long)m) - 1;
        list[o++] = (int64_t)(li << 6) + b;
        m &= m - 1;
    }
}
__global__ __launch_bounds__(256) void list_compact_kernel(const int64_t *in, const uint32_t *flag, const uint64_t *base, uint64_t n, int64_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && flag[i]) out[base[i]] = in[i];
}

// ---- tips -----------------------------------------------------------------------------------------------------------------------
template <bool OUT>
__global__ __launch_bounds__(64) void trim_walk_kernel(GraphDev g, const int64_t *starts, uint64_t n, int len, unsigned long long *removed,
                                                       unsigned long long *n_tips) {
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const int64_t x = starts[i];
    int64_t cur = x;
    bool is_tip = false;
    int steps = 0;
    for (int s = 1; s < len; ++s) {
        const int64_t nb = OUT ? unique_prev_node(g, cur) : unique_next_node(g, cur);
        if (nb == -1) { is_tip = OUT ? node_indegree_zero(g, cur) : node_outdegree_zero(g, cur); break; }
        if ((OUT ? unique_next_node(g, nb) : unique_prev_node(g, nb)) == -1) { is_tip = true; break; }
        cur = nb;
        ++steps;
    }
    if (!is_tip) return;
    cur = x;
    bit_set(removed, cur);
    for (int s = 0; s < steps; ++s) {      // the same walk again (nothing changes the graph while tips are being marked)
        cur = OUT ? unique_prev_node(g, cur) : unique_next_node(g, cur);
        bit_set(removed, cur);
    }
    atomicAdd(n_tips, 1ull);
}
__global__ __launch_bounds__(256) void trim_delete_kernel(Dn d, const unsigned long long *removed) {   // DeleteAllEdges of every marked node
    const uint64_t li = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (li >= d.g.n_lines) return;
    unsigned long long m = removed[li];
    while (m) {
        const int64_t node = (int64_t)(li << 6) + __ffsll((long long)m) - 1;
        m &= m - 1;
        int64_t t[8];
        int n = d_node_edges<true>(d.g, node, t);
        for (int i = 0; i < n; ++i) set_invalid(d, t[i]);
        n = d_incoming_scan<true>(d.g, node, t);
        for (int i = 0; i < n; ++i) set_invalid(d, t[i]);
    }
}

// ---- bubbles --------------------------------------------------------------------------------------------------------------------
// Stamps of one round, keyed by edge id: open addressing, linear probing.  A key word = tag << 40 | edge id with tag = the round (1 ..
// 2^24 - 1): whatever an earlier round left behind reads as an empty slot and is claimed by CAS, so nothing is ever cleared; the value
// keeps (round << 32 | ~rank) by atomicMax exactly as the per-edge array of round 2 did.  Inserts (reach kernel) and look-ups (check
// kernel) are separated by a kernel boundary, slots only ever turn from stale to current inside a round, so a probe sequence that
// found a key keeps finding it.  An insert that sees kStampProbes occupied slots gives up: the caller treats its candidate as one
// whose region does not fit (it and everything above it waits for the next round, whose window the host then shrinks).
constexpr int kStampProbes = 128;
struct StampTab {
    unsigned long long *key, *val;
    uint64_t mask;
    unsigned long long tag;       // round tag << 40
    __device__ __forceinline__ uint64_t slot(int64_t e) const { return (((uint64_t)e * 0x9E3779B97F4A7C15ull) >> 20) & mask; }
    __device__ bool stamp(int64_t e, unsigned long long v) const {
        const unsigned long long mine = tag | (unsigned long long)e;
        uint64_t h = slot(e);
        for (int probes = 0; probes < kStampProbes;) {
            unsigned long long cur = __hip_atomic_load(&key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur != mine && (cur & ~0xFFFFFFFFFFull) != tag) {
                if (!__hip_atomic_compare_exchange_strong(&key[h], &cur, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) && cur != mine)
                    continue;                                      // another edge took the slot: look at it again
                cur = mine;
            }
            if (cur == mine) { atomicMax(&val[h], v); return true; }
            h = (h + 1) & mask;
            ++probes;
        }
        return false;
    }
    __device__ unsigned long long lookup(int64_t e) const {       // 0 = no stamp this round
        const unsigned long long mine = tag | (unsigned long long)e;
        uint64_t h = slot(e);
        for (int probes = 0; probes < kStampProbes; ++probes) {
            const unsigned long long cur = key[h];
            if (cur == mine) return val[h];
            if ((cur & ~0xFFFFFFFFFFull) != tag) return 0ull;
            h = (h + 1) & mask;
        }
        return 0ull;
    }
};

struct SinkNone { __device__ void operator()(int64_t) {} __device__ bool stop() const { return false; } };
struct SinkStamp {
    StampTab tab;
    unsigned long long key;
    bool ok;
    __device__ void operator()(int64_t e) { if (!tab.stamp(e, key)) ok = false; }
    __device__ bool stop() const { return false; }
};
struct SinkCheck {
    StampTab tab;
    unsigned long long key;
    bool ok;
    __device__ void operator()(int64_t e) { if (ok && tab.lookup(e) != key) ok = false; }
    __device__ bool stop() const { return !ok; }                   // an edge somebody below reaches: the candidate waits, whatever the search would find
};

// BranchGroup::Search (branch_group.cpp:22-103).  br[b * max_len + j] = j-th edge of branch b; every still-valid edge whose validity
// the search reads goes through `sink` (an invalid edge stays invalid, reading it is not a dependency).
template <class Sink>
__device__ bool bubble_search(const GraphDev &g, int64_t begin, int max_len, int64_t *br, int *mult, int &nb, int &len, Sink &sink) {
    if (!g_valid(g, begin)) return false;
    sink(begin);
    int64_t out[8];
    const int outd = d_outgoing(g, begin, out);
    for (int x = 0; x < outd; ++x) sink(out[x]);
    if (outd <= 1 || outd > kMaxBranches || sink.stop()) return false;
    nb = 1; len = 1;
    br[0] = begin;
    mult[0] = 0;
    bool converged = false;
    int64_t end = -1;
    for (int j = 1; j < max_len; ++j) {
        const int nb0 = nb;
        for (int i = 0; i < nb0; ++i) {
            const int od = d_outgoing(g, br[(size_t)i * max_len + j - 1], out);
            for (int x = 0; x < od; ++x) sink(out[x]);
            if (od < 1) return false;                       // a dead end never converges (the reference runs on and fails later)
            br[(size_t)i * max_len + j] = out[0];
            const int m0 = mult[i];
            mult[i] = m0 + multiplicity(g, out[0]);
            if (nb + od - 1 > kMaxBranches) return false;
            for (int x = 1; x < od; ++x) {
                for (int q = 0; q < j; ++q) br[(size_t)nb * max_len + q] = br[(size_t)i * max_len + q];
                br[(size_t)nb * max_len + j] = out[x];
                mult[nb] = m0 + multiplicity(g, out[x]);
                ++nb;
            }
        }
        len = j + 1;
        for (int b = 0; b < nb; ++b) {                      // every edge into a branch head must come from the group
            int64_t in[8];
            const int id = d_incoming(g, br[(size_t)b * max_len + j], in);
            for (int x = 0; x < id; ++x) sink(in[x]);
            if (id == 1) continue;
            for (int x = 0; x < id; ++x) {
                bool found = false;
                for (int o = 0; o < nb && !found; ++o) found = br[(size_t)o * max_len + j - 1] == in[x];
                if (!found) return false;
            }
        }
        end = br[j];
        const int eo = d_outgoing(g, end, out);
        for (int x = 0; x < eo; ++x) sink(out[x]);
        if (sink.stop()) return false;
        if (eo == 1) {
            converged = true;
            for (int b = 1; b < nb && converged; ++b) converged = br[(size_t)b * max_len + j] == end;
            if (converged) break;
        }
    }
    return converged && begin != end;
}

// BranchGroup::Pop (branch_group.cpp:105-141) with one thread: fails (and undoes itself) when an inner edge shows up twice
__device__ bool bubble_pop(const Dn &d, unsigned long long *marked, const int64_t *br, const int *mult, int nb, int len, int max_len) {
    int best = 0, best_m = mult[0];
    for (int i = 1; i < nb; ++i)
        if (mult[i] >= best_m) { best = i; best_m = mult[i]; }
    for (int i = 0; i < nb; ++i)
        for (int j = 1; j + 1 < len; ++j) {
            const int64_t e = br[(size_t)i * max_len + j];
            if (!bit_try_lock(marked, e)) {
                for (int i2 = 0; i2 <= i; ++i2)
                    for (int j2 = 1; j2 + 1 < len && (i2 < i || j2 < j); ++j2) {
                        const int64_t u = br[(size_t)i2 * max_len + j2];
                        bit_unset(marked, u);
                        set_valid(d, u);
                    }
                return false;
            }
            set_invalid(d, e);
        }
    for (int j = 1; j + 1 < len; ++j) {
        const int64_t e = br[(size_t)best * max_len + j];
        set_valid(d, e);
        bit_unset(marked, e);
    }
    return true;
}

__global__ __launch_bounds__(64) void bubble_find_kernel(GraphDev g, const int64_t *cand, uint64_t n, int max_len, int64_t *scratch, size_t per, uint32_t *found) {
    const uint64_t t = (uint64_t)blockIdx.x * 64 + threadIdx.x, stride = (uint64_t)gridDim.x * 64;
    int64_t *br = scratch + t * per;
    for (uint64_t i = t; i < n; i += stride) {
        int mult[kMaxBranches], nb = 0, len = 0;
        SinkNone s;
        found[i] = bubble_search(g, cand[i], max_len, br, mult, nb, len, s) ? 1u : 0u;
    }
}
// Every edge a search from `begin` can read or write in ANY graph that has a subset of today's valid edges: the edges within max_len
// forward steps and the valid edges into them.  Stamped with `key`; false when the region does not fit the scratch (the caller then
// holds back every higher candidate of the round).
// The lanes of the workgroup take the edges of a level side by side (a region of thousands of edges in a repeat would
// otherwise keep one lane busy for milliseconds while the round waits for it).  Which edges end up stamped, and whether the region
// fits, depends on the region alone (counts per level and in all), not on the order the lanes find them in.
__device__ bool bubble_reach_stamp(const GraphDev &g, int64_t begin, int max_len, int64_t *scratch, const StampTab &owner, unsigned long long key,
                                   unsigned long long round, int reach_max, int *s_cnt /* LDS: [0] edges of the next level, [1] edges seen, [2] overflow */,
                                   unsigned long long tag_flag) {
    if (!g_valid(g, begin)) return true;
    const int lane = threadIdx.x;
    unsigned long long *hash = reinterpret_cast<unsigned long long *>(scratch);
    int64_t *cur = scratch + kReachHash, *nxt = cur + kReachFrontier;
    // hash entries carry the round in their top 24 bits: whatever an earlier round or a search left in the scratch reads as empty
    // (edge ids stay below 2^40), so nothing is cleared
    const unsigned long long tag = (((round & 0x3FFFFFull) + 1) << 40) | tag_flag;     // (tag_flag: what the narrow walk of the same round left reads as stale)
    auto insert = [&](int64_t e) -> bool {           // true = new; slots are claimed at the L2 (other lanes insert at the same time)
        uint32_t h = (uint32_t)(((uint64_t)e * 0x9E3779B97F4A7C15ull) >> 49) & (kReachHash - 1);
        const unsigned long long mine = tag | (unsigned long long)e;
        for (;;) {
            unsigned long long v = __hip_atomic_load(&hash[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == mine) return false;
            if ((v & ~0xFFFFFFFFFFull) != tag) {
                if (__hip_atomic_compare_exchange_strong(&hash[h], &v, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
                if (v == mine) return false;
                continue;                            // another lane took the slot for another edge: look at it again
            }
            h = (h + 1) & (kReachHash - 1);
        }
    };
    if (lane == 0) {
        s_cnt[0] = 0; s_cnt[1] = 1; s_cnt[2] = 0; s_cnt[3] = 0;
        insert(begin);
        __hip_atomic_store(&cur[0], begin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!owner.stamp(begin, key)) s_cnt[2] = 1;
    }
    __syncthreads();
    int n_cur = 1;
    for (int level = 0; level < max_len && n_cur > 0; ++level) {
        for (int i0 = 0; i0 < n_cur; i0 += (int)blockDim.x) {
            const int i = i0 + lane;
            if (i < n_cur) {
                int64_t out[8];
                const int od = d_outgoing(g, __hip_atomic_load(&cur[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), out);
                for (int x = 0; x < od; ++x) {
                    if (!insert(out[x])) continue;
                    const int seen = atomicAdd(&s_cnt[1], 1) + 1, at = atomicAdd(&s_cnt[0], 1);
                    if (seen > reach_max || at >= kReachFrontier) { s_cnt[2] = 1; s_cnt[3] = 1; continue; }   // [3]: the REGION is too large (not the table)
                    __hip_atomic_store(&nxt[at], out[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bool fits = owner.stamp(out[x], key);
                    int64_t in[8];
                    const int id = d_incoming(g, out[x], in);
                    for (int y = 0; y < id; ++y) fits = owner.stamp(in[y], key) && fits;
                    if (!fits) s_cnt[2] = 1;                            // the stamp table is crowded: treated like a region that does not fit
                }
            }
        }
        __syncthreads();
        if (s_cnt[2]) return false;
        n_cur = s_cnt[0];
        __syncthreads();
        if (lane == 0) s_cnt[0] = 0;
        __syncthreads();
        int64_t *t = cur; cur = nxt; nxt = t;
    }
    return true;
}

constexpr unsigned long long kKnownBig = 1ull << 63;   // in a window's position word: the candidate's region did not fit the scratch in an earlier round
constexpr unsigned long long kKnownWide = 1ull << 62;  // ... its region has more than kNarrowMax edges: walked by a whole wave from the start
constexpr unsigned long long kPosMask = ~(kKnownBig | kKnownWide);
constexpr int kNarrowLanes = 8, kNarrowMax = 512;

// The tail of a candidate whose region does not fit (or is known not to): what it can reach in a later graph is not known, so it stamps
// what its search reads TODAY and is marked (unknown[i]).  If nothing below it touches that read set (its check passes), its search at
// its turn is today's search -- same reads, same pop, all inside the stamped set -- and it commits like any other candidate without
// holding anybody back; only when its check FAILS (some lower pending candidate may change what it reads, so what it will read and
// write at its turn is open) does it become the round's barrier: nobody above it commits.  Round 3 raised the barrier at every such
// candidate, whatever its check said: every region that did not fit cost a round (100 M reads, k = 29: thousands of rounds).
__device__ void bubble_reach_failed(const GraphDev &g, const int64_t *cand, uint32_t i, int max_len, int64_t *scratch, size_t per, const StampTab &owner,
                                    unsigned long long key, uint32_t *barrier, uint32_t *unknown) {
    unknown[i] = 1u;
    int mult[kMaxBranches], nb = 0, len = 0;
    SinkStamp s{owner, key, true};
    bubble_search(g, cand[i], max_len, scratch + (size_t)i * per, mult, nb, len, s);
    if (!s.ok) atomicAdd(barrier + 2, 1u);                             // not even that fitted the table: the host shrinks the next window
}

// Most regions are a few dozen edges in levels of one to three: a wave per candidate keeps 60 lanes idle through ~60 levels of dependent
// line fetches and atomics (20 M reads: 17 of the 23 ms of a round).  Eight lanes per candidate, eight candidates per wave: the same walk
// (same stamps, same verdict) with eight times the candidates in flight.  A region of more than kNarrowMax edges is left to the wave-wide
// walk below (list), now and in later rounds (kKnownWide); what this walk stamped of it is a subset of what that one stamps.
__global__ __launch_bounds__(64) void bubble_reach_narrow_kernel(GraphDev g, const int64_t *cand, uint64_t *pos, uint32_t n, int max_len, int64_t *scratch, size_t per,
                                                                 StampTab owner, unsigned long long round, int reach_max, int narrow_max, uint32_t *barrier,
                                                                 uint32_t *wide_list, uint32_t *wide_count, uint32_t *unknown) {
    constexpr int G = kNarrowLanes, NG = 64 / G;
    __shared__ int s_cnt[NG][4];                                       // per candidate: [0] edges of the next level, [1] edges seen, [2] stop, [3] the region is over the limit
    const int lane = threadIdx.x, grp = lane / G, gl = lane % G;
    const uint32_t i = blockIdx.x * NG + grp;
    const bool have = i < n;
    const uint64_t pw = have ? pos[i] : 0ull;
    const bool known_big = (pw & kKnownBig) != 0ull, known_wide = (pw & kKnownWide) != 0ull;
    const int64_t begin = have ? cand[i] : 0;
    const unsigned long long key = (round << 32) | (0xFFFFFFFFull - i);
    const int limit = reach_max < narrow_max ? reach_max : narrow_max;
    enum { WALK, FITS, FAILED, WIDE };
    int state = !have ? FITS : known_big ? FAILED : known_wide ? WIDE : !g_valid(g, begin) ? FITS : WALK;
    int64_t *mine = scratch + (size_t)(have ? i : 0) * per;
    unsigned long long *hash = reinterpret_cast<unsigned long long *>(mine);
    int64_t *cur = mine + kReachHash, *nxt = cur + kReachFrontier;
    const unsigned long long tag = ((round & 0x3FFFFFull) + 1) << 40;
    auto insert = [&](int64_t e) -> bool {           // as in bubble_reach_stamp
        uint32_t h = (uint32_t)(((uint64_t)e * 0x9E3779B97F4A7C15ull) >> 49) & (kReachHash - 1);
        const unsigned long long me = tag | (unsigned long long)e;
        for (;;) {
            unsigned long long v = __hip_atomic_load(&hash[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == me) return false;
            if ((v & ~0xFFFFFFFFFFull) != tag) {
                if (__hip_atomic_compare_exchange_strong(&hash[h], &v, me, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
                if (v == me) return false;
                continue;
            }
            h = (h + 1) & (kReachHash - 1);
        }
    };
    if (gl == 0) {
        s_cnt[grp][0] = 0; s_cnt[grp][1] = 1; s_cnt[grp][2] = 0; s_cnt[grp][3] = 0;
        if (state == WALK) {
            insert(begin);
            __hip_atomic_store(&cur[0], begin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!owner.stamp(begin, key)) s_cnt[grp][2] = 1;
        }
    }
    __syncthreads();
    int n_cur = 1;
    for (int level = 0; level < max_len; ++level) {
        const bool walking = state == WALK && n_cur > 0;
        if (!__any(walking)) break;
        for (int i0 = 0; __any(walking && i0 < n_cur); i0 += G) {
            const int idx = i0 + gl;
            if (walking && idx < n_cur) {
                int64_t out[8];
                const int od = d_outgoing(g, __hip_atomic_load(&cur[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), out);
                for (int x = 0; x < od; ++x) {
                    if (!insert(out[x])) continue;
                    const int seen = atomicAdd(&s_cnt[grp][1], 1) + 1, at = atomicAdd(&s_cnt[grp][0], 1);
                    if (seen > limit || at >= kReachFrontier) { s_cnt[grp][2] = 1; s_cnt[grp][3] = 1; continue; }
                    __hip_atomic_store(&nxt[at], out[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bool fits = owner.stamp(out[x], key);
                    int64_t in[8];
                    const int id = d_incoming(g, out[x], in);
                    for (int y = 0; y < id; ++y) fits = owner.stamp(in[y], key) && fits;
                    if (!fits) s_cnt[grp][2] = 1;
                }
            }
        }
        __syncthreads();
        if (walking) {
            if (s_cnt[grp][2]) state = !s_cnt[grp][3] ? FAILED : (reach_max <= narrow_max ? FAILED : WIDE);
            else n_cur = s_cnt[grp][0];
        }
        __syncthreads();
        if (gl == 0) s_cnt[grp][0] = 0;
        __syncthreads();
        int64_t *t = cur; cur = nxt; nxt = t;
    }
    if (gl != 0 || !have || state == WALK || state == FITS) return;
    if (state == WIDE) {
        if (!known_wide) pos[i] = pw | kKnownWide;
        wide_list[atomicAdd(wide_count, 1u)] = i;
        return;
    }
    if (!known_big && s_cnt[grp][3]) pos[i] = pw | kKnownBig;          // (only with reach_max <= narrow_max: the region is too large for any walk)
    bubble_reach_failed(g, cand, i, max_len, scratch, per, owner, key, barrier, unknown);
}
// the listed candidates (regions of more than kNarrowMax edges), one workgroup of four waves each: the lanes take the edges of a level
// side by side (a region of thousands of edges in a repeat would otherwise keep one lane busy for milliseconds while the round waits)
constexpr int kWideThreads = 256;
__global__ __launch_bounds__(kWideThreads) void bubble_reach_kernel(GraphDev g, const int64_t *cand, uint64_t *pos, int max_len, int64_t *scratch, size_t per,
                                                          StampTab owner, unsigned long long round, int reach_max, uint32_t *barrier, const uint32_t *wide_list,
                                                          const uint32_t *wide_count, uint32_t *unknown) {
    __shared__ int s_cnt[4];
    const uint32_t n_wide = *wide_count;
    for (uint32_t b = blockIdx.x; b < n_wide; b += gridDim.x) {
        const uint32_t i = wide_list[b];
        const unsigned long long key = (round << 32) | (0xFFFFFFFFull - i);
        __syncthreads();                                                // the previous candidate's counters are no longer read
        if (bubble_reach_stamp(g, cand[i], max_len, scratch + (size_t)i * per, owner, key, round, reach_max, s_cnt, 1ull << 63)) continue;
        if (threadIdx.x != 0) continue;
        if (s_cnt[3]) pos[i] |= kKnownBig;
        bubble_reach_failed(g, cand, i, max_len, scratch, per, owner, key, barrier, unknown);
    }
}
// (the search's results: behind the reach walk's hash table and frontiers -- the packed multiplicities can have bits 40 and above set,
// where the table keeps its round tag, so inside the table a later round could take them for live entries: advisor r3)
constexpr size_t kResOffset = (size_t)kReachHash + 2 * kReachFrontier;
__global__ __launch_bounds__(64) void bubble_check_kernel(GraphDev g, const int64_t *cand, uint32_t n, int max_len, int64_t *scratch, size_t per,
                                                          StampTab owner, unsigned long long round, uint32_t *ok, uint32_t *barrier, const uint32_t *unknown) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    if (i > __hip_atomic_load(barrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok[i] = 0; return; }   // (already held back, whatever its stamps say: no search.  Only a
                                                                        // short cut -- the barrier is final when this kernel has ended, the commit reads it then)
    int mult[kMaxBranches], nb = 0, len = 0;
    SinkCheck s{owner, (round << 32) | (0xFFFFFFFFull - i), true};
    int64_t *br = scratch + (size_t)i * per;
    const bool found = bubble_search(g, cand[i], max_len, br, mult, nb, len, s);
    ok[i] = s.ok ? 1u : 0u;
    if (!s.ok && unknown[i]) atomicMin(barrier, i);                     // what this one will read and write at its turn is open: nobody above it commits
    // what the search found stays behind the branches for the commit: a candidate that passed read only edges nobody else of this round
    // writes (the committing candidates' read and write sets are disjoint by the stamp rule), so the search it would run again there,
    // after other candidates' pops, finds exactly this
    if (s.ok) {
        int64_t *res = br + kResOffset;
        res[0] = (int64_t)(found ? 1 : 0) | ((int64_t)nb << 8) | ((int64_t)len << 16);
        for (int b = 0; b < kMaxBranches; b += 2) res[1 + b / 2] = (int64_t)(uint32_t)mult[b] | ((int64_t)(uint32_t)mult[b + 1] << 32);
    }
}
// status (by position in the candidate list): 0 = the search fails now, 1 = popped, 2 = Pop undid itself (goes to the second list,
// assembly_algorithms.cpp:273-277).  keep[i] = 1: still pending after this round.
__global__ __launch_bounds__(64) void bubble_commit_kernel(Dn d, const int64_t *cand, const uint64_t *pos, uint32_t n, int max_len, int64_t *scratch, size_t per,
                                                           const uint32_t *ok, const uint32_t *barrier, unsigned long long *marked, uint32_t *status,
                                                           uint32_t *keep, uint32_t *n_done) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    if (!ok[i] || i > *barrier) { keep[i] = 1; return; }
    int mult[kMaxBranches];
    int64_t *br = scratch + (size_t)i * per;
    const int64_t *res = br + kResOffset;                                       // left by the check kernel (see there)
    const int nb = (int)((res[0] >> 8) & 255), len = (int)(res[0] >> 16);
    for (int b = 0; b < kMaxBranches; b += 2) { mult[b] = (int)(uint32_t)res[1 + b / 2]; mult[b + 1] = (int)(uint32_t)(res[1 + b / 2] >> 32); }
    uint32_t st = 0;
    if (res[0] & 1) st = bubble_pop(d, marked, br, mult, nb, len, max_len) ? 1u : 2u;
    status[pos[i] & kPosMask] = st;
    keep[i] = 0;
    atomicAdd(n_done, 1u);
}
__global__ __launch_bounds__(256) void window_fill_kernel(const int64_t *cand, uint64_t from, uint32_t take, uint32_t at, int64_t *win, uint64_t *pos) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < take) { win[at + i] = cand[from + i]; pos[at + i] = from + i; }
}
__global__ __launch_bounds__(256) void window_keep_kernel(const int64_t *win, const uint64_t *pos, const uint32_t *keep, const uint64_t *base, uint32_t n,
                                                          int64_t *win2, uint64_t *pos2) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && keep[i]) { win2[base[i]] = win[i]; pos2[base[i]] = pos[i]; }
}
__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t *p, uint32_t n, uint32_t v) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ __launch_bounds__(256) void flag_equals_kernel(const uint32_t *status, uint64_t n, uint32_t want, uint32_t *flag, unsigned long long *count) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t f = status[i] == want;
    flag[i] = f;
    if (f && count) atomicAdd(count, 1ull);
}

// ---- unitigs --------------------------------------------------------------------------------------------------------------------
struct PathRec {          // (the end edge of path p is ends[p])
    int64_t start, rc_start;
    int64_t target;       // path whose suffix the walk from RC(end) locks (-1: RC(end) is not on a path that ends)
    int64_t next;         // next claim on the same target
    uint32_t length;      // edges
    uint32_t extra_depth; // sum of the edge multiplicities (1 or 2 each) minus length
    uint32_t dist;        // edges from RC(end) to that path's end
    uint32_t pad;
};
static_assert(sizeof(PathRec) == 48, "path record");

// path id of an end edge: its rank among the end edges (the mask and its prefix sums are what built the ascending list)
__device__ __forceinline__ int64_t path_of_end(const unsigned long long *end_mask, const uint64_t *end_base, int64_t e) {
    const unsigned long long m = end_mask[e >> 6];
    return (int64_t)end_base[e >> 6] + __popcll(m & ((1ull << (e & 63)) - 1ull));
}

// The walk back from the end edge (unitig_graph.cpp:229-239): start, length, depth.  Nothing is written per edge.
__global__ __launch_bounds__(64) void unitig_walk_kernel(GraphDev g, const int64_t *ends, uint64_t n, PathRec *rec) {
    const uint64_t pid = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (pid >= n) return;
    const int64_t e = ends[pid];
    int64_t cur = e, p;
    uint64_t depth = (uint64_t)multiplicity(g, e), length = 1;
    while ((p = prev_simple(g, cur)) != -1) {       // never a cycle: e has no simple successor
        cur = p;
        depth += (uint64_t)multiplicity(g, cur);
        ++length;
    }
    PathRec r;
    r.start = cur; r.length = (uint32_t)length; r.extra_depth = (uint32_t)(depth - length);
    r.rc_start = -1; r.target = -1; r.dist = 0; r.next = -1; r.pad = length > 0xFFFFFFFFull ? 1u : 0u;   // (pad != 0: a path of 2^32 edges, reported by the host)
    rec[pid] = r;
}
// unitig_graph.cpp:241-273: the walk that locks forwards from RC(end) ends at the end edge of the path RC(end) lies on -- walked here
// the same way (an edge of a pure cycle lies on no path that ends: nothing to claim; the walk is cut when it comes round)
__global__ __launch_bounds__(64) void unitig_claim_kernel(GraphDev g, const int64_t *ends, uint64_t n, PathRec *rec, const unsigned long long *end_mask,
                                                          const uint64_t *end_base, int64_t *head) {
    const uint64_t pid = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (pid >= n) return;
    const int64_t r = edge_reverse_complement(g, ends[pid]);
    rec[pid].rc_start = r;
    if (r < 0 || !g_valid(g, r)) return;
    int64_t cur = r, nx;
    uint32_t dist = 0;
    while ((nx = next_simple(g, cur)) != -1) {
        cur = nx;
        ++dist;
        if (cur == r) return;                        // a cycle of simple edges
    }
    const int64_t t = path_of_end(end_mask, end_base, cur);
    rec[pid].target = t;
    rec[pid].dist = dist;
    rec[pid].next = (int64_t)atomicExch((unsigned long long *)&head[t], (unsigned long long)pid);
}

// state: 0 = not known yet, 1 = processed (its edges get locked by its own walk), 2 = skipped at `marked.try_lock(edge_idx)`
__global__ __launch_bounds__(256) void unitig_resolve_kernel(const int64_t *ends, const PathRec *rec, const int64_t *head, uint64_t n, uint32_t *state,
                                                             unsigned long long *undecided) {
    const uint64_t pid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (pid >= n || __hip_atomic_load(&state[pid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    bool pending = false, skipped = false;
    for (int64_t q = head[pid]; q >= 0; q = rec[q].next) {
        if ((uint64_t)q >= pid) continue;            // (ends ascend with the path id: an earlier end edge = a lower id)
        const uint32_t s = __hip_atomic_load(&state[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s == 1) { skipped = true; break; }
        if (s == 0) pending = true;
    }
    if (skipped) __hip_atomic_store(&state[pid], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (!pending) __hip_atomic_store(&state[pid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else atomicAdd(undecided, 1ull);
}

__global__ __launch_bounds__(64) void unitig_decide_kernel(GraphDev g, const int64_t *ends, const PathRec *rec, const int64_t *head, const uint32_t *state, uint64_t n,
                                                           int min_contig, uint32_t *emit_len) {
    const uint64_t pid = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (pid >= n) return;
    emit_len[pid] = 0;
    if (state[pid] != 1) return;
    const PathRec r = rec[pid];
    const int64_t end = ends[pid];
    bool add = true;
    if (r.target >= 0) {       // was RC(end) locked when this path was processed? (unitig_graph.cpp:245)
        const int64_t t = r.target;
        bool locked = (uint64_t)t == pid || (state[t] == 1 && (uint64_t)t < pid);
        for (int64_t q = head[t]; q >= 0 && !locked; q = rec[q].next)
            locked = (uint64_t)q != pid && state[q] == 1 && (uint64_t)q < pid && rec[q].dist > r.dist;
        if (locked) {
            const int64_t rc_end = edge_reverse_complement(g, r.start);
            const int64_t a = end > r.start ? end : r.start, b = r.rc_start > rc_end ? r.rc_start : rc_end;
            if (a < b) add = false;                                                  // :248-252
        }
    }
    const uint32_t len = r.length + (uint32_t)g.k;
    if (add && (int)len >= min_contig) emit_len[pid] = len;          // (a contig is at least k + 1 characters: 0 = not emitted)
}

struct ContigMeta { int64_t depth; uint32_t length, len; int32_t flag; uint32_t pad; uint64_t offset; };

// VertexToDNAString (unitig_graph.cpp:80-112) for the paths [p0, p0 + n) that are emitted: the start node's k symbols, then the W symbol
// of every edge of the path -- a second walk back from the end edge, writing from the end of the string; then the smaller of the label
// and its reverse complement (WriteContig, :134-150) and the record of the contig.  off / idx: exclusive sums of emit_len / of
// (emit_len > 0) over the chunk.
__global__ __launch_bounds__(64) void unitig_write_kernel(GraphDev g, const int64_t *ends, const PathRec *rec, const uint32_t *emit_len, const uint64_t *idx,
                                                          const uint64_t *off, uint64_t p0, uint64_t n, ContigMeta *meta, char *text) {
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n || !emit_len[p0 + i]) return;
    const uint64_t pid = p0 + i;
    const PathRec r = rec[pid];
    const int k = g.k;
    const uint32_t len = r.length + (uint32_t)k;
    char *s = text + off[i];
    int64_t cur = ends[pid];
    for (uint32_t at = len; at > (uint32_t)k;) {                      // the edge `dist` steps before the end is character len - 1 - dist
        const int w = g_W(g, cur);
        s[--at] = "ACGT"[(w > 4 ? w - 4 : w) - 1];
        if (at > (uint32_t)k) cur = prev_simple(g, cur);
    }
    uint8_t lab[kMaxK + 1];
    d_label(g, r.start, lab);
    for (int j = 0; j < k; ++j) s[j] = "ACGT"[lab[j] - 1];
    auto comp = [](char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; };
    int cmp = 0;
    for (uint32_t j = 0; j < len && cmp == 0; ++j) {
        const char a = s[j], b = comp(s[len - 1 - j]);
        cmp = a < b ? -1 : a > b ? 1 : 0;
    }
    if (cmp > 0)
        for (uint32_t a = 0, b = len - 1; a <= b && b != 0xFFFFFFFFu; ++a, --b) {
            const char x = comp(s[a]), y = comp(s[b]);
            s[a] = y; s[b] = x;
        }
    int64_t t[8];
    ContigMeta m;
    m.depth = (int64_t)r.length + (int64_t)r.extra_depth; m.length = r.length; m.len = len; m.pad = 0; m.offset = off[i];
    m.flag = (d_incoming(g, r.start, t) == 0 && d_outgoing(g, ends[pid], t) == 0) ? 1 : 0;   // contig_flag::kIsolated
    meta[idx[i]] = m;
}
__global__ __launch_bounds__(256) void nonzero_flag_kernel(const uint32_t *v, uint64_t n, uint32_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flag[i] = v[i] ? 1u : 0u;
}
__global__ __launch_bounds__(256) void any_pad_kernel(const PathRec *rec, uint64_t n, unsigned long long *count) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && rec[i].pad) atomicAdd(count, 1ull);
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
struct Work {
    mgta_ctx *ctx;
    hipStream_t st;
    Dn d;
    DevBuf mask, count, base, tmp, total;
    uint64_t *live() { return &ctx->live_bytes; }
    uint64_t *peak() { return &ctx->peak_bytes; }
};

static bool verbose() { static const bool v = getenv("MGTA_DENOVO_VERBOSE") != nullptr; return v; }
static void note(Work &w, const char *fmt, ...) {      // MGTA_DENOVO_VERBOSE: one line per step on stderr, after the stream has drained
    if (!verbose()) return;
    (void)hipStreamSynchronize(w.st);
    static const auto t0 = std::chrono::steady_clock::now();
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "[denovo %8.3f s] ", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
}

static uint64_t read_u64(Work &w, const void *p) {
    uint64_t v = 0;
    MGTA_HIP_CHECK(hipMemcpyAsync(&v, p, 8, hipMemcpyDeviceToHost, w.st));
    MGTA_HIP_CHECK(hipStreamSynchronize(w.st));
    return v;
}

template <class Pred>
static uint64_t edges_where(Work &w, Pred pred, DevBuf &list) {   // ascending ids of the edges that satisfy pred
    const GraphDev &g = w.d.g;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(edge_mask_kernel<Pred>), dim3((unsigned)std::min<uint64_t>((g.n_lines + 3) / 4, 1u << 22)), dim3(256), 0, w.st, g, pred,
                       w.mask.as<unsigned long long>(), w.count.as<uint32_t>());
    MGTA_HIP_CHECK(hipGetLastError());
    exclusive_scan_u32(w.st, w.count.as<uint32_t>(), g.n_lines, w.base.as<uint64_t>(), w.tmp.as<uint64_t>(), w.total.as<uint64_t>());
    const uint64_t n = read_u64(w, w.total.p);
    list.alloc(n * 8 + 64, w.live(), w.peak());
    if (n)
        hipLaunchKernelGGL(mask_expand_kernel, dim3((unsigned)((g.n_lines + 255) / 256)), dim3(256), 0, w.st, w.mask.as<unsigned long long>(),
                           w.base.as<uint64_t>(), g.n_lines, list.as<int64_t>());
    return n;
}

static uint64_t compact_list(Work &w, const DevBuf &in, const DevBuf &flag, uint64_t n, DevBuf &out) {
    if (n == 0) { out.alloc(64, w.live(), w.peak()); return 0; }
    DevBuf base, tmp;
    base.alloc(n * 8, w.live(), w.peak());
    tmp.alloc(scan_tmp_elems(n) * 8, w.live(), w.peak());
    exclusive_scan_u32(w.st, flag.as<uint32_t>(), n, base.as<uint64_t>(), tmp.as<uint64_t>(), w.total.as<uint64_t>());
    const uint64_t m = read_u64(w, w.total.p);
    out.alloc(m * 8 + 64, w.live(), w.peak());
    if (m)
        hipLaunchKernelGGL(list_compact_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, w.st, in.as<int64_t>(), flag.as<uint32_t>(),
                           base.as<uint64_t>(), n, out.as<int64_t>());
    MGTA_HIP_CHECK(hipStreamSynchronize(w.st));
    return m;
}

static uint64_t remove_tips(Work &w, int max_tip_len) {   // assembly_algorithms.cpp:161-183
    const GraphDev &g = w.d.g;
    DevBuf removed, counter;
    removed.alloc((g.n_lines + 1) * 8, w.live(), w.peak());
    counter.alloc(64, w.live(), w.peak());
    MGTA_HIP_CHECK(hipMemsetAsync(removed.p, 0, (g.n_lines + 1) * 8, w.st));
    MGTA_HIP_CHECK(hipMemsetAsync(counter.p, 0, 64, w.st));
    auto trim = [&](int len) {
        DevBuf starts;
        uint64_t n = edges_where(w, PredTipStartOut{removed.as<unsigned long long>()}, starts);
        if (n)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(trim_walk_kernel<true>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, w.st, g, starts.as<int64_t>(), n, len,
                               removed.as<unsigned long long>(), counter.as<unsigned long long>());
        n = edges_where(w, PredTipStartIn{removed.as<unsigned long long>()}, starts);
        if (n)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(trim_walk_kernel<false>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, w.st, g, starts.as<int64_t>(), n, len,
                               removed.as<unsigned long long>(), counter.as<unsigned long long>());
        hipLaunchKernelGGL(trim_delete_kernel, dim3((unsigned)((g.n_lines + 255) / 256)), dim3(256), 0, w.st, w.d, removed.as<unsigned long long>());
        MGTA_HIP_CHECK(hipStreamSynchronize(w.st));
        note(w, "trim len %d: %llu in-degree-0 starts", len, (unsigned long long)n);
    };
    for (int len = 2; len < max_tip_len; len *= 2) trim(len);
    trim(max_tip_len);
    return read_u64(w, counter.p);
}

struct BubbleWork {
    DevBuf scratch, stamp_key, stamp_val, marked, status, win[2], pos[2], ok, keep, base, tmp, small, unknown;
    int64_t *scratch_p = nullptr;   // the per-candidate scratch: `scratch`, or a buffer of the context's pool that nobody uses during a denovo
    uint64_t scratch_bytes = 0;
    uint64_t stamp_mask = 0;
    int64_t n_crowded = 0;   // rounds in which the stamp table turned a candidate away
    size_t per = 0;          // int64 of scratch per candidate
    uint32_t window = 0;
    int reach_max = kReachMax;
    int narrow_max = kNarrowMax;
    uint64_t round = 0;
};

// the ordered loop `for each candidate: Search, then Pop` (assembly_algorithms.cpp:266-279 and :283-292)
static void pop_in_order(Work &w, BubbleWork &b, const DevBuf &cand, uint64_t n, int max_len, int64_t &n_rounds) {
    const GraphDev &g = w.d.g;
    uint64_t p = 0;           // next candidate of the list not yet in a window
    uint32_t carry = 0;       // pending candidates kept from the previous window (front of win[cur])
    int cur = 0;
    uint32_t want = std::min<uint32_t>(4096, b.window);
    uint32_t *barrier = b.small.as<uint32_t>(), *n_done = b.small.as<uint32_t>() + 1;
    uint32_t cap = b.window;      // windows shrink while the stamp table overflows (a crowded table holds back what it could not stamp)
    while (carry > 0 || p < n) {
        // (a carry larger than the round is cut: the candidates beyond it stay pending, in order.  A round that committed little is followed
        // by a small one whatever is pending: behind a region that does not fit its scratch only the lowest candidates can commit, and a
        // round costs what its window costs -- 500 M reads spent 19 727 rounds of 40-86 k candidates committing a few hundred each)
        const uint32_t m = (uint32_t)std::min<uint64_t>(std::max<uint32_t>(1u, std::min<uint32_t>(want, std::max(cap, 1u))), carry + (n - p));
        const uint32_t take = m > carry ? m - carry : 0, shed = carry > m ? carry - m : 0;
        if (take) hipLaunchKernelGGL(window_fill_kernel, dim3((take + 255) / 256), dim3(256), 0, w.st, cand.as<int64_t>(), p, take, carry, b.win[cur].as<int64_t>(),
                                     b.pos[cur].as<uint64_t>());
        p += take;
        ++b.round; ++n_rounds;
        if (b.round >= 0xFFFFFEull) { set_error("mgta_denovo: more than 2^24 bubble rounds"); throw HipError{MGTA_EUNSUPPORTED}; }
        const int64_t *c = b.win[cur].as<int64_t>();
        const uint32_t init[4] = {0xFFFFFFFFu, 0u, 0u, 0u};
        MGTA_HIP_CHECK(hipMemcpyAsync(b.small.p, init, 16, hipMemcpyHostToDevice, w.st));
        MGTA_HIP_CHECK(hipMemsetAsync(b.unknown.p, 0, (size_t)m * 4, w.st));
        const StampTab tab{b.stamp_key.as<unsigned long long>(), b.stamp_val.as<unsigned long long>(), b.stamp_mask, (unsigned long long)(b.round + 1) << 40};
        // (the list of the wide candidates lives in `ok` until the check kernel writes that; its length in the fourth counter)
        hipLaunchKernelGGL(bubble_reach_narrow_kernel, dim3((m + 64 / kNarrowLanes - 1) / (64 / kNarrowLanes)), dim3(64), 0, w.st, g, c, b.pos[cur].as<uint64_t>(), m, max_len,
                           b.scratch_p, b.per, tab, (unsigned long long)b.round, b.reach_max, b.narrow_max, barrier, b.ok.as<uint32_t>(), barrier + 3, b.unknown.as<uint32_t>());
        hipLaunchKernelGGL(bubble_reach_kernel, dim3(std::min<uint32_t>(m, 8192u)), dim3(kWideThreads), 0, w.st, g, c, b.pos[cur].as<uint64_t>(), max_len, b.scratch_p, b.per,
                           tab, (unsigned long long)b.round, b.reach_max, barrier, b.ok.as<uint32_t>(), barrier + 3, b.unknown.as<uint32_t>());
        hipLaunchKernelGGL(bubble_check_kernel, dim3((m + 63) / 64), dim3(64), 0, w.st, g, c, m, max_len, b.scratch_p, b.per, tab,
                           (unsigned long long)b.round, b.ok.as<uint32_t>(), barrier, b.unknown.as<uint32_t>());
        hipLaunchKernelGGL(bubble_commit_kernel, dim3((m + 63) / 64), dim3(64), 0, w.st, w.d, c, b.pos[cur].as<uint64_t>(), m, max_len, b.scratch_p, b.per,
                           b.ok.as<uint32_t>(), barrier, b.marked.as<unsigned long long>(), b.status.as<uint32_t>(), b.keep.as<uint32_t>(), n_done);
        if (shed) {      // the candidates beyond the cut stay pending, behind the ones this round keeps
            hipLaunchKernelGGL(fill_u32_kernel, dim3((shed + 255) / 256), dim3(256), 0, w.st, b.keep.as<uint32_t>() + m, shed, 1u);
        }
        const uint32_t mm = m + shed;
        exclusive_scan_u32(w.st, b.keep.as<uint32_t>(), mm, b.base.as<uint64_t>(), b.tmp.as<uint64_t>(), w.total.as<uint64_t>());
        hipLaunchKernelGGL(window_keep_kernel, dim3((mm + 255) / 256), dim3(256), 0, w.st, c, b.pos[cur].as<uint64_t>(), b.keep.as<uint32_t>(), b.base.as<uint64_t>(), mm,
                           b.win[cur ^ 1].as<int64_t>(), b.pos[cur ^ 1].as<uint64_t>());
        uint32_t flags[4];
        MGTA_HIP_CHECK(hipMemcpyAsync(flags, b.small.p, 16, hipMemcpyDeviceToHost, w.st));
        carry = (uint32_t)read_u64(w, w.total.p);
        cur ^= 1;
        const uint32_t done = mm - carry;
        if (flags[2]) { cap = std::max<uint32_t>(1, m / 2); ++b.n_crowded; }   // some candidate could not stamp even what it reads: fewer candidates share the table next time
        else if (cap < b.window) cap = std::min<uint32_t>(b.window, cap * 2);
        if (done == 0 && !(flags[2] && m > 1)) { set_error("mgta_denovo: a bubble round committed nothing"); throw HipError{MGTA_EINTERNAL}; }   // the lowest always commits
        want = std::min<uint32_t>(b.window, std::max<uint32_t>(std::min<uint32_t>(4096, b.window), 4 * std::max(done, 1u)));
        if ((n_rounds & 15) == 1) note(w, "bubble round %lld: window %u, committed %u, %llu of %llu taken", (long long)n_rounds, m, done, (unsigned long long)p, (unsigned long long)n);     // a round that commits few (a region that does not fit the reach scratch) shrinks the next
    }
}

static uint64_t pop_bubbles(Work &w, int64_t &n_rounds, int64_t &n_candidates) {   // assembly_algorithms.cpp:245-301
    const GraphDev &g = w.d.g;
    const int max_len = g.k * 2 + 4;
    BubbleWork b;
    DevBuf branching, found, cand, flag, again, counter;
    uint64_t nb = edges_where(w, PredBranching{}, branching);
    b.per = kResOffset + 1 + kMaxBranches / 2;                         // reach hash (the search's branches overlay it: edge ids read as stale entries) | frontiers | results
    static_assert((size_t)kMaxBranches * (2 * kMaxK + 4) <= (size_t)kReachHash, "the branches of a search fit the hash region");
    // candidates per round: as many as an eighth of the free device memory (2 .. 32 GB) holds scratch for -- a round costs ~24 ms of
    // launches and look-ups whatever it commits, 100 M reads took 513 rounds with 8 GB
    size_t free_b = 0, total_b = 0;
    MGTA_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    uint64_t scratch_cap = 32ull << 30, borrow_cap = 128ull << 30;
    if (const char *e = getenv("MGTA_DENOVO_SCRATCH_GB")) scratch_cap = borrow_cap = (uint64_t)std::max(1, atoi(e)) << 30;   // (measurements)
    uint64_t scratch_budget = std::min<uint64_t>(scratch_cap, std::max<uint64_t>(2ull << 30, (uint64_t)free_b / 8));
    // ... but never more than the branching edges can use (a window slot per candidate, a read-only search per branching edge for as many
    // threads as the device holds at once), and obtained ONCE: 2 M reads: 32 GB of scratch for 263 k candidates cost 1.0-1.8 s of
    // hipMalloc per `denovo`, five to nine times the rounds themselves.  The build's key buffers sit idle in the context's pool while a
    // denovo runs (the worker keeps them between the steps): the larger of the two is borrowed when it holds at least a quarter of that --
    // and then ALL of it may be used (up to 128 GB): it costs nothing to obtain.  100 M reads, k = 29 / 35: windows of 228 k instead of
    // 61 k candidates, 130 / 88 rounds instead of 319 / 271, 7.6 / 6.8 s instead of 8.5 / 8.4 s (a round costs mostly what its window costs).
    const uint64_t per_find_b = (uint64_t)kMaxBranches * (uint64_t)max_len * 8;
    const uint64_t need = std::max<uint64_t>(256ull << 20, std::max<uint64_t>(std::min<uint64_t>((nb + 63) / 64 * 64, 1ull << 20) * per_find_b,
                                                                             std::min<uint64_t>((nb + 63) / 64 * 64 + 64, kBubbleWindowMax) * (uint64_t)b.per * 8));
    scratch_budget = std::min(scratch_budget, need);
    int64_t *scratch_p = nullptr;
    for (size_t slot = 4; slot <= 5 && slot < w.ctx->pool.size(); ++slot) {      // S_KEYS_A / S_KEYS_B of sdbg_build.hip
        DevBuf &kb = w.ctx->pool[slot];
        if (kb.p && kb.bytes >= scratch_budget / 4 && kb.bytes >= (256ull << 20) && (!scratch_p || kb.bytes > b.scratch_bytes)) {
            scratch_p = kb.as<int64_t>(); b.scratch_bytes = kb.bytes;
        }
    }
    if (getenv("MGTA_DENOVO_OWN_SCRATCH")) scratch_p = nullptr;          // (measurements)
    if (scratch_p) scratch_budget = std::min<uint64_t>(std::min<uint64_t>(borrow_cap, need), b.scratch_bytes);
    b.window = (uint32_t)std::min<uint64_t>(kBubbleWindowMax, std::max<uint64_t>(4096, scratch_budget / (b.per * 8)));
    // test knobs: tiny windows exercise the carry of pending candidates, a tiny reach limit the hold-back of a region that does not fit
    if (const char *e = getenv("MGTA_DENOVO_WINDOW")) b.window = (uint32_t)std::max(64, atoi(e)) & ~63u;
    if (const char *e = getenv("MGTA_DENOVO_REACH_MAX")) b.reach_max = std::min(kReachMax, std::max(1, atoi(e)));
    if (const char *e = getenv("MGTA_DENOVO_NARROW_MAX")) b.narrow_max = std::min(kReachFrontier, std::max(1, atoi(e)));   // (tiny: every region takes the wave-wide walk)
    if (scratch_p && (uint64_t)b.window * b.per * 8 <= b.scratch_bytes) b.scratch_p = scratch_p;
    else { b.scratch.alloc((size_t)b.window * b.per * 8, w.live(), w.peak()); b.scratch_p = b.scratch.as<int64_t>(); }
    found.alloc(nb * 4 + 64, w.live(), w.peak());
    if (nb) {
        // the read-only searches need kMaxBranches * max_len words each, not a reach table: as many threads as the scratch holds of those
        // (10 M reads: 8.5 M branching edges took 1 s with one thread per window slot)
        const size_t per_find = (size_t)kMaxBranches * max_len;
        const uint64_t threads = std::min<uint64_t>((nb + 63) / 64 * 64, ((uint64_t)b.window * b.per / per_find) / 64 * 64);
        hipLaunchKernelGGL(bubble_find_kernel, dim3((unsigned)(threads / 64)), dim3(64), 0, w.st, g, branching.as<int64_t>(), nb, max_len, b.scratch_p,
                           per_find, found.as<uint32_t>());
    }
    const uint64_t nc = compact_list(w, branching, found, nb, cand);
    n_candidates = (int64_t)nc;
    note(w, "bubbles: %llu branching edges, %llu candidates, window %u", (unsigned long long)nb, (unsigned long long)nc, b.window);
    if (nc == 0) return 0;
    // the stamp table: one slot of 16 bytes per edge up to 2^29 slots (8 GB, whatever the size of the graph beyond that) -- a window of
    // 29 k candidates stamps up to 16 k edges each, a few hundred as a rule; what does not fit is held back and the next window is
    // smaller.  MGTA_DENOVO_STAMP_LOG2 (tests): a tiny table exercises that path.
    int stamp_log = 22;
    while (stamp_log < 29 && (1ull << stamp_log) < (uint64_t)g.size) ++stamp_log;
    if (const char *e = getenv("MGTA_DENOVO_STAMP_LOG2")) stamp_log = std::min(30, std::max(8, atoi(e)));
    b.stamp_mask = (1ull << stamp_log) - 1;
    b.stamp_key.alloc((b.stamp_mask + 1) * 8, w.live(), w.peak());
    b.stamp_val.alloc((b.stamp_mask + 1) * 8, w.live(), w.peak());
    b.marked.alloc((g.n_lines + 1) * 8, w.live(), w.peak());
    b.status.alloc(nc * 4 + 64, w.live(), w.peak());
    for (int i = 0; i < 2; ++i) { b.win[i].alloc((size_t)b.window * 8, w.live(), w.peak()); b.pos[i].alloc((size_t)b.window * 8, w.live(), w.peak()); }
    b.ok.alloc((size_t)b.window * 4, w.live(), w.peak());
    b.keep.alloc((size_t)b.window * 4, w.live(), w.peak());
    b.unknown.alloc((size_t)b.window * 4, w.live(), w.peak());
    b.base.alloc((size_t)b.window * 8, w.live(), w.peak());
    b.tmp.alloc(scan_tmp_elems(b.window) * 8, w.live(), w.peak());
    b.small.alloc(64, w.live(), w.peak());
    flag.alloc(nc * 4 + 64, w.live(), w.peak());
    counter.alloc(64, w.live(), w.peak());
    MGTA_HIP_CHECK(hipMemsetAsync(b.stamp_key.p, 0, (b.stamp_mask + 1) * 8, w.st));
    MGTA_HIP_CHECK(hipMemsetAsync(b.stamp_val.p, 0, (b.stamp_mask + 1) * 8, w.st));
    MGTA_HIP_CHECK(hipMemsetAsync(b.marked.p, 0, (g.n_lines + 1) * 8, w.st));
    MGTA_HIP_CHECK(hipMemsetAsync(counter.p, 0, 64, w.st));
    pop_in_order(w, b, cand, nc, max_len, n_rounds);
    hipLaunchKernelGGL(flag_equals_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, w.st, b.status.as<uint32_t>(), nc, 1u, flag.as<uint32_t>(),
                       counter.as<unsigned long long>());
    hipLaunchKernelGGL(flag_equals_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, w.st, b.status.as<uint32_t>(), nc, 2u, flag.as<uint32_t>(),
                       (unsigned long long *)nullptr);
    const uint64_t na = compact_list(w, cand, flag, nc, again);
    if (na) {
        pop_in_order(w, b, again, na, max_len, n_rounds);
        hipLaunchKernelGGL(flag_equals_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, w.st, b.status.as<uint32_t>(), na, 1u, flag.as<uint32_t>(),
                           counter.as<unsigned long long>());
    }
    note(w, "bubbles: %lld rounds, the stamp table (2^%d slots) was crowded in %lld of them", (long long)n_rounds, stamp_log, (long long)b.n_crowded);
    return read_u64(w, counter.p);
}

struct Contigs {
    std::vector<ContigMeta> meta;
    std::vector<char> text;
    int64_t n_paths = 0, n_sweeps = 0;
};

static void unitigs(Work &w, int min_contig, Contigs &out) {
    const GraphDev &g = w.d.g;
    DevBuf ends, rec, head, state, counter, emit_len;
    const uint64_t n = edges_where(w, PredPathEnd{}, ends);         // w.mask / w.base keep the mask and its prefix sums: the rank of an end edge is its path id
    out.n_paths = (int64_t)n;
    if (n == 0) return;
    rec.alloc(n * sizeof(PathRec), w.live(), w.peak());
    head.alloc(n * 8, w.live(), w.peak());
    state.alloc(n * 4, w.live(), w.peak());
    counter.alloc(64, w.live(), w.peak());
    MGTA_HIP_CHECK(hipMemsetAsync(head.p, 0xFF, n * 8, w.st));
    MGTA_HIP_CHECK(hipMemsetAsync(state.p, 0, n * 4, w.st));
    MGTA_HIP_CHECK(hipMemsetAsync(counter.p, 0, 64, w.st));
    const unsigned grid64 = (unsigned)((n + 63) / 64), grid256 = (unsigned)((n + 255) / 256);
    if ((n + 63) / 64 >= (1ull << 31)) { set_error("mgta_denovo: %llu paths exceed one launch", (unsigned long long)n); throw HipError{MGTA_EUNSUPPORTED}; }
    hipLaunchKernelGGL(unitig_walk_kernel, dim3(grid64), dim3(64), 0, w.st, g, ends.as<int64_t>(), n, rec.as<PathRec>());
    hipLaunchKernelGGL(any_pad_kernel, dim3(grid256), dim3(256), 0, w.st, rec.as<PathRec>(), n, counter.as<unsigned long long>());
    if (read_u64(w, counter.p)) { set_error("mgta_denovo: a simple path of 2^32 edges or more"); throw HipError{MGTA_EUNSUPPORTED}; }
    note(w, "unitigs: %llu paths walked", (unsigned long long)n);
    hipLaunchKernelGGL(unitig_claim_kernel, dim3(grid64), dim3(64), 0, w.st, g, ends.as<int64_t>(), n, rec.as<PathRec>(), w.mask.as<unsigned long long>(),
                       w.base.as<uint64_t>(), head.as<int64_t>());
    for (;;) {
        MGTA_HIP_CHECK(hipMemsetAsync(counter.p, 0, 8, w.st));
        hipLaunchKernelGGL(unitig_resolve_kernel, dim3(grid256), dim3(256), 0, w.st, ends.as<int64_t>(), rec.as<PathRec>(), head.as<int64_t>(), n, state.as<uint32_t>(),
                           counter.as<unsigned long long>());
        ++out.n_sweeps;
        if (read_u64(w, counter.p) == 0) break;
    }
    note(w, "unitigs: claims resolved in %lld sweeps", (long long)out.n_sweeps);
    emit_len.alloc(n * 4, w.live(), w.peak());
    hipLaunchKernelGGL(unitig_decide_kernel, dim3(grid64), dim3(64), 0, w.st, g, ends.as<int64_t>(), rec.as<PathRec>(), head.as<int64_t>(), state.as<uint32_t>(), n, min_contig,
                       emit_len.as<uint32_t>());
    if (getenv("MGTA_DENOVO_DEBUG")) {      // the path table, for comparing with the oracle's model of this step
        std::vector<PathRec> h(n);
        std::vector<uint32_t> hs(n), he(n);
        std::vector<int64_t> hend(n);
        MGTA_HIP_CHECK(hipMemcpy(h.data(), rec.p, n * sizeof(PathRec), hipMemcpyDeviceToHost));
        MGTA_HIP_CHECK(hipMemcpy(hs.data(), state.p, n * 4, hipMemcpyDeviceToHost));
        MGTA_HIP_CHECK(hipMemcpy(he.data(), emit_len.p, n * 4, hipMemcpyDeviceToHost));
        MGTA_HIP_CHECK(hipMemcpy(hend.data(), ends.p, n * 8, hipMemcpyDeviceToHost));
        fprintf(stderr, "device: %llu paths\n", (unsigned long long)n);
        for (uint64_t i = 0; i < n; ++i)
            fprintf(stderr, "device p=%llu end=%lld start=%lld len=%u rc=%lld target=%lld dist=%u state=%u emit=%u\n", (unsigned long long)i, (long long)hend[i],
                    (long long)h[i].start, h[i].length, (long long)h[i].rc_start, (long long)h[i].target, h[i].dist, hs[i], he[i] ? 1u : 0u);
    }
    MGTA_HIP_CHECK(hipStreamSynchronize(w.st));
    head.release(); state.release();                                   // (only the records, the end edges and the verdicts are needed from here on)
    // the contigs leave in pieces of 2^25 paths: offsets, characters and records of one piece at a time
    const uint64_t kPiece = 1ull << 25;
    DevBuf flag, idx, off, tmp, meta, text;
    const uint64_t pn_max = std::min(n, kPiece);
    flag.alloc(pn_max * 4, w.live(), w.peak());
    idx.alloc(pn_max * 8, w.live(), w.peak());
    off.alloc(pn_max * 8, w.live(), w.peak());
    tmp.alloc(scan_tmp_elems(pn_max) * 8, w.live(), w.peak());
    uint64_t char_base = 0;
    for (uint64_t p0 = 0; p0 < n; p0 += kPiece) {
        const uint64_t pn = std::min(kPiece, n - p0);
        hipLaunchKernelGGL(nonzero_flag_kernel, dim3((unsigned)((pn + 255) / 256)), dim3(256), 0, w.st, emit_len.as<uint32_t>() + p0, pn, flag.as<uint32_t>());
        exclusive_scan_u32(w.st, flag.as<uint32_t>(), pn, idx.as<uint64_t>(), tmp.as<uint64_t>(), w.total.as<uint64_t>());
        const uint64_t n_contigs = read_u64(w, w.total.p);
        exclusive_scan_u32(w.st, emit_len.as<uint32_t>() + p0, pn, off.as<uint64_t>(), tmp.as<uint64_t>(), w.total.as<uint64_t>());
        const uint64_t n_chars = read_u64(w, w.total.p);
        if (n_contigs == 0) continue;
        if (meta.bytes < n_contigs * sizeof(ContigMeta)) meta.alloc(n_contigs * sizeof(ContigMeta), w.live(), w.peak());
        if (text.bytes < n_chars + 64) text.alloc(n_chars + 64, w.live(), w.peak());
        hipLaunchKernelGGL(unitig_write_kernel, dim3((unsigned)((pn + 63) / 64)), dim3(64), 0, w.st, g, ends.as<int64_t>(), rec.as<PathRec>(), emit_len.as<uint32_t>(),
                           idx.as<uint64_t>(), off.as<uint64_t>(), p0, pn, meta.as<ContigMeta>(), text.as<char>());
        MGTA_HIP_CHECK(hipGetLastError());
        const size_t m0 = out.meta.size(), t0 = out.text.size();
        out.meta.resize(m0 + n_contigs);
        out.text.resize(t0 + n_chars);
        MGTA_HIP_CHECK(hipMemcpyAsync(out.meta.data() + m0, meta.p, n_contigs * sizeof(ContigMeta), hipMemcpyDeviceToHost, w.st));
        MGTA_HIP_CHECK(hipMemcpyAsync(out.text.data() + t0, text.p, n_chars, hipMemcpyDeviceToHost, w.st));
        MGTA_HIP_CHECK(hipStreamSynchronize(w.st));
        for (size_t i = m0; i < out.meta.size(); ++i) out.meta[i].offset += char_base;
        char_base += n_chars;
    }
    note(w, "unitigs: %zu contigs, %zu characters written", out.meta.size(), out.text.size());
}

}  // namespace
}  // namespace mgta

using namespace mgta;

extern "C" {

int mgta_denovo(mgta_sdbg *graph, int max_tip_len, int no_bubble, int min_contig, char **fasta, uint64_t *fasta_len, mgta_denovo_stats *stats) {
    if (!graph || !fasta || !fasta_len) { set_error("mgta_denovo: bad argument"); return MGTA_EINVAL; }
    if (graph->dev.k > kMaxK) { set_error("mgta_denovo: k = %d > %d", graph->dev.k, kMaxK); return MGTA_EUNSUPPORTED; }
    *fasta = nullptr; *fasta_len = 0;
    mgta_denovo_stats s;
    memset(&s, 0, sizeof s);
    try {
        mgta_ctx *ctx = graph->ctx;
        MGTA_HIP_CHECK(hipSetDevice(ctx->device));
        Work w;
        w.ctx = ctx; w.st = ctx->stream;
        w.d.g = graph->dev;
        w.d.rw = graph->lines.as<GLine>();
        const GraphDev &g = w.d.g;
        if (g.size > 0) {
            w.mask.alloc((g.n_lines + 1) * 8, w.live(), w.peak());
            w.count.alloc((g.n_lines + 1) * 4, w.live(), w.peak());
            w.base.alloc((g.n_lines + 1) * 8, w.live(), w.peak());
            w.tmp.alloc(scan_tmp_elems(g.n_lines) * 8, w.live(), w.peak());
            w.total.alloc(64, w.live(), w.peak());
            hipEvent_t ev[4];
            for (auto &e : ev) MGTA_HIP_CHECK(hipEventCreate(&e));
            MGTA_HIP_CHECK(hipEventRecord(ev[0], w.st));
            if (max_tip_len == -1) max_tip_len = g.k * 2;                           // assembler.cpp:125-127
            note(w, "%lld edges", (long long)g.size);
            if (max_tip_len > 0) s.n_tips = (int64_t)remove_tips(w, max_tip_len);
            MGTA_HIP_CHECK(hipEventRecord(ev[1], w.st));
            if (!no_bubble) s.n_bubbles = (int64_t)pop_bubbles(w, s.n_bubble_rounds, s.n_bubble_candidates);
            MGTA_HIP_CHECK(hipEventRecord(ev[2], w.st));
            Contigs c;
            unitigs(w, min_contig, c);
            MGTA_HIP_CHECK(hipEventRecord(ev[3], w.st));
            MGTA_HIP_CHECK(hipEventSynchronize(ev[3]));
            MGTA_HIP_CHECK(hipEventElapsedTime(&s.ms_tips, ev[0], ev[1]));
            MGTA_HIP_CHECK(hipEventElapsedTime(&s.ms_bubbles, ev[1], ev[2]));
            MGTA_HIP_CHECK(hipEventElapsedTime(&s.ms_unitigs, ev[2], ev[3]));
            for (auto &e : ev) (void)hipEventDestroy(e);
            s.n_paths = c.n_paths;
            s.n_unitig_sweeps = c.n_sweeps;
            s.n_contigs = (int64_t)c.meta.size();
            // WriteContig, unitig_graph.cpp:134-150: ">k{K}_{id} flag={f} multi={%.4lf} len={L}\n{seq}\n", ids in emission order.  The
            // headers are formatted by several host threads (57.8 M of them took 16 s on one), each into its own piece of the text.
            const size_t nc = c.meta.size();
            const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)std::thread::hardware_concurrency(), (size_t)16, nc / 4096 + 1}));
            std::vector<std::string> piece(nt);
            std::vector<std::thread> th;
            auto fmt = [&](unsigned t) {
                const size_t i0 = nc * t / nt, i1 = nc * (t + 1) / nt;
                std::string &o = piece[t];
                size_t chars = 0;
                for (size_t i = i0; i < i1; ++i) chars += c.meta[i].len;
                o.reserve(chars + (i1 - i0) * 56);
                char head[160];
                for (size_t i = i0; i < i1; ++i) {
                    const ContigMeta &m = c.meta[i];
                    const double multi = std::min(65535.0, (double)m.depth / (double)m.length);
                    const int hl = snprintf(head, sizeof head, ">k%d_%lld flag=%d multi=%.4lf len=%d\n", g.k, (long long)(i + 1), m.flag, multi, (int)m.len);
                    o.append(head, (size_t)hl);
                    o.append(c.text.data() + m.offset, m.len);
                    o.push_back('\n');
                }
            };
            for (unsigned t = 1; t < nt; ++t) th.emplace_back(fmt, t);
            fmt(0);
            for (auto &x : th) x.join();
            for (const ContigMeta &m : c.meta) s.total_len += m.len;
            std::vector<char>().swap(c.text);
            size_t total = 0;
            for (const std::string &o : piece) total += o.size();
            char *buf = (char *)malloc(total + 1);
            if (!buf) { set_error("mgta_denovo: out of host memory"); return MGTA_ENOMEM; }
            size_t at = 0;
            for (std::string &o : piece) { memcpy(buf + at, o.data(), o.size()); at += o.size(); std::string().swap(o); }
            buf[total] = 0;
            *fasta = buf;
            *fasta_len = total;
            if (stats) *stats = s;
            return MGTA_OK;
        }
        char *buf = (char *)malloc(1);
        if (!buf) { set_error("mgta_denovo: out of host memory"); return MGTA_ENOMEM; }
        buf[0] = 0;
        *fasta = buf;
        *fasta_len = 0;
        if (stats) *stats = s;
        return MGTA_OK;
    } catch (const HipError &e) { return e.code; }
}

void mgta_host_free(void *p) { free(p); }

}  // extern "C"
