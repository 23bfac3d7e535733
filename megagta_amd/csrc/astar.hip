// astar.hip — batched HMM-guided A* over the device-resident succinct de Bruijn graph (gfx950).
//
// Replaces the OMP seed loop of search() (search.cpp:184-189) and, per seed and direction,
// HMMGraphSearch::astarSearch (hmm_graph_search.h:132-343) with NodeEnumerator::enumerateNodes
// (node_enumerator.h:65-246), AStarNode ordering (a_star_node.h:34-82) and the result walk
// getHighestScoreNode / partialResultFromGoal (hmm_graph_search.h:83-110,345-356).
//
// Mapping: one wavefront per (seed, direction) search, pulled from a work queue by persistent
// workgroups; all workgroups of one direction share that direction's profile-HMM tables, staged once
// per workgroup in LDS (fp64: msc, tsc, max_match, heuristic).
//   * frontier expansion is wave-parallel: lane l = (i,j,k) walks the 3-edge codon path
//     OutgoingEdges(curr)[i] -> [j] -> [k] through the 128-byte graph lines (<= 64 candidates,
//     reference order = ascending lane), ballot/popcount compaction of the surviving lanes;
//   * the order-sensitive bookkeeping (open list = binary heap with libstdc++'s exact
//     push_heap/pop_heap sift sequence, closed/open hash, node pool) is executed by lane 0 in the
//     reference's order, in per-search global-memory arenas (tagged entries: no clearing between searches);
//   * scores are IEEE fp64, compiled with -ffp-contract=off: path log-probabilities are bit-identical
//     to the x86-64 reference, fval = (int)(10000*(score+2h)) truncates identically.
// A search that outgrows its arena is re-run with a larger one (never on the CPU).
#include <algorithm>
#include <cmath>
#include <memory>
#include <string>

#include "common.hpp"
#include "device_utils.hpp"
#include "graph.hpp"

struct mgta_hmm {
    mgta_ctx *ctx = nullptr;
    int M = 0, A = 0;
    mgta::DevBuf tab;            // [msc (M+1)*A][tsc 7*(M+1)][maxm (M+1)][h 3*(M+1)]
    int8_t col[2][64];           // codon (c1*16+c2*4+c3) -> emission column; [0] codonTable, [1] rc_codonTable; -1 = stop
    mgta::DevBuf d_col;          // the same 128 bytes on the device
    size_t n_doubles = 0;
};

namespace mgta {

constexpr int kAstarWaves = 12;
constexpr int kAstarThreads = kAstarWaves * 64;
constexpr uint32_t kNone = 0x7FFFFFFFu;
constexpr int kMaxKmer = 160;

enum { T_MM = 0, T_MI = 1, T_MD = 2, T_IM = 3, T_II = 4, T_DM = 5, T_DD = 6 };   // profile_hmm.h:25
enum { ST_M = 0, ST_I = 1, ST_D = 2 };

struct ANode {                    // AStarNode, a_star_node.h:9-33
    double score, real_score, max_score;
    int64_t node_id;
    int32_t parent;               // index in the search's pool, -1 = none
    int32_t fval;
    int16_t state_no, length, negative_count;
    uint16_t em_state;            // nucl_emission (9 bits) | state << 9
};
static_assert(sizeof(ANode) == 48, "node layout");

struct HeapEnt {                  // 16 bytes; the priority (fval, -state_no, state rank) is rebuilt from key + fval
    uint64_t key;                 // node_id << 18 | state_no << 2 | (state + 1)
    int32_t fval;
    uint32_t node;                // index in the search's node pool
};
struct HashEnt {
    uint64_t key;                 // node_id << 18 | state_no << 2 | (state + 1)
    uint32_t val;                 // open-list node index (kNone = none) | closed << 31
    uint32_t tag;                 // entry is live iff tag == the search's tag
};

struct CacheEnt {
    unsigned long long key;       // parent key (0 = empty)
    unsigned long long val;       // ~(first seed that sees the entry << 16 | em_state of the child (nucl_emission | state << 9)); 0 = unset.
                                  // One atomicMax of the complement keeps the entry that becomes visible first (window B, no cost
                                  // term: the lowest owner = "first insert wins" of the sequential reference) and lets the table start
                                  // as all-zero bytes.
};

struct HmmView {
    const double *tab;
    int M, A;
    const int8_t *col_fwd;        // [64] codonTable -> column
    const int8_t *col_enum;       // [64] table used by the enumerator of this direction (codonTable / rc_codonTable)
};

struct AstarArgs {
    GraphDev g;
    HmmView hm[2];
    const char *kmers;            // n x klen, lower/upper ACGT
    const int32_t *start_state;
    const int64_t *start_node;    // [2n]: IndexBinarySearchEdge of the k-mer (dir 0) and of its reverse complement (dir 1)
    int64_t n_seeds;
    int klen;                     // k + 1
    int prune;
    double low_cov_penalty;       // -log(low_cov_pen)
    double log2v;
    const double *exit_prob;      // [3000]
    const int64_t *todo[2];       // seed indices still to run per direction
    int64_t n_todo[2];
    unsigned long long *queue;    // [2]
    ANode *nodes; HeapEnt *heap; HashEnt *hash;
    uint32_t cap_nodes, cap_hash; // cap_hash power of two
    uint32_t *slot_tag;
    mgta_astar_side *sides;       // [2n]
    char *out_seq; uint32_t out_cap; uint32_t *out_len;   // [2n]
    int32_t *status;              // [2n] 0 = pending, 1 = done, 2 = arena overflow, 3 = bad seed, 4 = gate timeout
    int use_lds;
    // shared term_nodes caches (search.cpp:182), one per direction.  window = 0: off (cold).  window = B >= 1: the path found by
    // seed j (c_j expansions) is seen by exactly the seeds >= j + B + c_j / cost_rate (cost_rate = 0: no cost term; B = 1 then is
    // the reference's sequential run).  The cost term lets later seeds start while a long search is still running: it cannot
    // become visible to them any more, however soon it ends.
    int window;
    int cost_rate;
    CacheEnt *cache[2];
    uint64_t cache_mask[2];
    unsigned long long *prof;     // [8] diagnostic cycle sums (MGTA_ASTAR_PROFILE builds only)
    long long *run_seed;          // [slots] seed a wave is working on (a lower bound while it is taking one from the queue), -1 = none
    unsigned long long *run_progress;   // [slots] expansions of that search so far (lags; only ever too small)
    unsigned long long *start_limit;    // [0..1] highest seed index known to be allowed to start (monotone cache of the gate), [2..3] time of the last refresh by a waiting wave
    uint32_t n_slots;
};

__device__ __forceinline__ int to_fval(double x) {   // (int)x as x86-64 cvttsd2si does it (INT_MIN when out of range / NaN)
    if (!(x > -2147483649.0 && x < 2147483648.0)) return (int)0x80000000;
    return (int)x;
}
__device__ __forceinline__ int srank(int st) { return st == ST_M ? 3 : st == ST_D ? 2 : 1; }
__device__ __forceinline__ uint64_t make_prio(int fval, int state_no, int st) {   // a < b  <=>  prio(a) < prio(b)
    return ((uint64_t)((uint32_t)fval ^ 0x80000000u) << 32) | ((uint64_t)(uint16_t)(0xFFFF - (uint16_t)state_no) << 2) | (uint64_t)srank(st);
}
__device__ __forceinline__ uint64_t make_key(int64_t node_id, int state_no, int st) {
    return ((uint64_t)node_id << 18) | ((uint64_t)(uint16_t)state_no << 2) | (uint64_t)(st + 1);
}
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// ---- open list: binary heap whose first kLdsHeap entries (levels 0..8) live in LDS, the rest in the search's arena.
// The sift sequences are libstdc++'s (bits/stl_heap.h __push_heap / __adjust_heap) so nodes of equal priority leave
// the list in the reference's order; they are executed by the whole wave: a pop prefetches five levels of the subtree
// under the hole with 62 lanes (one memory round trip per five levels), a push loads every ancestor at once.
constexpr uint32_t kLdsHeap = 255;

__device__ __forceinline__ uint64_t ent_prio(const HeapEnt &e) {   // AStarNode::operator< (a_star_node.h:34-82) as one integer
    int st = (int)(e.key & 3) - 1;
    return ((uint64_t)((uint32_t)e.fval ^ 0x80000000u) << 32) | ((uint64_t)(uint16_t)(0xFFFF - (uint16_t)((e.key >> 2) & 0xFFFF)) << 2) |
           (uint64_t)srank(st);
}
__device__ __forceinline__ HeapEnt hget(const HeapEnt *lds, const HeapEnt *glob, uint64_t i) { return i < kLdsHeap ? lds[i] : glob[i]; }
__device__ __forceinline__ void hset(HeapEnt *lds, HeapEnt *glob, uint64_t i, const HeapEnt &e) {
    if (i < kLdsHeap) lds[i] = e; else glob[i] = e;
}
__device__ __forceinline__ HeapEnt shfl_ent(const HeapEnt &e, int src) {
    HeapEnt r;
    r.key = __shfl(e.key, src, 64);
    r.fval = __shfl(e.fval, src, 64);
    r.node = __shfl(e.node, src, 64);
    return r;
}

// __push_heap(first, hole, 0, v): every lane calls it with the same arguments
__device__ __forceinline__ void heap_sift_up(HeapEnt *lds, HeapEnt *glob, uint64_t hole, const HeapEnt &v) {
    const int lane = lane_id();
    const int depth = 63 - __builtin_clzll(hole + 1);                 // number of ancestors of `hole`
    const uint64_t pv = ent_prio(v);
    HeapEnt e = v;
    bool less = false;
    if (lane < depth) {
        e = hget(lds, glob, ((hole + 1) >> (lane + 1)) - 1);
        less = ent_prio(e) < pv;
    }
    uint64_t bal = __ballot(less);
    int s = __builtin_ctzll(~bal);                                     // ancestors that move down one level
    if (s > depth) s = depth;
    if (lane < s) hset(lds, glob, ((hole + 1) >> lane) - 1, e);
    if (lane == 0) hset(lds, glob, ((hole + 1) >> s) - 1, v);
}

// pop_heap + pop_back; n = current size (> 0); returns the former top.
// __adjust_heap walks down from the root moving the larger child up (the right one unless right < left).  Five levels per memory
// round trip: lanes 0..61 hold the subtree under the hole in level order (siblings = lanes 2i, 2i+1), every lane decides locally
// whether its parent would pick it, a chain of five ballots tells which lanes lie on the path, and those lanes move their
// entries up at once.
__device__ __forceinline__ HeapEnt heap_pop(HeapEnt *lds, HeapEnt *glob, uint32_t n) {
    const int lane = lane_id();
    HeapEnt top = hget(lds, glob, 0);
    if (n > 1) {
        const HeapEnt v = hget(lds, glob, n - 1);
        const int64_t len = (int64_t)n - 1;
        const int64_t half = (len - 1) / 2;                            // nodes below `half` have two children
        int64_t hole = 0;
        const int d = 31 - __builtin_clz((unsigned)lane + 2);          // depth 1..5 below the hole for lanes 0..61
        const int o = lane + 2 - (1 << d);                             // offset inside the level
        const int parent_lane = d > 1 ? (1 << (d - 1)) - 2 + (o >> 1) : 0;
        while (hole < half) {
            const int64_t idx = ((hole + 1) << d) - 1 + o, pidx = (idx - 1) >> 1;
            HeapEnt e;
            e.key = 0; e.fval = 0; e.node = 0;
            if (lane < 62 && idx < len) e = hget(lds, glob, (uint64_t)idx);
            const uint64_t pr = ent_prio(e), sib = __shfl_xor(pr, 1, 64);
            // std::__adjust_heap: second = right child; if (right < left) second = left
            const bool right_less = (o & 1) ? (pr < sib) : (sib < pr);
            const bool chosen = lane < 62 && pidx < half && ((o & 1) ? !right_less : right_less);
            uint64_t path = 0;
#pragma unroll
            for (int t = 1; t <= 5; ++t) {
                const bool on = d == t && chosen && (t == 1 || ((path >> parent_lane) & 1ull));
                path |= __ballot(on);
            }
            if ((path >> lane) & 1ull) hset(lds, glob, (uint64_t)pidx, e);     // every node of the path moves up one level
            const int deepest = 63 - __builtin_clzll(path);                     // path != 0: the hole has two children
            hole = __shfl(idx, deepest, 64);
        }
        if ((len & 1) == 0 && hole == (len - 2) / 2) {                 // a last, single (left) child
            HeapEnt ce = hget(lds, glob, (uint64_t)(2 * hole + 1));
            if (lane == 0) hset(lds, glob, (uint64_t)hole, ce);
            hole = 2 * hole + 1;
        }
        heap_sift_up(lds, glob, (uint64_t)hole, v);
    }
    return top;
}

// ---- closed set + open_hash: one open-addressing table per search; every lane probes the same key (one request)
__device__ __forceinline__ uint32_t hash_find(const HashEnt *tab, uint32_t hmask, uint32_t tag, uint64_t key, bool &found, uint32_t &val) {
    uint32_t i = (uint32_t)mix64(key) & hmask;
    while (true) {
        HashEnt e = tab[i];
        if (e.tag != tag) { found = false; val = kNone; return i; }
        if (e.key == key) { found = true; val = e.val; return i; }
        i = (i + 1) & hmask;
    }
}

// agent-scope accesses: the caches / frontier are written by other CUs (and XCDs) of the same launch
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) {
    return __hip_atomic_fetch_add(const_cast<unsigned long long *>(p), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Protocol words (queue, table of running searches, start limit) are only ever touched by agent-scope atomics performed at the
// coherence point; they are ordered by waiting for the returning atomic before the next one is issued.  No acquire / release
// fences: on this part an agent-scope acquire invalidates, and a release writes back, the whole L2 of the XCD, and a gate that does
// that at polling rate slows every running search by an order of magnitude (measured: 5.3 s -> 257 s).
__device__ __forceinline__ void st_agent(long long *p, long long v) {
    (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v) {
    (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ long long ld_agent_ll(const long long *p) {
    long long v = __hip_atomic_fetch_add(const_cast<long long *>(p), 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return v;
}
// child descriptor cached for `key` and visible to seed `seed`, or -1
__device__ __forceinline__ int cache_lookup(const AstarArgs &a, int dir, uint64_t key, int64_t seed) {
    const CacheEnt *tab = dir ? a.cache[1] : a.cache[0];
    const uint64_t cmask = dir ? a.cache_mask[1] : a.cache_mask[0];
    uint64_t i = mix64(key) & cmask;
    while (true) {
        unsigned long long k = ld_agent(&tab[i].key);
        if (k == 0) return -1;
        if (k == key) {
            unsigned long long v = ld_agent(&tab[i].val);
            if (v == 0ull) return -1;
            v = ~v;
            return (int64_t)(v >> 16) <= seed ? (int)(v & 0xFFFF) : -1;
        }
        i = (i + 1) & cmask;
    }
}
__device__ __forceinline__ void cache_insert(const AstarArgs &a, int dir, uint64_t key, int64_t visible_from, int em_state) {
    CacheEnt *tab = dir ? a.cache[1] : a.cache[0];
    const uint64_t cmask = dir ? a.cache_mask[1] : a.cache_mask[0];
    uint64_t i = mix64(key) & cmask;
    unsigned long long v = ((unsigned long long)visible_from << 16) | (unsigned long long)(em_state & 0xFFFF);
    while (true) {
        unsigned long long k = ld_agent(&tab[i].key);
        if (k == 0) {
            unsigned long long expect = 0;
            if (__hip_atomic_compare_exchange_strong(&tab[i].key, &expect, (unsigned long long)key, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT))
                k = key;
            else
                k = expect;
        }
        if (k == key) {
            __hip_atomic_fetch_max(&tab[i].val, ~v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        i = (i + 1) & cmask;
    }
}

// Highest seed that may start now: no unfinished search can still become visible to it (see AstarArgs::window).  Every lane returns
// the same value; start_limit keeps the maximum ever computed (the bound only grows).  The table reads are atomics performed at the
// coherence point, four in flight per lane.
__device__ __forceinline__ long long start_bound(const AstarArgs &a, int dir, int lane) {
    const long long head = (long long)ld_agent(&a.queue[dir]);       // the queue first: every seed below it is in the table by now
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long bound = head + a.window - 1;
    const uint32_t n_dir = a.n_slots / 2;                            // this direction's waves: workgroups 2b + dir
    for (uint32_t t0 = 0; t0 < n_dir; t0 += 256) {
        long long js[4];
        unsigned long long pr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t t = t0 + u * 64 + (uint32_t)lane;
            const uint32_t sl = (2 * (t / kAstarWaves) + (uint32_t)dir) * kAstarWaves + t % kAstarWaves;
            js[u] = -1; pr[u] = 0;
            if (t < n_dir) {
                js[u] = __hip_atomic_fetch_add(&a.run_seed[sl], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.cost_rate > 0) pr[u] = __hip_atomic_fetch_add(&a.run_progress[sl], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (js[u] >= 0) {
                const long long b = js[u] + a.window - 1 + (a.cost_rate > 0 ? (long long)(pr[u] / (unsigned)a.cost_rate) : 0ll);
                bound = b < bound ? b : bound;
            }
    }
    bound = wave_min_ll(bound);
    if (lane == 0 && bound > 0) __hip_atomic_fetch_max(&a.start_limit[dir], (unsigned long long)bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return bound;
}

#ifdef MGTA_ASTAR_PROFILE   // diagnostic build only: per-phase cycle sums (s_memtime), lane 0 of every wave
#define PROF_DECL unsigned long long pt_ = __builtin_amdgcn_s_memtime(), pacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PROF(i) { unsigned long long n_ = __builtin_amdgcn_s_memtime(); pacc_[i] += n_ - pt_; pt_ = n_; }
#define PROF_FLUSH if (lane == 0) for (int q_ = 0; q_ < 8; ++q_) atomicAdd(&a.prof[q_], pacc_[q_]);
#else
#define PROF_DECL
#define PROF(i)
#define PROF_FLUSH
#endif

template <bool LDS>
__global__ __launch_bounds__(kAstarThreads) void astar_kernel(AstarArgs a) {
    extern __shared__ __align__(16) double s_tab[];
    __shared__ HeapEnt s_heap[kAstarWaves][kLdsHeap + 1];
    const int dir = blockIdx.x & 1;
    HmmView hv;                                                     // select by value: no indexed access into the kernel arguments
    hv.tab = dir ? a.hm[1].tab : a.hm[0].tab; hv.M = dir ? a.hm[1].M : a.hm[0].M; hv.A = dir ? a.hm[1].A : a.hm[0].A;
    hv.col_fwd = dir ? a.hm[1].col_fwd : a.hm[0].col_fwd; hv.col_enum = dir ? a.hm[1].col_enum : a.hm[0].col_enum;
    const int M = hv.M, A = hv.A;
    const double *tab = hv.tab;
    if (LDS) {
        size_t nd = (size_t)(M + 1) * (A + 11);
        for (size_t i = threadIdx.x; i < nd; i += kAstarThreads) s_tab[i] = hv.tab[i];
        __syncthreads();
        tab = s_tab;
    }
    const size_t M1 = (size_t)M + 1;
    const double *msc = tab, *tsc = tab + M1 * A, *maxm = tsc + 7 * M1, *hc = maxm + M1;
    const int lane = lane_id(), wv = wave_id();
    const uint32_t slot = blockIdx.x * kAstarWaves + wv;
    const GraphDev &g = a.g;
    const bool forward = dir == 0;
    const double NEG_INF = -__builtin_inf();

    // per-search arena (every lane holds the same scalars; stores are done by lane 0)
    ANode *const nodes = a.nodes + (size_t)slot * a.cap_nodes;
    HeapEnt *const gheap = a.heap + (size_t)slot * a.cap_nodes;
    HashEnt *const hash = a.hash + (size_t)slot * a.cap_hash;
    HeapEnt *const lheap = s_heap[wv];
    const uint32_t hmask = a.cap_hash - 1;
    uint32_t tag = a.slot_tag[slot];
    PROF_DECL

    while (true) {
        // ---- next search of this direction
        long long qi = 0;
        if (lane == 0) {
            if (a.window > 0) {
                // announce a lower bound of the seed about to be taken BEFORE taking it: whoever sees the queue beyond a seed also sees
                // a wave that holds it (or its committed paths)
                st_agent(&a.run_progress[slot], 0ull);
                st_agent(&a.run_seed[slot], (long long)ld_agent(&a.queue[dir]));
                qi = (long long)__hip_atomic_fetch_add(&a.queue[dir], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                qi = (long long)atomicAdd(&a.queue[dir], 1ull);
            }
        }
        qi = __shfl(qi, 0, 64);
        if (qi >= (dir ? a.n_todo[1] : a.n_todo[0])) break;
        const int64_t seed = (dir ? a.todo[1] : a.todo[0])[qi];
        const int64_t sid = seed * 2 + dir;
        if (a.window > 0) {
            if (lane == 0) st_agent(&a.run_seed[slot], (long long)seed);
            // Seed i may start once no unfinished search j can still become visible to it: i < j + B + progress_j / cost_rate for every
            // running j, and i < q + B for the next seed q of the queue.  The lowest running search always passes, so this terminates;
            // the wait is bounded anyway.
            // The limit moves when a search ends (its wave recomputes it) and, with a cost term, as the running searches progress: for that
            // ONE waiting wave per direction and ~50 us re-reads the table (ticket = time of the last refresh); the others poll one word.
            int gate_ok = 1;
            unsigned long long spins = 0;
            if (start_bound(a, dir, lane) < (long long)seed) {
                while (true) {
                    long long lim = 0;
                    int refresh = 0;
                    if (lane == 0) {
                        lim = (long long)ld_agent(&a.start_limit[dir]);
                        if (lim < (long long)seed) {
                            const unsigned long long now = __builtin_amdgcn_s_memrealtime();     // 100 MHz
                            unsigned long long last = ld_agent(&a.start_limit[2 + dir]);
                            if (now - last > 5000ull)
                                refresh = __hip_atomic_compare_exchange_strong(&a.start_limit[2 + dir], &last, now, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                               __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    lim = __shfl(lim, 0, 64);
                    if (lim >= (long long)seed) break;
                    refresh = __shfl(refresh, 0, 64);
                    if (refresh && start_bound(a, dir, lane) >= (long long)seed) break;
#pragma unroll
                    for (int z = 0; z < 8; ++z) __builtin_amdgcn_s_sleep(127);
                    if (++spins > (1ull << 21)) { gate_ok = 0; break; }
                }
            }
            if (!gate_ok) {
                if (lane == 0) { a.status[sid] = 4; st_agent(&a.run_seed[slot], -1ll); }
                break;
            }
        }
        ++tag;
        uint32_t n_nodes = 0, n_heap = 0, n_keys = 0;
        int64_t n_closed = 0, n_expanded = 0, n_opened = 0;
        int status = 1, partial = 0, ok = 0;
        int32_t goal = -1, inter = 0, cur = 0;
        bool first = true, done = false;
        ANode curr;                                                     // node being expanded

        // ---- start node (hmm_graph_search.h:132-189)
        {
            const char *km = a.kmers + seed * a.klen;
            int n_aa = a.klen / 3;
            int sstate = forward ? a.start_state[seed] : (M - a.start_state[seed] - n_aa);   // :73
            bool bad = sstate < 0 || sstate + n_aa > M || a.klen > kMaxKmer;
            double sc = 0, rs = 0;
            if (!bad) {
                for (int i = 1; i <= n_aa; ++i) {                      // scoreStart / realScoreStart, :112-130
                    int ci = forward ? (i - 1) : (n_aa - i);           // the reverse search scores the reversed protein
                    int c = 0;
                    for (int t = 0; t < 3; ++t) {
                        char ch = km[3 * ci + t];
                        int b = (ch == 'A' || ch == 'a') ? 0 : (ch == 'C' || ch == 'c') ? 1 : (ch == 'G' || ch == 'g' || ch == 'N' || ch == 'n') ? 2
                                : (ch == 'T' || ch == 't') ? 3 : -1;
                        if (b < 0) bad = true;
                        c = c * 4 + (b < 0 ? 0 : b);
                    }
                    int col = hv.col_fwd[c];
                    if (col < 0) { bad = true; break; }
                    double m = msc[(size_t)(sstate + i) * A + col], t = tsc[(size_t)T_MM * M1 + sstate + i - 1];
                    sc += m + t - maxm[sstate + i];
                    rs += m + t;
                }
            }
            if (bad) { status = 3; done = true; }
            else {
                curr.parent = -1; curr.state_no = (int16_t)(sstate + n_aa); curr.em_state = (uint16_t)(ST_M << 9); curr.length = (int16_t)n_aa;
                curr.fval = 0; curr.score = sc; curr.real_score = rs; curr.max_score = 0; curr.negative_count = 0;
                curr.node_id = a.start_node[sid];
                if (lane == 0) nodes[0] = curr;
                n_nodes = 1;
                if (curr.state_no >= M) { ok = 1; goal = 0; done = true; }            // :193-197
                else if (curr.node_id == -1) { ok = 0; done = true; n_opened = 1; }    // no children -> open.empty() -> false (:235-237)
            }
        }

        // ---- main loop: one expansion per iteration
        PROF(0)
        while (!done) {
            if (!first) {
                // pop until a node that is not closed (hmm_graph_search.h:243-257)
                bool have = false;
                uint32_t hs = 0, hval = kNone;
                uint64_t hkey = 0;
                while (n_heap > 0) {
                    HeapEnt top = heap_pop(lheap, gheap, n_heap);
                    --n_heap;
                    bool found;
                    hs = hash_find(hash, hmask, tag, top.key, found, hval);
                    if (found && (hval >> 31)) continue;                               // closed
                    cur = (int32_t)top.node;
                    hkey = top.key;
                    if (!found) ++n_keys;                                              // (children of the first expansion are not in open_hash)
                    have = true;
                    break;
                }
                PROF(1)
                if (!have) { partial = 1; ok = 1; goal = inter; break; }              // open list ran dry (:339-341)
                curr = nodes[cur];
                const ANode ig = nodes[inter];
                bool better = (curr.real_score + a.exit_prob[curr.length]) / a.log2v > (ig.real_score + a.exit_prob[ig.length]) / a.log2v;
                if (curr.state_no >= M) {                                              // goal (:259-270)
                    if (better) inter = cur;
                    ok = 1; goal = inter;
                    break;
                }
                if (lane == 0) { HashEnt e; e.key = hkey; e.val = hval | 0x80000000u; e.tag = tag; hash[hs] = e; }   // closed.insert (:272)
                n_closed++;
                if (better) inter = cur;                                               // :274-277
                if (n_keys * 2 > hmask) { status = 2; break; }
            }
            PROF(2)
            const int cst = curr.em_state >> 9;
            const int next_state = curr.state_no + 1;
            // term_nodes.find(curr) (hmm_graph_search.h:212,279): child recorded by an earlier seed, or -1
            int cached = -1;
            if (a.window > 0) {
                if (lane == 0) cached = cache_lookup(a, dir, make_key(curr.node_id, curr.state_no, cst), seed);
                cached = __shfl(cached, 0, 64);
            }
            const int cached_st = cached >= 0 ? (cached >> 9) : -1;

            PROF(3)
            // ---- wave-parallel enumeration of the <= 64 codon paths (node_enumerator.h:98-128)
            int64_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
            LineR Ls = g_load_line(g, (uint64_t)curr.node_id >> 6), Lt;
            uint64_t lt_idx = 0;
            int od1 = g_outgoing_line(g, Ls, curr.node_id, p0, p1, p2, p3, Lt, lt_idx);
            const int i = lane >> 4, j = (lane >> 2) & 3, kk = lane & 3;
            bool valid = i < od1;
            int64_t packed = 0;
            if (valid) {
                int64_t e1 = (int64_t)sel4((uint64_t)p0, (uint64_t)p1, (uint64_t)p2, (uint64_t)p3, i);
                if ((uint64_t)(e1 >> 10) != lt_idx) Lt = g_load_line(g, (uint64_t)(e1 >> 10));   // (e1 >> 4) >> 6: rare, a node's edges straddle two lines
                Ls = Lt;
                int od2 = g_outgoing_line(g, Ls, e1 >> 4, p0, p1, p2, p3, Lt, lt_idx);
                valid = j < od2;
                if (valid) {
                    int64_t e2 = (int64_t)sel4((uint64_t)p0, (uint64_t)p1, (uint64_t)p2, (uint64_t)p3, j);
                    if ((uint64_t)(e2 >> 10) != lt_idx) Lt = g_load_line(g, (uint64_t)(e2 >> 10));
                    Ls = Lt;
                    int od3 = g_outgoing_line(g, Ls, e2 >> 4, p0, p1, p2, p3, Lt, lt_idx);
                    valid = kk < od3;
                    if (valid) {
                        int64_t e3 = (int64_t)sel4((uint64_t)p0, (uint64_t)p1, (uint64_t)p2, (uint64_t)p3, kk);
                        int c1 = (int)(e1 & 7) - 1, c2 = (int)(e2 & 7) - 1, c3 = (int)(e3 & 7) - 1;
                        int low = (int)((e1 >> 3) & 1) & (int)((e2 >> 3) & 1) & (int)((e3 >> 3) & 1);
                        packed = ((e3 >> 4) << 16) | ((int64_t)low << 9) | (c1 << 6) | (c2 << 3) | c3;
                    }
                }
            }
            n_expanded++;
            if (a.window > 0 && a.cost_rate > 0 && (n_expanded & 63) == 0 && lane == 0)
                __hip_atomic_store(&a.run_progress[slot], (unsigned long long)n_expanded, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            PROF(4)

            // ---- children (node_enumerator.h:131-244): every lane scores ITS codon's match / insert child
            double mt, it, dt;
            if (cst == ST_M) { mt = tsc[T_MM * M1 + curr.state_no]; it = tsc[T_MI * M1 + curr.state_no]; dt = tsc[T_MD * M1 + curr.state_no]; }
            else if (cst == ST_D) { mt = tsc[T_DM * M1 + curr.state_no]; it = NEG_INF; dt = tsc[T_DD * M1 + curr.state_no]; }
            else { mt = tsc[T_IM * M1 + curr.state_no]; it = tsc[T_II * M1 + curr.state_no]; dt = NEG_INF; }
            const double max_match = maxm[next_state];
            const double h_m = hc[next_state], h_i = hc[M1 + curr.state_no], h_d = hc[2 * M1 + next_state];

            int col = -1;
            if (valid) col = hv.col_enum[(int)((packed >> 6) & 7) * 16 + (int)((packed >> 3) & 7) * 4 + (int)(packed & 7)];
            bool use = valid && col >= 0;                                              // stop codon (:142-144)
            if (use && cached >= 0)                                                    // child_node->node_id != packed >> 16 (:146-148)
                use = cached_st == ST_D ? ((packed >> 16) == curr.node_id) : ((int)(packed & 511) == (cached & 511));
            const bool any_pass = __ballot(use) != 0;
            // a cached match/insert child ends the enumeration at that child (:178-181,207-210)
            const bool want_ins = use && cst != ST_D && cached_st != ST_M;
            const bool want_del = cst != ST_I && !((cached_st == ST_M || cached_st == ST_I) && any_pass);

            ANode cm, ci;                                                              // this lane's match / insert child
            cm.parent = cur; ci.parent = cur;
            cm.node_id = packed >> 16; ci.node_id = packed >> 16;
            cm.length = (int16_t)(curr.length + 1); ci.length = cm.length;
            cm.state_no = (int16_t)next_state; ci.state_no = curr.state_no;
            cm.em_state = (uint16_t)((packed & 511) | (ST_M << 9)); ci.em_state = (uint16_t)((packed & 511) | (ST_I << 9));
            const double pen = (packed & (1 << 9)) ? a.low_cov_penalty : 0.0;          // :150
            {
                double e = mt + (use ? msc[(size_t)next_state * A + col] : 0.0);
                cm.real_score = curr.real_score + e - pen;
                if (cm.real_score >= curr.max_score) { cm.max_score = cm.real_score; cm.negative_count = 0; }
                else { cm.max_score = curr.max_score; cm.negative_count = (int16_t)(curr.negative_count + 1); }
                cm.score = curr.score + (e - pen - max_match);
                cm.fval = to_fval(10000 * (cm.score + 2.0 * h_m));                     // :173
                double ei = it + (next_state == M ? NEG_INF : 0.0);                    // isc == 0 except at node M
                ci.real_score = curr.real_score + ei - pen;
                ci.max_score = curr.max_score;
                ci.negative_count = (int16_t)(curr.negative_count + 1);
                ci.score = curr.score + (ei - pen);
                ci.fval = to_fval(10000 * (ci.score + 2.0 * h_i));
            }
            ANode cd;                                                                  // delete child (:218-244), same in every lane
            cd.parent = cur; cd.node_id = curr.node_id;
            cd.state_no = (int16_t)next_state; cd.length = curr.length;
            cd.real_score = curr.real_score + dt;
            cd.max_score = curr.max_score;
            cd.negative_count = (int16_t)(curr.negative_count + 1);
            cd.score = curr.score + (dt - max_match);
            cd.fval = to_fval(10000 * (cd.score + 2.0 * h_d));
            cd.em_state = (uint16_t)(((4 << 6) | (4 << 3) | 4) | (ST_D << 9));

            // ---- admission (hmm_graph_search.h:288-311): prune test + open_hash lookup, all children in parallel
            auto admissible = [&](const ANode &nx) {
                return a.prune > 0 ? ((nx.length < 5 || nx.negative_count <= a.prune) && nx.real_score > 0.0) : true;
            };
            auto probe_open = [&](const ANode &nx, bool want) -> bool {               // per-lane probe of this lane's own key
                if (!want) return false;
                if (first) return true;                                                // :212-233: no pruning / dedup
                if (!admissible(nx)) return false;
                uint64_t key = make_key(nx.node_id, nx.state_no, nx.em_state >> 9);
                uint32_t ii = (uint32_t)mix64(key) & hmask;
                while (true) {
                    HashEnt e = hash[ii];
                    if (e.tag != tag) return true;
                    if (e.key == key) {
                        uint32_t oi = e.val & kNone;
                        if (oi == kNone) return true;
                        return nodes[oi].fval < nx.fval;                               // got->second < next (:299-302); equal keys => only fval differs
                    }
                    ii = (ii + 1) & hmask;
                }
            };
            const bool open_m = probe_open(cm, use);
            const bool open_i = probe_open(ci, want_ins);
            const bool open_d = probe_open(cd, want_del && lane == 0);
            const uint64_t mm = __ballot(open_m), mi = __ballot(open_i);
            PROF(5)
            const bool del = __shfl((int)open_d, 0, 64) != 0;
            const uint32_t n_new = (uint32_t)__popcll(mm) + (uint32_t)__popcll(mi) + (del ? 1u : 0u);
            if (n_nodes + n_new > a.cap_nodes) { status = 2; break; }
            // node indices in reference order: codon by codon (ascending lane), match before insert, delete last
            const uint64_t lt = lanemask_lt();
            const uint32_t idx_m = n_nodes + (uint32_t)__popcll(mm & lt) + (uint32_t)__popcll(mi & lt);
            const uint32_t idx_i = idx_m + (open_m ? 1u : 0u);
            if (open_m) nodes[idx_m] = cm;
            if (open_i) nodes[idx_i] = ci;
            const uint32_t idx_d = n_nodes + n_new - 1;
            if (del && lane == 0) nodes[idx_d] = cd;
            // commit in order: open_hash[next] = next (:331) and open.push (:335)
            uint64_t todo_m = mm, todo_i = mi;
            while (todo_m | todo_i) {
                int lm = todo_m ? __builtin_ctzll(todo_m) : 64, li = todo_i ? __builtin_ctzll(todo_i) : 64;
                bool is_m = lm <= li;                                                  // same lane: match first
                int src = is_m ? lm : li;
                if (is_m) todo_m &= todo_m - 1; else todo_i &= todo_i - 1;
                HeapEnt he;
                he.key = __shfl(make_key(is_m ? cm.node_id : ci.node_id, is_m ? cm.state_no : ci.state_no, is_m ? ST_M : ST_I), src, 64);
                he.fval = __shfl(is_m ? cm.fval : ci.fval, src, 64);
                he.node = __shfl(is_m ? idx_m : idx_i, src, 64);
                if (!first) {
                    bool found; uint32_t val;
                    uint32_t hs = hash_find(hash, hmask, tag, he.key, found, val);
                    if (lane == 0) { HashEnt e; e.key = he.key; e.val = (found ? (val & 0x80000000u) : 0u) | he.node; e.tag = tag; hash[hs] = e; }
                    if (!found) ++n_keys;
                    n_opened++;
                }
                heap_sift_up(lheap, gheap, n_heap, he);
                ++n_heap;
            }
            if (del) {
                HeapEnt he;
                he.key = make_key(cd.node_id, cd.state_no, ST_D); he.fval = cd.fval; he.node = idx_d;
                if (!first) {
                    bool found; uint32_t val;
                    uint32_t hs = hash_find(hash, hmask, tag, he.key, found, val);
                    if (lane == 0) { HashEnt e; e.key = he.key; e.val = (found ? (val & 0x80000000u) : 0u) | he.node; e.tag = tag; hash[hs] = e; }
                    if (!found) ++n_keys;
                    n_opened++;
                }
                heap_sift_up(lheap, gheap, n_heap, he);
                ++n_heap;
            }
            n_nodes += n_new;
            PROF(6)
            if (n_keys * 2 > hmask) { status = 2; break; }
            if (first) {
                first = false;
                n_opened = 1;
                if (n_heap == 0) { ok = 0; break; }                                    // :235-237
            }
        }

        // ---- result: getHighestScoreNode + partialResultFromGoal (hmm_graph_search.h:83-110,345-356)
        {
            mgta_astar_side r;
            r.ok = ok; r.partial = partial; r.n_closed = n_closed; r.n_expanded = n_expanded; r.n_opened = n_opened;
            r.fval = 0; r.length = 0; r.state_no = -1; r.state = '-'; r.node_id = -1; r.real_score = 0; r.score = 0;
            uint32_t len = 0;
            char *dst = a.out_seq + (size_t)sid * a.out_cap;
            int32_t best = -1;
            if (status == 1 && ok && goal >= 0) {
                best = goal;
                double best_rs = nodes[goal].real_score;
                for (int32_t p = nodes[goal].parent; p >= 0;) {
                    ANode nd = nodes[p];
                    if (nd.real_score > best_rs) { best = p; best_rs = nd.real_score; }
                    p = nd.parent;
                }
                const ANode gn = nodes[best];
                r.fval = gn.fval; r.length = gn.length; r.state_no = gn.state_no;
                r.state = "mid"[gn.em_state >> 9]; r.node_id = gn.node_id; r.real_score = gn.real_score; r.score = gn.score;
                // 3 characters per non-delete node from the goal back to the start, then reversed (:92-108);
                // term_nodes.insert(parent -> child) along the same walk (:97-103)
                ANode nd = gn;
                while (nd.parent >= 0) {
                    if ((nd.em_state >> 9) != ST_D) {
                        if (len + 3 > a.out_cap) { status = 2; break; }
                        if (lane == 0)
                            for (int t = 0; t < 3; ++t) dst[len + t] = "acgt-"[(nd.em_state >> (3 * t)) & 7];
                        len += 3;
                    }
                    const ANode par = nodes[nd.parent];
                    if (a.window > 0 && lane == 0)
                        cache_insert(a, dir, make_key(par.node_id, par.state_no, par.em_state >> 9),
                                     seed + a.window + (a.cost_rate > 0 ? n_expanded / a.cost_rate : 0), nd.em_state);
                    nd = par;
                }
                if (lane == 0)
                    for (uint32_t x = 0; x < len / 2; ++x) { char t = dst[x]; dst[x] = dst[len - 1 - x]; dst[len - 1 - x] = t; }
            }
            if (lane == 0) {
                a.sides[sid] = r;
                a.out_len[sid] = len;
                a.status[sid] = status;
                if (a.window > 0) {      // the paths are in the cache (atomics, all performed): this search no longer holds anybody back
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    st_agent(&a.run_seed[slot], -1ll);
                }
            }
            if (a.window > 0) (void)start_bound(a, dir, lane);       // whoever finishes a search moves the limit for the waiting ones
        }
        PROF(7)
        __builtin_amdgcn_wave_barrier();
    }
    PROF_FLUSH
    if (lane == 0) {
        a.slot_tag[slot] = tag;
        if (a.window > 0) st_agent(&a.run_seed[slot], -1ll);
    }
}

}  // namespace mgta

using namespace mgta;

static const char kCodonAA[65] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";   // codon.h:9-106

extern "C" {

int mgta_hmm_load(mgta_ctx *ctx, int M, int A, const double *msc, const double *tsc, const double *max_match, const double *h,
                  const int32_t *alpha, mgta_hmm **out) {
    if (!ctx || !msc || !tsc || !max_match || !h || !alpha || !out || M < 1 || M > 30000 || A < 1 || A > 64) {
        set_error("mgta_hmm_load: bad argument");
        return MGTA_EINVAL;
    }
    try {
        MGTA_HIP_CHECK(hipSetDevice(ctx->device));
        auto hm = std::make_unique<mgta_hmm>();
        hm->ctx = ctx; hm->M = M; hm->A = A;
        size_t M1 = (size_t)M + 1;
        hm->n_doubles = M1 * (A + 11);
        std::vector<double> host(hm->n_doubles);
        std::copy(msc, msc + M1 * A, host.begin());
        for (size_t j = 0; j < (size_t)A; ++j) host[j] = -INFINITY;                // msc(0, .) = -inf (profile_hmm.h:58-64)
        std::copy(tsc, tsc + 7 * M1, host.begin() + M1 * A);
        std::copy(max_match, max_match + M1, host.begin() + M1 * (A + 7));
        std::copy(h, h + 3 * M1, host.begin() + M1 * (A + 8));
        hm->tab.alloc(hm->n_doubles * 8, &ctx->live_bytes, &ctx->peak_bytes);
        MGTA_HIP_CHECK(hipMemcpy(hm->tab.p, host.data(), hm->n_doubles * 8, hipMemcpyHostToDevice));
        for (int c = 0; c < 64; ++c) {
            int c1 = c >> 4, c2 = (c >> 2) & 3, c3 = c & 3;
            char f = kCodonAA[c], r = kCodonAA[(3 - c3) * 16 + (3 - c2) * 4 + (3 - c1)];   // rc_codonTable, codon.h:108-209
            hm->col[0][c] = (int8_t)(f == '*' ? -1 : alpha[(int)f]);
            hm->col[1][c] = (int8_t)(r == '*' ? -1 : alpha[(int)r]);
            if ((f != '*' && alpha[(int)f] < 0) || (r != '*' && alpha[(int)r] < 0)) {
                set_error("mgta_hmm_load: amino acid without a column in the model alphabet");
                return MGTA_EINVAL;
            }
        }
        hm->d_col.alloc(128, &ctx->live_bytes, &ctx->peak_bytes);
        MGTA_HIP_CHECK(hipMemcpy(hm->d_col.p, hm->col, 128, hipMemcpyHostToDevice));
        ctx_retain(ctx);
        *out = hm.release();
        return MGTA_OK;
    } catch (const HipError &e) { return e.code; }
}

void mgta_hmm_free(mgta_hmm *h) {
    if (!h) return;
    mgta_ctx *c = h->ctx;
    delete h;
    ctx_release(c);
}

int mgta_astar_batch(mgta_sdbg *g, const mgta_hmm *fwd, const mgta_hmm *rev, const char *kmers, const int32_t *start_state, int64_t n,
                     int prune_len, double low_cov_penalty, int cache_mode, mgta_contig_sink sink, void *user, mgta_astar_stats *stats) {
    if (!g || !fwd || !rev || n < 0 || (n > 0 && (!kmers || !start_state))) { set_error("mgta_astar_batch: bad argument"); return MGTA_EINVAL; }
    if (cache_mode < 0) { set_error("cache_mode must be >= 0"); return MGTA_EINVAL; }
    mgta_ctx *ctx = g->ctx;
    const int klen = g->dev.k + 1;
    if (klen > kMaxKmer) { set_error("k too large"); return MGTA_EINVAL; }
    try {
        MGTA_HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        mgta_astar_stats ST;
        memset(&ST, 0, sizeof(ST));
        ST.n_seeds = n;
        hipEvent_t ev0, ev1, evk0, evk1;
        MGTA_HIP_CHECK(hipEventCreate(&ev0)); MGTA_HIP_CHECK(hipEventCreate(&ev1));
        MGTA_HIP_CHECK(hipEventCreate(&evk0)); MGTA_HIP_CHECK(hipEventCreate(&evk1));
        MGTA_HIP_CHECK(hipEventRecord(ev0, st));
        if (n == 0) { if (stats) *stats = ST; return MGTA_OK; }

        // start edges: the k-mer (right search) and its reverse complement (left search), hmm_graph_search.h:163-186
        std::vector<uint8_t> seqs((size_t)n * 2 * klen);
        for (int64_t s = 0; s < n; ++s) {
            const char *km = kmers + s * klen;
            for (int i = 0; i < klen; ++i) {
                char c = km[i];
                int b = (c == 'A' || c == 'a') ? 1 : (c == 'C' || c == 'c') ? 2 : (c == 'G' || c == 'g' || c == 'N' || c == 'n') ? 3
                        : (c == 'T' || c == 't') ? 4 : 0;                           // dna_map, hmm_graph_search.h:54-58
                seqs[(size_t)(2 * s) * klen + i] = (uint8_t)b;
                seqs[(size_t)(2 * s + 1) * klen + (klen - 1 - i)] = (uint8_t)(b ? 5 - b : 0);
            }
        }
        std::vector<int64_t> start_node((size_t)n * 2);
        int rc = mgta_sdbg_index_edges(g, seqs.data(), n * 2, start_node.data());
        if (rc != MGTA_OK) return rc;

        const double lcp = -std::log(low_cov_penalty);                              // node_enumerator.h:42
        std::vector<double> exit_prob(3000);
        for (int i = 0; i < 3000; ++i) exit_prob[i] = std::log(2.0 / (i + 2)) * 2;  // hmm_graph_search.h:48-52

        DevBuf d_kmers, d_ss, d_sn, d_exit, d_queue, d_sides, d_out, d_len, d_status, d_todo[2];
        const uint32_t out_cap = (uint32_t)(3 * (2 * std::max(fwd->M, rev->M) + 64));
        d_kmers.alloc((size_t)n * klen); d_ss.alloc(n * 4); d_sn.alloc(n * 16); d_exit.alloc(3000 * 8); d_queue.alloc(16);
        d_sides.alloc((size_t)n * 2 * sizeof(mgta_astar_side)); d_out.alloc((size_t)n * 2 * out_cap); d_len.alloc(n * 8);
        d_status.alloc(n * 8);
        MGTA_HIP_CHECK(hipMemcpyAsync(d_kmers.p, kmers, (size_t)n * klen, hipMemcpyHostToDevice, st));
        MGTA_HIP_CHECK(hipMemcpyAsync(d_ss.p, start_state, n * 4, hipMemcpyHostToDevice, st));
        MGTA_HIP_CHECK(hipMemcpyAsync(d_sn.p, start_node.data(), n * 16, hipMemcpyHostToDevice, st));
        MGTA_HIP_CHECK(hipMemcpyAsync(d_exit.p, exit_prob.data(), 3000 * 8, hipMemcpyHostToDevice, st));
        MGTA_HIP_CHECK(hipMemsetAsync(d_status.p, 0, n * 8, st));

        AstarArgs a;
        memset(&a, 0, sizeof(a));
        a.g = g->dev;
        const mgta_hmm *hm[2] = {fwd, rev};
        size_t lds_bytes = 0;
        for (int d = 0; d < 2; ++d) {
            a.hm[d].tab = hm[d]->tab.as<double>(); a.hm[d].M = hm[d]->M; a.hm[d].A = hm[d]->A;
            a.hm[d].col_fwd = hm[d]->d_col.as<int8_t>();
            a.hm[d].col_enum = hm[d]->d_col.as<int8_t>() + 64 * d;
            lds_bytes = std::max(lds_bytes, hm[d]->n_doubles * 8);
        }
        a.kmers = d_kmers.as<char>(); a.start_state = d_ss.as<int32_t>(); a.start_node = d_sn.as<int64_t>();
        a.n_seeds = n; a.klen = klen; a.prune = prune_len; a.low_cov_penalty = lcp; a.log2v = std::log(2.0);
        a.exit_prob = d_exit.as<double>();
        a.queue = d_queue.as<unsigned long long>();
        DevBuf d_prof;
        d_prof.alloc(64);
        MGTA_HIP_CHECK(hipMemsetAsync(d_prof.p, 0, 64, st));
        a.prof = d_prof.as<unsigned long long>();
        a.sides = d_sides.as<mgta_astar_side>(); a.out_seq = d_out.as<char>(); a.out_cap = out_cap; a.out_len = d_len.as<uint32_t>();
        a.status = d_status.as<int32_t>();
        a.use_lds = lds_bytes + sizeof(HeapEnt) * (kLdsHeap + 1) * kAstarWaves + 2048 <= 160 * 1024;   // tables + LDS heap tops
        if (a.use_lds)
            MGTA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(astar_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds_bytes));

        std::vector<int64_t> todo[2];
        for (int d = 0; d < 2; ++d) { todo[d].resize(n); for (int64_t s = 0; s < n; ++s) todo[d][s] = s; }
        std::vector<int32_t> h_status((size_t)n * 2);
        uint32_t cap_nodes = 1u << 17;                                              // cold: searches that still overflow are re-run with 8x
        // warm modes: seeds must run in order in ONE launch (no re-runs), so the arenas are sized generously up front
        DevBuf d_cache[2], d_run_seed, d_run_progress, d_start_limit;
        a.window = cache_mode;
        a.cost_rate = cache_mode > 0 ? ctx->search_cost_rate : 0;
        if (cache_mode > 0) {
            cap_nodes = 1u << 18;
            for (int d = 0; d < 2; ++d) {
                uint64_t want = 2ull * (uint64_t)n * (2ull * (uint64_t)hm[d]->M + 64), cap = 1024;
                while (cap < want) cap <<= 1;
                d_cache[d].alloc(cap * sizeof(CacheEnt), &ctx->live_bytes, &ctx->peak_bytes);
                MGTA_HIP_CHECK(hipMemsetAsync(d_cache[d].p, 0, cap * sizeof(CacheEnt), st));
                a.cache[d] = d_cache[d].as<CacheEnt>(); a.cache_mask[d] = cap - 1;
            }
            d_start_limit.alloc(32);                                                // [0..1] limit per direction, [2..3] scan lock
            MGTA_HIP_CHECK(hipMemsetAsync(d_start_limit.p, 0, 32, st));
            a.start_limit = d_start_limit.as<unsigned long long>();
        }
        for (int attempt = 0; attempt < (cache_mode > 0 ? 1 : 5); ++attempt) {
            int64_t work = (int64_t)std::max(todo[0].size(), todo[1].size());
            if (work == 0) break;
            // persistent grid: one workgroup per CU and direction pair, fewer when there is little work
            int blocks = std::min<int64_t>((int64_t)ctx->num_cus * (a.use_lds ? 1 : 2), 2 * ((work + kAstarWaves - 1) / kAstarWaves));
            if (cache_mode > 0) {   // without the cost term at most `window` searches of a direction can be in flight; cap the arena footprint
                if (a.cost_rate == 0) blocks = std::min<int64_t>(blocks, 2 * (((int64_t)cache_mode + kAstarWaves - 1) / kAstarWaves));
                size_t free_b = 0, total_b = 0;
                MGTA_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
                uint64_t per_slot = (uint64_t)cap_nodes * (sizeof(ANode) + sizeof(HeapEnt) + 2 * sizeof(HashEnt));
                while (blocks > 2 && (uint64_t)blocks * kAstarWaves * per_slot > free_b * 0.6) blocks -= 2;
            }
            blocks = std::max(2, blocks + (blocks & 1));
            uint64_t slots = (uint64_t)blocks * kAstarWaves;
            uint32_t cap_hash = cap_nodes * 2;
            if (cache_mode > 0) {
                d_run_seed.alloc(slots * 8); d_run_progress.alloc(slots * 8);
                MGTA_HIP_CHECK(hipMemsetAsync(d_run_seed.p, 0xFF, slots * 8, st));
                MGTA_HIP_CHECK(hipMemsetAsync(d_run_progress.p, 0, slots * 8, st));
                a.run_seed = d_run_seed.as<long long>(); a.run_progress = d_run_progress.as<unsigned long long>(); a.n_slots = (uint32_t)slots;
            }
            // per-search arenas live in the context between calls: hash entries are tag-versioned and the tag counters
            // persist, so a re-used arena needs neither clearing nor re-allocation (only a geometry change does)
            AstarArenas &ar = ctx->astar;
            if (ar.slots != slots || ar.cap_nodes != cap_nodes || !ar.nodes.p) {
                ar.nodes.release(); ar.heap.release(); ar.hash.release(); ar.tag.release();
                ar.nodes.alloc(slots * cap_nodes * sizeof(ANode), &ctx->live_bytes, &ctx->peak_bytes);
                ar.heap.alloc(slots * cap_nodes * sizeof(HeapEnt), &ctx->live_bytes, &ctx->peak_bytes);
                ar.hash.alloc(slots * cap_hash * sizeof(HashEnt), &ctx->live_bytes, &ctx->peak_bytes);
                ar.tag.alloc(slots * 4, &ctx->live_bytes, &ctx->peak_bytes);
                MGTA_HIP_CHECK(hipMemsetAsync(ar.hash.p, 0, slots * cap_hash * sizeof(HashEnt), st));
                MGTA_HIP_CHECK(hipMemsetAsync(ar.tag.p, 0, slots * 4, st));
                ar.slots = slots; ar.cap_nodes = cap_nodes;
            }
            DevBuf &d_nodes = ar.nodes, &d_heap = ar.heap, &d_hash = ar.hash, &d_tag = ar.tag;
            MGTA_HIP_CHECK(hipMemsetAsync(d_queue.p, 0, 16, st));
            for (int d = 0; d < 2; ++d) {
                d_todo[d].alloc(std::max<size_t>(1, todo[d].size()) * 8);
                if (!todo[d].empty()) MGTA_HIP_CHECK(hipMemcpyAsync(d_todo[d].p, todo[d].data(), todo[d].size() * 8, hipMemcpyHostToDevice, st));
                a.todo[d] = d_todo[d].as<int64_t>(); a.n_todo[d] = (int64_t)todo[d].size();
            }
            a.nodes = d_nodes.as<ANode>(); a.heap = d_heap.as<HeapEnt>(); a.hash = d_hash.as<HashEnt>();
            a.cap_nodes = cap_nodes; a.cap_hash = cap_hash; a.slot_tag = d_tag.as<uint32_t>();
            MGTA_HIP_CHECK(hipEventRecord(evk0, st));
            if (a.use_lds) hipLaunchKernelGGL((astar_kernel<true>), dim3(blocks), dim3(kAstarThreads), lds_bytes, st, a);
            else hipLaunchKernelGGL((astar_kernel<false>), dim3(blocks), dim3(kAstarThreads), 0, st, a);
            MGTA_HIP_CHECK(hipEventRecord(evk1, st));
            MGTA_HIP_CHECK(hipMemcpyAsync(h_status.data(), d_status.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
            MGTA_HIP_CHECK(hipStreamSynchronize(st));
            MGTA_HIP_CHECK(hipGetLastError());
            float ms = 0;
            MGTA_HIP_CHECK(hipEventElapsedTime(&ms, evk0, evk1));
            ST.ms_kernel += ms;
            for (int d = 0; d < 2; ++d) {
                std::vector<int64_t> again;
                for (int64_t s : todo[d]) if (h_status[(size_t)s * 2 + d] == 2) again.push_back(s);
                todo[d].swap(again);
            }
            ST.n_retries += (int64_t)(todo[0].size() + todo[1].size());
            cap_nodes *= 8;                                                         // bigger arenas for the searches that overflowed
        }
        if (!todo[0].empty() || !todo[1].empty()) {
            set_error("%zu searches still overflow their arena (%u nodes) after all retries", todo[0].size() + todo[1].size(), cap_nodes / 8);
            return MGTA_EOVERFLOW;
        }
        for (int64_t s = 0; s < n * 2; ++s)
            if (h_status[(size_t)s] == 4 || h_status[(size_t)s] == 0) {
                set_error("search %lld did not run (ordered-commit gate timed out)", (long long)s);
                return MGTA_EHIP;
            }
#ifdef MGTA_ASTAR_PROFILE
        {
            unsigned long long hp[8];
            MGTA_HIP_CHECK(hipMemcpy(hp, d_prof.p, 64, hipMemcpyDeviceToHost));
            const char *nm[8] = {"setup", "pop+closedprobe", "node+closedins", "cache", "graph", "children+probe", "commit", "result"};
            unsigned long long tot = 0;
            for (int q = 0; q < 8; ++q) tot += hp[q];
            for (int q = 0; q < 8; ++q) fprintf(stderr, "[astar-prof] %-16s %6.2f %%\n", nm[q], 100.0 * hp[q] / (tot ? tot : 1));
        }
#endif
        // results
        std::vector<mgta_astar_side> h_sides((size_t)n * 2);
        std::vector<uint32_t> h_len((size_t)n * 2);
        std::vector<char> h_out((size_t)n * 2 * out_cap);
        MGTA_HIP_CHECK(hipMemcpyAsync(h_sides.data(), d_sides.p, h_sides.size() * sizeof(mgta_astar_side), hipMemcpyDeviceToHost, st));
        MGTA_HIP_CHECK(hipMemcpyAsync(h_len.data(), d_len.p, h_len.size() * 4, hipMemcpyDeviceToHost, st));
        MGTA_HIP_CHECK(hipMemcpyAsync(h_out.data(), d_out.p, h_out.size(), hipMemcpyDeviceToHost, st));
        MGTA_HIP_CHECK(hipEventRecord(ev1, st));
        MGTA_HIP_CHECK(hipStreamSynchronize(st));
        float ms = 0;
        MGTA_HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1));
        ST.ms_total = ms;
        (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1); (void)hipEventDestroy(evk0); (void)hipEventDestroy(evk1);
        std::string left;
        for (int64_t s = 0; s < n; ++s) {
            for (int d = 0; d < 2; ++d) {
                if (h_status[(size_t)s * 2 + d] == 3) {
                    set_error("seed %lld: k-mer / model position outside the model (start_state %d)", (long long)s, start_state[s]);
                    return MGTA_EINVAL;
                }
                ST.n_expansions += h_sides[(size_t)s * 2 + d].n_expanded;
                ST.n_opened += h_sides[(size_t)s * 2 + d].n_opened;
            }
            if (sink) {
                const char *r = h_out.data() + (size_t)(2 * s) * out_cap;
                const char *l = h_out.data() + (size_t)(2 * s + 1) * out_cap;
                uint32_t ll = h_len[(size_t)2 * s + 1];
                left.assign(ll, ' ');
                for (uint32_t i = 0; i < ll; ++i) {                                  // RevComp, hmm_graph_search.h:362-398
                    char c = l[ll - 1 - i];
                    left[i] = c == 'a' ? 't' : c == 'c' ? 'g' : c == 'g' ? 'c' : c == 't' ? 'a' : c;
                }
                int src = sink(user, s, left.data(), (int64_t)ll, r, (int64_t)h_len[(size_t)2 * s], &h_sides[(size_t)2 * s],
                               &h_sides[(size_t)2 * s + 1]);
                if (src != 0) { set_error("contig sink returned %d", src); return MGTA_ESINK; }
            }
        }
        if (stats) *stats = ST;
        return MGTA_OK;
    } catch (const HipError &e) { return e.code; }
}

}  // extern "C"
