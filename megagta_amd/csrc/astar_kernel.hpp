// astar_kernel.hpp — device side of the batched HMM-guided A* (included by astar.hip only).
//
// Mapping (gfx950, wave64): a wavefront carries 64 / G searches at once, G lanes each (G = 16 by default: one DPP row).
// The searches of a wave run in lockstep through the phases of one expansion (pop, closed-set probe, graph walk, child scoring,
// open-set probes, ordered commit), so the instruction stream that used to serve one search serves four, and the dependent
// memory round trips of the four overlap.  Inside a search:
//   * frontier expansion: lane (i, j) of the group walks OutgoingEdges(curr)[i] -> [j] and owns the <= 4 codon paths that
//     continue from there (k = 0..3); reference order of the children = ascending (lane, k), match before insert, delete last;
//   * open list = binary heap with libstdc++'s exact push_heap / pop_heap sift sequence (equal priorities leave in the
//     reference's order); the top levels sit in LDS; a pop reads log2(G) levels of sibling PAIRS per memory round trip;
//   * closed set + open_hash: one open-addressing table, empty = key 0;
//   * node pool, heap and hash table GROW IN PLACE (PoolST / HashMapST of the reference have no bound: pool_st.h:43,
//     hash_table_st.h:559-568): beyond the slot's base arena one 2 MB page at a time from a device-side pool (bump pointer + one free
//     list of pages); the hash table by linear hashing, one page-sized bucket split per step.  Nothing is ever re-run for lack of a
//     LARGE piece of memory; what happens when the pool itself runs dry is described at AstarArgs / in astar.hip.
#pragma once
#include "common.hpp"
#include "device_utils.hpp"
#include "graph.hpp"

namespace mgta {

#ifndef MGTA_ASTAR_WAVES
#define MGTA_ASTAR_WAVES 8
#endif
constexpr int kAstarWaves = MGTA_ASTAR_WAVES;                 // waves per workgroup (one workgroup per CU: the HMM tables take most of the LDS)
constexpr int kAstarThreads = kAstarWaves * 64;
constexpr uint32_t kNone = 0x7FFFFFFFu;
constexpr int kMaxKmer = 160;
// heap slots of a search kept in LDS: the root block and its eight child blocks (six tree levels); with 8 lanes per search (twice the
// searches per workgroup) the root block and four child blocks, so that the HMM tables of a 360-column model still fit beside them
#ifdef MGTA_ASTAR_LDS_HEAP
constexpr uint32_t lds_heap_slots(int G) { return MGTA_ASTAR_LDS_HEAP; }
#else
constexpr uint32_t lds_heap_slots(int G) { return G >= 16 ? 72u : 40u; }
#endif
constexpr int kUnitLog = 12;                   // pool offsets are kept in 4 KB units
constexpr int kNumClasses = 28;                // chunk size classes: 4 KB << c
constexpr uint32_t kNoChunk = 0xFFFFFFFFu;
// PAGES.  Beyond its base arena a search's node array, heap and hash table grow ONE 2 MB PAGE AT A TIME; the pages of an array are named
// by a small table (the first 16 in LDS, the rest in a 4 KB chunk).  Rounds 2-3 doubled every array with ever larger chunks from per-size
// free lists: late in a batch the free memory sat in the lists of the sizes the ended searches had used, and a search that needed its
// next 64 MB ... 1 GB in one piece waited for ever next to 80 GB of free 1-16 MB chunks (50 M reads, nirK: 141.7 GB handed out once, 55 GB
// in use, a thousand searches waiting, only the reserve's owner moving).  Paged LEVELS (a table per level, all or nothing) cured that and
// brought a livelock instead: thousands of searches taking 20 of the 64 pages they needed, failing, giving them back (100 M reads: the
// whole device at 250 expansions a second).  With one page per step every page that comes back lets some search go on, nothing is ever
// asked for and returned again, a search holds what it uses plus at most a page per array, and there is one size of chunk: no
// fragmentation at all.  The hash table grows by LINEAR HASHING (Litwin 1980): buckets of one page each, split one at a time.
constexpr int kPageClass = 9;                  // 4 KB << 9 = 2 MB
constexpr int kPageLog = kPageClass + kUnitLog;
constexpr int kLdsPages = 16;                  // pages of an array named in LDS (32 MB); beyond: a 4 KB table chunk (1024 pages = 2 GB per array)
constexpr int kMaxPages = 1024;
constexpr int kPtWords = 3 * kLdsPages + 4;    // LDS words per search: three page tables + the three table chunks' units (+ 1 spare)
constexpr uint32_t kStage = 8;                 // open-list entries of one walk pass staged in LDS in commit order (more than these: the lane-by-lane loop)
constexpr uint32_t kStarveLimit = 1u << 15;    // iterations a search waits for memory before it gives up (about a second)
constexpr uint32_t kMaxNew = 132;              // children one expansion can open (64 codons x {match, insert} + delete), rounded up

enum { T_MM = 0, T_MI = 1, T_MD = 2, T_IM = 3, T_II = 4, T_DM = 5, T_DD = 6 };   // profile_hmm.h:25
enum { ST_M = 0, ST_I = 1, ST_D = 2 };
enum { S_IDLE = 0, S_WAIT = 1, S_START = 2, S_RUN = 3, S_DONE = 4, S_EXIT = 5, S_BACKOFF = 6 };

struct ANode {                    // AStarNode, a_star_node.h:9-33
    double score, real_score, max_score;
    int64_t node_id;
    int32_t parent;               // index in the search's pool, -1 = none
    int32_t fval;
    int16_t state_no, length, negative_count;
    uint16_t em_state;            // nucl_emission (9 bits) | state << 9
    int64_t fwd_r;                // forward descriptor of node_id (graph.hpp FwdDesc): the expansion of this node starts at the TARGET line of
    uint32_t fwd_hint;            // its edge instead of loading the edge's own line first (kFdNone: not known, e.g. the seed's start edge)
    uint32_t pad;                 // one node = one aligned 64-byte sector: a node access is a single request
};
static_assert(sizeof(ANode) == 64, "node layout");

struct HeapEnt {                  // 16 bytes; the priority (fval, -state_no, state rank) is rebuilt from key + fval
    uint64_t key;                 // node_id << 18 | state_no << 2 | (state + 1)
    int32_t fval;
    uint32_t node;                // index in the search's node pool
};
struct HashEnt {
    uint64_t key;                 // node_id << 18 | state_no << 2 | (state + 1); 0 = empty
    uint32_t val;                 // open-list node index (kNone = none) | closed << 31
    int32_t fval;                 // fval of that node: the admission test `got->second < next` (hmm_graph_search.h:299-302) is answered by
                                  // the probe itself, without a second dependent fetch of the node
};
static_assert(sizeof(HeapEnt) == 16 && sizeof(HashEnt) == 16, "entry layout");

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

// Device-side chunk pool.  Every word below is touched by agent-scope atomics only (the allocator is shared by all CUs).
struct PoolDev {
    char *base;
    unsigned long long bytes;     // size of the region the bump pointer may hand out
    unsigned long long *bump;     // next never-used byte offset
    unsigned int *lock, *cnt;     // [kNumClasses] spin lock and number of free chunks per class
    unsigned int *stack;          // free chunks (4 KB units) of class c at stack[meta[c] .. meta[c] + meta[kNumClasses + c])
    const unsigned int *meta;
    unsigned long long *stat;     // [0] chunks served by the free lists, [1] failed allocations, [2] re-hashes, [3] searches that grew,
                                  // [4] bytes handed out and not yet returned, [5] its high-water mark (sampled when a search starts),
                                  // [6] searches that gave their memory back and started again in place (ordered launches only)
    unsigned long long soft_limit;   // no new search starts while more than this is in use: the ones that run keep room to grow
    // The RESERVE: the last reserve_bytes of the pool, behind `bytes`, with a bump pointer of its own and no free lists.  Only the LOWEST
    // running search of an ordered launch takes chunks from it, and only when the lists and the bump pointer above have nothing for it
    // (late in a batch the free memory sits in lists of other sizes, and the searches that hold the rest run on without needing more):
    // the one search every later seed waits for always finds room.  One owner at a time (the owner word is claimed by CAS: the two
    // directions take their seeds from two queues, so a search can be the lowest running one for a while and then see a lower one
    // start); the owner resets the bump pointer when it ends or yields; chunks of the reserve never enter a free list.
    unsigned long long reserve_off, reserve_bytes;
    unsigned long long *rbump;    // [0] next never-used byte of the reserve, [1] its high-water mark over the launch, [2] owner (search id + 1, 0 = free)
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
    PoolDev pool;
    unsigned long long base_off;  // byte offset of slot 0's base arena in the pool; slot s owns [base_off + s * slot_bytes, ...)
    unsigned long long slot_bytes;
    int log_b0;                   // base arena = 1 << log_b0 nodes, 2 << log_b0 heap slots and 2 << log_b0 hash entries
    mgta_astar_side *sides;       // [2n]
    char *out_seq; uint32_t out_cap; uint32_t *out_len;   // [2n]
    int32_t *status;              // [2n] 0 = pending, 1 = done, 2 = pool exhausted, 3 = bad seed, 4 = gate timeout, 5 = one array of the search reached
                                  // kMaxPages pages (2 GB beyond the base arena: ~33 M nodes): the library's limit, not the device's
    // shared term_nodes caches (search.cpp:182), one per direction.  window = 0: off (cold).  window = B >= 1: the path found by
    // seed j (c_j expansions) is seen by exactly the seeds >= j + B + c_j / cost_rate (cost_rate = 0: no cost term; B = 1 then is
    // the reference's sequential run).  The cost term lets later seeds start while a long search is still running: it cannot
    // become visible to them any more, however soon it ends.
    int window;
    int cost_rate;
    // the cost term is CONCAVE: c expansions delay a path by c / cost_rate seeds up to cost_knee expansions and by 1 / cost_rate2 seeds per
    // expansion beyond (cost_knee = 0: one rate throughout).  A first-of-its-gene-copy search of millions of expansions then stays
    // invisible for tens of thousands of seeds instead of millions -- with one rate every later seed of that copy explored it cold again --
    // while the seeds right behind an ordinary search still start without waiting for it.
    unsigned long long cost_knee;
    int cost_rate2;
    int free_share;               // 1 = every path is visible to every search from the moment it is inserted (the reference's multi-thread
                                  // behaviour: fastest, but the result depends on timing); no gate
    int gate;                     // 1 = seeds start in order behind the commit frontier (the normal shared-cache launch);
                                  // 0 with window > 0 = a re-run of searches the pool could not hold: the caches are read with the
                                  // same visibility rule but nothing waits
    CacheEnt *cache[2];
    uint64_t cache_mask[2];
    uint32_t cache_probe_limit;   // an insert gives up after this many probes (a missed entry is always correct)
    long long *run_seed;          // [slots] seed a search slot is working on (a lower bound while it is taking one from the queue), -1 = none
    unsigned long long *run_progress;   // [slots] expansions of that search so far (lags; only ever too small)
    unsigned long long *start_limit;    // [0..1] highest seed index known to be allowed to start (monotone cache of the gate), [2..3] time of the last
                                        // refresh by a waiting wave, [4] the pass has given up (no further seed is taken), [5] the call
                                        // for memory (call_for_memory)
    uint32_t n_slots;
    uint32_t blocks_dir0;         // workgroups [0, blocks_dir0) search direction 0 (the k-mer, forward model), the rest direction 1: split by the work the
                                  // seeds' model positions promise (a forward search covers M - s columns, a reverse one s), not in halves
    unsigned long long *prof;     // [16] per-phase cycle sums (MGTA_ASTAR_PROFILE builds only)
    unsigned long long *tmark;    // s_memrealtime ticks (10 ns), kept by atomic minimum, all-ones = never: [0] the launch's first workgroup at work, [1 + dir] the
                                  // first slot of the direction that found its queue empty (from then on the launch only finishes what is in flight: the tail)
    uint32_t ramp_base;           // ordered launches: searches in flight per direction before any has ended (slow start)
    int auto_unorder;             // ordered launches, OPT-IN (MEGAGTA_SEARCH_ALLOW_UNORDERED=1): when the searches in flight have outgrown the pool (thousands of refused requests) the
                                  // batch gives up the ORDER, not the searches: start_limit[14] is set, from then on every path is visible
                                  // to every search as soon as it is inserted and no seed waits at the gate -- the reference's multi-thread
                                  // behaviour (search.cpp:182-189).  Holding the order there means thousands of long searches waiting for
                                  // each other's memory while the seeds behind them wait for their progress (50 M reads, nirK on the multi-k
                                  // graph: 79 000 of 300 000 seeds taken after 150 s; without the order the 300 000 end in 94 s).
    uint32_t active_slots;        // search slots per workgroup that take seeds (all of them; 1 in the last-resort pass: one search per
                                  // direction at a time, with the whole pool to itself)
};

__device__ __forceinline__ int to_fval(double x) {   // (int)x as x86-64 cvttsd2si does it (INT_MIN when out of range / NaN)
    if (!(x > -2147483649.0 && x < 2147483648.0)) return (int)0x80000000;
    return (int)x;
}
__device__ __forceinline__ int srank(int st) { return st == ST_M ? 3 : st == ST_D ? 2 : 1; }
__device__ __forceinline__ uint64_t make_key(int64_t node_id, int state_no, int st) {
    return ((uint64_t)node_id << 18) | ((uint64_t)(uint16_t)state_no << 2) | (uint64_t)(st + 1);
}
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ uint64_t ent_prio(const HeapEnt &e) {   // AStarNode::operator< (a_star_node.h:34-82) as one integer: a < b <=> prio(a) < prio(b)
    int st = (int)(e.key & 3) - 1;
    return ((uint64_t)((uint32_t)e.fval ^ 0x80000000u) << 32) | ((uint64_t)(uint16_t)(0xFFFF - (uint16_t)((e.key >> 2) & 0xFFFF)) << 2) |
           (uint64_t)srank(st);
}

// ---- agent-scope accesses: protocol words, caches and the pool's bookkeeping are written by other CUs (and XCDs) of the same
// launch.  They are only ever touched by atomics performed at the coherence point, ordered by waiting for the returning atomic
// before the next one is issued.  No acquire / release fences on polled words: on this part an agent-scope acquire invalidates,
// and a release writes back, cache contents of the whole CU / XCD, and a gate that does that at polling rate slows every running
// search by an order of magnitude (measured in round 1: 5.3 s -> 257 s).
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) {
    return __hip_atomic_fetch_add(const_cast<unsigned long long *>(p), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned int ld_agent(const unsigned int *p) {
    return __hip_atomic_fetch_add(const_cast<unsigned int *>(p), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(long long *p, long long v) {
    (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v) {
    (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void st_agent(unsigned int *p, unsigned int v) {
    (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- chunk pool ------------------------------------------------------------------------------------------------------------
// Any subset of the lanes of a wave may call pool_alloc / pool_free, each for itself.  The callers of one wave take turns (one lane
// at a time runs the whole lock / unlock sequence): two lanes of a wave must never compete for a lock, because a lane that has left
// a spin loop waits at the reconvergence point for the lanes still in it -- with the lock in its hands (and the compiler is free to
// move a critical section behind the loop that guards it).  A single spinning lane only ever waits for OTHER waves, which run on.
// A chunk is named by its first 4 KB unit (28 bits: pools up to 1 TB); bits 28-29 say how many classes LARGER than asked for the
// chunk really is (it goes back to the list it came from).
constexpr uint32_t kUnitMask = 0x0FFFFFFFu;
constexpr int kBorrowShift = 28;
__device__ __forceinline__ uint32_t pool_pop(const PoolDev &P, int c) {
    uint32_t res = kNoChunk;
    if (ld_agent(&P.cnt[c]) == 0u) return res;
    for (int spins = 0; spins < (1 << 20); ++spins) {
        unsigned int expect = 0u;
        if (__hip_atomic_compare_exchange_strong(&P.lock[c], &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            const unsigned int n = ld_agent(&P.cnt[c]);
            if (n > 0u) {
                res = ld_agent(&P.stack[P.meta[c] + n - 1u]);
                st_agent(&P.cnt[c], n - 1u);
            }
            st_agent(&P.lock[c], 0u);
            break;
        }
        __builtin_amdgcn_s_sleep(4);                                  // (bounded: fall through to the bump pointer)
    }
    return res;
}
__device__ __forceinline__ uint32_t pool_alloc_one(const PoolDev &P, int c) {
    uint32_t res = pool_pop(P, c);
    if (res != kNoChunk) __hip_atomic_fetch_add(&P.stat[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (res == kNoChunk) {
        const unsigned long long size = 1ull << (c + kUnitLog);
        unsigned long long old = ld_agent(P.bump);
        while (old + size <= P.bytes) {
            if (__hip_atomic_compare_exchange_strong(P.bump, &old, old + size, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                res = (uint32_t)(old >> kUnitLog);
                break;
            }
        }
    }
    // The bump pointer never comes back and the lists are per size: late in a batch the memory that is free sits in the lists of
    // OTHER sizes.  Take a free chunk of the next two sizes up as it is (nothing is split: it returns to its own list).
    for (int e = 1; e <= 2 && res == kNoChunk && c + e < kNumClasses; ++e) {
        res = pool_pop(P, c + e);
        if (res != kNoChunk) res |= (uint32_t)e << kBorrowShift;
    }
    if (res == kNoChunk) __hip_atomic_fetch_add(&P.stat[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return res;
}
__device__ __forceinline__ uint32_t pool_alloc(const PoolDev &P, int c) {
    uint32_t res = kNoChunk;
    const int lane = lane_id();
    uint64_t turn = __ballot(true);                                   // the calling lanes
    while (turn) {
        const int l = __builtin_ctzll(turn);
        turn &= turn - 1;
        if (lane == l) res = pool_alloc_one(P, c);
    }
    // a chunk another CU may have used: drop whatever this CU's L1 still holds of it.  Once per chunk OBTAINED, never per poll: an
    // agent-scope acquire empties the CU's caches under every search that runs there, and a batch whose pool was exhausted had two
    // thousand starved searches asking every iteration (50 M reads, nirK: the one search that could run did 1 500 expansions a second)
    if (__ballot(res != kNoChunk) != 0ull) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (res != kNoChunk)                                                  // bytes in use
        __hip_atomic_fetch_add(&P.stat[4], 1ull << (c + (int)((res >> kBorrowShift) & 3u) + kUnitLog), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return res;
}
// The caller has issued pool_release_fence() since its last store into the chunk.
__device__ __forceinline__ void pool_free(const PoolDev &P, int c, uint32_t unit) {
    const int lane = lane_id();
    c += (int)((unit >> kBorrowShift) & 3u);                          // a borrowed chunk goes back to its own list
    unit &= kUnitMask;
    if (P.reserve_bytes != 0ull && ((unsigned long long)unit << kUnitLog) >= P.reserve_off) return;   // the reserve is reset as a whole by its owner
    __hip_atomic_fetch_sub(&P.stat[4], 1ull << (c + kUnitLog), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t turn = __ballot(true);
    while (turn) {
        const int l = __builtin_ctzll(turn);
        turn &= turn - 1;
        if (lane == l) {
            for (int spins = 0; spins < (1 << 20); ++spins) {
                unsigned int expect = 0u;
                if (__hip_atomic_compare_exchange_strong(&P.lock[c], &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    const unsigned int n = ld_agent(&P.cnt[c]);
                    if (n < P.meta[kNumClasses + c]) {                 // (a full list leaks the chunk: only possible with toy chunk sizes)
                        st_agent(&P.stack[P.meta[c] + n], unit);
                        st_agent(&P.cnt[c], n + 1u);
                    }
                    st_agent(&P.lock[c], 0u);
                    break;
                }
                __builtin_amdgcn_s_sleep(4);                          // (bounded: the chunk is given up rather than the wave hung)
            }
        }
    }
}
// Chunks travel between CUs of different XCDs, whose L2s are not coherent with each other for plain stores: write this XCD's
// dirty lines back before a chunk is offered to the others (once per search that grew, never per expansion).
__device__ __forceinline__ void pool_release_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// A chunk of the reserve (see PoolDev): called by lane 0 of the one search that owns it.
__device__ __forceinline__ uint32_t reserve_alloc(const PoolDev &P, int c) {
    const unsigned long long size = 1ull << (c + kUnitLog);
    const unsigned long long old = __hip_atomic_fetch_add(&P.rbump[0], size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + size > P.reserve_bytes) {
        __hip_atomic_fetch_sub(&P.rbump[0], size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return kNoChunk;
    }
    __hip_atomic_fetch_max(&P.rbump[1], old + size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                // an earlier owner ran on another CU
    return (uint32_t)((P.reserve_off + old) >> kUnitLog);
}

// lane 0 of a search that has found itself the lowest running one: claim the reserve (false: an earlier lowest search still holds it)
__device__ __forceinline__ bool reserve_claim(const PoolDev &P, long long sid) {
    unsigned long long expect = 0ull;
    return __hip_atomic_compare_exchange_strong(&P.rbump[2], &expect, (unsigned long long)sid + 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the owner ends or yields (after pool_release_fence(): its dirty lines are written back): everything in the reserve is free again
__device__ __forceinline__ void reserve_release(const PoolDev &P) {
    st_agent(&P.rbump[0], 0ull);
    st_agent(&P.rbump[2], 0ull);
}

// One piece of the pool for lane 0 of a search: from the lists / the bump pointer, else (the lowest running search) from the reserve.
__device__ __forceinline__ uint32_t chunk_alloc(const PoolDev &P, int c, bool use_reserve) {
    uint32_t u = pool_alloc(P, c);
    if (u == kNoChunk && use_reserve) u = reserve_alloc(P, c);
    return u;
}
// ---- growable arrays of one search ---------------------------------------------------------------------------------------------
// Element i < base_n lives in the slot's base arena, element base_n + j in page j >> (kPageLog - ELEM_LOG) of the array.
template <int ELEM_LOG>
struct PageArr {
    char *slot_base;              // the slot's base arena ...
    uint32_t base_off;            // ... and where the array's part of it begins
    uint32_t base_n;              // elements there
    const uint32_t *pt;           // LDS: units of the first kLdsPages pages
    const uint32_t *gt;           // LDS word: unit of the 4 KB chunk that names the pages from kLdsPages on
    char *pool;
    static constexpr int kPerPageLog = kPageLog - ELEM_LOG;
    __device__ __forceinline__ char *page(uint32_t g) const {
        const uint32_t u = g < (uint32_t)kLdsPages ? pt[g] : reinterpret_cast<const uint32_t *>(pool + ((uint64_t)(*gt & kUnitMask) << kUnitLog))[g];
        return pool + ((uint64_t)(u & kUnitMask) << kUnitLog);
    }
    __device__ __forceinline__ char *at(uint32_t i) const {        // (elements never straddle a page: 16 / 64 / 128-byte units)
        if (i < base_n) return slot_base + (base_off + (i << ELEM_LOG));
        const uint32_t j = i - base_n;
        return page(j >> kPerPageLog) + ((uint64_t)(j & ((1u << kPerPageLog) - 1u)) << ELEM_LOG);
    }
};
using NodeArr = PageArr<6>;
using HeapArr = PageArr<4>;
constexpr uint32_t kHashPerPageLog = kPageLog - 4;                     // 16-byte entries of one page (= one bucket of the paged table)
constexpr uint32_t kHashPerPage = 1u << kHashPerPageLog;
__device__ __forceinline__ ANode load_node(const ANode *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    union { uint4 v[4]; ANode n; } u;
    u.v[0] = q[0]; u.v[1] = q[1]; u.v[2] = q[2]; u.v[3] = q[3];
    return u.n;
}
__device__ __forceinline__ void store_node(ANode *p, const ANode &n) {
    union { uint4 v[4]; ANode n; } u;
    u.n = n;
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = u.v[0]; q[1] = u.v[1]; q[2] = u.v[2]; q[3] = u.v[3];
}
__device__ __forceinline__ HeapEnt load_ent(const HeapEnt *p) {
    const uint4 v = *reinterpret_cast<const uint4 *>(p);
    HeapEnt e;
    e.key = (uint64_t)v.x | ((uint64_t)v.y << 32); e.fval = (int32_t)v.z; e.node = v.w;
    return e;
}
__device__ __forceinline__ void store_ent(HeapEnt *p, const HeapEnt &e) {
    *reinterpret_cast<uint4 *>(p) = make_uint4((uint32_t)e.key, (uint32_t)(e.key >> 32), (uint32_t)e.fval, e.node);
}

// ---- group (= one search) primitives: G consecutive lanes ---------------------------------------------------------------------
template <int G> struct Grp {
    static constexpr int kGroups = 64 / G;
    static constexpr int kLog = G == 64 ? 6 : G == 32 ? 5 : G == 16 ? 4 : 3;
    static constexpr uint32_t kLdsHeap = lds_heap_slots(G);
    // the <= 16 (first edge, second edge) pairs of an expansion are walked by min(G, 16) lanes at a time: G = 8 takes the first edges
    // 0 and 1 in one pass and comes back for 2 and 3 only when the node has them (one node in a few hundred)
    static constexpr int kWalkLanes = G >= 16 ? 16 : G;
    static constexpr int kPasses = 16 / kWalkLanes;
    static constexpr uint64_t kMask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
    __device__ static __forceinline__ uint64_t ballot(bool p, int gbase) { return (__ballot(p) >> gbase) & kMask; }
    template <class T> __device__ static __forceinline__ T bcast(T v, int src, int gbase) { return __shfl(v, gbase + src, 64); }
    // OR over the lanes of the group, in every lane: DPP inside a row of 16 (quad swaps, then the mirrors: after a step all the lanes of a
    // block hold the block's OR, so mirroring a half row / a row brings in the other block's), shuffles beyond
    __device__ static __forceinline__ uint32_t or32(uint32_t x) {
        x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
        x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
        x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true);   // row_half_mirror
        if (G >= 16) x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, true);   // row_mirror
        if (G >= 32) x |= (uint32_t)__shfl_xor((int)x, 16, 64);
        if (G >= 64) x |= (uint32_t)__shfl_xor((int)x, 32, 64);
        return x;
    }
    __device__ static __forceinline__ uint64_t or64(uint64_t x) { return ((uint64_t)or32((uint32_t)(x >> 32)) << 32) | or32((uint32_t)x); }
};

// The open list is libstdc++'s binary heap LOGICALLY (index i, children 2i+1 and 2i+2: the sift sequences below are __push_heap /
// __adjust_heap word for word, so entries of equal priority leave in the reference's order), but entry i is STORED in blocks of
// three tree levels: one 128-byte line holds a node, its two children and its four grandchildren (slots 1..7 of the block, slot 0
// unused), blocks of one block level in tree order.  A walk down the tree then touches one line per three levels instead of one per
// level, and the two lines below a block's leaf are adjacent.
__device__ __forceinline__ uint32_t heap_slot(uint32_t i) {
    const uint32_t j = i + 1;
    const int l = 31 - __builtin_clz(j);
    const int b = (l * 11) >> 5, r = l - 3 * b;                        // block level (l / 3) and level inside the block
    const uint32_t lvl = 1u << (3 * b);
    const uint32_t blk = (0x49249249u & (lvl - 1u)) + ((j >> r) - lvl);   // blocks of the levels above: (8^b - 1) / 7
    return blk * 8u + ((1u << r) | (j & ((1u << r) - 1u)));
}
// slots a heap of n entries needs (the deepest block level is complete as soon as its second tree level has begun)
__device__ __forceinline__ uint32_t heap_slots_needed(uint32_t n) {
    const int l = 31 - __builtin_clz(n | 1u);
    const int b = (l * 11) >> 5, r = l - 3 * b;
    const uint32_t lvl = 1u << (3 * b);
    const uint32_t blocks = (0x49249249u & (lvl - 1u)) + (r == 0 ? n - lvl + 1u : lvl);
    return blocks * 8u;
}

template <int G> struct Heap {
    HeapEnt *lds;                 // this search's first Grp<G>::kLdsHeap slots
    HeapArr ar;
    int gl, gbase;
    __device__ __forceinline__ HeapEnt get(uint64_t i) const {
        const uint32_t s = heap_slot((uint32_t)i);
        return s < Grp<G>::kLdsHeap ? lds[s] : load_ent(reinterpret_cast<const HeapEnt *>(ar.at(s)));
    }
    __device__ __forceinline__ void set(uint64_t i, const HeapEnt &e) const {
        const uint32_t s = heap_slot((uint32_t)i);
        if (s < Grp<G>::kLdsHeap) lds[s] = e; else store_ent(reinterpret_cast<HeapEnt *>(ar.at(s)), e);
    }
    // __push_heap(first, hole, 0, v) (bits/stl_heap.h): every lane of the group calls it with the same arguments
    __device__ __forceinline__ void sift_up(uint64_t hole, const HeapEnt &v) const {
        const int depth = 63 - __builtin_clzll(hole + 1);              // number of ancestors of `hole`
        const uint64_t pv = ent_prio(v);
        int a0 = 0;                                                    // ancestors a0+1 .. a0+G are looked at together
        while (true) {
            const int a = a0 + gl + 1;
            HeapEnt e = v;
            bool less = false;
            if (a <= depth) {
                e = get(((hole + 1) >> a) - 1);
                less = ent_prio(e) < pv;
            }
            const uint64_t bal = Grp<G>::ballot(less, gbase);
            int s = bal == Grp<G>::kMask ? G : __builtin_ctzll(~bal);  // ancestors of this batch that move down one level
            if (gl < s) set(((hole + 1) >> (a - 1)) - 1, e);
            if (s == G && a0 + G < depth) { a0 += G; continue; }
            if (gl == 0) set(((hole + 1) >> (a0 + s)) - 1, v);
            break;
        }
    }
    // pop_heap + pop_back; n = current size (> 0); returns the former top.
    // __adjust_heap walks down from the root moving the larger child up (the right one unless right < left).  Three levels per
    // memory round trip (two below the root: the rounds then stay aligned with the blocks, and a round reads the two adjacent child
    // blocks of the hole): lane l < 7 holds one PAIR of siblings of the subtree under the hole (level t = floor(log2(l+1)) + 1,
    // pair p = l + 1 - 2^(t-1)), decides locally which of the two its parent would pick, a chain of ballots tells which lanes lie
    // on the path, and those lanes move their entries up at once.
    __device__ __forceinline__ void remove_top(uint32_t n) const {      // (the caller has read the top: get(0))
        if (n > 1) {
            const HeapEnt v = get(n - 1);
            const int64_t len = (int64_t)n - 1;
            const int64_t half = (len - 1) / 2;                        // nodes below `half` have two children
            int64_t hole = 0;
            const int t = 32 - __builtin_clz((unsigned)gl + 1);        // 1, 2, 2, 3, 3, 3, 3 for lanes 0 .. 6
            const int p = gl + 1 - (1 << (t - 1));
            const int parent_lane = t > 1 ? (1 << (t - 2)) - 1 + (p >> 1) : 0;
            int levels = 2;                                            // the root block holds two levels below the root
            // the entry that ends up right above the final hole is the one the last step moved there: known without a load, and it is
            // all __push_heap needs to look at when the former last entry stays below it (the usual case)
            HeapEnt par;
            par.key = 0; par.fval = 0; par.node = 0;
            bool have_par = false;
            // vmcnt counts stores and loads together, in order: a round's loads issued BEHIND the round before's stores wait for those stores
            // to be acknowledged as well.  The entry a lane moves up is therefore stored only after the next round's loads are on their way
            // (they read deeper levels than anything a round writes: nothing is read stale).
            bool pend = false;
            int64_t pend_P = 0;
            HeapEnt pend_ch;
            pend_ch.key = 0; pend_ch.fval = 0; pend_ch.node = 0;
            while (hole < half) {
                const int64_t P = ((hole + 1) << (t - 1)) - 1 + p;    // the node whose two children this lane holds
                const bool has = t <= levels && P < half;
                HeapEnt el, er;
                el.key = 0; el.fval = 0; el.node = 0; er = el;
                if (has) { el = get((uint64_t)(2 * P + 1)); er = get((uint64_t)(2 * P + 2)); }
                if (pend) { set((uint64_t)pend_P, pend_ch); pend = false; }
                const bool pick_left = ent_prio(er) < ent_prio(el);    // second = right child; if (right < left) second = left
                const uint64_t pl = Grp<G>::ballot(pick_left, gbase);
                uint64_t path = 0;
#pragma unroll
                for (int tt = 1; tt <= 3; ++tt) {
                    const bool on = has && t == tt &&
                                    (tt == 1 || (((path >> parent_lane) & 1ull) && (int)((pl >> parent_lane) & 1ull) == ((p & 1) ^ 1)));
                    path |= Grp<G>::ballot(on, gbase);
                }
                HeapEnt ch;                                                       // the child the parent picks (field by field: no address select)
                ch.key = pick_left ? el.key : er.key; ch.fval = pick_left ? el.fval : er.fval; ch.node = pick_left ? el.node : er.node;
                if ((path >> gl) & 1ull) { pend = true; pend_P = P; pend_ch = ch; }   // every node of the path moves up one level (stored behind the next round's loads)
                const int deepest = 63 - __builtin_clzll(path);                   // path != 0: the hole has two children
                hole = Grp<G>::bcast(pick_left ? 2 * P + 1 : 2 * P + 2, deepest, gbase);
                par.key = Grp<G>::bcast(ch.key, deepest, gbase); par.fval = Grp<G>::bcast(ch.fval, deepest, gbase); par.node = Grp<G>::bcast(ch.node, deepest, gbase);
                have_par = true;
                levels = 3;
            }
            const bool single = (len & 1) == 0 && hole == (len - 2) / 2;
            HeapEnt ce_early;
            ce_early.key = 0; ce_early.fval = 0; ce_early.node = 0;
            if (single) ce_early = get((uint64_t)(2 * hole + 1));
            if (pend) set((uint64_t)pend_P, pend_ch);
            if (single) {
                const HeapEnt ce = ce_early;
                if (gl == 0) set((uint64_t)hole, ce);
                hole = 2 * hole + 1;
                par = ce; have_par = true;
            }
            if (have_par && !(ent_prio(par) < ent_prio(v))) { if (gl == 0) set((uint64_t)hole, v); }   // __push_heap stops at once
            else sift_up((uint64_t)hole, v);
        }
    }
};
// start fetching a line the search is going to need a few microseconds from now (the popped node's hash slot, its pool entry, its
// graph line) while the heap is being repaired: by the time it is used it comes from L2 instead of HBM.  gfx950 has no prefetch
// instruction: a one-byte load whose result nobody reads.  The compiler does not know the asm is a load, so the register is kept
// reserved until touch_done(), which waits for the load (a result arriving later would overwrite whatever the register holds then).
__device__ __forceinline__ uint32_t touch(const void *p) {
    uint32_t t;
    asm volatile("global_load_ubyte %0, %1, off" : "=v"(t) : "v"(p) : "memory");
    return t;
}
__device__ __forceinline__ void touch_done(uint32_t a, uint32_t b, uint32_t c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" :: "v"(a), "v"(b), "v"(c));
}

// closed set + open_hash (the probes are in the kernel: hfirst / hfind)
__device__ __forceinline__ void hash_put(HashEnt *tab, uint32_t i, uint64_t key, uint32_t val, int32_t fval = 0) {
    *reinterpret_cast<uint4 *>(tab + i) = make_uint4((uint32_t)key, (uint32_t)(key >> 32), val, (uint32_t)fval);
}

// child descriptor cached for `key` and visible to seed `seed`, or -1
__device__ __forceinline__ int cache_lookup(const AstarArgs &a, int dir, uint64_t key, int64_t seed) {
    const CacheEnt *tab = dir ? a.cache[1] : a.cache[0];
    const uint64_t cmask = dir ? a.cache_mask[1] : a.cache_mask[0];
    uint64_t i = mix64(key) & cmask;
    for (uint32_t probes = 0; probes <= a.cache_probe_limit; ++probes) {
        const unsigned long long k = ld_agent(&tab[i].key);
        unsigned long long v = ld_agent(&tab[i].val);                  // (both in flight together: one round trip per probe)
        if (k == 0) return -1;
        if (k == key) {
            if (v == 0ull) return -1;
            v = ~v;
            return (int64_t)(v >> 16) <= seed ? (int)(v & 0xFFFF) : -1;
        }
        i = (i + 1) & cmask;
    }
    return -1;                                                         // (an insert never goes further than this either)
}
__device__ __forceinline__ void cache_insert(const AstarArgs &a, int dir, uint64_t key, int64_t visible_from, int em_state) {
    CacheEnt *tab = dir ? a.cache[1] : a.cache[0];
    const uint64_t cmask = dir ? a.cache_mask[1] : a.cache_mask[0];
    uint64_t i = mix64(key) & cmask;
    unsigned long long v = ((unsigned long long)visible_from << 16) | (unsigned long long)(em_state & 0xFFFF);
    for (uint32_t probes = 0; probes <= a.cache_probe_limit; ++probes) {
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
    __hip_atomic_fetch_add(&a.start_limit[13], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // no room: counted, reported by the host
}

// the cost term of the sharing rule: c expansions delay a path's visibility by c / rate seeds (rate > 0) or c * |rate| seeds (rate < 0)
__device__ __forceinline__ long long cost_term(const AstarArgs &a, unsigned long long c) {
    const int rate = a.cost_rate;
    if (rate > 0) {
        if (a.cost_knee != 0ull && c > a.cost_knee) return (long long)(a.cost_knee / (unsigned)rate + (c - a.cost_knee) / (unsigned)a.cost_rate2);
        return (long long)(c / (unsigned)rate);
    }
    return rate < 0 ? (long long)(c * (unsigned)(-rate)) : 0ll;
}
// Highest seed that may start now: no unfinished search can still become visible to it (see AstarArgs::window).  Every lane of
// the WAVE calls it and returns the same value; start_limit keeps the maximum ever computed (the bound only grows).  The table
// reads are atomics performed at the coherence point, four in flight per lane.
// The seed at position q of this direction's list of seeds to run (ascending seed indices; a launch that resumes a batch behind its
// commit frontier holds a sub-sequence of them): every seed below it is finished or in the table; past the end, past every seed.
__device__ __forceinline__ long long seed_at(const AstarArgs &a, int dir, unsigned long long q) {
    const int64_t n = dir ? a.n_todo[1] : a.n_todo[0];
    const int64_t *todo = dir ? a.todo[1] : a.todo[0];
    return q < (unsigned long long)n ? (long long)todo[q] : (long long)a.n_seeds;
}
template <int G>
__device__ __forceinline__ long long start_bound(const AstarArgs &a, int dir, int lane) {
    constexpr uint32_t SPB = kAstarWaves * Grp<G>::kGroups;           // search slots per workgroup
    const unsigned long long hq = ld_agent(&a.queue[dir]);            // the queue first: every seed below it is in the table by now
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long head = seed_at(a, dir, hq);
    long long bound = head + a.window - 1;
    const uint32_t s0 = a.blocks_dir0 * SPB;                          // this direction's slots: [0, s0) or [s0, n_slots)
    const uint32_t n_dir = dir ? a.n_slots - s0 : s0, first = dir ? s0 : 0u;
    for (uint32_t t0 = 0; t0 < n_dir; t0 += 256) {
        long long js[4];
        unsigned long long pr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t t = t0 + u * 64 + (uint32_t)lane;
            const uint32_t sl = first + t;
            js[u] = -1; pr[u] = 0;
            if (t < n_dir) {
                js[u] = __hip_atomic_fetch_add(&a.run_seed[sl], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.cost_rate != 0) pr[u] = __hip_atomic_fetch_add(&a.run_progress[sl], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (js[u] >= 0) {
                const long long b = js[u] + a.window - 1 + cost_term(a, pr[u]);
                bound = b < bound ? b : bound;
            }
    }
    bound = wave_min_ll(bound);
    if (lane == 0 && bound > 0) __hip_atomic_fetch_max(&a.start_limit[dir], (unsigned long long)bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return bound;
}

// start_limit[5]: the call for memory of a search nobody may overtake (the lowest running one when it starves, or one that waits to start
// again): first28 << 28 | last28, times in units of 1024 ticks of the 100 MHz clock (~10 us) of the first and the latest call of the
// episode (calls less than 50 ms apart).  The LEVEL of the call is its age in 10 ms steps: starved searches with fewer than
// 256 << 2 level expansions give their memory back, so the ones with the least work to lose go first and everybody after 100 ms.
// start_limit[11]: the lowest search (2 * seed + direction, + 1) that has called in this episode.  Any search that has waited for memory
// beyond its patience calls; a starved search ABOVE the caller gives its memory back (by the level rule above), the caller and
// everything below it keep theirs: the lowest starved search is always served, so a pool that the searches in flight have outgrown
// together drains from the top instead of standing still.
__device__ __forceinline__ void call_for_memory(const AstarArgs &a, long long sid) {
    const unsigned long long now = (__builtin_amdgcn_s_memrealtime() >> 10) & 0xFFFFFFFull, word = ld_agent(&a.start_limit[5]);
    const unsigned long long last = word & 0xFFFFFFFull, first = (word >> 28) & 0xFFFFFFFull;
    const bool going = word != 0ull && ((now - last) & 0xFFFFFFFull) < 5000ull;
    // the caller's seat ([11] id + 1, [12] when its holder last called): taken by a lower search, refreshed by its holder, and free again
    // 20 ms after the holder's last call (it was served, or it ended)
    const unsigned long long cur = ld_agent(&a.start_limit[11]), seen = ld_agent(&a.start_limit[12]);
    if (!going || cur == 0ull || ((now - seen) & 0xFFFFFFFull) > 2000ull || (unsigned long long)sid + 1ull <= cur) {
        st_agent(&a.start_limit[11], (unsigned long long)sid + 1ull);
        st_agent(&a.start_limit[12], now);
    }
    st_agent(&a.start_limit[5], ((going ? first : now) << 28) | now);
}
__device__ __forceinline__ long long memory_caller(const AstarArgs &a) { return (long long)ld_agent(&a.start_limit[11]) - 1ll; }
__device__ __forceinline__ int memory_call_level(const AstarArgs &a) {          // -1: nobody is calling
    const unsigned long long now = (__builtin_amdgcn_s_memrealtime() >> 10) & 0xFFFFFFFull, word = ld_agent(&a.start_limit[5]);
    const unsigned long long last = word & 0xFFFFFFFull, first = (word >> 28) & 0xFFFFFFFull;
    if (word == 0ull || ((now - last) & 0xFFFFFFFull) >= 5000ull) return -1;
    const unsigned long long age = ((last - first) & 0xFFFFFFFull) / 1000ull;
    return age > 12ull ? 12 : (int)age;
}

// Lowest search (2 * seed + direction) any slot of the launch is working on (-1: none).  Every lane of the WAVE calls it and returns the
// same value.  Under the gate seeds are taken in order, so this is the one search nobody is ahead of: the one that never yields its memory.
template <int G>
__device__ __forceinline__ long long lowest_running(const AstarArgs &a, int lane) {
    constexpr uint32_t SPB = kAstarWaves * Grp<G>::kGroups;
    long long lo = 0x7FFFFFFFFFFFFFFFll;
    for (uint32_t t0 = 0; t0 < a.n_slots; t0 += 256) {
        long long js[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t sl = t0 + u * 64 + (uint32_t)lane;
            js[u] = -1;
            if (sl < a.n_slots) js[u] = __hip_atomic_fetch_add(&a.run_seed[sl], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (js[u] >= 0) js[u] = js[u] * 2 + (long long)(sl >= a.blocks_dir0 * SPB ? 1u : 0u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (js[u] >= 0 && js[u] < lo) lo = js[u];
    }
    lo = wave_min_ll(lo);
    return lo == 0x7FFFFFFFFFFFFFFFll ? -1ll : lo;
}

__device__ __forceinline__ int base_of(char ch) {
    return (ch == 'A' || ch == 'a') ? 0 : (ch == 'C' || ch == 'c') ? 1 : (ch == 'G' || ch == 'g' || ch == 'N' || ch == 'n') ? 2
           : (ch == 'T' || ch == 't') ? 3 : -1;
}

#ifdef MGTA_ASTAR_PROFILE   // diagnostic build only: per-phase cycle sums (s_memtime)
// Every lane keeps its own sums (the macros run under the lane's own EXEC mask), flushed by the first lane of every GROUP: [3..8] = cycles a
// RUNNING search spends in the phases of an expansion, [10] = that search's expanding iterations (so [q] / [10] is "per expansion of one
// search"); [0] [1] [2] [9] (taking seeds, the gate, starting, results) and [11] = wave iterations with an expansion in them are flushed by
// lane 0 only (wave-level); [12] / [13] = the wave's lifetime in s_memrealtime (100 MHz) / s_memtime ticks: the clock the counts are in;
// [14] / [15] sleeps of a wave whose running searches all wait for memory.
#define PROF_DECL unsigned long long pt_ = __builtin_amdgcn_s_memtime(), pacc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    const unsigned long long prt0_ = __builtin_amdgcn_s_memrealtime(), pmt0_ = pt_; bool prun_ = false;
#ifdef MGTA_ASTAR_PROFILE_DRAIN   // every phase ends with its own memory operations done: the drain of a phase's stores is charged to that phase, not to the next wait
#define PROF(i) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); unsigned long long n_ = __builtin_amdgcn_s_memtime(); pacc_[i] += n_ - pt_; pt_ = n_; }
#else
#define PROF(i) { unsigned long long n_ = __builtin_amdgcn_s_memtime(); pacc_[i] += n_ - pt_; pt_ = n_; }
#endif
#define PROF_ITER(running) { prun_ = (running); if (prun_) pacc_[10] += 1; if (__ballot(prun_)) pacc_[11] += 1; }
// (a lane whose search did not expand in this iteration has not passed PROF(3..7): its clock is moved up instead of charging the others' expansion to its "run end")
#define PROF_IDLE_LANES { if (!prun_) pt_ = __builtin_amdgcn_s_memtime(); }
#define PROF_FLUSH { if (gl == 0) { for (int q_ = 3; q_ <= 8; ++q_) atomicAdd(&a.prof[q_], pacc_[q_]); atomicAdd(&a.prof[10], pacc_[10]); } \
    if (lane == 0) { atomicAdd(&a.prof[0], pacc_[0]); atomicAdd(&a.prof[1], pacc_[1]); atomicAdd(&a.prof[2], pacc_[2]); atomicAdd(&a.prof[9], pacc_[9]); \
    atomicAdd(&a.prof[11], pacc_[11]); atomicAdd(&a.prof[14], pacc_[14]); atomicAdd(&a.prof[15], pacc_[15]); \
    atomicAdd(&a.prof[12], __builtin_amdgcn_s_memrealtime() - prt0_); atomicAdd(&a.prof[13], __builtin_amdgcn_s_memtime() - pmt0_); } }
#else
#define PROF_DECL
#define PROF(i)
#define PROF_ITER(running)
#define PROF_IDLE_LANES
#define PROF_FLUSH
#endif

template <int G, bool LDS>
__global__ __launch_bounds__(kAstarThreads) void astar_kernel(AstarArgs a) {
    using GX = Grp<G>;
    constexpr int GROUPS = GX::kGroups;
    constexpr uint32_t SPB = kAstarWaves * GROUPS;
    extern __shared__ __align__(16) unsigned char s_mem[];            // [heap tops][page tables: nodes, heap, hash][HMM tables]
    HeapEnt *const s_heap = reinterpret_cast<HeapEnt *>(s_mem);
    constexpr uint32_t kLdsHeapSlots = GX::kLdsHeap;
    uint32_t *const s_seg = reinterpret_cast<uint32_t *>(s_mem + (size_t)SPB * kLdsHeapSlots * sizeof(HeapEnt));
    HeapEnt *const s_stage = reinterpret_cast<HeapEnt *>(s_mem + (size_t)SPB * (kLdsHeapSlots * sizeof(HeapEnt) + kPtWords * sizeof(uint32_t)));
    (void)s_stage;
    double *const s_tab = reinterpret_cast<double *>(s_mem + (size_t)SPB * (kLdsHeapSlots * sizeof(HeapEnt) + kPtWords * sizeof(uint32_t) + kStage * sizeof(HeapEnt)));

    const int dir = blockIdx.x < a.blocks_dir0 ? 0 : 1;
    if (threadIdx.x == 0) __hip_atomic_fetch_min(&a.tmark[0], (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    HmmView hv;                                                       // select by value: no indexed access into the kernel arguments
    hv.tab = dir ? a.hm[1].tab : a.hm[0].tab; hv.M = dir ? a.hm[1].M : a.hm[0].M; hv.A = dir ? a.hm[1].A : a.hm[0].A;
    hv.col_fwd = dir ? a.hm[1].col_fwd : a.hm[0].col_fwd; hv.col_enum = dir ? a.hm[1].col_enum : a.hm[0].col_enum;
    const int M = hv.M, A = hv.A;
    const double *tab = hv.tab;
    if (LDS) {
        const size_t nd = (size_t)(M + 1) * (A + 11);
        for (size_t i = threadIdx.x; i < nd; i += kAstarThreads) s_tab[i] = hv.tab[i];
        __syncthreads();
        tab = s_tab;
    }
    const size_t M1 = (size_t)M + 1;
    const double *msc = tab, *tsc = tab + M1 * A, *maxm = tsc + 7 * M1, *hc = maxm + M1;
    const int lane = lane_id(), wv = wave_id();
    const int gl = lane & (G - 1), gbase = lane & ~(G - 1), grp = lane / G;
    const uint32_t lslot = (uint32_t)wv * GROUPS + (uint32_t)grp;     // search slot inside the workgroup
    const uint32_t slot = blockIdx.x * SPB + lslot;
    const GraphDev &g = a.g;
    const bool forward = dir == 0;
    const double NEG_INF = -__builtin_inf();
    const uint64_t lt_mask = (1ull << gl) - 1ull;                     // lower lanes of the group
    const int64_t n_todo = dir ? a.n_todo[1] : a.n_todo[0];
    const int64_t *todo = dir ? a.todo[1] : a.todo[0];

    // this slot's base arena: B0 nodes (64 B), 2 B0 heap slots (16 B), 2 B0 hash entries (16 B); its page tables in LDS
    uint32_t *const pt_nodes = s_seg + (size_t)lslot * kPtWords, *const pt_heap = pt_nodes + kLdsPages, *const pt_hash = pt_heap + kLdsPages,
                    *const gt_words = pt_hash + kLdsPages;             // [0] nodes, [1] heap, [2] hash: unit of the array's table chunk (pages >= kLdsPages)
    const int log_b0 = a.log_b0;
    const uint32_t B0 = 1u << log_b0;
    char *const slot_base = a.pool.base + a.base_off + (uint64_t)slot * a.slot_bytes;
    auto base_hash_ptr = [&]() { return reinterpret_cast<HashEnt *>(slot_base + (96u << log_b0)); };   // (computed where it is used: a register less)
#define base_hash (base_hash_ptr())
    const uint32_t hmask0 = 2 * B0 - 1;                                // the base arena's table
    Heap<G> H;
    H.lds = s_heap + (size_t)lslot * kLdsHeapSlots;
    H.ar.slot_base = slot_base; H.ar.base_off = 64u << log_b0; H.ar.base_n = 2 * B0; H.ar.pt = pt_heap; H.ar.gt = gt_words + 1; H.ar.pool = a.pool.base;
    H.gl = gl; H.gbase = gbase;
    NodeArr AR;
    AR.slot_base = slot_base; AR.base_off = 0u; AR.base_n = B0; AR.pt = pt_nodes; AR.gt = gt_words; AR.pool = a.pool.base;
    auto node_at = [&](uint32_t i) { return reinterpret_cast<ANode *>(AR.at(i)); };

    // ---- per-search state (uniform inside a group)
    int st = S_IDLE;
    bool need_scan = false;
    uint32_t spins = 0;
    uint32_t last_lim = 0xFFFFFFFFu;                                  // (low word of) the start limit when this wave last looked (gate progress)
    long long seed = -1;
#define sid (seed * 2 + dir)                                          /* the search's id: computed, not kept (registers) */
    uint32_t n_nodes = 0, n_heap = 0, n_keys = 0;
    uint32_t np_nodes = 0, np_heap = 0;                               // pages of the node array / the heap: capacity = base arena + pages
#define cap_nodes (B0 + (np_nodes << NodeArr::kPerPageLog))
#define cap_heap (2 * B0 + (np_heap << HeapArr::kPerPageLog))
    // the hash table: the base arena's (hp_pages == 0: hmask0 + 1 entries, linear probing), or hp_pages = (1 << hL) + hp buckets of one
    // page each (linear hashing: a key's bucket is its hash modulo 2^hL, modulo 2^(hL+1) where that bucket has been split already;
    // linear probing inside the bucket, starting from other bits of the hash)
    uint32_t hp_pages = 0, hp = 0;
    int hL = 0;
    uint32_t n_closed = 0, n_expanded = 0, n_opened = 0;     // (a search of 2^32 expansions would run for a day)
    int status = 1, partial = 0, ok = 0;
    uint32_t starved = 0;                                             // iterations this search has waited for memory
    bool yield_check = false;                                         // ordered launch: starved for long -- give the memory back unless this is the lowest running seed
    bool lowest_check = false;                                        // ordered launch: waiting for memory -- is this the lowest running search (the reserve's owner)?
    bool use_reserve = false;                                         // this IS the lowest running search: what the pool cannot give it comes from the reserve
    bool order_off = a.free_share != 0;                               // paths are shared without an order (asked for, or the batch gave its order up: start_limit[14])
    uint32_t prog_floor = 0;                                          // expansions already announced for this seed before it started again in place
    bool have_curr = false;                                           // the node to expand is already popped (the search was waiting for memory)
    int32_t goal = -1, inter = 0, cur = 0;
    double inter_val = 0;                                             // (real_score + exit_prob[length]) / ln 2 of node `inter`
    bool first = true;
    ANode curr;
    curr.score = curr.real_score = curr.max_score = 0; curr.node_id = 0; curr.parent = -1; curr.fval = 0;
    curr.state_no = curr.length = curr.negative_count = 0; curr.em_state = 0; curr.fwd_r = 0; curr.fwd_hint = kFdNone; curr.pad = 0;

    auto hpage = [&](uint32_t b) -> HashEnt * {
        const uint32_t u = b < (uint32_t)kLdsPages ? pt_hash[b] : reinterpret_cast<const uint32_t *>(a.pool.base + ((uint64_t)(gt_words[2] & kUnitMask) << kUnitLog))[b];
        return reinterpret_cast<HashEnt *>(a.pool.base + ((uint64_t)(u & kUnitMask) << kUnitLog));
    };
    // where the probe sequence of `key` starts: table / bucket, first index; the sequence wraps inside `pmask`
    auto hfirst = [&](uint64_t key, uint32_t &s) -> HashEnt * {
        const uint64_t h = mix64(key);
        if (hp_pages == 0u) { s = (uint32_t)h & hmask0; return base_hash; }
        uint32_t b = (uint32_t)h & ((1u << hL) - 1u);
        if (b < hp) b = (uint32_t)h & ((2u << hL) - 1u);
        s = (uint32_t)(h >> 32) & (kHashPerPage - 1u);
        return hpage(b);
    };
    auto hfind = [&](uint64_t key, bool &found, uint32_t &val) -> HashEnt * {     // every lane of the group probes the same key (one request)
        uint32_t i;
        HashEnt *const tab = hfirst(key, i);
        const uint32_t pmask = hp_pages ? kHashPerPage - 1u : hmask0;
        while (true) {
            const uint4 v = *reinterpret_cast<const uint4 *>(tab + i);
            const uint64_t k = (uint64_t)v.x | ((uint64_t)v.y << 32);
            if (k == 0) { found = false; val = kNone; return tab + i; }
            if (k == key) { found = true; val = v.z; return tab + i; }
            i = (i + 1) & pmask;
        }
    };
    // lane 0: page g of an array (named in LDS below kLdsPages, else in the array's table chunk, obtained with the first such page)
    auto in_pool = [&](uint32_t u) { return a.pool.reserve_bytes == 0ull || ((unsigned long long)(u & kUnitMask) << kUnitLog) < a.pool.reserve_off; };
    // lane 0: page g of an array (named in LDS below kLdsPages, else in the array's table chunk, obtained with the first such page).
    // 0 = no memory now, 1 = a page of the reserve, 2 = a page of the pool, -1 = the array is at its limit (kMaxPages pages: no memory ever helps)
    auto take_page = [&](uint32_t *pt, uint32_t *gtw, uint32_t g) -> int {
        if (g >= (uint32_t)kMaxPages) return -1;
        if (g == (uint32_t)kLdsPages) {                                // the table chunk first
            const uint32_t t = chunk_alloc(a.pool, 0, use_reserve);
            if (t == kNoChunk) return 0;
            *gtw = t;
        }
        const uint32_t u = chunk_alloc(a.pool, kPageClass, use_reserve);
        if (u == kNoChunk) {
            if (g == (uint32_t)kLdsPages) { pool_free(a.pool, 0, *gtw); }
            return 0;
        }
        if (g < (uint32_t)kLdsPages) pt[g] = u;
        else reinterpret_cast<uint32_t *>(a.pool.base + ((uint64_t)(*gtw & kUnitMask) << kUnitLog))[g] = u;
        return in_pool(u) ? 2 : 1;
    };
    // (every lane) after lane 0 has written a page table: LDS words are fenced, a table chunk's line may sit in this CU's L1 from an earlier
    // look-up of a neighbouring entry
    auto tables_written = [&](uint32_t g) {
        wave_lds_fence();
        if (g >= (uint32_t)kLdsPages) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
    };
    auto free_pages = [&](const uint32_t *pt, const uint32_t *gtw, uint32_t n) {   // lane 0, after pool_release_fence()
        for (uint32_t g = 0; g < n; ++g)
            pool_free(a.pool, kPageClass, g < (uint32_t)kLdsPages ? pt[g] : reinterpret_cast<const uint32_t *>(a.pool.base + ((uint64_t)(*gtw & kUnitMask) << kUnitLog))[g]);
        if (n > (uint32_t)kLdsPages) pool_free(a.pool, 0, *gtw);
    };
    // everything beyond the base arena goes back to the pool (every lane of the group calls; lane 0 does it)
    auto release_all = [&]() {
        if (np_nodes | np_heap | hp_pages) {
            pool_release_fence();
            if (gl == 0) { free_pages(pt_nodes, gt_words, np_nodes); free_pages(pt_heap, gt_words + 1, np_heap); free_pages(pt_hash, gt_words + 2, hp_pages); }
        }
        np_nodes = 0; np_heap = 0; hp_pages = 0; hp = 0; hL = 0;
        if (gl == 0) gt_words[3] = kNoChunk;                             // (the spare page of the reserve's owner goes with the reserve)
    };

    PROF_DECL
    while (true) {
        // ================= next search for the idle slots
        if (st == S_IDLE && lslot >= a.active_slots) st = S_EXIT;
        if (st == S_BACKOFF) {    // a search that gave its memory back: it starts again (same seed, same place in the order) once there is room
            unsigned long long used = 0, quit = 0;
            if (gl == 0) { used = ld_agent(&a.pool.stat[4]); quit = ld_agent(&a.start_limit[4]); }
            used = GX::bcast(used, 0, gbase);
            quit = GX::bcast(quit, 0, gbase);
            if (quit) { if (gl == 0) st_agent(&a.run_seed[slot], -1ll); st = S_EXIT; }       // (status stays 0: the pass is run again)
            else if (used <= a.pool.soft_limit) { st = S_START; starved = 0; }
            else if ((++starved & 255u) == 0u && gl == 0) call_for_memory(a, sid);   // waiting for room is calling for room too (this may be the
                                                                                      // lowest search of all: the running ones must not sit on what it waits for)
        }
        if (__ballot(st == S_BACKOFF) != 0ull && __ballot(st == S_START || st == S_RUN || st == S_DONE) == 0ull) {
#pragma unroll
            for (int z = 0; z < 4; ++z) __builtin_amdgcn_s_sleep(127);
        }
        bool admit = true;
        unsigned long long used = 0;
        if (st == S_IDLE) {       // admission: searches in flight are bounded by the memory they hold, not only by the number of slots
            if (gl == 0) {
                used = ld_agent(&a.pool.stat[4]);
                if (used > ld_agent(&a.pool.stat[5])) __hip_atomic_fetch_max(&a.pool.stat[5], used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            used = GX::bcast(used, 0, gbase);
            admit = used <= a.pool.soft_limit;
        }
        if (st == S_IDLE && admit) {
            long long qi = 0;
            if (gl == 0) {
                if (a.gate && ld_agent(&a.start_limit[4]) != 0ull) {
                    qi = n_todo;                                                       // the pass has given up (a search that fits nowhere): take nothing more
                } else if (a.gate) {
                    // announce a lower bound of the seed about to be taken BEFORE taking it: whoever sees the queue beyond a seed also
                    // sees a slot that holds it (or its committed paths)
                    st_agent(&a.run_progress[slot], 0ull);
                    unsigned long long t = ld_agent(&a.queue[dir]);
                    st_agent(&a.run_seed[slot], seed_at(a, dir, t));
                    // SLOW START: the searches of a batch's first seconds find the caches empty and are all long ones (50 M reads, nirK:
                    // 8192 of them outgrew a 140 GB pool together within ten seconds and the batch stood still).  The searches in
                    // flight per direction start at ramp_base and grow by one with every search that ends, so the memory in use is
                    // known (and the admission rule above works) before every slot is busy.
                    const unsigned long long done = ld_agent(&a.start_limit[6 + dir]);
                    qi = -1;
                    for (int tries = 0; tries < 4; ++tries) {
                        if (t >= (unsigned long long)n_todo) { qi = (long long)t; break; }   // nothing left to take: the slot retires
                        // (only while the memory in use says the searches are long ones: a batch of short searches -- 2 M reads: 19.2 s
                        // instead of 15.5 s behind the ramp -- fills its slots at once)
                        if (t - done >= (unsigned long long)a.ramp_base + done && used * 16ull >= a.pool.soft_limit) break;
                        if (__hip_atomic_compare_exchange_strong(&a.queue[dir], &t, t + 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { qi = (long long)t; break; }
                    }
                    if (qi < 0) st_agent(&a.run_seed[slot], -1ll);                       // not now: the slot stays idle
                } else {
                    qi = (long long)atomicAdd(&a.queue[dir], 1ull);
                }
            }
            qi = GX::bcast(qi, 0, gbase);
            if (qi < 0) {
                // (slow start: no seed taken this time)
            } else if (qi >= n_todo) {
                st = S_EXIT;
                if (gl == 0) __hip_atomic_fetch_min(&a.tmark[1 + dir], (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.gate && gl == 0) st_agent(&a.run_seed[slot], -1ll);
            } else {
                seed = todo[qi];
                prog_floor = 0;
                if (a.gate) {
                    if (gl == 0) st_agent(&a.run_seed[slot], (long long)seed);
                    st = S_WAIT; need_scan = true; spins = 0;
                } else {
                    st = S_START;
                }
            }
        }
        if (__ballot(st != S_EXIT) == 0ull) break;
        if (__ballot(st != S_EXIT && st != S_IDLE) == 0ull) {               // every slot of this wave waits for memory
#pragma unroll
            for (int z = 0; z < 8; ++z) __builtin_amdgcn_s_sleep(127);
        }
        PROF(0)

        // ================= ordered-commit gate (shared-cache launches): wave-level, never blocks the searches that are running
        // Seed i may start once no unfinished search j can still become visible to it: i < j + B + progress_j / cost_rate for every
        // running j, and i < q + B for the next seed q of the queue.  The lowest running search always passes.  The limit moves when a
        // search ends (its wave recomputes it) and, with a cost term, as the running searches progress: for that ONE waiting wave per
        // direction and ~50 us re-reads the table (ticket = time of the last refresh); the others poll one word.
        if (a.gate) {
            if (__ballot(st == S_WAIT) != 0ull) {
                bool scan = __ballot(st == S_WAIT && need_scan) != 0ull;
                if (!scan) {
                    long long lim = 0, quit = 0;
                    if (lane == 0) {
                        lim = (long long)ld_agent(&a.start_limit[dir]); quit = (long long)ld_agent(&a.start_limit[4]);
                        if (a.auto_unorder && ld_agent(&a.start_limit[14]) != 0ull) lim = 0x7FFFFFFFFFFFFFFFll;   // the order is off: nobody waits
                    }
                    lim = __shfl(lim, 0, 64);
                    quit = __shfl(quit, 0, 64);
                    if (st == S_WAIT && quit) { if (gl == 0) st_agent(&a.run_seed[slot], -1ll); st = S_EXIT; }   // the pass has given up and is run again
                    if ((uint32_t)lim != last_lim) { last_lim = (uint32_t)lim; spins = 0; }     // the searches ahead are getting on: a long queue is not a hang
                    if (st == S_WAIT && lim >= seed) st = S_START;
                    if (__ballot(st == S_WAIT) != 0ull) {
                        int refresh = 0;
                        if (lane == 0) {
                            const unsigned long long now = __builtin_amdgcn_s_memrealtime();     // 100 MHz
                            unsigned long long last = ld_agent(&a.start_limit[2 + dir]);
                            if (now - last > 5000ull)
                                refresh = __hip_atomic_compare_exchange_strong(&a.start_limit[2 + dir], &last, now, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                               __HIP_MEMORY_SCOPE_AGENT);
                        }
                        scan = __shfl(refresh, 0, 64) != 0;
                    }
                }
                if (scan) {
                    const long long b = start_bound<G>(a, dir, lane);
                    if (st == S_WAIT && b >= seed) st = S_START;
                    need_scan = false;
                }
                if (st == S_WAIT && ++spins > (1u << 22)) {          // bounded wait (counted while the limit stands still): the host reports the seed
                    if (gl == 0) { a.status[sid] = 4; st_agent(&a.run_seed[slot], -1ll); }
                    st = S_EXIT;
                }
                if (__ballot(st == S_START || st == S_RUN) == 0ull) {  // nothing to do in this wave but wait
#pragma unroll
                    for (int z = 0; z < 8; ++z) __builtin_amdgcn_s_sleep(127);
                }
            }
        }

        // ================= ordered launches never hand a starved search to the host (a re-run after the others would see other paths than a
        // run in its place): a search that has waited long for memory gives back what it holds and starts again IN PLACE -- same seed,
        // its slot keeps announcing it, so every later seed waits for it exactly as before and nobody ever saw anything of it -- unless it
        // is the lowest running seed: that one keeps what it has and is served by the others' memory.  Wave-level scan, rare.
        if (a.gate && __ballot(st == S_RUN && (yield_check || lowest_check)) != 0ull) {
            const long long lo = lowest_running<G>(a, lane);
            if (st == S_RUN && lowest_check) {
                lowest_check = false;
                if (sid == lo && !use_reserve) {                      // asks again at once, now with the reserve behind it
                    int got = 0;
                    if (gl == 0) got = reserve_claim(a.pool, sid) ? 1 : 0;
                    if (GX::bcast(got, 0, gbase)) { use_reserve = true; starved = 0; yield_check = false; }
                }
            }
            if (st == S_RUN && yield_check) {
                yield_check = false;
                // Only the lowest RUNNING search calls (when its reserve is used up as well), and only its call makes the others give
                // their memory back.  A search that is merely starved WAITS with what it holds: what it has computed is kept, memory comes
                // back as others end, and the lowest running search -- served by the reserve -- always ends.  (Letting every starved
                // search call was measured on nirK at 50 M reads: 205 000 searches started again in place within six minutes, each one
                // throwing away hundreds of thousands of expansions.)
                int level = -1;
                if (gl == 0) { if (sid == lo) call_for_memory(a, sid); else level = memory_call_level(a); }
                level = GX::bcast(level, 0, gbase);
                if (sid == lo) {
                    // nobody is ahead of this search.  When everything the pool has handed out is its own, waiting cannot help: the pass
                    // gives up and the host starts the batch again with more room (or reports that one search does not fit the device)
                    // (an upper bound of what it holds of the pool proper -- the reserve's pages are not in `used` --: a search that holds
                    // everything that is handed out cannot be helped by waiting)
                    const unsigned long long own = (unsigned long long)(np_nodes + np_heap + hp_pages) << kPageLog;
                    unsigned long long used = 0;
                    if (gl == 0) used = ld_agent(&a.pool.stat[4]);
                    used = GX::bcast(used, 0, gbase);
                    int got = 0;
                    if (!use_reserve && a.pool.reserve_bytes != 0ull && gl == 0) got = reserve_claim(a.pool, sid) ? 1 : 0;
                    if (GX::bcast(got, 0, gbase)) { use_reserve = true; starved = 0; }
                    else if (used <= own || starved > (1u << 18)) {                  // (the second: a backstop -- no room for ten seconds of calling)
                        status = 2; st = S_DONE;
                        if (gl == 0) st_agent(&a.start_limit[4], 1ull);                // no further seed is taken: the pass is going to be run again
                    }
                } else if (level >= 0 && (uint64_t)n_expanded < (256ull << (2 * level))) {
                    release_all();
                    if (use_reserve) {   // (it took the reserve as the lowest running search and a lower one has started since)
                        if (gl == 0) reserve_release(a.pool);
                        use_reserve = false;
                    }
                    if (gl == 0) __hip_atomic_fetch_add(&a.pool.stat[6], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    prog_floor = n_expanded > prog_floor ? n_expanded : prog_floor;
                    starved = 0; have_curr = false;
                    st = S_BACKOFF;
                }
            }
        }

        PROF(1)
        // ================= start node (hmm_graph_search.h:132-189)
        if (st == S_START) {
            n_nodes = 0; n_heap = 0; n_keys = 0;
            np_nodes = 0; np_heap = 0; hp_pages = 0; hp = 0; hL = 0;
            n_closed = 0; n_expanded = 0; n_opened = 0;
            status = 1; partial = 0; ok = 0; goal = -1; inter = 0; cur = 0; first = true; starved = 0; have_curr = false;
            use_reserve = false; lowest_check = false; yield_check = false;
            if (gl == 0) gt_words[3] = kNoChunk;
            if (a.auto_unorder && !order_off) {
                int off = 0;
                if (gl == 0) off = ld_agent(&a.start_limit[14]) != 0ull;
                order_off = GX::bcast(off, 0, gbase) != 0;
            }
            for (uint32_t i = (uint32_t)gl; i <= hmask0; i += G) hash_put(base_hash, i, 0ull, 0u);
            const char *km = a.kmers + seed * a.klen;
            const int n_aa = a.klen / 3;
            const int sstate = forward ? a.start_state[seed] : (M - a.start_state[seed] - n_aa);   // :73
            bool bad = sstate < 0 || sstate + n_aa > M || a.klen > kMaxKmer;
            double sc = 0, rs = 0;
            if (!bad) {
                for (int i = 1; i <= n_aa; ++i) {                      // scoreStart / realScoreStart, :112-130
                    const int ci = forward ? (i - 1) : (n_aa - i);     // the reverse search scores the reversed protein
                    int c = 0;
                    for (int t = 0; t < 3; ++t) {
                        const int b = base_of(km[3 * ci + t]);
                        if (b < 0) bad = true;
                        c = c * 4 + (b < 0 ? 0 : b);
                    }
                    const int col = hv.col_fwd[c];
                    if (col < 0) { bad = true; break; }
                    const double m = msc[(size_t)(sstate + i) * A + col], t = tsc[(size_t)T_MM * M1 + sstate + i - 1];
                    sc += m + t - maxm[sstate + i];
                    rs += m + t;
                }
            }
            st = S_RUN;
            if (bad) { status = 3; st = S_DONE; }
            else {
                curr.parent = -1; curr.state_no = (int16_t)(sstate + n_aa); curr.em_state = (uint16_t)(ST_M << 9); curr.length = (int16_t)n_aa;
                curr.fval = 0; curr.score = sc; curr.real_score = rs; curr.max_score = 0; curr.negative_count = 0;
                curr.node_id = a.start_node[sid];
                curr.fwd_r = 0; curr.fwd_hint = kFdNone; curr.pad = 0;                  // the start edge's own line is read once, by the first expansion
                if (gl == 0) store_node(node_at(0), curr);
                n_nodes = 1;
                inter_val = (curr.real_score + a.exit_prob[curr.length]) / a.log2v;
                if (curr.state_no >= M) { ok = 1; goal = 0; st = S_DONE; }             // :193-197
                else if (curr.node_id == -1) { ok = 0; n_opened = 1; st = S_DONE; }    // no children -> open.empty() -> false (:235-237)
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }

        PROF(2)
        PROF_ITER(st == S_RUN && starved == 0u)
        // ================= one expansion
        if (st == S_RUN) {
            bool stop = false;
            if (!first && !have_curr) {
                // pop until a node that is not closed (hmm_graph_search.h:243-257)
                bool have = false;
                HashEnt *hs = nullptr;
                uint32_t hval = kNone;
                uint64_t hkey = 0;
                bool hfound = false;
                while (n_heap > 0) {
                    const HeapEnt top = H.get(0);
                    // (the node's edge needs no line of its own any more: the node carries where its Forward lands)
                    uint32_t ts;
                    HashEnt *const tb = hfirst(top.key, ts);
                    // The children of this expansion are STORED into node lines no access has touched yet: a store of part of a line that is not
                    // in the L2 is acknowledged only once the line has been filled from memory (scripts/probe_table_sizes.py: a dependent fetch
                    // behind such a store takes 1.3 us instead of 0.66), and vmcnt makes the commit's first fetch wait for it.  The next four node
                    // lines (eight nodes) are therefore asked for now, under the heap repair: by the time the children are written they are there.
                    uint32_t tn = 0;
                    const bool tn_on = gl < 4 && n_nodes + 2u * (uint32_t)gl < cap_nodes;
                    if (tn_on) tn = touch(node_at(n_nodes + 2u * (uint32_t)gl));
                    // the popped node and the first entry of its key's probe sequence are FETCHED (not only touched) while the heap is repaired:
                    // they arrive during the repair's own round trips, and the two dependent L2 reads that used to follow it are gone.  The node
                    // lands in `curr` itself (dead between two expansions: no register is added); a closed top is popped over as before.
                    uint4 hv = *reinterpret_cast<const uint4 *>(tb + ts);
                    curr = load_node(node_at(top.node));
                    H.remove_top(n_heap);
                    --n_heap;
                    touch_done(tn, tn, tn);
                    bool found = false;
                    {
                        const uint32_t pm_ = hp_pages ? kHashPerPage - 1u : hmask0;
                        uint32_t i = ts;
                        while (true) {
                            const uint64_t k = (uint64_t)hv.x | ((uint64_t)hv.y << 32);
                            if (k == 0) { found = false; hval = kNone; break; }
                            if (k == top.key) { found = true; hval = hv.z; break; }
                            i = (i + 1) & pm_;
                            hv = *reinterpret_cast<const uint4 *>(tb + i);
                        }
                        hs = tb + i;
                    }
                    if (found && (hval >> 31)) continue;                               // closed
                    cur = (int32_t)top.node;
                    hkey = top.key;
                    if (!found) ++n_keys;                                              // (children of the first expansion are not in open_hash)
                    hfound = found;
                    have = true;
                    break;
                }
                if (!have) { partial = 1; ok = 1; goal = inter; stop = true; }        // open list ran dry (:339-341)
                else {
                    const double cv = (curr.real_score + a.exit_prob[curr.length]) / a.log2v;
                    const bool better = cv > inter_val;
                    if (curr.state_no >= M) {                                          // goal (:259-270)
                        if (better) inter = cur;
                        ok = 1; goal = inter; stop = true;
                    } else {
                        if (gl == 0) {                                                 // closed.insert (:272); the entry keeps its node and fval
                            if (hfound) hs->val = hval | 0x80000000u;
                            else hash_put(hs, 0u, hkey, hval | 0x80000000u);
                        }
                        n_closed++;
                        if (better) { inter = cur; inter_val = cv; }                   // :274-277
                    }
                }
            }
            PROF(3)
            // room for this expansion's children: the arena grows in place, the table is re-hashed when half full.  When the pool has
            // nothing to give, the search keeps its popped node and asks again in the next iteration: memory comes back as other
            // searches end (bounded: after kStarveLimit iterations it gives up with status 2 and is run again by the host)
            bool wait_mem = false;
            // (a search that is waiting asks again every 8th, later every 64th iteration: thousands of waiting searches asking every
            // iteration keep the allocator's words -- and the memory channels they live in -- busy for everybody)
            const bool ask = starved == 0u || (starved & (starved < 1024u ? 7u : 63u)) == 0u;
            if (!stop && !ask) wait_mem = true;
            if (!stop && !wait_mem && n_nodes + kMaxNew > cap_nodes) {                 // one more page of nodes
                if ((np_nodes | np_heap | hp_pages) == 0u && gl == 0) __hip_atomic_fetch_add(&a.pool.stat[3], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int got = 0;
                if (gl == 0) got = take_page(pt_nodes, gt_words, np_nodes);
                got = GX::bcast(got, 0, gbase);
                tables_written(np_nodes);
                if (got > 0) ++np_nodes;
                else if (got < 0) { status = 5; stop = true; }                         // (its own status: not a starvation, nothing to wait for or to resume)
                else wait_mem = true;
            }
            if (!stop && !wait_mem && heap_slots_needed(n_heap + kMaxNew) > cap_heap) {    // pages of heap slots (a new block level can ask for several)
                const uint32_t need = heap_slots_needed(n_heap + kMaxNew);
                if ((np_nodes | np_heap | hp_pages) == 0u && gl == 0) __hip_atomic_fetch_add(&a.pool.stat[3], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (need > cap_heap) {                                              // (what was obtained is kept: the next attempt goes on from there)
                    int got = 0;
                    if (gl == 0) got = take_page(pt_heap, gt_words + 1, np_heap);
                    got = GX::bcast(got, 0, gbase);
                    tables_written(np_heap);
                    if (got < 0) { status = 5; stop = true; break; }
                    if (!got) { wait_mem = true; break; }
                    ++np_heap;
                }
            }
            // the hash table: out of the base arena into one bucket when that is half full; then a third full on average (the buckets not yet
            // split in the current round hold up to twice the average): one bucket is split per step
            while (!stop && !wait_mem && (hp_pages == 0u ? (uint64_t)(n_keys + kMaxNew) * 2 > (uint64_t)hmask0 + 1
                                                         : (uint64_t)(n_keys + kMaxNew) * 3 > (uint64_t)hp_pages * kHashPerPage)) {
                if ((np_nodes | np_heap | hp_pages) == 0u && gl == 0) __hip_atomic_fetch_add(&a.pool.stat[3], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool first_bucket = hp_pages == 0u;
                const uint32_t b_old = hp, b_new = hp_pages;                           // the bucket that is split, the bucket it splits into
                if (b_new >= (uint32_t)kMaxPages) { status = 5; stop = true; break; }  // the table is at its limit
                // lane 0: the new page(s).  uA takes the place of the split bucket (or is the first bucket), uB is the new bucket
                uint32_t uA = kNoChunk, uB = kNoChunk, tnew = kNoChunk;
                if (gl == 0 && b_new < (uint32_t)kMaxPages) {
                    bool ok_t = true;
                    if (b_new == (uint32_t)kLdsPages) { tnew = chunk_alloc(a.pool, 0, use_reserve); ok_t = tnew != kNoChunk; }
                    if (ok_t) {
                        // (the reserve's owner: the page its last split emptied serves this one -- pages of the reserve never go back to a list, so
                        // without this slot every split of the owner used up two new pages for one more bucket; advisor r4)
                        if (gt_words[3] != kNoChunk) { uA = gt_words[3]; gt_words[3] = kNoChunk; }
                        else uA = chunk_alloc(a.pool, kPageClass, use_reserve);
                        if (uA != kNoChunk && !first_bucket) {
                            uB = chunk_alloc(a.pool, kPageClass, use_reserve);
                            if (uB == kNoChunk) {
                                if (!in_pool(uA)) gt_words[3] = uA; else pool_free(a.pool, kPageClass, uA);
                                uA = kNoChunk;
                            }
                        }
                        if (uA == kNoChunk && tnew != kNoChunk) { pool_free(a.pool, 0, tnew); tnew = kNoChunk; }
                    }
                }
                uA = GX::bcast(uA, 0, gbase); uB = GX::bcast(uB, 0, gbase);
                if (uA == kNoChunk) { wait_mem = true; break; }
                HashEnt *const pa = reinterpret_cast<HashEnt *>(a.pool.base + ((uint64_t)(uA & kUnitMask) << kUnitLog));
                HashEnt *const pb = first_bucket ? pa : reinterpret_cast<HashEnt *>(a.pool.base + ((uint64_t)(uB & kUnitMask) << kUnitLog));
                for (uint32_t i = (uint32_t)gl; i < kHashPerPage; i += G) { hash_put(pa, i, 0ull, 0u); if (!first_bucket) hash_put(pb, i, 0ull, 0u); }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const HashEnt *const src = first_bucket ? base_hash : hpage(b_old);
                const uint32_t n_src = first_bucket ? hmask0 + 1u : kHashPerPage;
                for (uint32_t i0 = 0; i0 < n_src; i0 += G) {                           // every lane moves one entry; slots are claimed at the L2
                    const uint32_t i = i0 + (uint32_t)gl;
                    if (i < n_src) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(src + i);
                        const unsigned long long k = (unsigned long long)v.x | ((unsigned long long)v.y << 32);
                        if (k != 0ull) {
                            const uint64_t h = mix64(k);
                            HashEnt *const dst = (!first_bucket && (((uint32_t)h >> hL) & 1u)) ? pb : pa;
                            uint32_t j = (uint32_t)(h >> 32) & (kHashPerPage - 1u);
                            while (true) {
                                unsigned long long expect = 0ull;
                                if (__hip_atomic_compare_exchange_strong(reinterpret_cast<unsigned long long *>(&dst[j].key), &expect, k, __ATOMIC_RELAXED,
                                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                                    __hip_atomic_store(&dst[j].val, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    __hip_atomic_store(&dst[j].fval, (int32_t)v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    break;
                                }
                                j = (j + 1) & (kHashPerPage - 1u);
                            }
                        }
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                     // the new pages were filled behind this CU's L1
                // the tables: bucket b_old -> uA, bucket b_new -> uB; the old page goes back
                if (!first_bucket) pool_release_fence();
                if (gl == 0) {
                    auto put = [&](uint32_t b, uint32_t u) {
                        if (b < (uint32_t)kLdsPages) pt_hash[b] = u;
                        else reinterpret_cast<uint32_t *>(a.pool.base + ((uint64_t)(gt_words[2] & kUnitMask) << kUnitLog))[b] = u;
                    };
                    if (tnew != kNoChunk) gt_words[2] = tnew;
                    if (first_bucket) put(0u, uA);
                    else {
                        const uint32_t u_old = b_old < (uint32_t)kLdsPages ? pt_hash[b_old]
                                             : reinterpret_cast<const uint32_t *>(a.pool.base + ((uint64_t)(gt_words[2] & kUnitMask) << kUnitLog))[b_old];
                        put(b_old, uA); put(b_new, uB);
                        if (!in_pool(u_old) && gt_words[3] == kNoChunk) gt_words[3] = u_old;   // (a page of the reserve: kept for the owner's next split)
                        else pool_free(a.pool, kPageClass, u_old);
                    }
                    __hip_atomic_fetch_add(&a.pool.stat[2], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                tables_written(b_new);
                if (first_bucket) { hp_pages = 1; hL = 0; hp = 0; }
                else { ++hp_pages; ++hp; if (hp == (1u << hL)) { ++hL; hp = 0; } }
            }

            have_curr = false;
            if (wait_mem) {
                have_curr = !first;
                // the lowest running search does not wait: the reserve is there for it (a scan of the slot table: asked early, then rarely)
                if (a.gate && !use_reserve && a.pool.reserve_bytes != 0ull && (starved & 1023u) == 8u) lowest_check = true;
                if (a.auto_unorder && !order_off && (starved & 63u) == 1u) {
                    int off = 0;
                    if (gl == 0) {
                        off = ld_agent(&a.start_limit[14]) != 0ull;
                        if (!off && ld_agent(&a.pool.stat[1]) > 4096ull) { st_agent(&a.start_limit[14], 1ull); off = 1; }
                    }
                    order_off = GX::bcast(off, 0, gbase) != 0;
                }
                // when every search in flight waits, nobody ends and nothing comes back: a search gives up after a wait in proportion
                // to the work it would lose (a quarter of its expansions so far in iterations, 2^8 .. 2^17), so the young ones free
                // their memory for the others within milliseconds and the ones that have run for seconds wait for seconds
                uint32_t patience = n_expanded >> 2;
                patience = patience < 256u ? 256u : patience > (kStarveLimit << 2) ? (kStarveLimit << 2) : patience;
                if (++starved > patience) {
                    if (a.gate) {                                                      // ordered launch: yield in place (above), never a host re-run
                        if (((starved - patience) & 255u) == 1u) yield_check = true;
                    } else { status = 2; stop = true; }
                }
            } else {
                starved = 0;
            }
            PROF(4)
            if (!stop && !wait_mem) {
                const int cst = curr.em_state >> 9;
                const int next_state = curr.state_no + 1;
                // term_nodes.find(curr) (hmm_graph_search.h:212,279): child recorded by an earlier seed, or -1
                int cached = -1;
                if (a.window > 0) {
                    if (gl == 0) cached = cache_lookup(a, dir, make_key(curr.node_id, curr.state_no, cst), order_off ? 0x7FFFFFFFFFFFll : (long long)seed);
                    cached = GX::bcast(cached, 0, gbase);
                }
                const int cached_st = cached >= 0 ? (cached >> 9) : -1;

                n_expanded++;
                if (a.gate && a.cost_rate != 0 && (n_expanded & 63) == 0 && gl == 0 && n_expanded > prog_floor)
                    __hip_atomic_store(&a.run_progress[slot], (unsigned long long)n_expanded, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.auto_unorder && !order_off && (n_expanded & 255u) == 0u) {
                    int off = 0;
                    if (gl == 0) off = ld_agent(&a.start_limit[14]) != 0ull;
                    order_off = GX::bcast(off, 0, gbase) != 0;
                }

                // ---- children (node_enumerator.h:131-244)
                double mt, it, dt;
                if (cst == ST_M) { mt = tsc[T_MM * M1 + curr.state_no]; it = tsc[T_MI * M1 + curr.state_no]; dt = tsc[T_MD * M1 + curr.state_no]; }
                else if (cst == ST_D) { mt = tsc[T_DM * M1 + curr.state_no]; it = NEG_INF; dt = tsc[T_DD * M1 + curr.state_no]; }
                else { mt = tsc[T_IM * M1 + curr.state_no]; it = tsc[T_II * M1 + curr.state_no]; dt = NEG_INF; }
                const double max_match = maxm[next_state];
                const double h_m = hc[next_state], h_i = hc[M1 + curr.state_no], h_d = hc[2 * M1 + next_state];
                // a cached match/insert child ends the enumeration at that child (:178-181,207-210)
                const bool ins_ok = cst != ST_D && cached_st != ST_M;

                // ---- admission (hmm_graph_search.h:288-311): prune test + open_hash lookup, all children in parallel
                auto admissible = [&](int length, int negative_count, double real_score) {
                    return a.prune > 0 ? ((length < 5 || negative_count <= a.prune) && real_score > 0.0) : true;
                };
                // the probes of one expansion are independent: their first loads are issued together, then resolved, then the fvals of
                // the open-list entries they found are fetched together (three dependent round trips instead of six)
                const uint32_t pmask = hp_pages ? kHashPerPage - 1u : hmask0;         // probe sequences wrap inside the table / the bucket
                auto ld_first = [&](uint64_t key, uint32_t &ii) { HashEnt *const t = hfirst(key, ii); return *reinterpret_cast<const uint4 *>(t + ii); };
                auto resolve = [&](uint64_t key, uint32_t ii, uint4 v, int &old_fval) -> uint32_t {     // open-list node recorded for `key` (+ its fval), kNone = none
                    const HashEnt *t = nullptr;                                       // (the table / bucket is looked up again only when the first entry is another key's)
                    while (true) {
                        const uint64_t k = (uint64_t)v.x | ((uint64_t)v.y << 32);
                        if (k == 0) return kNone;
                        if (k == key) { old_fval = (int)v.w; return v.z & kNone; }
                        if (!t) { uint32_t dummy; t = hfirst(key, dummy); }
                        ii = (ii + 1) & pmask;
                        v = *reinterpret_cast<const uint4 *>(t + ii);
                    }
                };
                ANode cd;                                                              // delete child (:218-244), same in every lane
                cd.parent = cur; cd.node_id = curr.node_id;
                cd.state_no = (int16_t)next_state; cd.length = curr.length;
                cd.real_score = curr.real_score + dt;
                cd.max_score = curr.max_score;
                cd.negative_count = (int16_t)(curr.negative_count + 1);
                cd.score = curr.score + (dt - max_match);
                cd.fval = to_fval(10000 * (cd.score + 2.0 * h_d));
                cd.em_state = (uint16_t)(((4 << 6) | (4 << 3) | 4) | (ST_D << 9));
                cd.fwd_r = curr.fwd_r; cd.fwd_hint = curr.fwd_hint; cd.pad = 0;         // the same edge
                // (whether a cached match / insert child suppresses the delete child is known once every codon has been looked at: the
                // probe is issued now, the verdict follows the passes)
                bool del = cst != ST_I;
                if (del && !first) del = admissible(cd.length, cd.negative_count, cd.real_score);
                const bool probe_d = del && !first;
                const uint64_t key_d = make_key(cd.node_id, cd.state_no, ST_D);
                uint32_t slot_d = 0;
                uint4 vd = make_uint4(0, 0, 0, 0);
                if (probe_d) vd = ld_first(key_d, slot_d);

                // ---- commit in the reference's order: open_hash[next] = next (:331) and open.push (:335), codon by codon
                // (ascending (first edge, second edge), then third edge), match before insert, delete last
                // (every lane of the group, same arguments) open_hash[key] = node, probing from the key's first slot
                auto hash_commit = [&](uint64_t key, int fval, uint32_t node) {
                    bool found; uint32_t val;
                    HashEnt *const hs = hfind(key, found, val);
                    if (gl == 0) hash_put(hs, 0u, key, (found ? (val & 0x80000000u) : 0u) | node, fval);
                    if (!found) ++n_keys;
                };
                // a probe whose first entry is already loaded: the entry of `key` (found) or the empty slot the sequence ends at
                auto resolve_at = [&](uint64_t key, HashEnt *t, uint32_t ii, uint4 v, bool &found, uint32_t &val, int &old_fval) -> HashEnt * {
                    while (true) {
                        const uint64_t k = (uint64_t)v.x | ((uint64_t)v.y << 32);
                        if (k == 0) { found = false; return t + ii; }
                        if (k == key) { found = true; val = v.z; old_fval = (int)v.w; return t + ii; }
                        ii = (ii + 1) & pmask;
                        v = *reinterpret_cast<const uint4 *>(t + ii);
                    }
                };
                auto commit = [&](uint64_t key, int fval, uint32_t node, bool with_hash) {
                    HeapEnt he;
                    he.key = key; he.fval = fval; he.node = node;
                    if (!first && with_hash) {
                        hash_commit(key, fval, node);
                        n_opened++;
                    }
                    H.sift_up(n_heap, he);
                    ++n_heap;
                };
                constexpr bool kCommitHashes = false;                                  // (match / insert children: written where they were probed)

                // ---- enumeration of the <= 64 codon paths (node_enumerator.h:98-128): lane (i, j) walks two edges and owns the <= 4
                // third edges that continue from there.  kWalkLanes (i, j) pairs per pass, ascending: with 8 lanes per search the first
                // pass takes the first edges 0 and 1, a second one (rare: a node with three or four out-edges) the first edges 2 and 3.
                // The children of a pass are admitted, stored and committed before the next pass walks: their keys are distinct (distinct
                // end nodes), so a later child's look-up never meets an earlier child of the same expansion, as in the reference's loop.
                bool any_pass = false;
                bool more = true;
                uint32_t nbase = n_nodes;
#pragma unroll 1
                for (int pass = 0; pass < GX::kPasses; ++pass) {
                    if (more) {
                        int64_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
                        const int vl = gl + GX::kWalkLanes * pass;
                        const int ci = (vl >> 2) & 3, cj = vl & 3;
                        int64_t e1 = 0, e2 = 0;
                        int od1 = 0;
                        bool valid = gl < GX::kWalkLanes;
                        // one line per step: every edge comes with the place its own Forward lands (FwdDesc), the node's own edge included
                        FwdDesc fd1, fd2, f0, f1, f2, f3;
                        fd1.r = 0; fd1.hint = kFdNone; fd2 = fd1; f0 = fd1; f1 = fd1; f2 = fd1; f3 = fd1;
                        if (valid) {
                            FwdDesc fc;
                            fc.r = curr.fwd_r; fc.hint = curr.fwd_hint;
                            od1 = g_out_nth_fd(g, curr.node_id, fc, ci, e1, fd1);
                            valid = ci < od1;
                        }
                        if (GX::kPasses > 1) od1 = GX::bcast(od1, 0, gbase);
                        int od3 = 0, c12 = 0, low12 = 0;
                        if (valid) {
                            valid = cj < g_out_nth_fd(g, e1 >> 4, fd1, cj, e2, fd2);
                            if (valid) {
                                od3 = g_out_all_fd(g, e2 >> 4, fd2, p0, p1, p2, p3, f0, f1, f2, f3);
                                if (od3 < 0) od3 = 0;
                                c12 = (((int)(e1 & 7) - 1) << 6) | (((int)(e2 & 7) - 1) << 3);
                                low12 = (int)((e1 >> 3) & 1) & (int)((e2 >> 3) & 1);
                            }
                        }
                        if (!valid) od3 = 0;
                        PROF(5)
                        // which of this lane's codons pass (stop codons :142-144; a cached child keeps only its own codon :146-148)
                        int use_bits = 0, cols = 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int64_t e3 = k == 0 ? p0 : k == 1 ? p1 : k == 2 ? p2 : p3;
                            if (k < od3) {
                                const int codon = c12 | ((int)(e3 & 7) - 1);
                                const int col = hv.col_enum[((codon >> 6) & 7) * 16 + ((codon >> 3) & 7) * 4 + (codon & 7)];
                                bool use = col >= 0;
                                if (use && cached >= 0) use = cached_st == ST_D ? ((e3 >> 4) == curr.node_id) : (codon == (cached & 511));
                                if (use) { use_bits |= 1 << k; cols |= col << (8 * k); }
                            }
                        }
                        any_pass = any_pass || GX::ballot(use_bits != 0, gbase) != 0ull;
                        // node indices are handed out codon rank by codon rank (k), lane by lane, match before insert: the pool order is
                        // not observable, only the order of the commits below is
                        const uint32_t n_before = nbase;
                        uint64_t MM = 0, MI = 0;                                       // admitted match / insert children: bit 16 k + lane
                        int fm0 = 0, fm1 = 0, fm2 = 0, fm3 = 0, fi0 = 0, fi1 = 0, fi2 = 0, fi3 = 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (__ballot((use_bits >> k) & 1) == 0ull) continue;       // (uniform over the active lanes)
                            const bool use = (use_bits >> k) & 1;
                            const int64_t e3 = k == 0 ? p0 : k == 1 ? p1 : k == 2 ? p2 : p3;
                            const int codon = c12 | ((int)(e3 & 7) - 1);
                            const int low = low12 & (int)((e3 >> 3) & 1);
                            const int col = (cols >> (8 * k)) & 255;
                            ANode cm, cin;                                             // this codon's match / insert child
                            cm.parent = cur; cin.parent = cur;
                            cm.node_id = e3 >> 4; cin.node_id = e3 >> 4;
                            cm.length = (int16_t)(curr.length + 1); cin.length = cm.length;
                            cm.state_no = (int16_t)next_state; cin.state_no = curr.state_no;
                            cm.em_state = (uint16_t)(codon | (ST_M << 9)); cin.em_state = (uint16_t)(codon | (ST_I << 9));
                            cm.fwd_r = k == 0 ? f0.r : k == 1 ? f1.r : k == 2 ? f2.r : f3.r;
                            cm.fwd_hint = k == 0 ? f0.hint : k == 1 ? f1.hint : k == 2 ? f2.hint : f3.hint;
                            cm.pad = 0; cin.fwd_r = cm.fwd_r; cin.fwd_hint = cm.fwd_hint; cin.pad = 0;
                            const double pen = low ? a.low_cov_penalty : 0.0;          // :150
                            {
                                const double e = mt + (use ? msc[(size_t)next_state * A + col] : 0.0);
                                cm.real_score = curr.real_score + e - pen;
                                if (cm.real_score >= curr.max_score) { cm.max_score = cm.real_score; cm.negative_count = 0; }
                                else { cm.max_score = curr.max_score; cm.negative_count = (int16_t)(curr.negative_count + 1); }
                                cm.score = curr.score + (e - pen - max_match);
                                cm.fval = to_fval(10000 * (cm.score + 2.0 * h_m));     // :173
                                const double ei = it + (next_state == M ? NEG_INF : 0.0);   // isc == 0 except at node M
                                cin.real_score = curr.real_score + ei - pen;
                                cin.max_score = curr.max_score;
                                cin.negative_count = (int16_t)(curr.negative_count + 1);
                                cin.score = curr.score + (ei - pen);
                                cin.fval = to_fval(10000 * (cin.score + 2.0 * h_i));
                            }
                            bool open_m = use, open_i = use && ins_ok;
                            // the entry of every child's key (found) or the empty slot its probe ended at (new key): open_hash[next] = next (:331) is
                            // written from here, by the child's own lane, instead of probing again at commit time
                            HashEnt *pm = nullptr, *pi = nullptr;
                            uint32_t val_m = 0, val_i = 0;
                            bool fnd_m = false, fnd_i = false;
                            const uint64_t key_m = make_key(cm.node_id, cm.state_no, ST_M), key_i = make_key(cin.node_id, cin.state_no, ST_I);
                            if (!first) {                                              // :212-233: the first expansion neither prunes nor looks up
                                open_m = open_m && admissible(cm.length, cm.negative_count, cm.real_score);
                                open_i = open_i && admissible(cin.length, cin.negative_count, cin.real_score);
                                uint32_t slot_m = 0, slot_i = 0;
                                uint4 vm = make_uint4(0, 0, 0, 0), vi = vm;
                                HashEnt *tm = nullptr, *ti = nullptr;
                                if (open_m) { tm = hfirst(key_m, slot_m); vm = *reinterpret_cast<const uint4 *>(tm + slot_m); }
                                if (open_i) { ti = hfirst(key_i, slot_i); vi = *reinterpret_cast<const uint4 *>(ti + slot_i); }
                                int old_m = 0, old_i = 0;
                                if (open_m) pm = resolve_at(key_m, tm, slot_m, vm, fnd_m, val_m, old_m);
                                if (open_i) pi = resolve_at(key_i, ti, slot_i, vi, fnd_i, val_i, old_i);
                                if (fnd_m && (val_m & kNone) != kNone) open_m = old_m < cm.fval;   // got->second < next (:299-302); equal keys => only fval differs
                                if (fnd_i && (val_i & kNone) != kNone) open_i = old_i < cin.fval;
                            }
                            const uint64_t mm = GX::ballot(open_m, gbase), mi = GX::ballot(open_i, gbase);
                            const uint32_t idx_m = nbase + (uint32_t)__popcll(mm & lt_mask) + (uint32_t)__popcll(mi & lt_mask);
                            if (open_m) store_node(node_at(idx_m), cm);
                            if (open_i) store_node(node_at(idx_m + (open_m ? 1u : 0u)), cin);
                            if (!first && (mm | mi) != 0ull) {
                                // The children's keys are distinct, so their entries are too -- except that two NEW keys of this round may have ended
                                // their probes at the same empty slot.  A bit per new key (six address bits of its slot), OR-ed over the group:
                                // fewer bits than new keys = two of them may share a slot, and this round's entries are written one after the
                                // other, each probing again, as the commit used to (rare: the table is at most half full).
                                const bool new_m = open_m && !fnd_m, new_i = open_i && !fnd_i;
                                const uint64_t bit_m = new_m ? 1ull << ((reinterpret_cast<uintptr_t>(pm) >> 4) & 63u) : 0ull;
                                const uint64_t bit_i = new_i ? 1ull << ((reinterpret_cast<uintptr_t>(pi) >> 4) & 63u) : 0ull;
                                const uint32_t n_new = (uint32_t)__popcll(GX::ballot(new_m, gbase)) + (uint32_t)__popcll(GX::ballot(new_i, gbase));
                                const bool clash = n_new > 1u && (uint32_t)__popcll(GX::or64(bit_m | bit_i)) < n_new;
                                const uint32_t idx_i = idx_m + (open_m ? 1u : 0u);
                                if (!clash) {
                                    if (open_m) hash_put(pm, 0u, key_m, (fnd_m ? (val_m & 0x80000000u) : 0u) | idx_m, cm.fval);
                                    if (open_i) hash_put(pi, 0u, key_i, (fnd_i ? (val_i & 0x80000000u) : 0u) | idx_i, cin.fval);
                                    n_keys += n_new;
                                } else {
                                    uint32_t todo = (uint32_t)(mm | mi);
                                    while (todo) {
                                        const int l = __builtin_ctz(todo);
                                        todo &= todo - 1;
                                        if ((mm >> l) & 1ull) hash_commit(GX::bcast(key_m, l, gbase), GX::bcast(cm.fval, l, gbase), GX::bcast(idx_m, l, gbase));
                                        if ((mi >> l) & 1ull) hash_commit(GX::bcast(key_i, l, gbase), GX::bcast(cin.fval, l, gbase), GX::bcast(idx_i, l, gbase));
                                    }
                                }
                                n_opened += (uint32_t)__popcll(mm) + (uint32_t)__popcll(mi);
                            }
                            nbase += (uint32_t)__popcll(mm) + (uint32_t)__popcll(mi);
                            MM |= (mm & 0xFFFFull) << (16 * k);
                            MI |= (mi & 0xFFFFull) << (16 * k);
                            if (k == 0) { fm0 = cm.fval; fi0 = cin.fval; } else if (k == 1) { fm1 = cm.fval; fi1 = cin.fval; }
                            else if (k == 2) { fm2 = cm.fval; fi2 = cin.fval; } else { fm3 = cm.fval; fi3 = cin.fval; }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                        PROF(6)
                        const uint64_t anyk = MM | MI;
                        uint32_t lanes_todo = (uint32_t)((anyk | (anyk >> 16) | (anyk >> 32) | (anyk >> 48)) & 0xFFFFull);
                        // The searches of a wave have their children at different (lane, codon rank) places: walking those places one by one makes
                        // every search sit through the places of all the others.  Each lane writes its own children's open-list entries to
                        // the search's LDS stage at their rank in the commit order (lane, then codon rank, match before insert), and the
                        // pushes then go child by child: the wave's loop is as long as its busiest search's child count.
                        const uint32_t n_c = (uint32_t)__popcll(MM) + (uint32_t)__popcll(MI);
                        if (n_c != 0u && n_c <= kStage) {
                            HeapEnt *const stage = s_stage + (size_t)lslot * kStage;
                            const uint32_t lo16 = gl < 16 ? (1u << gl) - 1u : 0xFFFFu;
                            const uint64_t LB = 0x0001000100010001ull * (uint64_t)lo16;
                            uint32_t r = (uint32_t)__popcll(MM & LB) + (uint32_t)__popcll(MI & LB);
                            uint32_t kb = n_before;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const uint32_t mmk = (uint32_t)(MM >> (16 * k)) & 0xFFFFu, mik = (uint32_t)(MI >> (16 * k)) & 0xFFFFu;
                                const bool hm_ = gl < 16 && ((mmk >> gl) & 1u), hi_ = gl < 16 && ((mik >> gl) & 1u);
                                if (hm_ || hi_) {
                                    const int64_t e3 = k == 0 ? p0 : k == 1 ? p1 : k == 2 ? p2 : p3;
                                    const uint32_t idx = kb + (uint32_t)__popc(mmk & lo16) + (uint32_t)__popc(mik & lo16);
                                    HeapEnt he;
                                    if (hm_) {
                                        he.key = make_key(e3 >> 4, next_state, ST_M); he.fval = k == 0 ? fm0 : k == 1 ? fm1 : k == 2 ? fm2 : fm3; he.node = idx;
                                        stage[r++] = he;
                                    }
                                    if (hi_) {
                                        he.key = make_key(e3 >> 4, curr.state_no, ST_I); he.fval = k == 0 ? fi0 : k == 1 ? fi1 : k == 2 ? fi2 : fi3; he.node = idx + (hm_ ? 1u : 0u);
                                        stage[r++] = he;
                                    }
                                }
                                kb += (uint32_t)__popc(mmk) + (uint32_t)__popc(mik);
                            }
                            wave_lds_fence();
                            for (uint32_t c = 0; c < n_c; ++c) {
                                const HeapEnt he = stage[c];
                                commit(he.key, he.fval, he.node, kCommitHashes);
                            }
                            wave_lds_fence();
                            lanes_todo = 0u;
                        }
                        while (lanes_todo) {
                            const int l = __builtin_ctz(lanes_todo);
                            lanes_todo &= lanes_todo - 1;
                            // this lane's children in k order; what they need from lane l: the third edges and the fvals
                            const int64_t q0 = GX::bcast(p0, l, gbase), q1 = GX::bcast(p1, l, gbase), q2 = GX::bcast(p2, l, gbase), q3 = GX::bcast(p3, l, gbase);
                            const uint32_t below = (1u << l) - 1u;
                            uint32_t kbase = n_before;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const uint32_t mmk = (uint32_t)(MM >> (16 * k)) & 0xFFFFu, mik = (uint32_t)(MI >> (16 * k)) & 0xFFFFu;
                                const bool hm_ = (mmk >> l) & 1u, hi_ = (mik >> l) & 1u;
                                if (hm_ || hi_) {
                                    const int64_t e3 = k == 0 ? q0 : k == 1 ? q1 : k == 2 ? q2 : q3;
                                    const uint32_t idx = kbase + (uint32_t)__popc(mmk & below) + (uint32_t)__popc(mik & below);
                                    if (hm_) commit(make_key(e3 >> 4, next_state, ST_M), GX::bcast(k == 0 ? fm0 : k == 1 ? fm1 : k == 2 ? fm2 : fm3, l, gbase), idx, kCommitHashes);
                                    if (hi_) commit(make_key(e3 >> 4, curr.state_no, ST_I), GX::bcast(k == 0 ? fi0 : k == 1 ? fi1 : k == 2 ? fi2 : fi3, l, gbase),
                                                    idx + (hm_ ? 1u : 0u), kCommitHashes);
                                }
                                kbase += (uint32_t)__popc(mmk) + (uint32_t)__popc(mik);
                            }
                        }
                        PROF(7)
                        more = od1 > ((GX::kWalkLanes * (pass + 1)) >> 2);               // the node has first edges beyond this pass
                    }
                    if (GX::kPasses == 1 || __ballot(more) == 0ull) break;
                }
                // the delete child, last (:218-244)
                if (del && (cached_st == ST_M || cached_st == ST_I) && any_pass) del = false;
                if (del && probe_d) {
                    int old_d = 0;
                    const uint32_t od = resolve(key_d, slot_d, vd, old_d);
                    if (od != kNone) del = old_d < cd.fval;
                }
                const uint32_t idx_d = nbase;
                if (del && gl == 0) store_node(node_at(idx_d), cd);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (del) commit(make_key(cd.node_id, cd.state_no, ST_D), cd.fval, idx_d, true);
                n_nodes = nbase + (del ? 1u : 0u);
                if (first) {
                    first = false;
                    n_opened = 1;
                    if (n_heap == 0) { ok = 0; stop = true; }                          // :235-237
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
            PROF(7)
            if (stop) st = S_DONE;
        }
        PROF_IDLE_LANES
        PROF(8)
        if (__ballot(st == S_RUN && starved == 0) == 0ull && __ballot(st == S_RUN) != 0ull) {   // every running search of the wave waits for memory
#pragma unroll
            for (int z = 0; z < 4; ++z) __builtin_amdgcn_s_sleep(127);
#ifdef MGTA_ASTAR_PROFILE
            pacc_[15] += 1;
#endif
        }
        PROF(14)

        // ================= result: getHighestScoreNode + partialResultFromGoal (hmm_graph_search.h:83-110,345-356)
        const bool finishing = st == S_DONE;
        if (finishing) {
            mgta_astar_side r;
            r.ok = ok; r.partial = partial; r.n_closed = n_closed; r.n_expanded = n_expanded; r.n_opened = n_opened;
            r.fval = 0; r.length = 0; r.state_no = -1; r.state = '-'; r.node_id = -1; r.real_score = 0; r.score = 0;
            uint32_t len = 0;
            char *dst = a.out_seq + (size_t)sid * a.out_cap;
            if (status == 1 && ok && goal >= 0) {
                int32_t best = goal;
                ANode nd = load_node(node_at((uint32_t)goal));
                double best_rs = nd.real_score;
                for (int32_t p = nd.parent; p >= 0;) {
                    nd = load_node(node_at((uint32_t)p));
                    if (nd.real_score > best_rs) { best = p; best_rs = nd.real_score; }
                    p = nd.parent;
                }
                const ANode gn = load_node(node_at((uint32_t)best));
                r.fval = gn.fval; r.length = gn.length; r.state_no = gn.state_no;
                r.state = "mid"[gn.em_state >> 9]; r.node_id = gn.node_id; r.real_score = gn.real_score; r.score = gn.score;
                // 3 characters per non-delete node from the goal back to the start, then reversed (:92-108);
                // term_nodes.insert(parent -> child) along the same walk (:97-103)
                nd = gn;
                while (nd.parent >= 0) {
                    if ((nd.em_state >> 9) != ST_D) {
                        if (len + 3 > a.out_cap) { status = 2; break; }
                        if (gl == 0)
                            for (int t = 0; t < 3; ++t) dst[len + t] = "acgt-"[(nd.em_state >> (3 * t)) & 7];
                        len += 3;
                    }
                    const ANode par = load_node(node_at((uint32_t)nd.parent));
                    if (a.window > 0 && gl == 0)
                        cache_insert(a, dir, make_key(par.node_id, par.state_no, par.em_state >> 9),
                                     order_off ? 0 : seed + a.window + cost_term(a, n_expanded), nd.em_state);
                    nd = par;
                }
                if (gl == 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    for (uint32_t x = 0; x < len / 2; ++x) { char t = dst[x]; dst[x] = dst[len - 1 - x]; dst[len - 1 - x] = t; }
                }
            }
            if (gl == 0) {
                a.sides[sid] = r;
                a.out_len[sid] = len;
                a.status[sid] = status;
            }
            // everything above the base arena goes back to the pool
            release_all();
            if (use_reserve) {           // the reserve's owner ends: all of it is free for the next lowest search
                if (gl == 0) reserve_release(a.pool);
                use_reserve = false;
            }
            if (a.gate && gl == 0) {     // the paths are in the cache (atomics, all performed): this search no longer holds anybody back
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                st_agent(&a.run_seed[slot], -1ll);
                __hip_atomic_fetch_add(&a.start_limit[6 + dir], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (slow start: one more search may be in flight)
            }
            st = S_IDLE;
        }
        if (a.gate && __ballot(finishing) != 0ull) (void)start_bound<G>(a, dir, lane);   // whoever finishes a search moves the limit for the waiting ones
        PROF(9)
    }
    PROF_FLUSH
#undef base_hash
#undef sid
#undef cap_nodes
#undef cap_heap
}

}  // namespace mgta
