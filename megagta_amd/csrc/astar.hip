// astar.hip — batched HMM-guided A* over the device-resident succinct de Bruijn graph (gfx950): host side + C ABI.
//
// Replaces the OMP seed loop of search() (search.cpp:184-189) and, per seed and direction,
// HMMGraphSearch::astarSearch (hmm_graph_search.h:132-343) with NodeEnumerator::enumerateNodes
// (node_enumerator.h:65-246), AStarNode ordering (a_star_node.h:34-82) and the result walk
// getHighestScoreNode / partialResultFromGoal (hmm_graph_search.h:83-110,345-356).
// The kernel and its data structures are in astar_kernel.hpp.  Scores are IEEE fp64, compiled with -ffp-contract=off:
// path log-probabilities are bit-identical to the x86-64 reference, fval = (int)(10000*(score+2h)) truncates identically.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <string>
#include <thread>

#include "astar_kernel.hpp"
#include "scan.hpp"

struct mgta_hmm {
    mgta_ctx *ctx = nullptr;
    int M = 0, A = 0;
    mgta::DevBuf tab;            // [msc (M+1)*A][tsc 7*(M+1)][maxm (M+1)][h 3*(M+1)]
    int8_t col[2][64];           // codon (c1*16+c2*4+c3) -> emission column; [0] codonTable, [1] rc_codonTable; -1 = stop
    mgta::DevBuf d_col;          // the same 128 bytes on the device
    size_t n_doubles = 0;
};

using namespace mgta;

static const char kCodonAA[65] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";   // codon.h:9-106

namespace {
// the result strings of a batch, packed: side sid's `len[sid]` characters move from its fixed-size slot to packed[off[sid] ...) (a
// million-seed batch has 5 GB of slots for 0.7 GB of text: the slots never cross the bus)
__global__ __launch_bounds__(256) void pack_results_kernel(const char *slots, uint32_t slot_bytes, const uint32_t *len, const uint64_t *off, uint64_t n_sides,
                                                           char *packed) {
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    for (uint64_t sid = wave; sid < n_sides; sid += n_waves) {
        const uint32_t n = len[sid];
        const char *src = slots + sid * slot_bytes;
        char *dst = packed + off[sid];
        for (uint32_t i = (uint32_t)lane; i < n; i += 64) dst[i] = src[i];
    }
}
struct Events {                  // RAII: the events also go on the early-return and throw paths
    hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
    Events() { for (auto &x : e) MGTA_HIP_CHECK(hipEventCreate(&x)); }
    ~Events() { for (auto &x : e) if (x) (void)hipEventDestroy(x); }
};

template <int G>
void launch_astar(const mgta::AstarArgs &a, int blocks, size_t lds_bytes, bool use_lds, hipStream_t st) {
    using namespace mgta;
    if (use_lds) {
        MGTA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(astar_kernel<G, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL((astar_kernel<G, true>), dim3(blocks), dim3(kAstarThreads), lds_bytes, st, a);
    } else {
        MGTA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(astar_kernel<G, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL((astar_kernel<G, false>), dim3(blocks), dim3(kAstarThreads), lds_bytes, st, a);
    }
    MGTA_HIP_CHECK(hipGetLastError());
}
template <int G> size_t lds_fixed() {
    return (size_t)mgta::kAstarWaves * mgta::Grp<G>::kGroups * (mgta::Grp<G>::kLdsHeap * sizeof(mgta::HeapEnt) + mgta::kPtWords * sizeof(uint32_t) + mgta::kStage * sizeof(mgta::HeapEnt));
}
}  // namespace

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


int mgta_ctx_set_search_arena(mgta_ctx *ctx, int log2_base_nodes, uint64_t pool_bytes) {
    if (!ctx || (log2_base_nodes != 0 && (log2_base_nodes < 7 || log2_base_nodes > 20))) { set_error("mgta_ctx_set_search_arena: bad argument"); return MGTA_EINVAL; }
    ctx->astar_log_b0 = log2_base_nodes;
    ctx->astar_pool_bytes = pool_bytes;
    return MGTA_OK;
}

int mgta_ctx_set_search_share(mgta_ctx *ctx, int num, int den) {
    if (!ctx || num < 1 || den < num) { set_error("mgta_ctx_set_search_share: 1 <= num <= den"); return MGTA_EINVAL; }
    ctx->search_share_num = num; ctx->search_share_den = den;
    return MGTA_OK;
}

int mgta_astar_batch(mgta_sdbg *g, const mgta_hmm *fwd, const mgta_hmm *rev, const char *kmers, const int32_t *start_state, int64_t n,
                     int prune_len, double low_cov_penalty, int cache_mode, mgta_contig_sink sink, void *user, mgta_astar_stats *stats) {
    return mgta_astar_batch_on(g ? g->ctx : nullptr, g, fwd, rev, kmers, start_state, n, prune_len, low_cov_penalty, cache_mode, sink, user, stats);
}

}  // extern "C"

namespace {
struct PackedOut {               // mgta_astar_batch_packed: the contigs written straight into one malloc'd buffer
    char **contigs;
    uint64_t *offsets;           // [n + 1]
    mgta_astar_side *sides;      // [2 n] or null
};
int astar_batch_impl(mgta_ctx *ctx, mgta_sdbg *g, const mgta_hmm *fwd, const mgta_hmm *rev, const char *kmers, const int32_t *start_state,
                     int64_t n, int prune_len, double low_cov_penalty, int cache_mode, mgta_contig_sink sink, void *user,
                     mgta_astar_stats *stats, const PackedOut *packed);
}  // namespace

extern "C" {
int mgta_astar_batch_on(mgta_ctx *ctx, mgta_sdbg *g, const mgta_hmm *fwd, const mgta_hmm *rev, const char *kmers, const int32_t *start_state,
                        int64_t n, int prune_len, double low_cov_penalty, int cache_mode, mgta_contig_sink sink, void *user,
                        mgta_astar_stats *stats) {
    // (no exception crosses the C boundary: a host allocation that fails inside is an error code like any other)
    try {
        return astar_batch_impl(ctx, g, fwd, rev, kmers, start_state, n, prune_len, low_cov_penalty, cache_mode, sink, user, stats, nullptr);
    } catch (const std::bad_alloc &) { set_error("mgta_astar_batch: out of host memory"); return MGTA_ENOMEM; }
      catch (const std::exception &e) { set_error("mgta_astar_batch: %s", e.what()); return MGTA_EHIP; }
}
}  // extern "C"

namespace {
int astar_batch_impl(mgta_ctx *ctx, mgta_sdbg *g, const mgta_hmm *fwd, const mgta_hmm *rev, const char *kmers, const int32_t *start_state,
                     int64_t n, int prune_len, double low_cov_penalty, int cache_mode, mgta_contig_sink sink, void *user,
                     mgta_astar_stats *stats, const PackedOut *packed) {
    if (!ctx || !g || !fwd || !rev || n < 0 || (n > 0 && (!kmers || !start_state))) { set_error("mgta_astar_batch: bad argument"); return MGTA_EINVAL; }
    if (cache_mode < -1) { set_error("cache_mode must be >= -1"); return MGTA_EINVAL; }
    const bool free_share = cache_mode == -1;          // shared caches without any ordering (timing-dependent results, like the reference's OMP run)
    if (free_share) cache_mode = 1;
    if (ctx->device != g->ctx->device) { set_error("mgta_astar_batch_on: the context and the graph live on different devices"); return MGTA_EINVAL; }
    const int klen = g->dev.k + 1;
    if (klen > kMaxKmer) { set_error("k too large"); return MGTA_EINVAL; }
    try {
        MGTA_HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        mgta_astar_stats ST;
        memset(&ST, 0, sizeof(ST));
        ST.n_seeds = n;
        if (n == 0) { if (stats) *stats = ST; return MGTA_OK; }
        Events ev;
        MGTA_HIP_CHECK(hipEventRecord(ev.e[0], st));

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

        // lanes per search: 16 (four searches per wavefront, 8192 in flight).  One search per wavefront (64 lanes) was the choice for
        // shared-cache batches while their windows were thousands of seeds wide; with the small windows + cost term `megagta search` uses
        // 16 lanes win at every size measured (profiles/r02/e2e_window_sweep.log).  Eight lanes per search (16 384 in flight, two walk
        // passes) pay where the batch is large and independent: cold, 120 000 seeds on the 100 M-read graph 46.6 -> 43.6 s, 40 000 seeds
        // on the 10 M-read graph +14 %; they lose where single searches bound the run (30 000 seeds per gene at 100 M reads -5 %, the
        // ordered window on the 100 M-read graph 38.6 -> 42.7 s): profiles/r03/astar_ab.md.  MGTA_ASTAR_GROUP=8|16|32|64 overrides.
        // Round 5, ordered window on the driver's multi-k graphs (profiles/r05/trials_*_lanes*.log, same box back to back): 2 M reads (80 k / 108 k
        // seeds, 0.1 G edges) rplB 4.9 -> 4.7 s, nirK 8.0 -> 7.3 s; 20 M reads (0.8 M / 1.06 M seeds, 1.3 G edges) 23.2 -> 24.3 s and 44.5 -> 36.4 s;
        // 50 M reads (2.64 M seeds, 3.2 G edges: first-of-their-gene-copy searches of millions of expansions) nirK 179 -> 188 s: eight lanes for
        // ordered batches of 65 536 seeds and more on graphs of up to 2 G edges.
        int G = (cache_mode == 0 && n >= 32768) ? 8 : 16;
        if (cache_mode > 0 && !free_share && n >= 65536 && g->dev.size <= (2ll << 30)) G = 8;
        if (const char *e = getenv("MGTA_ASTAR_GROUP")) { int v = atoi(e); if (v == 8 || v == 16 || v == 32 || v == 64) G = v; }
        const int groups = 64 / G;
        const int64_t spb = (int64_t)kAstarWaves * groups;                          // search slots per workgroup

        AstarArgs a;
        memset(&a, 0, sizeof(a));
        a.g = g->dev;
        const mgta_hmm *hm[2] = {fwd, rev};
        size_t tab_bytes = 0;
        for (int d = 0; d < 2; ++d) {
            a.hm[d].tab = hm[d]->tab.as<double>(); a.hm[d].M = hm[d]->M; a.hm[d].A = hm[d]->A;
            a.hm[d].col_fwd = hm[d]->d_col.as<int8_t>();
            a.hm[d].col_enum = hm[d]->d_col.as<int8_t>() + 64 * d;
            tab_bytes = std::max(tab_bytes, hm[d]->n_doubles * 8);
        }
        a.kmers = d_kmers.as<char>(); a.start_state = d_ss.as<int32_t>(); a.start_node = d_sn.as<int64_t>();
        a.n_seeds = n; a.klen = klen; a.prune = prune_len; a.low_cov_penalty = lcp; a.log2v = std::log(2.0);
        a.exit_prob = d_exit.as<double>();
        a.queue = d_queue.as<unsigned long long>();
        a.sides = d_sides.as<mgta_astar_side>(); a.out_seq = d_out.as<char>(); a.out_cap = out_cap; a.out_len = d_len.as<uint32_t>();
        a.status = d_status.as<int32_t>();
        DevBuf d_prof;
        d_prof.alloc(128);
        MGTA_HIP_CHECK(hipMemsetAsync(d_prof.p, 0, 128, st));
        a.prof = d_prof.as<unsigned long long>();
        DevBuf d_tmark;
        d_tmark.alloc(32);
        a.tmark = d_tmark.as<unsigned long long>();
        const size_t lds_fix = G == 8 ? lds_fixed<8>() : G == 16 ? lds_fixed<16>() : G == 32 ? lds_fixed<32>() : lds_fixed<64>();
        const bool use_lds = lds_fix + tab_bytes + 1024 <= 160 * 1024;             // heap tops + level tables + HMM tables
        const size_t lds_bytes = lds_fix + (use_lds ? tab_bytes : 0);
        ST.hmm_in_lds = use_lds ? 1 : 0;

        std::vector<int64_t> todo[2];
        const int hm_M[2] = {fwd->M, rev->M};
        for (int d = 0; d < 2; ++d) { todo[d].resize(n); for (int64_t s = 0; s < n; ++s) todo[d][s] = s; }
        // Independent searches (cold) may be taken in any order, and a batch cannot end before its longest search does (one expansion of one
        // search is a chain of dependent line fetches: tens of microseconds, whatever else the device is doing).  The searches that promise the
        // most work -- the most model columns still to cover on their side -- are therefore started FIRST (longest processing time first), so
        // that the long ones run beside the bulk instead of after it.  Results are per seed and do not depend on the order.
        // MGTA_ASTAR_LPT=0 keeps the seed order.  (Shared-cache batches: the order IS the semantics, never touched.)
        if (cache_mode == 0) {
            const char *e = getenv("MGTA_ASTAR_LPT");
            if (!e || atoi(e) != 0) {
                for (int d = 0; d < 2; ++d) {
                    const int Md = hm_M[d];
                    std::stable_sort(todo[d].begin(), todo[d].end(), [&](int64_t x, int64_t y) {
                        const int cx = d == 0 ? Md - start_state[x] - klen / 3 : start_state[x], cy = d == 0 ? Md - start_state[y] - klen / 3 : start_state[y];
                        return cx > cy;
                    });
                }
            }
        }
        std::vector<int32_t> h_status((size_t)n * 2);
        std::vector<char> over_limit_seen((size_t)n * 2, 0);
        size_t free_b = 0, total_b = 0;
        MGTA_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));

        // shared term_nodes caches: one open-addressing table per direction, sized for the entries the searches can insert (one per
        // node of a result path: about the model length per search), never more than a quarter of the free memory each; an insert
        // that finds the neighbourhood of its slot full is dropped (a missed cache entry costs expansions, never correctness)
        DevBuf d_cache[2], d_run_seed, d_run_progress, d_start_limit;
        a.window = cache_mode;
        a.cost_rate = cache_mode > 0 && !free_share ? ctx->search_cost_rate : 0;
        a.cost_knee = a.cost_rate > 0 ? ctx->search_cost_knee : 0;
        a.cost_rate2 = a.cost_knee ? ctx->search_cost_rate2 : 0;
        a.cache_probe_limit = 256;
        if (cache_mode > 0) {
            for (int d = 0; d < 2; ++d) {
                // (one entry per DISTINCT node of the result paths.  Up to 1 GB the table holds "every seed a path of its own" twice over;
                // beyond that the paths of one gene copy's seeds share theirs and an eighth of it is generous -- 2 x 36 GB of tables
                // took a fifth of the device from nirK's searches at 50 M reads; MGTA_ASTAR_CACHE_DIV overrides the divisor.  An insert
                // that finds no room is counted: mgta_astar_stats.n_cache_drops, 0 in every run so far)
                uint64_t div = 8;
                if (const char *e = getenv("MGTA_ASTAR_CACHE_DIV")) div = (uint64_t)std::max(1, atoi(e));
                uint64_t want = 2ull * (uint64_t)n * ((uint64_t)hm[d]->M + 64), cap = 1024;
                const uint64_t gb = (1ull << 30) / sizeof(CacheEnt);                 // tables below 1 GB keep the worst-case size
                if (want > gb) want = std::max(gb, want / div);
                while (cap < want) cap <<= 1;
                while (cap > 1024 && cap * sizeof(CacheEnt) > free_b / 8) cap >>= 1;
                d_cache[d].alloc(cap * sizeof(CacheEnt), &ctx->live_bytes, &ctx->peak_bytes);
                MGTA_HIP_CHECK(hipMemsetAsync(d_cache[d].p, 0, cap * sizeof(CacheEnt), st));
                a.cache[d] = d_cache[d].as<CacheEnt>(); a.cache_mask[d] = cap - 1;
            }
            d_start_limit.alloc(128);                                               // [0..1] limit per direction, [2..3] scan lock, [4] pass given up, [5] call for memory;
            a.start_limit = d_start_limit.as<unsigned long long>();                 // [8..10] the reserve's bump pointer, high-water mark and owner
            a.pool.rbump = a.start_limit + 8;
            MGTA_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        }

        // ordered launches (window B >= 1) never re-run single searches: a search that starves yields its memory and starts again in place
        // (astar_kernel.hpp), so the result stays a function of (seed order, B, rate), and the lowest running search -- the one every
        // later seed waits for -- has a RESERVE of its own behind the pool.  Only when that search has used up the reserve as well does the
        // pass give up.  Everything that has ended by then is final (a seed only ever started once nothing unfinished could still become
        // visible to it), so the next pass RESUMES behind the commit frontier: results, caches and statuses stay, the seeds that have not
        // ended run again in their order, with a larger share of the memory in the reserve (1/8, 1/2, 7/8; the last pass one search per
        // direction at a time with everything).  The reference has no such limit to hit: PoolST / HashMapST grow until the host is out of
        // memory (pool_st.h:43, hash_table_st.h:559-568).
        const bool gated = cache_mode > 0 && !free_share;
        // base arena of a slot: 8192 nodes (1 MB with its heap slots and hash table; round 3: 4096).  Beyond it a search takes 2 MB pages -- three
        // at least --, and how many searches may be in flight is decided by the bytes they hold: with 4096 nodes the searches of 5-50 k nodes,
        // the bulk of a 2 M-read run, held 6 MB each where they used 1.5 (reads -> contigs there: 21.0 s with 4096, 17.1 s with 8192 / 16384)
        int log_b0 = ctx->astar_log_b0 ? ctx->astar_log_b0 : 13;
        if (const char *e = getenv("MGTA_ASTAR_LOG_B0")) { const int v = atoi(e); if (!ctx->astar_log_b0 && v >= 7 && v <= 20) log_b0 = v; }   // (experiments)
        const uint64_t slot_bytes = 128ull << log_b0;                               // per node of the base arena: 64 B + 2 heap slots + 2 hash entries of 16 B
        AstarArenas &ar = ctx->astar;
        for (int attempt = 0; attempt < 4; ++attempt) {
            const int64_t work = (int64_t)std::max(todo[0].size(), todo[1].size());
            if (work == 0) break;
            // persistent grid: one workgroup per CU and direction pair, fewer when there is little work; a pass that re-runs the
            // searches the pool could not hold runs fewer at a time
            int64_t blocks = std::min<int64_t>((int64_t)ctx->num_cus * (use_lds ? 1 : 2), 2 * ((work + spb - 1) / spb));
            // a context that shares the device with another batch (two genes searched side by side) takes its share of the CUs
            blocks = std::min<int64_t>(blocks, std::max<int64_t>(2, (int64_t)ctx->num_cus * (use_lds ? 1 : 2) * ctx->search_share_num / ctx->search_share_den));
            if (const char *e = getenv("MGTA_ASTAR_BLOCKS")) blocks = std::min<int64_t>(blocks, std::max(2, atoi(e)));   // (diagnostic)
            if (!gated && attempt == 2) blocks = std::max<int64_t>(2, blocks / 8);
            if (attempt == 3) blocks = 2;
            blocks = std::max<int64_t>(2, blocks + (blocks & 1));
            const uint64_t slots = (uint64_t)blocks * spb;
            // the two directions share the workgroups by the work their seeds promise: a forward search from model position s has
            // M - s - (k+1)/3 columns to cover, a reverse one s (50 M reads, rplB: with halves the reverse searches were done after 35 s
            // and their half of the device idled for the other 36 s); between a quarter and three quarters each
            double w_dir[2] = {0, 0};
            for (int d = 0; d < 2; ++d)
                for (int64_t sd : todo[d]) {
                    const double cols = d == 0 ? (double)(hm[0]->M - start_state[sd] - klen / 3) : (double)start_state[sd];
                    w_dir[d] += std::max(1.0, cols);
                }
            int64_t blocks0 = (int64_t)std::llround((double)blocks * w_dir[0] / std::max(1.0, w_dir[0] + w_dir[1]));
            blocks0 = std::max<int64_t>(std::max<int64_t>(1, blocks / 4), std::min<int64_t>(blocks - std::max<int64_t>(1, blocks / 4), blocks0));
            if (todo[0].empty()) blocks0 = 1;
            if (todo[1].empty()) blocks0 = blocks - 1;
            if (getenv("MGTA_ASTAR_EVEN_SPLIT")) blocks0 = blocks / 2;
            a.blocks_dir0 = (uint32_t)blocks0;
            // pool = the slots' base arenas + what the searches grow into (+ the reserve).  Device memory beyond the first ~24 GB of a
            // process costs 20-90 ms/GB to obtain (profiles/r02/vmm_probe.log), so the pool follows the job: at least 4 GB; 24 MB per
            // search in flight (196 GB for a full grid) for independent searches -- at 100 M reads they average 29 k expansions
            // and hold 62 GB together -- and 8 MB per slot (64 GB) where the searches share their paths and most end after a few
            // hundred; 16 MB for batches of a million searches and more (20 M reads: 68 GB in use at most, 93 GB handed out).  No new search starts while half
            // of it is in use, so a small pool costs searches in flight, not failures.
            const uint64_t n_search = (uint64_t)work * 2;
            // (graphs of billions of edges: the searches run longer, the 8 MB that serve a 2 M-read graph starved nirK's at 100 M reads)
            // (round 4, multi-k graphs of 50 M reads and more: the first-of-their-gene-copy searches of nirK reach 1-4 M nodes, 150-500 MB
            // each, and how many of them run side by side is what the first minutes of such a batch cost: 24 MB per slot there too)
            const uint64_t per_slot = cache_mode == 0 ? (24ull << 20) : (n_search >= (1ull << 20) || g->dev.size > (3ll << 30)) ? (n_search >= (1ull << 21) ? (24ull << 20) : (16ull << 20)) : (8ull << 20);
            uint64_t per_slot_used = per_slot;
            if (const char *e = getenv("MGTA_ASTAR_PER_SLOT_MB")) per_slot_used = (uint64_t)std::max(1, atoi(e)) << 20;   // (experiments)
            uint64_t dyn = ctx->astar_pool_bytes ? ctx->astar_pool_bytes
                                                 : std::max<uint64_t>(4ull << 30, std::min<uint64_t>(slots, n_search) * per_slot_used);
            const uint64_t avail = (uint64_t)((double)(free_b + ar.pool.bytes) * 0.8);
            const uint64_t room = avail > slots * slot_bytes ? avail - slots * slot_bytes : 0;
            uint64_t reserve = 0;
            if (gated) {
                // dyn is what the searches share, the reserve comes on top in the first pass (an eighth of it, at least 1 GB where the
                // pool is sized by the job); a pass that resumes takes all the memory there is and moves the border
                static const int kShare8[4] = {1, 4, 7, 0};                         // eighths of the whole in the reserve
                if (ctx->astar_pool_bytes) {                                        // (tests: pool_bytes is the whole)
                    reserve = attempt < 3 ? dyn * kShare8[attempt] / 8 : 0;
                    dyn -= reserve;
                } else if (attempt == 0) {
                    reserve = std::max<uint64_t>(dyn / 8, 1ull << 30);
                    if (dyn + reserve > room) { const uint64_t whole = std::min(dyn + reserve, room); reserve = whole / 8; dyn = whole - reserve; }
                } else {
                    reserve = attempt < 3 ? room / 8 * kShare8[attempt] : 0;
                    dyn = room - reserve;
                }
            } else {
                // the first re-run has the pool of the first pass to itself with a fraction of the searches; only the later ones ask for
                // everything that is free (obtaining 200 GB takes seconds)
                if (attempt == 1 && !ctx->astar_pool_bytes && ar.pool.bytes > slots * slot_bytes) dyn = std::max<uint64_t>(dyn, ar.pool.bytes - slots * slot_bytes);
                if (attempt > 1 && !ctx->astar_pool_bytes) dyn = avail;
                dyn = std::min<uint64_t>(dyn, room);
            }
            // a pool of nearly that size is there (the previous gene's, sized from a slightly different count of free bytes): keep it
            // rather than obtain 100+ GB again for a few per cent more
            if (attempt == 0 && !ctx->astar_pool_bytes && ar.pool.p && ar.pool.bytes > slots * slot_bytes + reserve &&
                ar.pool.bytes - slots * slot_bytes - reserve >= dyn - dyn / 4 && ar.pool.bytes - slots * slot_bytes - reserve < dyn)
                dyn = ar.pool.bytes - slots * slot_bytes - reserve;
            dyn &= ~((1ull << kUnitLog) - 1);
            reserve &= ~((1ull << kUnitLog) - 1);
            const uint64_t pool_bytes = slots * slot_bytes + dyn + reserve;
            if (ar.pool.bytes < pool_bytes || !ar.pool.p) {
                ar.pool.release();
                ar.pool.alloc(pool_bytes, &ctx->live_bytes, &ctx->peak_bytes);
            }
            // free lists: one stack per class, as many entries as chunks of that class fit into the pool (capped)
            std::vector<uint32_t> meta(2 * kNumClasses);
            uint64_t stack_words = 0;
            for (int c = 0; c < kNumClasses; ++c) {
                const uint64_t fit = std::min<uint64_t>((dyn >> (c + kUnitLog)) + 1, 1ull << 20);
                meta[c] = (uint32_t)stack_words; meta[kNumClasses + c] = (uint32_t)fit;
                stack_words += fit;
            }
            const size_t meta_words = 2 + 2 * kNumClasses /*lock, cnt*/ + 14 /*stat (u64 x 7)*/ + 2 * kNumClasses /*meta*/;
            if (ar.meta.bytes < (meta_words + stack_words) * 4 + 64) ar.meta.alloc((meta_words + stack_words) * 4 + 64, &ctx->live_bytes, &ctx->peak_bytes);
            {
                // layout (32-bit words): [bump u64][stat u64 x 7][lock NC][cnt NC][meta 2 NC][stacks]
                uint32_t *w = ar.meta.as<uint32_t>();
                MGTA_HIP_CHECK(hipMemsetAsync(w, 0, (16 + 2 * kNumClasses) * 4, st));
                MGTA_HIP_CHECK(hipMemcpyAsync(w + 16 + 2 * kNumClasses, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, st));
                const unsigned long long bump0 = slots * slot_bytes;
                MGTA_HIP_CHECK(hipMemcpyAsync(w, &bump0, 8, hipMemcpyHostToDevice, st));
                a.pool.base = ar.pool.as<char>(); a.pool.bytes = slots * slot_bytes + dyn;
                a.pool.reserve_off = slots * slot_bytes + dyn; a.pool.reserve_bytes = reserve;
                a.pool.bump = reinterpret_cast<unsigned long long *>(w);
                a.pool.stat = reinterpret_cast<unsigned long long *>(w + 2);
                a.pool.lock = w + 16; a.pool.cnt = w + 16 + kNumClasses;
                a.pool.meta = w + 16 + 2 * kNumClasses;
                a.pool.stack = w + 16 + 4 * kNumClasses;
            }
            a.base_off = 0; a.slot_bytes = slot_bytes; a.log_b0 = log_b0;
            // admission: no new search starts while half of the pool is in use.  (A third, while the arrays still doubled: what is in flight
            // goes on growing, the cold searches of a batch's first minute a hundredfold.  With pages an overcommitted pool only makes
            // searches wait for the next page that comes back, and a third cost the 2 M-read run 4 of its 21 s.)
            // No new search starts while more than this of the pool is in use: the searches that run keep room to grow.  HALF of the pool, and it
            // stays half: round 6 measured other values in the ordered mode where admission is memory-bound (profiles/r06/soft_limit/, the same
            // contigs throughout).  nirK on the multi-k graph of 50 M reads (2.64 M seeds): 50 % 166.0 s, 67 % 149.9 s, 80 % 142.5 s, 90 % 154.8 s
            // (searches starve and wait), 100 % > 400 s (they take each other's room in turns) -- but at 100 M reads (5.14 M seeds, the pool no
            // larger) 75 % is already beyond the cliff: nirK had not ended after 900 s where it takes 391 s with half.  Where the cliff lies
            // depends on how much the searches in flight still have to grow, which nothing knows when they are admitted.
            // MGTA_ASTAR_SOFT_PCT / _DIV: experiments.
            a.pool.soft_limit = dyn / 2;
            if (const char *e = getenv("MGTA_ASTAR_SOFT_DIV")) a.pool.soft_limit = dyn / (uint64_t)std::max(1, atoi(e));   // (experiments)
            if (const char *e = getenv("MGTA_ASTAR_SOFT_PCT")) a.pool.soft_limit = dyn / 100 * (uint64_t)std::min(100, std::max(1, atoi(e)));   // (experiments: per cent of the pool)
            a.gate = gated;
            a.free_share = free_share;
            a.active_slots = attempt == 3 ? 1u : (uint32_t)spb;
            a.ramp_base = (uint32_t)std::max<uint64_t>(64, slots / 16);              // an eighth of a direction's slots
            // The order is HELD by default, whatever it costs (advisor r4: giving it up silently made the contigs of large inputs depend on timing and
            // on the rank count).  MEGAGTA_SEARCH_ALLOW_UNORDERED=1 opts into the last resort: a batch whose searches in flight have outgrown the
            // pool (thousands of refused requests) goes on WITHOUT the order, says so on stderr and in mgta_astar_stats.order_abandoned.
            a.auto_unorder = gated && getenv("MEGAGTA_SEARCH_ALLOW_UNORDERED") && atoi(getenv("MEGAGTA_SEARCH_ALLOW_UNORDERED")) ? 1 : 0;
            if (cache_mode > 0) {
                MGTA_HIP_CHECK(hipMemsetAsync(d_start_limit.p, 0, 128, st));           // (limits are recomputed: a conservative restart of the gate)
                if (ST.order_abandoned) {                                            // (a batch that gave its order up resumes without one)
                    const unsigned long long one = 1;
                    MGTA_HIP_CHECK(hipMemcpyAsync(d_start_limit.as<unsigned long long>() + 14, &one, 8, hipMemcpyHostToDevice, st));
                }
                d_run_seed.alloc(slots * 8); d_run_progress.alloc(slots * 8);
                MGTA_HIP_CHECK(hipMemsetAsync(d_run_seed.p, 0xFF, slots * 8, st));
                MGTA_HIP_CHECK(hipMemsetAsync(d_run_progress.p, 0, slots * 8, st));
                a.run_seed = d_run_seed.as<long long>(); a.run_progress = d_run_progress.as<unsigned long long>(); a.n_slots = (uint32_t)slots;
            }
            MGTA_HIP_CHECK(hipMemsetAsync(d_queue.p, 0, 16, st));
            for (int d = 0; d < 2; ++d) {
                d_todo[d].alloc(std::max<size_t>(1, todo[d].size()) * 8);
                if (!todo[d].empty()) MGTA_HIP_CHECK(hipMemcpyAsync(d_todo[d].p, todo[d].data(), todo[d].size() * 8, hipMemcpyHostToDevice, st));
                a.todo[d] = d_todo[d].as<int64_t>(); a.n_todo[d] = (int64_t)todo[d].size();
            }
            MGTA_HIP_CHECK(hipEventRecord(ev.e[2], st));
            MGTA_HIP_CHECK(hipMemsetAsync(d_tmark.p, 0xFF, 32, st));
            if (G == 8) launch_astar<8>(a, (int)blocks, lds_bytes, use_lds, st);
#ifndef MGTA_ASTAR_G8_ONLY                                                     /* (experiment builds: one instantiation compiles in a quarter of the time) */
            else if (G == 16) launch_astar<16>(a, (int)blocks, lds_bytes, use_lds, st);
            else if (G == 32) launch_astar<32>(a, (int)blocks, lds_bytes, use_lds, st);
            else launch_astar<64>(a, (int)blocks, lds_bytes, use_lds, st);
#else
            else { set_error("this experiment build holds the 8-lane kernel only (MGTA_ASTAR_GROUP=8)"); return MGTA_EUNSUPPORTED; }
#endif
            MGTA_HIP_CHECK(hipEventRecord(ev.e[3], st));
            // MGTA_ASTAR_MONITOR=<seconds>: while the launch runs, a line on stderr every so often -- where the queues are, how many slots
            // hold a search, the lowest running seed and how far it is, the memory in use.  Copies on a stream of their own: the words are
            // written by atomics performed at the memory side, so what the copy engine reads is recent.
            if (const char *me = getenv("MGTA_ASTAR_MONITOR")) {
                const double every = std::max(1.0, atof(me));
                hipStream_t ms = nullptr;
                MGTA_HIP_CHECK(hipStreamCreateWithFlags(&ms, hipStreamNonBlocking));
                std::vector<long long> h_rs(cache_mode > 0 ? slots : 0);
                std::vector<unsigned long long> h_rp(cache_mode > 0 ? slots : 0);
                const auto t_start = std::chrono::steady_clock::now();
                double next = every;
                while (hipEventQuery(ev.e[3]) == hipErrorNotReady) {
                    std::this_thread::sleep_for(std::chrono::milliseconds(50));
                    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
                    if (el < next) continue;
                    next += every;
                    unsigned long long q[2] = {0, 0}, pl[8] = {0}, lim[16] = {0};
                    (void)hipMemcpyAsync(q, d_queue.p, 16, hipMemcpyDeviceToHost, ms);
                    (void)hipMemcpyAsync(pl, ar.meta.p, 64, hipMemcpyDeviceToHost, ms);
                    if (cache_mode > 0) {
                        (void)hipMemcpyAsync(lim, d_start_limit.p, 128, hipMemcpyDeviceToHost, ms);
                        (void)hipMemcpyAsync(h_rs.data(), d_run_seed.p, slots * 8, hipMemcpyDeviceToHost, ms);
                        (void)hipMemcpyAsync(h_rp.data(), d_run_progress.p, slots * 8, hipMemcpyDeviceToHost, ms);
                    }
                    (void)hipStreamSynchronize(ms);
                    long long lo = -1, lo_dir = 0, busy = 0, big = -1, big_dir = 0;
                    unsigned long long lo_prog = 0, big_prog = 0;
                    for (size_t sl = 0; sl < h_rs.size(); ++sl)
                        if (h_rs[sl] >= 0) {
                            ++busy;
                            const long long d = sl >= (size_t)a.blocks_dir0 * (size_t)spb ? 1 : 0;
                            if (lo < 0 || h_rs[sl] * 2 + d < lo * 2 + lo_dir) { lo = h_rs[sl]; lo_dir = d; lo_prog = h_rp[sl]; }
                            if (h_rp[sl] > big_prog) { big = h_rs[sl]; big_dir = d; big_prog = h_rp[sl]; }
                        }
                    if (big >= 0)
                        fprintf(stderr, "[astar]          longest running search: seed %lld (direction %lld, start state %d) at >= %llu expansions, k-mer %.45s\n", big, big_dir,
                                start_state[big], big_prog, kmers + (size_t)big * klen);
                    const long long q0 = (long long)std::min<unsigned long long>(q[0], todo[0].size()), q1 = (long long)std::min<unsigned long long>(q[1], todo[1].size());
                    fprintf(stderr, "[astar] %6.0f s: seeds taken %lld + %lld of %zu + %zu; %lld slots hold a search; lowest running seed %lld (direction %lld) at >= %llu "
                            "expansions%s%.45s; start limits %llu / %llu; pool %.1f GB in use (%.1f GB handed out once), reserve %.2f GB in use, owner %lld; %llu in-place restarts\n",
                            el, q0, q1, todo[0].size(), todo[1].size(), busy, lo, lo_dir, lo_prog, lo >= 0 ? ", k-mer " : "", lo >= 0 ? kmers + (size_t)lo * klen : "",
                            lim[0], lim[1], pl[5] / 1e9, pl[0] / 1e9, lim[8] / 1e9, (long long)lim[10] - 1, pl[7]);
                }
                (void)hipStreamDestroy(ms);
            }
            MGTA_HIP_CHECK(hipMemcpyAsync(h_status.data(), d_status.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
            unsigned long long h_pool[8], h_lim[16];                                // bump, stat[0..6]; the gate's words and the reserve's
            uint32_t h_cnt[kNumClasses];
            memset(h_lim, 0, sizeof(h_lim));
            MGTA_HIP_CHECK(hipMemcpyAsync(h_pool, ar.meta.p, 64, hipMemcpyDeviceToHost, st));
            MGTA_HIP_CHECK(hipMemcpyAsync(h_cnt, ar.meta.as<uint32_t>() + 16 + kNumClasses, sizeof(h_cnt), hipMemcpyDeviceToHost, st));
            if (cache_mode > 0) MGTA_HIP_CHECK(hipMemcpyAsync(h_lim, d_start_limit.p, 128, hipMemcpyDeviceToHost, st));
            unsigned long long h_tmark[4] = {~0ull, ~0ull, ~0ull, ~0ull};
            MGTA_HIP_CHECK(hipMemcpyAsync(h_tmark, d_tmark.p, 32, hipMemcpyDeviceToHost, st));
            MGTA_HIP_CHECK(hipStreamSynchronize(st));
            MGTA_HIP_CHECK(hipGetLastError());
            float ms = 0;
            MGTA_HIP_CHECK(hipEventElapsedTime(&ms, ev.e[2], ev.e[3]));
            ST.ms_kernel += ms;
            if (attempt == 0 && h_tmark[0] != ~0ull) {                                // when the last seed of the batch was TAKEN: what follows is the tail
                double drained = 0;
                for (int d = 0; d < 2; ++d)
                    if (h_tmark[1 + d] != ~0ull && h_tmark[1 + d] >= h_tmark[0]) drained = std::max(drained, (double)(h_tmark[1 + d] - h_tmark[0]) * 1e-5);
                ST.ms_queue_drained = drained;
            }
            ST.n_recycled += (int64_t)h_pool[1]; ST.n_rehash += (int64_t)h_pool[3]; ST.n_grown += (int64_t)h_pool[4];
            ST.n_retries += (int64_t)h_pool[7];                                      // searches that started again in place
            ST.pool_bytes = pool_bytes;
            ST.pool_used = std::max<uint64_t>(ST.pool_used, slots * slot_bytes + h_pool[6]);   // base arenas + most ever handed out at once
            ST.reserve_bytes = reserve;
            ST.reserve_used = std::max<uint64_t>(ST.reserve_used, h_lim[9]);
            ST.n_cache_drops += (int64_t)h_lim[13];
            if (h_lim[14] && !ST.order_abandoned) {
                ST.order_abandoned = 1;
                fprintf(stderr, "[megagta_amd] search: the searches in flight outgrew their pool (%.1f GB, %llu requests refused): the batch of %lld seeds gave up the ORDER of its "
                        "cache sharing from there on -- every path is seen by every search as soon as it is found, as in the reference's multi-thread search "
                        "(search.cpp:182-189); which of several equally good paths a later seed takes depends on timing (MEGAGTA_SEARCH_ALLOW_UNORDERED=1 asked for this).\n",
                        pool_bytes / 1e9, h_pool[2], (long long)n);
            }
            if (gated && h_lim[13])
                fprintf(stderr, "[megagta_amd] search: %llu path entries found no room in the shared cache (%.1f GB per direction): later seeds may have searched "
                        "where they could have followed a path -- which ones depends on timing\n", h_lim[13], d_cache[0].bytes / 1e9);
            // a search that reached the page tables' limit is terminal at once (more memory or another pass cannot help, and the searches
            // behind it have seen nothing of it).  It is ONE seed's side: reported as a failed search (ok = 0, no extension on that side,
            // named on stderr and counted in mgta_astar_stats.n_over_limit) while the batch goes on -- a multi-k run of hours is not thrown
            // away for it (advisor r5).  MEGAGTA_SEARCH_STRICT_LIMIT=1: the batch fails with MGTA_EOVERFLOW instead.
            for (int64_t s = 0; s < n * 2; ++s)
                if (h_status[(size_t)s] == 5 && !over_limit_seen[(size_t)s]) {
                    over_limit_seen[(size_t)s] = 1;     // (named and counted once, however many passes the batch takes)
                    char msg[384];
                    snprintf(msg, sizeof(msg), "search %lld (seed %lld, %s) outgrew the library's limit of %d pages of %d KB per array (~%lld M nodes): the reference's pool has no "
                             "bound (pool_st.h:43), this build's page tables do", (long long)s, (long long)(s / 2), (s & 1) ? "left" : "right", kMaxPages,
                             1 << (kPageLog - 10), (long long)(((uint64_t)kMaxPages << (kPageLog - 6)) >> 20));
                    const char *strict = getenv("MEGAGTA_SEARCH_STRICT_LIMIT");
                    if (strict && atoi(strict) != 0) { set_error("%s", msg); return MGTA_EOVERFLOW; }
                    fprintf(stderr, "[megagta_amd] search: %s -- this side of the seed is reported as a failed search, the batch goes on\n", msg);
                    ++ST.n_over_limit;
                }
            size_t left = 0, starved_out = 0;
            for (int d = 0; d < 2; ++d) {
                std::vector<int64_t> again;
                for (int64_t s : todo[d]) {
                    const int32_t v = h_status[(size_t)s * 2 + d];
                    if (v == 2) ++starved_out;
                    if (v == 2 || (gated && v == 0)) again.push_back(s);            // (ordered: whatever has not ended runs again, in seed order)
                }
                todo[d].swap(again);
                left += todo[d].size();
            }
            const bool verbose = getenv("MGTA_ASTAR_VERBOSE") != nullptr;
            if (verbose || (gated && left)) {
                std::string lists;
                for (int c = 0; c < kNumClasses; ++c)
                    if (h_cnt[c]) { char b[48]; snprintf(b, sizeof(b), " %u x %s", h_cnt[c], c + kUnitLog >= 30 ? (std::to_string(1u << (c + kUnitLog - 30)) + " GB").c_str() : c + kUnitLog >= 20 ? (std::to_string(1u << (c + kUnitLog - 20)) + " MB").c_str() : (std::to_string(1u << (c + kUnitLog - 10)) + " KB").c_str()); lists += b; }
                fprintf(stderr, "[astar] pass %d: %lld workgroups, pool %.1f GB (reserve %.1f GB, %.1f GB of it used), handed out once %.1f GB, most in use %.1f GB, "
                        "%llu chunks reused, %llu requests refused, %llu searches started again in place, %.0f ms, %zu searches to run again; free lists at the end:%s\n",
                        attempt, (long long)blocks, pool_bytes / 1e9, reserve / 1e9, h_lim[9] / 1e9, h_pool[0] / 1e9, (slots * slot_bytes + h_pool[6]) / 1e9, h_pool[1],
                        h_pool[2], h_pool[7], ms, left, lists.empty() ? " none" : lists.c_str());
            }
            if (gated && left) {
                ++ST.n_resumes;
                fprintf(stderr, "[megagta_amd] search: %zu search(es) found no memory even as the lowest running seed with a reserve of %.1f GB (pool %.1f GB); "
                        "the batch of %lld seeds resumes behind its commit frontier (%zu searches left) with a larger reserve\n", starved_out, reserve / 1e9,
                        pool_bytes / 1e9, (long long)n, left);
                for (int d = 0; d < 2; ++d)                                          // (status 2 -> 0: a search that is cut off again must not look starved)
                    for (int64_t s : todo[d]) h_status[(size_t)s * 2 + d] = 0;
                MGTA_HIP_CHECK(hipMemcpyAsync(d_status.p, h_status.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
            } else {
                ST.n_retries += (int64_t)left;                                       // independent searches run again by the host
            }
            MGTA_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        }
        if (!todo[0].empty() || !todo[1].empty()) {
            set_error("%zu searches do not fit the device memory left for them (pool of %llu bytes)", todo[0].size() + todo[1].size(),
                      (unsigned long long)ST.pool_bytes);
            return MGTA_EOVERFLOW;
        }
        for (int64_t s = 0; s < n * 2; ++s)
            if (h_status[(size_t)s] == 4 || h_status[(size_t)s] == 0) {
                set_error("search %lld did not run (ordered-commit gate timed out)", (long long)s);
                return MGTA_EHIP;
            }
#ifdef MGTA_ASTAR_PROFILE
        {
            unsigned long long hp[16];
            MGTA_HIP_CHECK(hipMemcpy(hp, d_prof.p, 128, hipMemcpyDeviceToHost));
            // the clock the counts are in (s_memtime ticks per 10 ns of s_memrealtime); [3..8] are sums over the SEARCHES that expanded (per
            // expansion of one search), the rest per wave (see PROF_DECL in astar_kernel.hpp)
            const double mhz = hp[12] ? 100.0 * (double)hp[13] / (double)hp[12] : 0.0, exps = (double)(hp[10] ? hp[10] : 1), iters = (double)(hp[11] ? hp[11] : 1);
            const char *nm[10] = {"fetch", "gate", "start", "pop+closed", "grow", "cache+walk", "score+probe", "commit", "(run end)", "result+free"};
            double sum = 0;
            for (int q = 3; q <= 8; ++q) sum += (double)hp[q];
            fprintf(stderr, "[astar-prof] lanes per search %d; clock of the counts %.0f MHz; %llu expansions in %llu expanding wave iterations (%.2f of %d searches expanding in each); "
                    "one expansion of one search: %.2f us\n", G, mhz, hp[10], hp[11], exps / iters, 64 / G, mhz > 0 ? sum / exps / mhz : 0.0);
            for (int q = 3; q <= 8; ++q)
                fprintf(stderr, "[astar-prof]   %-12s %6.2f %%  %8.3f us per expansion of one search\n", nm[q], 100.0 * hp[q] / (sum > 0 ? sum : 1), mhz > 0 ? (double)hp[q] / exps / mhz : 0.0);
            for (int q : {0, 1, 2, 9})
                fprintf(stderr, "[astar-prof]   %-12s %8.3f us per expanding wave iteration (wave-level)\n", nm[q], mhz > 0 ? (double)hp[q] / iters / mhz : 0.0);
            fprintf(stderr, "[astar-prof]   asleep (every running search of the wave waits for memory): %llu times, %.3f us per expanding wave iteration\n",
                    hp[15], mhz > 0 ? (double)hp[14] / iters / mhz : 0.0);
        }
#endif
        // results: records and lengths as they are, the strings packed on the device first
        std::vector<mgta_astar_side> h_sides((size_t)n * 2);
        std::vector<uint32_t> h_len((size_t)n * 2);
        std::vector<uint64_t> h_off((size_t)n * 2);
        DevBuf d_off, d_scan_tmp, d_tot, d_packed;
        d_off.alloc((size_t)n * 2 * 8); d_scan_tmp.alloc(scan_tmp_elems((uint64_t)n * 2) * 8); d_tot.alloc(64);
        exclusive_scan_u32(st, d_len.as<uint32_t>(), (uint64_t)n * 2, d_off.as<uint64_t>(), d_scan_tmp.as<uint64_t>(), d_tot.as<uint64_t>());
        uint64_t n_chars = 0;
        MGTA_HIP_CHECK(hipMemcpyAsync(&n_chars, d_tot.p, 8, hipMemcpyDeviceToHost, st));
        MGTA_HIP_CHECK(hipStreamSynchronize(st));
        d_packed.alloc(n_chars + 64);
        hipLaunchKernelGGL(pack_results_kernel, dim3((unsigned)std::min<uint64_t>(((uint64_t)n * 2 + 3) / 4, 1u << 16)), dim3(256), 0, st, d_out.as<char>(), out_cap,
                           d_len.as<uint32_t>(), d_off.as<uint64_t>(), (uint64_t)n * 2, d_packed.as<char>());
        MGTA_HIP_CHECK(hipGetLastError());
        std::vector<char> h_out(n_chars + 1);
        MGTA_HIP_CHECK(hipMemcpyAsync(h_sides.data(), d_sides.p, h_sides.size() * sizeof(mgta_astar_side), hipMemcpyDeviceToHost, st));
        MGTA_HIP_CHECK(hipMemcpyAsync(h_len.data(), d_len.p, h_len.size() * 4, hipMemcpyDeviceToHost, st));
        MGTA_HIP_CHECK(hipMemcpyAsync(h_off.data(), d_off.p, h_off.size() * 8, hipMemcpyDeviceToHost, st));
        if (n_chars) MGTA_HIP_CHECK(hipMemcpyAsync(h_out.data(), d_packed.p, n_chars, hipMemcpyDeviceToHost, st));
        MGTA_HIP_CHECK(hipEventRecord(ev.e[1], st));
        MGTA_HIP_CHECK(hipStreamSynchronize(st));
        float ms = 0;
        MGTA_HIP_CHECK(hipEventElapsedTime(&ms, ev.e[0], ev.e[1]));
        ST.ms_total = ms;
        uint32_t max_nodes = 0, max_exp = 0;
        for (int64_t s = 0; s < n * 2; ++s) {
            if (h_status[(size_t)s] == 3) {
                set_error("seed %lld: k-mer / model position outside the model (start_state %d)", (long long)(s / 2), start_state[s / 2]);
                return MGTA_EINVAL;
            }
            ST.n_expansions += h_sides[(size_t)s].n_expanded;
            ST.n_opened += h_sides[(size_t)s].n_opened;
            max_nodes = std::max<uint32_t>(max_nodes, (uint32_t)h_sides[(size_t)s].n_opened);
            max_exp = std::max<uint32_t>(max_exp, (uint32_t)h_sides[(size_t)s].n_expanded);
        }
        ST.max_search_nodes = max_nodes; ST.max_search_expansions = max_exp;
        auto rev_comp = [](char *dst, const char *l, uint32_t ll) {                // RevComp, hmm_graph_search.h:362-398
            for (uint32_t i = 0; i < ll; ++i) {
                const char c = l[ll - 1 - i];
                dst[i] = c == 'a' ? 't' : c == 'c' ? 'g' : c == 'g' ? 'c' : c == 't' ? 'a' : c;
            }
        };
        if (packed) {
            // contig i = left + lower-cased seed k-mer + right (hmm_graph_search.h:60-81), written once into its final place
            const uint64_t total = n_chars + (uint64_t)n * (uint64_t)klen;
            char *buf = static_cast<char *>(malloc(total + 1));
            if (!buf) { set_error("mgta_astar_batch_packed: out of host memory"); return MGTA_ENOMEM; }
            uint64_t at = 0;
            for (int64_t s = 0; s < n; ++s) {
                packed->offsets[s] = at;
                const uint32_t ll = h_len[(size_t)2 * s + 1], rl = h_len[(size_t)2 * s];
                rev_comp(buf + at, h_out.data() + h_off[(size_t)2 * s + 1], ll);
                at += ll;
                const char *km = kmers + (size_t)s * klen;
                for (int j = 0; j < klen; ++j) buf[at + j] = (char)tolower((unsigned char)km[j]);   // the seed k-mer, lower case (search.cpp:156)
                at += klen;
                memcpy(buf + at, h_out.data() + h_off[(size_t)2 * s], rl);
                at += rl;
            }
            packed->offsets[n] = at;
            buf[at] = 0;
            if (packed->sides) memcpy(packed->sides, h_sides.data(), h_sides.size() * sizeof(mgta_astar_side));
            *packed->contigs = buf;
        } else if (sink) {
            std::string left;
            for (int64_t s = 0; s < n; ++s) {
                const char *r = h_out.data() + h_off[(size_t)2 * s];
                const uint32_t ll = h_len[(size_t)2 * s + 1];
                left.assign(ll, ' ');
                rev_comp(&left[0], h_out.data() + h_off[(size_t)2 * s + 1], ll);
                int src = sink(user, s, left.data(), (int64_t)ll, r, (int64_t)h_len[(size_t)2 * s], &h_sides[(size_t)2 * s],
                               &h_sides[(size_t)2 * s + 1]);
                if (src != 0) { set_error("contig sink returned %d", src); return MGTA_ESINK; }
            }
        }
        if (stats) *stats = ST;
        return MGTA_OK;
    } catch (const HipError &e) { return e.code; }
}
}  // namespace

extern "C" {
// The same batch with the results in flat arrays instead of one call-back per seed (a Python caller pays microseconds per call-back:
// minutes at millions of seeds): contig i = (*contigs)[offsets[i] .. offsets[i + 1]) = left + lower-cased k-mer + right, exactly the
// sequence line `search` writes (hmm_graph_search.h:60-81).
int mgta_astar_batch_packed(mgta_sdbg *g, const mgta_hmm *fwd, const mgta_hmm *rev, const char *kmers, const int32_t *start_state, int64_t n,
                            int prune_len, double low_cov_penalty, int cache_mode, char **contigs, uint64_t *offsets, mgta_astar_side *sides,
                            mgta_astar_stats *stats) {
    if (!g || !contigs || !offsets) { set_error("mgta_astar_batch_packed: bad argument"); return MGTA_EINVAL; }
    *contigs = nullptr;
    const PackedOut po{contigs, offsets, sides};
    try {
        if (n == 0) {
            char *buf = static_cast<char *>(malloc(1));
            if (!buf) { set_error("mgta_astar_batch_packed: out of host memory"); return MGTA_ENOMEM; }
            buf[0] = 0; offsets[0] = 0; *contigs = buf;
        }
        const int rc = astar_batch_impl(g->ctx, g, fwd, rev, kmers, start_state, n, prune_len, low_cov_penalty, cache_mode, nullptr, nullptr, stats, &po);
        if (rc != MGTA_OK && *contigs) { free(*contigs); *contigs = nullptr; }   // (a call that fails hands nothing over: the caller frees only what MGTA_OK gave it)
        return rc;
    } catch (const std::bad_alloc &) { set_error("mgta_astar_batch_packed: out of host memory"); }
      catch (const std::exception &e) { set_error("mgta_astar_batch_packed: %s", e.what()); if (*contigs) { free(*contigs); *contigs = nullptr; } return MGTA_EHIP; }
    if (*contigs) { free(*contigs); *contigs = nullptr; }
    return MGTA_ENOMEM;
}

}  // extern "C"
