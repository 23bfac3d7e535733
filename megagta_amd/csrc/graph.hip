// graph.hip — build the device-resident succinct de Bruijn graph from the logical edge stream and
// answer batched navigation queries (test hooks of the C ABI).
//
// Replaces SuccinctDBG::LoadFromMultiFile + init (succinct_dbg.cpp:595-723, succinct_dbg.h:62-86)
// and RankAndSelect{4Bits,1Bit}::Build (rank_and_select.h:80-150,430-500).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <memory>
#include <thread>
#include <vector>

#include "common.hpp"
#include "device_utils.hpp"
#include "graph.hpp"
#include "scan.hpp"

namespace mgta {

// G1: pack 64 records into a line, count ones per line: cnt[c*n_lines + line], c = 0 last, 1 tip, 2..5 symbols 1..4
// (the lines [line_lo, line_hi) from the records recs[0 ...) = records rec_base ...: a graph too large to hold its records AND its lines at once
// is packed range by range, mgta_sdbg_load_files)
__global__ __launch_bounds__(256) void graph_pack_kernel(const uint16_t *recs, int64_t rec_base, int64_t size, GLine *lines, uint64_t n_lines,
                                                         uint64_t line_lo, uint64_t line_hi, uint32_t *cnt) {
    // one wave per line, one lane per edge; the grid is capped (a dispatch holds < 2^32 work-items: 6.3 G edges do not fit one lane each)
    const int lane = lane_id();
    for (uint64_t li = line_lo + (uint64_t)blockIdx.x * 4 + wave_id(); li < line_hi; li += (uint64_t)gridDim.x * 4) {
    int64_t e = (int64_t)(li << 6) + lane;
    uint32_t it = e < size ? recs[e - rec_base] : 0;
    bool in = e < size;
    uint32_t w = it & 15;
    uint64_t b_last = __ballot(in && ((it >> 4) & 1));
    uint64_t b_tip = __ballot(in && ((it >> 5) & 1));
    uint64_t b_inv = __ballot(in && (((it >> 5) & 1) || w == 0));
    uint64_t b_m1 = __ballot(in && ((it >> 8) <= 1));
    uint64_t sym[4];
#pragma unroll
    for (int a = 1; a <= 4; ++a) sym[a - 1] = __ballot(in && w == (uint32_t)a);
    // W nibbles: lane j contributes w << 4*(j&15) to word j>>4 : OR-reduce inside each group of 16 lanes
    uint64_t nib = (uint64_t)w << ((lane & 15) * 4);
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) nib |= __shfl_xor(nib, d, 64);
    uint64_t w0 = __shfl(nib, 0, 64), w1 = __shfl(nib, 16, 64), w2 = __shfl(nib, 32, 64), w3 = __shfl(nib, 48, 64);
    if (lane == 0) {
        GLine L;
        L.w[0] = w0; L.w[1] = w1; L.w[2] = w2; L.w[3] = w3;
        L.last = b_last; L.tip = b_tip; L.invalid = b_inv; L.multi1 = b_m1;
        L.rank_last = 0; L.rank_tip = 0;
        L.rank_w[0] = L.rank_w[1] = L.rank_w[2] = L.rank_w[3] = 0;
        L.fwd_hint[0] = L.fwd_hint[1] = L.fwd_hint[2] = L.fwd_hint[3] = 0;
        lines[li] = L;
        cnt[0 * n_lines + li] = (uint32_t)__popcll(b_last);
        cnt[1 * n_lines + li] = (uint32_t)__popcll(b_tip);
#pragma unroll
        for (int a = 0; a < 4; ++a) cnt[(2 + a) * n_lines + li] = (uint32_t)__popcll(sym[a]);
    }
    }
}

// G3: write the absolute ranks into the lines and fill the select samples
__global__ __launch_bounds__(256) void graph_rank_kernel(GLine *lines, uint64_t n_lines, const uint32_t *cnt, const uint64_t *base,
                                                         uint32_t *sel_last, uint32_t *sel_w1, uint32_t *sel_w2, uint32_t *sel_w3,
                                                         uint32_t *sel_w4) {
    uint64_t li = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (li >= n_lines) return;
    uint32_t *sel[5] = {sel_last, sel_w1, sel_w2, sel_w3, sel_w4};
    uint64_t b[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) b[c] = base[c * n_lines + li];
    lines[li].rank_last = b[0];
    lines[li].rank_tip = b[1];
#pragma unroll
    for (int a = 0; a < 4; ++a) lines[li].rank_w[a] = b[2 + a];
    // ranks [b, b+c) live in this line: every multiple of 64 among them gets this line as its sample
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        int c = s == 0 ? 0 : s + 1;
        uint64_t lo = b[c], hi = lo + cnt[c * n_lines + li];
        for (uint64_t m = (lo + 63) & ~63ull; m < hi; m += 64) sel[s][m >> 6] = (uint32_t)li;
    }
}

// G4: forward hints (needs rank_f): line of Select(rank_f[a] + #a before this line - 1).  One rank early on purpose: an edge with
// W = a + 4 at the start of the line, before any plain a, forwards to the target of the last plain a of an EARLIER line (Forward counts
// plain symbols only, succinct_dbg.h:155-164), and the look-up only ever walks forwards from the hint.
__global__ __launch_bounds__(256) void graph_hint_kernel(GraphDev g, GLine *lines) {
    uint64_t li = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (li >= g.n_lines) return;
#pragma unroll
    for (int a = 1; a <= 4; ++a) {
        int64_t r0 = g.rank_f[a] + (int64_t)lines[li].rank_w[a - 1] - 1;
        uint64_t h = g.n_lines - 1;
        if (r0 < g.total_last) {
            if (r0 < 0) r0 = 0;
            h = g.sel_last[r0 >> 6];
            while (h + 1 < g.n_lines && (int64_t)lines[h + 1].rank_last <= r0) ++h;
        }
        lines[li].fwd_hint[a - 1] = (uint32_t)h;
    }
}

// PREFIX.sdbg.N -> logical edge stream (SdbgReader::NextItem, sdbg_multi_io.h:335-382).  A bucket's records are variable-length (a
// record word, then the full multiplicity when the stored one is 255, then words_per_tip label words when it is a tip), so a bucket is
// parsed front to back; the 65536 buckets are independent and their places in the output are known from the index file: one lane per
// bucket.  `bytes` = a piece of one file, already on the device; every size was checked against the file by the host.
struct BucketSrc {
    uint64_t src;        // byte offset of the bucket inside `bytes` (even)
    uint64_t n_bytes;    // 2 items + 2 large + 4 words_per_tip tips
    int64_t items, rec_out, tip_out;   // records, index of its first record, index of its first tip
};
__global__ __launch_bounds__(64) void sdbg_decode_kernel(const uint16_t *bytes, const BucketSrc *buckets, uint32_t n, int words_per_tip, uint16_t *recs,
                                                         uint32_t *tips, uint32_t *bad) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const BucketSrc b = buckets[i];
    const uint16_t *p = bytes + (b.src >> 1), *end = p + (b.n_bytes >> 1);
    uint16_t *out = recs + b.rec_out;
    uint32_t *tp = tips + b.tip_out * words_per_tip;
    bool ok = true;
    for (int64_t r = 0; r < b.items; ++r) {
        if (p >= end) { ok = false; break; }
        const uint32_t it = *p++;
        out[r] = (uint16_t)it;
        if ((it >> 8) == 255u) ++p;                                      // the full multiplicity: not part of the graph (need_multiplicity = false)
        if ((it >> 5) & 1u) {
            if (p + 2 * words_per_tip > end) { ok = false; break; }
            for (int w = 0; w < words_per_tip; ++w) { *tp++ = (uint32_t)p[0] | ((uint32_t)p[1] << 16); p += 2; }
        }
    }
    if (!ok || p != end) atomicAdd(bad, 1u);
}

__global__ void graph_rankf_kernel(GraphDev g, int64_t *rank_f) {
    if (threadIdx.x < 6) rank_f[threadIdx.x] = g_rank_last(g, g.f[threadIdx.x] - 1);
}

__global__ __launch_bounds__(256) void graph_outgoing_kernel(GraphDev g, const int64_t *edges, int64_t n, int64_t *out4, int8_t *outdeg) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int64_t o[4] = {-1, -1, -1, -1};
    int od = g_outgoing(g, edges[i], o);
    outdeg[i] = (int8_t)od;
    for (int j = 0; j < 4; ++j) out4[i * 4 + j] = (j < od) ? (o[j] >> 4) : -1;
}

__global__ __launch_bounds__(64) void graph_index_kernel(GraphDev g, const uint8_t *seqs, int64_t n, int64_t *ids) {
    int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint8_t *s = seqs + i * (g.k + 1);
    bool ok = true;
    for (int j = 0; j <= g.k; ++j) ok = ok && s[j] >= 1 && s[j] <= 4;
    ids[i] = ok ? g_index_edge(g, s) : -1;
}

}  // namespace mgta

using namespace mgta;

__global__ __launch_bounds__(256) void graph_invalid_kernel(const GLine *lines, uint64_t n_lines, uint64_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_lines) out[i] = lines[i].invalid;
}

extern "C" {

// A graph under construction: graph_begin() allocates the lines, the per-line counts and the tip labels, graph_pack() packs a range of lines
// from records on the device, graph_finish() builds the rank / select tables.  (succinct_dbg.cpp:595-723, rank_and_select.h:80,430)
struct GraphBuild {
    std::unique_ptr<mgta_sdbg> g;
    DevBuf d_cnt;
    uint64_t n_lines = 0;
    int64_t size = 0;
};
// tips: the labels (host or device memory, `tips_on_device`), or null = the caller fills g->tips itself before graph_finish()
// adopt_lines: a device buffer that holds the RECORDS and is large enough for the lines: the lines are then packed IN PLACE (a GLine is exactly
// as large as the 64 two-byte records it is made of, and a wavefront reads its 64 records before it stores its line)
static int graph_begin(mgta_ctx *ctx, int k, int64_t size, const int64_t *bucket_items, const uint32_t *tips, int64_t n_tip_words, int words_per_tip,
                       bool tips_on_device, GraphBuild &B, DevBuf *adopt_lines = nullptr) {
    MGTA_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    B.g = std::make_unique<mgta_sdbg>();
    auto &g = B.g;
    g->ctx = ctx;
    GraphDev &d = g->dev;
    memset(&d, 0, sizeof(d));
    d.size = size; d.k = k; d.words_per_tip = words_per_tip;
    d.f[0] = -1; d.f[1] = 0;                                        // sdbg_multi_io.h:254-268
    int64_t acc = 0;
    for (int b = 0; b < MGTA_NUM_BUCKETS; ++b) { acc += bucket_items[b]; d.f[b / (MGTA_NUM_BUCKETS / 4) + 2] = acc; }
    if (acc != size) { set_error("mgta_sdbg_load: bucket_items sum %lld != size %lld", (long long)acc, (long long)size); return MGTA_EINVAL; }
    const uint64_t n_lines = (uint64_t)((size + 63) / 64);
    if (n_lines >= 0xFFFFFFFFull) { set_error("graph too large for 32-bit line samples"); return MGTA_EUNSUPPORTED; }
    d.n_lines = n_lines;
    B.n_lines = n_lines; B.size = size;
    if (!adopt_lines) {
        g->lines.alloc((n_lines + 1) * sizeof(GLine), &ctx->live_bytes, &ctx->peak_bytes);
        MGTA_HIP_CHECK(hipMemsetAsync(g->lines.p, 0, (n_lines + 1) * sizeof(GLine), st));
    }
    g->tips.alloc((size_t)n_tip_words * 4 + 16, &ctx->live_bytes, &ctx->peak_bytes);
    if (n_tip_words && tips)
        MGTA_HIP_CHECK(hipMemcpyAsync(g->tips.p, tips, (size_t)n_tip_words * 4, tips_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    if (size > 0) B.d_cnt.alloc(n_lines * 6 * 4, &ctx->live_bytes, &ctx->peak_bytes);
    // the caller's buffer changes hands only once nothing in here can fail any more: an allocation that throws above leaves the stream where
    // it was, still valid (advisor r5)
    if (adopt_lines) g->lines = std::move(*adopt_lines);
    d.lines = g->lines.as<GLine>();
    d.tip_labels = g->tips.as<uint32_t>();
    return MGTA_OK;
}
static void graph_pack(GraphBuild &B, const uint16_t *dev_recs, int64_t rec_base, uint64_t line_lo, uint64_t line_hi) {
    if (line_hi <= line_lo) return;
    mgta_ctx *ctx = B.g->ctx;
    hipLaunchKernelGGL(graph_pack_kernel, dim3((unsigned)std::min<uint64_t>((line_hi - line_lo + 3) / 4, 1u << 22)), dim3(256), 0, ctx->stream, dev_recs, rec_base,
                       B.size, B.g->lines.as<GLine>(), B.n_lines, line_lo, line_hi, B.d_cnt.as<uint32_t>());
    MGTA_HIP_CHECK(hipGetLastError());
}
static int graph_finish(GraphBuild &B, mgta_sdbg **out) {
    auto &g = B.g;
    mgta_ctx *ctx = g->ctx;
    hipStream_t st = ctx->stream;
    GraphDev &d = g->dev;
    const uint64_t n_lines = B.n_lines;
    if (B.size > 0) {
        DevBuf d_base, d_tmp, d_tot;
        d_base.alloc(n_lines * 6 * 8, &ctx->live_bytes, &ctx->peak_bytes);
        d_tmp.alloc(scan_tmp_elems(n_lines) * 8, &ctx->live_bytes, &ctx->peak_bytes);
        d_tot.alloc(64, &ctx->live_bytes, &ctx->peak_bytes);
        uint64_t tot[6];
        for (int c = 0; c < 6; ++c)
            exclusive_scan_u32(st, B.d_cnt.as<uint32_t>() + c * n_lines, n_lines, d_base.as<uint64_t>() + c * n_lines, d_tmp.as<uint64_t>(),
                               d_tot.as<uint64_t>() + c);
        MGTA_HIP_CHECK(hipMemcpyAsync(tot, d_tot.p, 48, hipMemcpyDeviceToHost, st));
        MGTA_HIP_CHECK(hipStreamSynchronize(st));
        d.total_last = (int64_t)tot[0];
        for (int a = 1; a <= 4; ++a) d.total_w[a] = (int64_t)tot[1 + a];
        g->sel_last.alloc((tot[0] / 64 + 2) * 4, &ctx->live_bytes, &ctx->peak_bytes);
        for (int a = 1; a <= 4; ++a) g->sel_w[a].alloc((tot[1 + a] / 64 + 2) * 4, &ctx->live_bytes, &ctx->peak_bytes);
        d.sel_last = g->sel_last.as<uint32_t>();
        for (int a = 1; a <= 4; ++a) d.sel_w[a] = g->sel_w[a].as<uint32_t>();
        hipLaunchKernelGGL(graph_rank_kernel, dim3((unsigned)((n_lines + 255) / 256)), dim3(256), 0, st, g->lines.as<GLine>(), n_lines,
                           B.d_cnt.as<uint32_t>(), d_base.as<uint64_t>(), g->sel_last.as<uint32_t>(), g->sel_w[1].as<uint32_t>(),
                           g->sel_w[2].as<uint32_t>(), g->sel_w[3].as<uint32_t>(), g->sel_w[4].as<uint32_t>());
        DevBuf d_rf;
        d_rf.alloc(64, &ctx->live_bytes, &ctx->peak_bytes);
        hipLaunchKernelGGL(graph_rankf_kernel, dim3(1), dim3(64), 0, st, d, d_rf.as<int64_t>());
        MGTA_HIP_CHECK(hipMemcpyAsync(d.rank_f, d_rf.p, 48, hipMemcpyDeviceToHost, st));
        MGTA_HIP_CHECK(hipStreamSynchronize(st));
        hipLaunchKernelGGL(graph_hint_kernel, dim3((unsigned)((n_lines + 255) / 256)), dim3(256), 0, st, d, g->lines.as<GLine>());
        MGTA_HIP_CHECK(hipGetLastError());
        MGTA_HIP_CHECK(hipStreamSynchronize(st));
    }
    B.d_cnt.release();
    ctx_retain(ctx);
    *out = g.release();
    return MGTA_OK;
}

// records / tip labels in host memory (`resident` false) or still on the device where the build left them (true: no copy of the records)
// (`recs_owner`: a device buffer of the caller that holds `recs` and nothing else -- released as soon as the lines are packed, before
// the rank tables are built: at 2 bytes per edge it is as large as the graph itself)
static int load_graph(mgta_ctx *ctx, int k, const uint16_t *recs, int64_t size, const int64_t *bucket_items, const uint32_t *tips,
                      int64_t n_tip_words, int words_per_tip, bool resident, mgta_sdbg **out, DevBuf *recs_owner = nullptr) {
    try {
        GraphBuild B;
        const int rc = graph_begin(ctx, k, size, bucket_items, tips, n_tip_words, words_per_tip, resident, B);
        if (rc != MGTA_OK) return rc;
        hipStream_t st = ctx->stream;
        if (size > 0) {
            DevBuf d_recs;
            const uint16_t *dev_recs = recs;
            if (!resident) {
                d_recs.alloc((size_t)size * 2, &ctx->live_bytes, &ctx->peak_bytes);
                MGTA_HIP_CHECK(hipMemcpyAsync(d_recs.p, recs, (size_t)size * 2, hipMemcpyHostToDevice, st));
                dev_recs = d_recs.as<uint16_t>();
            }
            graph_pack(B, dev_recs, 0, 0, B.n_lines);
            if (!resident || recs_owner) {                                   // the records have done their part
                MGTA_HIP_CHECK(hipStreamSynchronize(st));
                d_recs.release();
                if (recs_owner) recs_owner->release();
            }
        }
        return graph_finish(B, out);
    } catch (const HipError &e) { return e.code; }
}

int mgta_sdbg_load(mgta_ctx *ctx, int k, const uint16_t *recs, int64_t size, const int64_t *bucket_items, const uint32_t *tips,
                   int64_t n_tip_words, int words_per_tip, mgta_sdbg **out) {
    if (!ctx || !out || size < 0 || (size > 0 && !recs) || !bucket_items) { set_error("mgta_sdbg_load: bad argument"); return MGTA_EINVAL; }
    return load_graph(ctx, k, recs, size, bucket_items, tips, n_tip_words, words_per_tip, false, out);
}

// SuccinctDBG::LoadFromMultiFile (succinct_dbg.cpp:595-723) from the files themselves: the host reads the index, maps the record files
// and copies them to the device piece by piece (pinned staging, several copy threads); the records are parsed THERE.
int mgta_sdbg_load_files(mgta_ctx *ctx, const char *prefix_c, mgta_sdbg **out) {
    if (!ctx || !prefix_c || !out) { set_error("mgta_sdbg_load_files: bad argument"); return MGTA_EINVAL; }
    const std::string prefix = prefix_c;
    struct Fd { int fd = -1; const unsigned char *map = nullptr; size_t size = 0; };
    std::vector<Fd> files;
    struct Cleanup {
        std::vector<Fd> &f; void *pin[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr};
        ~Cleanup() { for (Fd &x : f) { if (x.map && x.size) munmap(const_cast<unsigned char *>(x.map), x.size); if (x.fd >= 0) close(x.fd); }
                     for (void *q : pin) if (q) (void)hipHostFree(q);
                     for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
    } cleanup{files};
    try {
        FILE *info = fopen((prefix + ".sdbg_info").c_str(), "r");
        if (!info) { set_error("cannot open %s.sdbg_info", prefix.c_str()); return MGTA_EINVAL; }
        int k = 0, wpt = 0, nb = 0, nf = 0;
        long long total = 0, ntips = 0, nlarge = 0;
        struct Line { int tid; long long off, items, tips, large; };
        std::vector<Line> bl(MGTA_NUM_BUCKETS);
        bool good = fscanf(info, "k %d\n", &k) == 1 && fscanf(info, "words_per_tip_label %d\n", &wpt) == 1 && fscanf(info, "num_buckets %d\n", &nb) == 1 &&
                    fscanf(info, "num_threads %d\n", &nf) == 1 && fscanf(info, "total_size %lld\n", &total) == 1 && fscanf(info, "num_tips %lld\n", &ntips) == 1 &&
                    fscanf(info, "large_multi %lld\n", &nlarge) == 1 && nb == MGTA_NUM_BUCKETS && nf >= 1 && nf <= 65536 && k >= 1 && wpt == (2 * k + 31) / 32 &&
                    total >= 0 && ntips >= 0;
        for (int b = 0; good && b < nb; ++b) {
            int id = -1;
            good = fscanf(info, "%d %d %lld %lld %lld %lld\n", &id, &bl[b].tid, &bl[b].off, &bl[b].items, &bl[b].tips, &bl[b].large) == 6 && id == b &&
                   bl[b].tid < nf && bl[b].items >= 0 && bl[b].tips >= 0 && bl[b].large >= 0 && bl[b].off >= 0 && (bl[b].off & 1) == 0;
        }
        fclose(info);
        if (!good) { set_error("%s.sdbg_info: not an SdBG index (sdbg_multi_io.h:117-187)", prefix.c_str()); return MGTA_EINVAL; }
        files.resize(nf);
        for (int t = 0; t < nf; ++t) {
            const std::string path = prefix + ".sdbg." + std::to_string(t);
            files[t].fd = open(path.c_str(), O_RDONLY);
            struct stat sb;
            if (files[t].fd < 0 || fstat(files[t].fd, &sb) != 0) { set_error("cannot open %s", path.c_str()); return MGTA_EINVAL; }
            files[t].size = (size_t)sb.st_size;
            if (files[t].size) {
                void *m = mmap(nullptr, files[t].size, PROT_READ, MAP_PRIVATE, files[t].fd, 0);
                if (m == MAP_FAILED) { set_error("cannot map %s", path.c_str()); return MGTA_EINVAL; }
                files[t].map = static_cast<const unsigned char *>(m);
                (void)madvise(m, files[t].size, MADV_SEQUENTIAL);
            }
        }
        // where every bucket's records and tips go, and the pieces to copy: the buckets of a file in offset order, cut every ~512 MB
        std::vector<int64_t> items(MGTA_NUM_BUCKETS), rec_out(MGTA_NUM_BUCKETS), tip_out(MGTA_NUM_BUCKETS);
        long long acc_r = 0, acc_t = 0;
        std::vector<std::vector<int>> of_file(nf);
        for (int b = 0; b < nb; ++b) {
            const bool has = bl[b].tid >= 0 && bl[b].items > 0;
            items[b] = has ? bl[b].items : 0;
            rec_out[b] = acc_r; tip_out[b] = acc_t;
            if (!has) continue;
            const unsigned long long nbytes = 2ull * bl[b].items + 2ull * bl[b].large + 4ull * wpt * bl[b].tips;
            if ((unsigned long long)bl[b].off + nbytes > files[bl[b].tid].size || bl[b].tips > bl[b].items || bl[b].large > bl[b].items) {
                set_error("%s.sdbg_info: bucket %d does not fit %s.sdbg.%d", prefix.c_str(), b, prefix.c_str(), bl[b].tid);
                return MGTA_EINVAL;
            }
            acc_r += bl[b].items; acc_t += bl[b].tips;
            of_file[bl[b].tid].push_back(b);
        }
        if (acc_r != total || acc_t != ntips) { set_error("%s.sdbg_info: the bucket lines hold %lld records / %lld tips, the header says %lld / %lld",
                                                          prefix.c_str(), acc_r, acc_t, total, ntips); return MGTA_EINVAL; }
        MGTA_HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        // The records never sit on the device all at once: they are decoded into a buffer of at most `range` records (MGTA_LOAD_RANGE_RECORDS;
        // 2^32 = 8 GB), the lines they fill are packed at once, the buffer serves the next range.  Round 4 held all records (2 B per edge) next
        // to the lines (2 B) and the line counts: 285 GB for the 63 G-edge graph of a 1 G-read set -- it could not be loaded at all; now the
        // peak is the graph + its rank prefix sums (3.2 B per edge) + the range buffer.  A line (64 edges) that straddles two ranges is packed
        // with the second: its first records are carried over to the front of the buffer.
        GraphBuild B;
        {
            const int rc = graph_begin(ctx, k, (int64_t)total, items.data(), nullptr, (int64_t)ntips * wpt, wpt, true, B);
            if (rc != MGTA_OK) return rc;
        }
        uint64_t range = 1ull << 32;
        if (const char *e = getenv("MGTA_LOAD_RANGE_RECORDS")) range = std::max<uint64_t>(64, strtoull(e, nullptr, 10));
        DevBuf d_range, d_piece[2], d_desc[2], d_bad;
        d_bad.alloc(64);
        MGTA_HIP_CHECK(hipMemsetAsync(d_bad.p, 0, 64, st));
        const size_t kPiece = 512ull << 20, kStage = 64ull << 20;
        for (auto &q : cleanup.pin) MGTA_HIP_CHECK(hipHostMalloc(&q, kStage, hipHostMallocDefault));
        hipEvent_t (&ev)[2] = cleanup.ev;
        for (auto &e : ev) MGTA_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        int stage = 0;
        bool used_stage[2] = {false, false};
        const unsigned n_copy = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
        auto upload = [&](const unsigned char *src, size_t n, char *dst) {       // file bytes -> device, through the two pinned buffers
            for (size_t o = 0; o < n; o += kStage) {
                const size_t m = std::min(kStage, n - o);
                if (used_stage[stage]) MGTA_HIP_CHECK(hipEventSynchronize(ev[stage]));
                char *pin = static_cast<char *>(cleanup.pin[stage]);
                std::vector<std::thread> th;
                const size_t per = (m + n_copy - 1) / n_copy;
                for (unsigned t = 1; t < n_copy && t * per < m; ++t)
                    th.emplace_back([=] { memcpy(pin + t * per, src + o + t * per, std::min(per, m - t * per)); });
                memcpy(pin, src + o, std::min(per, m));
                for (auto &x : th) x.join();
                MGTA_HIP_CHECK(hipMemcpyAsync(dst + o, pin, m, hipMemcpyHostToDevice, st));
                MGTA_HIP_CHECK(hipEventRecord(ev[stage], st));
                used_stage[stage] = true;
                stage ^= 1;
            }
        };
        for (int t = 0; t < nf; ++t) std::sort(of_file[t].begin(), of_file[t].end(), [&](int a, int b) { return bl[a].off < bl[b].off; });
        int pc = 0, n_ranges = 0;
        std::vector<BucketSrc> desc;
        int64_t carry = 0;                                                // records at the front of the buffer that belong to a line not packed yet (< 64)
        for (int b0 = 0; b0 < nb;) {
            // the buckets [b0, b1) of this range: as many as fit the buffer beside the carried records (one at least)
            const int64_t first = rec_out[b0];
            int b1 = b0;
            int64_t n_in = 0;
            while (b1 < nb && (b1 == b0 || (uint64_t)(carry + n_in + items[b1]) <= range)) { n_in += items[b1]; ++b1; }
            const int64_t rec_base = first - carry;                      // record held at the front of the buffer: a multiple of 64
            if (d_range.bytes < (size_t)(carry + n_in + 64) * 2) {
                DevBuf bigger;
                bigger.alloc((size_t)std::max<uint64_t>((uint64_t)(carry + n_in), std::min<uint64_t>(range, (uint64_t)total)) * 2 + 256, &ctx->live_bytes, &ctx->peak_bytes);
                if (carry) MGTA_HIP_CHECK(hipMemcpyAsync(bigger.p, d_range.p, (size_t)carry * 2, hipMemcpyDeviceToDevice, st));
                MGTA_HIP_CHECK(hipStreamSynchronize(st));
                d_range = std::move(bigger);
            }
            for (int t = 0; t < nf; ++t) {
                const std::vector<int> &bs = of_file[t];
                for (size_t i = 0; i < bs.size();) {
                    if (bs[i] < b0 || bs[i] >= b1) { ++i; continue; }
                    const unsigned long long lo = (unsigned long long)bl[bs[i]].off;
                    unsigned long long hi = lo;
                    desc.clear();
                    size_t j = i;
                    for (; j < bs.size(); ++j) {
                        if (bs[j] < b0 || bs[j] >= b1) break;            // (a bucket of another range: the piece ends here)
                        const Line &L = bl[bs[j]];
                        const unsigned long long nbytes = 2ull * L.items + 2ull * L.large + 4ull * wpt * L.tips;
                        if (j > i && (unsigned long long)L.off + nbytes - lo > kPiece) break;
                        hi = std::max(hi, (unsigned long long)L.off + nbytes);
                        desc.push_back(BucketSrc{(uint64_t)L.off - lo, nbytes, L.items, rec_out[bs[j]] - rec_base, tip_out[bs[j]]});
                    }
                    // (the piece buffers alternate: the kernel of one piece runs while the next is being staged; a buffer is re-used two
                    // pieces later, behind that kernel in stream order)
                    if (d_piece[pc].bytes < hi - lo + 64) d_piece[pc].alloc((size_t)(hi - lo) + 64, &ctx->live_bytes, &ctx->peak_bytes);
                    if (d_desc[pc].bytes < desc.size() * sizeof(BucketSrc)) d_desc[pc].alloc(std::max<size_t>(desc.size(), 4096) * sizeof(BucketSrc));
                    upload(files[t].map + lo, (size_t)(hi - lo), d_piece[pc].as<char>());
                    MGTA_HIP_CHECK(hipMemcpyAsync(d_desc[pc].p, desc.data(), desc.size() * sizeof(BucketSrc), hipMemcpyHostToDevice, st));
                    MGTA_HIP_CHECK(hipStreamSynchronize(st));                             // (desc is re-used by the host; pieces are hundreds of MB)
                    hipLaunchKernelGGL(sdbg_decode_kernel, dim3((unsigned)((desc.size() + 63) / 64)), dim3(64), 0, st, d_piece[pc].as<uint16_t>(),
                                       d_desc[pc].as<BucketSrc>(), (uint32_t)desc.size(), wpt, d_range.as<uint16_t>(), B.g->tips.as<uint32_t>(), d_bad.as<uint32_t>());
                    MGTA_HIP_CHECK(hipGetLastError());
                    pc ^= 1;
                    i = j;
                }
            }
            // the lines these records complete (all that are left, at the end)
            const int64_t end = first + n_in;
            const bool last = b1 == nb;
            const uint64_t line_lo = (uint64_t)(rec_base >> 6), line_hi = last ? B.n_lines : (uint64_t)(end >> 6);
            graph_pack(B, d_range.as<uint16_t>(), rec_base, line_lo, line_hi);
            const int64_t new_carry = last ? 0 : end - (int64_t)(line_hi << 6);
            if (new_carry) {
                // (behind the pack kernel in stream order; source and destination cannot overlap: the source starts >= 64 records in, or the
                // range held less than a line and nothing moves)
                const int64_t src = (int64_t)(line_hi << 6) - rec_base;
                if (src >= new_carry) MGTA_HIP_CHECK(hipMemcpyAsync(d_range.p, d_range.as<uint16_t>() + src, (size_t)new_carry * 2, hipMemcpyDeviceToDevice, st));
                else if (src > 0) {                                      // (a range shorter than a line: through a bounce buffer)
                    uint16_t tmp[64];
                    MGTA_HIP_CHECK(hipMemcpyAsync(tmp, d_range.as<uint16_t>() + src, (size_t)new_carry * 2, hipMemcpyDeviceToHost, st));
                    MGTA_HIP_CHECK(hipStreamSynchronize(st));
                    MGTA_HIP_CHECK(hipMemcpyAsync(d_range.p, tmp, (size_t)new_carry * 2, hipMemcpyHostToDevice, st));
                    MGTA_HIP_CHECK(hipStreamSynchronize(st));
                }
            }
            MGTA_HIP_CHECK(hipStreamSynchronize(st));                     // (the next range decodes into the same buffer)
            carry = new_carry;
            b0 = b1;
            ++n_ranges;
        }
        uint32_t bad = 0;
        MGTA_HIP_CHECK(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, st));
        MGTA_HIP_CHECK(hipStreamSynchronize(st));
        d_piece[0].release(); d_piece[1].release(); d_range.release();
        if (bad) { set_error("%s: %u buckets do not parse to the sizes the index gives", prefix.c_str(), bad); return MGTA_EINVAL; }
        if (getenv("MGTA_LOAD_VERBOSE")) fprintf(stderr, "[load] %lld records in %d range(s) of <= %llu\n", total, n_ranges, (unsigned long long)range);
        return graph_finish(B, out);
    } catch (const HipError &e) { return e.code; }
}

// ---- the whole edge stream of a keep-stream build, taken out of the context: it stays in device memory until it is freed, and a host
// thread of its own brings it to the host (`megagta buildgraph` in the worker: the files are written behind the step that already uses
// the resident graph; per-pass copies into pageable memory cost more than the passes themselves at 50 M reads)
struct mgta_stream {
    mgta_ctx *ctx = nullptr;      // (its memory counters outlive the buffers below)
    int device = 0, k = 0, words_per_tip = 0;
    mgta::DevBuf rec, tips;
    uint64_t n_rec = 0, n_tip_words = 0;
};

int mgta_sdbg_stream_detach(mgta_ctx *ctx, mgta_stream **out) {
    if (!ctx || !out) { set_error("mgta_sdbg_stream_detach: bad argument"); return MGTA_EINVAL; }
    if (!ctx->acc_valid) { set_error("mgta_sdbg_stream_detach: the last build did not keep its whole stream (mgta_ctx_keep_stream)"); return MGTA_EINVAL; }
    try {
        MGTA_HIP_CHECK(hipSetDevice(ctx->device));
        MGTA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        auto st = std::make_unique<mgta_stream>();
        st->ctx = ctx; st->device = ctx->device; st->k = ctx->last_k; st->words_per_tip = ctx->last_words_per_tip;
        st->n_rec = ctx->acc_n_rec; st->n_tip_words = ctx->acc_n_tips * (uint64_t)ctx->last_words_per_tip;
        st->rec = std::move(ctx->acc_rec); st->tips = std::move(ctx->acc_tips);
        ctx->acc_valid = false;
        ctx->last_rec = nullptr; ctx->last_tips = nullptr; ctx->last_first = nullptr; ctx->last_n_rec = 0; ctx->last_n_tips = 0; ctx->last_k = 0;
        ctx_retain(ctx);
        *out = st.release();
        return MGTA_OK;
    } catch (const HipError &e) { return e.code; }
}

int mgta_stream_sizes(const mgta_stream *s, uint64_t *n_recs, uint64_t *n_tip_words) {
    if (!s) { set_error("mgta_stream_sizes: bad argument"); return MGTA_EINVAL; }
    if (n_recs) *n_recs = s->n_rec;
    if (n_tip_words) *n_tip_words = s->n_tip_words;
    return MGTA_OK;
}

// Any host thread; a HIP stream and two pinned staging buffers of its own; touches nothing of the context.
int mgta_stream_download(mgta_stream *s, uint16_t *recs, uint32_t *tips) {
    if (!s || (s->n_rec && !recs) || (s->n_tip_words && !tips)) { set_error("mgta_stream_download: bad argument"); return MGTA_EINVAL; }
    hipStream_t st = nullptr;
    void *stage[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    int rc = MGTA_OK;
    try {
        MGTA_HIP_CHECK(hipSetDevice(s->device));
        MGTA_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        const size_t piece = 128ull << 20;
        for (int i = 0; i < 2; ++i) { MGTA_HIP_CHECK(hipHostMalloc(&stage[i], piece, hipHostMallocDefault)); MGTA_HIP_CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
        struct Part { const char *src; char *dst; size_t bytes; };
        const Part parts[2] = {{s->rec.as<char>(), reinterpret_cast<char *>(recs), (size_t)s->n_rec * 2}, {s->tips.as<char>(), reinterpret_cast<char *>(tips), (size_t)s->n_tip_words * 4}};
        for (const Part &pt : parts) {
            const size_t n_pieces = (pt.bytes + piece - 1) / piece;
            auto issue = [&](size_t i) {
                const size_t len = std::min(piece, pt.bytes - i * piece);
                MGTA_HIP_CHECK(hipMemcpyAsync(stage[i & 1], pt.src + i * piece, len, hipMemcpyDeviceToHost, st));
                MGTA_HIP_CHECK(hipEventRecord(ev[i & 1], st));
            };
            if (n_pieces) issue(0);
            for (size_t i = 0; i < n_pieces; ++i) {                     // piece i + 1 crosses the bus while piece i leaves its staging buffer
                if (i + 1 < n_pieces) issue(i + 1);                     // (its buffer held piece i - 1: copied out in the previous round)
                MGTA_HIP_CHECK(hipEventSynchronize(ev[i & 1]));
                memcpy(pt.dst + i * piece, stage[i & 1], std::min(piece, pt.bytes - i * piece));
            }
        }
    } catch (const HipError &e) { rc = e.code; }
    for (int i = 0; i < 2; ++i) { if (stage[i]) (void)hipHostFree(stage[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); }
    if (st) (void)hipStreamDestroy(st);
    return rc;
}

void mgta_stream_free(mgta_stream *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    mgta_ctx *c = s->ctx;
    delete s;
    ctx_release(c);
}

int mgta_sdbg_load_resident(mgta_ctx *ctx, mgta_sdbg **out) {
    if (!ctx || !out) { set_error("mgta_sdbg_load_resident: bad argument"); return MGTA_EINVAL; }
    if (ctx->last_k == 0 || ctx->last_bucket_lo != 0 || ctx->last_bucket_hi != (uint32_t)MGTA_NUM_BUCKETS || (ctx->last_n_rec && !ctx->last_rec)) {
        set_error("mgta_sdbg_load_resident: the last build of this context did not leave a whole edge stream on the device "
                  "(none yet, a bucket sub-range, or several memory-bound passes)");
        return MGTA_EINVAL;
    }
    try {
        MGTA_HIP_CHECK(hipSetDevice(ctx->device));
        if (ctx->acc_valid) {    // the build's key buffers (grow-only pool) are scratch; a graph of tens of billions of edges needs their room
            size_t free_b = 0, total_b = 0;
            MGTA_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
            const uint64_t need = (uint64_t)((double)ctx->last_n_rec * 3.3) + (1ull << 30);      // lines 2.0 + line counts 0.4 + their prefix sums 0.75 bytes per edge
            if ((uint64_t)free_b < need) { MGTA_HIP_CHECK(hipStreamSynchronize(ctx->stream)); ctx->pool.clear(); }
        }
        // A stream that nobody else is going to want (mgta_ctx_keep_stream 1, not the worker's hand-over to its file writer) and that the
        // device cannot hold a second time as lines: the lines are packed INTO the stream's buffer.  The graph of a 1 G-read set is 63 G
        // edges: 126 GB of records + 126 GB of lines + 24 GB of line counts do not fit 288 GB, records-turned-lines + counts + rank sums do.
        // MGTA_LOAD_INPLACE=1 / 0 forces / forbids it (tests).
        if (ctx->acc_valid && ctx->keep_stream != 2) {
            const int64_t size = (int64_t)ctx->last_n_rec;
            const uint64_t n_lines = (uint64_t)((size + 63) / 64);
            size_t free_b = 0, total_b = 0;
            MGTA_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
            const char *force = getenv("MGTA_LOAD_INPLACE");
            const bool want = force ? atoi(force) != 0 : (uint64_t)free_b < (uint64_t)((double)size * 3.3) + (1ull << 30);
            if (want && size > 0 && ctx->acc_rec.p == ctx->last_rec && ctx->acc_rec.bytes >= (n_lines + 1) * sizeof(GLine)) {
                GraphBuild B;
                const int rc = graph_begin(ctx, ctx->last_k, size, ctx->acc_items.data(), static_cast<const uint32_t *>(ctx->last_tips),
                                           (int64_t)ctx->last_n_tips * ctx->last_words_per_tip, ctx->last_words_per_tip, true, B, &ctx->acc_rec);
                if (rc != MGTA_OK) return rc;
                // from here on the stream's buffer belongs to the graph under construction: whatever happens below, the context must not
                // go on naming it (a failure frees it with B; a retry or an export would read freed memory)
                auto forget_stream = [&]() {
                    ctx->acc_tips.release(); ctx->acc_valid = false;
                    ctx->last_rec = nullptr; ctx->last_tips = nullptr; ctx->last_first = nullptr; ctx->last_n_rec = 0; ctx->last_n_tips = 0; ctx->last_k = 0;
                };
                try {
                    graph_pack(B, B.g->lines.as<uint16_t>(), 0, 0, B.n_lines);
                    MGTA_HIP_CHECK(hipMemsetAsync(B.g->lines.as<GLine>() + n_lines, 0, sizeof(GLine), ctx->stream));   // (the line past the end reads as empty)
                    MGTA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                    forget_stream();                               // the stream is gone: it IS the graph now
                    return graph_finish(B, out);
                } catch (const HipError &) { forget_stream(); throw; }
            }
        }
        if (ctx->acc_valid)      // a multi-pass build that kept its whole stream (mgta_ctx_keep_stream): records per bucket are on the host
            return load_graph(ctx, ctx->last_k, static_cast<const uint16_t *>(ctx->last_rec), (int64_t)ctx->last_n_rec, ctx->acc_items.data(),
                              static_cast<const uint32_t *>(ctx->last_tips), (int64_t)ctx->last_n_tips * ctx->last_words_per_tip,
                              ctx->last_words_per_tip, true, out);
        // records before every bucket (-1 = empty bucket) -> records per bucket
        std::vector<int64_t> first((size_t)MGTA_NUM_BUCKETS * 3), items(MGTA_NUM_BUCKETS);
        if (ctx->last_n_rec) {
            MGTA_HIP_CHECK(hipMemcpyAsync(first.data(), ctx->last_first, first.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            MGTA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        } else {
            std::fill(first.begin(), first.end(), (int64_t)-1);
        }
        int64_t nxt = (int64_t)ctx->last_n_rec;
        for (int64_t b = MGTA_NUM_BUCKETS - 1; b >= 0; --b) {
            int64_t f = first[(size_t)b * 3];
            if (f < 0) f = nxt;
            items[(size_t)b] = nxt - f;
            nxt = f;
        }
        return load_graph(ctx, ctx->last_k, static_cast<const uint16_t *>(ctx->last_rec), (int64_t)ctx->last_n_rec, items.data(),
                          static_cast<const uint32_t *>(ctx->last_tips), (int64_t)ctx->last_n_tips * ctx->last_words_per_tip,
                          ctx->last_words_per_tip, true, out);
    } catch (const HipError &e) { return e.code; }
}

void mgta_sdbg_free(mgta_sdbg *g) {
    if (!g) return;
    mgta_ctx *c = g->ctx;
    delete g;
    ctx_release(c);
}
int64_t mgta_sdbg_size(const mgta_sdbg *g) { return g ? g->dev.size : -1; }
int mgta_sdbg_k(const mgta_sdbg *g) { return g ? g->dev.k : -1; }

int mgta_sdbg_outgoing(mgta_sdbg *g, const int64_t *edges, int64_t n, int64_t *out4, int8_t *outdeg) {
    if (!g || n < 0 || (n > 0 && (!edges || !out4 || !outdeg))) { set_error("mgta_sdbg_outgoing: bad argument"); return MGTA_EINVAL; }
    for (int64_t i = 0; i < n; ++i)
        if (edges[i] < 0 || edges[i] >= g->dev.size) { set_error("edge id %lld out of range", (long long)edges[i]); return MGTA_EINVAL; }
    if (n == 0) return MGTA_OK;
    try {
        mgta_ctx *ctx = g->ctx;
        MGTA_HIP_CHECK(hipSetDevice(ctx->device));
        DevBuf d_e, d_o, d_d;
        d_e.alloc(n * 8); d_o.alloc(n * 32); d_d.alloc(n);
        MGTA_HIP_CHECK(hipMemcpyAsync(d_e.p, edges, n * 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(graph_outgoing_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, g->dev, d_e.as<int64_t>(), n,
                           d_o.as<int64_t>(), d_d.as<int8_t>());
        MGTA_HIP_CHECK(hipMemcpyAsync(out4, d_o.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
        MGTA_HIP_CHECK(hipMemcpyAsync(outdeg, d_d.p, n, hipMemcpyDeviceToHost, ctx->stream));
        MGTA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return MGTA_OK;
    } catch (const HipError &e) { return e.code; }
}

int mgta_sdbg_invalid_bits(mgta_sdbg *g, uint64_t *words) {
    if (!g || !words) { set_error("mgta_sdbg_invalid_bits: bad argument"); return MGTA_EINVAL; }
    if (g->dev.size == 0) return MGTA_OK;
    try {
        mgta_ctx *ctx = g->ctx;
        MGTA_HIP_CHECK(hipSetDevice(ctx->device));
        const uint64_t n = g->dev.n_lines;
        DevBuf d;
        d.alloc(n * 8);
        hipLaunchKernelGGL(graph_invalid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, g->dev.lines, n, d.as<uint64_t>());
        MGTA_HIP_CHECK(hipMemcpyAsync(words, d.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        MGTA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return MGTA_OK;
    } catch (const HipError &e) { return e.code; }
}

int mgta_sdbg_index_edges(mgta_sdbg *g, const uint8_t *seqs, int64_t n, int64_t *edge_ids) {
    if (!g || n < 0 || (n > 0 && (!seqs || !edge_ids))) { set_error("mgta_sdbg_index_edges: bad argument"); return MGTA_EINVAL; }
    if (n == 0) return MGTA_OK;
    try {
        mgta_ctx *ctx = g->ctx;
        MGTA_HIP_CHECK(hipSetDevice(ctx->device));
        size_t sb = (size_t)n * (g->dev.k + 1);
        DevBuf d_s, d_i;
        d_s.alloc(sb); d_i.alloc(n * 8);
        MGTA_HIP_CHECK(hipMemcpyAsync(d_s.p, seqs, sb, hipMemcpyHostToDevice, ctx->stream));
        if (g->dev.size > 0)
            hipLaunchKernelGGL(graph_index_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, g->dev, d_s.as<uint8_t>(), n,
                               d_i.as<int64_t>());
        else
            MGTA_HIP_CHECK(hipMemsetAsync(d_i.p, 0xFF, n * 8, ctx->stream));
        MGTA_HIP_CHECK(hipMemcpyAsync(edge_ids, d_i.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        MGTA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return MGTA_OK;
    } catch (const HipError &e) { return e.code; }
}

}  // extern "C"
