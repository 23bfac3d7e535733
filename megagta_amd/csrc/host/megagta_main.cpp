// megagta_main.cpp — `megagta buildgraph` / `megagta search`: the process-level drop-in boundary
// (argv, files, stderr, exit code) of the reference's multi-call binary (megagta.cpp:33-78) for the
// two sub-commands on the hot path.  Host code only; all compute goes through the C ABI of
// libmegagta_hip.so.  Other sub-commands (buildlib, denovo, findstart, filterbylen, translate,
// readstat) are outside the path and are NOT provided here.
//
//   buildgraph  -k INT -m INT --host_mem F --mem_flag INT --gpu_mem F --output_prefix STR
//               --num_cpu_threads INT --num_output_threads INT --read_lib_file STR
//               [--need_mercy] [--assist_seq FASTA]                       build_graph.cpp:38-48
//   search      <sdbg_prefix> <gene_list> <starting_kmers_prefix> <output_prefix> <prune_len>
//               <low_cov_penalty> [num_threads]                            search.cpp:72-90
#include <fcntl.h>
#include <getopt.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sys/time.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <parallel/algorithm>
#include <atomic>
#include <map>
#include <memory>
#include <thread>
#include <string>
#include <vector>

#include "../../../include/megagta_hip.h"
#include "formats.hpp"

using namespace mgta_host;

static double now_s() {
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    return tv.tv_sec + 1e-6 * tv.tv_usec;
}

// ------------------------------------------------------------------------------------------------
// `megagta serve`: one process for all the steps of a driver run.  The sub-commands are the same functions; what a worker keeps
// between them is what a one-shot process pays for again at every step: the device context with its work memory (mapping device
// memory costs ~27 ms/GB), the read library (reads.lib.bin unpacked once), and the graph of the last `buildgraph`, which stays on
// the device for the `denovo` / `search` that follows instead of travelling through PREFIX.sdbg.* (succinct_dbg.cpp:595-723; the
// files are still written: they are the run's artefacts and what `--continue` resumes from).
struct Session {
    bool active = false;
    mgta_ctx *ctx = nullptr;
    std::string lib_key;          // path of the .bin the library was read from
    PackedReads lib;              // reversed; the contigs one step appended behind it stay for the next step that wants the same file
    bool lib_loaded = false;
    PackedReads::Mark lib_mark{}; // end of the library proper
    std::string extra_key;        // the contigs file behind it (path | size | mtime)
    bool have_extra = false;
    mgta_ctx *ctx2 = nullptr;     // second context of the device: the other lane of a two-gene search
    mgta_sdbg *graph = nullptr;   // graph of the last buildgraph, not used yet
    std::string graph_prefix;
    std::thread writer;           // PREFIX.sdbg.* of the last buildgraph being written while the next step already runs on the resident graph
    std::string writer_error;     // why that thread failed (set by the thread, read after the join)
    mgta_stream *stream = nullptr;   // the edge stream that thread downloads from the device (freed on this thread once it has joined)
    // the contigs of the last `denovo`: their FASTA text stays in memory for the `buildgraph` / `findstart` that take them as --assist_seq /
    // extra sequences (100 M reads: 4 GB of text, 8 s to read back and parse on one thread), and is written to PREFIX.contigs.fa by a thread
    // of its own behind the next step
    char *ctext = nullptr;
    uint64_t ctext_len = 0;
    std::string ctext_path;       // the file the text is (being) written to; stays set after the text is freed
    std::thread cwriter;
    std::atomic<bool> cwriter_done{true};
    std::string cwriter_error;
};
static Session g_sess;
// The graph files of the last buildgraph are complete: called before anything reads them, before the next build, at the end, and by the
// driver's "sync" request before it writes the checkpoint that says "graph built" (the reference writes its checkpoint after the files:
// megagta.py:538-586).  Returns 1 when the writer failed (disk full, ...): the failure belongs to the build, not to whatever step
// happened to be running when the thread hit it.
static int writer_join() {
    if (g_sess.writer.joinable()) g_sess.writer.join();
    if (g_sess.stream) { mgta_stream_free(g_sess.stream); g_sess.stream = nullptr; }
    if (g_sess.writer_error.empty()) return 0;
    fprintf(stderr, "    [ERROR] writing the graph files of the last buildgraph failed: %s\n", g_sess.writer_error.c_str());
    fflush(stderr);
    g_sess.writer_error.clear();
    return 1;
}
// The contigs file of the last denovo is complete (and its text, if no step can want it any more, is freed): before the next denovo, at
// "sync" / "release" / the end.  Returns 1 when the writer failed.
static int contigs_join(bool free_text) {
    if (g_sess.cwriter.joinable()) g_sess.cwriter.join();
    if (free_text && g_sess.ctext) { mgta_host_free(g_sess.ctext); g_sess.ctext = nullptr; g_sess.ctext_len = 0; }
    if (g_sess.cwriter_error.empty()) return 0;
    fprintf(stderr, "    [ERROR] writing the contigs of the last denovo failed: %s\n", g_sess.cwriter_error.c_str());
    fflush(stderr);
    g_sess.cwriter_error.clear();
    return 1;
}

static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}
// MEGAGTA_DEVICE: the GPU of this process (a rank of a multi-GPU step is started with its own; default 0)
static mgta_ctx *ctx_get() {
    if (g_sess.active && g_sess.ctx) return g_sess.ctx;
    mgta_ctx *ctx = mgta_ctx_create(env_int("MEGAGTA_DEVICE", 0));
    if (!ctx) die("%s", mgta_last_error());
    if (g_sess.active) g_sess.ctx = ctx;
    return ctx;
}
static void ctx_put(mgta_ctx *ctx) {
    if (!g_sess.active) mgta_ctx_destroy(ctx);
}
// The library of `bin_path` (reads.lib.bin), reversed, with the sequences of `extra` (a FASTA of contigs: the assist sequences of
// buildgraph, the contigs findstart scans with the reads) appended behind it, finished for upload.  `mk` = the end of the library
// proper.  The worker keeps both between steps: a multi-k run reads the contigs of a k once for the buildgraph and the findstart calls
// that use them (at 20 M reads: 0.8 GB of text, 1.5 s per parse), and never re-reads the library.
static std::string file_key(const std::string &path) {
    struct stat sb;
    if (path.empty() || stat(path.c_str(), &sb) != 0) return path;
    return path + "|" + std::to_string((long long)sb.st_size) + "|" + std::to_string((long long)sb.st_mtime) + "|" + std::to_string((long long)sb.st_mtim.tv_nsec);
}
static PackedReads &lib_get(const std::string &bin_path, const std::string &lib_prefix, const std::string &extra, bool extra_is_assist, PackedReads &local,
                            PackedReads::Mark &mk) {
    PackedReads &pr = g_sess.active ? g_sess.lib : local;
    if (!(g_sess.active && g_sess.lib_loaded && g_sess.lib_key == bin_path)) {
        if (g_sess.active) { g_sess.lib = PackedReads(); g_sess.lib_loaded = false; }
        if (!lib_prefix.empty()) load_read_lib(lib_prefix, /*reverse=*/true, pr);       // cx1_read2sdbg_s1.cpp:97,117
        else load_read_bin(bin_path, /*reverse=*/true, pr);
        if (g_sess.active) { g_sess.lib_loaded = true; g_sess.lib_key = bin_path; g_sess.lib_mark = pr.mark(); g_sess.extra_key.clear(); g_sess.have_extra = false; }
        else g_sess.lib_mark = pr.mark();
    }
    mk = g_sess.lib_mark;
    // the contigs this worker's own denovo has just made: taken from their text in memory (the file may still be on its way to the disk)
    // (the same FILE however it is spelled -- relative, `//`, a symlinked directory: compared by device + inode once the writer has
    // created it, by name before; advisor r5)
    auto same_file = [](const std::string &a, const std::string &b) {
        if (a == b) return true;
        struct stat sa, sb;
        return !a.empty() && !b.empty() && stat(a.c_str(), &sa) == 0 && stat(b.c_str(), &sb) == 0 && sa.st_dev == sb.st_dev && sa.st_ino == sb.st_ino;
    };
    const bool from_memory = g_sess.active && !extra.empty() && !g_sess.ctext_path.empty() && same_file(extra, g_sess.ctext_path);
    // any other extra file is read from the disk: never while the background writer may still be in the middle of one
    if (g_sess.active && !extra.empty() && !from_memory && g_sess.cwriter.joinable() && contigs_join(false) != 0) die("the contigs of the last denovo are incomplete");
    const std::string key = extra.empty() ? std::string() : from_memory ? "mem|" + extra : file_key(extra);
    if (g_sess.active && g_sess.have_extra && g_sess.extra_key == key) {
        logf("library%s: still in memory", extra.empty() ? "" : " + contigs");
        return pr;
    }
    pr.rewind(mk);
    if (!extra.empty()) {
        if (from_memory && g_sess.ctext) {
            const double t0 = now_s();
            load_fasta_text(g_sess.ctext, (size_t)g_sess.ctext_len, /*reverse=*/true, pr);
            logf("contigs of %s: %zu sequences packed from memory (%.3f s)", extra.c_str(), pr.start.size() - mk.n_start, now_s() - t0);
            if (g_sess.cwriter_done.load()) contigs_join(true);                          // (written already: the text has served)
        } else {
            if (from_memory && contigs_join(true) != 0) die("%s is incomplete", extra.c_str());   // (the text is gone: the file is complete by now)
            if (extra_is_assist) load_assist_fasta(extra, /*reverse=*/true, pr);        // :121-134
            else load_fastx(extra, true, pr);
        }
    }
    pr.finish();
    if (g_sess.active) { g_sess.extra_key = key; g_sess.have_extra = true; }
    return pr;
}
static void graph_drop() {
    if (g_sess.graph) { mgta_sdbg_free(g_sess.graph); g_sess.graph = nullptr; g_sess.graph_prefix.clear(); }
}
// the graph PREFIX names: the one the last buildgraph of this process left on the device, else read from the files
static mgta_sdbg *graph_get(mgta_ctx *ctx, const std::string &prefix, int *k_out, size_t *n_edges) {
    if (g_sess.active && g_sess.graph && g_sess.graph_prefix == prefix) {
        mgta_sdbg *g = g_sess.graph;
        g_sess.graph = nullptr; g_sess.graph_prefix.clear();
        *k_out = mgta_sdbg_k(g); *n_edges = (size_t)mgta_sdbg_size(g);
        logf("graph %s: still on the device", prefix.c_str());
        return g;
    }
    graph_drop();
    if (writer_join() != 0) die("the graph files of %s are incomplete", prefix.c_str());
    mgta_sdbg *g = nullptr;                                              // the files are copied to the device as they are and parsed there
    if (mgta_sdbg_load_files(ctx, prefix.c_str(), &g) != MGTA_OK) die("mgta_sdbg_load_files: %s", mgta_last_error());
    *k_out = mgta_sdbg_k(g); *n_edges = (size_t)mgta_sdbg_size(g);
    return g;
}

struct RssLine {   // AutoMaxRssRecorder, utils.h:99-128
    double t0 = now_s();
    ~RssLine() {
        struct rusage u;
        getrusage(RUSAGE_SELF, &u);
        logf("Real: %.4lf\tuser: %.4lf\tsys: %.4lf\tmaxrss: %ld", now_s() - t0, u.ru_utime.tv_sec + 1e-6 * u.ru_utime.tv_usec,
             u.ru_stime.tv_sec + 1e-6 * u.ru_stime.tv_usec, u.ru_maxrss);
    }
};

// ------------------------------------------------------------------------------------------------
static int sink_collect(void *user, int32_t b0, int32_t b1, const int64_t *counts, const uint16_t *recs, int64_t n, const uint16_t *large,
                        int64_t nl, const uint32_t *tips, int64_t ntw) {
    EdgeStream &s = *static_cast<EdgeStream *>(user);
    for (int32_t b = b0; b < b1; ++b) {
        s.bucket_items[b] = counts[3 * (b - b0)];
        s.bucket_large[b] = counts[3 * (b - b0) + 1];
        s.bucket_tips[b] = counts[3 * (b - b0) + 2];
    }
    if (recs) s.recs.insert(s.recs.end(), recs, recs + n);           // (NULL: the stream stays on the device, mgta_ctx_keep_stream 2)
    s.large.insert(s.large.end(), large, large + nl);
    if (tips) s.tips.insert(s.tips.end(), tips, tips + ntw);
    return 0;
}

static int main_buildgraph(int argc, char **argv) {
    RssLine rss;
    int k = 0, min_count = 0, mem_flag = 1, n_threads = 0, n_out_threads = 0, need_mercy = 0;
    double host_mem = 0, gpu_mem = 0;
    std::string out_prefix, lib_file, assist;
    static struct option opts[] = {{"kmer_k", required_argument, 0, 'k'}, {"min_kmer_frequency", required_argument, 0, 'm'},
                                   {"host_mem", required_argument, 0, 1}, {"gpu_mem", required_argument, 0, 2},
                                   {"num_cpu_threads", required_argument, 0, 3}, {"num_output_threads", required_argument, 0, 4},
                                   {"read_lib_file", required_argument, 0, 5}, {"assist_seq", required_argument, 0, 6},
                                   {"output_prefix", required_argument, 0, 7}, {"mem_flag", required_argument, 0, 8},
                                   {"need_mercy", no_argument, 0, 9}, {0, 0, 0, 0}};
    optind = 1;
    int ch;
    bool bad = false;
    while ((ch = getopt_long(argc, argv, "k:m:", opts, nullptr)) != -1) {
        switch (ch) {
        case 'k': k = atoi(optarg); break;
        case 'm': min_count = atoi(optarg); break;
        case 1: host_mem = atof(optarg); break;
        case 2: gpu_mem = atof(optarg); break;
        case 3: n_threads = atoi(optarg); break;
        case 4: n_out_threads = atoi(optarg); break;
        case 5: lib_file = optarg; break;
        case 6: assist = optarg; break;
        case 7: out_prefix = optarg; break;
        case 8: mem_flag = atoi(optarg); break;
        case 9: need_mercy = 1; break;
        default: bad = true;
        }
    }
    // same argument checks and messages as build_graph.cpp:52-83
    const char *why = nullptr;
    if (bad) why = "uknown option";
    else if (lib_file.empty()) why = "No input file!";
    else if (host_mem == 0) why = "Please specify the host memory!";
    else if (n_threads == 1) why = "Number of CPU threads should be at least 2!";
    else if (n_threads != 0 && n_out_threads >= n_threads) why = "Number of output threads must be less than number of CPU threads!";
    if (why) {
        fprintf(stderr, "%s\nUsage: sdbg_builder read2sdbg --read_lib_file fastx_file -o out\n", why);
        return 1;
    }
    (void)mem_flag;
    if (min_count < 1) min_count = 1;
    // A build over several GPUs (`megagta.py --gpus N`): N processes, MEGAGTA_RANK / MEGAGTA_WORLD / MEGAGTA_DEVICE each.  The 65536
    // prefix buckets are independent once every rank holds the reads (cx1.h:494-590 loops over bucket ranges for the same reason): rank r
    // builds its share and writes it as PREFIX.sdbg.r, the file a writer thread r of the reference would have written
    // (sdbg_multi_io.h:83-187); `megagta sdbgmerge PREFIX N` then writes the index.  No data leaves a GPU except into its file.
    const int world = std::max(1, env_int("MEGAGTA_WORLD", 1)), rank = env_int("MEGAGTA_RANK", 0);
    if (rank < 0 || rank >= world || world > 65536) die("MEGAGTA_RANK = %d outside MEGAGTA_WORLD = %d", rank, world);
    const int share = (65536 + world - 1) / world, b_lo = std::min(65536, rank * share), b_hi = std::min(65536, (rank + 1) * share);

    double t0 = now_s();
    if (writer_join() != 0) return 1;
    PackedReads local;
    PackedReads::Mark mk;
    PackedReads &pr = lib_get(lib_file + ".bin", lib_file, assist, true, local, mk);
    pr.n_short = mk.n_start ? mk.n_start - 1 : 0;                        // the library's reads; assist sequences follow
    logf("%zu reads, %d max read length, %llu total bases (load %.3f s)", pr.start.size() - 1, pr.max_len,
         (unsigned long long)pr.start.back(), now_s() - t0);

    mgta_ctx *ctx = ctx_get();
    graph_drop();
    const double t_dev = now_s();
    const bool hand_over = g_sess.active && world == 1;
    const bool device_stream = hand_over && !getenv("MEGAGTA_SYNC_WRITES");
    if (hand_over) mgta_ctx_keep_stream(ctx, device_stream ? 2 : 1);
    // --gpu_mem: device budget in bytes.  Unset, a one-shot process takes 64 GB at most: device memory is mapped at ~27 ms/GB
    // (measured: 194 GB cost 5.3 s before the first kernel ran), which outweighs the few extra bucket-range passes of a tighter budget
    // (100 M reads: 3 passes in 1.5 s with 194 GB, 10 passes in 2.0 s with 70 GB).  The worker keeps its pool between the builds of a run,
    // so it pays that once and takes what fits beside the next step's needs: 3/5 of the device (a `denovo` on the graph of 100 M reads
    // holds ~80 GB; 50 M reads, k = 44: 7 passes of 540 ms under 64 GB).
    uint64_t budget = gpu_mem > 0 ? (uint64_t)gpu_mem : (64ull << 30);
    if (gpu_mem <= 0 && g_sess.active) {
        uint64_t total = 0;
        if (mgta_ctx_device_memory(ctx, nullptr, &total) == MGTA_OK && total) budget = std::max<uint64_t>(budget, total / 5 * 3);
    }
    mgta_ctx_set_mem_limit(ctx, budget);
    std::shared_ptr<EdgeStream> sp = std::make_shared<EdgeStream>();
    EdgeStream &s = *sp;
    s.k = k; s.words_per_tip = (2 * k + 31) / 32;
    mgta_build_stats st;
    mgta_reads *rd = nullptr;
    if (mgta_reads_upload(ctx, pr.words.data(), pr.words.size(), pr.start.data(), pr.start.size() - 1, &rd) != MGTA_OK) die("mgta_reads_upload: %s", mgta_last_error());
    int rc = mgta_sdbg_build_resident(ctx, rd, pr.n_short, k, min_count, min_count > 1 ? need_mercy : 0, b_lo, b_hi, sink_collect, &s, &st);
    if (rc != MGTA_OK) die("mgta_sdbg_build: %s", mgta_last_error());
    mgta_reads_free(rd);
    if (hand_over) {                                                     // the graph stays on the device for the step that uses it
        if (mgta_sdbg_load_resident(ctx, &g_sess.graph) != MGTA_OK) die("mgta_sdbg_load_resident: %s", mgta_last_error());
        g_sess.graph_prefix = out_prefix;
        // the stream leaves the context in one piece: the writer thread below brings it to the host and writes the files
        if (device_stream && mgta_sdbg_stream_detach(ctx, &g_sess.stream) != MGTA_OK) die("mgta_sdbg_stream_detach: %s", mgta_last_error());
        mgta_ctx_keep_stream(ctx, 0);
    }
    if (min_count > 1 && rank == 0) {                                    // PREFIX.counting (s1_post_proc, cx1_read2sdbg_s1.cpp:923-930); every rank counts all (k+1)-mers
        std::vector<int64_t> hist(65536);
        if (mgta_sdbg_last_counting(ctx, hist.data()) != MGTA_OK) die("%s", mgta_last_error());
        FILE *cf = fopen((out_prefix + ".counting").c_str(), "w");
        if (!cf) die("cannot write %s.counting", out_prefix.c_str());
        long long acc = 0;
        for (int i = 1; i <= 65535; ++i) { acc += hist[i]; fprintf(cf, "%d %lld\n", i, acc); }
        fclose(cf);
    }
    ctx_put(ctx);
    logf("device build: %.1f ms (%d pass%s, %lld sort items, %.3f Gk-mer/s; count %.0f, keys %.0f, sort %.0f, emit %.0f, copy to the host %.0f ms; stage 1 %.0f ms; "
         "upload + graph hand-over: wall %.2f s)", st.ms_total, st.n_passes, st.n_passes > 1 ? "es" : "",
         (long long)st.n_items, st.n_kmers / (st.ms_total * 1e-3) / 1e9, st.ms_count, st.ms_gen, st.ms_sort, st.ms_emit, st.ms_d2h, st.ms_stage1, now_s() - t_dev);
    // the files: the run's artefacts, what `--continue` resumes from and what a one-shot process reads.  In the worker the step that
    // follows works on the resident graph, so they are written by a host thread behind it (joined before anything reads them)
    mgta_stream *dev_stream = device_stream ? g_sess.stream : nullptr;
    auto write_files = [sp, out_prefix, world, rank, b_lo, b_hi, dev_stream]() {
        EdgeStream &s = *sp;
        double t1 = now_s();
        if (dev_stream) {                                                // records and tip labels: one download of the whole stream
            uint64_t nr = 0, nt = 0;
            mgta_stream_sizes(dev_stream, &nr, &nt);
            s.recs.resize(nr); s.tips.resize(nt);
            if (mgta_stream_download(dev_stream, s.recs.data(), s.tips.data()) != MGTA_OK) die("mgta_stream_download: %s", mgta_last_error());
            logf("edge stream on the host: %.3f s (%llu records)", now_s() - t1, (unsigned long long)nr);
        }
        if (world == 1) write_sdbg(out_prefix, s);
        else {
            write_sdbg(out_prefix, s, rank, b_lo, b_hi, true);
            logf("rank %d of %d: buckets [%d, %d) -> %s.sdbg.%d", rank, world, b_lo, b_hi, out_prefix.c_str(), rank);
        }
        long long nw[9] = {0};
        for (uint16_t r : s.recs) nw[r & 15]++;
        logf("Number of $ A C G T A- C- G- T-:");                           // s2_post_proc, cx1_read2sdbg_s2.cpp:899-915
        logf("%lld %lld %lld %lld %lld %lld %lld %lld %lld", nw[0], nw[1], nw[2], nw[3], nw[4], nw[5], nw[6], nw[7], nw[8]);
        logf("Total number of edges: %zu", s.recs.size());
        logf("Total number of $v edges: %zu (write %.3f s)", s.words_per_tip ? s.tips.size() / s.words_per_tip : 0, now_s() - t1);
    };
    if (hand_over && !getenv("MEGAGTA_SYNC_WRITES"))
        g_sess.writer = std::thread([write_files]() {
            set_soft_die(true);
            try { write_files(); }
            catch (const std::exception &e) { g_sess.writer_error = e.what(); }
        });
    else write_files();
    return 0;
}

// ------------------------------------------------------------------------------------------------
struct FastaOut {
    FILE *f;
    const std::string *gene;
    const std::vector<std::string> *kmers;
};
static int sink_contig(void *user, int64_t i, const char *left, int64_t ll, const char *right, int64_t rl, const mgta_astar_side *,
                       const mgta_astar_side *) {
    FastaOut &o = *static_cast<FastaOut *>(user);                        // hmm_graph_search.h:79
    fprintf(o.f, ">%s_contig_%lld_contig_%lld\n%.*s%s%.*s\n", o.gene->c_str(), (long long)(2 * i), (long long)(2 * i + 1), (int)ll, left,
            (*o.kmers)[i].c_str(), (int)rl, right);
    return 0;
}

static mgta_hmm *upload_hmm(mgta_ctx *ctx, const std::string &path) {
    ProfileHmm hm;
    if (!parse_hmm(path, hm)) die("cannot open HMM %s", path.c_str());
    mgta_hmm *out = nullptr;
    if (mgta_hmm_load(ctx, hm.M, hm.A, hm.msc.data(), hm.tsc.data(), hm.max_match.data(), hm.h.data(), hm.alpha, &out) != MGTA_OK)
        die("mgta_hmm_load(%s): %s", path.c_str(), mgta_last_error());
    return out;
}

// ordered-commit window B and cost term R of a gene's batch by its number of seeds, with the MEGAGTA_CACHE_WINDOW / MEGAGTA_CACHE_COST_RATE
// overrides (search_dist.py::window_and_rate is the same table for the multi-GPU ranks; `megagta searchplan N` prints it, tests compare)
// an integer from the environment: unset or empty = not given; anything that is not an integer is refused (search_dist.py::_env_int is the
// same rule for the multi-GPU ranks, so that one environment means one mode on both paths)
constexpr long long kDefaultCostKnee = 0;        // expansions up to which the cost term runs at the plan's rate ...
constexpr int kDefaultCostRate2 = 0;             // ... and the expansions per seed beyond (0: no knee)
static bool env_int_strict(const char *name, int *out) {
    const char *e = getenv(name);
    if (!e || !*e) return false;
    char *end = nullptr;
    const long v = strtol(e, &end, 10);
    if (end == e || *end) die("%s must be an integer (got '%s')", name, e);
    *out = (int)std::max(-1000000000l, std::min(1000000000l, v));
    return true;
}
static void search_plan(size_t ns, int *window, int *rate, long long *knee, int *rate2) {
    int cache_window = -2;                          // < -1 = choose per gene; -1 = no ordering at all (timing-dependent, like the reference's OMP run)
    env_int_strict("MEGAGTA_CACHE_WINDOW", &cache_window);
    int cost_rate = 0;                              // MEGAGTA_CACHE_COST_RATE: see mgta_ctx_set_search_cost_rate (> 0: expansions per seed, < 0: seeds
                                                    // per expansion, 0: no cost term); unset = chosen per gene by its number of seeds
    const bool cost_rate_set = env_int_strict("MEGAGTA_CACHE_COST_RATE", &cost_rate);
    *window = cache_window >= -1 ? cache_window : ns < 32768 ? 1024 : ns < 65536 ? 2048 : ns < 196608 ? 4096 : 8192;
    // (window 1 without an explicit rate = the reference's sequential run: no cost term; window 0 / -1 ignore it)
    *rate = cost_rate_set ? cost_rate : *window == 1 ? 0 : (ns < 65536 ? 4 : ns < 393216 ? 2 : 1);
    // the cost term's knee (MEGAGTA_CACHE_COST_KNEE expansions) and its rate beyond (MEGAGTA_CACHE_COST_RATE2): see mgta_ctx_set_search_cost_curve
    int k_env = 0, r2_env = 0;
    const bool knee_set = env_int_strict("MEGAGTA_CACHE_COST_KNEE", &k_env), r2_set = env_int_strict("MEGAGTA_CACHE_COST_RATE2", &r2_env);
    *knee = knee_set ? std::max(0, k_env) : kDefaultCostKnee;
    *rate2 = r2_set ? r2_env : kDefaultCostRate2;
    if (*rate < 1 || *knee <= 0 || *rate2 <= *rate) { *knee = 0; *rate2 = 0; }           // (one rate throughout)
}

static int main_search(int argc, char **argv) {
    RssLine rss;
    if (argc < 7) {
        fprintf(stderr, "Usage: %s <succinct_dbg> <gene_list> <starting_kmers_prefix> <output_prefix> <prune_len> <low_cov_penalty> [num_threads=0]\n",
                argv[0]);
        return 1;
    }
    int prune = atoi(argv[5]);
    double pen = atof(argv[6]);
    // argv[7] (num_threads) is accepted and ignored: the batch runs on the device.  The shared term_nodes cache (search.cpp:182)
    // runs with an ordered-commit window (deterministic); MEGAGTA_CACHE_WINDOW overrides: 0 = no sharing, 1 = exactly `search ... 1`
    double t0 = now_s();
    logf("Loading SdBG...");
    mgta_ctx *ctx = ctx_get();
    int gk = 0;
    size_t n_edges = 0;
    mgta_sdbg *g = graph_get(ctx, argv[1], &gk, &n_edges);
    logf("Done! Time elapsed: %.4lf", now_s() - t0);
    // the build's key buffers (tens of GB in the worker) make room for the searches' pool: the graph is packed, its files are written from
    // the host copy, and no build follows the search of a run
    mgta_ctx_release_scratch(ctx);
    const size_t klen = (size_t)gk + 1;
    auto run_gene = [&](const GeneEntry &gene, mgta_ctx *ctx) {
        double tg = now_s();
        logf("START %s", gene.name.c_str());
        mgta_hmm *fw = upload_hmm(ctx, gene.fwd_hmm), *rv = upload_hmm(ctx, gene.rev_hmm);
        std::vector<std::string> kmers;
        std::vector<int32_t> start;
        std::string sk = std::string(argv[3]) + "_" + gene.name + "_starting_kmers.txt";
        std::string on = std::string(argv[4]) + "_raw_contigs_" + gene.name + ".fasta";
        FILE *out = fopen(on.c_str(), "w");
        if (!out) die("cannot write %s", on.c_str());
        if (!read_seeds(sk, kmers, start)) {                              // search.cpp:163-167: report and go on
            fprintf(stderr, "    [ERROR] Fail to open %s\n", sk.c_str());
            fclose(out);
            mgta_hmm_free(fw); mgta_hmm_free(rv);
            return;
        }
        logf("Searching from %zu starting kmers", kmers.size());
        std::string flat;
        flat.reserve(kmers.size() * klen);
        for (const std::string &km : kmers) {
            if (km.size() < klen) die("%s: seed k-mer shorter than k+1 = %zu", sk.c_str(), klen);
            flat.append(km, 0, klen);
        }
        FastaOut fo{out, &gene.name, &kmers};
        mgta_astar_stats st;
        // The ordered-commit window B and the cost term R (a search that has run r expansions no longer holds back the seeds below
        // j + B + r / R: the window slides past the long searches) by the number of seeds, from reads -> contigs runs of 0.2 .. 5 M reads
        // (profiles/r02/e2e_window_sweep.log; seeds of rplB / nirK, search seconds):
        //    7 k / 10 k: 1024 + 4  2.0 / 3.0    (half the seeds, no cost term: 2.4 / 3.7)
        //   18 k / 24 k: 1024 + 4  3.0 / 3.9    (half the seeds: 3.9 / 5.8;  2048 + 4: 3.0 / 4.0)
        //   37 k / 50 k: 2048 + 4  3.5 / 5.4    (8192 + 2: 4.1 / 7.4;  1024 + 4: 3.7 / 5.4;  4096 + 4: 3.6 / 5.5)
        //   80 k / 108 k: 4096 + 2  5.3 / 8.7   (8192 + 2: 5.6 / 9.2;  4096 + 4: 5.5 / 8.6;  2048 + 2: 7.0 / 8.4)
        //  200 k / 270 k: 8192 + 2  9.0 / 15.8  (8192 + 4: 10.3 / 16.7;  4096 + 4: 11.8 / 18.2);  414 k: 8192 + 2 14.4 (4096 + 2: 15.4)
        //  from ~400 k seeds on one seed per expansion: 414 k seeds (10 M-read graph) 8192 + 1 12.7 s; 400 k seeds of rplB on the 100 M-read
        //  graph (profiles/r03/sweep_window_400k_100M.log): 8192 + 2 26.8 s, 8192 + 1 22.1 / 22.2 / 25.3 s, 4096 + 1 21.9 s, unordered 15.9 s
        const size_t ns = kmers.size();
        int window = 0, rate = 0, rate2 = 0;
        long long knee = 0;
        search_plan(ns, &window, &rate, &knee, &rate2);
        if ((knee ? mgta_ctx_set_search_cost_curve(ctx, rate, (uint64_t)knee, rate2) : mgta_ctx_set_search_cost_rate(ctx, rate)) != MGTA_OK)
            die("MEGAGTA_CACHE_COST_RATE must be >= -64 (%s)", mgta_last_error());
        logf("sharing rule: window %d, cost term %d expansions per seed%s", window, rate,
             knee ? (" up to " + std::to_string(knee) + " expansions, " + std::to_string(rate2) + " beyond").c_str() : "");
        if (mgta_astar_batch_on(ctx, g, fw, rv, flat.data(), start.data(), (int64_t)kmers.size(), prune, pen, window, sink_contig, &fo, &st) != MGTA_OK)
            die("mgta_astar_batch: %s", mgta_last_error());
        fclose(out);
        mgta_hmm_free(fw); mgta_hmm_free(rv);
        logf("Done %s: time %.4lf (%lld expansions, %.1f ms on device; %lld searches grew in place, %lld run again, %lld resumed passes, pool %.1f of %.1f GB, "
             "reserve %.2f of %.1f GB; largest search %lld nodes / %lld expansions%s)",
             gene.name.c_str(), now_s() - tg, (long long)st.n_expansions, st.ms_total, (long long)st.n_grown, (long long)st.n_retries, (long long)st.n_resumes,
             st.pool_used / 1e9, st.pool_bytes / 1e9, st.reserve_used / 1e9, st.reserve_bytes / 1e9, (long long)st.max_search_nodes,
             (long long)st.max_search_expansions, st.order_abandoned ? "; the searches outgrew the pool: paths shared WITHOUT an order from there on" : "");
    };
    // the genes of the list one after the other (search.cpp:124).  MEGAGTA_SEARCH_LANES=2 searches two genes side by side on one
    // graph, each batch on its own context (stream + work memory) and half of the CUs (mgta_astar_batch_on): measured and NOT the
    // default -- rplB + nirK at 2 M reads took 18 s side by side (9.5 s and 17.7 s on the device) against 15.9 s one after the other:
    // a batch on half the CUs takes twice as long, the idle time of one gene's window is not there for the other to use.
    const std::vector<GeneEntry> genes = read_gene_list(argv[2]);
    int lanes = 1;
    if (const char *e = getenv("MEGAGTA_SEARCH_LANES")) lanes = std::max(1, std::min(2, atoi(e)));
    if (lanes == 1) {
        for (const GeneEntry &gene : genes) run_gene(gene, ctx);
    } else {
        mgta_ctx *ctx2 = g_sess.active && g_sess.ctx2 ? g_sess.ctx2 : mgta_ctx_create(0);
        if (!ctx2) die("%s", mgta_last_error());
        if (g_sess.active) g_sess.ctx2 = ctx2;
        mgta_ctx_set_search_share(ctx, 1, 2);
        mgta_ctx_set_search_share(ctx2, 1, 2);
        std::atomic<size_t> next{0};
        auto lane = [&](mgta_ctx *c) {
            for (size_t i; (i = next.fetch_add(1)) < genes.size();) run_gene(genes[i], c);
        };
        std::thread other(lane, ctx2);
        lane(ctx);
        other.join();
        mgta_ctx_set_search_share(ctx, 1, 1);
        mgta_ctx_set_search_share(ctx2, 1, 1);
        if (!g_sess.active) mgta_ctx_destroy(ctx2);
    }
    mgta_sdbg_free(g);
    ctx_put(ctx);
    return 0;
}

// ---- the two one-screen text filters of the driver's last step (host only) ------------------------------------------------
// FASTA/FASTQ records with name and comment as kseq.h splits them (name = header up to the first blank)
struct TextRecord { std::string name, comment, seq; bool has_comment; };
static bool next_text_record(FILE *f, std::string &pending, TextRecord &r) {
    static bool pending_cr = false;                                      // goes with `pending` (one input stream per process)
    bool cr = false;                                                     // the last line ended in "\r\n"
    auto getline = [&](std::string &out) {                               // one line without its end ('\n' or "\r\n"); false at the end of the file
        static char *lp = nullptr;
        static size_t cap = 0;
        ssize_t n = ::getline(&lp, &cap, f);
        out.clear();
        if (n <= 0) return false;
        if (lp[n - 1] == '\n') --n;
        out.assign(lp, (size_t)n);
        cr = !out.empty() && out.back() == '\r';
        if (cr) out.pop_back();
        return true;
    };
    std::string line;
    if (pending.empty()) {
        for (;;) {
            if (!getline(line)) return false;
            if (!line.empty() && (line[0] == '>' || line[0] == '@')) break;
        }
    } else { line = pending; pending.clear(); cr = pending_cr; }
    size_t sp = line.find_first_of(" \t");
    r.name = line.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
    r.has_comment = sp != std::string::npos || cr;                       // kseq.h: the '\r' of a bare "name\r\n" header is a blank, the comment is empty
    r.comment = sp != std::string::npos ? line.substr(sp + 1) : "";
    r.seq.clear();
    for (;;) {
        if (!getline(line)) return true;
        if (line.empty()) continue;
        if (line[0] == '>' || line[0] == '@') { pending = line; pending_cr = cr; return true; }
        if (line[0] == '+') {
            size_t got = 0;
            while (got < r.seq.size() && getline(line)) got += line.size();
            return true;
        }
        r.seq += line;
    }
}

// cat contigs.fa | megagta filterbylen <min_len>     (filter_by_len.cpp:33-61)
static int main_filterbylen(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "Usage: cat contigs.fa | %s <min_len>\n", argv[0]); return 1; }
    const unsigned min_len = (unsigned)atoi(argv[1]);
    std::map<long long, size_t> hist;
    std::string pending, comment = "(null)";                         // kseq never clears its comment buffer: a record without a
    TextRecord r;                                                      // comment prints the previous one, the first ones "(null)"
    while (next_text_record(stdin, pending, r)) {
        if (r.has_comment) comment = r.comment;
        if (r.seq.size() >= min_len) {
            ++hist[(long long)r.seq.size()];
            printf(">%s %s\n%s\n", r.name.c_str(), comment.c_str(), r.seq.c_str());
        }
    }
    double sum = 0;
    size_t n = 0;
    for (auto &kv : hist) { sum += 1.0 * kv.first * kv.second; n += kv.second; }
    long long n50 = 0;
    double acc = 0;
    for (auto it = hist.rbegin(); it != hist.rend(); ++it) { acc += (double)it->second * it->first; if (acc >= sum * 0.5) { n50 = it->first; break; } }
    fprintf(stderr, "%d contigs, total %lld bp, min %lld bp, max %lld bp, avg %d bp, N50 %lld bp\n", (int)n, (long long)sum,
            hist.empty() ? 0LL : hist.begin()->first, hist.empty() ? 0LL : hist.rbegin()->first, int((n ? sum / n : 0) + 0.5), n50);
    return 0;
}

// megagta translate <nucl_seq>     (translate.cpp:14-35): frame 0, standard code, lower case, any codon with a letter outside ACGT -> x
static int main_translate(int argc, char **argv) {
    if (argc == 1) { fprintf(stderr, "Usage: %s <nucl_seq> \n", argv[0]); return 1; }
    FILE *f = fopen(argv[1], "r");
    if (!f) die("cannot open %s", argv[1]);
    static const char *kCodon = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";
    auto code = [](char c) { switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                                          case 'T': case 't': case 'U': case 'u': return 3; default: return -1; } };
    std::string pending;
    TextRecord r;
    while (next_text_record(f, pending, r)) {
        std::string aa;
        for (size_t i = 0; i + 3 <= r.seq.size(); i += 3) {
            int a = code(r.seq[i]), b = code(r.seq[i + 1]), c = code(r.seq[i + 2]);
            char ch = (a < 0 || b < 0 || c < 0) ? 'X' : kCodon[16 * a + 4 * b + c];
            aa.push_back((char)tolower(ch));
        }
        printf(">%s\n%s\n", r.name.c_str(), aa.c_str());
    }
    fclose(f);
    return 0;
}

// megagta findstart <ref_seq> <read.lib.bin> <k_size> [num_threads=0] [contigs.fa]     (fast_kmer_filter.cpp:49-190)
static int main_findstart(int argc, char **argv) {
    if (argc == 1) {
        fprintf(stderr, "Usage: %s <ref_seq> <read.lib> <k_size> [num_threads=0]\n", argv[0]);
        return 1;
    }
    if (argc < 4) die("findstart: <ref_seq> <read.lib> <k_size> are required");
    for (int i = 1; i <= 2; ++i) {
        FILE *t = fopen(argv[i], "rb");
        if (!t) { fprintf(stderr, "File %s doesn't exist\n", argv[i]); return 1; }       // :58-64
        fclose(t);
    }
    const int k = atoi(argv[3]);
    if (k < 9 || k % 3 != 0 || k / 3 > 24) die("findstart: k_size = %d: a multiple of 3 in [9, 72] is required (k/3 residues, at most 24)", k);
    RssLine rss;
    const RefWords ref = load_reference_words(argv[1], k / 3);
    logf("reference kmer set size: %lld\n", (long long)ref.model_pos.size());
    PackedReads local;
    PackedReads::Mark mk;
    PackedReads &pr = lib_get(argv[2], "", argc > 5 ? argv[5] : "", false, local, mk);   // stored as buildgraph wants them: the scan handles both orders
    const uint64_t n_lib = mk.n_start ? mk.n_start - 1 : 0;
    const uint64_t n_reads = pr.start.size() - 1;
    logf("Processing %llu reads, %llu contigs\n", (unsigned long long)n_lib, (unsigned long long)(n_reads - n_lib));
    mgta_ctx *ctx = ctx_get();
    mgta_reads *rd = nullptr;
    if (mgta_reads_upload(ctx, pr.words.data(), pr.words.size(), pr.start.data(), n_reads, &rd) != MGTA_OK) die("%s", mgta_last_error());
    // room for the hits: a call that finds more than fit only counts them and the scan runs again, so start generously (16 B per hit:
    // sized for one hit per 200 bases, about twice what the synthetic sets give; round 1 started at 65536 and rescanned every library of more than a few thousand reads)
    std::vector<mgta_seed_hit> hits(std::max<size_t>(1u << 16, (size_t)(pr.start.back() / 200)));
    int64_t n_hits = 0;
    double ms = 0;
    for (;;) {
        if (mgta_findstart(ctx, rd, 1, k, ref.words.data(), (int64_t)ref.model_pos.size(), hits.data(), (int64_t)hits.size(), &n_hits, &ms) != MGTA_OK)
            die("%s", mgta_last_error());
        if (n_hits <= (int64_t)hits.size()) break;
        hits.resize((size_t)n_hits + 1024);
    }
    hits.resize((size_t)n_hits);
    logf("seed scan: %.3f ms on the device, %lld hits\n", ms, (long long)n_hits);
    // unique by nucleotide k-mer (:181-182); the reference then shuffles, any order is as good: sorted
    auto base_at = [&](uint64_t r, uint32_t fwd_pos) {                  // reversed storage -> base of the read as sequenced
        const uint64_t len = pr.start[r + 1] - pr.start[r], q = pr.start[r] + (len - 1 - fwd_pos);
        return (int)((pr.words[q >> 4] >> (30 - 2 * (q & 15))) & 3);
    };
    // a hit as its k-mer packed two bits per base, first base in the top bits (k <= 72: three words): sorting the packed words IS the
    // lexicographic order of the strings, and equal k-mers end up side by side (a std::map of strings took 5 s for 7 M hits)
    struct Seed { uint64_t w[3]; int32_t ref; int64_t contig; };      // contig: the lowest-numbered contig of the previous k that holds the k-mer, -1 = reads only
    std::vector<Seed> seeds(hits.size());
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)hits.size(); ++i) {
        const mgta_seed_hit &h = hits[(size_t)i];
        const uint32_t pos = h.pos_strand >> 1, len = (uint32_t)(pr.start[h.read + 1] - pr.start[h.read]);
        Seed sd{{0, 0, 0}, h.ref, h.read >= n_lib ? (int64_t)(h.read - n_lib) : -1};
        for (int j = 0; j < k; ++j) {
            const int b = (h.pos_strand & 1) ? 3 - base_at(h.read, len - 1 - (pos + (uint32_t)j)) : base_at(h.read, pos + (uint32_t)j);
            sd.w[j >> 5] |= (uint64_t)b << (62 - 2 * (j & 31));
        }
        seeds[(size_t)i] = sd;
    }
    // of equal k-mers: the reference position of the first hit in scan order (what the map kept), the lowest contig
    if (seeds.size() >= (1ull << 32)) die("findstart: too many hits (%zu)", seeds.size());
    std::vector<uint32_t> order(seeds.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (uint32_t)i;
    // (a total order -- ties broken by the scan index -- so the parallel sort gives the one result; 36 M hits at 100 M reads: 9 s on one thread)
    __gnu_parallel::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        const Seed &x = seeds[a], &y = seeds[b];
        if (x.w[0] != y.w[0]) return x.w[0] < y.w[0];
        if (x.w[1] != y.w[1]) return x.w[1] < y.w[1];
        if (x.w[2] != y.w[2]) return x.w[2] < y.w[2];
        return a < b;
    });
    FILE *cf_out = nullptr;
    // side file for `search` (MEGAGTA_CLUSTER_FILE): one line per seed, in the order of the seed lines: the contig of the previous k that
    // holds the k-mer (-1 = none).  Seeds of one contig lie on one stretch of one gene copy: see the chain mode of `megagta search`.
    if (const char *cf = getenv("MEGAGTA_CLUSTER_FILE")) {
        cf_out = fopen(cf, "w");
        if (!cf_out) die("cannot write %s", cf);
    }
    std::string nucl((size_t)k, 'A');
    for (size_t i = 0; i < order.size();) {
        const Seed &first = seeds[order[i]];
        int64_t contig = first.contig;
        size_t j = i + 1;
        for (; j < order.size(); ++j) {
            const Seed &o = seeds[order[j]];
            if (o.w[0] != first.w[0] || o.w[1] != first.w[1] || o.w[2] != first.w[2]) break;
            if (o.contig >= 0 && (contig < 0 || o.contig < contig)) contig = o.contig;
        }
        for (int q = 0; q < k; ++q) nucl[(size_t)q] = "ACGT"[(first.w[q >> 5] >> (62 - 2 * (q & 31))) & 3];
        printf("dump_gene_name\tdump_seq_name\tdump\t%s\ttrue\t%d\t%s\t%d\n", nucl.c_str(), 1, ref.prot[(size_t)first.ref].c_str(),
               ref.model_pos[(size_t)first.ref]);
        if (cf_out) fprintf(cf_out, "%lld\n", (long long)contig);
        i = j;
    }
    if (cf_out) fclose(cf_out);
    mgta_reads_free(rd);
    ctx_put(ctx);
    return 0;
}

// ---- denovo: main_assemble (assembler.cpp:60-167) --------------------------------------------------------------------------------
static int main_denovo(int argc, char **argv) {
    RssLine rss;
    std::string sdbg_name, out_prefix = "out";
    int max_tip_len = -1, no_bubble = 0, min_contig = 0;
    static struct option opts[] = {{"sdbg_name", required_argument, 0, 's'}, {"output_prefix", required_argument, 0, 'o'},
                                   {"num_cpu_threads", required_argument, 0, 't'}, {"max_tip_len", required_argument, 0, 1},
                                   {"no_bubble", no_argument, 0, 2}, {"min_standalone", required_argument, 0, 3},
                                   {"min_contig", required_argument, 0, 4}, {0, 0, 0, 0}};
    optind = 1;
    int ch;
    bool bad = false;
    while ((ch = getopt_long(argc, argv, "s:o:t:", opts, nullptr)) != -1) {
        switch (ch) {
        case 's': sdbg_name = optarg; break;
        case 'o': out_prefix = optarg; break;
        case 't': break;                      // threads: the device decides; the result is the reference's one-thread result
        case 1: max_tip_len = atoi(optarg); break;
        case 2: no_bubble = 1; break;
        case 3: break;                        // min_standalone is parsed and never used by the reference either (assembly_algorithms.cpp:92,128)
        case 4: min_contig = atoi(optarg); break;
        default: bad = true;
        }
    }
    if (bad || sdbg_name.empty()) {
        fprintf(stderr, "%s\nUsage: %s -s sdbg_name -o output_prefix\n", bad ? "unknown option" : "no succinct de Bruijn graph name!", argv[0]);
        return 1;
    }
    double t0 = now_s();
    mgta_ctx *ctx = ctx_get();
    int gk = 0;
    size_t n_edges = 0;
    mgta_sdbg *g = graph_get(ctx, sdbg_name, &gk, &n_edges);
    logf("Number of Edges: %lld; K value: %d (load %.3f s)", (long long)n_edges, gk, now_s() - t0);
    char *fasta = nullptr;
    uint64_t len = 0;
    mgta_denovo_stats st;
    if (mgta_denovo(g, max_tip_len, no_bubble, min_contig, &fasta, &len, &st) != MGTA_OK) die("mgta_denovo: %s", mgta_last_error());
    logf("Tips removed: %lld (%.1f ms); bubbles removed: %lld of %lld candidates in %lld rounds (%.1f ms); %lld simple paths, %lld contigs, "
         "total length %lld (%.1f ms)", (long long)st.n_tips, st.ms_tips, (long long)st.n_bubbles, (long long)st.n_bubble_candidates,
         (long long)st.n_bubble_rounds, st.ms_bubbles, (long long)st.n_paths, (long long)st.n_contigs, (long long)st.total_len, st.ms_unitigs);
    const std::string cpath = out_prefix + ".contigs.fa";
    const long long info_n = (long long)st.n_contigs, info_len = (long long)st.total_len;
    auto write_contigs = [cpath, info_n, info_len](const char *text, uint64_t n) {
        FILE *f = fopen(cpath.c_str(), "w");
        if (!f) die("cannot write %s", cpath.c_str());
        if (n && fwrite(text, 1, n, f) != n) die("short write to %s", cpath.c_str());
        if (fclose(f) != 0) die("short write to %s", cpath.c_str());
        // the .info file LAST: whoever finds it finds a complete contigs file beside it
        f = fopen((cpath + ".info").c_str(), "w");
        if (!f) die("cannot write %s.info", cpath.c_str());
        fprintf(f, "%lld %lld\n", info_n, info_len);                                    // assembler.cpp:162
        if (fclose(f) != 0) die("short write to %s.info", cpath.c_str());
    };
    (void)remove((cpath + ".info").c_str());                                            // (an earlier run's: it would vouch for a file being rewritten)
    if (g_sess.active && !getenv("MEGAGTA_SYNC_WRITES")) {
        // the worker: the text stays for the step that takes these contigs, the file is written behind it (joined by "sync" before the
        // driver's checkpoint says "assembled", and before anything reads the file)
        if (contigs_join(true) != 0) return 1;
        g_sess.ctext = fasta; g_sess.ctext_len = len; g_sess.ctext_path = cpath;
        g_sess.cwriter_done = false;
        g_sess.cwriter = std::thread([write_contigs, fasta, len]() {
            set_soft_die(true);
            try { write_contigs(fasta, len); }
            catch (const std::exception &e) { g_sess.cwriter_error = e.what(); }
            g_sess.cwriter_done = true;
        });
    } else {
        write_contigs(fasta, len);
        mgta_host_free(fasta);
    }
    mgta_sdbg_free(g);
    ctx_put(ctx);
    return 0;
}

static int dispatch(int argc, char **argv);

// megagta serve: requests on stdin, one per line: the sub-command's argv, tab separated; a field "<PATH" / ">PATH" redirects the
// step's stdin / stdout (filterbylen, translate, findstart).  Reply "DONE <exit code>" on the descriptor stdout had at start.  A step
// that dies takes the worker with it: the driver then reports the step as failed, as it does for a one-shot process.
static int main_serve() {
    g_sess.active = true;
    FILE *req = fdopen(dup(0), "r"), *rep = fdopen(dup(1), "w");
    if (!req || !rep) die("serve: cannot duplicate the standard descriptors");
    const int null_in = open("/dev/null", O_RDONLY);
    dup2(null_in, 0);                                                     // steps never read the request channel
    dup2(2, 1);                                                           // ... nor write into the reply channel: a step's stdout that no ">PATH" field
                                                                          // redirects (dumpversion, a stray printf) goes to stderr, never between the DONE lines
    char *line = nullptr;
    size_t cap = 0;
    ssize_t n;
    while ((n = getline(&line, &cap, req)) > 0) {
        while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
        if (n == 0) continue;
        std::vector<std::string> f;
        for (char *p = line, *e; ; p = e + 1) {
            e = strchr(p, '\t');
            f.emplace_back(p, e ? (size_t)(e - p) : strlen(p));
            if (!e) break;
        }
        if (f[0] == "quit") break;
        if (f[0] == "sync") {                                             // the files of the last buildgraph / denovo are on disk (or: why not)
            const int crc = contigs_join(false);
            fprintf(rep, "DONE %d\n", writer_join() | crc);
            fflush(rep);
            continue;
        }
        if (f[0] == "release") {                                          // hand the device memory back (another process is going to need it:
            graph_drop();                                                 // it reads the graph from the files, so they are complete first)
            const int wrc = writer_join() | contigs_join(true);
            if (g_sess.ctx) mgta_ctx_release_scratch(g_sess.ctx);
            if (g_sess.ctx2) mgta_ctx_release_scratch(g_sess.ctx2);
            fprintf(rep, "DONE %d\n", wrc);
            fflush(rep);
            continue;
        }
        std::string in_path, out_path;
        std::vector<char *> av;
        static char prog[] = "megagta";
        av.push_back(prog);
        for (std::string &x : f) {
            if (!x.empty() && x[0] == '<') in_path = x.substr(1);
            else if (!x.empty() && x[0] == '>') out_path = x.substr(1);
            else av.push_back(const_cast<char *>(x.c_str()));
        }
        av.push_back(nullptr);
        int save_in = -1, save_out = -1, rc = 1;
        fflush(stdout);
        if (!in_path.empty()) {
            int fd = open(in_path.c_str(), O_RDONLY);
            if (fd < 0) { fprintf(stderr, "serve: cannot open %s\n", in_path.c_str()); goto reply; }
            save_in = dup(0); dup2(fd, 0); close(fd);
            clearerr(stdin);
        }
        if (!out_path.empty()) {
            int fd = open(out_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (fd < 0) { fprintf(stderr, "serve: cannot write %s\n", out_path.c_str()); goto reply; }
            save_out = dup(1); dup2(fd, 1); close(fd);
        }
        rc = dispatch((int)av.size() - 1, av.data());
    reply:
        fflush(stdout);
        if (save_out >= 0) { dup2(save_out, 1); close(save_out); }
        if (save_in >= 0) { dup2(save_in, 0); close(save_in); clearerr(stdin); }
        fprintf(rep, "DONE %d\n", rc);
        fflush(rep);
    }
    graph_drop();
    const int wrc = writer_join() | contigs_join(true);
    if (g_sess.ctx2) mgta_ctx_destroy(g_sess.ctx2);
    if (g_sess.ctx) mgta_ctx_destroy(g_sess.ctx);
    return wrc;
}

int main(int argc, char **argv) {
    if (argc >= 2 && std::string(argv[1]) == "serve") return main_serve();
    return dispatch(argc, argv);
}

static int dispatch(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "Usage: %s <sub_program> [sub options]\n    sub-programs on the MI355X hot path:\n        buildgraph    build succinct de Bruijn graph\n"
                        "        denovo        tips, bubbles, contigs of an intermediate k\n        search        HMM-guided search of gene contigs\n        findstart     find starting kmers of the search\n        dumpversion   dump version\n", argv[0]);
        return 1;
    }
    std::string sub = argv[1];
    if (sub == "buildgraph") return main_buildgraph(argc - 1, argv + 1);
    if (sub == "search") return main_search(argc - 1, argv + 1);
    if (sub == "findstart") return main_findstart(argc - 1, argv + 1);
    if (sub == "denovo") return main_denovo(argc - 1, argv + 1);
    if (sub == "filterbylen") return main_filterbylen(argc - 1, argv + 1);
    if (sub == "translate") return main_translate(argc - 1, argv + 1);
    if (sub == "buildlib") {                                             // build_read_lib.cpp:8-20 (host only: file formats, no kernel)
        if (argc < 4) { fprintf(stderr, "Usage %s <read_lib_file> <out_prefix>\n", argv[1]); return 1; }
        RssLine rss;
        // the reads are packed on the device (mgta_reads_pack_text); the host inflates the files and cuts them into records.  Without a
        // device (or with MEGAGTA_BUILDLIB_HOST=1) the same bytes come from the host packer, and the log says so.
        const char *force_host = getenv("MEGAGTA_BUILDLIB_HOST");
        mgta_ctx *ctx = (force_host && atoi(force_host)) ? nullptr : (g_sess.active ? ctx_get() : mgta_ctx_create(0));
        if (!ctx) {
            logf("buildlib: packing reads on the host (%s)", (force_host && atoi(force_host)) ? "MEGAGTA_BUILDLIB_HOST" : mgta_last_error());
            build_read_lib(argv[2], argv[3]);
            return 0;
        }
        struct Pack {
            static bool run(void *user, const char *text, uint64_t n_bytes, const uint64_t *off, uint64_t n, std::vector<uint32_t> &out) {
                uint64_t words = 0;
                for (uint64_t r = 0; r < n; ++r) words += 1 + (off[r + 1] - off[r] + 15) / 16;
                out.resize(words);
                uint64_t got = 0;
                if (mgta_reads_pack_text(static_cast<mgta_ctx *>(user), text, n_bytes, off, n, out.data(), words, &got) != MGTA_OK) {
                    fprintf(stderr, "mgta_reads_pack_text: %s\n", mgta_last_error());
                    return false;
                }
                return got == words;
            }
        };
        double t0 = now_s();
        build_read_lib(argv[2], argv[3], &Pack::run, ctx);
        logf("buildlib: reads packed on the device (%.3f s)", now_s() - t0);
        ctx_put(ctx);
        return 0;
    }
    if (sub == "libdump") {      // host-only check of the read loaders (tests): writes what buildgraph / findstart would upload
        if (argc < 5) { fprintf(stderr, "Usage %s <read_lib_prefix> lib|bin <out_prefix> [assist.fa]\n", argv[1]); return 1; }
        PackedReads pr;
        if (std::string(argv[3]) == "bin") load_read_bin(std::string(argv[2]) + ".bin", true, pr);
        else load_read_lib(argv[2], true, pr);
        if (argc > 5) {
            if (env_int("MEGAGTA_LIBDUMP_TEXT", 0)) {                   // the in-memory route a worker takes for the contigs its `denovo` has just made
                FILE *tf = fopen(argv[5], "rb");
                if (!tf) die("cannot open %s", argv[5]);
                std::string text;
                char tb[1 << 16];
                for (size_t got; (got = fread(tb, 1, sizeof tb, tf)) > 0;) text.append(tb, got);
                fclose(tf);
                load_fasta_text(text.data(), text.size(), true, pr);
            } else load_assist_fasta(argv[5], true, pr);
        }
        pr.finish();
        FILE *fw = fopen((std::string(argv[4]) + ".words").c_str(), "wb"), *fs = fopen((std::string(argv[4]) + ".start").c_str(), "wb");
        if (!fw || !fs) die("cannot write %s.words / .start", argv[4]);
        fwrite(pr.words.data(), 4, pr.words.size(), fw);
        fwrite(pr.start.data(), 8, pr.start.size(), fs);
        fclose(fw); fclose(fs);
        printf("%zu %zu %d %llu\n", pr.start.size() - 1, pr.words.size(), pr.max_len, (unsigned long long)pr.n_short);
        return 0;
    }
    if (sub == "sdbgcopy") {     // host-only check of the graph file reader + writer (tests): <in_prefix> -> <out_prefix> (one .sdbg.0 file)
        if (argc < 4) { fprintf(stderr, "Usage %s <in_prefix> <out_prefix>\n", argv[1]); return 1; }
        EdgeStream s;
        double t0 = now_s();
        read_sdbg(argv[2], s);
        double t1 = now_s();
        write_sdbg(argv[3], s);
        logf("%zu records: read %.3f s, write %.3f s", s.recs.size(), t1 - t0, now_s() - t1);
        return 0;
    }
    if (sub == "sdbgmerge") {    // after a build over N GPUs: <prefix> <N> -> PREFIX.sdbg_info naming the N files (host only)
        if (argc < 4 || atoi(argv[3]) < 1) { fprintf(stderr, "Usage %s <sdbg_prefix> <num_parts>\n", argv[1]); return 1; }
        merge_sdbg_parts(argv[2], atoi(argv[3]));
        return 0;
    }
    if (sub == "hmmdump") {      // host-only check of the HMMER3 text parser + heuristic (tests): the tables `search` uploads, as hex doubles
        if (argc != 3) { fprintf(stderr, "Usage: megagta hmmdump <file.hmm>\n"); return 1; }
        ProfileHmm hm;
        if (!parse_hmm(argv[2], hm)) die("cannot open HMM %s", argv[2]);
        const size_t M1 = (size_t)hm.M + 1;
        printf("M %d\nA %d\nalpha", hm.M, hm.A);
        for (int c = 0; c < 127; ++c) printf(" %d", hm.alpha[c]);
        printf("\ncompo");
        for (int j = 0; j < hm.A; ++j) printf(" %a", hm.compo[j]);
        printf("\n");
        for (int k = 0; k <= hm.M; ++k) {
            printf("msc %d", k);
            for (int j = 0; j < hm.A; ++j) printf(" %a", hm.msc[(size_t)k * hm.A + j]);
            printf("\ntsc %d", k);
            for (int t = 0; t < 7; ++t) printf(" %a", hm.tsc[(size_t)t * M1 + k]);
            printf("\nmaxm %d %a\nh %d %a %a %a\n", k, hm.max_match[k], k, hm.h[k], hm.h[M1 + k], hm.h[2 * M1 + k]);
        }
        return 0;
    }
    if (sub == "searchplan") {   // <n_seeds>...: the window and cost term `search` would take for batches of that many seeds (host only)
        for (int i = 2; i < argc; ++i) {
            int window = 0, rate = 0, rate2 = 0;
            long long knee = 0;
            search_plan((size_t)atoll(argv[i]), &window, &rate, &knee, &rate2);
            printf("%s %d %d %lld %d\n", argv[i], window, rate, knee, rate2);
        }
        return 0;
    }
    if (sub == "dumpversion") { printf("%s\n", mgta_version()); return 0; }
    fprintf(stderr, "sub-command '%s' is not built here (buildlib, buildgraph, denovo, findstart, search, filterbylen, translate are): run it with the reference's megagta binary\n", sub.c_str());
    return 1;
}
