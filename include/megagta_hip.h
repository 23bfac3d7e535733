/* megagta_hip.h — C ABI of libmegagta_hip.so: the MI355X (gfx950) implementation of MegaGTA's
 * HMM-guided succinct-de-Bruijn-graph assembly hot path.
 *
 * This is the in-process successor of the reference's dead GPU plug-point
 *     lv2_gpu_sort(uint32_t *lv2_substrings, uint32_t *permutation, int words_per_substring,
 *                  int64_t lv2_num_items, void *key1, void *key2, void *val1, void *val2)
 *     alloc_gpu_buffers(...) / free_gpu_buffers(...)
 * (call sites cx1_read2sdbg_s1.cpp:308,619-620,945 and cx1_read2sdbg_s2.cpp:391,697-698,926; the
 * functions themselves exist nowhere in the reference tree).  Instead of sorting one lv2 batch per
 * call, the whole read -> SdBG-edge pipeline, the graph and the A* search live on the device.
 *
 * Conventions: extern "C"; plain pointers and sizes; every function returns 0 on success and a
 * negative MGTA_E* code on failure (mgta_last_error() gives the message of the calling thread's
 * last failure); no exceptions cross the boundary; one mgta_ctx per GPU (thread-compatible).
 * There is NO CPU fallback: without a usable HIP device every entry point fails.
 */
#ifndef MEGAGTA_HIP_H_
#define MEGAGTA_HIP_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MGTA_OK 0
#define MGTA_EINVAL (-1)      /* bad argument */
#define MGTA_EHIP (-2)        /* HIP runtime error / no device */
#define MGTA_ENOMEM (-3)      /* device memory exhausted */
#define MGTA_EUNSUPPORTED (-4)/* feature of the reference not built yet (fails loudly, never silently) */
#define MGTA_ESINK (-5)       /* caller's sink returned non-zero */
#define MGTA_EOVERFLOW (-6)   /* per-search arena exhausted after all retries */
#define MGTA_EINTERNAL (-7)   /* a device-side protocol gave up (bounded wait expired); the call produced nothing */

#define MGTA_NUM_BUCKETS 65536 /* kNumBuckets, cx1_read2sdbg.h:64 (8-character key prefix) */

typedef struct mgta_ctx mgta_ctx;
typedef struct mgta_reads mgta_reads;
typedef struct mgta_sdbg mgta_sdbg;
typedef struct mgta_hmm mgta_hmm;

const char *mgta_last_error(void);
const char *mgta_version(void);

mgta_ctx *mgta_ctx_create(int device_id);            /* NULL on failure (see mgta_last_error) */
void mgta_ctx_destroy(mgta_ctx *);
/* memory the build may use on the device; 0 = 90 % of what is free (cf. --host_mem/--mem_flag, build_graph.cpp:40-47) */
int mgta_ctx_set_mem_limit(mgta_ctx *, uint64_t bytes);
int mgta_ctx_device_memory(mgta_ctx *, uint64_t *free_bytes, uint64_t *total_bytes);   /* of the context's device, now (either may be NULL) */
/* on: a build also leaves the WHOLE edge stream of its bucket range on the device when memory forces it into several bucket-range
 * passes (each pass is appended to a stream buffer), so that mgta_sdbg_load_resident can hand the graph of a whole-range build to
 * `denovo` / `search` without the disk or host round trip of `<prefix>.sdbg.*` (succinct_dbg.cpp:595-723) at any input size, and
 * mgta_sdbg_export_records_device hands a rank's shard (a bucket sub-range) to the all-gather in one piece.  off (default): only the
 * last pass stays.  Turning it off frees the buffer. */
int mgta_ctx_keep_stream(mgta_ctx *, int on);
/* on = 2: as 1, and the records and tip labels of a pass are NOT copied to the host for the sink (it is called with recs = tips = NULL,
 * counts and large multiplicities as usual): the caller takes the whole stream afterwards with the calls below.  `megagta buildgraph`
 * in the driver's worker: the graph is packed from the resident stream, the step ends, and a host thread downloads the stream and
 * writes PREFIX.sdbg.* (sdbg_multi_io.h:83-187) behind the step that follows.
 * mgta_sdbg_stream_detach: the whole stream of the last build leaves the context (which forgets it); it stays in device memory until
 * mgta_stream_free.  mgta_stream_download: records (uint16) and tip label words (uint32) into the caller's host memory, from any host
 * thread (a HIP stream and staging buffers of its own; nothing of the context is touched).  mgta_stream_free: on the context's thread. */
typedef struct mgta_stream mgta_stream;
int mgta_sdbg_stream_detach(mgta_ctx *, mgta_stream **out);
int mgta_stream_sizes(const mgta_stream *, uint64_t *n_recs, uint64_t *n_tip_words);
int mgta_stream_download(mgta_stream *, uint16_t *recs, uint32_t *tips);
void mgta_stream_free(mgta_stream *);
/* frees the grow-only work memory the context keeps between calls (build pool, search pool); a stream left in the build pool by a
 * single-pass build is dropped with it (one kept by mgta_ctx_keep_stream stays) */
int mgta_ctx_release_scratch(mgta_ctx *);
/* diagnostic switch (bit mask): 1 = sort every key with global LSD passes only (also the fallback for oversized segments);
 * 2 = the segment-local sort runs LSD passes over every remaining digit instead of finishing short runs by comparison
 *     (also the fallback for tiles whose runs are long) */
int mgta_ctx_set_full_lsd(mgta_ctx *, int on);
/* diagnostic, host only (no device needed): the global sort passes a build of `n_items` keys of `words_per_key` 32-bit words over the
 * buckets [bucket_begin, bucket_end) takes: *n_passes 8-bit digits taken below *skip_bits leading bits that every key of the range shares
 * once (bucket_begin << 16) is subtracted from key word 0 (0 for a whole-range build); the segment-local finish then sees segments of
 * equal leading 8 * n_passes + skip_bits bits.  Honours MGTA_SORT_BIAS. */
int mgta_sort_plan(uint64_t n_items, int words_per_key, uint32_t bucket_begin, uint32_t bucket_end, int *n_passes, int *skip_bits);
/* Shared-cache searches (mgta_astar_batch with cache_mode = B >= 1): the path found by seed j after c_j node expansions is seen by
 * exactly the seeds >= j + B + c_j / expansions_per_seed.  0 (default) = no cost term: seed i sees the seeds <= i - B, and every
 * later seed waits for the longest unfinished search.  > 0: a search that has already run r expansions cannot become visible to the
 * seeds below j + B + r / expansions_per_seed any more, so those start without waiting for it.  Either way the result is a function
 * of (seed order, B, expansions_per_seed) only, never of timing. */
int mgta_ctx_set_search_cost_rate(mgta_ctx *, int expansions_per_seed);   /* < 0 (down to -64): seeds per expansion, i.e. the path is
                                                                            * seen from seed j + B + c_j * |value| on */
/* The same with a CONCAVE cost term: c_j expansions delay the path of seed j by cost(c_j) = c_j / expansions_per_seed seeds while
 * c_j <= knee_expansions and by knee / expansions_per_seed + (c_j - knee) / expansions_per_seed_beyond seeds beyond (knee 0 = one rate;
 * expansions_per_seed_beyond >= expansions_per_seed >= 1).  The first search of a gene copy on a large graph runs for millions of
 * expansions: with one rate its path stayed invisible for as many seeds and every seed of that copy among them explored the copy cold
 * again; beyond the knee the delay grows slowly, the seeds right behind an ordinary search still start without waiting for it.  The
 * result is a function of (seed order, B, the three numbers) only.  (The reference has no such rule: its threads share term_nodes by
 * timing, search.cpp:182-189; B = 1 without a cost term is its one-thread run.) */
int mgta_ctx_set_search_cost_curve(mgta_ctx *, int expansions_per_seed, uint64_t knee_expansions, int expansions_per_seed_beyond);
/* Work memory of the searches.  The reference's node pool, open list and hash maps grow without bound (pool_st.h:43,
 * hash_table_st.h:559-568); here every search slot owns a base arena of 1 << log2_base_nodes nodes (0 = default 12; 7..20) and a search
 * that outgrows it takes further chunks from a device-side pool of pool_bytes (0 = sized from the number of searches in flight), its
 * hash table being re-built at twice the size when half full.  When the pool itself runs dry: an independent search (cache_mode 0 / -1)
 * that has waited in vain is run again by the host with fewer searches at a time (mgta_astar_stats.n_retries); under the ordered-commit
 * window (cache_mode B >= 1) a starved search gives its memory back and starts again IN PLACE -- its slot keeps holding the window, the
 * lowest running seed never yields and has a reserve of its own (an eighth of the memory on top of the pool; with an explicit pool_bytes
 * an eighth OF it) -- so the result stays a function of (seed order, B, cost rate) whatever starved when; only if the lowest seed outgrows
 * the reserve as well does the pass give up, and the batch RESUMES behind its commit frontier (everything that has ended is final; the
 * caches stay) with a larger share in the reserve: 1/2, then 7/8, then one search at a time (mgta_astar_stats.n_resumes, a note on
 * stderr).  Small values exercise these paths on small inputs (tests). */
int mgta_ctx_set_search_arena(mgta_ctx *, int log2_base_nodes, uint64_t pool_bytes);

/* ------------------------------------------------------------------------------------------------
 * Read ingestion (`megagta buildlib`: SequenceManager::ReadShortReads + WriteBinarySequences, sequence_manager.cpp:109-216,375-410;
 * base codes sequence_package.h:67-69).  `text` = the sequence characters of a batch of reads back to back (the host has inflated the
 * file and cut it into records), read i = text[offsets[i] .. offsets[i+1]).  bin_words receives the batch's share of PREFIX.bin: per
 * read uint32 length + ceil(length / 16) words, 2 bits per base (base j of a word at bits 30 - 2j, zero padded, forward orientation,
 * A0 C1 G2 T3, N -> G, any other byte -> A); *n_words_out = words written (sum of 1 + ceil(len / 16)).
 * ------------------------------------------------------------------------------------------------ */
int mgta_reads_pack_text(mgta_ctx *, const char *text, uint64_t n_bytes, const uint64_t *offsets /* [n_reads + 1] */, uint64_t n_reads,
                         uint32_t *bin_words, uint64_t capacity_words, uint64_t *n_words_out);

/* ------------------------------------------------------------------------------------------------
 * SdBG construction  (replaces CX1::run() with the s2 plug-ins: cx1.h:443-623,
 * s2_lv0_calc_bucket_size / s2_lv1_fill_offset / s2_lv2_extract_substr_ / lv2_cpu_radix_sort_st /
 * output_: cx1_read2sdbg_s2.cpp:252-315,475-677,742-835; lv2_cpu_sort.h:133-150)
 * ------------------------------------------------------------------------------------------------ */

/* One call per bucket range [bucket_begin,bucket_end), ranges ascending and disjoint: the logical
 * edge stream in bucket order, exactly the arguments SdbgWriter::write receives
 * (sdbg_multi_io.h:83-112).  recs[i] = w | last<<4 | tip<<5 | min(mult,255)<<8; `large` holds the
 * full 16-bit multiplicities of the records with mult > 254, `tips` words_per_tip words per tip
 * record, both in stream order.  Buffers are host memory owned by the library, valid during the call. */
typedef int (*mgta_edge_sink)(void *user, int32_t bucket_begin, int32_t bucket_end,
                              const int64_t *bucket_counts /* [bucket_end-bucket_begin][3]: records, large, tips */,
                              const uint16_t *recs, int64_t n_recs, const uint16_t *large, int64_t n_large,
                              const uint32_t *tips, int64_t n_tip_words);

typedef struct mgta_build_stats {
    int32_t k, words_per_key, words_per_tip, n_passes;
    int64_t n_reads, n_kmers;        /* n_kmers = sum max(0,len-k): (k+1)-mer occurrences fed to stage 2 */
    int64_t n_items;                 /* sort items generated */
    int64_t n_edges, n_tips, n_large;
    int64_t n_sort_launches;         /* launches of the dominant kernel (radix scatter) */
    double ms_total;                 /* device time, reads resident -> last record in device memory */
    double ms_count, ms_gen, ms_sort, ms_emit, ms_d2h;
    double ms_sort_scatter;          /* summed duration of the radix scatter launches (HIP events) */
    double ms_local_sort;            /* duration of the segment-local (LDS) finishing sort */
    int64_t n_big_segments;          /* key segments deferred by the tiled LDS sort (sorted alone in LDS, or by global passes) */
    int64_t n_lsd_tiles;             /* LDS tiles whose runs were too long to finish by comparison (LSD passes over every digit) */
    uint64_t bytes_peak;             /* device bytes allocated at the peak */
    double ms_stage1;                /* min_count >= 2: solid-edge counting + mercy edges, before (and not part of) ms_total */
} mgta_build_stats;

/* a1: packed reads as `buildgraph` holds them — every read REVERSED (cx1_read2sdbg_s1.cpp:97,117),
 * 2 bits/base, base j of word at bits 30-2j (sequence_package.h:126-129), reads concatenated;
 * start_idx[n_reads+1] in bases (sequence_package.h:44).  Copies host -> device. */
int mgta_reads_upload(mgta_ctx *, const uint32_t *packed_seq, uint64_t n_words, const uint64_t *start_idx,
                      uint64_t n_reads, mgta_reads **out);
/* adopt buffers that already live on this device (e.g. torch tensors); not freed by mgta_reads_free */
int mgta_reads_adopt_device(mgta_ctx *, const uint32_t *d_packed_seq, uint64_t n_words,
                            const uint64_t *d_start_idx, uint64_t n_reads, mgta_reads **out);
void mgta_reads_free(mgta_reads *);

/* a2-a7: reads resident -> edge stream.  min_count = 1 (reference `-m 1`): every position solid.  min_count >= 2 runs
 * stage 1 first (cx1_read2sdbg_s1.cpp: (k+1)-mer counting, solid marks; need_mercy adds mercy edges, s2.cpp:106-250).
 * n_short_reads: reads [n_short_reads, n_reads) are assist sequences (always solid, s2.cpp:276).
 * sink may be NULL (records stay on the device, e.g. for timing). */
int mgta_sdbg_build_resident(mgta_ctx *, const mgta_reads *, uint64_t n_short_reads, int k, int min_count,
                             int need_mercy, int32_t bucket_begin, int32_t bucket_end /* this GPU's share of the 65536 buckets */,
                             mgta_edge_sink sink, void *user, mgta_build_stats *stats);
/* (k+1)-mer multiplicity histogram of the last min_count >= 2 build: hist[i] = number of distinct (k+1)-mers seen i times
 * (i = 65535: that or more) — what s1_post_proc accumulates into PREFIX.counting (cx1_read2sdbg_s1.cpp:905-930). */
int mgta_sdbg_last_counting(mgta_ctx *, int64_t *hist /* [65536] */);
/* Records of the LAST build pass are still on the device: copy them (device -> device) into a caller-owned device
 * buffer (e.g. a torch tensor that is then all-gathered over RCCL).  d_dst = NULL only queries *n_records. */
int mgta_sdbg_export_records_device(mgta_ctx *, void *d_dst, uint64_t capacity_bytes, uint64_t *n_records);
/* convenience: upload + build + free */
int mgta_sdbg_build(mgta_ctx *, const uint32_t *packed_seq, uint64_t n_words, const uint64_t *start_idx,
                    uint64_t n_reads, uint64_t n_short_reads, int k, int min_count, int need_mercy,
                    mgta_edge_sink sink, void *user, mgta_build_stats *stats);

/* ------------------------------------------------------------------------------------------------
 * Succinct de Bruijn graph on the device (replaces SuccinctDBG::LoadFromMultiFile/init and the
 * rank/select indexes: succinct_dbg.cpp:595-723, succinct_dbg.h:62-86, rank_and_select.h)
 * ------------------------------------------------------------------------------------------------ */
/* From the logical edge stream (all buckets): recs[size], bucket_items[65536], tip labels. */
int mgta_sdbg_load(mgta_ctx *, int k, const uint16_t *recs, int64_t size, const int64_t *bucket_items,
                   const uint32_t *tips, int64_t n_tip_words, int words_per_tip, mgta_sdbg **out);
/* row f-4 (succinct_dbg.cpp:595-723 without the disk round trip): the graph of the edge stream the LAST
 * mgta_sdbg_build* call of this context left on the device (one pass over all 65536 buckets), read where it lies — the records
 * never visit the host.  MGTA_EINVAL when there is no such stream (no build yet, a bucket sub-range, several passes). */
int mgta_sdbg_load_resident(mgta_ctx *, mgta_sdbg **out);
/* The same load from the files `buildgraph` wrote: PREFIX.sdbg_info + PREFIX.sdbg.0 .. N-1 (SdbgReader + LoadFromMultiFile,
 * sdbg_multi_io.h:201-417, succinct_dbg.cpp:595-723).  The host maps the files and copies them to the device; the variable-length records
 * are parsed there, one bucket per lane.  What every rank of a multi-GPU search and every one-shot `megagta denovo|search` calls.
 * The records are never on the device all at once: they are decoded range by range (at most MGTA_LOAD_RANGE_RECORDS of them, default 2^32 =
 * 8 GB) and the 64-edge lines each range completes are packed at once, so the peak is the graph (2 B per edge) + its rank prefix sums
 * (1.2 B per edge) + one range -- the 63 G-edge graph of a 1 G-read set loads into 288 GB (round 4 held all records beside the lines). */
int mgta_sdbg_load_files(mgta_ctx *, const char *prefix, mgta_sdbg **out);
void mgta_sdbg_free(mgta_sdbg *);
int64_t mgta_sdbg_size(const mgta_sdbg *);
int mgta_sdbg_k(const mgta_sdbg *);                 /* the graph's k (node length), -1 for NULL */
/* batched navigation (test hook + building block of the search): for each edge id the valid
 * outgoing edges in the reference's order (descending id) [succinct_dbg.cpp:78-97];
 * outdeg[i] = -1 for an invalid edge.  out4 = n x 4 int64 (unused slots -1). Host pointers. */
int mgta_sdbg_outgoing(mgta_sdbg *, const int64_t *edges, int64_t n, int64_t *out4, int8_t *outdeg);
/* batched IndexBinarySearchEdge over (k+1)-symbol strings (symbols 1..4) [succinct_dbg.cpp:427-549]. */
/* the validity bits as they are now (SuccinctDBG::invalid_, succinct_dbg.h:117-131): bit e of words[e / 64]; ceil(size / 64) words.
 * Set for tips and $ edges after a load, and for everything `mgta_denovo` removed. */
int mgta_sdbg_invalid_bits(mgta_sdbg *, uint64_t *words);
int mgta_sdbg_index_edges(mgta_sdbg *, const uint8_t *seqs /* n x (k+1) */, int64_t n, int64_t *edge_ids);

/* ------------------------------------------------------------------------------------------------
 * Seed finder (replaces the read scan of `megagta findstart`, fast_kmer_filter.cpp:108-176,193-215):
 * every window of k nucleotides (k a multiple of 3, k/3 <= 24) of every read, on both strands, whose translation is one of
 * the n_ref reference words.  A word = its residues in the code of prot_kmer.h:31-43 (ARNDCQEGHILKMFPSTWYV = 0..19, '*' = 20),
 * 5 bits each, first residue highest: ref_words[2i] = the first min(12, k/3) residues, ref_words[2i+1] = the rest
 * (kmer.h:66-84); of equal words the first one wins (insert_unique, fast_kmer_filter.cpp:88).
 * reads_reversed: the reads were uploaded reversed, as mgta_sdbg_build wants them (cx1_read2sdbg_s1.cpp:97) — one upload serves
 * both.  A hit names the read, the strand (0 = as sequenced, 1 = reverse complement), the window start inside that strand's
 * string (ProcessSequenceMulti's nucl_pos) and the reference word.  At most `cap` hits are stored; *n_hits is the number found
 * (call again with a larger buffer when it exceeds cap).  Order of the hits is unspecified (the reference shuffles its output).
 * ---------------------------------------------------------------------------------------------- */
typedef struct mgta_seed_hit {
    uint64_t read;
    uint32_t pos_strand;             /* window start << 1 | strand */
    int32_t ref;                     /* index into ref_words */
} mgta_seed_hit;
int mgta_findstart(mgta_ctx *, const mgta_reads *reads, int reads_reversed, int k, const uint64_t *ref_words, int64_t n_ref,
                   mgta_seed_hit *hits, int64_t cap, int64_t *n_hits, double *ms_kernel /* optional */);

/* ------------------------------------------------------------------------------------------------
 * `megagta denovo` on a loaded graph (multi-k runs: the contigs of k feed `buildgraph --assist_seq` of the next k and `findstart`):
 * RemoveTips (assembly_algorithms.cpp:76-183), PopBubbles (:245-301, branch_group.cpp:22-141), then the maximal simple paths
 * written as contigs (main_assemble, assembler.cpp:98-167; UnitigGraph::InitFromSdBG with a file, unitig_graph.cpp:80-150,208-303).
 * The reference's loops race between threads; the result here is that of its ONE-thread run, byte for byte, computed in parallel
 * on the device (see denovo.hip).  CONSUMES the validity bits of the graph (as the reference does): do not search it afterwards.
 * max_tip_len: -1 = 2k, 0 = keep tips; min_contig: shortest contig written (the driver passes the next k + 1).
 * *fasta receives the malloc'd text of PREFIX.contigs.fa (">k{K}_{id} flag={f} multi={%.4lf} len={L}\n{seq}\n" per contig, ids in
 * ascending end-edge order = the one-thread order); free it with mgta_host_free.  PREFIX.contigs.fa.info is "n_contigs total_len\n".
 * ------------------------------------------------------------------------------------------------ */
typedef struct mgta_denovo_stats {
    int64_t n_tips, n_bubbles;               /* what the reference logs: tips removed, bubbles popped */
    int64_t n_bubble_candidates, n_bubble_rounds;
    int64_t n_paths, n_unitig_sweeps;
    int64_t n_contigs, total_len;
    float ms_tips, ms_bubbles, ms_unitigs;
} mgta_denovo_stats;
int mgta_denovo(mgta_sdbg *, int max_tip_len, int no_bubble, int min_contig, char **fasta, uint64_t *fasta_len,
                mgta_denovo_stats *stats /* optional */);
void mgta_host_free(void *);

/* ------------------------------------------------------------------------------------------------
 * Profile HMM tables (parsed on the host exactly like Parser::readHMM, hmmer3b_parser.h:19-177;
 * heuristic like MostProbablePath, most_probable_path.h:48-118)
 * ------------------------------------------------------------------------------------------------ */
int mgta_hmm_load(mgta_ctx *, int M, int A, const double *msc /* [(M+1)*A] */, const double *tsc /* [7*(M+1)] */,
                  const double *max_match /* [M+1] */, const double *h /* [3*(M+1)] */,
                  const int32_t *alpha /* [127] residue letter -> column */, mgta_hmm **out);
void mgta_hmm_free(mgta_hmm *);

/* ------------------------------------------------------------------------------------------------
 * Batched HMM-guided A* (replaces the OMP seed loop of search(): search.cpp:184-189,
 * HMMGraphSearch::search/astarSearch: hmm_graph_search.h:60-343, NodeEnumerator::enumerateNodes:
 * node_enumerator.h:65-246)
 * ------------------------------------------------------------------------------------------------ */
typedef struct mgta_astar_side {
    int32_t ok, fval, length, state_no, state, partial;
    int64_t node_id, n_closed, n_expanded, n_opened;
    double real_score, score;
} mgta_astar_side;

typedef struct mgta_astar_stats {
    int64_t n_seeds, n_expansions, n_opened, n_retries;   /* n_retries: searches run again because the pool was exhausted (normally 0): by
                                                           * the host after the pass (independent searches) or in place (ordered window) */
    double ms_total, ms_kernel;
    int64_t n_grown, n_rehash, n_recycled;                /* searches that outgrew their base arena, hash tables re-built, chunks re-used */
    uint64_t pool_bytes, pool_used;                       /* device memory set aside for the searches / most of it in use at once */
    int64_t n_resumes;                                    /* ordered window only: passes that gave up (the lowest running search outgrew its
                                                           * reserve) and were resumed behind the commit frontier (normally 0) */
    uint64_t reserve_bytes, reserve_used;                 /* the lowest running search's reserve (last pass) / most of it ever in use */
    int64_t max_search_nodes, max_search_expansions;      /* the largest single search of the batch: nodes opened, nodes expanded */
    int64_t order_abandoned;                              /* ordered window only, and only with MEGAGTA_SEARCH_ALLOW_UNORDERED=1 in the environment: 1 = the
                                                           * searches in flight outgrew the pool and the batch went on sharing its paths WITHOUT an order
                                                           * (the reference's multi-thread behaviour); by default the order is held whatever it costs */
    int64_t n_cache_drops;                                /* shared-cache inserts that found no room within the probe limit (0 in every measured run;
                                                           * > 0: later seeds may have searched where they could have followed a cached path) */
    int64_t hmm_in_lds;                                   /* 1: the HMM tables were staged in LDS; 0: (M + 1)(A + 11) * 8 B beside the heap tops
                                                           * exceed the CU's 160 KB (models longer than ~400 columns): read from device memory */
    int64_t n_over_limit;                                 /* search sides that outgrew the library's page tables (2 GB per array, ~33 M nodes) and are
                                                           * reported as failed searches (ok = 0, no extension); named on stderr.  0 in every measured run */
    double ms_queue_drained;                              /* first pass: milliseconds from the kernel's start to the moment the last seed of the batch was
                                                           * TAKEN by a search slot; ms_kernel - ms_queue_drained = the tail in which the launch only finishes
                                                           * the searches in flight (a batch cannot end before its longest search does) */
} mgta_astar_stats;

/* sink gets one call per seed, in seed order: left (already reverse-complemented) + right halves. */
typedef int (*mgta_contig_sink)(void *user, int64_t seed_index, const char *left, int64_t left_len,
                                const char *right, int64_t right_len, const mgta_astar_side *right_side,
                                const mgta_astar_side *left_side);

/* kmers: n x (k+1) characters ACGT (any case), start_state[i] = model position - 1 (search.cpp:157).
 * cache_mode 0 = cold: every seed independent (empty term_nodes caches), embarrassingly parallel;
 * cache_mode B >= 1 = shared term_nodes caches (search.cpp:182) with an ordered-commit window: the search of seed j
 *   sees exactly the paths found by seeds <= j - B, whatever the scheduling.  B = 1 is the reference's sequential
 *   run (`search ... 1`, bit-identical FASTA); larger B trades that equivalence for up to B searches in flight per
 *   direction while staying deterministic. */
int mgta_astar_batch(mgta_sdbg *, const mgta_hmm *fwd, const mgta_hmm *rev, const char *kmers,
                     const int32_t *start_state, int64_t n, int prune_len, double low_cov_penalty,
                     int cache_mode, mgta_contig_sink sink, void *user, mgta_astar_stats *stats);
/* cache_mode -1 = shared caches WITHOUT any ordering: every search sees whatever paths have been inserted when it looks (what the
 * reference's `search` with more than one thread does, search.cpp:182-189).  Least work and no waiting, but which of several equally
 * good paths a seed takes depends on timing: not the default anywhere; MEGAGTA_CACHE_WINDOW=-1 selects it in `megagta search`. */
/* The same batch run with the stream and the work memory of `run` (another context of the graph's device) instead of the graph's own:
 * two host threads can search two genes of one gene_list side by side on one graph (search.cpp:124 loops over the genes one after the
 * other).  mgta_ctx_set_search_share(run, 1, 2) on both contexts gives each batch half of the CUs.  Results do not depend on it.
 * (Measured on two genes of 76 k / 103 k seeds: no faster than one after the other -- `megagta search` keeps the genes sequential
 * unless MEGAGTA_SEARCH_LANES=2.) */
int mgta_astar_batch_on(mgta_ctx *run, mgta_sdbg *, const mgta_hmm *fwd, const mgta_hmm *rev, const char *kmers,
                        const int32_t *start_state, int64_t n, int prune_len, double low_cov_penalty,
                        int cache_mode, mgta_contig_sink sink, void *user, mgta_astar_stats *stats);
/* The same batch with its results in flat arrays instead of one call-back per seed: contig i = (*contigs)[offsets[i] .. offsets[i + 1]) =
 * left + lower-cased seed k-mer (k + 1 characters) + right, the sequence line `search` writes (hmm_graph_search.h:60-81).  *contigs is
 * malloc'd (mgta_host_free); offsets [n + 1] and sides [2 n] (right, left per seed; may be NULL) are the caller's. */
int mgta_astar_batch_packed(mgta_sdbg *, const mgta_hmm *fwd, const mgta_hmm *rev, const char *kmers, const int32_t *start_state, int64_t n,
                            int prune_len, double low_cov_penalty, int cache_mode, char **contigs, uint64_t *offsets, mgta_astar_side *sides,
                            mgta_astar_stats *stats);
int mgta_ctx_set_search_share(mgta_ctx *, int num, int den);     /* this context's search batches use num/den of the CUs (default 1/1) */

/* ---- measurement: the device's random 128-byte line ceiling -------------------------------------------------------------------------
 * Not part of the reference's interface: the yardstick the search's roofline is priced against (bench.py `search.roofline`).  The A*
 * expansion (hmm_graph_search.h:191-343) and the succinct-graph walks under it (succinct_dbg.cpp:78-97, rank_and_select.h:153-280)
 * are chains of random 128-byte line reads.  Every configuration is run once over a table of `table_bytes` (>= 8 GB to leave every
 * cache behind) with the kernels' own access shape: groups of 8 lanes read one aligned line each (16 B per lane); a wavefront carries
 * `groups` groups, each with `unroll` independent lines in flight, `waves_per_cu` wavefronts per CU.  dependent = 1 chases pointers
 * (the next line's index is read from the line just fetched): ns_per_step is then the loaded latency of one dependent line;
 * dependent = 2: the same chase with a store to another random line in every step (a store in front of a dependent fetch): 16 bytes
 * per lane = the whole line; dependent = 3: one lane's 16 bytes = a partial line.
 * In: waves_per_cu 1..32, groups 1..8, unroll 1|2|4|8, dependent 0..3, steps >= 1.  Out: the rest. */
typedef struct mgta_line_probe {
    int32_t waves_per_cu, groups, unroll, dependent;
    uint64_t steps;                  /* line reads per group and chain */
    int32_t lines_in_flight_per_cu, pad_;
    uint64_t lines;                  /* lines read by the launch */
    double ms, gb_per_s, ns_per_step;
} mgta_line_probe;
int mgta_probe_random_lines(mgta_ctx *, uint64_t table_bytes, mgta_line_probe *cfg, int n_cfg);

#ifdef __cplusplus
}
#endif
#endif /* MEGAGTA_HIP_H_ */
