/* mgta_oracle.h — C ABI of the CPU ORACLE.   *** TEST INFRASTRUCTURE, NOT PRODUCT CODE ***
 *
 * A plain, single-purpose CPU restatement of the MegaGTA hot path,
 * written from the reference's behaviour, every function citing the reference file:line it follows.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (megagta_amd/, libmegagta_hip.so) never links, imports or calls it.
 *
 * Pinning: checked against golden vectors captured in the build container from the compiled
 * reference itself (oracle/_ref/megagta + oracle/_ref/probe, recipe oracle/ref/Makefile) — see
 * tests/golden/make_golden.py and tests/test_oracle_golden.py.  The reference's own test-suite
 * pins nothing for this path.
 */
#ifndef MGTA_ORACLE_H_
#define MGTA_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NUM_BUCKETS 65536

/* ---- SdBG logical edge stream (what buildgraph emits, in bucket order) ---------------------- */
typedef struct orc_stream orc_stream;
/* stage-2 build, -c 1 semantics (every position solid)  [cx1_read2sdbg_s2.cpp:252-315,475-677,742-835] */
orc_stream *orc_sdbg_build(const uint32_t *packed_seq, uint64_t n_words, const uint64_t *start_idx,
                           uint64_t n_reads, int k, int n_threads);
/* -m min_count [--need_mercy]: stage 1 (solid (k+1)-mers per read position, mercy edges) then stage 2 over the solid runs
 * [cx1_read2sdbg_s1.cpp:177-951, cx1_read2sdbg_s2.cpp:106-250,276-297].  Reads >= n_short (assist sequences) are always solid.
 * counting[65536] (optional): number of distinct (k+1)-mers per multiplicity, what s1_post_proc turns into PREFIX.counting. */
orc_stream *orc_sdbg_build_solid(const uint32_t *packed_seq, uint64_t n_words, const uint64_t *start_idx, uint64_t n_reads,
                                 uint64_t n_short, int k, int min_count, int need_mercy, int n_threads, int64_t *counting,
                                 int64_t *n_mercy);
/* decode <prefix>.sdbg_info + <prefix>.sdbg.N into the logical stream  [sdbg_multi_io.h:240-382] */
orc_stream *orc_sdbg_read(const char *prefix);
void orc_stream_free(orc_stream *);
int orc_stream_k(const orc_stream *);
int orc_stream_words_per_tip(const orc_stream *);
int64_t orc_stream_num_edges(const orc_stream *);
int64_t orc_stream_num_tips(const orc_stream *);
int64_t orc_stream_num_large(const orc_stream *);
int64_t orc_stream_num_items_sorted(const orc_stream *);          /* sort items fed (build only) */
const int64_t *orc_stream_bucket_items(const orc_stream *);       /* [65536] records per bucket */
const uint16_t *orc_stream_records(const orc_stream *);           /* [num_edges] w|last<<4|tip<<5|min(m,255)<<8 */
const uint16_t *orc_stream_large(const orc_stream *);             /* [num_large] full multiplicities, stream order */
const uint32_t *orc_stream_tips(const orc_stream *);              /* [num_tips*words_per_tip] */

/* ---- succinct de Bruijn graph + rank/select navigation -------------------------------------- */
typedef struct orc_graph orc_graph;
orc_graph *orc_graph_from_stream(const orc_stream *);             /* LoadFromMultiFile(prefix,false) [succinct_dbg.cpp:595-723] */
void orc_graph_free(orc_graph *);
int64_t orc_graph_size(const orc_graph *);
int orc_graph_k(const orc_graph *);
const int64_t *orc_graph_f(const orc_graph *);                    /* [6] */
const uint64_t *orc_graph_w(const orc_graph *);
const uint64_t *orc_graph_last(const orc_graph *);
const uint64_t *orc_graph_tip(const orc_graph *);
const uint64_t *orc_graph_invalid(const orc_graph *);
const uint64_t *orc_graph_multi1(const orc_graph *);
const uint32_t *orc_graph_tip_labels(const orc_graph *);
int64_t orc_graph_num_tips(const orc_graph *);
int64_t orc_rank_last(const orc_graph *, int64_t pos);            /* #1 in [0..pos]      [rank_and_select.h:492] */
int64_t orc_select_last(const orc_graph *, int64_t r);            /* 0-based r-th one    [rank_and_select.h:560] */
int64_t orc_rank_w(const orc_graph *, int c, int64_t pos);        /* [rank_and_select.h:153] */
int64_t orc_select_w(const orc_graph *, int c, int64_t r);        /* [rank_and_select.h:220] */
int64_t orc_forward(const orc_graph *, int64_t e);                /* [succinct_dbg.h:155] */
int64_t orc_backward(const orc_graph *, int64_t e);               /* [succinct_dbg.h:166] */
int orc_outgoing(const orc_graph *, int64_t e, int64_t out[4]);   /* [succinct_dbg.cpp:78] */
int orc_incoming(const orc_graph *, int64_t e, int64_t in[4]);    /* [succinct_dbg.cpp:99] */
int orc_label(const orc_graph *, int64_t e, uint8_t *seq);        /* [succinct_dbg.cpp:503] */
int64_t orc_index_edge(const orc_graph *, const uint8_t *seq);    /* k+1 symbols 1..4 [succinct_dbg.cpp:427,530] */

/* ---- denovo (tips, bubbles, unitigs) as the reference's ONE-thread run ---------------------------
 * [assembler.cpp:98-167, assembly_algorithms.cpp:76-183,245-301, branch_group.cpp:22-141, unitig_graph.cpp:80-150,208-303]
 * Consumes the graph's validity bits.  Returns the malloc'd FASTA text (free with orc_free). */
char *orc_denovo(orc_graph *, int max_tip_len, int no_bubble, int min_contig, int64_t *n_contigs, int64_t *total_len,
                 int64_t *n_tips, int64_t *n_bubbles);
void orc_free(void *);
const uint64_t *orc_graph_invalid_now(const orc_graph *);

/* ---- profile HMM ------------------------------------------------------------------------------ */
typedef struct orc_hmm orc_hmm;
orc_hmm *orc_hmm_parse(const char *path);                         /* [hmmer3b_parser.h:19-201], normalized=true */
void orc_hmm_free(orc_hmm *);
int orc_hmm_M(const orc_hmm *);
int orc_hmm_A(const orc_hmm *);
const double *orc_hmm_msc(const orc_hmm *);                       /* [(M+1)*A] */
const double *orc_hmm_isc(const orc_hmm *);                       /* [(M+1)*A] */
const double *orc_hmm_tsc(const orc_hmm *);                       /* [7*(M+1)] transition-major MM MI MD IM II DM DD */
const double *orc_hmm_maxm(const orc_hmm *);                      /* [M+1] */
const double *orc_hmm_h(const orc_hmm *);                         /* [3*(M+1)] m,i,d  [most_probable_path.h:48-118] */
const int *orc_hmm_alpha(const orc_hmm *);                        /* [127] */

/* ---- HMM-guided A* ------------------------------------------------------------------------------ */
typedef struct {
    int32_t ok;            /* astarSearch() return value */
    int32_t fval, length, state_no;
    int32_t state;         /* 'm' 'i' 'd' or '-' */
    int32_t partial;       /* 1 = open list ran dry (inter-goal result) */
    int64_t node_id;
    int64_t n_closed;      /* closed-set insertions */
    int64_t n_expanded;    /* enumerateNodes calls on a valid node (metric: expansions) */
    int64_t n_opened;
    double real_score, score;
} orc_astar_result;

typedef struct orc_searcher orc_searcher;
orc_searcher *orc_searcher_new(const orc_graph *, const orc_hmm *fwd, const orc_hmm *rev, int prune_len,
                               double low_cov_penalty);
void orc_searcher_free(orc_searcher *);
void orc_searcher_clear_cache(orc_searcher *);                    /* drop the term_nodes caches (cold mode) */
/* windowed sharing: seed j sees the paths found by seeds <= j - window; 1 = the reference's sequential run */
void orc_searcher_set_window(orc_searcher *, int window);
/* cost term of the sharing rule: the path of seed j (c_j expansions) is seen by the seeds >= j + window + c_j / rate; 0 = none;
 * rate < 0: by the seeds >= j + window + c_j * |rate| */
void orc_searcher_set_cost_rate(orc_searcher *, int rate);
/* concave cost term (mgta_ctx_set_search_cost_curve): c / rate seeds up to `knee` expansions, knee / rate + (c - knee) / rate2 beyond */
void orc_searcher_set_cost_curve(orc_searcher *, int rate, int64_t knee, int rate2);
/* one seed = HMMGraphSearch::search [hmm_graph_search.h:60-81]; kmer is lower/upper-case ACGT of
 * length k+1; contig receives "<left><kmer><right>" (lower case).  Returns contig length or <0. */
int64_t orc_search_seed(orc_searcher *, const char *kmer, int start_state, orc_astar_result *right,
                        orc_astar_result *left, char *contig, int64_t contig_cap);

#ifdef __cplusplus
}
#endif
#endif
