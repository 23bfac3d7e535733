// probe.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A small driver of OUR OWN that #includes the reference's header-only classes from
// /root/reference/src (never copied) and prints the quantities the reference computes but never
// writes to disk, so that golden vectors for the oracle restatement can be captured in this
// container:
//
//   probe hmm    <file.hmm>                               parsed tables + A* heuristic (hex doubles)
//   probe graph  <sdbg_prefix> <n_queries> <seed>         bit-vector checksums + navigation answers
//   probe index  <sdbg_prefix> <kmers.txt>                IndexBinarySearchEdge of (k+1)-mers
//   probe astar  <sdbg_prefix> <fwd.hmm> <rev.hmm> <seeds.txt> <prune> <low_cov_pen> <cold|warm>
//
// Reference entry points exercised: Parser::readHMM (hmmer3b_parser.h:19), MostProbablePath
// (most_probable_path.h:18), SuccinctDBG::LoadFromMultiFile (succinct_dbg.cpp:595), OutgoingEdges
// (succinct_dbg.cpp:78), IndexBinarySearchEdge (:530), HMMGraphSearch::astarSearch
// (hmm_graph_search.h:132) and partialResultFromGoal (:83), in the order search() uses them (:60-81).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <queue>
#include <sstream>
#include <string>
#include <vector>
#include <set>
#include <cmath>
#include <limits>
#include <map>

// the A* bookkeeping containers (closed / pool_) are private; the probe needs their sizes.
#define private public
#include "succinct_dbg.h"
#include "hmmer3b_parser.h"
#include "most_probable_path.h"
#include "node_enumerator.h"
#include "hmm_graph_search.h"
#undef private

static uint64_t splitmix64(uint64_t &x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static uint64_t fnv1a(const void *p, size_t n) {
    const unsigned char *c = (const unsigned char *)p;
    uint64_t h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ULL; }
    return h;
}

static int cmd_hmm(int argc, char **argv) {
    std::ifstream f(argv[2]);
    ProfileHMM hmm(true);
    Parser::readHMM(f, hmm);
    MostProbablePath mpp(hmm);
    int M = hmm.modelLength(), A = hmm.alphabetLength();
    printf("M %d\nA %d\n", M, A);
    printf("alpha");
    for (int c = 0; c < 127; ++c) printf(" %d", hmm.alpha_mapping[c]);
    printf("\n");
    printf("compo");
    for (int j = 0; j < A; ++j) printf(" %a", hmm.compo[j]);
    printf("\n");
    for (int k = 0; k <= M; ++k) {
        printf("msc %d", k);
        for (int j = 0; j < A; ++j) printf(" %a", hmm.emissions[k][j][0]);
        printf("\n");
        printf("isc %d", k);
        for (int j = 0; j < A; ++j) printf(" %a", hmm.emissions[k][j][1]);
        printf("\n");
        printf("tsc %d", k);
        for (int t = 0; t < 7; ++t) printf(" %a", hmm.transitions[t][k]);
        printf("\n");
        printf("maxm %d %a\n", k, hmm.max_match_emissions[k]);
        printf("h %d %a %a %a\n", k, mpp.computeHeuristicCost('m', k), mpp.computeHeuristicCost('i', k),
               mpp.computeHeuristicCost('d', k));
    }
    return 0;
}

static int cmd_graph(int argc, char **argv) {
    SuccinctDBG dbg;
    dbg.LoadFromMultiFile(argv[2], false);
    int64_t nq = atoll(argv[3]);
    uint64_t seed = strtoull(argv[4], NULL, 10);
    int64_t n = dbg.size;
    size_t words_w = (n + 15) / 16, words_b = (n + 63) / 64;
    printf("size %lld\nk %d\n", (long long)n, dbg.kmer_k);
    printf("f");
    for (int i = 0; i < 6; ++i) printf(" %lld", dbg.f_[i]);
    printf("\n");
    printf("num_tips %lld words_per_tip %d\n", (long long)dbg.num_tip_nodes_, dbg.uint32_per_tip_nodes_);
    printf("fnv_w %016llx\n", (unsigned long long)fnv1a(dbg.w_, words_w * 8));
    printf("fnv_last %016llx\n", (unsigned long long)fnv1a(dbg.last_, words_b * 8));
    printf("fnv_tip %016llx\n", (unsigned long long)fnv1a(dbg.is_tip_, words_b * 8));
    printf("fnv_invalid %016llx\n", (unsigned long long)fnv1a(dbg.invalid_, words_b * 8));
    printf("fnv_multi1 %016llx\n", (unsigned long long)fnv1a(dbg.is_multi_1_, words_b * 8));
    printf("fnv_tiplabels %016llx\n",
           (unsigned long long)fnv1a(dbg.tip_node_seq_, (size_t)dbg.num_tip_nodes_ * dbg.uint32_per_tip_nodes_ * 4));
    for (int64_t q = 0; q < nq; ++q) {
        int64_t e = (int64_t)(splitmix64(seed) % (uint64_t)n);
        int64_t out[4];
        int od = dbg.OutgoingEdges(e, out);
        printf("q %lld w %d last %d tip %d valid %d multi1 %d od %d", (long long)e, (int)dbg.GetW(e), (int)dbg.IsLast(e),
               (int)dbg.IsTip(e), (int)dbg.IsValidEdge(e), (int)dbg.IsMulti1(e), od);
        for (int i = 0; i < od; ++i) printf(" %lld", (long long)out[i]);
        // rank / select answers at this position (semantics: rank_and_select.h:153,220,492,560)
        printf(" rl %lld", (long long)dbg.rs_last_.Rank(e));
        for (int c = 0; c < 9; ++c) printf(" %lld", (long long)dbg.rs_w_.Rank(c, e));
        int64_t r = dbg.rs_last_.Rank(e);
        printf(" sl %lld", (long long)dbg.rs_last_.Select(r - 1));
        if (dbg.IsValidEdge(e)) {
            printf(" fwd %lld", (long long)dbg.Forward(e));
            uint8_t lab[SuccinctDBG::kMaxKmerK + 1];
            dbg.Label(e, lab);
            printf(" label ");
            for (int i = 0; i < dbg.kmer_k; ++i) putchar("$ACGT"[lab[i]]);
            int64_t in[4];
            int id = dbg.IncomingEdges(e, in);
            printf(" id %d", id);
            for (int i = 0; i < id; ++i) printf(" %lld", (long long)in[i]);
        }
        printf("\n");
    }
    return 0;
}

static int cmd_index(int argc, char **argv) {
    SuccinctDBG dbg;
    dbg.LoadFromMultiFile(argv[2], false);
    HMMGraphSearch::setUp();
    std::ifstream f(argv[3]);
    std::string line;
    while (std::getline(f, line)) {
        if ((int)line.size() < dbg.kmer_k + 1) { printf("%s -2\n", line.c_str()); continue; }
        uint8_t seq[SuccinctDBG::kMaxKmerK + 2];
        for (int i = 0; i < dbg.kmer_k + 1; ++i) seq[i] = HMMGraphSearch::dna_map[(int)line[i]];
        printf("%s %lld\n", line.c_str(), (long long)dbg.IndexBinarySearchEdge(seq));
    }
    return 0;
}

static void print_goal(const char *tag, AStarNode &goal, HMMGraphSearch &s, const std::string &seq, bool ok) {
    // `partial` is never initialised on most nodes in the reference (a_star_node.h:31) -> not printed.
    // ok==false: astarSearch returned before touching goal (default-constructed node, fields indeterminate).
    printf(" %s ok %d real %a score %a fval %d len %d state_no %d state %c node %lld closed %zu seq %s", tag, (int)ok,
           ok ? goal.real_score : 0.0, ok ? goal.score : 0.0, ok ? goal.fval : 0, ok ? (int)goal.length : 0,
           ok ? (int)goal.state_no : -1, ok ? goal.state : '-', ok ? (long long)goal.node_id : -1LL, s.closed.size(),
           seq.empty() ? "." : seq.c_str());
}

static int cmd_astar(int argc, char **argv) {
    SuccinctDBG dbg;
    dbg.LoadFromMultiFile(argv[2], false);
    HMMGraphSearch::setUp();
    std::ifstream f1(argv[3]), f2(argv[4]);
    ProfileHMM fwd(true), rev(true);
    Parser::readHMM(f1, fwd);
    Parser::readHMM(f2, rev);
    MostProbablePath fh(fwd), rh(rev);
    int prune = atoi(argv[6]);
    double pen = atof(argv[7]);
    bool warm = std::string(argv[8]) == "warm";
    NodeEnumerator fe(fwd, fh, pen), re(rev, rh, pen);
    HMMGraphSearch s(prune);
    s.constructPool();

    std::ifstream sf(argv[5]);
    std::string line, col[8];
    HashMapST<AStarNode, AStarNode> *tn = new HashMapST<AStarNode, AStarNode>, *tnr = new HashMapST<AStarNode, AStarNode>;
    int idx = 0;
    while (std::getline(sf, line)) {
        std::istringstream iss(line);
        for (int i = 0; i < 8; ++i) iss >> col[i];
        std::transform(col[3].begin(), col[3].end(), col[3].begin(), ::tolower);   // search.cpp:156
        std::string kmer = col[3];
        int start_state = std::stoi(col[7]) - 1;                                       // search.cpp:157
        if (!warm) { delete tn; delete tnr; tn = new HashMapST<AStarNode, AStarNode>; tnr = new HashMapST<AStarNode, AStarNode>; }
        // same call sequence as HMMGraphSearch::search (hmm_graph_search.h:60-81)
        AStarNode *g1 = s.pool_->construct(), *g2 = s.pool_->construct();
        std::string right, left;
        printf("seed %d %s %d", idx, kmer.c_str(), start_state);
        // astarSearch clears `closed` itself (hmm_graph_search.h:202) except on its early return
        // (:193-197); clearing here only keeps the printed count from going stale in that case.
        s.closed.clear();
        bool ok1 = s.astarSearch(fwd, start_state, kmer, dbg, true, fe, *g1, *tn);
        s.partialResultFromGoal(*g1, true, right, *tn);
        print_goal("R", *g1, s, right, ok1);
        int lstate = rev.modelLength() - start_state - (int)kmer.size() / 3;
        s.closed.clear();
        bool ok2 = s.astarSearch(rev, lstate, kmer, dbg, false, re, *g2, *tnr);
        s.partialResultFromGoal(*g2, false, left, *tnr);
        print_goal("L", *g2, s, left, ok2);
        s.deleteAStarNodes();
        std::string left_rc = left;
        s.RevComp(left_rc);
        printf(" contig %s%s%s\n", left_rc.c_str(), kmer.c_str(), right.c_str());
        ++idx;
    }
    return 0;
}

static int cmd_codon() {
    // codon.h:5-209: forward and reverse-complement codon tables, index c1,c2,c3 with A0 C1 G2 T3
    printf("fwd ");
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4; ++c) putchar(Codon::codonTable[a][b][c]);
    printf("\nrc ");
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4; ++c) putchar(Codon::rc_codonTable[a][b][c]);
    printf("\n");
    // libseq translation used for the start k-mer (hmm_graph_search.h:137-148)
    printf("libseq ");
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4; ++c) {
        std::string cod; cod += "ACGT"[a]; cod += "ACGT"[b]; cod += "ACGT"[c];
        seq::NTSequence nts = seq::NTSequence("", "", cod);
        seq::AASequence aa = seq::AASequence::translate(nts.begin(), nts.begin() + 3);
        printf("%s", aa.asString().c_str());
    }
    printf("\n");
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && std::string(argv[1]) == "codon") return cmd_codon();
    if (argc < 3) { fprintf(stderr, "usage: probe hmm|graph|index|astar ...\n"); return 2; }
    std::string c = argv[1];
    if (c == "hmm") return cmd_hmm(argc, argv);
    if (c == "graph" && argc >= 5) return cmd_graph(argc, argv);
    if (c == "index" && argc >= 4) return cmd_index(argc, argv);
    if (c == "astar" && argc >= 9) return cmd_astar(argc, argv);
    fprintf(stderr, "bad arguments\n");
    return 2;
}
