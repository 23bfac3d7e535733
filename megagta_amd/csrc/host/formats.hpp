// formats.hpp — host-side file formats of the drop-in boundary (C++17, no reference code):
//   reads.lib.bin / .lib_info   sequence_manager.cpp:375-410, read_lib_functions-inl.h:216-261
//   assist FASTA + .info         cx1_read2sdbg_s1.cpp:104-134
//   .sdbg.N / .sdbg_info         sdbg_multi_io.h:34-417
//   HMMER3 text models           hmmer3b_parser.h:19-201, most_probable_path.h:48-118
//   gene_list.txt, *_starting_kmers.txt   search.cpp:105-162
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace mgta_host {

[[noreturn]] void die(const char *fmt, ...);
// A thread that has called set_soft_die(true) gets a StepFailure exception from die() instead of the process exiting under the feet of
// whatever step the main thread is in (the worker's background writer of PREFIX.sdbg.*).
struct StepFailure : std::runtime_error { using std::runtime_error::runtime_error; };
void set_soft_die(bool on);
void logf(const char *fmt, ...);   // stderr progress line, "    [file:line] ..." style is not required by the driver

// ---- packed reads ---------------------------------------------------------------------------------
struct PackedReads {
    std::vector<uint32_t> words;     // 2 bit / base, base j of a word at bits 30-2j, reads concatenated
    std::vector<uint64_t> start;     // [n+1] in bases
    uint64_t n_short = 0;            // reads that came from the library (the rest are assist sequences)
    int max_len = 0;
    void append(const uint8_t *codes, size_t n, bool reverse);
    void append_packed(const uint32_t *w, size_t n, bool reverse);   // n bases as the .bin files hold them (base j of a word at bits 30-2j)
    // the same result as n calls of append_packed() / of append() on the codes of the text, computed by all host threads: every sequence's
    // place in the bit stream follows from the lengths, the pieces are OR-ed into zeroed words (100 M reads: the one-thread loop took 6 s per
    // `buildgraph`, a k's contigs -- 4 GB of text -- 8 s)
    void append_packed_many(const uint32_t *const *w, const uint32_t *len, size_t n, bool reverse);
    void append_text_many(const char *const *seq, const uint32_t *len, size_t n, bool reverse);   // ACGT in any case, N -> G (sequence_package.h:67-69)
    void finish();
    // a loaded library kept between the steps of one process: mark() before sequences are appended for one step (assist contigs),
    // rewind() afterwards (finish() only flushes the last partial word, which rewind() takes back)
    struct Mark { size_t n_words, n_start; uint64_t acc, n_bases; int acc_bits, max_len; };
    Mark mark() const { return Mark{words.size(), start.size(), acc_, n_bases_, acc_bits_, max_len}; }
    void rewind(const Mark &m) { words.resize(m.n_words); start.resize(m.n_start); acc_ = m.acc; n_bases_ = m.n_bases; acc_bits_ = m.acc_bits; max_len = m.max_len; }
  private:
    template <class Put> void append_many(const uint32_t *len, size_t n, Put put);
    uint64_t acc_ = 0;
    int acc_bits_ = 0;
    uint64_t n_bases_ = 0;
    void push_bits(uint32_t v, int nbits) {                          // nbits <= 32, acc_bits_ < 32 on entry
        if (nbits == 0) return;
        acc_ = (acc_ << nbits) | (nbits == 32 ? (uint64_t)v : (uint64_t)(v & ((1u << nbits) - 1u)));
        acc_bits_ += nbits;
        if (acc_bits_ >= 32) { acc_bits_ -= 32; words.push_back((uint32_t)(acc_ >> acc_bits_)); acc_ &= (1ull << acc_bits_) - 1ull; }
    }
    void push2(unsigned c) {
        acc_ = (acc_ << 2) | c; acc_bits_ += 2; ++n_bases_;
        if (acc_bits_ == 32) { words.push_back((uint32_t)acc_); acc_ = 0; acc_bits_ = 0; }
    }
};
void load_read_lib(const std::string &prefix, bool reverse, PackedReads &out);          // ReadBinaryLibs
void load_assist_fasta(const std::string &path, bool reverse, PackedReads &out);        // s1.cpp:104-134
void load_read_bin(const std::string &bin_path, bool reverse, PackedReads &out);         // a bare reads.lib.bin, read to EOF (findstart)
void load_fastx(const std::string &path, bool reverse, PackedReads &out);                // FASTA / FASTQ (plain or .gz), N -> G (sequence_package.h:67-69)
// FASTA text in memory whose records are one header line + ONE sequence line each (what `denovo` writes): the same sequences load_fastx
// would append from the file, found and packed by all host threads
void load_fasta_text(const char *text, size_t len, bool reverse, PackedReads &out);
class FastxReader {                                                                      // kseq.h's record rules over zlib
  public:
    explicit FastxReader(const std::string &path);
    ~FastxReader();
    bool next(std::vector<uint8_t> &codes);                                              // false at end of file
    bool next_text(std::string &text);                                                   // the same record, its sequence characters appended to `text`
    FastxReader(const FastxReader &) = delete;
    FastxReader &operator=(const FastxReader &) = delete;
    struct Impl;
  private:
    Impl *p_;
};
// `megagta buildlib`, build_read_lib.cpp.  pack != nullptr: batches of sequence text are packed by it (the device: mgta_reads_pack_text)
typedef bool (*PackTextFn)(void *user, const char *text, uint64_t n_bytes, const uint64_t *offsets, uint64_t n_reads, std::vector<uint32_t> &bin_words);
void build_read_lib(const std::string &lib_file, const std::string &out_prefix, PackTextFn pack = nullptr, void *pack_user = nullptr);

// ---- findstart ----------------------------------------------------------------------------------------
struct RefWords {
    std::vector<uint64_t> words;     // [n][2]: residues 5 bits each, first residue highest; first min(12, k/3) | the rest
    std::vector<int> model_pos;      // model column of the first residue (1-based)
    std::vector<std::string> prot;   // lower case, as Kmer::decodePacked prints it
};
RefWords load_reference_words(const std::string &faa_path, int kaa);                     // fast_kmer_filter.cpp:81-91

// ---- SdBG files -------------------------------------------------------------------------------------
struct EdgeStream {
    int k = 0, words_per_tip = 0;
    std::vector<int64_t> bucket_items = std::vector<int64_t>(65536, 0), bucket_tips = std::vector<int64_t>(65536, 0),
                         bucket_large = std::vector<int64_t>(65536, 0);
    std::vector<uint16_t> recs, large;
    std::vector<uint32_t> tips;
};
void write_sdbg(const std::string &prefix, const EdgeStream &s);   // one file PREFIX.sdbg.0 + PREFIX.sdbg_info
// one rank's share of a build over several GPUs: the buckets [b_lo, b_hi) -> PREFIX.sdbg.<file_id> (+ PREFIX.sdbg_info.part<file_id> when `part`)
void write_sdbg(const std::string &prefix, const EdgeStream &s, int file_id, int b_lo, int b_hi, bool part);
void merge_sdbg_parts(const std::string &prefix, int n_parts);      // the parts of n_parts ranks -> PREFIX.sdbg_info (num_threads = n_parts)
void read_sdbg(const std::string &prefix, EdgeStream &s);

// ---- profile HMM --------------------------------------------------------------------------------------
struct ProfileHmm {
    std::string name;
    int M = 0, A = 0;
    int32_t alpha[127];
    std::vector<double> compo, msc, tsc, max_match, h;   // msc [(M+1)*A], tsc [7*(M+1)], h [3*(M+1)]
};
bool parse_hmm(const std::string &path, ProfileHmm &hm);

// ---- search inputs ------------------------------------------------------------------------------------
struct GeneEntry { std::string name, fwd_hmm, rev_hmm; };
std::vector<GeneEntry> read_gene_list(const std::string &path);                           // search.cpp:105-122
bool read_seeds(const std::string &path, std::vector<std::string> &kmers, std::vector<int32_t> &start_state);   // :146-162

}  // namespace mgta_host
