// formats.cpp — see formats.hpp
#include "formats.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include <cstring>
#include <fstream>
#include <map>
#include <limits>
#include <sstream>

namespace mgta_host {

static thread_local bool t_soft_die = false;
void set_soft_die(bool on) { t_soft_die = on; }
void die(const char *fmt, ...) {
    char msg[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof(msg), fmt, ap);
    va_end(ap);
    if (t_soft_die) throw StepFailure(msg);                            // (a background thread: the failure is reported by the step that waits for it)
    fprintf(stderr, "    [ERROR] %s\n", msg);
    fflush(stderr);
    exit(1);
}
void logf(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "    [megagta_amd] ");
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, "\n");
    va_end(ap);
    fflush(stderr);
}

// ----------------------------------------------------------------------------------------------------
void PackedReads::append(const uint8_t *codes, size_t n, bool reverse) {
    if (start.empty()) start.push_back(0);
    if (reverse) for (size_t i = n; i-- > 0;) push2(codes[i]);
    else for (size_t i = 0; i < n; ++i) push2(codes[i]);
    start.push_back(n_bases_);
    max_len = std::max<int>(max_len, (int)n);
}
// the same as append() of the unpacked codes, 16 bases at a time
void PackedReads::append_packed(const uint32_t *w, size_t n, bool reverse) {
    if (start.empty()) start.push_back(0);
    const size_t nw = (n + 15) / 16;
    const int tail = (int)(n - (nw ? (nw - 1) * 16 : 0));               // bases in the last word (1..16), 0 for an empty read
    auto rev16 = [](uint32_t x) {                                       // the 16 characters of a word in reverse order
        x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
        x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
        return __builtin_bswap32(x);
    };
    if (reverse) {
        if (nw) push_bits(rev16(w[nw - 1]), 2 * tail);                  // its characters end up in the low 2 * tail bits
        for (size_t j = nw; j-- > 1;) push_bits(rev16(w[j - 1]), 32);   // (nw = 0, an empty read: nothing -- `nw - 1` would wrap)
    } else {
        for (size_t j = 0; j + 1 < nw; ++j) push_bits(w[j], 32);
        if (nw) push_bits(tail == 16 ? w[nw - 1] : w[nw - 1] >> (32 - 2 * tail), 2 * tail);
    }
    n_bases_ += n;
    start.push_back(n_bases_);
    max_len = std::max<int>(max_len, (int)n);
}
// ---- many sequences at once, by all host threads ---------------------------------------------------------------------------------------
namespace {
inline uint32_t rev16(uint32_t x) {                                     // the 16 characters of a word in reverse order
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
}
struct BitOr {                                                          // the bit stream as push_bits() builds it: first bit = the top bit of word 0
    uint32_t *w;
    inline void put(uint64_t bitpos, uint32_t v, int nbits) const {     // v: nbits <= 32 bits, right-aligned
        if (nbits == 0) return;
        const size_t j = (size_t)(bitpos >> 5);
        const int off = (int)(bitpos & 31);
        const uint64_t x = (uint64_t)(nbits == 32 ? v : (v & ((1u << nbits) - 1u))) << (64 - nbits - off);
        const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
        if (hi) __atomic_fetch_or(&w[j], hi, __ATOMIC_RELAXED);         // (neighbouring sequences share their boundary words)
        if (lo) __atomic_fetch_or(&w[j + 1], lo, __ATOMIC_RELAXED);
    }
};
}  // namespace

template <class Put> void PackedReads::append_many(const uint32_t *len, size_t n, Put put) {
    if (start.empty()) start.push_back(0);
    if (n == 0) return;
    const size_t s0 = start.size();
    start.resize(s0 + n);
    uint64_t at = n_bases_;
    int ml = max_len;
    for (size_t i = 0; i < n; ++i) { at += len[i]; start[s0 + i] = at; ml = std::max<int>(ml, (int)len[i]); }
    max_len = ml;
    // the words that are complete now + the one the accumulator holds a part of + what the new bases need (+ one of slack: a piece may touch j + 1)
    const size_t full = words.size();
    const uint64_t bits_end = 2 * at;
    if (acc_bits_ > 0) words.push_back((uint32_t)(acc_ << (32 - acc_bits_)));
    words.resize((size_t)((bits_end + 31) / 32) + 1, 0u);
    (void)full;
    const BitOr bo{words.data()};
    const uint64_t base0 = n_bases_;
    const uint64_t *st = start.data() + s0;
#pragma omp parallel for schedule(dynamic, 4096)
    for (long long i = 0; i < (long long)n; ++i) put((size_t)i, bo, 2 * (i == 0 ? base0 : st[i - 1]));
    n_bases_ = at;
    words.pop_back();                                                   // the slack word
    const int rem = (int)(bits_end & 31);
    if (rem) { acc_ = (uint64_t)(words.back() >> (32 - rem)); acc_bits_ = rem; words.pop_back(); }
    else { acc_ = 0; acc_bits_ = 0; }
}

void PackedReads::append_packed_many(const uint32_t *const *w, const uint32_t *len, size_t n, bool reverse) {
    append_many(len, n, [&](size_t i, const BitOr &bo, uint64_t bit) {
        const uint32_t *s = w[i];
        const size_t nb = len[i], nw = (nb + 15) / 16;
        const int tail = (int)(nb - (nw ? (nw - 1) * 16 : 0));
        if (reverse) {
            if (nw) { bo.put(bit, rev16(s[nw - 1]), 2 * tail); bit += 2 * tail; }
            for (size_t j = nw; j-- > 1;) { bo.put(bit, rev16(s[j - 1]), 32); bit += 32; }   // (safe for nw = 0: buildlib keeps empty records)
        } else {
            for (size_t j = 0; j + 1 < nw; ++j) { bo.put(bit, s[j], 32); bit += 32; }
            if (nw) bo.put(bit, tail == 16 ? s[nw - 1] : s[nw - 1] >> (32 - 2 * tail), 2 * tail);
        }
    });
}

void PackedReads::append_text_many(const char *const *seq, const uint32_t *len, size_t n, bool reverse) {
    static const struct CodeTable {
        uint8_t t[256];
        CodeTable() {
            memset(t, 0, sizeof(t));                         // (as load_fastx: anything that is not a base reads as A)
            t[(int)'C'] = t[(int)'c'] = 1;
            t[(int)'G'] = t[(int)'g'] = t[(int)'N'] = t[(int)'n'] = 2;
            t[(int)'T'] = t[(int)'t'] = 3;
        }
    } code;
    append_many(len, n, [&](size_t i, const BitOr &bo, uint64_t bit) {
        const unsigned char *s = reinterpret_cast<const unsigned char *>(seq[i]);
        const size_t nb = len[i];
        for (size_t b = 0; b < nb; b += 16) {                           // 16 bases per piece
            const int m = (int)std::min<size_t>(16, nb - b);
            uint32_t v = 0;
            if (reverse) for (int q = 0; q < m; ++q) v = (v << 2) | code.t[s[nb - 1 - b - (size_t)q]];
            else for (int q = 0; q < m; ++q) v = (v << 2) | code.t[s[b + (size_t)q]];
            bo.put(bit, v, 2 * m);
            bit += 2 * (uint64_t)m;
        }
    });
}

void PackedReads::finish() {
    if (start.empty()) start.push_back(0);
    if (acc_bits_ > 0) { words.push_back((uint32_t)(acc_ << (32 - acc_bits_))); acc_ = 0; acc_bits_ = 0; }
    if (words.empty()) words.push_back(0);
}

void load_read_lib(const std::string &prefix, bool reverse, PackedReads &out) {
    std::ifstream info(prefix + ".lib_info");
    long long total_bases = 0, num_reads = 0;
    if (!(info >> total_bases >> num_reads)) die("cannot read %s.lib_info", prefix.c_str());
    FILE *f = fopen((prefix + ".bin").c_str(), "rb");
    if (!f) die("cannot open %s.bin", prefix.c_str());
    setvbuf(f, nullptr, _IOFBF, 1 << 20);
    out.words.reserve((size_t)total_bases / 16 + 16);
    out.start.reserve((size_t)num_reads + 2);
    // the file in pieces of ~256 MB (whole records): the records of a piece are found by one pass over their length words and appended by
    // all host threads (sequence_package.h:126-129 for the layout)
    std::vector<uint32_t> buf;
    std::vector<const uint32_t *> ptr;
    std::vector<uint32_t> lens;
    long long r = 0;
    size_t have = 0;                                                    // words of `buf` that hold unread file content
    const size_t piece = 64u << 20;                                     // words
    buf.resize(piece + 64);
    bool eof = false;
    while (r < num_reads) {
        if (!eof) {
            if (buf.size() < have + piece) buf.resize(have + piece);
            const size_t got = fread(buf.data() + have, 4, piece, f);
            have += got;
            eof = got < piece;
        }
        ptr.clear(); lens.clear();
        size_t at = 0;
        while (r + (long long)ptr.size() < num_reads && at < have) {
            const uint32_t len = buf[at];
            const size_t nw = (len + 15) / 16;
            if (at + 1 + nw > have) break;                              // the record continues in the next piece
            ptr.push_back(buf.data() + at + 1);
            lens.push_back(len);
            at += 1 + nw;
        }
        if (ptr.empty()) {
            if (eof) die("%s.bin: truncated at read %lld", prefix.c_str(), r);
            buf.resize(buf.size() * 2);                                 // (a record longer than a piece)
            continue;
        }
        out.append_packed_many(ptr.data(), lens.data(), ptr.size(), reverse);
        r += (long long)ptr.size();
        memmove(buf.data(), buf.data() + at, (have - at) * 4);
        have -= at;
    }
    fclose(f);
    out.n_short = (uint64_t)num_reads;
}

void load_read_bin(const std::string &bin_path, bool reverse, PackedReads &out) {
    FILE *f = fopen(bin_path.c_str(), "rb");
    if (!f) die("cannot open %s", bin_path.c_str());
    setvbuf(f, nullptr, _IOFBF, 1 << 20);
    std::vector<uint32_t> w;
    uint32_t len;
    uint64_t n = 0;
    while (fread(&len, 4, 1, f) == 1) {                                                      // until EOF, sequence_manager.cpp:375-410
        size_t nw = (len + 15) / 16;
        w.resize(nw);
        if (nw && fread(w.data(), 4, nw, f) != nw) die("%s: truncated at read %llu", bin_path.c_str(), (unsigned long long)n);
        out.append_packed(w.data(), len, reverse);
        ++n;
    }
    fclose(f);
    out.n_short = n;
}

void load_assist_fasta(const std::string &path, bool reverse, PackedReads &out) {
    std::ifstream info(path + ".info");
    long long ns = 0, nb = 0;
    if (!(info >> ns >> nb)) die("cannot read %s.info (num_seq num_bases)", path.c_str());   // s1.cpp:105-108
    load_fastx(path, reverse, out);
}

// ---- FASTA / FASTQ records, plain or gzip'ed (the reference reads them with kseq.h over zlib) -------------------
struct FastxReader::Impl {
    gzFile f = nullptr;
    std::string path, line, pending;     // pending: a header line already consumed
    bool have_pending = false, eof = false;
    std::vector<char> buf = std::vector<char>(1 << 20);              // inflated text, refilled by gzread; lines are cut out of it with memchr
    size_t pos = 0, len = 0;
    bool fill() {
        const int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n <= 0) return false;
        pos = 0; len = (size_t)n;
        return true;
    }
    // one line without its '\n' (and without a '\r' before it); false at the end of the file unless text without a final '\n' is left
    bool getline(std::string &out) {
        out.clear();
        for (;;) {
            if (pos == len && !fill()) return !out.empty();
            const char *p = buf.data() + pos;
            const char *nl = static_cast<const char *>(memchr(p, '\n', len - pos));
            if (nl) {
                out.append(p, (size_t)(nl - p));
                pos = (size_t)(nl - buf.data()) + 1;
                if (!out.empty() && out.back() == '\r') out.pop_back();
                return true;
            }
            out.append(p, len - pos);
            pos = len;
        }
    }
};

FastxReader::FastxReader(const std::string &path) : p_(new Impl) {
    p_->path = path;
    p_->f = gzopen(path.c_str(), "rb");
    if (!p_->f) die("cannot open %s", path.c_str());
    gzbuffer(p_->f, 1u << 20);
}
FastxReader::~FastxReader() {
    if (p_->f) gzclose(p_->f);
    delete p_;
}

// one record: a header line ('>' or '@'), sequence lines up to the next line that starts with '>', '@' or '+'; after '+' as many
// quality characters as the sequence has are skipped (kseq.h kseq_read).  Codes: sequence_package.h:67-69 (N -> G)
static bool fastx_next(FastxReader::Impl &r, std::vector<uint8_t> *codes_p, std::string *text_p);
bool FastxReader::next(std::vector<uint8_t> &codes) { codes.clear(); return fastx_next(*p_, &codes, nullptr); }
bool FastxReader::next_text(std::string &text) { return fastx_next(*p_, nullptr, &text); }
static bool fastx_next(FastxReader::Impl &r, std::vector<uint8_t> *codes_p, std::string *text_p) {
    static const struct CodeTable {
        uint8_t t[256];
        CodeTable() {
            memset(t, 0, sizeof(t));                         // anything else: 0 (the reference indexes an uninitialised table here)
            t[(int)'C'] = t[(int)'c'] = 1;
            t[(int)'G'] = t[(int)'g'] = t[(int)'N'] = t[(int)'n'] = 2;
            t[(int)'T'] = t[(int)'t'] = 3;
        }
    } code;
    if (r.eof) return false;
    const size_t text_at = text_p ? text_p->size() : 0;
    if (!r.have_pending) {                                   // look for the first header
        for (;;) {
            if (!r.getline(r.line)) { r.eof = true; return false; }
            if (!r.line.empty() && (r.line[0] == '>' || r.line[0] == '@')) break;
        }
    }
    r.have_pending = false;
    for (;;) {
        if (!r.getline(r.line)) { r.eof = true; return true; }
        if (r.line.empty()) continue;
        const char c = r.line[0];
        if (c == '>' || c == '@') { r.have_pending = true; return true; }
        if (c == '+') {
            const size_t n_seq = codes_p ? codes_p->size() : text_p->size() - text_at;
            size_t got = 0;
            while (got < n_seq && r.getline(r.line)) got += r.line.size();
            return true;
        }
        if (text_p) { text_p->append(r.line); continue; }
        std::vector<uint8_t> &codes = *codes_p;
        const size_t at = codes.size();
        codes.resize(at + r.line.size());
        for (size_t i = 0; i < r.line.size(); ++i) codes[at + i] = code.t[(unsigned char)r.line[i]];
    }
}

void load_fasta_text(const char *text, size_t len, bool reverse, PackedReads &out) {
    // records: '>' at the start of the text or right after a '\n'; the sequence = the line after the header line
    int nt = 1;
#ifdef _OPENMP
    nt = std::max(1, omp_get_max_threads());
#endif
    std::vector<std::vector<const char *>> seqs((size_t)nt);
    std::vector<std::vector<uint32_t>> lens((size_t)nt);
#pragma omp parallel for schedule(static, 1) num_threads(nt)
    for (int t = 0; t < nt; ++t) {
        const size_t lo = len * (size_t)t / (size_t)nt, hi = len * ((size_t)t + 1) / (size_t)nt;
        const char *p = text + lo, *const end = text + len, *const stop = text + hi;
        while (p < stop) {                                              // headers that START in [lo, hi)
            if (!(*p == '>' && (p == text || p[-1] == '\n'))) {
                const char *q = static_cast<const char *>(memchr(p, '\n', (size_t)(stop - p)));
                if (!q) break;
                p = q + 1;
                continue;
            }
            const char *h = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
            if (!h) break;                                              // a header without a sequence line: no record
            const char *s = h + 1;
            const char *e = static_cast<const char *>(memchr(s, '\n', (size_t)(end - s)));
            if (!e) e = end;
            size_t n = (size_t)(e - s);
            if (n && s[n - 1] == '\r') --n;
            seqs[(size_t)t].push_back(s);
            lens[(size_t)t].push_back((uint32_t)n);
            p = e < end ? e + 1 : end;
        }
    }
    size_t total = 0;
    for (auto &v : seqs) total += v.size();
    std::vector<const char *> all;
    std::vector<uint32_t> all_len;
    all.reserve(total); all_len.reserve(total);
    for (int t = 0; t < nt; ++t) { all.insert(all.end(), seqs[(size_t)t].begin(), seqs[(size_t)t].end()); all_len.insert(all_len.end(), lens[(size_t)t].begin(), lens[(size_t)t].end()); }
    out.append_text_many(all.data(), all_len.data(), total, reverse);
}

void load_fastx(const std::string &path, bool reverse, PackedReads &out) {
    FastxReader rd(path);
    std::vector<uint8_t> codes;
    while (rd.next(codes)) out.append(codes.data(), codes.size(), reverse);
}

// ---- buildlib (build_read_lib.cpp, read_lib_functions-inl.h:116-225, sequence_manager.cpp:109-216,375-410) ----------------
// read_lib_file: per library one free-text line, then `pe f1 f2` | `se f` | `interleaved f`.  PREFIX.bin: per read uint32 length +
// ceil(length / 16) words, 2 bit per base, base j of a word at bits 30-2j, zero padded, forward orientation, no N trimming;
// PREFIX.lib_info: "total_bases total_reads" then per library its text line and "from to max_read_len pe|se".
void build_read_lib(const std::string &lib_file, const std::string &out_prefix, PackTextFn pack, void *pack_user) {
    std::ifstream cfg(lib_file);
    if (!cfg.is_open()) die("File to open read_lib file: %s", lib_file.c_str());
    FILE *bin = fopen((out_prefix + ".bin").c_str(), "wb");
    if (!bin) die("cannot write %s.bin", out_prefix.c_str());
    struct Lib { std::string metadata; long long from, to; int max_len; bool pe; };
    std::vector<Lib> libs;
    long long total_reads = 0, total_bases = 0;
    std::vector<uint8_t> codes;
    std::vector<uint32_t> words;
    auto write_read = [&](const std::vector<uint8_t> &c, int &max_len) {
        const uint32_t len = (uint32_t)c.size();
        words.assign((len + 15) / 16, 0u);
        const uint8_t *cp = c.data();
        uint32_t i = 0;
        for (; i + 16 <= len; i += 16) {
            uint32_t w = 0;
            for (int j = 0; j < 16; ++j) w = (w << 2) | cp[i + j];
            words[i >> 4] = w;
        }
        for (; i < len; ++i) words[i >> 4] |= (uint32_t)cp[i] << (30 - 2 * (i & 15));
        fwrite(&len, 4, 1, bin);
        if (!words.empty()) fwrite(words.data(), 4, words.size(), bin);
        ++total_reads;
        total_bases += len;
        max_len = std::max(max_len, (int)len);
    };
    // device path: the sequence characters of a batch of reads back to back + their offsets; `pack` returns the batch's bytes of
    // PREFIX.bin (mgta_reads_pack_text).  The host keeps what belongs to the file: inflating it and cutting it into records.
    std::string batch;
    std::vector<uint64_t> batch_off{0};
    std::vector<uint32_t> batch_bin;
    constexpr size_t kBatchBytes = 256u << 20;
    auto flush_batch = [&]() {
        if (batch_off.size() == 1) return;
        if (!pack(pack_user, batch.data(), batch.size(), batch_off.data(), batch_off.size() - 1, batch_bin)) die("packing reads on the device failed");
        if (!batch_bin.empty() && fwrite(batch_bin.data(), 4, batch_bin.size(), bin) != batch_bin.size()) die("short write to %s.bin", out_prefix.c_str());
        batch.clear();
        batch_off.assign(1, 0);
    };
    auto end_read = [&](int &max_len) {                               // the read's characters are the tail of `batch`
        const uint64_t len = batch.size() - batch_off.back();
        if (len > 0xFFFFFFFFull) die("read longer than 2^32 bases");
        batch_off.push_back(batch.size());
        ++total_reads;
        total_bases += (long long)len;
        max_len = std::max(max_len, (int)len);
        if (batch.size() >= kBatchBytes) flush_batch();
    };
    std::string metadata, rest;
    while (std::getline(cfg, metadata)) {
        std::string type, f1, f2;
        if (!(cfg >> type)) break;
        Lib lib{metadata, total_reads, 0, 0, type != "se"};
        if (type == "pe") {
            if (!(cfg >> f1 >> f2)) die("pe library needs two files: %s", metadata.c_str());
            FastxReader r1(f1), r2(f2);
            std::vector<uint8_t> c2;
            for (;;) {
                if (pack) {
                    const bool a = r1.next_text(batch);
                    if (a) end_read(lib.max_len);
                    const bool b = r2.next_text(batch);
                    if (b) end_read(lib.max_len);
                    if (a != b) die("PE library files hold different numbers of reads: %s", metadata.c_str());
                    if (!a) break;
                    continue;
                }
                const bool a = r1.next(codes), b = r2.next(c2);
                if (a != b) die("PE library files hold different numbers of reads: %s", metadata.c_str());
                if (!a) break;
                write_read(codes, lib.max_len);
                write_read(c2, lib.max_len);
            }
        } else if (type == "se" || type == "interleaved") {
            if (!(cfg >> f1)) die("library needs a file: %s", metadata.c_str());
            FastxReader r1(f1);
            if (pack) { while (r1.next_text(batch)) end_read(lib.max_len); }
            else while (r1.next(codes)) write_read(codes, lib.max_len);
        } else {
            fprintf(stderr, "Cannot identify read library type %s\n", type.c_str());
            die("Valid types: pe, se, interleaved");
        }
        lib.to = total_reads - 1;
        if (lib.pe && (total_reads - lib.from) % 2 != 0) {
            fprintf(stderr, "PE library number of reads is odd: %lld!\n", total_reads - lib.from);
            die("File(s): %s", metadata.c_str());
        }
        logf("Lib %zu (%s): %s, %lld reads, %d max length\n", libs.size(), metadata.c_str(), type.c_str(), total_reads - lib.from, lib.max_len);
        libs.push_back(lib);
        std::getline(cfg, rest);                                     // the rest of the type line
    }
    if (pack) flush_batch();
    fclose(bin);
    FILE *info = fopen((out_prefix + ".lib_info").c_str(), "w");
    if (!info) die("cannot write %s.lib_info", out_prefix.c_str());
    fprintf(info, "%lld %lld\n", total_bases, total_reads);
    for (const Lib &l : libs) fprintf(info, "%s\n%lld %lld %d %s\n", l.metadata.c_str(), l.from, l.to, l.max_len, l.pe ? "pe" : "se");
    fclose(info);
}

// ----------------------------------------------------------------------------------------------------
// The records of the buckets [b_lo, b_hi) go to PREFIX.sdbg.<file_id>.  part = false: that is the whole graph, PREFIX.sdbg_info is
// written with it (one file).  part = true: one rank's share of a build over several GPUs: PREFIX.sdbg_info.part<file_id> keeps this
// file's bucket lines and counts until merge_sdbg_parts puts the index together (the reference's writer also deals the buckets to
// num_threads files, sdbg_multi_io.h:83-187: a reader does not care who wrote which).
void write_sdbg(const std::string &prefix, const EdgeStream &s, int file_id, int b_lo, int b_hi, bool part) {
    const std::string fname = prefix + ".sdbg." + std::to_string(file_id);
    FILE *f = fopen(fname.c_str(), "wb");
    if (!f) die("cannot write %s", fname.c_str());
    const std::string iname = part ? prefix + ".sdbg_info.part" + std::to_string(file_id) : prefix + ".sdbg_info";
    FILE *info = fopen(iname.c_str(), "w");
    if (!info) die("cannot write %s", iname.c_str());
    int64_t n_tips = s.words_per_tip ? (int64_t)s.tips.size() / s.words_per_tip : 0;
    if (part) fprintf(info, "%d %d %d %d %lld %lld %lld\n", s.k, s.words_per_tip, b_lo, b_hi, (long long)s.recs.size(), (long long)n_tips, (long long)s.large.size());
    else {
        fprintf(info, "k %d\nwords_per_tip_label %d\nnum_buckets %d\nnum_threads %d\n", s.k, s.words_per_tip, 65536, 1);
        fprintf(info, "total_size %lld\nnum_tips %lld\nlarge_multi %lld\n", (long long)s.recs.size(), (long long)n_tips, (long long)s.large.size());
    }
    std::vector<unsigned char> buf;
    size_t ri = 0, li = 0, ti = 0;
    long long off = 0;
    for (int b = part ? b_lo : 0; b < (part ? b_hi : 65536); ++b) {
        int64_t n = (b >= b_lo && b < b_hi) ? s.bucket_items[b] : 0;
        if (n == 0) { fprintf(info, "%d -1 0 0 0 0\n", b); continue; }
        const size_t worst = (size_t)n * (size_t)(4 + 4 * s.words_per_tip);      // record + large multiplicity + tip label
        if (buf.size() < worst) buf.resize(worst);
        unsigned char *q = buf.data();
        int64_t nt = 0, nl = 0;
        const uint16_t *rp = s.recs.data() + ri;
        for (int64_t i = 0; i < n; ++i) {
            const uint16_t it = rp[i];
            memcpy(q, &it, 2); q += 2;
            if ((it >> 8) == 255) { const uint16_t m = s.large[li++]; memcpy(q, &m, 2); q += 2; ++nl; }
            if ((it >> 5) & 1) {
                memcpy(q, &s.tips[ti], 4 * (size_t)s.words_per_tip); q += 4 * (size_t)s.words_per_tip;
                ti += s.words_per_tip; ++nt;
            }
        }
        ri += (size_t)n;
        const size_t bytes = (size_t)(q - buf.data());
        if (fwrite(buf.data(), 1, bytes, f) != bytes) die("write error on %s", fname.c_str());
        fprintf(info, "%d %d %lld %lld %lld %lld\n", b, file_id, off, (long long)n, (long long)nt, (long long)nl);
        off += (long long)bytes;
    }
    if (fclose(f) != 0 || fclose(info) != 0) die("write error on %s", fname.c_str());
}
void write_sdbg(const std::string &prefix, const EdgeStream &s) { write_sdbg(prefix, s, 0, 0, 65536, false); }

// PREFIX.sdbg_info from the parts of `n_parts` ranks (file r = PREFIX.sdbg.r holds the buckets of part r); the parts are removed
void merge_sdbg_parts(const std::string &prefix, int n_parts) {
    int k = -1, wpt = -1, next_b = 0;
    long long total = 0, tips = 0, large = 0;
    std::vector<std::string> lines;
    lines.reserve(65536);
    char *lp = nullptr;
    size_t cap = 0;
    for (int r = 0; r < n_parts; ++r) {
        const std::string pn = prefix + ".sdbg_info.part" + std::to_string(r);
        FILE *f = fopen(pn.c_str(), "r");
        if (!f) die("cannot open %s (rank %d of the build did not finish?)", pn.c_str(), r);
        int pk, pw, lo, hi;
        long long n, nt, nl;
        if (fscanf(f, "%d %d %d %d %lld %lld %lld\n", &pk, &pw, &lo, &hi, &n, &nt, &nl) != 7) die("%s: bad header", pn.c_str());
        if (r == 0) { k = pk; wpt = pw; }
        if (pk != k || pw != wpt || lo != next_b || hi < lo) die("%s: part %d covers buckets [%d, %d), expected to start at %d with k = %d", pn.c_str(), r, lo, hi, next_b, k);
        total += n; tips += nt; large += nl;
        for (int b = lo; b < hi; ++b) {
            if (getline(&lp, &cap, f) <= 0 || atoi(lp) != b) die("%s: bucket line %d missing", pn.c_str(), b);
            lines.emplace_back(lp);
        }
        next_b = hi;
        fclose(f);
    }
    free(lp);
    if (next_b != 65536) die("%s: the %d parts cover %d of 65536 buckets", prefix.c_str(), n_parts, next_b);
    FILE *info = fopen((prefix + ".sdbg_info").c_str(), "w");
    if (!info) die("cannot write %s.sdbg_info", prefix.c_str());
    fprintf(info, "k %d\nwords_per_tip_label %d\nnum_buckets %d\nnum_threads %d\n", k, wpt, 65536, n_parts);
    fprintf(info, "total_size %lld\nnum_tips %lld\nlarge_multi %lld\n", total, tips, large);
    for (const std::string &l : lines) fputs(l.c_str(), info);
    if (fclose(info) != 0) die("write error on %s.sdbg_info", prefix.c_str());
    for (int r = 0; r < n_parts; ++r) remove((prefix + ".sdbg_info.part" + std::to_string(r)).c_str());
}

void read_sdbg(const std::string &prefix, EdgeStream &s) {
    FILE *info = fopen((prefix + ".sdbg_info").c_str(), "r");
    if (!info) die("cannot open %s.sdbg_info", prefix.c_str());
    int nb = 0, nf = 0;
    long long total = 0, ntips = 0, nlarge = 0;
    if (fscanf(info, "k %d\n", &s.k) != 1 || fscanf(info, "words_per_tip_label %d\n", &s.words_per_tip) != 1 ||
        fscanf(info, "num_buckets %d\n", &nb) != 1 || fscanf(info, "num_threads %d\n", &nf) != 1 ||
        fscanf(info, "total_size %lld\n", &total) != 1 || fscanf(info, "num_tips %lld\n", &ntips) != 1 ||
        fscanf(info, "large_multi %lld\n", &nlarge) != 1 || nb != 65536)
        die("%s.sdbg_info: bad header", prefix.c_str());
    struct Rec { int tid; long long off, items, tips, large; };
    std::vector<Rec> recs(nb);
    for (int b = 0; b < nb; ++b) {
        int dummy;
        if (fscanf(info, "%d %d %lld %lld %lld %lld\n", &dummy, &recs[b].tid, &recs[b].off, &recs[b].items, &recs[b].tips, &recs[b].large) != 6)
            die("%s.sdbg_info: bad bucket line %d", prefix.c_str(), b);
    }
    fclose(info);
    std::vector<std::vector<unsigned char>> files(nf);
    for (int t = 0; t < nf; ++t) {
        std::string p = prefix + ".sdbg." + std::to_string(t);
        FILE *f = fopen(p.c_str(), "rb");
        if (!f) die("cannot open %s", p.c_str());
        fseek(f, 0, SEEK_END);
        long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        files[t].resize((size_t)sz);
        if (sz && fread(files[t].data(), 1, (size_t)sz, f) != (size_t)sz) die("read error on %s", p.c_str());
        fclose(f);
    }
    s.recs.clear(); s.large.clear(); s.tips.clear();
    long long listed = 0;
    for (int b = 0; b < nb; ++b) if (recs[b].tid >= 0) listed += recs[b].items;
    if (listed != total) die("%s: the bucket lines hold %lld records, the header says %lld", prefix.c_str(), listed, total);
    s.recs.resize((size_t)total);
    s.large.reserve((size_t)std::max(0ll, nlarge));
    s.tips.reserve((size_t)std::max(0ll, ntips) * (size_t)std::max(0, s.words_per_tip));
    uint16_t *out = s.recs.data();
    for (int b = 0; b < nb; ++b) {
        s.bucket_items[b] = recs[b].items; s.bucket_tips[b] = recs[b].tips; s.bucket_large[b] = recs[b].large;
        if (recs[b].tid < 0 || recs[b].items == 0) continue;
        if (recs[b].tid >= nf) die("%s.sdbg_info: bucket %d names file %d of %d", prefix.c_str(), b, recs[b].tid, nf);
        const unsigned char *p = files[recs[b].tid].data() + recs[b].off;
        for (long long i = 0; i < recs[b].items; ++i) {
            uint16_t it;
            memcpy(&it, p, 2); p += 2;
            *out++ = it;
            if ((it >> 8) == 255) { uint16_t m; memcpy(&m, p, 2); p += 2; s.large.push_back(m); }
            if ((it >> 5) & 1)
                for (int t = 0; t < s.words_per_tip; ++t) { uint32_t w; memcpy(&w, p, 4); p += 4; s.tips.push_back(w); }
        }
    }
}

// ----------------------------------------------------------------------------------------------------
static const double NEG_INF = -std::numeric_limits<double>::infinity();

static double prob(const std::string &tok) { return tok == "*" ? 0.0 : std::exp(-1 * std::stod(tok)); }   // hmmer3b_parser.h:111-116

static double heuristic(const ProfileHmm &hm, char pre, int state_no) {                                    // most_probable_path.h:48-118
    enum { MM, MI, MD, IM, II, DM, DD };
    const size_t M1 = (size_t)hm.M + 1;
    double h = 0;
    for (int i = state_no + 1; i <= hm.M; ++i) {
        double mt, it, dt;
        if (pre == 'm') { mt = hm.tsc[MM * M1 + i - 1]; it = hm.tsc[MI * M1 + i - 1]; dt = hm.tsc[MD * M1 + i - 1]; }
        else if (pre == 'd') { mt = hm.tsc[DM * M1 + i - 1]; it = NEG_INF; dt = hm.tsc[DD * M1 + i - 1]; }
        else { mt = hm.tsc[IM * M1 + i - 1]; it = hm.tsc[II * M1 + i - 1]; dt = NEG_INF; }
        double best_m = NEG_INF;
        for (int j = 0; j < hm.A; ++j) best_m = std::max(best_m, hm.msc[(size_t)i * hm.A + j]);
        mt += best_m - hm.max_match[i];
        dt -= hm.max_match[i];
        it = NEG_INF;                                                                                      // :100
        if (it > mt && it > dt) { h += it; pre = 'i'; --i; }
        else if (dt > mt && dt > it) { h += dt; pre = 'd'; }
        else { h += mt; pre = 'm'; }
    }
    return h;
}

bool parse_hmm(const std::string &path, ProfileHmm &hm) {
    std::ifstream f(path);
    if (!f.is_open()) return false;
    std::fill(hm.alpha, hm.alpha + 127, -1);
    std::string line, w1, w2;
    std::getline(f, line);
    bool got_hmm = false;
    while (std::getline(f, line)) {
        std::istringstream iss(line);
        w1.clear(); w2.clear();
        iss >> w1 >> w2;
        if (w1 == "NAME") hm.name = w2;
        else if (w1 == "LENG") hm.M = std::stoi(w2);
        else if (w1 == "HMM") {
            std::istringstream a(line);
            std::string tok;
            a >> tok;
            int c = 0;
            while (a >> tok) { hm.alpha[toupper(tok[0])] = c; hm.alpha[tolower(tok[0])] = c; ++c; }
            hm.A = c;
            got_hmm = true;
            break;
        }
    }
    if (!got_hmm || hm.M <= 0 || hm.A <= 0) die("%s: not a HMMER3 text model (LENG / HMM line missing)", path.c_str());
    std::getline(f, line);
    std::getline(f, line);
    {
        std::istringstream iss(line);
        std::string tag, tok;
        iss >> tag;
        if (tag != "COMPO") die("%s: COMPO line required (hmmer3b_parser.h:63-75)", path.c_str());
        for (int j = 0; j < hm.A; ++j) { iss >> tok; hm.compo.push_back(std::exp(-1 * std::stod(tok))); }
    }
    const int M = hm.M, A = hm.A;
    const size_t M1 = (size_t)M + 1;
    hm.msc.assign(M1 * A, 0.0);
    hm.tsc.assign(7 * M1, 0.0);
    hm.max_match.assign(M1, NEG_INF);
    for (int i = 0; i <= M; ++i) {
        std::string tok;
        if (i > 0) {
            std::getline(f, line);
            std::istringstream iss(line);
            iss >> tok;
            for (int j = 0; j < A; ++j) {
                iss >> tok;
                double v = std::log(prob(tok) / hm.compo[j]);
                hm.msc[(size_t)i * A + j] = v;
                if (v > hm.max_match[i]) hm.max_match[i] = v;
            }
        }
        std::getline(f, line);
        std::getline(f, line);
        std::istringstream iss(line);
        for (int t = 0; t < 7; ++t) { iss >> tok; hm.tsc[(size_t)t * M1 + i] = std::log(prob(tok)); }
    }
    hm.h.assign(3 * M1, 0.0);
    for (int i = 0; i <= M; ++i) {
        hm.h[i] = heuristic(hm, 'm', i);
        hm.h[M1 + i] = heuristic(hm, 'i', i);
        hm.h[2 * M1 + i] = heuristic(hm, 'd', i);
    }
    return true;
}

// ----------------------------------------------------------------------------------------------------
std::vector<GeneEntry> read_gene_list(const std::string &path) {
    std::vector<GeneEntry> out;
    std::ifstream f(path);
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream iss(line);
        GeneEntry g;
        iss >> g.name >> g.fwd_hmm >> g.rev_hmm;
        out.push_back(g);
    }
    return out;
}

bool read_seeds(const std::string &path, std::vector<std::string> &kmers, std::vector<int32_t> &start_state) {
    std::ifstream f(path);
    if (!f.is_open()) return false;
    std::string line, col[8];
    while (std::getline(f, line)) {
        std::istringstream iss(line);
        for (int i = 0; i < 8; ++i) { col[i].clear(); iss >> col[i]; }
        std::transform(col[3].begin(), col[3].end(), col[3].begin(), ::tolower);      // search.cpp:156
        kmers.push_back(col[3]);
        start_state.push_back(std::stoi(col[7]) - 1);                                 // search.cpp:157
    }
    return true;
}

// ----------------------------------------------------------------------------------------------------
// findstart: the reference word set (find_start, fast_kmer_filter.cpp:81-91; ProtKmerGenerator in model-only mode,
// prot_kmer_generator.h:60-135).  A window is broken by lower case (insert columns), '-' and 'X' ('-' and 'X' still occupy a
// model column); '.', '*' and letters outside the alphabet are skipped; of equal words the first one stays (insert_unique).
RefWords load_reference_words(const std::string &faa_path, int kaa) {
    std::ifstream f(faa_path);
    if (!f.is_open()) die("File %s doesn't exist", faa_path.c_str());
    static const char *kAlphabet = "ARNDCQEGHILKMFPSTWYV";              // prot_kmer.h:31-40
    int code[128];
    for (int &c : code) c = -1;
    for (int i = 0; i < 20; ++i) code[(int)kAlphabet[i]] = i;
    RefWords out;
    std::map<std::pair<uint64_t, uint64_t>, int> seen;
    std::string line, seq;
    bool have = false;
    auto flush = [&]() {
        int position = 1, run = 0;
        std::vector<int> window;
        for (char base : seq) {
            if ((base >= 'a' && base <= 'z') || base == '-' || base == 'X') {
                if (base == '-' || base == 'X') ++position;
                run = 0;
                continue;
            }
            if ((unsigned char)base < 128 && code[(int)base] >= 0) {
                window.push_back(code[(int)base]);
                ++position;
                if (++run >= kaa) {
                    uint64_t w0 = 0, w1 = 0;
                    for (int j = 0; j < kaa; ++j) {
                        uint64_t c = (uint64_t)window[window.size() - kaa + j];
                        if (j < 12) w0 = (w0 << 5) | c; else w1 = (w1 << 5) | c;
                    }
                    if (seen.emplace(std::make_pair(w0, w1), 0).second) {
                        out.words.push_back(w0); out.words.push_back(w1);
                        out.model_pos.push_back(position - kaa);
                        std::string prot;
                        for (int j = 0; j < kaa; ++j) prot.push_back((char)(kAlphabet[window[window.size() - kaa + j]] | 0x20));   // decodePacked: lower case
                        out.prot.push_back(prot);
                    }
                }
            }
        }
        seq.clear();
    };
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!line.empty() && line[0] == '>') { if (have) flush(); have = true; continue; }
        if (have) for (char c : line) if (c != ' ' && c != '\t') seq.push_back(c);
    }
    if (have) flush();
    return out;
}

}  // namespace mgta_host
