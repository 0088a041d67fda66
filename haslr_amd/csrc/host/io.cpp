// io.cpp — ingest: contigs FASTA, long-read FASTA/FASTQ (optionally gz, optionally a file of file
// names) and the PAF, into the SoA arrays of include/haslr_types.h.
//
// Behaviour follows the reference loaders (paths under /root/reference/src/haslr_assemble/src/):
//   load_contig_compressed  Contig.cpp:43-117   (KC:i: / km:f: taken from the FASTA comment, :60-63)
//   calc_uniq_freq          Contig.cpp:162-174  (mean km of the <=20 longest contigs, ties by larger km)
//   load_longread_compressed Longread.cpp:109-162 (every record becomes read #ordinal; non-ACGT packs as A)
//   load_alignment          Longread.cpp:234-302 (field positions :262-289, first "cg:Z:" tag :276-283)
// The reference parses with kseq.h (name = header up to first blank, comment = rest, multi-line
// sequences, FASTA or FASTQ); this reader implements the same record grammar on zlib's gzread.
// The filters of load_alignment are NOT applied here: every PAF line becomes a raw record and the
// predicate runs in the chain kernel.
#include <chrono>
#include "host_internal.h"

#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <thread>

namespace hxh {

thread_local std::string g_err;

namespace {

class GzLines {
  public:
    explicit GzLines(const std::string& path) : f_(gzopen(path.c_str(), "r")) {
        if (f_) gzbuffer(f_, 1 << 20);
    }
    ~GzLines() { if (f_) gzclose(f_); }
    bool ok() const { return f_ != nullptr; }
    // next line without its terminator; false at EOF
    bool next(std::string& line) {
        line.clear();
        for (;;) {
            if (pos_ == len_) {
                int n = gzread(f_, buf_, sizeof(buf_));
                if (n <= 0) return !line.empty();
                pos_ = 0; len_ = (size_t)n;
            }
            char* nl = (char*)memchr(buf_ + pos_, '\n', len_ - pos_);
            if (nl) {
                line.append(buf_ + pos_, nl - (buf_ + pos_));
                pos_ = (size_t)(nl - buf_) + 1;
                if (!line.empty() && line.back() == '\r') line.pop_back();
                return true;
            }
            line.append(buf_ + pos_, len_ - pos_);
            pos_ = len_;
        }
    }
  private:
    gzFile f_;
    char buf_[1 << 16];
    size_t pos_ = 0, len_ = 0;
};

// FASTA / FASTQ records; cb(name, comment, seq)
bool read_seq_file(const std::string& path, const std::function<void(const std::string&, const std::string&, const std::string&)>& cb) {
    GzLines in(path);
    if (!in.ok()) return false;
    std::string line, name, comment, seq;
    bool have = false, in_qual = false;
    size_t qual_left = 0;
    auto flush = [&]() { if (have) cb(name, comment, seq); have = false; };
    while (in.next(line)) {
        if (in_qual) {
            qual_left = line.size() >= qual_left ? 0 : qual_left - line.size();
            if (qual_left == 0) in_qual = false;
            continue;
        }
        if (line.empty()) continue;
        char c0 = line[0];
        if (c0 == '>' || c0 == '@') {            // record start (kseq: '>' or '@' at line start ends a sequence)
            flush();
            size_t sp = line.find_first_of(" \t", 1);
            name = line.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
            comment = sp == std::string::npos ? "" : line.substr(sp + 1);
            seq.clear(); have = true;
        } else if (c0 == '+' && have) {          // FASTQ: skip as many quality characters as bases
            qual_left = seq.size();
            in_qual = qual_left > 0;
            flush();
        } else if (have) {
            for (char c : line) if (c != ' ' && c != '\t') seq.push_back(c);
        }
    }
    flush();
    return true;
}

// 2-bit code of a base: C 1, G 2, T 3, and 0 for A, N and everything else (Compressed_sequence.cpp:10-19 with "& 3"). A table, not a
// switch: the bases are random, so a compare chain mispredicts on every other character (measured: 10 ns per base against < 1 ns)
struct BaseCodes { uint8_t v[256]; constexpr BaseCodes() : v() { v['C'] = v['c'] = 1; v['G'] = v['g'] = 2; v['T'] = v['t'] = 3; } };
constexpr BaseCodes kBaseCodes;
inline uint8_t base_code(char c) { return kBaseCodes.v[(uint8_t)c]; }

template <class Arena> void pack_into(Arena& dst, std::vector<uint64_t>& off, const std::string& s) {
    off.push_back(dst.size());
    size_t nbytes = ((s.size() + 15) / 16) * 4;   // whole dwords: every sequence starts 4-byte aligned
    size_t base = dst.size();
    dst.resize(base + nbytes);
    uint8_t* out = dst.data() + base;
    uint32_t w = 0, k = 0;
    for (const char c : s) {                      // 16 bases per dword, little-endian, first base in the low bits
        w |= (uint32_t)base_code(c) << k; k += 2;
        if (k == 32) { memcpy(out, &w, 4); out += 4; w = 0; k = 0; }
    }
    if (k) memcpy(out, &w, 4);
}

// CIGAR text -> op words, parsed like sscanf("%u%c") until it stops matching (Common.cpp:108-121). `odd` is set when the op words do
// not spell the text again (letters other than M/I/D, zero lengths, leading zeros, trailing garbage): such a record keeps its text,
// because index.longread stores the cg:Z: string as it was (index_cache.cpp).
bool parse_cigar_text(const char* p, const char* end, U32Arena& ops, bool& odd, bool& too_long) {
    odd = false; too_long = false;
    while (p < end) {
        uint64_t len = 0;
        const char* q = p;
        while (q < end && *q >= '0' && *q <= '9') { len = len * 10 + (uint64_t)(*q - '0'); q++; }
        if (q == p || q == end) { odd = true; break; }
        if (len >= (1u << 30)) { too_long = true; return false; }
        const uint32_t code = *q == 'M' ? HX_CG_M : *q == 'I' ? HX_CG_I : *q == 'D' ? HX_CG_D : HX_CG_OTHER;
        if (code == HX_CG_OTHER || len == 0 || *p == '0') odd = true;
        if (len) ops.push_back(((uint32_t)len << 2) | code);
        p = q + 1;
    }
    return true;
}

bool parse_u32(const char* b, const char* e, uint32_t& v) {
    if (b == e) return false;
    uint64_t x = 0;
    for (const char* p = b; p < e; p++) { if (*p < '0' || *p > '9') return false; x = x * 10 + (uint64_t)(*p - '0'); if (x > 0xffffffffULL) return false; }
    v = (uint32_t)x;
    return true;
}

// ---- multi-threaded ingest of PLAIN (not gzip) files: the file is mapped, cut at record boundaries, parsed by `threads` workers into
// private buffers and stitched together in file order. Results are the arrays the streaming readers above produce, element for element
// (tests/test_ingest_mt.py); gzip input, FASTQ and anything unusual falls back to the streaming readers.
struct Mapped {
    const char* p = nullptr; size_t n = 0; int fd = -1;
    explicit Mapped(const std::string& path) {
        fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) return;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); fd = -1; return; }
        n = (size_t)st.st_size;
        if (n == 0) return;
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); fd = -1; n = 0; return; }
        p = (const char*)m;
    }
    ~Mapped() { if (p) munmap((void*)p, n); if (fd >= 0) close(fd); }
    bool ok() const { return fd >= 0; }
    bool gz() const { return n >= 2 && (unsigned char)p[0] == 0x1f && (unsigned char)p[1] == 0x8b; }
};

struct IoLap {   // HASLR_IO_DEBUG=1: phase times of the parallel loaders on stderr
    const char* who; bool on; std::chrono::steady_clock::time_point t0;
    explicit IoLap(const char* w) : who(w), on(getenv("HASLR_IO_DEBUG") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void lap(const char* what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[io] %s: %s %.3f s\n", who, what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};

// start of the line that contains or follows position `pos` ... i.e. first line start >= pos
inline size_t next_line_start(const char* p, size_t n, size_t pos) {
    if (pos == 0) return 0;
    const char* nl = (const char*)memchr(p + pos - 1, '\n', n - (pos - 1));
    return nl ? (size_t)(nl - p) + 1 : n;
}

struct PafPart {
    std::vector<uint32_t> q_id, q_start, q_end, t_id, t_len, t_start, t_end, n_match, n_block;
    U32Arena cg_ops;
    std::vector<uint8_t> is_rev, mapq;
    std::vector<uint64_t> cg_off;       // local offsets
    uint64_t lines = 0;                  // lines in this part (for global line numbers)
    uint64_t first_rec_line = 0;         // local line number of the first record (1-based), 0 = no record
    uint64_t err_line = 0; std::string err;   // first error of this part (local line number)
    std::vector<std::pair<uint64_t, std::string>> odd;   // (local record, cg text) of records whose op words do not spell their text
};

// one PAF line -> part; returns false with part.err set
bool parse_paf_line(const char* b, const char* end, PafPart& d, size_t n_reads, size_t n_contigs, std::vector<std::pair<const char*, const char*>>& f, const std::string& path) {
    f.clear();
    for (const char* p = b;; p++) {
        if (p == end || *p == '\t') { f.push_back({b, p}); b = p + 1; if (p == end) break; }
    }
    if (f.size() < 12) { d.err = "has fewer than 12 columns"; return false; }
    uint32_t v[12] = {0};
    static const int numeric[] = {0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11};
    for (int k : numeric)
        if (!parse_u32(f[k].first, f[k].second, v[k])) {
            d.err = ": column " + std::to_string(k + 1) + " is not an unsigned integer (read and contig names must be the ordinals haslr.py assigns)";
            return false;
        }
    if (v[0] >= n_reads) { d.err = "Q" + std::to_string(v[0]); return false; }
    if (v[5] >= n_contigs) { d.err = "T" + std::to_string(v[5]); return false; }
    if (!d.q_id.empty() && v[0] < d.q_id.back()) { d.err = "ORDER"; return false; }
    d.q_id.push_back(v[0]); d.q_start.push_back(v[2]); d.q_end.push_back(v[3]);
    d.is_rev.push_back(*f[4].first == '-' ? 1 : 0);
    d.t_id.push_back(v[5]); d.t_len.push_back(v[6]); d.t_start.push_back(v[7]); d.t_end.push_back(v[8]);
    d.n_match.push_back(v[9]); d.n_block.push_back(v[10]); d.mapq.push_back((uint8_t)v[11]);
    d.cg_off.push_back(d.cg_ops.size());
    for (size_t k = 12; k < f.size(); k++) {
        if (f[k].second - f[k].first >= 5 && memcmp(f[k].first, "cg:Z:", 5) == 0) {
            bool odd, too_long;
            if (!parse_cigar_text(f[k].first + 5, f[k].second, d.cg_ops, odd, too_long)) { d.err = "CIGAR"; return false; }
            if (odd) d.odd.push_back({d.q_id.size() - 1, std::string(f[k].first + 5, f[k].second)});
            break;
        }
    }
    return true;
}

std::string paf_error_text(const std::string& code, uint64_t lineno, const std::string& path) {
    if (code == "ORDER") return "[ERROR] PAF is not grouped by ascending query id at line " + std::to_string(lineno) + " (the reference silently mis-assigns alignments in that case, Longread.cpp:57-84)";
    if (code == "CIGAR") return "[ERROR] CIGAR operation longer than 2^30 at PAF line " + std::to_string(lineno);
    if (code[0] == 'Q') return "[ERROR] PAF query " + code.substr(1) + " is not a loaded long read";
    if (code[0] == 'T') return "[ERROR] PAF target " + code.substr(1) + " is not a loaded contig";
    if (code[0] == ':') return "[ERROR] PAF line " + std::to_string(lineno) + " of " + path + code;
    return "[ERROR] PAF line " + std::to_string(lineno) + " of " + path + " " + code;
}

// returns 1 done, 0 error (g_err set), -1 not applicable (caller streams)
int load_paf_parallel(Dataset& d, const std::string& path, unsigned threads) {
    Mapped m(path);
    if (!m.ok()) return -1;
    if (m.gz() || m.n < (1u << 16) || threads < 2) return -1;
    std::vector<size_t> cut(threads + 1);
    for (unsigned t = 0; t <= threads; t++) cut[t] = t == threads ? m.n : next_line_start(m.p, m.n, m.n / threads * t);
    std::vector<PafPart> parts(threads);
    const size_t n_reads = d.read_len.size(), n_contigs = d.contig_len.size();
    IoLap lap("paf");
    run_parallel(threads, [&](unsigned t) {
        PafPart& P = parts[t];
        std::vector<std::pair<const char*, const char*>> f;
        const char* p = m.p + cut[t];
        const char* pe = m.p + cut[t + 1];
        while (p < pe) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(pe - p));
            const char* le = nl ? nl : pe;
            const char* e2 = le;
            if (e2 > p && e2[-1] == '\r') e2--;
            P.lines++;
            if (e2 > p) {
                if (!parse_paf_line(p, e2, P, n_reads, n_contigs, f, path)) { P.err_line = P.lines; return; }
                if (!P.first_rec_line) P.first_rec_line = P.lines;
            }
            p = nl ? nl + 1 : pe;
        }
    });
    lap.lap("parse");
    // errors in file order, including the ordering rule across part boundaries
    uint64_t line0 = 0;
    bool have_last = !d.q_id.empty();
    uint32_t last_q = have_last ? d.q_id.back() : 0;
    for (unsigned t = 0; t < threads; t++) {
        const PafPart& P = parts[t];
        const bool boundary_bad = have_last && !P.q_id.empty() && P.q_id.front() < last_q;
        if (boundary_bad && (!P.err_line || P.first_rec_line <= P.err_line)) { g_err = paf_error_text("ORDER", line0 + P.first_rec_line, path); return 0; }
        if (P.err_line) { g_err = paf_error_text(P.err, line0 + P.err_line, path); return 0; }
        if (!P.q_id.empty()) { have_last = true; last_q = P.q_id.back(); }
        line0 += P.lines;
    }
    lap.lap("checks");
    // stitch
    std::vector<size_t> rec0(threads + 1, d.q_id.size()), op0(threads + 1, d.cg_ops.size());
    for (unsigned t = 0; t < threads; t++) { rec0[t + 1] = rec0[t] + parts[t].q_id.size(); op0[t + 1] = op0[t] + parts[t].cg_ops.size(); }
    const size_t nr = rec0[threads], no = op0[threads];
    d.q_id.resize(nr); d.q_start.resize(nr); d.q_end.resize(nr); d.t_id.resize(nr); d.t_len.resize(nr); d.t_start.resize(nr); d.t_end.resize(nr);
    d.n_match.resize(nr); d.n_block.resize(nr); d.is_rev.resize(nr); d.mapq.resize(nr); d.cg_off.resize(nr); d.cg_ops.resize(no);
    if (lap.on) fprintf(stderr, "[io] paf: %zu records, %zu CIGAR ops\n", nr, no);
    lap.lap("resize");
    run_parallel(threads, [&](unsigned t) {
        const PafPart& P = parts[t];
        const size_t r = rec0[t], k = P.q_id.size();
        auto cp32 = [&](std::vector<uint32_t>& dst, const std::vector<uint32_t>& src) { if (k) memcpy(dst.data() + r, src.data(), k * 4); };
        cp32(d.q_id, P.q_id); cp32(d.q_start, P.q_start); cp32(d.q_end, P.q_end); cp32(d.t_id, P.t_id); cp32(d.t_len, P.t_len); cp32(d.t_start, P.t_start);
        cp32(d.t_end, P.t_end); cp32(d.n_match, P.n_match); cp32(d.n_block, P.n_block);
        if (k) { memcpy(d.is_rev.data() + r, P.is_rev.data(), k); memcpy(d.mapq.data() + r, P.mapq.data(), k); }
        for (size_t i = 0; i < k; i++) d.cg_off[r + i] = P.cg_off[i] + op0[t];
        if (!P.cg_ops.empty()) memcpy(d.cg_ops.data() + op0[t], P.cg_ops.data(), P.cg_ops.size() * 4);
    });
    for (unsigned t = 0; t < threads; t++) for (auto& o : parts[t].odd) d.cg_text_odd[rec0[t] + o.first] = o.second;
    lap.lap("stitch");
    return 1;
}

// plain FASTA (records start with '>' or '@' at a line start, no '+' lines): lengths first, then every record is packed in place
int load_reads_parallel(Dataset& d, const std::string& path, unsigned threads) {
    Mapped m(path);
    if (!m.ok()) return -1;
    if (m.gz() || m.n < (1u << 16) || threads < 2 || m.p[0] != '>') return -1;
    std::vector<size_t> cut(threads + 1);
    for (unsigned t = 0; t <= threads; t++) cut[t] = t == threads ? m.n : next_line_start(m.p, m.n, m.n / threads * t);
    std::vector<std::vector<size_t>> starts(threads);
    std::vector<uint8_t> odd(threads, 0);
    IoLap lap("reads");
    run_parallel(threads, [&](unsigned t) {   // record starts and a scan for anything FASTQ-like
        const char* p = m.p + cut[t];
        const char* pe = m.p + cut[t + 1];
        while (p < pe) {
            if (*p == '>' || *p == '@') starts[t].push_back((size_t)(p - m.p));
            else if (*p == '+') odd[t] = 1;
            const char* nl = (const char*)memchr(p, '\n', (size_t)(pe - p));
            p = nl ? nl + 1 : pe;
        }
    });
    lap.lap("record starts");
    for (unsigned t = 0; t < threads; t++) if (odd[t]) return -1;
    std::vector<size_t> st;
    for (unsigned t = 0; t < threads; t++) st.insert(st.end(), starts[t].begin(), starts[t].end());
    const size_t nrec = st.size();
    st.push_back(m.n);
    const size_t r0 = d.read_len.size();
    d.read_len.resize(r0 + nrec);
    // bases of a record: every character of its sequence lines except blanks, tabs and the line terminator (\n, or \r\n)
    auto for_seq = [&](size_t r, auto&& fn) {
        const char* p = m.p + st[r];
        const char* pe = m.p + st[r + 1];
        const char* nl = (const char*)memchr(p, '\n', (size_t)(pe - p));   // header line
        p = nl ? nl + 1 : pe;
        while (p < pe) {
            nl = (const char*)memchr(p, '\n', (size_t)(pe - p));
            const char* le = nl ? nl : pe;
            const char* e2 = le;
            if (e2 > p && e2[-1] == '\r') e2--;
            for (const char* q = p; q < e2; q++) if (*q != ' ' && *q != '\t') fn(*q);
            p = nl ? nl + 1 : pe;
        }
    };
    run_parallel(threads, [&](unsigned t) {
        for (size_t r = nrec * t / threads; r < nrec * (t + 1) / threads; r++) { size_t n = 0; for_seq(r, [&](char) { n++; }); d.read_len[r0 + r] = (uint32_t)n; }
    });
    lap.lap("lengths");
    size_t base = d.read_packed.size();
    d.read_off.resize(r0 + nrec);
    for (size_t r = 0; r < nrec; r++) { d.read_off[r0 + r] = base; base += (((size_t)d.read_len[r0 + r] + 15) / 16) * 4; d.total_read_bases += d.read_len[r0 + r]; }
    d.read_packed.resize(base);   // (not initialised: every dword is written below, by the thread that packs its record)
    lap.lap("offsets");
    run_parallel(threads, [&](unsigned t) {
        for (size_t r = nrec * t / threads; r < nrec * (t + 1) / threads; r++) {
            uint32_t* dst = reinterpret_cast<uint32_t*>(d.read_packed.data() + d.read_off[r0 + r]);   // records start 4-byte aligned and own whole dwords
            uint32_t w = 0, k = 0;
            for_seq(r, [&](char c) { w |= (uint32_t)base_code(c) << k; k += 2; if (k == 32) { *dst++ = w; w = 0; k = 0; } });
            if (k) *dst = w;
        }
    });
    lap.lap("pack");
    return 1;
}

}  // namespace

unsigned g_io_threads = 1;   // set by load_dataset

bool load_contigs(Dataset& d, const std::string& path) {
    bool bad = false;
    bool ok = read_seq_file(path, [&](const std::string&, const std::string& comment, const std::string& seq) {
        const char* p1 = strstr(comment.c_str(), "KC:i:");
        const char* p2 = strstr(comment.c_str(), "km:f:");
        if (!p1 || !p2) { bad = true; return; }
        d.contig_kc.push_back((uint32_t)strtoul(p1 + 5, nullptr, 10));
        d.contig_km.push_back(strtod(p2 + 5, nullptr));
        d.contig_len.push_back((uint32_t)seq.size());
        pack_into(d.contig_packed, d.contig_off, seq);
    });
    if (!ok) { g_err = "[ERROR] (Contig::load_contig_compressed) could not open file: " + path; return false; }
    if (bad) { g_err = "[ERROR] contig header without KC:i:/km:f: comment in " + path; return false; }
    d.contig_off.push_back(d.contig_packed.size());
    finish_contigs(d);
    return true;
}

// calc_uniq_freq (Contig.cpp:162-174)
void finish_contigs(Dataset& d) {
    std::vector<std::pair<uint32_t, double>> cf(d.contig_len.size());
    for (size_t i = 0; i < cf.size(); i++) cf[i] = {d.contig_len[i], d.contig_km[i]};
    std::sort(cf.begin(), cf.end(), std::greater<std::pair<uint32_t, double>>());
    double f = 0;
    size_t i = 0;
    for (; i < 20 && i < cf.size(); i++) f += cf[i].second;
    d.uniq_freq = f / i;
}

// one alignment's CIGAR from text (index.longread): offsets, op words, and the text itself when the words do not spell it
bool parse_cigar_ops(const char* b, const char* e, U32Arena& ops, bool& odd, bool& too_long) { return parse_cigar_text(b, e, ops, odd, too_long); }

bool append_cigar(Dataset& d, const char* b, const char* e) {
    d.cg_off.push_back(d.cg_ops.size());
    bool odd, too_long;
    if (!parse_cigar_text(b, e, d.cg_ops, odd, too_long)) { g_err = "[ERROR] CIGAR operation longer than 2^30 in index.longread"; return false; }
    if (odd) d.cg_text_odd[d.q_id.size() - 1] = std::string(b, e);
    return true;
}

bool load_reads_file(Dataset& d, const std::string& path) {
    {
        const int r = load_reads_parallel(d, path, g_io_threads);
        if (r >= 0) return r == 1;
    }
    bool ok = read_seq_file(path, [&](const std::string&, const std::string&, const std::string& seq) {
        d.read_len.push_back((uint32_t)seq.size());
        d.total_read_bases += seq.size();
        pack_into(d.read_packed, d.read_off, seq);
    });
    if (!ok) { g_err = "[ERROR] (DNA-Seq::load_seq_compressed) could not open file: " + path; return false; }
    return true;
}

bool load_paf_file(Dataset& d, const std::string& path) {
    {
        const int r = load_paf_parallel(d, path, g_io_threads);
        if (r >= 0) return r == 1;
    }
    GzLines in(path);
    if (!in.ok()) { g_err = "[ERROR] (Longread::load_alignment) could not open file: " + path; return false; }
    std::string line;
    std::vector<std::pair<const char*, const char*>> f;
    uint64_t lineno = 0;
    while (in.next(line)) {
        lineno++;
        if (line.empty()) continue;
        f.clear();
        const char* b = line.data();
        const char* end = b + line.size();
        for (const char* p = b;; p++) {
            if (p == end || *p == '\t') { f.push_back({b, p}); b = p + 1; if (p == end) break; }
        }
        if (f.size() < 12) { g_err = "[ERROR] PAF line " + std::to_string(lineno) + " of " + path + " has fewer than 12 columns"; return false; }
        uint32_t v[12] = {0};
        static const int numeric[] = {0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11};
        for (int k : numeric)
            if (!parse_u32(f[k].first, f[k].second, v[k])) {
                g_err = "[ERROR] PAF line " + std::to_string(lineno) + " of " + path + ": column " + std::to_string(k + 1) +
                        " is not an unsigned integer (read and contig names must be the ordinals haslr.py assigns)";
                return false;
            }
        uint32_t qid = v[0];
        if (qid >= d.read_len.size()) { g_err = "[ERROR] PAF query " + std::to_string(qid) + " is not a loaded long read"; return false; }
        if (v[5] >= d.contig_len.size()) { g_err = "[ERROR] PAF target " + std::to_string(v[5]) + " is not a loaded contig"; return false; }
        if (!d.q_id.empty() && qid < d.q_id.back()) {
            g_err = "[ERROR] PAF is not grouped by ascending query id at line " + std::to_string(lineno) +
                    " (the reference silently mis-assigns alignments in that case, Longread.cpp:57-84)";
            return false;
        }
        d.q_id.push_back(qid); d.q_start.push_back(v[2]); d.q_end.push_back(v[3]);
        d.is_rev.push_back(*f[4].first == '-' ? 1 : 0);
        d.t_id.push_back(v[5]); d.t_len.push_back(v[6]); d.t_start.push_back(v[7]); d.t_end.push_back(v[8]);
        d.n_match.push_back(v[9]); d.n_block.push_back(v[10]); d.mapq.push_back((uint8_t)v[11]);
        // CIGAR: first tag starting with cg:Z:
        d.cg_off.push_back(d.cg_ops.size());
        for (size_t k = 12; k < f.size(); k++) {
            if (f[k].second - f[k].first >= 5 && memcmp(f[k].first, "cg:Z:", 5) == 0) {
                bool odd, too_long;
                if (!parse_cigar_text(f[k].first + 5, f[k].second, d.cg_ops, odd, too_long)) { g_err = "[ERROR] CIGAR operation longer than 2^30 at PAF line " + std::to_string(lineno); return false; }
                if (odd) d.cg_text_odd[d.q_id.size() - 1] = std::string(f[k].first + 5, f[k].second);
                break;
            }
        }
    }
    return true;
}

static bool for_each_path(const std::string& path, bool fofn, const std::function<bool(const std::string&)>& fn, const char* who) {
    if (!fofn) return fn(path);
    std::ifstream fin(path);
    if (!fin.is_open()) { g_err = std::string("[ERROR] (") + who + ") could not open file: " + path; return false; }
    std::string line;
    while (std::getline(fin, line)) if (!fn(line)) return false;
    return true;
}

static bool file_exists(const std::string& p) { FILE* f = fopen(p.c_str(), "rb"); if (f) fclose(f); return f != nullptr; }

Dataset* load_dataset_cached(const char* index_dir, const char* contig_path, const char* long_path, bool long_fofn, const char* mapping_path, bool mapping_fofn,
                             unsigned threads, int* used_contig_index, int* used_longread_index) {
    std::unique_ptr<Dataset> d(new Dataset);
    if (threads == 0) {   // automatic: HASLR_IO_THREADS, else up to 16 hardware threads
        const char* e = getenv("HASLR_IO_THREADS");
        threads = e ? (unsigned)atoi(e) : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    }
    g_io_threads = std::max(1u, std::min(threads, 256u));
    const std::string ci = index_dir ? std::string(index_dir) + "/index.contig" : std::string();
    const std::string li = index_dir ? std::string(index_dir) + "/index.longread" : std::string();
    const bool use_ci = index_dir && file_exists(ci), use_li = index_dir && file_exists(li);
    if (used_contig_index) *used_contig_index = use_ci;
    if (used_longread_index) *used_longread_index = use_li;
    // the contigs (one streaming reader) and the long reads (the parallel loader) touch different arrays of the data set: with several threads the contigs
    // are read on a thread of their own beside the reads (0.3 s of a 1.7 s load at 140 Mb); an error of the contig file is reported first, as before
    std::string contig_error;
    bool contigs_ok = true;
    auto contigs = [&]() {
        contigs_ok = use_ci ? read_contig_index(*d, ci) : load_contigs(*d, contig_path);
        if (!contigs_ok) contig_error = g_err;
    };
    std::thread contig_thread;
    if (g_io_threads > 1 && !use_li) contig_thread = std::thread(contigs); else contigs();
    struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join_contigs{contig_thread};
    if (!contig_thread.joinable() && !contigs_ok) return nullptr;
    if (use_li) { if (!read_longread_index(*d, li)) return nullptr; return d.release(); }
    const bool reads_ok = for_each_path(long_path, long_fofn, [&](const std::string& p) { return load_reads_file(*d, p); }, "Longread::load_longread_compressed_fofn");
    if (contig_thread.joinable()) contig_thread.join();
    if (!contigs_ok) { g_err = contig_error; return nullptr; }
    if (!reads_ok) return nullptr;
    d->read_off.push_back(d->read_packed.size());
    if (!for_each_path(mapping_path, mapping_fofn, [&](const std::string& p) { return load_paf_file(*d, p); }, "Longread::load_alignment_fofn")) return nullptr;
    d->cg_off.push_back(d->cg_ops.size());
    if (d->q_id.size() >= 0xffffffffULL) { g_err = "[ERROR] more than 2^32-1 PAF records"; return nullptr; }
    // per-read ranges of raw records
    d->read_hit_off.assign(d->read_len.size() + 1, 0);
    for (uint32_t q : d->q_id) d->read_hit_off[q + 1]++;
    for (size_t r = 0; r < d->read_len.size(); r++) d->read_hit_off[r + 1] += d->read_hit_off[r];
    // keep vectors non-empty so that views never carry NULL
    if (d->cg_ops.empty()) d->cg_ops.push_back(0);
    return d.release();
}

Dataset* load_dataset(const char* contig_path, const char* long_path, bool long_fofn, const char* mapping_path, bool mapping_fofn, unsigned threads) {
    return load_dataset_cached(nullptr, contig_path, long_path, long_fofn, mapping_path, mapping_fofn, threads, nullptr, nullptr);
}

}  // namespace hxh

using namespace hxh;

extern "C" const char* hxh_last_error(void) { return g_err.c_str(); }

extern "C" hxh_dataset* hxh_dataset_load_mt(const char* contig_path, const char* long_path, int long_fofn, const char* mapping_path, int mapping_fofn, unsigned threads) {
    return reinterpret_cast<hxh_dataset*>(load_dataset(contig_path, long_path, long_fofn != 0, mapping_path, mapping_fofn != 0, threads));
}
extern "C" hxh_dataset* hxh_dataset_load_cached(const char* index_dir, const char* contig_path, const char* long_path, int long_fofn, const char* mapping_path,
                                                int mapping_fofn, unsigned threads, int* used_contig_index, int* used_longread_index) {
    try {   // (no exception may cross the C boundary: a corrupt cache file or an exhausted host becomes an [ERROR] like any other)
        return reinterpret_cast<hxh_dataset*>(load_dataset_cached(index_dir, contig_path, long_path, long_fofn != 0, mapping_path, mapping_fofn != 0, threads,
                                                                  used_contig_index, used_longread_index));
    } catch (const std::exception& e) { g_err = std::string("[ERROR] loading the inputs: ") + e.what(); return nullptr; }
}
extern "C" int hxh_dataset_write_contig_index(const hxh_dataset* p, const char* path) {
    try { return write_contig_index(*reinterpret_cast<const Dataset*>(p), path) ? 0 : -1; }
    catch (const std::exception& e) { g_err = std::string("[ERROR] writing index.contig: ") + e.what(); return -1; }
}
extern "C" hxh_dataset* hxh_dataset_load(const char* contig_path, const char* long_path, int long_fofn, const char* mapping_path, int mapping_fofn) {
    try { return reinterpret_cast<hxh_dataset*>(load_dataset(contig_path, long_path, long_fofn != 0, mapping_path, mapping_fofn != 0, 0)); }   // automatic thread count
    catch (const std::exception& e) { g_err = std::string("[ERROR] loading the inputs: ") + e.what(); return nullptr; }
}
extern "C" void hxh_dataset_free(hxh_dataset* p) { delete reinterpret_cast<Dataset*>(p); }
extern "C" double hxh_dataset_uniq_freq(const hxh_dataset* p) { return reinterpret_cast<const Dataset*>(p)->uniq_freq; }
extern "C" uint64_t hxh_dataset_total_read_bases(const hxh_dataset* p) { return reinterpret_cast<const Dataset*>(p)->total_read_bases; }

extern "C" void hxh_dataset_views(const hxh_dataset* p, hx_contigs* c, hx_reads* r, hx_hits* h, const uint64_t** rho) {
    const Dataset& d = *reinterpret_cast<const Dataset*>(p);
    if (c) { c->n = (uint32_t)d.contig_len.size(); c->mean_kmer = d.contig_km.data(); c->len = d.contig_len.data(); }
    if (r) { r->n = (uint32_t)d.read_len.size(); r->len = d.read_len.data(); r->off = d.read_off.data(); r->packed = d.read_packed.data(); }
    if (h) {
        h->n = d.q_id.size();
        h->q_id = d.q_id.data(); h->q_start = d.q_start.data(); h->q_end = d.q_end.data(); h->t_id = d.t_id.data(); h->t_len = d.t_len.data();
        h->t_start = d.t_start.data(); h->t_end = d.t_end.data(); h->n_match = d.n_match.data(); h->n_block = d.n_block.data();
        h->is_rev = d.is_rev.data(); h->mapq = d.mapq.data(); h->cg_off = d.cg_off.data(); h->cg_ops = d.cg_ops.data();
    }
    if (rho) *rho = d.read_hit_off.data();
}

extern "C" void hxh_dataset_contig_seq(const hxh_dataset* p, uint32_t id, char* dst) {
    const Dataset& d = *reinterpret_cast<const Dataset*>(p);
    std::string s = d.contig_seq(id);
    memcpy(dst, s.data(), s.size());
}
