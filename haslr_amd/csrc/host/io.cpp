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
#include "host_internal.h"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>

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

inline uint8_t base_code(char c) {
    switch (c) {
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 0;   // A, a, N and everything else (Compressed_sequence.cpp:10-19 with "& 3")
    }
}

void pack_into(std::vector<uint8_t>& dst, std::vector<uint64_t>& off, const std::string& s) {
    off.push_back(dst.size());
    size_t nbytes = ((s.size() + 15) / 16) * 4;   // whole dwords: every sequence starts 4-byte aligned
    size_t base = dst.size();
    dst.resize(base + nbytes, 0);
    for (size_t i = 0; i < s.size(); i++) dst[base + (i >> 2)] |= (uint8_t)(base_code(s[i]) << ((i & 3) * 2));
}

bool parse_u32(const char* b, const char* e, uint32_t& v) {
    if (b == e) return false;
    uint64_t x = 0;
    for (const char* p = b; p < e; p++) { if (*p < '0' || *p > '9') return false; x = x * 10 + (uint64_t)(*p - '0'); if (x > 0xffffffffULL) return false; }
    v = (uint32_t)x;
    return true;
}

}  // namespace

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
    // calc_uniq_freq
    std::vector<std::pair<uint32_t, double>> cf(d.contig_len.size());
    for (size_t i = 0; i < cf.size(); i++) cf[i] = {d.contig_len[i], d.contig_km[i]};
    std::sort(cf.begin(), cf.end(), std::greater<std::pair<uint32_t, double>>());
    double f = 0;
    size_t i = 0;
    for (; i < 20 && i < cf.size(); i++) f += cf[i].second;
    d.uniq_freq = f / i;
    return true;
}

bool load_reads_file(Dataset& d, const std::string& path) {
    bool ok = read_seq_file(path, [&](const std::string&, const std::string&, const std::string& seq) {
        d.read_len.push_back((uint32_t)seq.size());
        d.total_read_bases += seq.size();
        pack_into(d.read_packed, d.read_off, seq);
    });
    if (!ok) { g_err = "[ERROR] (DNA-Seq::load_seq_compressed) could not open file: " + path; return false; }
    return true;
}

bool load_paf_file(Dataset& d, const std::string& path) {
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
        // CIGAR: first tag starting with cg:Z:, parsed like sscanf("%u%c") until it stops matching (Common.cpp:108-121)
        d.cg_off.push_back(d.cg_ops.size());
        for (size_t k = 12; k < f.size(); k++) {
            if (f[k].second - f[k].first >= 5 && memcmp(f[k].first, "cg:Z:", 5) == 0) {
                const char* p = f[k].first + 5;
                while (p < f[k].second) {
                    uint64_t len = 0;
                    const char* q = p;
                    while (q < f[k].second && *q >= '0' && *q <= '9') { len = len * 10 + (uint64_t)(*q - '0'); q++; }
                    if (q == p || q == f[k].second) break;
                    if (len >= (1u << 30)) { g_err = "[ERROR] CIGAR operation longer than 2^30 at PAF line " + std::to_string(lineno); return false; }
                    uint32_t code = *q == 'M' ? HX_CG_M : *q == 'I' ? HX_CG_I : *q == 'D' ? HX_CG_D : HX_CG_OTHER;
                    if (len) d.cg_ops.push_back(((uint32_t)len << 2) | code);
                    p = q + 1;
                }
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

Dataset* load_dataset(const char* contig_path, const char* long_path, bool long_fofn, const char* mapping_path, bool mapping_fofn) {
    std::unique_ptr<Dataset> d(new Dataset);
    if (!load_contigs(*d, contig_path)) return nullptr;
    if (!for_each_path(long_path, long_fofn, [&](const std::string& p) { return load_reads_file(*d, p); }, "Longread::load_longread_compressed_fofn")) return nullptr;
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

}  // namespace hxh

using namespace hxh;

extern "C" const char* hxh_last_error(void) { return g_err.c_str(); }

extern "C" hxh_dataset* hxh_dataset_load(const char* contig_path, const char* long_path, int long_fofn, const char* mapping_path, int mapping_fofn) {
    return reinterpret_cast<hxh_dataset*>(load_dataset(contig_path, long_path, long_fofn != 0, mapping_path, mapping_fofn != 0));
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
