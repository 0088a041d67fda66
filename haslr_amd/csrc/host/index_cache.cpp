// index_cache.cpp — the reference's `index.contig` / `index.longread` cache files.
//
// haslr_assemble writes both files into the output directory on its first run and reloads them instead of parsing FASTA + PAF when
// they exist (main.cpp:39-52, :65-103). They are raw dumps (Contig.cpp:119-132, Longread.cpp:322-339):
//
//   index.contig    u64 n | n x Contig_t | u64 block_size | block
//       Contig_t (32 B, Contig.hpp:14-21): len u32 @0, comp_len u32 @4, kmer_count u32 @8, mean_kmer f64 @16, comp_seq pointer @24
//   index.longread  u64 n | n x Longread_t | u64 seqs_size | seqs | u64 n_aln | n_aln x Align_Seq_t | u64 cigars_size | cigars
//       Longread_t (32 B, Longread.hpp:50-57): len u32 @0, comp_len u32 @4, two pointers @8 @16, contig_aln_num u32 @24
//       Align_Seq_t (48 B, Longread.hpp:33-48): q_id q_start q_end t_id t_start t_end n_match n_block (8 x u32), is_rev u8 @32,
//       mapq u8 @33, flag u8 @34, cigar_len u32 @36, pointer @40; cigars = the cg:Z: strings, each followed by a NUL
//   The pointers are stale process addresses which the reference recomputes on load (update_contigs / update_longreads, :57-84);
//   written as zero here. Alignments are the set AFTER filters 1-5, the per-read sort and the palindrome rule
//   (Longread.cpp:182-232,262-272), in read order: exactly the `hit` list of hx_chain_out with the raw record's fields.
//   Sequences use the reference's 2-bit codec (Compressed_sequence.cpp:52-70): packed from the END of the sequence, 4 bases per byte,
//   the base with the smaller position in the lower bits; the last byte holds the first len%4 bases.
//
// Parity is pinned both ways by the compiled reference (oracle/_ref/ref_front): it loads the files written here and reproduces its own
// text-parse results, and the files it writes load here into the arrays a text parse gives (tests/test_index_cache.py).
#include "host_internal.h"

#include <cstdio>
#include <cstring>

namespace hxh {

namespace {

inline uint32_t comp_len_of(uint32_t len) { return len / 4 + (len % 4 ? 1 : 0); }

// own layout (base i at bits 2*(i%4) of byte i/4) -> reference layout
void to_ref_codec(const uint8_t* mine, uint32_t len, uint8_t* dst) {
    // The reference's last byte holds the first len % 4 bases, the bytes before it (going backwards) the following groups of four, each
    // with its first base in the low bits. Our arena is the plain little-endian 2-bit stream, so a group of four starting at base p is
    // the 8 bits at bit 2p of the stream: one (unaligned) byte per output byte.
    const uint32_t cl = comp_len_of(len), r = len % 4, sh = r * 2;
    if (r) dst[cl - 1] = (uint8_t)(mine[0] & ((1u << sh) - 1));
    uint8_t* out = dst + cl - 1 - (r ? 1 : 0);
    const uint32_t groups = len / 4;
    if (!sh) for (uint32_t g = 0; g < groups; g++) *out-- = mine[g];
    else for (uint32_t g = 0; g < groups; g++) *out-- = (uint8_t)((mine[g] | ((uint32_t)mine[g + 1] << 8)) >> sh);
}
void from_ref_codec(const uint8_t* src, uint32_t len, uint8_t* mine /* zeroed, ((len+15)/16)*4 bytes */) {
    const uint32_t cl = comp_len_of(len), r = len % 4;
    for (uint32_t p = 0; p < len; p++) {
        uint32_t byte, sh;
        if (p < r) { byte = cl - 1; sh = 2 * p; }
        else { const uint32_t q = (p - r) / 4, j = (p - r) % 4; byte = cl - 1 - (r ? 1 : 0) - q; sh = 2 * j; }
        const uint8_t c = (src[byte] >> sh) & 3;
        mine[p >> 2] |= (uint8_t)(c << ((p & 3) * 2));
    }
}

struct File {
    FILE* fp;
    File(const std::string& path, const char* mode) : fp(fopen(path.c_str(), mode)) {}
    ~File() { if (fp) fclose(fp); }
    bool w(const void* p, size_t n) { return n == 0 || fwrite(p, 1, n, fp) == n; }
    bool r(void* p, size_t n) { return n == 0 || fread(p, 1, n, fp) == n; }
};

void put32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }
uint32_t get32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

}  // namespace

void append_cigar_text(const Dataset& d, uint64_t rec, std::string& s) {
    if (!d.cg_text_odd.empty()) {
        auto it = d.cg_text_odd.find(rec);
        if (it != d.cg_text_odd.end()) { s += it->second; return; }
    }
    char buf[12];
    for (uint64_t k = d.cg_off[rec]; k < d.cg_off[rec + 1]; k++) {
        const uint32_t w = d.cg_ops[k];
        uint32_t v = HX_CG_LEN(w);
        int n = 0;
        do { buf[n++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (n) s.push_back(buf[--n]);
        s.push_back("MID?"[w & 3u]);   // '?' never reaches a file: records with other letters keep their text in cg_text_odd
    }
}

std::string cigar_text(const Dataset& d, uint64_t rec) {
    std::string s;
    append_cigar_text(d, rec, s);
    return s;
}

bool write_contig_index(const Dataset& d, const std::string& path) {
    File f(path, "wb");
    if (!f.fp) { g_err = "[ERROR] (Contig::write_contig_index) could not open file: " + path; return false; }
    const uint64_t n = d.contig_len.size();
    std::vector<uint8_t> recs(n * 32, 0);
    uint64_t block = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint8_t* p = recs.data() + i * 32;
        put32(p, d.contig_len[i]); put32(p + 4, comp_len_of(d.contig_len[i])); put32(p + 8, d.contig_kc[i]);
        memcpy(p + 16, &d.contig_km[i], 8);
        block += comp_len_of(d.contig_len[i]);
    }
    std::vector<uint8_t> blk(block);
    uint64_t off = 0;
    for (uint64_t i = 0; i < n; i++) { to_ref_codec(d.contig_packed.data() + d.contig_off[i], d.contig_len[i], blk.data() + off); off += comp_len_of(d.contig_len[i]); }
    if (!f.w(&n, 8) || !f.w(recs.data(), recs.size()) || !f.w(&block, 8) || !f.w(blk.data(), blk.size())) { g_err = "[ERROR] could not write " + path; return false; }
    return true;
}

bool write_longread_index(const Dataset& d, const hx_chain_out& ch, const std::string& path) {
    File f(path, "wb");
    if (!f.fp) { g_err = "[ERROR] (Longread::write_longread_index) could not open file: " + path; return false; }
    const uint64_t n = d.read_len.size();
    if (ch.n_reads != n) { g_err = "[ERROR] index.longread: chain output does not match the data set"; return false; }
    std::vector<uint8_t> recs(n * 32, 0);
    uint64_t seqs = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint8_t* p = recs.data() + i * 32;
        put32(p, d.read_len[i]); put32(p + 4, comp_len_of(d.read_len[i]));
        put32(p + 24, (uint32_t)(ch.read_off[i + 1] - ch.read_off[i]));
        seqs += comp_len_of(d.read_len[i]);
    }
    // both big blocks are independent per record: the reference codec of every read and the cg:Z: text of every kept alignment are made
    // by the ingest threads (reads / alignments dealt in contiguous ranges), the texts are joined in alignment order afterwards
    const unsigned T = std::max(1u, g_io_threads);
    U8Arena sq;
    sq.resize(seqs);
    std::vector<uint64_t> sq_off(n + 1, 0);
    for (uint64_t i = 0; i < n; i++) sq_off[i + 1] = sq_off[i] + comp_len_of(d.read_len[i]);
    run_parallel(T, [&](unsigned t) {
        for (uint64_t i = n * t / T; i < n * (t + 1) / T; i++) to_ref_codec(d.read_packed.data() + d.read_off[i], d.read_len[i], sq.data() + sq_off[i]);
    });
    const uint64_t na = ch.n_aln;
    std::vector<uint8_t> al(na * 48, 0);
    std::vector<std::string> part(T);
    run_parallel(T, [&](unsigned t) {
        std::string& out = part[t];
        for (uint64_t a = na * t / T; a < na * (t + 1) / T; a++) {
            const uint32_t h = ch.hit[a];
            uint8_t* p = al.data() + a * 48;
            put32(p, d.q_id[h]); put32(p + 4, d.q_start[h]); put32(p + 8, d.q_end[h]); put32(p + 12, d.t_id[h]);
            put32(p + 16, d.t_start[h]); put32(p + 20, d.t_end[h]); put32(p + 24, d.n_match[h]); put32(p + 28, d.n_block[h]);
            p[32] = d.is_rev[h]; p[33] = d.mapq[h]; p[34] = 0;
            const size_t before = out.size();
            append_cigar_text(d, h, out);
            put32(p + 36, (uint32_t)(out.size() - before));
            out.push_back('\0');
        }
    });
    uint64_t cs = 0;
    for (const std::string& x : part) cs += x.size();
    bool ok = f.w(&n, 8) && f.w(recs.data(), recs.size()) && f.w(&seqs, 8) && f.w(sq.data(), sq.size()) && f.w(&na, 8) && f.w(al.data(), al.size()) && f.w(&cs, 8);
    for (const std::string& x : part) ok = ok && (x.empty() || f.w(x.data(), x.size()));
    if (!ok) { g_err = "[ERROR] could not write " + path; return false; }
    return true;
}

bool read_contig_index(Dataset& d, const std::string& path) {
    File f(path, "rb");
    if (!f.fp) { g_err = "[ERROR] (Contig::read_contig_index) could not open file: " + path; return false; }
    uint64_t n = 0, block = 0;
    if (!f.r(&n, 8)) { g_err = "[ERROR] truncated " + path; return false; }
    std::vector<uint8_t> recs(n * 32);
    if (!f.r(recs.data(), recs.size()) || !f.r(&block, 8)) { g_err = "[ERROR] truncated " + path; return false; }
    std::vector<uint8_t> blk(block);
    if (!f.r(blk.data(), block)) { g_err = "[ERROR] truncated " + path; return false; }
    uint64_t off = 0;
    for (uint64_t i = 0; i < n; i++) {
        const uint8_t* p = recs.data() + i * 32;
        const uint32_t len = get32(p), cl = get32(p + 4);
        if (cl != comp_len_of(len) || off + cl > block) { g_err = "[ERROR] inconsistent record in " + path; return false; }
        double km; memcpy(&km, p + 16, 8);
        d.contig_len.push_back(len); d.contig_kc.push_back(get32(p + 8)); d.contig_km.push_back(km);
        d.contig_off.push_back(d.contig_packed.size());
        const size_t base = d.contig_packed.size();
        d.contig_packed.resize(base + (((size_t)len + 15) / 16) * 4, 0);
        from_ref_codec(blk.data() + off, len, d.contig_packed.data() + base);
        off += cl;
    }
    d.contig_off.push_back(d.contig_packed.size());
    finish_contigs(d);
    return true;
}

bool read_longread_index(Dataset& d, const std::string& path) {
    File f(path, "rb");
    if (!f.fp) { g_err = "[ERROR] (Longread::read_longread_index) could not open file: " + path; return false; }
    auto trunc = [&]() { g_err = "[ERROR] truncated or inconsistent " + path; return false; };
    uint64_t n = 0, seqs = 0, na = 0, cs = 0;
    if (!f.r(&n, 8)) return trunc();
    std::vector<uint8_t> recs(n * 32);
    if (!f.r(recs.data(), recs.size()) || !f.r(&seqs, 8)) return trunc();
    std::vector<uint8_t> sq(seqs);
    if (!f.r(sq.data(), seqs) || !f.r(&na, 8)) return trunc();
    std::vector<uint8_t> al(na * 48);
    if (!f.r(al.data(), al.size()) || !f.r(&cs, 8)) return trunc();
    std::vector<char> cg(cs);
    if (!f.r(cg.data(), cs)) return trunc();
    uint64_t off = 0, a = 0, coff = 0;
    d.read_hit_off.assign(1, 0);
    for (uint64_t i = 0; i < n; i++) {
        const uint8_t* p = recs.data() + i * 32;
        const uint32_t len = get32(p), cl = get32(p + 4), nal = get32(p + 24);
        if (cl != comp_len_of(len) || off + cl > seqs || a + nal > na) return trunc();
        d.read_len.push_back(len); d.total_read_bases += len;
        d.read_off.push_back(d.read_packed.size());
        const size_t base = d.read_packed.size();
        d.read_packed.resize(base + (((size_t)len + 15) / 16) * 4, 0);
        from_ref_codec(sq.data() + off, len, d.read_packed.data() + base);
        off += cl;
        for (uint32_t k = 0; k < nal; k++, a++) {
            const uint8_t* q = al.data() + a * 48;
            const uint32_t tid = get32(q + 12), cgl = get32(q + 36);
            if (tid >= d.contig_len.size() || coff + cgl + 1 > cs) return trunc();
            d.q_id.push_back(get32(q)); d.q_start.push_back(get32(q + 4)); d.q_end.push_back(get32(q + 8));
            d.t_id.push_back(tid); d.t_len.push_back(d.contig_len[tid]); d.t_start.push_back(get32(q + 16)); d.t_end.push_back(get32(q + 20));
            d.n_match.push_back(get32(q + 24)); d.n_block.push_back(get32(q + 28));
            d.is_rev.push_back(q[32]); d.mapq.push_back(q[33]);
            if (!append_cigar(d, cg.data() + coff, cg.data() + coff + cgl)) return false;
            coff += cgl + 1;
        }
        d.read_hit_off.push_back(d.q_id.size());
    }
    d.read_off.push_back(d.read_packed.size());
    d.cg_off.push_back(d.cg_ops.size());
    if (d.cg_ops.empty()) d.cg_ops.push_back(0);
    return true;
}

}  // namespace hxh
