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

#include <unistd.h>

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
void from_ref_codec(const uint8_t* src, uint32_t len, uint8_t* mine /* ((len+15)/16)*4 bytes, all of them written */) {
    // inverse of to_ref_codec: the first len % 4 bases come from the last byte, then one group of four per byte going backwards; the
    // groups are appended to the little-endian 2-bit stream through a small bit accumulator
    const uint32_t cl = comp_len_of(len), r = len % 4, groups = len / 4;
    const size_t nbytes = (((size_t)len + 15) / 16) * 4;
    uint32_t acc = r ? (uint32_t)(src[cl - 1] & ((1u << (2 * r)) - 1)) : 0, nb = 2 * r;
    const uint8_t* in = src + cl - 1 - (r ? 1 : 0);
    size_t o = 0;
    for (uint32_t g = 0; g < groups; g++) {
        acc |= (uint32_t)(*in--) << nb;
        mine[o++] = (uint8_t)acc;
        acc >>= 8;
    }
    if (nb) mine[o++] = (uint8_t)acc;
    while (o < nbytes) mine[o++] = 0;
}

struct File {
    FILE* fp;
    uint64_t size = 0, pos = 0;   // (readers: a count that promises more bytes than the file has left is refused before anything is allocated for it)
    std::string final_path, part_path;
    File(const std::string& path, const char* mode) : fp(nullptr) {
        if (mode[0] == 'w') {   // writers fill "<path>.part"; commit() renames it, so a run that is killed half way leaves no cache file that looks complete
            final_path = path; part_path = path + ".part";
            fp = fopen(part_path.c_str(), mode);
        } else {
            fp = fopen(path.c_str(), mode);
            if (fp && fseeko(fp, 0, SEEK_END) == 0) { size = (uint64_t)ftello(fp); fseeko(fp, 0, SEEK_SET); }
        }
    }
    ~File() { if (fp) fclose(fp); if (!part_path.empty()) remove(part_path.c_str()); }
    bool w(const void* p, size_t n) { return n == 0 || fwrite(p, 1, n, fp) == n; }
    bool r(void* p, size_t n) { if (n == 0) return true; if (fread(p, 1, n, fp) != n) return false; pos += n; return true; }
    bool has(uint64_t count, uint64_t each) const { return each == 0 || count <= (size - pos) / each; }
    bool commit() {
        const bool ok = fclose(fp) == 0;
        fp = nullptr;
        if (ok && rename(part_path.c_str(), final_path.c_str()) == 0) { part_path.clear(); return true; }
        return false;
    }
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
    if (!f.w(&n, 8) || !f.w(recs.data(), recs.size()) || !f.w(&block, 8) || !f.w(blk.data(), blk.size()) || !f.commit()) { g_err = "[ERROR] could not write " + path; return false; }
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
    // The two big blocks (every read re-coded, every kept CIGAR spelled out: 3.3 GB at 140 Mb) are produced and written by several threads AT ONCE, each its own
    // range of the file (round 6; the lengths of every text are counted first - the alignment records carry them - so every byte's place in the file is known
    // before any is produced): chunks of a few megabytes through a buffer per thread that is reused, positional writes. (Until round 5: eight threads produced a
    // chunk, one wrote it: 1.35 s at 140 Mb, more than what is left of a one-shot run when the filtered set is known.)
    const unsigned T = std::max(1u, std::min(g_io_threads, 16u));
    const uint64_t na = ch.n_aln;
    std::vector<uint8_t> al(na * 48, 0);
    std::vector<uint64_t> cg_at(na + 1, 0);   // place of every alignment's text in the block (each text is followed by a NUL)
    run_parallel(T, [&](unsigned t) {
        for (uint64_t a = na * t / T; a < na * (t + 1) / T; a++) {
            const uint32_t h = ch.hit[a];
            uint8_t* p = al.data() + a * 48;
            put32(p, d.q_id[h]); put32(p + 4, d.q_start[h]); put32(p + 8, d.q_end[h]); put32(p + 12, d.t_id[h]);
            put32(p + 16, d.t_start[h]); put32(p + 20, d.t_end[h]); put32(p + 24, d.n_match[h]); put32(p + 28, d.n_block[h]);
            p[32] = d.is_rev[h]; p[33] = d.mapq[h]; p[34] = 0;
            uint64_t len = 0;
            auto it = d.cg_text_odd.empty() ? d.cg_text_odd.end() : d.cg_text_odd.find(h);
            if (it != d.cg_text_odd.end()) len = it->second.size();
            else for (uint64_t k = d.cg_off[h]; k < d.cg_off[h + 1]; k++) { const uint32_t v = HX_CG_LEN(d.cg_ops[k]); len += 2 + (v >= 10) + (v >= 100) + (v >= 1000) + (v >= 10000) + (v >= 100000) + (v >= 1000000) + (v >= 10000000) + (v >= 100000000) + (v >= 1000000000); }
            put32(p + 36, (uint32_t)len);
            cg_at[a + 1] = len + 1;
        }
    });
    for (uint64_t a = 0; a < na; a++) cg_at[a + 1] += cg_at[a];
    const uint64_t cs = cg_at[na];
    std::vector<uint64_t> sq_at(n + 1, 0);
    for (uint64_t i = 0; i < n; i++) sq_at[i + 1] = sq_at[i] + comp_len_of(d.read_len[i]);
    // layout: n | records | seqs | <sequences> | na | alignments | cs | <texts>
    const uint64_t o_seq = 8 + recs.size() + 8, o_na = o_seq + seqs, o_cg = o_na + 8 + al.size() + 8;
    bool ok = f.w(&n, 8) && f.w(recs.data(), recs.size()) && f.w(&seqs, 8) && fflush(f.fp) == 0;
    const int fd = fileno(f.fp);
    auto put = [fd](const void* p, uint64_t bytes, uint64_t at) -> bool {
        const char* q = (const char*)p;
        while (bytes) {
            const ssize_t w = pwrite(fd, q, bytes, (off_t)at);
            if (w <= 0) return false;
            q += w; at += (uint64_t)w; bytes -= (uint64_t)w;
        }
        return true;
    };
    ok = ok && put(&na, 8, o_na) && put(al.data(), al.size(), o_na + 8) && put(&cs, 8, o_na + 8 + al.size());
    std::vector<uint8_t> bad(T, 0);
    run_parallel(T, [&](unsigned t) {
        const uint64_t CHUNK = 8ull << 20;
        U8Arena buf;
        // sequences of this thread's reads, a chunk at a time
        for (uint64_t i0 = n * t / T, iend = n * (t + 1) / T; i0 < iend && !bad[t];) {
            uint64_t i1 = i0;
            while (i1 < iend && (sq_at[i1] - sq_at[i0] < CHUNK || i1 == i0)) i1++;
            const uint64_t bytes = sq_at[i1] - sq_at[i0];
            if (buf.size() < bytes) buf.resize(bytes);
            for (uint64_t i = i0; i < i1; i++) to_ref_codec(d.read_packed.data() + d.read_off[i], d.read_len[i], buf.data() + (sq_at[i] - sq_at[i0]));
            if (!put(buf.data(), bytes, o_seq + sq_at[i0])) bad[t] = 1;
            i0 = i1;
        }
        // CIGAR texts of this thread's alignments, each followed by a NUL
        std::string out;
        for (uint64_t a0 = na * t / T, aend = na * (t + 1) / T; a0 < aend && !bad[t];) {
            uint64_t a1 = a0;
            while (a1 < aend && (cg_at[a1] - cg_at[a0] < CHUNK || a1 == a0)) a1++;
            out.clear();
            for (uint64_t a = a0; a < a1; a++) { append_cigar_text(d, ch.hit[a], out); out.push_back('\0'); }
            if (out.size() != cg_at[a1] - cg_at[a0] || !put(out.data(), out.size(), o_cg + cg_at[a0])) bad[t] = 1;
            a0 = a1;
        }
    });
    for (uint8_t x : bad) ok = ok && !x;
    if (ok && o_cg + cs > 0 && cs == 0) ok = ftruncate(fd, (off_t)o_cg) == 0;   // (no alignment at all: the file still ends behind the empty block's size word)
    if (!ok || !f.commit()) { g_err = "[ERROR] could not write " + path; return false; }
    return true;
}

bool read_contig_index(Dataset& d, const std::string& path) {
    File f(path, "rb");
    if (!f.fp) { g_err = "[ERROR] (Contig::read_contig_index) could not open file: " + path; return false; }
    uint64_t n = 0, block = 0;
    if (!f.r(&n, 8) || !f.has(n, 32)) { g_err = "[ERROR] truncated " + path; return false; }
    std::vector<uint8_t> recs(n * 32);
    if (!f.r(recs.data(), recs.size()) || !f.r(&block, 8) || !f.has(block, 1)) { g_err = "[ERROR] truncated " + path; return false; }
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
    if (!f.r(&n, 8) || !f.has(n, 32) || n >= 0xffffffffull) return trunc();
    std::vector<uint8_t> recs(n * 32);
    if (!f.r(recs.data(), recs.size()) || !f.r(&seqs, 8) || !f.has(seqs, 1)) return trunc();
    std::vector<uint8_t> sq(seqs);
    if (!f.r(sq.data(), seqs) || !f.r(&na, 8) || !f.has(na, 48)) return trunc();
    std::vector<uint8_t> al(na * 48);
    if (!f.r(al.data(), al.size()) || !f.r(&cs, 8) || !f.has(cs, 1)) return trunc();
    std::vector<char> cg(cs);
    if (!f.r(cg.data(), cs)) return trunc();
    // serial: validate the record tables, lay out the arenas. parallel (reads / alignments dealt in contiguous ranges to the ingest
    // threads): re-code every read, spell every CIGAR into op words (per-thread parts, stitched in order like the PAF loader's)
    if (!d.read_len.empty() || !d.q_id.empty()) { g_err = "[ERROR] index.longread is loaded into an empty data set only"; return false; }
    std::vector<uint64_t> sq_off(n + 1, 0), al0(n + 1, 0);
    d.read_len.resize(n); d.read_off.resize(n + 1); d.read_hit_off.assign(n + 1, 0);
    uint64_t base = 0;
    for (uint64_t i = 0; i < n; i++) {
        const uint8_t* p = recs.data() + i * 32;
        const uint32_t len = get32(p), cl = get32(p + 4), nal = get32(p + 24);
        if (cl != comp_len_of(len) || sq_off[i] + cl > seqs || al0[i] + nal > na) return trunc();
        sq_off[i + 1] = sq_off[i] + cl; al0[i + 1] = al0[i] + nal;
        d.read_len[i] = len; d.total_read_bases += len;
        d.read_off[i] = base; base += (((size_t)len + 15) / 16) * 4;
        d.read_hit_off[i + 1] = al0[i + 1];
    }
    d.read_off[n] = base;
    const uint64_t nrec = al0[n];
    std::vector<uint64_t> cg0(nrec + 1, 0);   // start of every alignment's text
    for (uint64_t a = 0; a < nrec; a++) {
        const uint32_t cgl = get32(al.data() + a * 48 + 36);
        if (cg0[a] + cgl + 1 > cs || get32(al.data() + a * 48 + 12) >= d.contig_len.size()) return trunc();
        cg0[a + 1] = cg0[a] + cgl + 1;
    }
    d.read_packed.resize(base);
    d.q_id.resize(nrec); d.q_start.resize(nrec); d.q_end.resize(nrec); d.t_id.resize(nrec); d.t_len.resize(nrec); d.t_start.resize(nrec); d.t_end.resize(nrec);
    d.n_match.resize(nrec); d.n_block.resize(nrec); d.is_rev.resize(nrec); d.mapq.resize(nrec); d.cg_off.resize(nrec + 1);
    const unsigned T = std::max(1u, g_io_threads);
    struct Part { U32Arena ops; std::vector<uint64_t> off; std::vector<std::pair<uint64_t, std::string>> odd; bool too_long = false; };
    std::vector<Part> part(T);
    run_parallel(T, [&](unsigned t) {
        for (uint64_t i = n * t / T; i < n * (t + 1) / T; i++) from_ref_codec(sq.data() + sq_off[i], d.read_len[i], d.read_packed.data() + d.read_off[i]);
        Part& P = part[t];
        for (uint64_t a = nrec * t / T; a < nrec * (t + 1) / T; a++) {
            const uint8_t* q = al.data() + a * 48;
            const uint32_t tid = get32(q + 12);
            d.q_id[a] = get32(q); d.q_start[a] = get32(q + 4); d.q_end[a] = get32(q + 8);
            d.t_id[a] = tid; d.t_len[a] = d.contig_len[tid]; d.t_start[a] = get32(q + 16); d.t_end[a] = get32(q + 20);
            d.n_match[a] = get32(q + 24); d.n_block[a] = get32(q + 28);
            d.is_rev[a] = q[32]; d.mapq[a] = q[33];
            P.off.push_back(P.ops.size());
            bool odd, too_long;
            const char* b = cg.data() + cg0[a];
            const char* e = cg.data() + cg0[a + 1] - 1;
            if (!parse_cigar_ops(b, e, P.ops, odd, too_long)) { P.too_long = true; return; }
            if (odd) P.odd.push_back({a, std::string(b, e)});
        }
    });
    std::vector<uint64_t> op0(T + 1, 0);
    for (unsigned t = 0; t < T; t++) {
        if (part[t].too_long) { g_err = "[ERROR] CIGAR operation longer than 2^30 in index.longread"; return false; }
        op0[t + 1] = op0[t] + part[t].ops.size();
    }
    d.cg_ops.resize(std::max<uint64_t>(1, op0[T]));
    if (!op0[T]) d.cg_ops[0] = 0;
    run_parallel(T, [&](unsigned t) {
        const Part& P = part[t];
        const uint64_t a0 = nrec * t / T;
        for (size_t k = 0; k < P.off.size(); k++) d.cg_off[a0 + k] = P.off[k] + op0[t];
        if (!P.ops.empty()) memcpy(d.cg_ops.data() + op0[t], P.ops.data(), P.ops.size() * 4);
    });
    d.cg_off[nrec] = op0[T];
    for (unsigned t = 0; t < T; t++) for (auto& o : part[t].odd) d.cg_text_odd[o.first] = o.second;
    return true;
}

}  // namespace hxh
