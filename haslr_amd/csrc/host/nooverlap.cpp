// minia_nooverlap — trims the (k-1)-base overlaps Minia leaves between linked unitigs / contigs, so that neighbouring short-read
// contigs do not share sequence (behaviour of the reference's src/minia_nooverlap/nooverlap.cpp:30-89).
//
//   minia_nooverlap <assembly.fa[.gz]> <kmerSize>   > trimmed.fa
//
// Every record is printed as ">name comment" + one sequence line. The comment holds three Minia fields (LN:i: KC:i: km:f:) and then
// the links, "L:<from strand>:<to id>:<to strand>": a link leaving on '+' means the record has a successor, one leaving on '-' a
// predecessor (nooverlap.cpp:62-71). With h = (k-1)/2, a record with a predecessor loses its first h bases and one with a successor
// its last h (nooverlap.cpp:75-83) — each side of an overlap gives up half of it.
//
// The record grammar is the one of the reader the reference uses (kseq.h:183-219 in the same directory), restated on a buffer that
// holds the whole (decompressed) file: the name ends at the first white-space character; the comment is the rest of the header line;
// sequence lines are concatenated as they are; a line starting with '+' opens FASTQ qualities, which are skipped. Two consequences of
// that reader are kept because a drop-in must print the same bytes: a header WITHOUT a comment prints (and links are taken from) the
// previous record's comment, and a truncated FASTQ record ends the output without an error. Where the reference dies on an uncaught
// exception (no comment on the very first record; a predecessor trim longer than the sequence) this tool exits with an error message.
//
// Pinned byte for byte against the compiled reference tool (oracle/_ref/ref_nooverlap) by tests/test_nooverlap.py.
#include <zlib.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>

namespace {

bool slurp(const char* path, std::string& out) {
    gzFile f = gzopen(path, "r");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    char buf[1 << 16];
    int n;
    while ((n = gzread(f, buf, sizeof(buf))) > 0) out.append(buf, (size_t)n);
    gzclose(f);
    return true;
}

struct Records {
    const std::string& d;
    size_t p = 0;
    bool at_header = false;          // the record's first character ('>' / '@') has been consumed already
    bool has_comment = false;        // a comment has been seen on some record so far
    std::string name, comment, seq;
    explicit Records(const std::string& data) : d(data) {}

    // appends [p, end of line) to s, moves past the newline; a '\r' that ends the collected text is dropped when it is not the only character
    bool rest_of_line(std::string& s) {
        if (p >= d.size()) return false;
        size_t e = d.find('\n', p);
        if (e == std::string::npos) e = d.size();
        s.append(d, p, e - p);
        p = e < d.size() ? e + 1 : e;
        if (s.size() > 1 && s.back() == '\r') s.pop_back();
        return true;
    }

    // 1 = a record was read, 0 = end of input
    int next() {
        if (!at_header) {
            while (p < d.size() && d[p] != '>' && d[p] != '@') p++;
            if (p >= d.size()) return 0;
            p++;
        }
        at_header = false;
        if (p >= d.size()) return 0;
        size_t e = p;
        while (e < d.size() && !isspace((unsigned char)d[e])) e++;
        name.assign(d, p, e - p);
        const char sep = e < d.size() ? d[e] : '\n';
        p = e < d.size() ? e + 1 : e;
        if (sep != '\n') { comment.clear(); rest_of_line(comment); has_comment = true; }
        seq.clear();
        char c = 0;
        while (p < d.size()) {
            c = d[p++];
            if (c == '>' || c == '+' || c == '@') break;
            if (c == '\n') { c = 0; continue; }
            seq.push_back(c);
            rest_of_line(seq);
            c = 0;
        }
        if (c == '>' || c == '@') at_header = true;
        if (c != '+') return 1;
        // FASTQ: skip the '+' line, then as many quality characters as there are bases
        size_t e2 = d.find('\n', p);
        if (e2 == std::string::npos) return 0;                     // no quality string: the reference's loop stops here
        p = e2 + 1;
        std::string qual;
        do { if (!rest_of_line(qual)) break; } while (qual.size() < seq.size());   // (one line is read even for an empty sequence)
        if (qual.size() != seq.size()) return 0;
        return 1;
    }
};

}  // namespace

int main(int argc, char** argv) {
    const char* usage = "usage: ./nooverlap unitigs.fa kmerSize\n";
    if (argc == 2 && strcmp(argv[1], "-h") == 0) { fputs(usage, stderr); return EXIT_SUCCESS; }
    if (argc < 3) { fputs(usage, stderr); return EXIT_FAILURE; }
    int k = 0;
    { std::istringstream in(argv[2]); in >> k; }                   // the reference's conversion: 0 when the text is not a number
    const long half = ((long)k - 1) / 2;
    std::string data;
    if (!slurp(argv[1], data)) { fprintf(stderr, "[ERROR] could not open file: %s\n", argv[1]); return EXIT_FAILURE; }
    Records r(data);
    std::string out;
    out.reserve(data.size() + 1024);
    while (r.next() == 1) {
        if (!r.has_comment) {
            fwrite(out.data(), 1, out.size(), stdout);
            fprintf(stderr, "[ERROR] record %s has no comment (expected Minia's LN:i: KC:i: km:f: L:... fields)\n", r.name.c_str());
            return EXIT_FAILURE;
        }
        out += '>'; out += r.name; out += ' '; out += r.comment; out += '\n';
        bool pred = false, succ = false;
        std::istringstream fields(r.comment);
        std::string tok;
        for (int i = 0; i < 3; i++) fields >> tok;
        while (fields >> tok) {
            if (tok.size() < 3) continue;
            if (tok[2] == '+') succ = true;
            else if (tok[2] == '-') pred = true;
        }
        size_t b = 0, e = r.seq.size();
        if (pred) {
            if (half < 0 || (size_t)half > e) {
                fwrite(out.data(), 1, out.size(), stdout);
                fprintf(stderr, "[ERROR] record %s is shorter than half the overlap (%ld)\n", r.name.c_str(), half);
                return EXIT_FAILURE;
            }
            b = (size_t)half;
        }
        if (succ && half >= 0 && (size_t)half <= e - b) e -= (size_t)half;   // (a longer trim than what is left keeps it all, as the reference's unsigned arithmetic does)
        out.append(r.seq, b, e - b);
        out += '\n';
        if (out.size() > (1u << 22)) { fwrite(out.data(), 1, out.size(), stdout); out.clear(); }
    }
    fwrite(out.data(), 1, out.size(), stdout);
    return EXIT_SUCCESS;
}
