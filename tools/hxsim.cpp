// hxsim — seeded synthetic input generator for the haslr_assemble stage.
//
// Produces what haslr.py would hand to haslr_assemble (bin/haslr.py:66 in the
// reference): a short-read-contig FASTA with Minia-style "KC:i:/km:f:" comments,
// a long-read FASTA whose names are the ordinals 0..N-1, and a minimap2-style
// PAF ("-c", so with cg:Z: CIGARs) of reads against contigs. Nothing here is
// taken from the reference; minimap2/minia are not available in the build image,
// so the PAF is synthesised from the simulation's own truth.
//
// The generator is build-owned test/bench infrastructure. It is deterministic for
// a given (seed, options): its own xoshiro256** PRNG, no std:: distributions.
//
// --chromosomes K (round 5): the genome is K independent chromosomes of genome-len / K bases, simulated by --threads T threads (chromosome k
// with seed + 7919 k), written in chromosome order with contig and read ids running on - 140 Mb took 42 s on one thread. K = 1 (the default)
// writes exactly what the single-threaded generator always wrote. --filler-reads N --filler-len L: N unreferenced reads of L bases ('A') ahead of
// the real ones (test: a packed read arena above 4 GiB, every supporting read at a byte offset above 2^32).
//
// Planted features (each exercises a reference code path, cited in DESIGN.md):
//   * repeat families collapsed into one contig with km ~ 30*copies
//   * contigs < 250 bp (never mapped) and 250..500 bp (fail --aln-block)
//   * negative gaps (adjacent contigs overlap on the genome => read-space overlap)
//   * variant molecules: deletion / substitution haplotypes (bubbles), dead-end
//     branches (tips), low-coverage branches (weak edges)
//   * both read strands, both contig strands, hits below identity / mapq cut-offs
#include <algorithm>
#include <cinttypes>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) { for (auto& v : s) v = splitmix(seed); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return n ? next() % n : 0; }
    double normal() {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
    double lognormal(double median, double sigma) { return median * std::exp(sigma * normal()); }
};

const char kBases[5] = "ACGT";
inline char comp(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; }
    return 'N';
}
std::string revcomp(const std::string& s) {
    std::string r(s.size(), 'N');
    for (size_t i = 0; i < s.size(); i++) r[i] = comp(s[s.size() - 1 - i]);
    return r;
}

struct Contig {
    std::string seq;
    double km;
    uint32_t kc;
};

// A stretch of some molecule-independent "source axis" that is represented by a contig.
// Source axes: 0 = the main genome; 1+k = novel segment k (exists only in variant molecules).
struct Placement {
    uint32_t axis;
    int64_t s, e;       // [s,e) on the axis
    uint32_t contig;    // index before id shuffling
    bool rev;           // contig sequence = revcomp(axis[s,e))
    int64_t coff;       // contig coordinate of axis position s (fwd) / of axis position e-1 (rev)
    uint8_t mapq_lo, mapq_hi;
};

struct Piece { uint32_t axis; int64_t s, e; };   // forward slice of an axis
struct Molecule { std::vector<Piece> pieces; double cov; };

struct Opts {
    uint64_t seed = 0x4841534cULL;
    int64_t glen = 300000;
    double cov = 25.0;
    std::string model = "pacbio";
    std::string prefix = "sim";
    double contig_median = 8000, gap_median = 600;
    double read_median = 9000, read_sigma = 0.5;
    int64_t read_cap = 120000;
    double variant_per_mb = 6.0;   // planted variant molecules per Mb, per kind
    bool variants = true;
    double hairpin_frac = 0;       // fraction of reads that continue with the reverse complement of their own tail (missed-adapter reads:
                                   // the same contigs hit twice on opposite strands, Longread.cpp:182-232's palindrome rule)
};

struct ErrModel { double ins, del, sub; };

struct Hit { uint32_t qs, qe, tid, tlen, ts, te, nm, nb; bool rev; uint8_t mapq; std::string cg; };

// where a simulation's products go: straight to the files (one chromosome), or into memory with local ids (one of several chromosomes)
struct Sink {
    virtual ~Sink() {}
    virtual void contigs(const std::vector<Contig>& contigs, const std::vector<uint32_t>& order) = 0;   // order[id] = index of the contig with that id
    virtual void genome(const std::string& g) = 0;
    virtual void read(const std::string& seq, uint32_t rlen, const std::vector<Hit>& hits) = 0;
};
struct FileSink : Sink {
    std::string prefix; FILE *fr = nullptr, *fp = nullptr; uint64_t read_id = 0;
    bool open_reads() { fr = fopen((prefix + ".reads.fa").c_str(), "w"); fp = fopen((prefix + ".paf").c_str(), "w"); return fr && fp; }
    void contigs(const std::vector<Contig>& cs, const std::vector<uint32_t>& order) override {
        FILE* fc = fopen((prefix + ".contigs.fa").c_str(), "w");
        if (!fc) { perror("hxsim: contigs"); exit(1); }
        for (size_t id = 0; id < order.size(); id++) {
            const Contig& c = cs[order[id]];
            fprintf(fc, ">%zu LN:i:%zu KC:i:%u km:f:%.3f\n%s\n", id, c.seq.size(), c.kc, c.km, c.seq.c_str());
        }
        fclose(fc);
    }
    void genome(const std::string& g) override {
        FILE* fg = fopen((prefix + ".genome.fa").c_str(), "w");
        if (!fg) { perror("hxsim: genome"); exit(1); }
        fprintf(fg, ">genome\n%s\n", g.c_str());
        fclose(fg);
    }
    void read(const std::string& seq, uint32_t rlen, const std::vector<Hit>& hits) override {
        fprintf(fr, ">%" PRIu64 "\n%s\n", read_id, seq.c_str());
        for (const Hit& h : hits)
            fprintf(fp, "%" PRIu64 "\t%u\t%u\t%u\t%c\t%u\t%u\t%u\t%u\t%u\t%u\t%u\ttp:A:P\tcm:i:%u\ts1:i:%u\tcg:Z:%s\n",
                    read_id, rlen, h.qs, h.qe, h.rev ? '-' : '+', h.tid, h.tlen, h.ts, h.te, h.nm, h.nb,
                    (unsigned)h.mapq, h.nm / 10, h.nm, h.cg.c_str());
        read_id++;
    }
};
struct MemSink : Sink {
    struct Rd { std::string seq; uint32_t rlen; std::vector<Hit> hits; };
    std::vector<Contig> cs; std::string g; std::vector<Rd> reads;   // contigs in id order (local ids)
    void contigs(const std::vector<Contig>& c, const std::vector<uint32_t>& order) override { cs.reserve(order.size()); for (uint32_t i : order) cs.push_back(c[i]); }
    void genome(const std::string& gg) override { g = gg; }
    void read(const std::string& seq, uint32_t rlen, const std::vector<Hit>& hits) override { reads.push_back({seq, rlen, hits}); }
};

struct SimStats { size_t contigs = 0, families = 0, molecules = 0; uint64_t reads = 0, bases = 0, hits = 0; };
int simulate(Opts o, Sink& sink, SimStats& st);

}  // namespace

int main(int argc, char** argv) {
    Opts o;
    int chromosomes = 1, threads = 0;
    uint64_t filler_reads = 0, filler_len = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&](const char* name) -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "hxsim: %s needs a value\n", name); exit(2); }
            return argv[++i];
        };
        if (a == "--seed") o.seed = strtoull(val("--seed"), nullptr, 0);
        else if (a == "--genome-len") o.glen = atoll(val("--genome-len"));
        else if (a == "--cov") o.cov = atof(val("--cov"));
        else if (a == "--model") o.model = val("--model");
        else if (a == "--out-prefix") o.prefix = val("--out-prefix");
        else if (a == "--contig-median") o.contig_median = atof(val("--contig-median"));
        else if (a == "--gap-median") o.gap_median = atof(val("--gap-median"));
        else if (a == "--read-median") o.read_median = atof(val("--read-median"));
        else if (a == "--variant-per-mb") o.variant_per_mb = atof(val("--variant-per-mb"));
        else if (a == "--no-variants") o.variants = false;
        else if (a == "--hairpin-frac") o.hairpin_frac = atof(val("--hairpin-frac"));
        else if (a == "--chromosomes") chromosomes = std::max(1, atoi(val("--chromosomes")));
        else if (a == "--threads") threads = atoi(val("--threads"));
        else if (a == "--filler-reads") filler_reads = strtoull(val("--filler-reads"), nullptr, 0);
        else if (a == "--filler-len") filler_len = strtoull(val("--filler-len"), nullptr, 0);
        else { fprintf(stderr, "hxsim: unknown option %s\n", a.c_str()); return 2; }
    }
    if (o.model != "nanopore" && o.model != "perfect" && o.model != "pacbio") { fprintf(stderr, "hxsim: unknown model %s\n", o.model.c_str()); return 2; }
    if (filler_reads && (filler_len < 1 || filler_len > 0xfffffff0ull)) { fprintf(stderr, "hxsim: --filler-len must be 1 .. 2^32 - 16\n"); return 2; }
    SimStats tot;
    FileSink out;
    out.prefix = o.prefix;
    if (!out.open_reads()) { perror("hxsim: reads/paf"); return 1; }
    if (filler_reads) {   // unreferenced reads ahead of the real ones: no PAF record names them
        std::string blk((size_t)std::min<uint64_t>(filler_len, 1u << 24), 'A');
        for (uint64_t f = 0; f < filler_reads; f++) {
            fprintf(out.fr, ">%" PRIu64 "\n", out.read_id++);
            for (uint64_t w = 0; w < filler_len; w += blk.size()) fwrite(blk.data(), 1, (size_t)std::min<uint64_t>(blk.size(), filler_len - w), out.fr);
            fputc('\n', out.fr);
        }
    }
    if (chromosomes == 1) {
        if (int rc = simulate(o, out, tot)) return rc;
    } else {
        // K chromosomes on T threads into memory (local ids), then written in chromosome order with the ids running on
        if (threads <= 0) threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), (unsigned)chromosomes);
        std::vector<MemSink> ms((size_t)chromosomes);
        std::vector<SimStats> sts((size_t)chromosomes);
        std::vector<int> rcs((size_t)chromosomes, 0);
        std::atomic<int> next{0};
        auto work = [&]() {
            for (;;) {
                const int k = next.fetch_add(1);
                if (k >= chromosomes) break;
                Opts ok = o;
                ok.glen = o.glen / chromosomes; ok.seed = o.seed + 7919ull * (uint64_t)k;
                rcs[(size_t)k] = simulate(ok, ms[(size_t)k], sts[(size_t)k]);
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back(work);
        for (auto& t : th) t.join();
        for (int rc : rcs) if (rc) return rc;
        FILE* fc = fopen((o.prefix + ".contigs.fa").c_str(), "w");
        FILE* fg = fopen((o.prefix + ".genome.fa").c_str(), "w");
        if (!fc || !fg) { perror("hxsim: contigs/genome"); return 1; }
        size_t cbase = 0;
        for (int k = 0; k < chromosomes; k++) {
            MemSink& m = ms[(size_t)k];
            for (size_t id = 0; id < m.cs.size(); id++) fprintf(fc, ">%zu LN:i:%zu KC:i:%u km:f:%.3f\n%s\n", cbase + id, m.cs[id].seq.size(), m.cs[id].kc, m.cs[id].km, m.cs[id].seq.c_str());
            fprintf(fg, ">chr%d\n%s\n", k, m.g.c_str());
            for (MemSink::Rd& r : m.reads) {
                for (Hit& h : r.hits) h.tid += (uint32_t)cbase;
                out.read(r.seq, r.rlen, r.hits);
            }
            cbase += m.cs.size();
            tot.contigs += sts[(size_t)k].contigs; tot.families += sts[(size_t)k].families; tot.molecules += sts[(size_t)k].molecules;
            tot.reads += sts[(size_t)k].reads; tot.bases += sts[(size_t)k].bases; tot.hits += sts[(size_t)k].hits;
            m = MemSink();
        }
        fclose(fc); fclose(fg);
    }
    fclose(out.fr); fclose(out.fp);
    fprintf(stderr, "hxsim: genome=%" PRId64 " contigs=%zu (repeat families=%zu) molecules=%zu reads=%" PRIu64 " bases=%" PRIu64 " hits=%" PRIu64 "\n",
            o.glen, tot.contigs, tot.families, tot.molecules, tot.reads, tot.bases, tot.hits);
    return 0;
}

namespace {
int simulate(Opts o, Sink& sink, SimStats& st) {
    ErrModel em{0.08, 0.03, 0.02};
    if (o.model == "nanopore") { em = {0.03, 0.05, 0.04}; o.read_median = 7000; o.read_sigma = 0.8; }
    else if (o.model == "perfect") em = {0, 0, 0};

    Rng rng(o.seed);
    // ---------------------------------------------------------------- genome + repeats
    std::vector<std::string> axes(1);   // axes may grow (novel segments): never hold a reference into it
    axes[0].resize(o.glen);
    for (auto& c : axes[0]) c = kBases[rng.next() & 3];

    struct RepCopy { int64_t s, e; uint32_t fam; bool rev; };
    std::vector<RepCopy> copies;
    std::vector<std::string> fam_unit;
    std::vector<uint32_t> fam_copies;
    {
        const int copy_choices[] = {2, 2, 2, 3, 4, 6, 10, 20};
        int64_t budget = (int64_t)(o.glen * 0.035);
        std::vector<std::pair<int64_t, int64_t>> taken;
        int guard = 0;
        while (budget > 0 && guard++ < 10000) {
            int64_t U = 1000 + (int64_t)rng.below(5000);
            int c = copy_choices[rng.below(8)];
            if (U * c > budget + 6000) { if (U * 2 > budget + 6000) break; c = 2; }
            std::string unit(U, 'A');
            for (auto& ch : unit) ch = kBases[rng.next() & 3];
            uint32_t fam = fam_unit.size();
            int placed = 0;
            for (int k = 0; k < c; k++) {
                for (int tries = 0; tries < 50; tries++) {
                    int64_t s = 2000 + (int64_t)rng.below(std::max<int64_t>(1, o.glen - U - 4000));
                    if (s + U + 2000 > o.glen) continue;
                    bool ok = true;
                    for (auto& t : taken) if (s < t.second + 1500 && t.first < s + U + 1500) { ok = false; break; }
                    if (!ok) continue;
                    bool rev = rng.next() & 1;
                    std::string ins = rev ? revcomp(unit) : unit;
                    memcpy(&axes[0][s], ins.data(), U);
                    taken.push_back({s, s + U});
                    copies.push_back({s, s + U, fam, rev});
                    placed++;
                    break;
                }
            }
            if (placed == 0) continue;
            fam_unit.push_back(unit);
            fam_copies.push_back(placed);
            budget -= U * placed;
        }
        std::sort(copies.begin(), copies.end(), [](const RepCopy& a, const RepCopy& b) { return a.s < b.s; });
    }

    // ---------------------------------------------------------------- contigs on the main genome
    std::vector<Contig> contigs;
    std::vector<Placement> places;
    auto add_unique_contig = [&](uint32_t axis, int64_t s, int64_t e) {
        bool rev = rng.next() & 1;
        Contig c;
        c.seq = axes[axis].substr(s, e - s);
        if (rev) c.seq = revcomp(c.seq);
        c.km = 30.0 * (0.97 + 0.06 * rng.uni());
        int64_t L = e - s;
        c.kc = L > 48 ? (uint32_t)std::llround(c.km * (L - 48)) : 0;
        places.push_back({axis, s, e, (uint32_t)contigs.size(), rev, rev ? L - 1 : 0, 60, 60});
        contigs.push_back(std::move(c));
    };
    // repeat family contigs first (so that their index is known), one per family
    std::vector<uint32_t> fam_contig(fam_unit.size());
    for (size_t f = 0; f < fam_unit.size(); f++) {
        Contig c;
        c.seq = fam_unit[f];
        c.km = 30.0 * fam_copies[f] * (0.97 + 0.06 * rng.uni());
        c.kc = (uint32_t)std::llround(c.km * ((int64_t)c.seq.size() - 48));
        fam_contig[f] = contigs.size();
        contigs.push_back(std::move(c));
    }
    for (auto& rc : copies) {
        int64_t U = rc.e - rc.s;
        // two-copy families sometimes get a confident mapq (they survive filter 3 and 4 at load,
        // and are then dropped by the uniqueness test in chaining)
        bool confident = fam_copies[rc.fam] == 2 && rng.uni() < 0.4;
        places.push_back({0, rc.s, rc.e, fam_contig[rc.fam], rc.rev, rc.rev ? U - 1 : 0,
                          (uint8_t)(confident ? 60 : 0), (uint8_t)(confident ? 60 : 40)});
    }
    {
        size_t ci = 0;
        int64_t pos = 0;
        while (pos < o.glen) {
            int64_t lim = ci < copies.size() ? copies[ci].s : o.glen;
            // tile [pos, lim) with unique contigs
            int64_t p = pos;
            while (p < lim) {
                int64_t L = (int64_t)rng.lognormal(o.contig_median, 0.8);
                if (rng.uni() < 0.08) L = 60 + (int64_t)rng.below(640);   // short contigs: <250 never mapped, <500 fail aln-block
                L = std::max<int64_t>(60, L);
                if (p + L > lim) L = lim - p;
                if (L >= 60) add_unique_contig(0, p, p + L);
                int64_t gap = (int64_t)rng.lognormal(o.gap_median, 0.7);
                if (rng.uni() < 0.06 && L > 400) gap = -(5 + (int64_t)rng.below(36));
                p += L + gap;
            }
            if (ci < copies.size()) { pos = copies[ci].e + 30 + (int64_t)rng.below(300); ci++; }
            else break;
        }
    }

    // ---------------------------------------------------------------- variant molecules
    std::vector<Molecule> mols;
    mols.push_back({{{0, 0, o.glen}}, o.cov});
    if (o.variants) {
        // unique placements sorted along the genome, long enough to anchor
        std::vector<size_t> uniq;
        for (size_t i = 0; i < places.size(); i++)
            if (places[i].axis == 0 && places[i].mapq_lo == 60 && places[i].e - places[i].s >= 1500 &&
                contigs[places[i].contig].km < 34.0)
                uniq.push_back(i);
        std::sort(uniq.begin(), uniq.end(), [&](size_t a, size_t b) { return places[a].s < places[b].s; });
        int nvar = std::max<int>(1, (int)std::llround(o.variant_per_mb * o.glen / 1e6));
        auto novel = [&](int64_t L) -> uint32_t {
            std::string s(L, 'A');
            for (auto& ch : s) ch = kBases[rng.next() & 3];
            axes.push_back(s);
            uint32_t axis = axes.size() - 1;
            add_unique_contig(axis, 0, L);
            return axis;
        };
        const int64_t flank = 35000;
        for (int kind = 0; kind < 4; kind++) {
            for (int v = 0; v < nvar && uniq.size() > 8; v++) {
                size_t k = 2 + rng.below(uniq.size() - 4);
                const Placement A = places[uniq[k - 1]], X = places[uniq[k]], B = places[uniq[k + 1]];
                int64_t left_s = std::max<int64_t>(0, A.s - flank), right_e = std::min<int64_t>(o.glen, B.e + flank);
                double vc = 6.0 + 8.0 * rng.uni();
                Molecule m;
                if (kind == 0) {            // deletion haplotype: A -> B directly (triangle with A->X->B)
                    m.pieces = {{0, left_s, A.e + 40}, {0, B.s - 40, right_e}};
                    (void)X;
                } else if (kind == 1) {     // substitution haplotype: A -> Y -> B (simple bubble vs A->X->B)
                    uint32_t ax = novel(1500 + (int64_t)rng.below(4000));
                    m.pieces = {{0, left_s, A.e + 40}, {ax, 0, (int64_t)axes[ax].size()}, {0, B.s - 40, right_e}};
                } else if (kind == 2) {     // dead end: A -> T (tip)
                    uint32_t ax = novel(2500 + (int64_t)rng.below(3000));
                    m.pieces = {{0, left_s, A.e + 40}, {ax, 0, (int64_t)axes[ax].size()}};
                } else {                    // low coverage branch: A -> W -> far away contig (weak edges)
                    uint32_t ax = novel(1200 + (int64_t)rng.below(2000));
                    size_t k2 = 1 + rng.below(uniq.size() - 2);
                    const Placement C = places[uniq[k2]];
                    m.pieces = {{0, left_s, A.e + 40}, {ax, 0, (int64_t)axes[ax].size()},
                                {0, C.s, std::min<int64_t>(o.glen, C.e + flank)}};
                    vc = 1.5;
                }
                bool sane = true;
                for (auto& pc : m.pieces) if (pc.e <= pc.s) sane = false;
                if (!sane) continue;
                m.cov = vc;
                mols.push_back(m);
            }
        }
    }

    // ---------------------------------------------------------------- contig id shuffle
    std::vector<uint32_t> new_id(contigs.size());
    for (size_t i = 0; i < new_id.size(); i++) new_id[i] = i;
    for (size_t i = new_id.size(); i > 1; i--) std::swap(new_id[i - 1], new_id[rng.below(i)]);
    {
        std::vector<uint32_t> order(contigs.size());
        for (size_t i = 0; i < contigs.size(); i++) order[new_id[i]] = i;
        sink.contigs(contigs, order);
        sink.genome(axes[0]);
    }
    // placements per axis sorted by start
    std::vector<std::vector<Placement>> axis_places(axes.size());
    for (auto& p : places) axis_places[p.axis].push_back(p);
    for (auto& v : axis_places) std::sort(v.begin(), v.end(), [](const Placement& a, const Placement& b) { return a.s < b.s; });

    // ---------------------------------------------------------------- reads + PAF
    uint64_t read_id = 0, total_bases = 0, total_hits = 0;
    std::string tmpl, rseq;
    std::vector<uint32_t> t_axis_piece;       // piece index of each template base
    std::vector<char> col_op;                 // per column: 'M' match, 'X' mismatch, 'I', 'D'
    std::vector<uint32_t> col_q;              // read bases consumed before this column
    std::vector<uint32_t> tcol;               // column index of each template base
    std::vector<Hit> hits;

    for (size_t mi = 0; mi < mols.size(); mi++) {
        const Molecule& mol = mols[mi];
        std::vector<int64_t> pstart(mol.pieces.size() + 1, 0);
        for (size_t i = 0; i < mol.pieces.size(); i++) pstart[i + 1] = pstart[i] + (mol.pieces[i].e - mol.pieces[i].s);
        int64_t mlen = pstart.back();
        double target = mol.cov * mlen;
        double acc = 0;
        while (acc < target) {
            int64_t L = (int64_t)rng.lognormal(o.read_median, o.read_sigma);
            L = std::max<int64_t>(1000, std::min<int64_t>(L, o.read_cap));
            if (L > mlen) L = mlen;
            int64_t a = (int64_t)rng.below(mlen - L + 1), b = a + L;
            acc += L;
            // template
            tmpl.clear(); t_axis_piece.clear();
            for (size_t pi = 0; pi < mol.pieces.size(); pi++) {
                int64_t s = std::max(a, pstart[pi]), e = std::min(b, pstart[pi + 1]);
                if (s >= e) continue;
                const Piece& pc = mol.pieces[pi];
                tmpl.append(axes[pc.axis], pc.s + (s - pstart[pi]), e - s);
                t_axis_piece.insert(t_axis_piece.end(), e - s, (uint32_t)pi);
            }
            // errors
            rseq.clear(); col_op.clear(); col_q.clear(); tcol.assign(tmpl.size(), 0);
            for (size_t t = 0; t <= tmpl.size(); t++) {
                while (em.ins > 0 && rng.uni() < em.ins) {
                    col_op.push_back('I'); col_q.push_back(rseq.size());
                    rseq.push_back(kBases[rng.next() & 3]);
                }
                if (t == tmpl.size()) break;
                double r = rng.uni();
                tcol[t] = col_op.size();
                col_q.push_back(rseq.size());
                if (r < em.del) { col_op.push_back('D'); }
                else if (r < em.del + em.sub) {
                    char c;
                    do { c = kBases[rng.next() & 3]; } while (c == tmpl[t]);
                    col_op.push_back('X'); rseq.push_back(c);
                } else { col_op.push_back('M'); rseq.push_back(tmpl[t]); }
            }
            col_q.push_back(rseq.size());
            uint32_t rlen = rseq.size();
            if (rlen < 500) continue;
            bool read_rev = rng.next() & 1;
            // hits
            hits.clear();
            for (size_t pi = 0; pi < mol.pieces.size(); pi++) {
                int64_t s = std::max(a, pstart[pi]), e = std::min(b, pstart[pi + 1]);
                if (s >= e) continue;
                const Piece& pc = mol.pieces[pi];
                int64_t as = pc.s + (s - pstart[pi]), ae = pc.s + (e - pstart[pi]);   // axis interval covered
                const auto& pl = axis_places[pc.axis];
                for (const Placement& P : pl) {
                    if (P.e <= as) continue;
                    if (P.s >= ae) break;
                    int64_t os = std::max(as, P.s), oe = std::min(ae, P.e);
                    if (oe - os < 100) continue;
                    const Contig& C = contigs[P.contig];
                    if (C.seq.size() < 250) continue;          // haslr.py maps only contigs >= 250 bp
                    // template indices of [os, oe)
                    int64_t t0 = (s - a) + (os - as), t1 = t0 + (oe - os);
                    // shrink to match columns at both ends
                    while (t0 < t1 && col_op[tcol[t0]] != 'M') t0++;
                    while (t1 > t0 && col_op[tcol[t1 - 1]] != 'M') t1--;
                    if (t1 - t0 < 50) continue;
                    uint32_t c0 = tcol[t0], c1 = tcol[t1 - 1] + 1;
                    Hit h;
                    uint32_t qs = col_q[c0], qe = col_q[c1];
                    h.nb = c1 - c0; h.nm = 0;
                    std::string ops; ops.reserve(h.nb);
                    for (uint32_t c = c0; c < c1; c++) {
                        char op = col_op[c];
                        if (op == 'M') h.nm++;
                        ops.push_back(op == 'X' ? 'M' : op);
                    }
                    int64_t g0 = os + (t0 - ((s - a) + (os - as))), g1 = g0 + (t1 - t0);   // axis coords of the aligned block
                    if (!P.rev) { h.ts = (uint32_t)(P.coff + (g0 - P.s)); h.te = (uint32_t)(P.coff + (g1 - P.s)); }
                    else { h.ts = (uint32_t)(P.coff - (g1 - 1 - P.s)); h.te = (uint32_t)(P.coff - (g0 - P.s) + 1); std::reverse(ops.begin(), ops.end()); }
                    h.rev = P.rev; h.qs = qs; h.qe = qe;      // in the orientation the read was synthesised in; flipped below with the read
                    h.tid = new_id[P.contig]; h.tlen = C.seq.size();
                    h.mapq = P.mapq_lo + (uint8_t)rng.below(P.mapq_hi - P.mapq_lo + 1);
                    // run-length CIGAR
                    h.cg.clear();
                    for (size_t i2 = 0; i2 < ops.size();) {
                        size_t j2 = i2;
                        while (j2 < ops.size() && ops[j2] == ops[i2]) j2++;
                        h.cg += std::to_string(j2 - i2); h.cg.push_back(ops[i2]);
                        i2 = j2;
                    }
                    hits.push_back(std::move(h));
                }
            }
            if (o.hairpin_frac > 0 && rng.uni() < o.hairpin_frac) {
                // the polymerase went round the hairpin adapter: the read continues with the reverse complement of its last X bases, and
                // every hit inside that stretch appears a second time, mirrored, on the other strand (same target interval, same CIGAR)
                const uint32_t X = (uint32_t)(rlen * (0.3 + 0.6 * rng.uni()));
                const size_t n0 = hits.size();
                for (size_t k = 0; k < n0; k++) {
                    if (hits[k].qs < rlen - X) continue;
                    Hit m = hits[k];
                    m.qs = 2 * rlen - hits[k].qe; m.qe = 2 * rlen - hits[k].qs; m.rev = !m.rev;
                    hits.push_back(std::move(m));
                }
                rseq += revcomp(rseq.substr(rlen - X));
                rlen = rseq.size();
            }
            if (read_rev) for (Hit& h : hits) { const uint32_t qs = h.qs; h.qs = rlen - h.qe; h.qe = rlen - qs; h.rev = !h.rev; }
            std::sort(hits.begin(), hits.end(), [](const Hit& x, const Hit& y) { return x.qs != y.qs ? x.qs < y.qs : x.qe < y.qe; });
            std::string out = read_rev ? revcomp(rseq) : rseq;
            sink.read(out, rlen, hits);
            total_hits += hits.size();
            total_bases += rlen;
            read_id++;
        }
    }
    st.contigs = contigs.size(); st.families = fam_unit.size(); st.molecules = mols.size(); st.reads = read_id; st.bases = total_bases; st.hits = total_hits;
    return 0;
}
}  // namespace
