// hxident — length-weighted identity of assembled contigs against a truth genome (test/bench tool).
// For every query record: choose the strand by a vote of exact 24-mer seeds of both orientations, place 20 kb windows by seed clusters, then a banded
// global edit distance of the query against the implied genome window. Prints one line per record and a
// summary `identity <weighted> aligned_bases <n> records <k> unplaced <u>`.
#include <algorithm>
#include <atomic>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <cstdint>
#include <climits>
#include <vector>

static std::vector<std::pair<std::string, std::string>> read_fa(const char* p) {
    std::vector<std::pair<std::string, std::string>> v;
    std::ifstream f(p);
    std::string l;
    while (std::getline(f, l)) {
        if (l.empty()) continue;
        if (l[0] == '>') v.push_back({l.substr(1), ""});
        else if (!v.empty()) v.back().second += l;
    }
    return v;
}
static std::string rc(const std::string& s) {
    std::string r(s.size(), 'N');
    for (size_t i = 0; i < s.size(); i++) { char c = s[s.size() - 1 - i]; r[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }
    return r;
}
// banded edit distance, q vs t (global in q, t window chosen by caller), band half-width B
static long banded_ed(const std::string& q, const std::string& t, long B) {
    const long n = q.size(), m = t.size(), INF = 1L << 40;
    std::vector<long> prev(2 * B + 1, INF), cur(2 * B + 1, INF);
    // cell (i, j) with j in [i - B, i + B] stored at j - i + B
    for (long k = 0; k <= 2 * B; k++) { long j = k - B; if (j >= 0 && j <= m) prev[k] = 0; }   // free start in the target window (semi-global)
    for (long i = 1; i <= n; i++) {
        for (long k = 0; k <= 2 * B; k++) {
            long j = i + k - B;
            long best = INF;
            if (j >= 0 && j <= m) {
                if (j == 0) best = i;   // query prefix unaligned
                else {
                    if (prev[k] < INF) best = std::min(best, prev[k] + (q[i - 1] != t[j - 1]));          // diag: (i-1, j-1) -> same k
                    if (k + 1 <= 2 * B && prev[k + 1] < INF) best = std::min(best, prev[k + 1] + 1);     // up: (i-1, j)
                    if (k - 1 >= 0 && cur[k - 1] < INF) best = std::min(best, cur[k - 1] + 1);           // left: (i, j-1)
                }
            }
            cur[k] = best;
        }
        std::swap(prev, cur);
    }
    long best = INF;   // semi-global: the target window may extend beyond the query's end for free
    for (long k = 0; k <= 2 * B; k++) { long j = n + k - B; if (j >= 0 && j <= m) best = std::min(best, prev[k]); }
    return best;
}

// Seed index of the genome: every 7th 24-mer (ACGT only) packed into 48 bits, with its position, sorted by (key, position) - a lookup returns the FIRST position
// of the key, what the hash map of the first version (emplace in ascending order) kept. Round 6: that map took minutes on a 400 Mb genome (57 M std::string
// keys, built on one thread); this builds on every thread (buckets by the key's top byte, each sorted on its own) and holds 16 bytes per seed.
struct SeedIndex {
    std::vector<std::pair<uint64_t, int64_t>> v;
    static bool pack(const char* p, int K, uint64_t& key) {
        key = 0;
        for (int i = 0; i < K; i++) {
            const char c = p[i];
            const uint64_t b = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
            if (b > 3) return false;
            key = key << 2 | b;
        }
        return true;
    }
    void build(const std::string& G, int K, long step, unsigned T) {
        const long n = (long)G.size() >= K ? ((long)G.size() - K) / step + 1 : 0;
        std::vector<std::vector<std::pair<uint64_t, int64_t>>> part(T);
        std::vector<std::thread> th;
        auto scan = [&](unsigned t) {
            auto& o = part[t];
            o.reserve((size_t)(n / T + 1));
            for (long q = n * t / T; q < n * (t + 1) / T; q++) { uint64_t k; if (pack(G.data() + q * step, K, k)) o.push_back({k, q * step}); }
        };
        for (unsigned t = 1; t < T; t++) th.emplace_back(scan, t);
        scan(0);
        for (auto& x : th) x.join();
        th.clear();
        const int shift = 2 * K - 8;
        std::vector<size_t> cnt(257, 0);
        for (auto& o : part) for (auto& e : o) cnt[(e.first >> shift) + 1]++;
        for (int b = 0; b < 256; b++) cnt[b + 1] += cnt[b];
        v.resize(cnt[256]);
        {
            std::vector<size_t> at(cnt.begin(), cnt.end() - 1);
            for (auto& o : part) { for (auto& e : o) v[at[e.first >> shift]++] = e; std::vector<std::pair<uint64_t, int64_t>>().swap(o); }   // (parts in position order: equal keys stay in ascending position)
        }
        std::atomic<int> next{0};
        auto sorter = [&]() { for (;;) { const int b = next.fetch_add(1); if (b >= 256) return; std::sort(v.begin() + (long)cnt[b], v.begin() + (long)cnt[b + 1]); } };
        for (unsigned t = 1; t < T; t++) th.emplace_back(sorter);
        sorter();
        for (auto& x : th) x.join();
    }
    long find(const char* p, int K) const {   // first position of the 24-mer at p, -1 if it is not a seed
        uint64_t k;
        if (!pack(p, K, k)) return -1;
        auto it = std::lower_bound(v.begin(), v.end(), std::make_pair(k, (int64_t)INT64_MIN));
        return it != v.end() && it->first == k ? (long)it->second : -1;
    }
};

// identity of one record on one strand; hits = windows that found a placement
struct StrandResult { long ed = 0, len = 0, bad = 0; };
static StrandResult eval_strand(const std::string& q, const std::string& G, const SeedIndex& idx, int K, long B) {
    StrandResult r;
    const long WIN = 20000;
    // windows are placed independently (seeded inside the window), so cumulative indel drift never leaves the band
    for (long w0 = 0; w0 < (long)q.size(); w0 += WIN) {
        long wl = std::min<long>(WIN, (long)q.size() - w0);
        std::string win = q.substr(w0, wl);
        // offset by majority vote of seeds spread over the window (a single seed may sit in a repeat copy or straddle an error)
        std::vector<long> offs;
        for (long i = 0; i + K <= wl; i += 31) {
            const long at = idx.find(win.data() + i, K);
            if (at >= 0) offs.push_back(at - i);
        }
        long off = -1000000000L;
        if (!offs.empty()) {
            std::sort(offs.begin(), offs.end());
            size_t bi = 0, bc = 0;
            for (size_t a2 = 0, b2 = 0; a2 < offs.size(); a2++) {       // densest cluster within +-150
                while (b2 < offs.size() && offs[b2] - offs[a2] <= 300) b2++;
                if (b2 - a2 > bc) { bc = b2 - a2; bi = a2; }
            }
            off = offs[bi];   // smallest offset of the cluster = alignment of the window start (insertions only push later seeds right)
        }
        if (off < -1000000 || wl < K) { r.bad++; r.ed += wl; r.len += wl; continue; }
        long st = std::max(0L, off - 200);
        std::string t = G.substr(st, std::min<long>((long)G.size() - st, wl + 400 + B / 2));
        long ed = banded_ed(win, t, B);
        r.ed += std::min(ed, wl); r.len += wl;
    }
    return r;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: hxident genome.fa asm.fa [band [threads]]\n"); return 2; }
    auto g = read_fa(argv[1]);
    auto qs = read_fa(argv[2]);
    long B = argc > 3 ? atol(argv[3]) : 400;
    unsigned T = argc > 4 ? (unsigned)atoi(argv[4]) : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    if (g.empty()) return 2;
    // a genome of several records (hxsim --chromosomes) is searched as one sequence, the records a thousand N apart (no seed, no window spans two)
    std::string joined;
    if (g.size() > 1) {
        size_t tot = 0;
        for (auto& r : g) tot += r.second.size() + 1000;
        joined.reserve(tot);
        for (auto& r : g) { joined += r.second; joined.append(1000, 'N'); std::string().swap(r.second); }
    }
    const std::string& G = g.size() > 1 ? joined : g[0].second;
    const int K = 24;
    SeedIndex idx;
    idx.build(G, K, 7, T);
    // records are independent: dealt to threads, reported in file order
    struct Out { std::string line; double ident = 0; long len = 0; bool placed = false; };
    std::vector<Out> out(qs.size());
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= qs.size()) return;
            const auto& rec = qs[k];
            // strand: seeds of both orientations vote (a single seed may hit an inverted repeat); without any hit the record is unplaced
            const std::string r = rc(rec.second);
            long votes[2] = {0, 0};
            for (int s2 = 0; s2 < 2; s2++) {
                const std::string& q = s2 == 0 ? rec.second : r;
                const long lim = std::min<long>((long)q.size(), 200000);
                for (long i = 0; i + K <= lim; i += 13) votes[s2] += idx.find(q.data() + i, K) >= 0;
            }
            char buf[256];
            if (!votes[0] && !votes[1]) { snprintf(buf, sizeof buf, "%s\tlen=%zu\tUNPLACED\n", rec.first.c_str(), rec.second.size()); out[k].line = buf; continue; }
            int strand = votes[1] > votes[0] ? 1 : 0;
            StrandResult sr = eval_strand(strand ? r : rec.second, G, idx, K, B);
            if (sr.len && (double)sr.ed / (double)sr.len > 0.5) {   // a vote decided by repeats: the other strand may be the real one
                StrandResult o = eval_strand(strand ? rec.second : r, G, idx, K, B);
                if (o.ed < sr.ed) { sr = o; strand ^= 1; }
            }
            const double ident = 1.0 - (double)sr.ed / (double)sr.len;
            snprintf(buf, sizeof buf, "%s\tlen=%zu\t%c\ted=%ld\tunplaced_windows=%ld\tidentity=%.6f\n", rec.first.c_str(), rec.second.size(), strand ? '-' : '+', sr.ed, sr.bad, ident);
            out[k].line = buf; out[k].ident = ident; out[k].len = sr.len; out[k].placed = true;
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; t++) th.emplace_back(worker);
    worker();
    for (auto& x : th) x.join();
    double wsum = 0; long wlen = 0; int unplaced = 0;
    for (const Out& o : out) {
        fputs(o.line.c_str(), stdout);
        if (!o.placed) { unplaced++; continue; }
        wsum += o.ident * o.len; wlen += o.len;
    }
    printf("identity %.6f aligned_bases %ld records %zu unplaced %d\n", wlen ? wsum / wlen : 0.0, wlen, qs.size(), unplaced);
    return 0;
}
