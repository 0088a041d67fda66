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
#include <unordered_map>
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

// identity of one record on one strand; hits = windows that found a placement
struct StrandResult { long ed = 0, len = 0, bad = 0; };
static StrandResult eval_strand(const std::string& q, const std::string& G, const std::unordered_map<std::string, long>& idx, int K, long B) {
    StrandResult r;
    const long WIN = 20000;
    // windows are placed independently (seeded inside the window), so cumulative indel drift never leaves the band
    for (long w0 = 0; w0 < (long)q.size(); w0 += WIN) {
        long wl = std::min<long>(WIN, (long)q.size() - w0);
        std::string win = q.substr(w0, wl);
        // offset by majority vote of seeds spread over the window (a single seed may sit in a repeat copy or straddle an error)
        std::vector<long> offs;
        for (long i = 0; i + K <= wl; i += 31) {
            auto it = idx.find(win.substr(i, K));
            if (it != idx.end()) offs.push_back(it->second - i);
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
    const std::string& G = g[0].second;
    const int K = 24;
    std::unordered_map<std::string, long> idx;
    for (long i = 0; i + K <= (long)G.size(); i += 7) idx.emplace(G.substr(i, K), i);
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
                for (long i = 0; i + K <= lim; i += 13) votes[s2] += idx.count(q.substr(i, K));
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
