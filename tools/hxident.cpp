// hxident — length-weighted identity of assembled contigs against a truth genome (test/bench tool).
// For every query record: locate it on the genome (either strand) by exact 24-mer seeds, then a banded
// global edit distance of the query against the implied genome window. Prints one line per record and a
// summary `identity <weighted> aligned_bases <n> records <k> unplaced <u>`.
#include <algorithm>
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

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: hxident genome.fa asm.fa [band]\n"); return 2; }
    auto g = read_fa(argv[1]);
    auto qs = read_fa(argv[2]);
    long B = argc > 3 ? atol(argv[3]) : 400;
    if (g.empty()) return 2;
    const std::string& G = g[0].second;
    const int K = 24;
    std::unordered_map<std::string, long> idx;
    for (long i = 0; i + K <= (long)G.size(); i += 7) idx.emplace(G.substr(i, K), i);
    double wsum = 0; long wlen = 0; int unplaced = 0;
    const long WIN = 20000;
    for (auto& rec : qs) {
        // strand: the one whose first placeable window hits
        std::string r = rc(rec.second);
        long total_ed = 0, total_len = 0, bad_win = 0;
        int strand = -1;
        for (int s2 = 0; s2 < 2 && strand < 0; s2++) {
            const std::string& q = s2 == 0 ? rec.second : r;
            for (long i = 0; i + K <= (long)q.size() && i < 50000; i++)
                if (idx.count(q.substr(i, K))) { strand = s2; break; }
        }
        if (strand < 0) { unplaced++; printf("%s\tlen=%zu\tUNPLACED\n", rec.first.c_str(), rec.second.size()); continue; }
        const std::string& q = strand == 0 ? rec.second : r;
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
            if (off < -1000000 || wl < K) { bad_win++; total_ed += wl; total_len += wl; continue; }
            long st = std::max(0L, off - 200);
            std::string t = G.substr(st, std::min<long>((long)G.size() - st, wl + 400 + B / 2));
            long ed = banded_ed(win, t, B);
            total_ed += std::min(ed, wl); total_len += wl;
        }
        double ident = 1.0 - (double)total_ed / (double)total_len;
        printf("%s\tlen=%zu\t%c\ted=%ld\tunplaced_windows=%ld\tidentity=%.6f\n", rec.first.c_str(), rec.second.size(), strand ? '-' : '+', total_ed, bad_win, ident);
        wsum += ident * total_len; wlen += total_len;
    }
    printf("identity %.6f aligned_bases %ld records %zu unplaced %d\n", wlen ? wsum / wlen : 0.0, wlen, qs.size(), unplaced);
    return 0;
}
