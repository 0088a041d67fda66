// graph.cpp — backbone graph, its writers, and the four serial cleaning passes.
//
// These are order-dependent edits of a small graph and stay on the host.
// Output bytes (GFA, .stat, cleaning logs) are a parity gate and are checked against files produced
// by the compiled reference (tests/golden/, oracle/_ref/ref_front). Reference functions followed
// (paths under /root/reference/src/haslr_assemble/src/):
//   bbg_remove_weak_edges Backbone_graph.cpp:348-375     bbg_print_graph_gfa :540-588
//   bbg_general_stats :595-659                            bbg_report_branching_nodes :682-694
//   bbg_find_simple_path_from_source :378-402             clean_tips Cleaning.cpp:59-96
//   clean_simple_bubbles_old Cleaning.cpp:98-184          detect_super_bubble/clean_super_bubbles :488-648
//   clean_small_bubbles Cleaning.cpp:7-57
#include <cstring>
#include <cstdlib>
#include <thread>
#include <algorithm>
#include <cstdio>
#include <map>
#include <queue>
#include <set>
#include <tuple>
#include <unordered_map>

#include "host_internal.h"

namespace hxh {

#define LOGF(fp, ...) do { if (fp) fprintf(fp, __VA_ARGS__); } while (0)

FILE* open_or_null(const std::string& path, const char* mode) {
    if (path.empty()) return nullptr;
    FILE* fp = fopen(path.c_str(), mode);
    if (!fp) { fprintf(stderr, "[ERROR] (Common::file_open_write) could not open file: %s\n", path.c_str()); exit(EXIT_FAILURE); }
    return fp;
}

std::string revcomp(const std::string& s) {
    // complement by table (anything but ACGT / acgt becomes N), written back to front
    static const struct Tab { char c[256]; Tab() { memset(c, 'N', 256); c['A'] = c['a'] = 'T'; c['C'] = c['c'] = 'G'; c['G'] = c['g'] = 'C'; c['T'] = c['t'] = 'A'; } } tab;
    const size_t n = s.size();
    std::string r(n, 'N');
    const unsigned char* in = reinterpret_cast<const unsigned char*>(s.data());
    char* out = &r[0];
    for (size_t i = 0; i < n; i++) out[n - 1 - i] = tab.c[in[i]];
    return r;
}

Arc* Graph::find(uint32_t v, uint32_t key) {
    auto& a = adj[v];
    auto it = std::lower_bound(a.begin(), a.end(), key, [](const Arc& x, uint32_t k) { return x.key < k; });
    return it != a.end() && it->key == key ? &*it : nullptr;
}

void Graph::erase_arc(uint32_t v, uint32_t key) {
    auto& a = adj[v];
    auto it = std::lower_bound(a.begin(), a.end(), key, [](const Arc& x, uint32_t k) { return x.key < k; });
    if (it != a.end() && it->key == key) a.erase(it);
}

// Edge keys arrive sorted by (source vertex, map key), so every arc vector is born sorted.
void graph_build(Graph& g, uint32_t n_nodes, const hx_edges_out& e) {
    g.n_nodes = n_nodes;
    g.adj.assign((size_t)n_nodes * 2, {});
    for (uint64_t i = 0; i < e.n_edge; i++) {
        uint32_t v = (uint32_t)(e.edge_key[i] >> 32), key = (uint32_t)e.edge_key[i];
        Arc a;
        a.key = key; a.supp = (uint32_t)(e.edge_off[i + 1] - e.edge_off[i]); a.dev_edge = (uint32_t)i;
        g.adj[v].push_back(a);
    }
}

int graph_remove_weak_edges(Graph& g, uint32_t min_edge_sup) {
    int removed = 0;
    for (uint32_t v = 0; v < g.adj.size(); v++) {
        for (size_t k = 0; k < g.adj[v].size();) {
            if (g.adj[v][k].supp < min_edge_sup) {
                uint32_t key = g.adj[v][k].key;
                g.adj[v].erase(g.adj[v].begin() + k);
                g.erase_arc(Graph::twin_vertex(key), Graph::twin_key(v));
                removed++;
            } else k++;
        }
    }
    return removed;
}

// the arcs in the order the GFA lists them (vertex ascending, key ascending): all a writer needs, so that the file can be written
// while the cleaning passes go on changing the graph
std::vector<std::pair<uint32_t, uint32_t>> graph_arc_list(const Graph& g) {
    std::vector<std::pair<uint32_t, uint32_t>> arcs;
    for (uint32_t v = 0; v < g.adj.size(); v++)
        for (const Arc& a : g.adj[v]) arcs.push_back({v, a.key});
    return arcs;
}

void graph_write_gfa_arcs(const std::vector<std::pair<uint32_t, uint32_t>>& arcs, const Dataset& d, const std::string& path) {
    FILE* fp = open_or_null(path, "w");
    if (!fp) return;
    std::vector<char> iobuf(1 << 22);
    setvbuf(fp, iobuf.data(), _IOFBF, iobuf.size());
    std::set<uint32_t> to_print;
    for (const auto& a : arcs) { to_print.insert(a.first >> 1); to_print.insert(a.second >> 1); }
    for (uint32_t id : to_print) {
        std::string s = d.contig_seq(id);
        fprintf(fp, "S\t%u\t", id);
        fwrite(s.data(), 1, s.size(), fp);
        fprintf(fp, "\tLN:i:%zu\tKC:i:%u\n", s.size(), d.contig_kc[id]);
    }
    for (const auto& a : arcs)
        fprintf(fp, "L\t%u\t%c\t%u\t%c\t0M\n", a.first >> 1, "+-"[a.first & 1], a.second >> 1, (a.second & 1) ? '-' : '+');
    fclose(fp);
}

void graph_write_gfa(const Graph& g, const Dataset& d, const std::string& path) {
    if (path.empty()) return;
    graph_write_gfa_arcs(graph_arc_list(g), d, path);
}

void graph_write_stats(const Graph& g, const Dataset& d, const std::string& path) {
    FILE* fp = open_or_null(path, "w");
    if (!fp) return;
    uint32_t num = g.n_nodes, nb_node = 0, nb_edge = 0;
    for (uint32_t i = 0; i < num; i++) {
        nb_node += (g.deg(i, 0) > 0 || g.deg(i, 1) > 0);
        nb_edge += g.deg(i, 0) + g.deg(i, 1);
    }
    fprintf(fp, "nodes: %d\n", nb_node);
    fprintf(fp, "edges: %d\n", nb_edge / 2);
    std::vector<bool> visited(num, false);
    // (size, nodes, representative), narrowed to 32 bits exactly like the reference's tuple (:612,:644)
    std::vector<std::tuple<uint32_t, uint32_t, uint32_t>> comps;
    for (uint32_t i = 0; i < num; i++) {
        if (visited[i] || !(g.deg(i, 0) > 0 || g.deg(i, 1) > 0)) continue;
        uint64_t cc_size = d.contig_len[i], cc_node = 1;
        std::queue<uint32_t> q;
        q.push(i);
        visited[i] = true;
        while (!q.empty()) {
            uint32_t cur = q.front();
            q.pop();
            for (int side = 0; side < 2; side++)
                for (const Arc& a : g.adj[(cur << 1) | side]) {
                    uint32_t nx = a.key >> 1;
                    if (!visited[nx]) { q.push(nx); cc_node++; cc_size += d.contig_len[nx]; visited[nx] = true; }
                }
        }
        comps.push_back(std::make_tuple((uint32_t)cc_size, (uint32_t)cc_node, i));
    }
    // comparator on size only, std::sort: equal sizes land where libstdc++'s introsort puts them, as in the reference (:651)
    std::sort(comps.begin(), comps.end(), [](const std::tuple<uint32_t, uint32_t, uint32_t>& a, const std::tuple<uint32_t, uint32_t, uint32_t>& b) {
        return std::get<0>(a) > std::get<0>(b);
    });
    fprintf(fp, "connected_components: %zu\n", comps.size());
    for (uint32_t i = 0; i < comps.size(); i++)
        fprintf(fp, "\tcomponent:%u\tsize:%u\tnodes:%u\trepresentative:%u\n", i, std::get<0>(comps[i]), std::get<1>(comps[i]), std::get<2>(comps[i]));
    fclose(fp);
}

void graph_report_branching(const Graph& g, const std::string& path) {
    FILE* fp = open_or_null(path, "w");
    if (!fp) return;
    for (uint32_t i = 0; i < g.n_nodes; i++)
        if (g.deg(i, 0) >= 2 || g.deg(i, 1) >= 2)   // labels are swapped in the reference too (:691)
            fprintf(fp, "node:%u\tincoming:%zu\toutgoing:%zu\n", i, g.deg(i, 0), g.deg(i, 1));
    fclose(fp);
}

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// Parallel formulation of the four cleaning passes (Cleaning.cpp:7-184, :488-648). Each pass of the reference walks the
// nodes in ascending order and, at almost every node, does nothing: a cheap test of the node's own adjacency sends it on. Edits only ever
// REMOVE edges. So a pass is split into
//   (1) the test for every node on the graph as the pass finds it - read-only, independent per node: all host threads, a bit per node;
//   (2) the reference's loop over the nodes whose bit is set, in ascending order, deciding on the CURRENT graph exactly like the reference;
//       every removal sets the bits of the two nodes it touches, so a node whose test only became true through an earlier removal of this
//       pass (a degree that fell to the tested value) is visited as well, at its place in the order.
// A node the reference would have acted on is never skipped - its test was true at the start or became true by a removal next to it - and a
// node visited without need takes the reference's own early exit: the edits, their order and the logs are the serial algorithm's, whatever
// the thread count. (The small-bubble test - "has a triangle" - can only turn false by removals, so its set needs no additions.)
// ---------------------------------------------------------------------------------------------------------------------------
unsigned clean_threads_(uint32_t n_nodes) {
    if (const char* e = getenv("HASLR_CLEAN_THREADS")) return (unsigned)std::max(1, atoi(e));
    if (n_nodes < 200000) return 1;                   // (below that the scan is microseconds: threads would cost more than they save)
    return std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
}

template <class Pred>
std::vector<uint64_t> candidate_bits(const Graph& g, Pred pred) {
    const uint32_t n = g.n_nodes, words = (n + 63) / 64;
    std::vector<uint64_t> bits(words + 1, 0);
    const unsigned T = clean_threads_(n);
    auto work = [&](uint32_t w0, uint32_t w1) {
        for (uint32_t w = w0; w < w1; w++) {
            uint64_t b = 0;
            for (uint32_t k = 0; k < 64 && w * 64 + k < n; k++) if (pred(w * 64 + k)) b |= 1ull << k;
            bits[w] = b;
        }
    };
    if (T <= 1) { work(0, words); return bits; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++) th.emplace_back(work, (uint32_t)((uint64_t)words * t / T), (uint32_t)((uint64_t)words * (t + 1) / T));
    for (auto& x : th) x.join();
    return bits;
}

// the loop of a pass over the set bits, ascending; `body(i)` returns true when the reference would look at node i again (its `i--; continue`)
template <class Body>
void for_candidates(Graph& g, std::vector<uint64_t>& bits, Body body) {
    g.touched = &bits;
    for (uint32_t w = 0; w + 1 < bits.size(); w++)
        while (bits[w]) {                             // (a removal may set bits of this very word: lower ones are behind us and are cleared unseen)
            const uint32_t k = (uint32_t)__builtin_ctzll(bits[w]);
            const uint32_t i = w * 64 + k;
            bits[w] &= ~((2ull << k) - 1);            // this bit and everything below it
            while (body(i)) {}
            bits[w] &= ~((2ull << k) - 1);            // (the node's own removals set its bit again: it has just been looked at)
        }
    g.touched = nullptr;
}

struct PathElem { uint32_t strand, id; };

// bbg_find_simple_path_from_source: follow the arc at index `k` of vertex (src,side) while nodes are 1-in 1-out.
bool simple_path_from(const Graph& g, uint32_t src, uint32_t side, size_t k, int max_depth, std::vector<PathElem>& path, float& cov) {
    path.clear();
    cov = 0;
    path.push_back({side, src});
    const Arc* it = &g.adj[(src << 1) | side][k];
    uint32_t cn = it->key >> 1, cs = it->key & 1;
    int depth = 1;
    while (depth <= max_depth) {
        path.push_back({cs, cn});
        cov += it->supp;
        if (g.deg(cn, cs) == 0) break;
        if (g.deg(cn, cs) > 1 || g.deg(cn, 1 - cs) > 1) break;
        it = &g.adj[(cn << 1) | cs][0];
        cn = it->key >> 1; cs = it->key & 1;
        depth++;
    }
    if (depth > max_depth) return false;
    cov = cov / depth;
    return true;
}

void remove_path(Graph& g, const std::vector<PathElem>& p) {
    for (size_t j = 0; j + 1 < p.size(); j++) g.remove_edge(p[j].id, p[j].strand, p[j + 1].id, p[j + 1].strand);
}

}  // namespace
unsigned clean_threads(uint32_t n_nodes) { return clean_threads_(n_nodes); }

int clean_tips(Graph& g, int max_depth, const std::string& logpath) {
    FILE* fp = open_or_null(logpath, max_depth == 1 ? "w" : "a");
    int removed = 0;
    std::vector<uint64_t> cand = candidate_bits(g, [&](uint32_t i) { return g.deg(i, 0) + g.deg(i, 1) == 1; });   // a dead end: one arc on one side, none on the other
    for_candidates(g, cand, [&](uint32_t i) -> bool {
        uint32_t side;
        if (g.deg(i, 1) == 0 && g.deg(i, 0) == 1) side = 0;
        else if (g.deg(i, 1) == 1 && g.deg(i, 0) == 0) side = 1;
        else return false;
        std::vector<PathElem> p;
        float cov;
        if (simple_path_from(g, i, side, 0, max_depth, p, cov)) {
            if (g.deg(p.back().id, p.back().strand) == 0) return false;
            LOGF(fp, "tip_len:%zu\t%u:%c -> %u:%c\n", p.size() - 1, p.front().id, "+-"[p.front().strand], p.back().id, "+-"[p.back().strand]);
            remove_path(g, p);
            removed++;
        }
        return false;
    });
    if (fp) fclose(fp);
    return removed;
}

int clean_simple_bubbles(Graph& g, int max_depth, const std::string& logpath) {
    FILE* fp = open_or_null(logpath, "w");
    int removed = 0;
    std::vector<uint64_t> cand = candidate_bits(g, [&](uint32_t i) { return g.deg(i, 0) >= 2 || g.deg(i, 1) >= 2; });
    for_candidates(g, cand, [&](uint32_t i) -> bool {
        if (g.deg(i, 0) < 2 && g.deg(i, 1) < 2) return false;
        bool again = false;
        for (uint32_t side = 0; side < 2 && !again; side++) {
            if (g.deg(i, side) != 2) continue;
            std::vector<PathElem> p1, p2;
            float c1, c2;
            bool f1 = simple_path_from(g, i, side, 0, max_depth, p1, c1);
            bool f2 = simple_path_from(g, i, side, 1, max_depth, p2, c2);
            if (f1 && f2 && p1.back().id == p2.back().id && p1.back().strand == p2.back().strand) {
                LOGF(fp, "simple_bubble cov:%.2lf ", c1);
                for (auto& e : p1) LOGF(fp, "%u:%c ", e.id, "+-"[e.strand]);
                LOGF(fp, "\n              cov:%.2lf ", c2);
                for (auto& e : p2) LOGF(fp, "%u:%c ", e.id, "+-"[e.strand]);
                LOGF(fp, "\n");
                remove_path(g, c1 < c2 ? p1 : p2);
                removed++;
                again = true;   // the reference re-examines the same node (i--; continue)
            }
        }
        return again;
    });
    if (fp) fclose(fp);
    return removed;
}

namespace {

// Super bubbles (Cleaning.cpp:488-562 is miniasm's tour: a vertex is expanded once all of its in-arcs have been seen, and the
// bubble closes when exactly one vertex waits and nothing else is open). State is kept per vertex in flat epoch-stamped tables
// instead of hash maps, and the best path to a vertex as (parent, length, support): a vertex is final when it is expanded, so the
// chain of parents at that moment is the path the reference stores by value.
class BubbleTour {
    struct Slot { uint32_t epoch = 0, waiting_in = 0, parent = 0, hops = 0, support = 0; };
    std::vector<Slot> at_;
    uint32_t epoch_ = 0;

public:
    std::vector<uint32_t> best;                              // source ... sink, as vertices
    std::vector<std::pair<uint32_t, uint32_t>> touched;      // every arc looked at, (vertex, key), sorted, unique

    bool close_from(const Graph& g, uint32_t source) {
        if (at_.size() != g.adj.size()) at_.assign(g.adj.size(), Slot());
        epoch_++;
        best.clear(); touched.clear();
        std::vector<uint32_t> ready{source};   // used as a stack
        at_[source] = Slot{epoch_, 0, source, 1, 0};
        int open = 0;                          // vertices seen but not yet ready
        bool closed = false;
        uint32_t sink = 0;
        while (!ready.empty() && !closed) {
            const uint32_t v = ready.back();
            ready.pop_back();
            const Slot from = at_[v];
            for (const Arc& a : g.adj[v]) {
                const uint32_t w = a.key;
                touched.push_back({v, w});
                if ((w >> 1) == (v >> 1)) return false;      // an arc back into the same contig: not a bubble
                Slot& to = at_[w];
                const bool first = to.epoch != epoch_;
                if (first) { to = Slot{epoch_, (uint32_t)g.adj[w ^ 1u].size(), v, 0, 0}; open++; }
                // the reference compares mean support per hop, dividing the old value by (hops of v) - 1: at the source that is a division
                // by zero whose result (inf or NaN) never wins - IEEE doubles reproduce it
                const double via_v = double(from.support + a.supp) / double(from.hops);
                const double old = double(to.support) / double(from.hops - 1);
                if (first || via_v > old) { to.parent = v; to.hops = from.hops + 1; to.support = from.support + a.supp; }
                if (--to.waiting_in == 0 && !g.adj[w].empty()) { ready.push_back(w); open--; }
            }
            if (ready.size() == 1 && open == 0) { closed = true; sink = ready.back(); }
        }
        if (!closed) return false;
        for (uint32_t v = sink;; v = at_[v].parent) { best.push_back(v); if (v == source) break; }
        std::reverse(best.begin(), best.end());
        std::sort(touched.begin(), touched.end());
        touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
        return true;
    }
};

}  // namespace

int clean_super_bubbles(Graph& g, const std::string& logpath) {
    FILE* fp = open_or_null(logpath, "w");
    int removed = 0;
    BubbleTour tour;
    std::vector<uint64_t> cand = candidate_bits(g, [&](uint32_t i) { return g.deg(i, 0) >= 2 || g.deg(i, 1) >= 2; });
    for_candidates(g, cand, [&](uint32_t i) -> bool {
        bool again = false;
        for (uint32_t side = 0; side < 2 && !again; side++) {
            if (g.deg(i, side) < 2 || !tour.close_from(g, (i << 1) | side)) continue;
            const std::vector<uint32_t>& best = tour.best;
            LOGF(fp, "bubble_src %u:%c\tbubble_sink %u:%c\n", i, "+-"[side], best.back() >> 1, "+-"[best.back() & 1]);
            LOGF(fp, "\tbest_path ");
            for (uint32_t v : best) LOGF(fp, "%u:%c ", v >> 1, "+-"[v & 1]);
            LOGF(fp, "\n\tremoved_edges:\n");
            std::vector<std::pair<uint32_t, uint32_t>> on_best;
            for (size_t j = 0; j + 1 < best.size(); j++) on_best.push_back({best[j], best[j + 1]});
            std::sort(on_best.begin(), on_best.end());
            for (const auto& e : tour.touched) {   // every arc of the bubble that is not on the best path goes, in (vertex, key) order
                if (std::binary_search(on_best.begin(), on_best.end(), e)) continue;
                g.remove_edge(e.first >> 1, e.first & 1, e.second >> 1, e.second & 1);
                LOGF(fp, "\t\t%u:%c -> %u:%c\n", e.first >> 1, "+-"[e.first & 1], e.second >> 1, "+-"[e.second & 1]);
            }
            LOGF(fp, "\n");
            removed++;
            again = true;   // the node is looked at again, like the reference's i--
        }
        return again;
    });
    if (fp) fclose(fp);
    return removed;
}

namespace {
// Small bubbles (Cleaning.cpp:7-57): a triangle p -> i -> s with the shortcut p -> s. `p` is the vertex that leaves the predecessor
// towards i, `s` the key under which the successor is entered.
struct Triangle { uint32_t p, s, direct, in, out; };

bool first_triangle(const Graph& g, uint32_t i, Triangle& t) {
    for (const Arc& back : g.adj[(i << 1) | 1])        // arcs leaving i through its 5' end = its predecessors, seen from the other side
        for (const Arc& fwd : g.adj[i << 1]) {
            const uint32_t p = back.key ^ 1u;
            if (const Arc* d = g.find(p, fwd.key)) { t = Triangle{p, fwd.key, d->supp, back.supp, fwd.supp}; return true; }
        }
    return false;
}
}  // namespace

int clean_small_bubbles(Graph& g, const std::string& logpath) {
    FILE* fp = open_or_null(logpath, "w");
    int removed = 0;
    Triangle t;
    std::vector<uint64_t> cand = candidate_bits(g, [&](uint32_t i) { Triangle x; return first_triangle(g, i, x); });   // (removals only destroy triangles: no additions)
    for_candidates(g, cand, [&](uint32_t i) -> bool {
        if (!first_triangle(g, i, t)) return false;    // at most one per node and pass, and the node is not looked at again
        const double shortcut = t.direct, detour = (t.in + t.out) / 2.0;
        LOGF(fp, "small_bubble cov:%.2lf %u:%c -> %u:%c\n", shortcut, t.p >> 1, "+-"[t.p & 1], t.s >> 1, "+-"[t.s & 1]);
        LOGF(fp, "             cov:%.2lf %u:%c -> %u:%c -> %u:%c\n", detour, t.p >> 1, "+-"[t.p & 1], i, '+', t.s >> 1, "+-"[t.s & 1]);
        if (shortcut < detour) g.remove_edge(t.p >> 1, t.p & 1, t.s >> 1, t.s & 1);
        else { g.remove_edge(t.p >> 1, t.p & 1, i, 0); g.remove_edge(i, 0, t.s >> 1, t.s & 1); }
        removed++;
        return false;
    });
    if (fp) fclose(fp);
    return removed;
}

}  // namespace hxh
