// pipeline.cpp — one run of the haslr_assemble stage on the host, calling the compute backend
// (the HIP C-ABI) for everything per-read and per-edge.
//
// Stage order is the reference's main() (main.cpp:115-219). Host-side pieces restated here
// (paths under /root/reference/src/haslr_assemble/src/):
//   print_compact_longreads        Longread.cpp:675-693
//   edge work queue                Assemble.cpp:365-434 (each undirected edge once, from its lowest vertex)
//   asm_extract_all_simple_paths   Assemble.cpp:757-810
//   asm_assemble_single_path       Assemble.cpp:624-755  (asm.final.fa / asm.final.ann bytes)
//   asm_get_assembly               Assemble.cpp:1045-1077
#include <thread>
#include <mutex>
#include <condition_variable>
#include <memory>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <deque>

#include "host_internal.h"

namespace hxh {

#define LOGF(fp, ...) do { if (fp) fprintf(fp, __VA_ARGS__); } while (0)

// what the two per-edge stages leave behind for one entry of the work queue; in a multi-GPU run the entries of the other ranks arrive
// through hxh_run_results_import
struct EdgeResult {
    bool have_coords = false, have_cns = false;
    uint32_t head_end = 0, tail_beg = 0;
    std::vector<uint32_t> supp_lr, spos, epos;
    std::string cns;
};

struct Run {
    const Dataset* d = nullptr;
    hx_params prm{};
    hx_backend be{};
    std::string out_dir;
    hx_chain_out chain{};
    hx_edges_out edges{};
    hx_coords_out coords{};
    hx_cns_out cnsout{};
    bool have_chain = false, have_edges = false, have_coords = false, have_cns = false;
    Graph g;
    // the work queue (Assemble.cpp:365-434): every surviving undirected edge once, as (vertex, key) of the direction that is processed
    std::vector<std::pair<uint32_t, uint32_t>> work;
    std::vector<uint32_t> mine;     // entries of `work` this run computes (all of them unless the edges are sharded), ascending;
                                    // position = index in coords / cnsout
    std::vector<EdgeResult> res;    // per entry of `work`
    std::vector<std::string> cns;   // consensus strings; Arc::cns_id indexes this
    std::string fasta;
    std::vector<uint8_t> blob;      // last hxh_run_results_export
    std::string compact_text;       // last hxh_run_compact_text
    double t[5] = {0, 0, 0, 0, 0};
    bool async_writers = false;                 // hxh_run_set_async_writers / hxh_run_all: the graph stage returns while its GFA snapshots are still being written
    std::vector<std::thread> gfa_writers;       // the six GFA snapshots are written beside the per-edge stages; joined before the assembly is written (and on free)
    void join_writers() { for (auto& th : gfa_writers) if (th.joinable()) th.join(); gfa_writers.clear(); }
    uint32_t shard_rank = 0, shard_world = 1;   // multi-GPU: this run computes coordinates/consensus for its share of the edges
    uint32_t lr_begin = 0;                      // multi-GPU: first long read of the backend's read shard (ids in compact_uniq.txt)

    std::string path(const char* name) const { return out_dir.empty() ? std::string() : out_dir + "/" + name; }
    void release() {
        join_writers();
        if (have_cns) be.free_cns(be.ctx, &cnsout), have_cns = false;
        if (have_coords) be.free_coords(be.ctx, &coords), have_coords = false;
        if (have_edges) be.free_edges(be.ctx, &edges), have_edges = false;
        if (have_chain) be.free_chain(be.ctx, &chain), have_chain = false;
    }
};

namespace {
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int backend_fail(Run& r, const char* what) {
    g_err = std::string(what) + ": " + (r.be.last_error ? r.be.last_error() : "backend error");
    return -1;
}

// compact_uniq.txt lines (Longread.cpp:675-693) of the reads this run chained; read ids start at r.lr_begin
std::string compact_lines(const Run& r) {
    std::string out;
    char buf[96];
    const hx_chain_out& c = r.chain;
    for (uint32_t i = 0; i < c.n_reads; i++) {
        out.append(buf, snprintf(buf, sizeof(buf), ">%u\t", r.lr_begin + i));
        for (uint64_t j = c.cmp_off[i]; j < c.cmp_off[i + 1]; j++) {
            uint32_t a = c.cmp_aln[j], h = c.hit[a];
            out.append(buf, snprintf(buf, sizeof(buf), "%u-%u:%u:%c:%u-%u\t", c.q_start[a], c.q_end[a], r.d->t_id[h], r.d->is_rev[h] ? '-' : '+', c.t_start[a], c.t_end[a]));
        }
        out.push_back('\n');
    }
    return out;
}

void write_compact(const Run& r) {
    if (r.shard_world > 1) return;   // sharded reads: the launcher gathers hxh_run_compact_text of every rank and writes the file once
    FILE* fp = open_or_null(r.path("compact_uniq.txt"), "w");
    if (!fp) return;
    const std::string t = compact_lines(r);
    fwrite(t.data(), 1, t.size(), fp);
    fclose(fp);
}

// Assemble.cpp:365-434: walk vertices ascending, arcs ascending; an arc whose flag differs from `mark` is
// handed out and it and its twin are marked. Returns the hand-out order.
std::vector<std::pair<uint32_t, uint32_t>> work_queue(Graph& g, uint8_t mark) {
    std::vector<std::pair<uint32_t, uint32_t>> order;
    for (uint32_t v = 0; v < g.adj.size(); v++)
        for (Arc& a : g.adj[v]) {
            if (a.flag == mark) continue;
            a.flag = mark;
            Arc* tw = g.find(Graph::twin_vertex(a.key), Graph::twin_key(v));
            if (tw) tw->flag = mark;
            order.push_back({v, a.key});
        }
    return order;
}
}  // namespace

static int run_chain(Run& r) {
    double t0 = now();
    if (r.have_chain) r.be.free_chain(r.be.ctx, &r.chain), r.have_chain = false;
    if (r.be.chain_reads(r.be.ctx, &r.prm, &r.chain) != 0) return backend_fail(r, "chain_reads");
    r.have_chain = true;
    r.t[0] = now() - t0;
    write_compact(r);
    return 0;
}

// the six GFA snapshots carry every contig sequence (6 x the assembly size of text): each is written by its own thread from a copy of
// the arc list, while the cleaning passes AND the per-edge GPU stages (coordinates, consensus) go on - they depend only on the cleaned
// graph. They are joined before the assembly is written (run_assemble), when the stage is repeated, and when the run is freed.
// That overlap is what the pipeline as a whole does (hxh_run_all, the CLI: Run::async_writers); a graph stage called on its own returns
// when all six are on disk. HASLR_GFA_SYNC=1 forces that everywhere (A/B timing).
struct GfaWriters {
    Run& r;
    void start(const char* name) {
        if (r.out_dir.empty()) return;
        auto arcs = std::make_shared<std::vector<std::pair<uint32_t, uint32_t>>>(graph_arc_list(r.g));
        const Dataset* d = r.d;
        const std::string path = r.path(name);
        r.gfa_writers.emplace_back([arcs, d, path]() { graph_write_gfa_arcs(*arcs, *d, path); });
    }
};

static int run_graph(Run& r) {
    double t0 = now();
    // HASLR_GRAPH_DEBUG=1: where the host graph stage spends its time (stderr)
    const bool dbg = getenv("HASLR_GRAPH_DEBUG") != nullptr;
    double tl = t0;
    auto lap = [&](const char* what) { if (dbg) { const double t = now(); fprintf(stderr, "[hxh] graph stage: %-28s %7.1f ms\n", what, (t - tl) * 1e3); tl = t; } };
    r.join_writers();
    GfaWriters gfa{r};
    if (r.have_edges) r.be.free_edges(r.be.ctx, &r.edges), r.have_edges = false;
    if (r.be.edge_support(r.be.ctx, &r.prm, &r.edges) != 0) return backend_fail(r, "edge_support");
    r.have_edges = true;
    lap("edge_support (backend)");
    const Dataset& d = *r.d;
    Graph& g = r.g;
    graph_build(g, (uint32_t)d.contig_len.size(), r.edges);
    lap("build");
    if (dbg) fprintf(stderr, "[hxh] graph stage: cleaning passes on %u threads (%u nodes)\n", clean_threads(g.n_nodes), g.n_nodes);
    graph_write_stats(g, d, r.path("backbone.01.init.stat"));
    gfa.start("backbone.01.init.gfa");
    lap("stat + gfa 01");
    int nb = graph_remove_weak_edges(g, r.prm.min_edge_sup);
    if (!r.out_dir.empty()) fprintf(stderr, "       removed %d edges\n", nb);
    lap("weak edges");
    graph_write_stats(g, d, r.path("backbone.02.weakEdge.stat"));
    gfa.start("backbone.02.weakEdge.gfa");
    lap("stat + gfa 02");
    nb = clean_tips(g, 1, r.path("backbone.03.tip.log"));
    nb += clean_tips(g, 2, r.path("backbone.03.tip.log"));
    nb += clean_tips(g, 3, r.path("backbone.03.tip.log"));
    if (!r.out_dir.empty()) fprintf(stderr, "       removed %d tips\n", nb);
    lap("tips x3");
    graph_write_stats(g, d, r.path("backbone.03.tip.stat"));
    gfa.start("backbone.03.tip.gfa");
    lap("stat + gfa 03");
    nb = clean_simple_bubbles(g, 4, r.path("backbone.04.simplebubble.log"));
    if (!r.out_dir.empty()) fprintf(stderr, "       removed %d simple bubbles\n", nb);
    lap("simple bubbles");
    graph_write_stats(g, d, r.path("backbone.04.simplebubble.stat"));
    gfa.start("backbone.04.simplebubble.gfa");
    lap("stat + gfa 04");
    nb = clean_super_bubbles(g, r.path("backbone.05.superbubble.log"));
    if (!r.out_dir.empty()) fprintf(stderr, "       removed %d super bubbles\n", nb);
    lap("super bubbles");
    graph_write_stats(g, d, r.path("backbone.05.superbubble.stat"));
    gfa.start("backbone.05.superbubble.gfa");
    lap("stat + gfa 05");
    nb = clean_small_bubbles(g, r.path("backbone.06.smallbubble.log"));
    if (!r.out_dir.empty()) fprintf(stderr, "       removed %d small bubbles\n", nb);
    lap("small bubbles");
    graph_write_stats(g, d, r.path("backbone.06.smallbubble.stat"));
    gfa.start("backbone.06.smallbubble.gfa");
    graph_report_branching(g, r.path("backbone.branching.log"));
    lap("stat + gfa 06 + branching");
    if (!r.async_writers || getenv("HASLR_GFA_SYNC")) { r.join_writers(); lap("gfa writers joined"); }
    r.t[1] = now() - t0;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Per-edge stages. Both work on the queue of Assemble.cpp:365-434; a run computes the entries in `mine`
// and keeps what it learns in `res`, from where it is applied to the arcs (and, in a multi-GPU run,
// exported to / imported from the other ranks).
// ---------------------------------------------------------------------------------------------
namespace {

// DP cost of an edge before its coordinates are known: sequences x longest gap x nodes of the finished graph, with the gap
// taken from the read-space distance of the two anchors of every support and the graph growth hx_poa_batch plans with
uint64_t edge_cost_estimate(const Run& r, uint32_t dev_edge) {
    const hx_edges_out& e = r.edges;
    uint64_t lmax = 1, n = 0;
    for (uint64_t k = e.edge_off[dev_edge]; k < e.edge_off[dev_edge + 1]; k++, n++) {
        const bool twin = e.lr[k] >> 31;   // a twin record's head is the later anchor on the read
        const uint32_t from = twin ? e.tail.q_end[k] : e.head.q_end[k], to = twin ? e.head.q_start[k] : e.tail.q_start[k];
        if (to > from) lmax = std::max<uint64_t>(lmax, (uint64_t)to - from + 1);
    }
    const uint64_t nodes = lmax * (120 + 9 * n) / 100 + 1024;
    return nodes * lmax * std::max<uint64_t>(1, n);
}

// longest-processing-time dealing of the queue to the ranks: entries in descending cost, each to the rank with the least load so far
std::vector<uint32_t> my_share(const Run& r) {
    std::vector<uint32_t> all(r.work.size());
    for (uint32_t i = 0; i < all.size(); i++) all[i] = i;
    if (r.shard_world <= 1) return all;
    std::vector<uint64_t> cost(all.size());
    for (uint32_t i = 0; i < all.size(); i++) cost[i] = edge_cost_estimate(r, r.g.find(r.work[i].first, r.work[i].second)->dev_edge);
    std::stable_sort(all.begin(), all.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
    std::vector<uint64_t> load(r.shard_world, 0);
    std::vector<uint32_t> mine;
    for (uint32_t i : all) {
        const uint32_t to = (uint32_t)(std::min_element(load.begin(), load.end()) - load.begin());
        load[to] += cost[i];
        if (to == r.shard_rank) mine.push_back(i);
    }
    std::sort(mine.begin(), mine.end());
    return mine;
}

struct ArcPair { Arc* a; Arc* tw; };
ArcPair arcs_of(Run& r, uint32_t gi) {
    const uint32_t v = r.work[gi].first, key = r.work[gi].second;
    return {r.g.find(v, key), r.g.find(Graph::twin_vertex(key), Graph::twin_key(v))};
}

void apply_coords(Run& r, uint32_t gi) {
    const EdgeResult& x = r.res[gi];
    const ArcPair p = arcs_of(r, gi);
    const uint32_t n = (uint32_t)x.supp_lr.size();
    p.a->head_end = x.head_end; p.a->tail_beg = x.tail_beg; p.a->n_cns_supp = n;
    if (p.tw == p.a) p.a->head_end = p.a->tail_beg = x.tail_beg;   // self-twin (hairpin): edge1 and edge2 are one object in the reference,
                                                                    // the second chained assignment wins (Assemble.cpp:351-352)
    else if (p.tw) { p.tw->tail_beg = x.head_end; p.tw->head_end = x.tail_beg; p.tw->n_cns_supp = n; }
}

void apply_cns(Run& r, uint32_t gi) {
    const EdgeResult& x = r.res[gi];
    const ArcPair p = arcs_of(r, gi);
    p.a->cns_id = (int32_t)r.cns.size();
    r.cns.push_back(x.cns);
    if (p.tw && p.tw != p.a) { p.tw->cns_id = (int32_t)r.cns.size(); r.cns.push_back(revcomp(x.cns)); }   // Assemble.cpp:555
    else if (p.tw == p.a) r.cns.back() = revcomp(x.cns);   // self-twin (hairpin): the reference's edge2 assignment overwrites edge1's
}

// reduced forms of the reference's diagnostic logs (Assemble.cpp:176-362, :501-557), every entry of the queue in queue order
void write_stage_logs(Run& r) {
    FILE* fk = open_or_null(r.path("log_coordinate.txt"), "w");
    FILE* fc = open_or_null(r.path("log_consensus.txt"), "w");
    for (uint32_t gi = 0; gi < r.work.size() && (fk || fc); gi++) {
        const uint32_t v = r.work[gi].first, key = r.work[gi].second;
        const EdgeResult& x = r.res[gi];
        const Arc* a = r.g.find(v, key);
        if (fk && x.have_coords) {
            fprintf(fk, "edge      %u:%c -> %u:%c\n", v >> 1, "+-"[v & 1], key >> 1, "+-"[key & 1]);
            fprintf(fk, "edge_twin %u:%c -> %u:%c\n", key >> 1, "+-"[1 - (key & 1)], v >> 1, "+-"[1 - (v & 1)]);
            fprintf(fk, "\tedge_supp size:%u\n", a->supp);
            fprintf(fk, "coordinates contig1_pos: %u\tcontig2_pos: %u\n", a->head_end, a->tail_beg);
            for (size_t k = 0; k < x.supp_lr.size(); k++)
                fprintf(fk, "    +++ lr:%u strand:%c [coordinate] lr_start:%u lr_end:%u\n", x.supp_lr[k] & 0x7fffffffu, "+-"[x.supp_lr[k] >> 31], x.spos[k], x.epos[k]);
            fprintf(fk, "\n");
        }
        if (fc && x.have_cns) {
            fprintf(fc, "calc_cns %u:%c -> %u:%c\n", v >> 1, "+-"[v & 1], key >> 1, "+-"[key & 1]);
            fprintf(fc, "[shared_region] head_end:%u\ttail_beg:%u\n", a->head_end, a->tail_beg);
            fprintf(fc, ">CONSENSUS\n%s\n", x.cns.c_str());
        }
    }
    if (fk) fclose(fk);
    if (fc) fclose(fc);
}

size_t missing_results(const Run& r) {
    size_t n = 0;
    for (const EdgeResult& x : r.res) n += !(x.have_coords && x.have_cns);
    return n;
}

}  // namespace

static int run_coords(Run& r) {
    double t0 = now();
    r.work = work_queue(r.g, 11);
    r.res.assign(r.work.size(), EdgeResult());
    r.cns.clear();
    r.mine = my_share(r);
    std::vector<uint32_t> sel(r.mine.size());
    for (size_t i = 0; i < r.mine.size(); i++) sel[i] = r.g.find(r.work[r.mine[i]].first, r.work[r.mine[i]].second)->dev_edge;
    if (r.have_coords) r.be.free_coords(r.be.ctx, &r.coords), r.have_coords = false;
    if (r.be.edge_coords(r.be.ctx, (uint32_t)sel.size(), sel.data(), &r.coords) != 0) return backend_fail(r, "edge_coords");
    r.have_coords = true;
    for (size_t i = 0; i < r.mine.size(); i++) {
        EdgeResult& x = r.res[r.mine[i]];
        const uint64_t b = r.coords.supp_off[i], e = r.coords.supp_off[i + 1];
        x.have_coords = true;
        x.head_end = r.coords.head_end[i]; x.tail_beg = r.coords.tail_beg[i];
        x.supp_lr.assign(r.coords.supp_lr + b, r.coords.supp_lr + e);
        x.spos.assign(r.coords.spos + b, r.coords.spos + e);
        x.epos.assign(r.coords.epos + b, r.coords.epos + e);
        apply_coords(r, r.mine[i]);
    }
    r.t[2] = now() - t0;
    return 0;
}

static int run_consensus(Run& r) {
    double t0 = now();
    // second pass of the work queue (flag 12) hands out the same arcs in the same order as the first
    if (work_queue(r.g, 12) != r.work) { g_err = "internal: consensus work queue differs from coordinate work queue"; return -1; }
    hx_poa_params pp{5, -4, -8};   // Assemble.cpp:8-11
    if (r.have_cns) r.be.free_cns(r.be.ctx, &r.cnsout), r.have_cns = false;
    if (!r.cns.empty()) {   // a repeated call: the strings of the earlier one go, those imported from other ranks are applied again
        r.cns.clear();
        for (uint32_t gi : r.mine) r.res[gi].have_cns = false;
        for (uint32_t gi = 0; gi < r.res.size(); gi++) if (r.res[gi].have_cns) apply_cns(r, gi);
    }
    if (r.be.poa_batch(r.be.ctx, &pp, &r.cnsout) != 0) return backend_fail(r, "poa_batch");
    const double t_poa = now();
    r.have_cns = true;
    if (r.cnsout.n_edge != r.mine.size()) { g_err = "internal: consensus count differs from this run's share of the work queue"; return -1; }
    {
        // The strings go to their arcs in two steps: the ids in queue order (serial: an id is the string's place in r.cns, Assemble.cpp:555 gives the twin the
        // reverse complement), then the copies and the reverse complements on several threads - 60 MB of byte work at 140 Mb, 30 ms of a 0.5 s step on one.
        struct Job { uint32_t i; int32_t fwd, rev; bool hairpin; };
        std::vector<Job> jobs(r.mine.size());
        for (size_t i = 0; i < r.mine.size(); i++) {
            const ArcPair p = arcs_of(r, r.mine[i]);
            Job& j = jobs[i];
            j = Job{(uint32_t)i, (int32_t)r.cns.size(), -1, false};
            p.a->cns_id = j.fwd;
            r.cns.emplace_back();
            if (p.tw && p.tw != p.a) { j.rev = (int32_t)r.cns.size(); p.tw->cns_id = j.rev; r.cns.emplace_back(); }
            else if (p.tw == p.a) j.hairpin = true;   // self-twin (hairpin): the reference's edge2 assignment overwrites edge1's
        }
        const uint64_t bytes = r.cnsout.cns_off[r.mine.size()];
        const unsigned T = bytes < (4u << 20) ? 1u : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        run_parallel(T, [&](unsigned t) {
            for (size_t q = jobs.size() * t / T; q < jobs.size() * (t + 1) / T; q++) {
                const Job& j = jobs[q];
                EdgeResult& x = r.res[r.mine[j.i]];
                x.cns.assign(r.cnsout.cns + r.cnsout.cns_off[j.i], r.cnsout.cns + r.cnsout.cns_off[j.i + 1]);
                x.have_cns = true;
                if (j.hairpin) r.cns[(size_t)j.fwd] = revcomp(x.cns);
                else {
                    r.cns[(size_t)j.fwd] = x.cns;
                    if (j.rev >= 0) r.cns[(size_t)j.rev] = revcomp(x.cns);
                }
            }
        });
    }
    const double t_apply = now();
    if (r.shard_world == 1) write_stage_logs(r);
    r.t[3] = now() - t0;
    if (getenv("HASLR_GRAPH_DEBUG")) fprintf(stderr, "[consensus] queue check + backend %.2f ms, results applied %.2f ms, stage logs %.2f ms\n", (t_poa - t0) * 1e3, (t_apply - t_poa) * 1e3, (now() - t_apply) * 1e3);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Multi-GPU: results of this run's share of the queue as one self-delimiting blob of 32-bit words
//   { magic, n_items, words in the blob } then per item { queue index, head_end, tail_beg, n_supp, cns bytes,
//   n_supp x (lr, spos, epos), consensus padded to a word }.
// A rank imports the concatenation of every rank's blob (its own entries are skipped).
// ---------------------------------------------------------------------------------------------
namespace {
constexpr uint32_t kBlobMagic = 0x31525848u;   // "HXR1"

int export_results(Run& r) {
    std::vector<uint32_t> w{kBlobMagic, (uint32_t)r.mine.size(), 0u};
    for (uint32_t gi : r.mine) {
        const EdgeResult& x = r.res[gi];
        const uint32_t hdr[5] = {gi, x.head_end, x.tail_beg, (uint32_t)x.supp_lr.size(), (uint32_t)x.cns.size()};
        w.insert(w.end(), hdr, hdr + 5);
        for (size_t k = 0; k < x.supp_lr.size(); k++) { w.push_back(x.supp_lr[k]); w.push_back(x.spos[k]); w.push_back(x.epos[k]); }
        const size_t at = w.size();
        w.resize(at + (x.cns.size() + 3) / 4, 0u);
        memcpy(w.data() + at, x.cns.data(), x.cns.size());
    }
    if (w.size() > 0xffffffffull) { g_err = "results export: this rank's share exceeds 2^32 words (the blob's length field is 32 bits)"; return -1; }
    w[2] = (uint32_t)w.size();
    r.blob.resize(w.size() * 4);
    memcpy(r.blob.data(), w.data(), r.blob.size());
    return 0;
}

int import_results(Run& r, const uint8_t* buf, uint64_t len) {
    if (len % 4) { g_err = "results import: length is not a multiple of 4"; return -1; }
    std::vector<uint32_t> w(len / 4);
    memcpy(w.data(), buf, len);
    size_t at = 0;
    while (at < w.size()) {
        if (w.size() - at < 3 || w[at] != kBlobMagic || w[at + 2] < 3 || w[at + 2] > w.size() - at) { g_err = "results import: malformed blob header"; return -1; }
        const size_t end = at + w[at + 2];
        const uint32_t n_items = w[at + 1];
        at += 3;
        for (uint32_t it = 0; it < n_items; it++) {
            if (end - at < 5) { g_err = "results import: truncated item"; return -1; }
            const uint32_t gi = w[at], ns = w[at + 3], nc = w[at + 4];
            const size_t need = 5 + (size_t)ns * 3 + ((size_t)nc + 3) / 4;
            if (gi >= r.work.size() || end - at < need) { g_err = "results import: item does not fit (different graph on the sending rank?)"; return -1; }
            EdgeResult& x = r.res[gi];
            if (!(x.have_coords && x.have_cns)) {   // (own entries and repeats are skipped)
                x.head_end = w[at + 1]; x.tail_beg = w[at + 2];
                x.supp_lr.resize(ns); x.spos.resize(ns); x.epos.resize(ns);
                for (uint32_t k = 0; k < ns; k++) { x.supp_lr[k] = w[at + 5 + 3 * k]; x.spos[k] = w[at + 6 + 3 * k]; x.epos[k] = w[at + 7 + 3 * k]; }
                x.cns.assign(reinterpret_cast<const char*>(w.data() + at + 5 + (size_t)ns * 3), nc);
                x.have_coords = x.have_cns = true;
                apply_coords(r, gi);
                apply_cns(r, gi);
            }
            at += need;
        }
        if (at != end) { g_err = "results import: blob length does not match its items"; return -1; }
    }
    if (missing_results(r) == 0) write_stage_logs(r);
    return 0;
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// Path extraction and stitching (Assemble.cpp:757-810, :624-755): a path of anchor contigs becomes one output
// record, or several when an edge without consensus support breaks it. A record is planned as a list of
// pieces (contig slices and consensus strings) and rendered once into asm.final.fa / .ann / log_asmfinal.txt.
// ---------------------------------------------------------------------------------------------
namespace {
struct Anchor { uint32_t strand, id; };

struct Piece {
    const std::string* cns = nullptr;   // a gap consensus ...
    uint32_t n_supp = 0;
    uint32_t contig = 0, strand = 0;    // ... or bases [lo, lo + n) of a contig, reverse-complemented on strand 1
    uint64_t lo = 0, n = 0, ann_lo = 0, ann_hi = 0;
};

struct Record {
    Anchor from{}, to{};
    std::vector<Piece> pieces;
    std::vector<std::string> notes;     // log_asmfinal.txt lines that precede the record
};

// what std::string::substr(pos, count) keeps of a string of `len` characters
uint64_t kept(uint64_t len, uint64_t pos, uint64_t count) { return pos >= len ? 0 : std::min(count, len - pos); }

std::string slice_text(const Dataset& d, const Piece& p) {
    std::string s(p.n, 'A');
    const uint8_t* q = d.contig_packed.data() + d.contig_off[p.contig];
    if (p.strand == 0)
        for (uint64_t i = 0; i < p.n; i++) { const uint64_t b = p.lo + i; s[i] = "ACGT"[(q[b >> 2] >> ((b & 3) * 2)) & 3]; }
    else
        for (uint64_t i = 0; i < p.n; i++) { const uint64_t b = p.lo + p.n - 1 - i; s[i] = "TGCA"[(q[b >> 2] >> ((b & 3) * 2)) & 3]; }
    return s;
}

std::string note(const char* what, uint64_t len, uint32_t cursor, const Arc& e) {
    char buf[160];
    snprintf(buf, sizeof(buf), "[%s] contig1_len:%zu    contig1_start:%u    prev_end:%u     next_beg:%u\n", what, (size_t)len, cursor, e.head_end, e.tail_beg);
    return buf;
}

// the rest of a contig from the cursor to the end it is left through (path orientation)
Piece to_contig_end(const Dataset& d, Anchor c, uint32_t cursor) {
    Piece p;
    const uint64_t len = d.contig_len[c.id];
    p.contig = c.id; p.strand = c.strand;
    if (c.strand == 0) { p.lo = cursor; p.n = kept(len, cursor, UINT64_MAX); p.ann_lo = cursor; p.ann_hi = len; }
    else { p.lo = 0; p.n = kept(len, 0, (uint64_t)cursor + 1); p.ann_lo = 0; p.ann_hi = (uint64_t)cursor + 1; }
    return p;
}

// from the cursor up to the base where the edge's consensus takes over (head_end, inclusive); counts are 32-bit like the reference's
Piece to_head_end(const Dataset& d, Anchor c, uint32_t cursor, const Arc& e) {
    Piece p;
    const uint64_t len = d.contig_len[c.id];
    p.contig = c.id; p.strand = c.strand;
    if (c.strand == 0) { p.lo = cursor; p.n = kept(len, cursor, (uint32_t)(e.head_end - cursor + 1)); }
    else { p.lo = e.head_end; p.n = kept(len, e.head_end, (uint32_t)(cursor - e.head_end + 1)); }
    p.ann_lo = p.lo; p.ann_hi = p.lo + p.n;
    return p;
}

std::vector<Record> plan_path(Run& r, const std::deque<Anchor>& path) {
    const Dataset& d = *r.d;
    std::vector<Record> out;
    auto entry = [&](Anchor a) { return a.strand == 0 ? 0u : d.contig_len[a.id] - 1; };   // where a contig is entered when nothing precedes it
    Record cur;
    cur.from = path.front();
    uint32_t cursor = entry(path.front());
    for (size_t i = 0; i + 1 < path.size(); i++) {
        const Anchor c1 = path[i], c2 = path[i + 1];
        const Arc& e = *r.g.find((c1.id << 1) | c1.strand, (c2.id << 1) | c2.strand);
        if (e.n_cns_supp == 0) {   // no read bridges the gap: the record ends with the rest of c1, the next one starts at c2
            cur.notes.push_back(note("breaking", d.contig_len[c1.id], cursor, e));
            cur.pieces.push_back(to_contig_end(d, c1, cursor));
            cur.to = c1;
            out.push_back(std::move(cur));
            cur = Record();
            cur.from = c2;
            cursor = entry(c2);
            if (!r.out_dir.empty())
                fprintf(stderr, "[WARNING] breaking assembly for path %u:%c --> %u:%c between anchors %u:%c --> %u:%c\n", c2.id, "+-"[c2.strand], path.back().id,
                        "+-"[path.back().strand], c1.id, "+-"[c1.strand], c2.id, "+-"[c2.strand]);
        } else {
            cur.notes.push_back(note("stitching", d.contig_len[c1.id], cursor, e));
            cur.pieces.push_back(to_head_end(d, c1, cursor, e));
            Piece g;
            g.cns = &r.cns[e.cns_id]; g.n_supp = e.n_cns_supp;
            cur.pieces.push_back(g);
            cursor = e.tail_beg;
        }
    }
    if (path.size() == 1) {   // a lone anchor is written as it is, whatever its strand in the path, and leaves no annotation
        Piece whole;
        whole.contig = path.front().id; whole.n = d.contig_len[whole.contig];
        cur.pieces.push_back(whole);
    } else cur.pieces.push_back(to_contig_end(d, path.back(), cursor));
    cur.to = path.back();
    out.push_back(std::move(cur));
    return out;
}

void render(Run& r, const Record& rec, bool annotate, int id, FILE* fp_asm, FILE* fp_ann, FILE* fp_log) {
    std::string seq;
    for (const std::string& n : rec.notes) LOGF(fp_log, "%s", n.c_str());
    for (const Piece& p : rec.pieces) {
        const std::string text = p.cns ? *p.cns : slice_text(*r.d, p);
        if (annotate && p.cns) LOGF(fp_ann, "%d\t%zu\t%zu\tcns\t%zu\t%u\n", id, seq.size(), seq.size() + text.size(), text.size(), p.n_supp);
        else if (annotate)
            LOGF(fp_ann, "%d\t%zu\t%zu\tctg\t%c\t%u\t%zu\t%zu\t%zu\n", id, seq.size(), seq.size() + text.size(), "+-"[p.strand], p.contig, (size_t)r.d->contig_len[p.contig],
                 (size_t)p.ann_lo, (size_t)p.ann_hi);
        seq += text;
    }
    char hdr[128];
    const int n = snprintf(hdr, sizeof(hdr), ">%d from:%u:%c to:%u:%c\n", id, rec.from.id, "+-"[rec.from.strand], rec.to.id, "+-"[rec.to.strand]);
    LOGF(fp_log, "%s%s\n\n", hdr, seq.c_str());
    r.fasta.append(hdr, n); r.fasta += seq; r.fasta.push_back('\n');
    if (fp_asm) { fputs(hdr, fp_asm); fwrite(seq.data(), 1, seq.size(), fp_asm); fputc('\n', fp_asm); }
}

// asm_extract_all_simple_paths (Assemble.cpp:757-810): paths start at every node that is not a plain link (one arc on each side)
std::vector<std::deque<Anchor>> simple_paths(Graph& g) {
    const uint8_t seen = 21;
    std::vector<std::deque<Anchor>> paths;
    auto mark = [&](Anchor a, Anchor b) {
        const uint32_t v = (a.id << 1) | a.strand, key = (b.id << 1) | b.strand;
        if (Arc* x = g.find(v, key)) x->flag = seen;
        if (Arc* x = g.find(Graph::twin_vertex(key), Graph::twin_key(v))) x->flag = seen;
    };
    for (uint32_t i = 0; i < g.n_nodes; i++) {
        const size_t d0 = g.deg(i, 0), d1 = g.deg(i, 1);
        if (d0 == 1 && d1 == 1) continue;
        if (d0 > 1 && d1 > 1) paths.push_back({Anchor{0, i}});
        for (uint32_t side = 0; side < 2; side++)
            for (size_t k = 0; k < g.adj[(i << 1) | side].size(); k++) {
                if (g.adj[(i << 1) | side][k].flag == seen) continue;
                std::deque<Anchor> p{Anchor{side, i}};
                for (uint32_t key = g.adj[(i << 1) | side][k].key;;) {   // asm_find_simple_path_from_source (Assemble.cpp:607-622)
                    const Anchor at{key & 1, key >> 1};
                    p.push_back(at);
                    if (g.deg(at.id, at.strand) != 1 || g.deg(at.id, 1 - at.strand) > 1) break;
                    key = g.adj[(at.id << 1) | at.strand][0].key;
                }
                for (size_t j = 0; j + 1 < p.size(); j++) mark(p[j], p[j + 1]);
                if (g.deg(p.front().id, p.front().strand) > 1) p.pop_front();                  // a branching end belongs to no path
                if (!p.empty() && g.deg(p.back().id, 1 - p.back().strand) > 1) p.pop_back();
                if (!p.empty()) paths.push_back(std::move(p));
            }
    }
    return paths;
}
}  // namespace

static int run_assemble(Run& r) {
    double t0 = now();
    r.join_writers();   // the GFA snapshots are complete before the stage's last outputs appear
    if (const size_t miss = missing_results(r)) {
        g_err = "assemble: " + std::to_string(miss) + " of " + std::to_string(r.res.size()) + " edges have no coordinates / consensus in this run" +
                (r.shard_world > 1 ? " (multi-GPU: the other ranks' results must be imported first, hxh_run_results_import)" : " (coords and consensus stages must run first)");
        return -1;
    }
    const std::vector<std::deque<Anchor>> paths = simple_paths(r.g);
    FILE* fp_asm = open_or_null(r.path("asm.final.fa"), "w");
    FILE* fp_ann = open_or_null(r.path("asm.final.ann"), "w");
    FILE* fp_log = open_or_null(r.path("log_asmfinal.txt"), "w");
    for (size_t i = 0; i < paths.size(); i++)
        LOGF(fp_log, "simple_path %zu size:%zu\tfrom:%u:%c\tto:%u:%c\n", i, paths[i].size(), paths[i].front().id, "+-"[paths[i].front().strand], paths[i].back().id, "+-"[paths[i].back().strand]);
    r.fasta.clear();
    int n_out = 0;
    for (const auto& p : paths)
        for (const Record& rec : plan_path(r, p)) render(r, rec, p.size() > 1, n_out++, fp_asm, fp_ann, fp_log);
    if (fp_log) fclose(fp_log);
    if (fp_ann) fclose(fp_ann);
    if (fp_asm) fclose(fp_asm);
    r.t[4] = now() - t0;
    return 0;
}

}  // namespace hxh

using namespace hxh;

extern "C" hxh_run* hxh_run_create(const hxh_dataset* ds, const hx_params* prm, const hx_backend* be, const char* out_dir) {
    Run* r = new Run;
    r->d = reinterpret_cast<const Dataset*>(ds);
    r->prm = *prm;
    r->be = *be;
    r->out_dir = out_dir ? out_dir : "";
    return reinterpret_cast<hxh_run*>(r);
}
extern "C" void hxh_run_free(hxh_run* p) { Run* r = reinterpret_cast<Run*>(p); if (r) { r->release(); delete r; } }
extern "C" int hxh_run_chain(hxh_run* p) { return run_chain(*reinterpret_cast<Run*>(p)); }
extern "C" int hxh_run_graph(hxh_run* p) { return run_graph(*reinterpret_cast<Run*>(p)); }
extern "C" int hxh_run_coords(hxh_run* p) { return run_coords(*reinterpret_cast<Run*>(p)); }
extern "C" int hxh_run_consensus(hxh_run* p) { return run_consensus(*reinterpret_cast<Run*>(p)); }
extern "C" int hxh_run_assemble(hxh_run* p) { return run_assemble(*reinterpret_cast<Run*>(p)); }
extern "C" void hxh_run_set_async_writers(hxh_run* p, int on) { reinterpret_cast<Run*>(p)->async_writers = on != 0; }
extern "C" int hxh_run_all(hxh_run* p) {
    Run& r = *reinterpret_cast<Run*>(p);
    struct Async { Run& r; bool was; ~Async() { r.async_writers = was; } } guard{r, r.async_writers};
    r.async_writers = true;   // (run_assemble joins the writers)
    int rc;
    if ((rc = run_chain(r))) return rc;
    if ((rc = run_graph(r))) return rc;
    if ((rc = run_coords(r))) return rc;
    if ((rc = run_consensus(r))) return rc;
    return run_assemble(r);
}
extern "C" void hxh_run_set_edge_shard(hxh_run* p, uint32_t rank, uint32_t world) {
    Run* r = reinterpret_cast<Run*>(p);
    r->shard_rank = rank; r->shard_world = world ? world : 1;
}
extern "C" void hxh_run_set_read_shard(hxh_run* p, uint32_t lr_begin) { reinterpret_cast<Run*>(p)->lr_begin = lr_begin; }
extern "C" int hxh_run_results_export(hxh_run* p, const uint8_t** buf, uint64_t* len) {
    Run* r = reinterpret_cast<Run*>(p);
    for (uint32_t gi : r->mine) if (!(r->res[gi].have_coords && r->res[gi].have_cns)) { g_err = "results export: the coords and consensus stages have not run"; return -1; }
    if (export_results(*r) != 0) return -1;
    *buf = r->blob.data(); *len = r->blob.size();
    return 0;
}
extern "C" int hxh_run_results_import(hxh_run* p, const uint8_t* buf, uint64_t len) { return import_results(*reinterpret_cast<Run*>(p), buf, len); }
extern "C" uint64_t hxh_run_results_missing(const hxh_run* p) { return missing_results(*reinterpret_cast<const Run*>(p)); }
extern "C" const char* hxh_run_compact_text(hxh_run* p, uint64_t* len) {
    Run* r = reinterpret_cast<Run*>(p);
    r->compact_text = r->have_chain ? compact_lines(*r) : std::string();
    if (len) *len = r->compact_text.size();
    return r->compact_text.c_str();
}
extern "C" int hxh_run_write_longread_index(const hxh_run* p, const char* path) {
    const Run* r = reinterpret_cast<const Run*>(p);
    if (!r->have_chain) { g_err = "index.longread needs the chain stage"; return -1; }
    return write_longread_index(*r->d, r->chain, path) ? 0 : -1;
}
extern "C" void hxh_run_timings(const hxh_run* p, double out[5]) { memcpy(out, reinterpret_cast<const Run*>(p)->t, sizeof(double) * 5); }
extern "C" uint32_t hxh_run_n_edges(const hxh_run* p) { return (uint32_t)reinterpret_cast<const Run*>(p)->mine.size(); }
extern "C" uint32_t hxh_run_n_edges_total(const hxh_run* p) { return (uint32_t)reinterpret_cast<const Run*>(p)->work.size(); }
extern "C" const hx_chain_out* hxh_run_chain_out(const hxh_run* p) { return &reinterpret_cast<const Run*>(p)->chain; }
extern "C" const hx_edges_out* hxh_run_edges_out(const hxh_run* p) { return &reinterpret_cast<const Run*>(p)->edges; }
extern "C" const hx_coords_out* hxh_run_coords_out(const hxh_run* p) { return &reinterpret_cast<const Run*>(p)->coords; }
extern "C" const hx_cns_out* hxh_run_cns_out(const hxh_run* p) { return &reinterpret_cast<const Run*>(p)->cnsout; }
extern "C" const char* hxh_run_assembly_fasta(const hxh_run* p, uint64_t* len) {
    const Run* r = reinterpret_cast<const Run*>(p);
    if (len) *len = r->fasta.size();
    return r->fasta.c_str();
}

// ---------------------------------------------------------------------------------------------
// The whole stage over n ranks INSIDE one process (one host thread per rank, one GPU per rank behind the ranks' backend tables): what the
// reference's worker threads are to asm_calc_edge_coordinates_MT / asm_cal_cns_seq_MT (Assemble.cpp:453-477, :580-605), with the reads
// sharded as well. Between the stages the ranks agree on success - a rank that failed never leaves the others waiting inside the
// edge-record collective - and the per-edge results travel through the process's memory (export -> import), no second collective.
// ---------------------------------------------------------------------------------------------
extern "C" void hxh_shard_bounds(const hxh_dataset* p, uint32_t n, uint32_t* bounds) {
    // contiguous read-id ranges with about equal numbers of raw PAF records (SURVEY.md 8e phase 1)
    const Dataset* d = reinterpret_cast<const Dataset*>(p);
    const std::vector<uint64_t>& rho = d->read_hit_off;
    const uint32_t nr = (uint32_t)rho.size() - 1;
    const uint64_t total = rho[nr];
    bounds[0] = 0;
    for (uint32_t r = 1; r < n; r++) {
        const uint64_t want = total / n * r + total % n * r / n;
        bounds[r] = std::max<uint32_t>(bounds[r - 1], (uint32_t)(std::lower_bound(rho.begin(), rho.end(), want) - rho.begin()));
        if (bounds[r] > nr) bounds[r] = nr;
    }
    bounds[n] = nr;
}

namespace {
struct RankBarrier {   // everybody arrives with a status, everybody leaves with the worst one
    std::mutex mu; std::condition_variable cv;
    uint32_t n, arrived = 0; int worst = 0, agreed = 0; uint64_t gen = 0;
    explicit RankBarrier(uint32_t n_) : n(n_) {}
    int meet(int status) {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t g = gen;
        worst = std::max(worst, status);
        if (++arrived == n) { agreed = worst; worst = 0; arrived = 0; gen++; cv.notify_all(); return agreed; }
        cv.wait(lk, [&] { return gen != g; });
        return agreed;
    }
};
}  // namespace

extern "C" int hxh_runs_all_sharded(hxh_run** runs, uint32_t n, const uint32_t* read_begin, void (*on_stage)(int stage, int begin, void* user), void* user) {
    if (n == 0) { g_err = "sharded run: no ranks"; return -1; }
    std::vector<Run*> R(n);
    for (uint32_t r = 0; r < n; r++) R[r] = reinterpret_cast<Run*>(runs[r]);
    RankBarrier bar(n);
    std::vector<std::string> errs(n);
    std::vector<int> rcs(n, 0);
    auto body = [&](uint32_t r) {
        Run& me = *R[r];
        me.shard_rank = r; me.shard_world = n; me.lr_begin = read_begin[r];
        me.async_writers = true;
        int (*stages[4])(Run&) = {run_chain, run_graph, run_coords, run_consensus};
        for (int s = 0; s < 4; s++) {
            if (r == 0 && on_stage) on_stage(s, 1, user);
            int rc = stages[s](me);
            if (rc) errs[r] = g_err;
            if (bar.meet(rc != 0)) { rcs[r] = -1; if (errs[r].empty()) errs[r] = "another rank failed"; return; }
            if (r == 0 && on_stage) on_stage(s, 0, user);
        }
        // results: every rank publishes its share, every rank takes the others'
        int rc = export_results(me);
        if (rc) errs[r] = g_err;
        if (bar.meet(rc != 0)) { rcs[r] = -1; if (errs[r].empty()) errs[r] = "another rank failed"; return; }
        for (uint32_t q = 0; q < n && !rc; q++)
            if (q != r && import_results(me, R[q]->blob.data(), R[q]->blob.size()) != 0) { rc = -1; errs[r] = g_err; }
        if (!rc && missing_results(me)) { rc = -1; errs[r] = std::to_string(missing_results(me)) + " edges are without results after the exchange"; }
        if (bar.meet(rc != 0)) { rcs[r] = -1; if (errs[r].empty()) errs[r] = "another rank failed"; return; }
        if (r == 0) {
            if (on_stage) on_stage(4, 1, user);
            // compact_uniq.txt lists every read: the ranks' lines in rank order = read order
            if (FILE* fp = open_or_null(me.path("compact_uniq.txt"), "w")) {
                for (uint32_t q = 0; q < n; q++) { const std::string t = compact_lines(*R[q]); fwrite(t.data(), 1, t.size(), fp); }
                fclose(fp);
            }
            if (run_assemble(me) != 0) { rcs[r] = -1; errs[r] = g_err; }
            if (on_stage) on_stage(4, 0, user);
        }
    };
    std::vector<std::thread> th;
    for (uint32_t r = 1; r < n; r++) th.emplace_back(body, r);
    body(0);
    for (auto& t : th) t.join();
    for (uint32_t r = 0; r < n; r++) R[r]->join_writers();
    for (uint32_t r = 0; r < n; r++)
        if (rcs[r]) {   // the first rank with an error of its own tells what went wrong
            uint32_t w = r;
            for (uint32_t q = 0; q < n; q++) if (rcs[q] && errs[q] != "another rank failed") { w = q; break; }
            g_err = "rank " + std::to_string(w) + ": " + errs[w];
            return -1;
        }
    return 0;
}

// index.longread of a sharded pass: the filtered alignments of ALL reads = the ranks' chain outputs in rank order (ranks own ascending read
// ranges). The writer looks at the per-read alignment counts and the PAF record of every alignment only (Longread.cpp:322-339).
extern "C" int hxh_runs_write_longread_index(hxh_run* const* runs, uint32_t n, const char* path) {
    if (n == 0) { g_err = "index.longread: no ranks"; return -1; }
    std::vector<uint32_t> hit;
    std::vector<uint64_t> read_off{0};
    const Dataset* d = reinterpret_cast<const Run*>(runs[0])->d;
    for (uint32_t r = 0; r < n; r++) {
        const Run* R = reinterpret_cast<const Run*>(runs[r]);
        if (!R->have_chain) { g_err = "index.longread needs the chain stage of every rank"; return -1; }
        const hx_chain_out& c = R->chain;
        const uint64_t base = hit.size();
        hit.insert(hit.end(), c.hit, c.hit + c.n_aln);
        for (uint32_t i = 0; i < c.n_reads; i++) read_off.push_back(base + c.read_off[i + 1]);
    }
    hx_chain_out all{};
    all.n_reads = (uint32_t)(read_off.size() - 1); all.n_aln = hit.size();
    all.hit = hit.data(); all.read_off = read_off.data();
    return write_longread_index(*d, all, path) ? 0 : -1;
}
