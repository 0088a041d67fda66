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
#include <memory>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <deque>

#include "host_internal.h"

namespace hxh {

#define LOGF(fp, ...) do { if (fp) fprintf(fp, __VA_ARGS__); } while (0)

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
    // processed arcs in work-queue order: (vertex, key); index = position in coords / cnsout
    std::vector<std::pair<uint32_t, uint32_t>> work;
    std::vector<std::string> cns;   // consensus strings; Arc::cns_id indexes this
    std::string fasta;
    double t[5] = {0, 0, 0, 0, 0};
    uint32_t shard_rank = 0, shard_world = 1;   // multi-GPU: this run computes coordinates/consensus for its share of the edges

    std::string path(const char* name) const { return out_dir.empty() ? std::string() : out_dir + "/" + name; }
    void release() {
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

void write_compact(const Run& r) {
    FILE* fp = open_or_null(r.path("compact_uniq.txt"), "w");
    if (!fp) return;
    const hx_chain_out& c = r.chain;
    for (uint32_t i = 0; i < c.n_reads; i++) {
        fprintf(fp, ">%u\t", i);
        for (uint64_t j = c.cmp_off[i]; j < c.cmp_off[i + 1]; j++) {
            uint32_t a = c.cmp_aln[j], h = c.hit[a];
            fprintf(fp, "%u-%u:%u:%c:%u-%u\t", c.q_start[a], c.q_end[a], r.d->t_id[h], r.d->is_rev[h] ? '-' : '+', c.t_start[a], c.t_end[a]);
        }
        fprintf(fp, "\n");
    }
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
// the arc list, while the cleaning passes go on; run_graph returns when all are on disk
struct GfaWriters {
    std::vector<std::thread> th;
    void start(const Run& r, const char* name) {
        if (r.out_dir.empty()) return;
        auto arcs = std::make_shared<std::vector<std::pair<uint32_t, uint32_t>>>(graph_arc_list(r.g));
        const Dataset* d = r.d;
        const std::string path = r.path(name);
        th.emplace_back([arcs, d, path]() { graph_write_gfa_arcs(*arcs, *d, path); });
    }
    ~GfaWriters() { for (auto& t : th) t.join(); }
};

static int run_graph(Run& r) {
    double t0 = now();
    GfaWriters gfa;
    if (r.have_edges) r.be.free_edges(r.be.ctx, &r.edges), r.have_edges = false;
    if (r.be.edge_support(r.be.ctx, &r.prm, &r.edges) != 0) return backend_fail(r, "edge_support");
    r.have_edges = true;
    const Dataset& d = *r.d;
    Graph& g = r.g;
    graph_build(g, (uint32_t)d.contig_len.size(), r.edges);
    graph_write_stats(g, d, r.path("backbone.01.init.stat"));
    gfa.start(r, "backbone.01.init.gfa");
    int nb = graph_remove_weak_edges(g, r.prm.min_edge_sup);
    if (!r.out_dir.empty()) fprintf(stderr, "       removed %d edges\n", nb);
    graph_write_stats(g, d, r.path("backbone.02.weakEdge.stat"));
    gfa.start(r, "backbone.02.weakEdge.gfa");
    nb = clean_tips(g, 1, r.path("backbone.03.tip.log"));
    nb += clean_tips(g, 2, r.path("backbone.03.tip.log"));
    nb += clean_tips(g, 3, r.path("backbone.03.tip.log"));
    if (!r.out_dir.empty()) fprintf(stderr, "       removed %d tips\n", nb);
    graph_write_stats(g, d, r.path("backbone.03.tip.stat"));
    gfa.start(r, "backbone.03.tip.gfa");
    nb = clean_simple_bubbles(g, 4, r.path("backbone.04.simplebubble.log"));
    if (!r.out_dir.empty()) fprintf(stderr, "       removed %d simple bubbles\n", nb);
    graph_write_stats(g, d, r.path("backbone.04.simplebubble.stat"));
    gfa.start(r, "backbone.04.simplebubble.gfa");
    nb = clean_super_bubbles(g, r.path("backbone.05.superbubble.log"));
    if (!r.out_dir.empty()) fprintf(stderr, "       removed %d super bubbles\n", nb);
    graph_write_stats(g, d, r.path("backbone.05.superbubble.stat"));
    gfa.start(r, "backbone.05.superbubble.gfa");
    nb = clean_small_bubbles(g, r.path("backbone.06.smallbubble.log"));
    if (!r.out_dir.empty()) fprintf(stderr, "       removed %d small bubbles\n", nb);
    graph_write_stats(g, d, r.path("backbone.06.smallbubble.stat"));
    gfa.start(r, "backbone.06.smallbubble.gfa");
    graph_report_branching(g, r.path("backbone.branching.log"));
    for (auto& t : gfa.th) t.join();
    gfa.th.clear();
    r.t[1] = now() - t0;
    return 0;
}

static int run_coords(Run& r) {
    double t0 = now();
    r.work = work_queue(r.g, 11);
    if (r.shard_world > 1) {   // deal the queue out by descending support count (a cost proxy), round-robin
        std::vector<uint32_t> ord(r.work.size());
        for (uint32_t i = 0; i < ord.size(); i++) ord[i] = i;
        std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
            return r.g.find(r.work[a].first, r.work[a].second)->supp > r.g.find(r.work[b].first, r.work[b].second)->supp;
        });
        std::vector<char> mine(r.work.size(), 0);
        for (uint32_t k = 0; k < ord.size(); k++) if (k % r.shard_world == r.shard_rank) mine[ord[k]] = 1;
        std::vector<std::pair<uint32_t, uint32_t>> w;
        for (uint32_t i = 0; i < r.work.size(); i++) if (mine[i]) w.push_back(r.work[i]);
        r.work.swap(w);
    }
    std::vector<uint32_t> sel(r.work.size());
    for (size_t i = 0; i < r.work.size(); i++) sel[i] = r.g.find(r.work[i].first, r.work[i].second)->dev_edge;
    if (r.have_coords) r.be.free_coords(r.be.ctx, &r.coords), r.have_coords = false;
    if (r.be.edge_coords(r.be.ctx, (uint32_t)sel.size(), sel.data(), &r.coords) != 0) return backend_fail(r, "edge_coords");
    r.have_coords = true;
    FILE* fp = open_or_null(r.path("log_coordinate.txt"), "w");
    for (size_t i = 0; i < r.work.size(); i++) {
        uint32_t v = r.work[i].first, key = r.work[i].second;
        Arc* a = r.g.find(v, key);
        Arc* tw = r.g.find(Graph::twin_vertex(key), Graph::twin_key(v));
        uint32_t n = (uint32_t)(r.coords.supp_off[i + 1] - r.coords.supp_off[i]);
        a->head_end = r.coords.head_end[i]; a->tail_beg = r.coords.tail_beg[i]; a->n_cns_supp = n;
        if (tw == a) a->head_end = a->tail_beg = r.coords.tail_beg[i];   // self-twin (hairpin): edge1 and edge2 are one object in the
                                                                          // reference, the second chained assignment wins (Assemble.cpp:351-352)
        else if (tw) { tw->tail_beg = r.coords.head_end[i]; tw->head_end = r.coords.tail_beg[i]; tw->n_cns_supp = n; }
        if (fp) {   // reduced form of the reference's diagnostic log (Assemble.cpp:176-362)
            fprintf(fp, "edge      %u:%c -> %u:%c\n", v >> 1, "+-"[v & 1], key >> 1, "+-"[key & 1]);
            fprintf(fp, "edge_twin %u:%c -> %u:%c\n", key >> 1, "+-"[1 - (key & 1)], v >> 1, "+-"[1 - (v & 1)]);
            fprintf(fp, "\tedge_supp size:%u\n", a->supp);
            fprintf(fp, "coordinates contig1_pos: %u\tcontig2_pos: %u\n", a->head_end, a->tail_beg);
            for (uint64_t k = r.coords.supp_off[i]; k < r.coords.supp_off[i + 1]; k++)
                fprintf(fp, "    +++ lr:%u strand:%c [coordinate] lr_start:%u lr_end:%u\n", r.coords.supp_lr[k] & 0x7fffffffu,
                        "+-"[r.coords.supp_lr[k] >> 31], r.coords.spos[k], r.coords.epos[k]);
            fprintf(fp, "\n");
        }
    }
    if (fp) fclose(fp);
    r.t[2] = now() - t0;
    return 0;
}

static int run_consensus(Run& r) {
    double t0 = now();
    // second pass of the work queue (flag 12) hands out the same arcs in the same order as the first
    std::vector<std::pair<uint32_t, uint32_t>> again = work_queue(r.g, 12);
    if (r.shard_world == 1 && again != r.work) { g_err = "internal: consensus work queue differs from coordinate work queue"; return -1; }
    hx_poa_params pp{5, -4, -8};   // Assemble.cpp:8-11
    if (r.have_cns) r.be.free_cns(r.be.ctx, &r.cnsout), r.have_cns = false;
    if (r.be.poa_batch(r.be.ctx, &pp, &r.cnsout) != 0) return backend_fail(r, "poa_batch");
    r.have_cns = true;
    r.cns.clear();
    r.cns.reserve(r.work.size() * 2);
    FILE* fp = open_or_null(r.path("log_consensus.txt"), "w");
    for (size_t i = 0; i < r.work.size(); i++) {
        uint32_t v = r.work[i].first, key = r.work[i].second;
        std::string c(r.cnsout.cns + r.cnsout.cns_off[i], r.cnsout.cns + r.cnsout.cns_off[i + 1]);
        Arc* a = r.g.find(v, key);
        Arc* tw = r.g.find(Graph::twin_vertex(key), Graph::twin_key(v));
        a->cns_id = (int32_t)r.cns.size();
        r.cns.push_back(c);
        if (tw && tw != a) { tw->cns_id = (int32_t)r.cns.size(); r.cns.push_back(revcomp(c)); }   // Assemble.cpp:555
        else if (tw == a) { r.cns.back() = revcomp(c); }   // self-twin (hairpin): the reference's edge2 assignment overwrites edge1's
        if (fp) {
            fprintf(fp, "calc_cns %u:%c -> %u:%c\n", v >> 1, "+-"[v & 1], key >> 1, "+-"[key & 1]);
            fprintf(fp, "[shared_region] head_end:%u\ttail_beg:%u\n", a->head_end, a->tail_beg);
            fprintf(fp, ">CONSENSUS\n%s\n", c.c_str());
        }
    }
    if (fp) fclose(fp);
    r.t[3] = now() - t0;
    return 0;
}

namespace {
struct PE { uint32_t strand, id; };

void assemble_path(Run& r, const std::deque<PE>& path, int& nb_ctg, FILE* fp_asm, FILE* fp_ann, FILE* fp_log) {
    const Dataset& d = *r.d;
    Graph& g = r.g;
    auto emit = [&](uint32_t fs, uint32_t fstrand, uint32_t ts, uint32_t tstrand, const std::string& seq) {
        LOGF(fp_log, ">%d from:%u:%c to:%u:%c\n%s\n\n", nb_ctg, fs, "+-"[fstrand], ts, "+-"[tstrand], seq.c_str());
        char hdr[128];
        int n = snprintf(hdr, sizeof(hdr), ">%d from:%u:%c to:%u:%c\n", nb_ctg, fs, "+-"[fstrand], ts, "+-"[tstrand]);
        r.fasta.append(hdr, n); r.fasta += seq; r.fasta.push_back('\n');
        if (fp_asm) { fputs(hdr, fp_asm); fputs(seq.c_str(), fp_asm); fputc('\n', fp_asm); }
        nb_ctg++;
    };
    if (path.size() == 1) {
        emit(path.front().id, path.front().strand, path.front().id, path.front().strand, d.contig_seq(path.front().id));
        return;
    }
    std::string assembled;
    uint32_t src = path[0].id, src_strand = path[0].strand;
    uint32_t c1_start = src_strand == 0 ? 0 : d.contig_len[src] - 1;
    uint32_t tgt = path.back().id, tgt_strand = path.back().strand;
    size_t i;
    for (i = 0; i + 1 < path.size(); i++) {
        uint32_t c1 = path[i].id, s1 = path[i].strand, c2 = path[i + 1].id, s2 = path[i + 1].strand;
        std::string c1s = d.contig_seq(c1);
        Arc* e = g.find((c1 << 1) | s1, (c2 << 1) | s2);
        std::string prefix;
        if (e->n_cns_supp == 0) {   // break the assembly here (Assemble.cpp:682-706)
            LOGF(fp_log, "[breaking] contig1_len:%zu    contig1_start:%u    prev_end:%u     next_beg:%u\n", c1s.size(), c1_start, e->head_end, e->tail_beg);
            if (s1 == 0) {
                prefix = c1s.substr(c1_start);
                LOGF(fp_ann, "%d\t%zu\t%zu\tctg\t+\t%u\t%zu\t%u\t%zu\n", nb_ctg, assembled.size(), assembled.size() + prefix.size(), c1, c1s.size(), c1_start, c1s.size());
            } else {
                prefix = c1s.substr(0, c1_start + 1);
                LOGF(fp_ann, "%d\t%zu\t%zu\tctg\t-\t%u\t%zu\t%u\t%u\n", nb_ctg, assembled.size(), assembled.size() + prefix.size(), c1, c1s.size(), 0, c1_start + 1);
                prefix = revcomp(prefix);
            }
            assembled += prefix;
            emit(src, src_strand, c1, s1, assembled);
            assembled.clear();
            src = c2; src_strand = s2;
            c1_start = src_strand == 0 ? 0 : d.contig_len[src] - 1;
            if (!r.out_dir.empty())
                fprintf(stderr, "[WARNING] breaking assembly for path %u:%c --> %u:%c between anchors %u:%c --> %u:%c\n", src, "+-"[src_strand], tgt, "+-"[tgt_strand], c1, "+-"[s1], c2, "+-"[s2]);
        } else {                    // stitch contig piece + consensus (Assemble.cpp:708-731)
            LOGF(fp_log, "[stitching] contig1_len:%zu    contig1_start:%u    prev_end:%u     next_beg:%u\n", c1s.size(), c1_start, e->head_end, e->tail_beg);
            if (s1 == 0) {
                prefix = c1s.substr(c1_start, e->head_end - c1_start + 1);
                LOGF(fp_ann, "%d\t%zu\t%zu\tctg\t+\t%u\t%zu\t%u\t%zu\n", nb_ctg, assembled.size(), assembled.size() + prefix.size(), c1, c1s.size(), c1_start, c1_start + prefix.size());
            } else {
                prefix = c1s.substr(e->head_end, c1_start - e->head_end + 1);
                LOGF(fp_ann, "%d\t%zu\t%zu\tctg\t-\t%u\t%zu\t%u\t%zu\n", nb_ctg, assembled.size(), assembled.size() + prefix.size(), c1, c1s.size(), e->head_end, e->head_end + prefix.size());
                prefix = revcomp(prefix);
            }
            assembled += prefix;
            const std::string& cs = r.cns[e->cns_id];
            LOGF(fp_ann, "%d\t%zu\t%zu\tcns\t%zu\t%u\n", nb_ctg, assembled.size(), assembled.size() + cs.size(), cs.size(), e->n_cns_supp);
            assembled += cs;
            c1_start = e->tail_beg;
        }
    }
    uint32_t c2 = path[i].id, s2 = path[i].strand;
    std::string c2s = d.contig_seq(c2), suffix;
    if (s2 == 0) {
        suffix = c2s.substr(c1_start);
        LOGF(fp_ann, "%d\t%zu\t%zu\tctg\t+\t%u\t%zu\t%u\t%zu\n", nb_ctg, assembled.size(), assembled.size() + suffix.size(), c2, c2s.size(), c1_start, c2s.size());
    } else {
        suffix = c2s.substr(0, c1_start + 1);
        LOGF(fp_ann, "%d\t%zu\t%zu\tctg\t-\t%u\t%zu\t%u\t%u\n", nb_ctg, assembled.size(), assembled.size() + suffix.size(), c2, c2s.size(), 0, c1_start + 1);
        suffix = revcomp(suffix);
    }
    assembled += suffix;
    emit(src, src_strand, c2, s2, assembled);
}
}  // namespace

static int run_assemble(Run& r) {
    double t0 = now();
    Graph& g = r.g;
    const uint8_t mark = 21;
    std::vector<std::deque<PE>> paths;
    for (uint32_t i = 0; i < g.n_nodes; i++) {
        if (g.deg(i, 0) == 1 && g.deg(i, 1) == 1) continue;
        if (g.deg(i, 0) > 1 && g.deg(i, 1) > 1) paths.push_back({PE{0, i}});
        for (uint32_t side = 0; side < 2; side++) {
            for (size_t k = 0; k < g.adj[(i << 1) | side].size(); k++) {
                if (g.adj[(i << 1) | side][k].flag == mark) continue;
                std::deque<PE> p;
                p.push_back({side, i});
                const Arc* it = &g.adj[(i << 1) | side][k];
                uint32_t cn = it->key >> 1, cs = it->key & 1;
                for (;;) {   // asm_find_simple_path_from_source (Assemble.cpp:607-622)
                    p.push_back({cs, cn});
                    if (g.deg(cn, cs) == 0) break;
                    if (g.deg(cn, cs) > 1 || g.deg(cn, 1 - cs) > 1) break;
                    it = &g.adj[(cn << 1) | cs][0];
                    cn = it->key >> 1; cs = it->key & 1;
                }
                for (size_t j = 0; j + 1 < p.size(); j++) {
                    uint32_t v = (p[j].id << 1) | p[j].strand, key = (p[j + 1].id << 1) | p[j + 1].strand;
                    if (Arc* a = g.find(v, key)) a->flag = mark;
                    if (Arc* tw = g.find(Graph::twin_vertex(key), Graph::twin_key(v))) tw->flag = mark;
                }
                if (g.deg(p.front().id, p.front().strand) > 1) p.pop_front();
                if (!p.empty() && g.deg(p.back().id, 1 - p.back().strand) > 1) p.pop_back();
                if (!p.empty()) paths.push_back(p);
            }
        }
    }
    FILE* fp_asm = open_or_null(r.path("asm.final.fa"), "w");
    FILE* fp_ann = open_or_null(r.path("asm.final.ann"), "w");
    FILE* fp_log = open_or_null(r.path("log_asmfinal.txt"), "w");
    for (size_t i = 0; i < paths.size(); i++)
        LOGF(fp_log, "simple_path %zu size:%zu\tfrom:%u:%c\tto:%u:%c\n", i, paths[i].size(), paths[i].front().id, "+-"[paths[i].front().strand], paths[i].back().id, "+-"[paths[i].back().strand]);
    r.fasta.clear();
    int nb_ctg = 0;
    for (auto& p : paths) assemble_path(r, p, nb_ctg, fp_asm, fp_ann, fp_log);
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
extern "C" int hxh_run_all(hxh_run* p) {
    Run& r = *reinterpret_cast<Run*>(p);
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
extern "C" int hxh_run_write_longread_index(const hxh_run* p, const char* path) {
    const Run* r = reinterpret_cast<const Run*>(p);
    if (!r->have_chain) { g_err = "index.longread needs the chain stage"; return -1; }
    return write_longread_index(*r->d, r->chain, path) ? 0 : -1;
}
extern "C" void hxh_run_timings(const hxh_run* p, double out[5]) { memcpy(out, reinterpret_cast<const Run*>(p)->t, sizeof(double) * 5); }
extern "C" uint32_t hxh_run_n_edges(const hxh_run* p) { return (uint32_t)reinterpret_cast<const Run*>(p)->work.size(); }
extern "C" const hx_chain_out* hxh_run_chain_out(const hxh_run* p) { return &reinterpret_cast<const Run*>(p)->chain; }
extern "C" const hx_edges_out* hxh_run_edges_out(const hxh_run* p) { return &reinterpret_cast<const Run*>(p)->edges; }
extern "C" const hx_coords_out* hxh_run_coords_out(const hxh_run* p) { return &reinterpret_cast<const Run*>(p)->coords; }
extern "C" const hx_cns_out* hxh_run_cns_out(const hxh_run* p) { return &reinterpret_cast<const Run*>(p)->cnsout; }
extern "C" const char* hxh_run_assembly_fasta(const hxh_run* p, uint64_t* len) {
    const Run* r = reinterpret_cast<const Run*>(p);
    if (len) *len = r->fasta.size();
    return r->fasta.c_str();
}
