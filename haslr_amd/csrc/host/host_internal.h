// host_internal.h — C++ internals of the host side (not part of any ABI).
#ifndef HASLR_HOST_INTERNAL_H
#define HASLR_HOST_INTERNAL_H
#include <cstdint>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "haslr_host.h"

namespace hxh {

extern thread_local std::string g_err;

// Allocator for the big arenas (CIGAR op words, packed reads): resize() leaves new elements uninitialised, so growing an arena by
// gigabytes neither zero-fills it on one thread nor touches its pages before the worker threads write them (first touch in parallel).
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
    template <class U> void construct(U* p) noexcept { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
using U32Arena = std::vector<uint32_t, NoInitAlloc<uint32_t>>;
using U8Arena = std::vector<uint8_t, NoInitAlloc<uint8_t>>;

extern unsigned g_io_threads;   // threads of the loaders and the index writers (set by load_dataset: -t / HASLR_IO_THREADS / up to 16)
template <class F> void run_parallel(unsigned threads, F&& f) {   // f(thread index) on `threads` threads, the caller being one of them
    std::vector<std::thread> th;
    for (unsigned t = 1; t < threads; t++) th.emplace_back([&f, t]() { f(t); });
    f(0);
    for (auto& x : th) x.join();
}

struct Dataset {
    // contigs
    std::vector<uint32_t> contig_len, contig_kc;
    std::vector<double> contig_km;
    std::vector<uint64_t> contig_off;
    std::vector<uint8_t> contig_packed;
    double uniq_freq = 0;
    // reads
    std::vector<uint32_t> read_len;
    std::vector<uint64_t> read_off;
    U8Arena read_packed;
    uint64_t total_read_bases = 0;
    // raw PAF records
    std::vector<uint32_t> q_id, q_start, q_end, t_id, t_len, t_start, t_end, n_match, n_block;
    std::vector<uint8_t> is_rev, mapq;
    std::vector<uint64_t> cg_off;
    U32Arena cg_ops;
    std::vector<uint64_t> read_hit_off;
    std::unordered_map<uint64_t, std::string> cg_text_odd;   // cg:Z: text of the records whose op words do not spell it (index.longread keeps the text)

    std::string contig_seq(uint32_t id) const {
        std::string s(contig_len[id], 'A');
        const uint8_t* p = contig_packed.data() + contig_off[id];
        for (uint32_t i = 0; i < contig_len[id]; i++) s[i] = "ACGT"[(p[i >> 2] >> ((i & 3) * 2)) & 3];
        return s;
    }
};

Dataset* load_dataset(const char* contig_path, const char* long_path, bool long_fofn, const char* mapping_path, bool mapping_fofn, unsigned threads /* 0 = automatic, 1 = streaming readers only */);
// the same, but `index.contig` / `index.longread` of index_dir are loaded instead of the text files when they exist (main.cpp:39-103)
Dataset* load_dataset_cached(const char* index_dir, const char* contig_path, const char* long_path, bool long_fofn, const char* mapping_path, bool mapping_fofn,
                             unsigned threads, int* used_contig_index, int* used_longread_index);
void finish_contigs(Dataset& d);
bool append_cigar(Dataset& d, const char* b, const char* e);
bool parse_cigar_ops(const char* b, const char* e, U32Arena& ops, bool& odd, bool& too_long);   // cg:Z: text -> op words (io.cpp)
std::string cigar_text(const Dataset& d, uint64_t rec);
void append_cigar_text(const Dataset& d, uint64_t rec, std::string& out);
bool write_contig_index(const Dataset& d, const std::string& path);
bool write_longread_index(const Dataset& d, const hx_chain_out& chain, const std::string& path);
bool read_contig_index(Dataset& d, const std::string& path);
bool read_longread_index(Dataset& d, const std::string& path);

// ---------------------------------------------------------------------------------------------
// Backbone graph. The reference keeps `vector<BBG_Node_t>` with two std::map<uint32_t,BBG_Edge_t>
// per node (Backbone_graph.hpp:39-54). Here: one sorted arc vector per vertex (vertex = node*2 + side,
// side 0 = leaving through the contig's 3' end), which iterates in the same ascending-key order.
// Every undirected edge is two arcs (the arc and its twin), exactly as in the reference.
// ---------------------------------------------------------------------------------------------
struct Arc {
    uint32_t key;        // (to_node << 1) | to_rev — the reference's map key
    uint32_t supp;       // edge_supp.size()
    uint32_t dev_edge;   // index of this directed edge in hx_edges_out
    uint8_t flag = 0;    // traversal marker (Assemble.cpp:365-434: 11, 12, 21)
    uint32_t head_end = 0, tail_beg = 0;
    uint32_t n_cns_supp = 0;
    int32_t cns_id = -1; // index into Run::cns (consensus in THIS arc's direction), -1 = none yet
};

struct Graph {
    std::vector<std::vector<Arc>> adj;   // 2 * n_nodes
    uint32_t n_nodes = 0;

    static uint32_t twin_vertex(uint32_t key) { return key ^ 1u; }                 // (node2<<1)|(1-rev2)
    static uint32_t twin_key(uint32_t vertex) { return vertex ^ 1u; }              // (node1<<1)|(1-rev1)
    Arc* find(uint32_t v, uint32_t key);
    const Arc* find(uint32_t v, uint32_t key) const { return const_cast<Graph*>(this)->find(v, key); }
    void erase_arc(uint32_t v, uint32_t key);
    // nodes whose adjacency changed while a cleaning pass runs (a bit per node; null outside a pass): the pass looks at them again
    std::vector<uint64_t>* touched = nullptr;
    // bbg_remove_edge (Backbone_graph.cpp:45-51)
    void remove_edge(uint32_t node1, uint32_t rev1, uint32_t node2, uint32_t rev2) {
        erase_arc((node1 << 1) | rev1, (node2 << 1) | rev2);
        erase_arc((node2 << 1) | (1 - rev2), (node1 << 1) | (1 - rev1));
        if (touched) { (*touched)[node1 >> 6] |= 1ull << (node1 & 63); (*touched)[node2 >> 6] |= 1ull << (node2 & 63); }
    }
    size_t deg(uint32_t node, uint32_t side) const { return adj[(node << 1) | side].size(); }
};

void graph_build(Graph& g, uint32_t n_nodes, const hx_edges_out& e);
int graph_remove_weak_edges(Graph& g, uint32_t min_edge_sup);
void graph_write_stats(const Graph& g, const Dataset& d, const std::string& path);
void graph_write_gfa(const Graph& g, const Dataset& d, const std::string& path);
std::vector<std::pair<uint32_t, uint32_t>> graph_arc_list(const Graph& g);
void graph_write_gfa_arcs(const std::vector<std::pair<uint32_t, uint32_t>>& arcs, const Dataset& d, const std::string& path);
void graph_report_branching(const Graph& g, const std::string& path);
unsigned clean_threads(uint32_t n_nodes);   // host threads the candidate scans of the cleaning passes use (HASLR_CLEAN_THREADS; 1 below 200 000 nodes)
int clean_tips(Graph& g, int max_depth, const std::string& logpath);
int clean_simple_bubbles(Graph& g, int max_depth, const std::string& logpath);
int clean_super_bubbles(Graph& g, const std::string& logpath);
int clean_small_bubbles(Graph& g, const std::string& logpath);

std::string revcomp(const std::string& s);
FILE* open_or_null(const std::string& path, const char* mode);   // "" -> nullptr (no file output)

}  // namespace hxh
#endif
