// oracle_backend.cpp — TEST INFRASTRUCTURE: wraps the oracle's four operators in the `hx_backend`
// table of haslr_amd/csrc/host/haslr_host.h so that tests and bench.py's cpu_baseline leg can run
// the host pipeline with the CPU restatement in place of the HIP C-ABI. The product never loads this.
#include <cstdlib>
#include <cstring>

#include "../haslr_amd/csrc/host/haslr_host.h"
#include "oracle.h"

struct orc_ctx {
    hx_contigs contigs;
    hx_reads reads;
    hx_hits hits;
    const uint64_t* read_hit_off;
    int n_threads;
    hx_chain_out chain;
    hx_edges_out edges;
    hx_coords_out coords;
    bool have_chain, have_edges, have_coords;
    uint32_t lr_begin, lr_end;   // read shard shown to the host pipeline by chain_reads (multi-rank tests); 0, 0 = all reads
    hx_chain_out full_chain;     // the chain output of ALL reads, which the later stages work on (kept when a shard is shown)
    bool sliced;
    int prefiltered;             // the records come from an index.longread
};

template <class T> static T* dup_range(const T* p, uint64_t b, uint64_t e) {
    T* q = (T*)malloc((e - b + 1) * sizeof(T));
    memcpy(q, p + b, (e - b) * sizeof(T));
    return q;
}
// the part of a chain output that belongs to reads [b, e), with offsets and alignment indices rebased: what a rank that chained only
// those reads would hold
static void slice_chain(const hx_chain_out& f, uint32_t b, uint32_t e, hx_chain_out* o) {
    const uint64_t a0 = f.read_off[b], a1 = f.read_off[e], c0 = f.cmp_off[b], c1 = f.cmp_off[e];
    memset(o, 0, sizeof(*o));
    o->n_aln = a1 - a0; o->n_reads = e - b; o->n_cmp = c1 - c0;
    o->hit = dup_range(f.hit, a0, a1); o->q_start = dup_range(f.q_start, a0, a1); o->q_end = dup_range(f.q_end, a0, a1);
    o->t_start = dup_range(f.t_start, a0, a1); o->t_end = dup_range(f.t_end, a0, a1); o->n_match = dup_range(f.n_match, a0, a1);
    o->n_block = dup_range(f.n_block, a0, a1); o->cg_begin = dup_range(f.cg_begin, a0, a1); o->cg_end = dup_range(f.cg_end, a0, a1);
    o->cg_skip_front = dup_range(f.cg_skip_front, a0, a1); o->cg_skip_back = dup_range(f.cg_skip_back, a0, a1);
    o->read_off = dup_range(f.read_off, b, (uint64_t)e + 1); o->cmp_off = dup_range(f.cmp_off, b, (uint64_t)e + 1);
    o->cmp_aln = dup_range(f.cmp_aln, c0, c1);
    for (uint32_t i = 0; i <= e - b; i++) { o->read_off[i] -= a0; o->cmp_off[i] -= c0; }
    for (uint64_t i = 0; i < c1 - c0; i++) o->cmp_aln[i] -= (uint32_t)a0;
}

static int be_chain(void* p, const hx_params* prm, hx_chain_out* out) {
    orc_ctx* c = (orc_ctx*)p;
    int rc = orc_chain_reads_ex(&c->contigs, &c->hits, c->read_hit_off, c->reads.n, prm, c->prefiltered, out);
    if (rc) return rc;
    c->chain = *out; c->have_chain = true; c->sliced = false;
    if (c->lr_end > c->lr_begin || c->lr_begin) {   // show the host pipeline this rank's reads only; edges / coords go on using the full table
        c->full_chain = *out; c->sliced = true;
        slice_chain(c->full_chain, c->lr_begin, c->lr_end, out);
    }
    return 0;
}
static int be_edges(void* p, const hx_params* prm, hx_edges_out* out) {
    orc_ctx* c = (orc_ctx*)p;
    if (!c->have_chain) return -1;
    int rc = orc_edge_support(&c->contigs, &c->hits, prm, &c->chain, 0, c->reads.n, out);
    if (rc == 0) { c->edges = *out; c->have_edges = true; }
    return rc;
}
static int be_coords(void* p, uint32_t n_sel, const uint32_t* sel, hx_coords_out* out) {
    orc_ctx* c = (orc_ctx*)p;
    if (!c->have_edges) return -1;
    int rc = orc_edge_coords(&c->contigs, c->reads.len, &c->hits, &c->edges, n_sel, sel, out);
    if (rc == 0) { c->coords = *out; c->have_coords = true; }
    return rc;
}
static int be_poa(void* p, const hx_poa_params* pp, hx_cns_out* out) {
    orc_ctx* c = (orc_ctx*)p;
    if (!c->have_coords) return -1;
    return orc_poa_batch(&c->reads, &c->coords, pp, c->n_threads, out);
}
static void be_free_chain(void* p, hx_chain_out* o) {
    orc_ctx* c = (orc_ctx*)p;
    c->have_chain = false;
    if (c->sliced) { orc_free_chain(&c->full_chain); c->sliced = false; }
    orc_free_chain(o);
}
static void be_free_edges(void* p, hx_edges_out* o) { ((orc_ctx*)p)->have_edges = false; orc_free_edges(o); }
static void be_free_coords(void* p, hx_coords_out* o) { ((orc_ctx*)p)->have_coords = false; orc_free_coords(o); }
static void be_free_cns(void*, hx_cns_out* o) { orc_free_cns(o); }

extern "C" orc_ctx* orc_ctx_create(const hx_contigs* c, const hx_reads* r, const hx_hits* h, const uint64_t* rho, int n_threads) {
    orc_ctx* x = (orc_ctx*)calloc(1, sizeof(orc_ctx));
    x->contigs = *c; x->reads = *r; x->hits = *h; x->read_hit_off = rho; x->n_threads = n_threads;
    return x;
}
extern "C" void orc_ctx_destroy(orc_ctx* x) { free(x); }
extern "C" void orc_ctx_set_prefiltered(orc_ctx* x, int on) { x->prefiltered = on; }
extern "C" void orc_ctx_set_read_shard(orc_ctx* x, uint32_t b, uint32_t e) { x->lr_begin = b; x->lr_end = e; }
/* the emission of one read shard (multi-rank tests): records of reads [b, e) from the full chain table, key-sorted like orc_edge_support returns them */
extern "C" int orc_ctx_shard_edges(orc_ctx* c, const hx_params* prm, uint32_t b, uint32_t e, hx_edges_out* out) {
    if (!c->have_chain) return -1;
    return orc_edge_support(&c->contigs, &c->hits, prm, &c->chain, b, e, out);
}
extern "C" void orc_backend_fill(orc_ctx* x, hx_backend* b) {
    b->ctx = x; b->chain_reads = be_chain; b->edge_support = be_edges; b->edge_coords = be_coords; b->poa_batch = be_poa;
    b->free_chain = be_free_chain; b->free_edges = be_free_edges; b->free_coords = be_free_coords; b->free_cns = be_free_cns;
    b->last_error = orc_last_error;
}
