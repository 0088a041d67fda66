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
};

static int be_chain(void* p, const hx_params* prm, hx_chain_out* out) {
    orc_ctx* c = (orc_ctx*)p;
    int rc = orc_chain_reads(&c->contigs, &c->hits, c->read_hit_off, c->reads.n, prm, out);
    if (rc == 0) { c->chain = *out; c->have_chain = true; }
    return rc;
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
static void be_free_chain(void* p, hx_chain_out* o) { ((orc_ctx*)p)->have_chain = false; orc_free_chain(o); }
static void be_free_edges(void* p, hx_edges_out* o) { ((orc_ctx*)p)->have_edges = false; orc_free_edges(o); }
static void be_free_coords(void* p, hx_coords_out* o) { ((orc_ctx*)p)->have_coords = false; orc_free_coords(o); }
static void be_free_cns(void*, hx_cns_out* o) { orc_free_cns(o); }

extern "C" orc_ctx* orc_ctx_create(const hx_contigs* c, const hx_reads* r, const hx_hits* h, const uint64_t* rho, int n_threads) {
    orc_ctx* x = (orc_ctx*)calloc(1, sizeof(orc_ctx));
    x->contigs = *c; x->reads = *r; x->hits = *h; x->read_hit_off = rho; x->n_threads = n_threads;
    return x;
}
extern "C" void orc_ctx_destroy(orc_ctx* x) { free(x); }
extern "C" void orc_backend_fill(orc_ctx* x, hx_backend* b) {
    b->ctx = x; b->chain_reads = be_chain; b->edge_support = be_edges; b->edge_coords = be_coords; b->poa_batch = be_poa;
    b->free_chain = be_free_chain; b->free_edges = be_free_edges; b->free_coords = be_free_coords; b->free_cns = be_free_cns;
    b->last_error = orc_last_error;
}
