/* oracle.h — TEST INFRASTRUCTURE. CPU restatement of the reference hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so.
 * The product (libhaslr_hip.so, haslr_assemble) never links, dlopens or executes anything here.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   orc_chain_reads, orc_edge_support : PINNED against the real reference's own front half
 *       (oracle/_ref/ref_front, built from /root/reference sources) — compact_uniq.txt,
 *       alignments.fixed.paf, edge_supp dumps, on seeded inputs; golden copies in tests/golden/.
 *   orc_edge_coords, orc_poa_batch    : PARITY UNPINNED. Assemble.cpp cannot be built here
 *       (needs spoa.hpp from rvaser/spoa 1.1.3, un-vendored, no network) and the reference has
 *       no tests or golden vectors. orc_edge_coords restates Assemble.cpp:24-363 line by line;
 *       orc_poa_batch restates the published algorithm of rvaser/spoa tag 1.1.3 (graph.cpp,
 *       sisd_alignment_engine.cpp) as called from Assemble.cpp:499-554.
 */
#ifndef HASLR_ORACLE_H
#define HASLR_ORACLE_H
#include "haslr_types.h"
#ifdef __cplusplus
extern "C" {
#endif

/* read_hit_off: n_reads+1 offsets into hits (PAF must be grouped by query in ascending read id,
 * the reference's precondition, Longread.cpp:57-84,253). Returns 0, or <0 with orc_last_error(). */
int orc_chain_reads(const hx_contigs* contigs, const hx_hits* hits, const uint64_t* read_hit_off,
                    uint32_t n_reads, const hx_params* prm, hx_chain_out* out);
/* prefiltered != 0: the records are the filtered set of an index.longread (no filters, sort or group rule: straight to trim + chain) */
int orc_chain_reads_ex(const hx_contigs* contigs, const hx_hits* hits, const uint64_t* read_hit_off,
                       uint32_t n_reads, const hx_params* prm, int prefiltered, hx_chain_out* out);
int orc_edge_support(const hx_contigs* contigs, const hx_hits* hits, const hx_params* prm,
                     const hx_chain_out* chain, uint32_t lr_begin, uint32_t lr_end, hx_edges_out* out);
int orc_edge_coords(const hx_contigs* contigs, const uint32_t* read_len, const hx_hits* hits,
                    const hx_edges_out* edges, uint32_t n_sel, const uint32_t* sel_edge, hx_coords_out* out);
int orc_poa_batch(const hx_reads* reads, const hx_coords_out* coords, const hx_poa_params* pp,
                  int n_threads, hx_cns_out* out);
void orc_free_chain(hx_chain_out*);
void orc_free_edges(hx_edges_out*);
void orc_free_coords(hx_coords_out*);
void orc_free_cns(hx_cns_out*);
const char* orc_last_error(void);
/* row kernels of the POA aligner in use: "avx2 (8 x int32)" when the CPU has AVX2 (same integers as the scalar restatement; ORC_POA_SCALAR=1 forces "scalar") */
const char* orc_poa_kernel_name(void);

/* single POA problem on plain ASCII sequences (known-answer tests): returns malloc'd consensus */
char* orc_poa_consensus(const char* const* seqs, uint32_t n, const hx_poa_params* pp);
void orc_free_str(char*);

#ifdef __cplusplus
}
#endif
#endif
