/* haslr_host.h — host side of the stage (C++ implementation, C interface so that the
 * Python mirror in haslr_amd/ and the tests can drive it through ctypes).
 *
 * This is the part of haslr_assemble that stays on the CPU by design:
 * text ingest, the order-dependent serial graph cleaning, path stitching and the writers whose
 * bytes are the parity gate. All per-read and per-edge arithmetic goes through the `hx_backend`
 * table, which the product fills with the HIP entry points of include/haslr_hip.h.
 */
#ifndef HASLR_HOST_H
#define HASLR_HOST_H
#include "haslr_types.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hxh_dataset hxh_dataset; /* parsed inputs, resident in host memory */
typedef struct hxh_graph hxh_graph;     /* backbone graph */
typedef struct hxh_run hxh_run;         /* one execution of the stage: dataset + params + backend + state */

const char* hxh_last_error(void);

/* ---- ingest (Contig.cpp:43-117, Longread.cpp:109-162, :234-302; kseq-compatible FASTA/FASTQ[.gz]) */
hxh_dataset* hxh_dataset_load(const char* contig_path, const char* long_path, int long_fofn,
                              const char* mapping_path, int mapping_fofn);
/* the same with an explicit number of ingest threads (0 = automatic: HASLR_IO_THREADS or up to 16; 1 = the streaming single-thread
 * readers). Plain FASTA and PAF files are mapped, cut at record boundaries and parsed in parallel; gzip, FASTQ and small files stream.
 * The arrays are identical for every thread count. */
hxh_dataset* hxh_dataset_load_mt(const char* contig_path, const char* long_path, int long_fofn,
                                 const char* mapping_path, int mapping_fofn, unsigned threads);
/* index caches of the reference (index.contig / index.longread, main.cpp:39-103; layouts in host/index_cache.cpp): when index_dir holds
 * them they are loaded instead of the text files (used_* report which); byte-compatible with the reference's own files both ways */
hxh_dataset* hxh_dataset_load_cached(const char* index_dir, const char* contig_path, const char* long_path, int long_fofn, const char* mapping_path,
                                     int mapping_fofn, unsigned threads, int* used_contig_index, int* used_longread_index);
int hxh_dataset_write_contig_index(const hxh_dataset*, const char* path);
void hxh_dataset_free(hxh_dataset*);
void hxh_dataset_views(const hxh_dataset*, hx_contigs*, hx_reads*, hx_hits*, const uint64_t** read_hit_off);
double hxh_dataset_uniq_freq(const hxh_dataset*);   /* Contig.cpp:162-174 */
uint64_t hxh_dataset_total_read_bases(const hxh_dataset*);
/* contig sequence as ASCII into caller buffer of contig length bytes */
void hxh_dataset_contig_seq(const hxh_dataset*, uint32_t id, char* dst);

/* ---- compute backend: the four hot-path operators. ctx is opaque to the host pipeline.
 * The signatures are exactly the C-ABI of include/haslr_hip.h. */
typedef struct {
    void* ctx;
    int (*chain_reads)(void* ctx, const hx_params*, hx_chain_out*);
    int (*edge_support)(void* ctx, const hx_params*, hx_edges_out*);
    int (*edge_coords)(void* ctx, uint32_t n_sel, const uint32_t* sel_edge, hx_coords_out*);
    int (*poa_batch)(void* ctx, const hx_poa_params*, hx_cns_out*);
    void (*free_chain)(void* ctx, hx_chain_out*);
    void (*free_edges)(void* ctx, hx_edges_out*);
    void (*free_coords)(void* ctx, hx_coords_out*);
    void (*free_cns)(void* ctx, hx_cns_out*);
    const char* (*last_error)(void);
} hx_backend;

/* ---- one run of the stage (main.cpp:115-219). out_dir may be NULL: no files are written
 * (bench timing of the compute path); otherwise every reference output file is produced. */
hxh_run* hxh_run_create(const hxh_dataset*, const hx_params*, const hx_backend*, const char* out_dir);
void hxh_run_free(hxh_run*);
/* multi-GPU (one run per rank, every rank sees the same graph because hxh_run_graph works on the merged edge multiset):
 *   hxh_run_set_edge_shard   this run computes coordinates + consensus only for its share of the work queue (Assemble.cpp:386-434); the
 *                            queue is dealt by estimated DP cost, longest first, to the least loaded rank - the same on every rank
 *   hxh_run_set_read_shard   first long read of the backend's read shard (read ids of hxh_run_compact_text)
 *   hxh_run_results_export   head_end / tail_beg / supports / consensus of this run's share as one blob (valid until the next export)
 *   hxh_run_results_import   the other ranks' blobs, in any order, singly or concatenated (own entries are skipped)
 *   hxh_run_results_missing  queue entries without results: hxh_run_assemble fails unless it is 0 (the role of asm_cal_cns_seq_MT's
 *                            join before asm_get_assembly, Assemble.cpp:580-605 -> :1045-1077)
 *   hxh_run_compact_text     the compact_uniq.txt lines of the reads this run chained (a sharded run does not write the file itself) */
void hxh_run_set_edge_shard(hxh_run*, uint32_t rank, uint32_t world);
/* the same inside ONE process (one host thread per rank; the multi-GPU mode of the haslr_assemble binary):
 *   hxh_shard_bounds        n + 1 boundaries of contiguous read-id ranges with about equal numbers of raw PAF records
 *   hxh_runs_all_sharded    the whole stage over runs[0..n): runs[r] was created on rank r's backend table (include/haslr_hip.h
 *                           hx_group_backend_fill: its edge_support is the collective hx_edge_merge), runs[0] owns the output directory. One
 *                           thread per rank: chain (own reads) -> graph (merged multiset, cleaned redundantly) -> coordinates + consensus (own
 *                           share of the queue) -> results exchanged through the process's memory -> rank 0 writes compact_uniq.txt and the
 *                           assembly. The ranks agree on success after every stage: an error on one rank ends all of them, none waits
 *                           forever. on_stage (may be NULL) is called on rank 0's thread at the begin (1) / end (0) of stage 0..4
 *                           (chain, graph, coords, consensus, assemble). */
void hxh_shard_bounds(const hxh_dataset*, uint32_t n, uint32_t* bounds);
int hxh_runs_all_sharded(hxh_run** runs, uint32_t n, const uint32_t* read_begin, void (*on_stage)(int stage, int begin, void* user), void* user);
/* index.longread of a sharded pass (after hxh_runs_all_sharded, or at least the chain stage of every rank): the ranks' filtered alignments in
 * rank order = read order, same bytes as the single-rank file */
int hxh_runs_write_longread_index(hxh_run* const* runs, uint32_t n, const char* path);
void hxh_run_set_read_shard(hxh_run*, uint32_t lr_begin);
int hxh_run_results_export(hxh_run*, const uint8_t** buf, uint64_t* len);
int hxh_run_results_import(hxh_run*, const uint8_t* buf, uint64_t len);
uint64_t hxh_run_results_missing(const hxh_run*);
const char* hxh_run_compact_text(hxh_run*, uint64_t* len);
/* stages, in reference order; each returns 0 or <0 */
int hxh_run_chain(hxh_run*);          /* fix_alignments + build_compact_longreads (+ compact_uniq.txt) */
int hxh_run_graph(hxh_run*);          /* bbg_build_graph .. clean_small_bubbles + branching log (+ gfa/stat/log files) */
int hxh_run_coords(hxh_run*);         /* asm_calc_edge_coordinates_MT */
int hxh_run_consensus(hxh_run*);      /* asm_cal_cns_seq_MT */
int hxh_run_assemble(hxh_run*);       /* asm_get_assembly: asm.final.fa / .ann / log_asmfinal.txt */
int hxh_run_all(hxh_run*);            /* all of the above */
/* on: hxh_run_graph returns while its six GFA snapshots are still being written by their threads - beside the coordinate and consensus
 * stages, which need only the cleaned graph; hxh_run_assemble (and hxh_run_free) waits for them. Off (default): a graph stage called on
 * its own returns when the files are complete. hxh_run_all always overlaps. */
void hxh_run_set_async_writers(hxh_run*, int on);
/* wall seconds of the last call of each stage: chain, graph(host), coords, consensus, assemble */
/* index.longread: the alignments that survived the chain stage's filters, with their raw fields (needs hxh_run_chain) */
int hxh_run_write_longread_index(const hxh_run*, const char* path);
void hxh_run_timings(const hxh_run*, double out[5]);
/* results for tests: number of surviving undirected edges this run processed (= all of them unless sharded) / in total; their consensus */
uint32_t hxh_run_n_edges(const hxh_run*);
uint32_t hxh_run_n_edges_total(const hxh_run*);
const hx_chain_out* hxh_run_chain_out(const hxh_run*);
const hx_edges_out* hxh_run_edges_out(const hxh_run*);
const hx_coords_out* hxh_run_coords_out(const hxh_run*);
const hx_cns_out* hxh_run_cns_out(const hxh_run*);
/* the assembled contigs, FASTA text (same bytes as asm.final.fa); valid until the run is freed */
const char* hxh_run_assembly_fasta(const hxh_run*, uint64_t* len);

#ifdef __cplusplus
}
#endif
#endif
