/* haslr_types.h — plain-C data layouts shared by the C-ABI (haslr_hip.h), the host
 * pipeline and the test oracle. Everything is structure-of-arrays: on the device each
 * array is one contiguous HBM allocation so that a wavefront's 64 lanes read 64
 * consecutive elements (coalesced), which the reference's 48-byte AoS `Align_Seq_t`
 * (Longread.hpp:32-48) does not allow.
 *
 * Reference types these replace (all under /root/reference/src/haslr_assemble/src):
 *   Align_Seq2_t / Align_Seq_t   Longread.hpp:16-48     -> hx_hits (raw) + hx_alns (surviving)
 *   Longread_t / Longread_List_t Longread.hpp:50-77     -> hx_reads
 *   Contig_t / Contig_List_t     Contig.hpp:14-32       -> hx_contigs
 *   Edge_Supp_t                  Backbone_graph.hpp:23-29 -> hx_edge_recs (key + lr + cmp ids [+ hit copies])
 *   Consensus_Supp_t             Backbone_graph.hpp:31-37 -> hx_coords (supp_lr/spos/epos)
 */
#ifndef HASLR_TYPES_H
#define HASLR_TYPES_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CIGAR op word: (length << 2) | code. Codes keep the distinctions the reference makes when it
 * walks a per-base expanded CIGAR (Longread.cpp:384-396 forward, :401-413 undo; Assemble.cpp:141-153):
 * 'M' consumes read+contig, 'I' read only, anything else contig only; but only a literal 'D' is undone. */
enum { HX_CG_M = 0, HX_CG_I = 1, HX_CG_D = 2, HX_CG_OTHER = 3 };
#define HX_CG_LEN(w) ((uint32_t)(w) >> 2)
#define HX_CG_OP(w) ((uint32_t)(w) & 3u)

/* Options of the stage (Common.hpp:44-65, defaults Commandline.cpp:46-66). */
typedef struct {
    uint32_t min_aln_block; /* --aln-block, 500 */
    double min_aln_sim;     /* --aln-sim, 0.85 */
    uint32_t min_aln_mapq;  /* hard-wired 55 (Commandline.cpp:60) */
    double max_uniq_dev;    /* --uniq-dev, 0.15 */
    uint32_t min_edge_sup;  /* --edge-sup, 3 */
    double uniq_freq;       /* calc_uniq_freq, Contig.cpp:162-174 */
} hx_params;

/* Short-read contigs: what the filters look up (Contig_t.mean_kmer, .len). */
typedef struct {
    uint32_t n;
    const double* mean_kmer;
    const uint32_t* len;
} hx_contigs;

/* Long reads: 2-bit packed bases. Layout is the build's own (NOT the reference's reversed-byte
 * codec, Compressed_sequence.cpp:46-62): base i of a read lives in byte off+i/4, bits 2*(i%4),
 * A=0 C=1 G=2 T=3, and anything else packs as A exactly like the reference (`_dna_tableVal[..] & 3`).
 * Every read starts on a 4-byte boundary so a lane can fetch 16 bases with one aligned dword. */
typedef struct {
    uint32_t n;
    const uint32_t* len;   /* bases */
    const uint64_t* off;   /* byte offset of read i in `packed`; n+1 entries, multiples of 4 */
    const uint8_t* packed;
} hx_reads;

/* Raw PAF records, one per PAF line, in file order (Align_Seq2_t minus q_len, plus CIGAR ops). */
typedef struct {
    uint64_t n;
    const uint32_t *q_id, *q_start, *q_end, *t_id, *t_len, *t_start, *t_end, *n_match, *n_block;
    const uint8_t *is_rev, *mapq;
    const uint64_t* cg_off; /* n+1 */
    const uint32_t* cg_ops; /* cg_off[n] words */
} hx_hits;

/* Surviving alignments after filters 1-5 + palindrome rule, grouped by read in (q_end,q_start) order,
 * then overlap-trimmed (the reference's `alignments` arena after fix_alignments, Longread.cpp:626).
 * A trimmed CIGAR is always a contiguous piece of the raw one: ops [cg_begin,cg_end) with
 * cg_skip_front bases removed from the first op and cg_skip_back from the last. */
typedef struct {
    uint64_t n_aln;
    uint32_t n_reads;
    uint32_t* hit; /* index into hx_hits */
    uint32_t *q_start, *q_end, *t_start, *t_end, *n_match, *n_block;
    uint64_t *cg_begin, *cg_end;
    uint32_t *cg_skip_front, *cg_skip_back;
    uint64_t* read_off; /* n_reads+1: alignments of read r are [read_off[r], read_off[r+1]) */
    /* compact long reads (find_best_scheduling, Longread.cpp:524-610) */
    uint64_t n_cmp;
    uint64_t* cmp_off; /* n_reads+1 */
    uint32_t* cmp_aln; /* global alignment index of each compact element */
} hx_chain_out;

/* One side (head or tail) of an edge-support record: a copy of the compact element it points at, so
 * that a record is self-contained after the multi-GPU all-gather (raw CIGAR ops and packed reads are
 * replicated on every GPU; the alignment table is not). */
typedef struct {
    uint32_t *q_start, *q_end, *t_start, *t_end;
    uint8_t* is_rev;
    uint64_t *cg_begin, *cg_end;
    uint32_t *cg_skip_front, *cg_skip_back;
} hx_rec_side;

/* Edge-support multiset, sorted by (key, emission order) which reproduces the reference's per-edge
 * `edge_supp` vectors (Backbone_graph.cpp:10-25,148-171).  key = (n1<<1|rev1)<<32 | (n2<<1|rev2):
 * high word = source vertex (node, which end is left), low word = the reference's map key. */
typedef struct {
    uint64_t n_rec;
    uint64_t* key;
    uint32_t* lr; /* lr_id | lr_strand<<31 */
    uint32_t *cmp_head, *cmp_tail;
    hx_rec_side head, tail;
    uint64_t n_edge;
    uint64_t* edge_key;
    uint64_t* edge_off; /* n_edge+1; support count = edge_off[e+1]-edge_off[e] */
} hx_edges_out;

/* Per processed edge (asm_calc_single_edge_coordinates, Assemble.cpp:157-363). */
typedef struct {
    uint32_t n_edge;
    uint32_t *head_end, *tail_beg;
    uint64_t* supp_off; /* n_edge+1 */
    uint32_t *supp_lr;  /* lr_id | lr_strand<<31 */
    uint32_t *spos, *epos;
} hx_coords_out;

/* Per processed edge consensus (asm_calc_single_cns_seq, Assemble.cpp:479-560): ASCII ACGT. */
typedef struct {
    uint32_t n_edge;
    uint64_t* cns_off; /* n_edge+1 */
    char* cns;
    /* work counters, for the roofline report */
    uint64_t dp_cells;   /* sum over alignments of graph_nodes * seq_len (what the full-matrix reference computes) */
    uint64_t seq_bases;  /* bases fed to POA */
    uint64_t n_aligned;  /* sequences aligned */
} hx_cns_out;

/* POA scoring (Assemble.cpp:8-11): global NW, linear gap. */
typedef struct {
    int32_t match, mismatch, gap;
} hx_poa_params;

#ifdef __cplusplus
}
#endif
#endif
