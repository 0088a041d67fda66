// kernels.h — host-callable launchers of the gfx950 kernels. All launch on `stream` and return
// immediately; temporary buffers they need come from the caller (sizes via the *_temp_* helpers)
// or are allocated internally where noted.
#ifndef HX_KERNELS_H
#define HX_KERNELS_H
#include <vector>

#include "common.h"

namespace hxk {

// ---- primitives.hip
// Scratch of the launchers: bump allocation from device blocks that live as long as their owner (the context), so that no operator
// allocates, frees or synchronises for its temporaries. Everything is handed out in stream order on ONE stream; reset() declares all of it
// free again for the work enqueued afterwards.
struct Workspace {
    struct Block { char* p; size_t cap, used; };
    std::vector<Block> blocks;
    bool oom = false;               // sticky: an allocation failed since the owner last looked (the launchers then skip their work; the operator must report it)
    void* take(size_t bytes);       // 256-byte aligned; nullptr when the device is out of memory (and oom is set)
    void reset(hipStream_t s);      // (several blocks are merged into one of their total size, after the stream has drained)
    void release();
    ~Workspace() { release(); }
};
// out[i] = sum(in[0..i)), out[n] = total.  (n+1 outputs)
void exclusive_scan_u32(const uint32_t* in, uint64_t* out, uint64_t n, hipStream_t s, Workspace& ws);
// stable LSD radix sort of (key, val) pairs on bits [0,bits_lo) and [32,32+bits_hi) of the key
void radix_sort_pairs(uint64_t* key, uint32_t* val, uint64_t* key_tmp, uint32_t* val_tmp, uint64_t n,
                      int bits_lo, int bits_hi, hipStream_t s, Workspace& ws);

// strings scattered over `src` moved side by side into `dst`: desc = n_items x (source offset lo / hi, destination offset lo / hi, length)
void gather_bytes(const char* src, const uint32_t* desc, uint32_t n_items, char* dst, hipStream_t s);
// one wave with 1 KB of private memory per lane: the stream's hardware queue grows its scratch to the largest any POA instance needs
void scratch_warm(hipStream_t s);

// ---- chain.hip (K0-K3)
struct ChainScratch {   // all sized by the number of raw hits in the shard (+1)
    uint32_t *hit, *qs, *qe, *ts, *te, *nm, *nb, *skf, *skb;
    uint64_t *cb, *ce;
    uint32_t* dp;
    int32_t* from;
    uint32_t* cmp;        // local compact indices
    uint32_t *n_aln, *n_cmp;   // per read of the shard
};
void contig_class(const double* mean_kmer, uint32_t n, double thr_load, double thr_uniq, uint8_t* cls, hipStream_t s);
void chain_reads(const DevHits& h, const uint64_t* read_hit_off, const uint8_t* cls, uint32_t n_contigs,
                 uint32_t lr_begin, uint32_t lr_end, uint32_t min_aln_block, double min_aln_sim, uint32_t min_mapq,
                 const ChainScratch& sc, uint32_t* err, bool prefiltered /* records of an index.longread: no filters, sort or group rule */, hipStream_t s);
struct ChainFinal {
    uint32_t *hit, *qs, *qe, *ts, *te, *nm, *nb, *skf, *skb;
    uint64_t *cb, *ce;
    uint32_t* cmp_aln;
};
void chain_compact(const ChainScratch& sc, const uint64_t* read_hit_off, uint32_t lr_begin, uint32_t lr_end,
                   const uint64_t* aln_off, const uint64_t* cmp_off, const ChainFinal& out, hipStream_t s);

// ---- edges.hip (K4)
struct EdgeRecs {
    uint64_t* key;
    uint32_t *lr, *cmp_head, *cmp_tail;
    DevSide head, tail;
};
void edge_count(const DevHits& h, const uint8_t* cls, const ChainFinal& c, const uint64_t* cmp_off,
                uint32_t lr_begin, uint32_t lr_end, uint32_t* n_pairs, hipStream_t s);
void edge_emit(const DevHits& h, const uint8_t* cls, const ChainFinal& c, const uint64_t* cmp_off,
               uint32_t lr_begin, uint32_t lr_end, const uint64_t* pair_off, const EdgeRecs& out, hipStream_t s);
void edge_gather(const EdgeRecs& in, const uint32_t* perm, uint64_t n, const EdgeRecs& out, hipStream_t s);
constexpr int EDGE_REC_WORDS = 11;   // packed exchange layout: dwords per record (a forward + twin pair travels as one unit of 22); n = records (even)
void edge_pack(const EdgeRecs& r, uint64_t n, uint32_t* dst, hipStream_t s);
void edge_unpack(const uint32_t* src, uint64_t n, const EdgeRecs& r, hipStream_t s);
void iota_u32(uint32_t* p, uint64_t n, hipStream_t s);
// head flags of equal-key segments (flag[i] = key[i] != key[i-1])
void segment_flags(const uint64_t* key, uint64_t n, uint32_t* flag, hipStream_t s);
void segment_scatter(const uint64_t* key, const uint64_t* flag_scan, uint64_t n, uint64_t* edge_key, uint64_t* edge_off, hipStream_t s);

// ---- coords.hip (K5)
struct CoordsScratch {   // sized by records of the selected edges (x2 for the output lists)
    uint64_t *beg1, *end1, *beg2, *end2;   // sorted (value<<32 | support index)
    uint8_t *cur, *best1, *best2;
    uint32_t* best_list;
};
void edge_coords(const EdgeRecs& recs, const uint64_t* edge_key, const uint64_t* edge_off, const uint32_t* cg_ops,
                 const uint32_t* contig_len, const uint32_t* read_len, uint32_t n_sel, const uint32_t* sel_edge,
                 const uint64_t* sel_rec_off,   // n_sel+1: scratch / output capacity offsets (records, doubled for hairpins)
                 const CoordsScratch& sc, uint32_t* head_end, uint32_t* tail_beg, uint32_t* n_supp,
                 uint32_t* supp_lr, uint32_t* spos, uint32_t* epos, int lds_supp /* supports per edge sorted in LDS: -1 = the kernel's capacity */, hipStream_t s);
void coords_compact(const uint64_t* cap_off, const uint64_t* out_off, uint32_t n_sel, const uint32_t* lr_in, const uint32_t* sp_in,
                    const uint32_t* ep_in, uint32_t* lr_out, uint32_t* sp_out, uint32_t* ep_out, hipStream_t s);

// ---- poa.hip (K6)
struct PoaSeq { uint32_t rid; uint32_t strand; uint32_t spos; uint32_t len; };
struct PoaEdge {
    uint32_t seq_begin, seq_end;   // into the PoaSeq table
    uint32_t vcap, ecap, lmax, hrows;   // capacities the kernel checks this edge against (all within its workspace slot)
    uint64_t cns_off;              // into the consensus output (bytes, capacity vcap): per edge, it outlives the workspace
    uint64_t cl_off;               // into the cluster pools (members * (vcap+1) entries per edge), members > 1 only
    uint32_t members;              // workgroups ("members", one CU each) that share this edge's DP columns; 1 = the usual single workgroup
    uint32_t wrows;                // rows of the wide-row pool (an estimate, overflow -> retry)
    uint32_t slot;                 // workspace slot of a shared edge (members > 1); other edges run in the slot of the workgroup that pulls them
    uint32_t passes;               // column passes (members == 1): the workgroup's waves take the DP columns of a sequence in this many windows, one after the other (1: all at once)
};
// A workspace slot: offsets into the pools. A persistent workgroup owns one for its lifetime (sized for the largest edge of its launch), a shared
// edge owns one for the call.
struct PoaSlot {
    uint64_t node_off, edge_off;   // into the node / edge pools (elements)
    uint64_t h_off;                // into the H pool (int32 cells): hrows rows of W + one word per wave of the edge's pipeline. Direction-byte traceback: only the rows a far successor reads
                                   // (hrows = an estimate, overflow -> retry); score-matrix traceback: all vcap + 1 rows
    uint64_t d_off;                // into the direction pool (bytes): vcap + 1 rows of W / 2 (a 4-bit move code per cell)
    uint64_t w_off;                // into the wide-row pool (bytes): wrows rows of W (a move byte per cell of the rows with more than 4 predecessors)
    uint64_t seq_off;              // into the decoded-sequence pool (bytes, lmax per edge)
    uint64_t stack_off;            // into the toposort stack pool (4*(vcap+1) + ecap entries per edge)
    uint64_t aln_off;              // into the alignment pools (vcap + lmax + 2 entries per edge)
    uint64_t mbox_off;             // into the mailbox pool: passes x (vcap + 1) words for the carries handed from one column pass to the next (edges with passes > 1)
};
struct PoaPools {
    // per node (pool length = sum (vcap+1))
    uint8_t* code; uint8_t* n_aligned; uint32_t* aligned;   // aligned: 3 per node
    uint32_t *in_head, *in_tail, *out_head, *out_tail;
    uint32_t *rank2node, *node2rank;
    uint8_t *mark, *check; uint32_t* stack;                 // toposort scratch (stack has its own pool, PoaEdge::stack_off)
    int32_t* score; int32_t* pred;                          // consensus scratch (score as int64 is not needed: weights < 2^31)
    // rank-order CSR rebuilt after every toposort
    uint8_t* row_code; uint8_t* row_sink; uint32_t* row_pred_off; uint32_t* pred_rank;   // row_pred_off: vcap+1 per edge; pred_rank: ecap
    uint32_t *row_meta, *row_pred0, *row_pred1;   // code | sink<<2 | far<<3 | npred<<8 ; first two predecessor ranks
    uint4* nrec;   // per node 16-byte record for the serial graph walks (first in-edge source, counts, packed aligned ids)
    uint4* nrec2;  // ... second record: first two in-edge ids, first two out-edge targets (kernels/poa.hip G)
    // per graph edge (pool length = sum ecap)
    uint32_t *e_from, *e_to, *e_next_in, *e_next_out; int32_t* e_w;
    // alignment output of the traceback (node|-1, pos|-1), own pool, PoaEdge::aln_off
    int32_t* aln_node; int32_t* aln_pos;
    int32_t* H;
    uint8_t* dir;   // traceback move codes, 4 bits per cell (PoaEdge::d_off)
    uint8_t* dirw;  // move bytes of the rows with more than 4 predecessors (PoaEdge::w_off)
    uint32_t* wslot;   // per rank: row of the wide-row pool
    uint8_t* seq;
    // cluster mode (an edge's DP columns spread over several workgroups):
    unsigned long long* mbox;   // [member][row] = {tag, carry}: prefix maximum of the row through the member's last column (PoaEdge::cl_off)
    uint32_t* csync;            // 8 words per edge: go, done, V, L, error
    int32_t* sinkbuf;           // 1 + 2*1024 words per edge: sink rows / scores when the last column lives in another member
    uint16_t* row_al;           // per rank: ranks of the node's aligned nodes in list order, 3 x 3 bits (rank delta + 4, 0 = none)
    int32_t* pred_w;            // per entry of pred_rank: the weight of that in-edge (the heaviest bundle runs on the rank-ordered rows)
};
// kernel instance (largest workgroup it is compiled for) that serves workgroups of `block_threads` lanes, and the columns per lane it can be had with
inline int poa_kernel_lanes(int block_threads) { return block_threads <= 64 ? 64 : block_threads <= 256 ? 256 : block_threads <= 512 ? 512 : 1024; }
inline int poa_kernel_max_cm(int block_threads) { return block_threads <= 256 ? 32 : block_threads <= 512 ? 16 : 32; }
// (2 columns per lane: instances for the members of shared edges only - 256-lane workgroups and the 1024-lane wide members, direction bytes, one workgroup per
// entry of the launch. A lone wave's row is bound by its instruction count, and a row of 2 columns is ~28 instructions shorter than one of 4.)
inline int poa_kernel_min_cm(int block_threads, bool shared, bool use_dir) { return shared && use_dir && (block_threads == 256 || block_threads == 1024) ? 2 : 4; }
inline bool poa_persistent_ok(bool use_dir) { return use_dir; }   // launches that can run persistent (the score-matrix flavour is rare: one workgroup per edge)
// Exact score-bound pruning of the DP (kernels/poa.hip "PRUNE"): instances exist for the direction-byte flavour with 4 or 8 columns per lane, for
// launches of one workgroup per edge (a shared edge's members would have to repeat an attempt together)
inline bool poa_prune_ok(bool use_dir, int cm) { return use_dir && cm <= 8; }
constexpr int32_t PRUNE_OFF = -(1 << 24);   // a threshold no real cell is below (|scores| < 2^24: the host checks 8 (nodes + columns))
constexpr int POA_PHASE_WORDS = 21;         // per edge: 6 phase cycle counters, 6 row statistics, 4 of the pruning (wave-rows, wave-rows skipped, attempts repeated, alignments with a threshold), the edge's begin and end on the 100 MHz wall clock, cycles of the final consensus and whether it took the reference's order, wave-rows skipped a batch at a time
// One launch of a class. counter == nullptr: one workgroup per entry of `order` (edge | member << 24; shared edges, each in its own slot
// PoaEdge::slot); else PERSISTENT: n_blocks workgroups, workgroup b owns slots[b]; the n_items edges of `order` come in nb buckets of workspace need
// (largest first), bucket k = order[item_begin[k] .. item_begin[k + 1]) behind counter[k], its workgroups = the slots [slot_end[k - 1], slot_end[k]);
// btab (device) = {nb, slot_end[0 .. nb), item_begin[0 .. nb], est[0 .. n_items)}: est = the estimated chain time of every entry of `order` (any unit). A
// workgroup takes, of the next edges of its bucket and of the later (smaller) ones, the one with the longest chain.
struct PoaLaunch {
    const PoaEdge* edges; const uint32_t* order; uint32_t n_items; const PoaSlot* slots; uint32_t* counter; const uint32_t* btab; uint32_t n_blocks;
    const PoaSeq* seqs; const uint8_t* packed; const uint64_t* read_off; const uint32_t* read_len; PoaPools pools;
    int32_t match, mismatch, gap; char* cns; uint32_t *cns_len, *status;
    unsigned long long *cells, *phase /* POA_PHASE_WORDS per edge or null */;
    int block_threads /* multiple of 64, <= 1024 */;
    int cm /* columns per lane of the launch: 4, 8, 16 or 32; every edge's longest sequence fits members x block_threads x cm columns */;
    uint32_t poll_limit /* polls before a wave gives up waiting for another (-> HXE_POA_STALLED) */, ring_bytes /* dynamic LDS */;
    bool use_dir /* direction-byte traceback (in-degrees <= max_indeg <= 16, else the edge comes back with HXE_POA_NODIR) */;
    uint32_t max_indeg, dp_lanes /* 0: every lane of the workgroup; else the lanes that take part in the DP (a wide cluster member) */;
    int* occupancy /* not null: no launch - the workgroups of this launch's shape a CU holds (hipOccupancyMaxActiveBlocksPerMultiprocessor), for the debug output */;
    uint32_t* started /* null, or a word of HOST memory (mapped): every workgroup of the launch adds 1 when it begins */;
    uint32_t prune_pct /* 0: full matrix; else the pruned instance (poa_prune_ok, unshared edges) with thresholds at this percentage of the previous alignment's score per base */;
};
void poa_run(const PoaLaunch& q, hipStream_t s);

}  // namespace hxk
#endif
