/* haslr_hip.h — C-ABI of the MI355X (gfx950) implementation of haslr_assemble's hot path.
 *
 * Plain C: opaque context, caller-visible PODs from haslr_types.h, `int` results (0 = ok, <0 = error,
 * text via hx_last_error()), no exceptions and no C++/torch types across the boundary. One context per GPU,
 * one host thread per context. The library fails loudly (error return) when no HIP device is usable;
 * there is no CPU fallback behind these symbols.
 *
 * What each entry point replaces in the reference (paths under /root/reference/src/haslr_assemble/src/):
 *
 *   hx_upload          the in-memory products of load_contig_compressed (Contig.cpp:43-117),
 *                      load_longread_compressed (Longread.cpp:109-162) and the record parse of
 *                      load_alignment (Longread.cpp:250-289), copied once to HBM
 *   hx_chain_reads     filters 1-4 of load_alignment (Longread.cpp:262-272) + per-read sort (:256) +
 *                      process_lr_alignment_group (:182-232) + fix_alignments (:626-635, :430-512) +
 *                      build_compact_longreads (:612-624, :524-610); called from main.cpp:98,116,124
 *   hx_edge_support    bbg_build_graph / bbg_add_edge (Backbone_graph.cpp:148-171, :10-25); main.cpp:131.
 *                      (bbg_remove_weak_edges :348-375 is a count test the host applies to the result.)
 *   hx_edge_coords     asm_calc_edge_coordinates_MT (Assemble.cpp:453-477 -> :157-363); main.cpp:203
 *   hx_poa_batch       asm_cal_cns_seq_MT (Assemble.cpp:580-605 -> :479-560), i.e. the five SPOA 1.1.3 calls
 *                      createAlignmentEngine/createGraph/align_sequence_with_graph/add_alignment/
 *                      generate_consensus at Assemble.cpp:499,500,539,540,554; main.cpp:207
 *
 * The four operators have the signature of the host pipeline's `hx_backend` table
 * (haslr_amd/csrc/host/haslr_host.h); hx_backend_fill() wires them.
 */
#ifndef HASLR_HIP_H
#define HASLR_HIP_H
#include "haslr_types.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hx_ctx hx_ctx;

const char* hx_last_error(void);
int hx_device_count(void); /* number of HIP devices, <0 on error */

/* stream: a hipStream_t to launch on (e.g. torch's current stream), or NULL to create one.
 * Process environment: the library never modifies it. hx_poa_batch launches its lane-count classes on separate streams; they run side
 * by side only if the HIP runtime was initialised with GPU_MAX_HW_QUEUES >= 8 (the runtime's default is 4), so an application that wants
 * the measured POA throughput exports that variable before its first HIP call (haslr_assemble, haslr_amd/hip.py and bench.py do).
 * Results do not depend on it. */
int hx_ctx_create(int device, void* stream, hx_ctx** out);
void hx_ctx_destroy(hx_ctx*);

/* Tuning and test switches of a context: what used to be HX_* environment variables read inside the library (31 of them, on every call) is
 * state of the context. The library reads NO environment variable; applications that want the old behaviour copy the HX_* variables of their
 * environment in when they create a context (haslr_assemble and haslr_amd/hip.py do, bench.py through hip.py). The reference keeps its knobs in
 * one options struct filled by its command line (Common.hpp:44-65, Commandline.cpp:46-66); these are the knobs it does not have.
 *   hx_set_option    name = an entry of hx_option_names() (the old spelling works too: "HX_POA_SLOTS" = "poa_slots"), value = a number as text;
 *                    NULL or "" = back to the default. Unknown name or malformed value: error. Takes effect with the next operator call.
 *   hx_get_option    current value as a double
 *   hx_option_names  comma-separated list: debug, prof, poa_workspace_gb (cap of the consensus workspace, GB; 0 = 90 % of the free memory),
 *                    poa_prune (exact score-bound pruning of the DP: -1 automatic = calls of thousands of edges, 0 never, else the threshold as a
 *                    percentage of the previous alignment's score per base), launch-shape knobs (poa_cols, poa_cols2_top, poa_member_lanes, poa_cluster_min /
 *                    _max / _topk / _cols, poa_wide_members, poa_wave_max, poa_ring_kb, poa_balance, poa_balance_pct, poa_balance_lanes, poa_streams,
 *                    poa_wide_delay_us, poa_prune_lanes, poa_prune_shared, poa_prune_lazy, poa_order_by_cells, poa_pass_lanes, poa_chain_ms, poa_chain_pct, poa_big_first,
 *                    poa_slots_by_work, poa_bucket_half_octaves, poa_own_bucket_first, poa_resident_first, poa_far_shift, poa_scratch_warm) and test switches that force rare paths (poa_poll_limit, poa_max_indeg, poa_node_est_pct,
 *                    poa_far_rows, poa_ring_zero, poa_slots, poa_slots_pct, poa_batches, poa_force_cm, poa_no_xcd_map, coords_lds_supp).
 *                    Results never depend on any of them. */
int hx_set_option(hx_ctx*, const char* name, const char* value);
int hx_get_option(const hx_ctx*, const char* name, double* value);
const char* hx_option_names(void);

/* copy inputs to HBM; they stay resident for the life of the context (replicated on every GPU in a
 * multi-GPU run). Read shard defaults to all reads. */
int hx_upload(hx_ctx*, const hx_contigs*, const hx_reads*, const hx_hits*, const uint64_t* read_hit_off);
int hx_set_read_shard(hx_ctx*, uint32_t lr_begin, uint32_t lr_end);
/* the uploaded records are the reference's FILTERED set, read back from an index.longread (Longread.cpp:341-372): hx_chain_reads then takes
 * them as they are - no filters 1-5, no sort, no group / palindrome rule - and goes straight to trim + chain, like main.cpp:90-116 does
 * after read_longread_index. Reset by hx_upload. */
void hx_set_prefiltered(hx_ctx*, int on);

int hx_chain_reads(hx_ctx*, const hx_params*, hx_chain_out* out);
int hx_edge_support(hx_ctx*, const hx_params*, hx_edges_out* out);
int hx_edge_coords(hx_ctx*, uint32_t n_sel, const uint32_t* sel_edge, hx_coords_out* out);
int hx_poa_batch(hx_ctx*, const hx_poa_params*, hx_cns_out* out);
/* The consensus operator on its own (what stands where the five SPOA calls stand, Assemble.cpp:499-554), for callers that hold the
 * support lists themselves:
 *   hx_poa_supports   consensus of caller-given edges: `sup` has hx_edge_coords' layout (n_edge, supp_off, supp_lr = read id |
 *                     strand << 31, spos, epos; head_end / tail_beg are not read) and points into the resident reads. The sub-sequence
 *                     rule is the reference's: epos - spos + 1 in 32 bits, clamped to the read like std::string::substr
 *                     (Assemble.cpp:530-532), empty ones skipped (:537), no sequence at all -> empty consensus (:544-551).
 *   hx_poa_sequences  consensus of caller-given sequence sets (plain ACGT text, set i = sequences [set_off[i], set_off[i+1]), sequence
 *                     k = bases[seq_off[k] .. seq_off[k+1])), aligned in the given order; nothing has to be resident. This is the
 *                     entry include/spoa_hx.hpp (the spoa.hpp-shaped C++ header over this library) calls. */
int hx_poa_supports(hx_ctx*, const hx_coords_out* sup, const hx_poa_params*, hx_cns_out* out);
int hx_poa_sequences(hx_ctx*, uint32_t n_sets, const uint64_t* set_off, const uint64_t* seq_off, const char* bases, const hx_poa_params*, hx_cns_out* out);
void hx_free_chain(hx_ctx*, hx_chain_out*);
void hx_free_edges(hx_ctx*, hx_edges_out*);
void hx_free_coords(hx_ctx*, hx_coords_out*);
void hx_free_cns(hx_ctx*, hx_cns_out*);

/* multi-GPU exchange of the edge-support multiset (one all-gather between hx_chain_reads and the sort):
 *   hx_edge_emit            emit this shard's records (unsorted) on the device, returns their number
 *   hx_edge_records_bytes   bytes per record in the packed exchange layout (44: a forward record and its twin travel as one
 *                           88-byte unit - key, read id, compact indices and the two trimmed anchor alignments, once)
 *   hx_edge_records_export  pack the local records into a caller-owned DEVICE buffer (n * bytes; n is even)
 *   hx_edge_records_import  replace the record set by n records unpacked from a DEVICE buffer (all ranks,
 *                           rank order), then sort + segment; fills `out` like hx_edge_support */
int hx_edge_emit(hx_ctx*, const hx_params*, uint64_t* n_records);
uint32_t hx_edge_records_bytes(void);
int hx_edge_records_export(hx_ctx*, void* dst_device, uint64_t capacity_records);
int hx_edge_records_import(hx_ctx*, const void* src_device, uint64_t n_records, hx_edges_out* out);

/* Multi-GPU inside ONE process: the GPUs of a node behind the same binary, like the reference's worker threads behind asm_calc_edge_coordinates_MT
 * / asm_cal_cns_seq_MT (Assemble.cpp:453-477, :580-605, called from main.cpp:203-208). A group holds one context per rank (one host thread
 * each) and one RCCL communicator per rank (ncclCommInitAll; librccl is loaded at run time, only by this call).
 *   hx_group_create        n ranks on `devices` (NULL: ranks 0..n-1 on devices 0..n-1). `transport`: NULL = RCCL over xGMI when every rank has its
 *                          own device; "rccl" insists on it; "host" stages the exchange through host memory and lets ranks share devices (a
 *                          rehearsal of the multi-GPU logic on a box with fewer GPUs than ranks; haslr_assemble passes its HASLR_GROUP_TRANSPORT).
 *   hx_group_ctx           the rank's context: hx_upload (inputs are replicated), hx_set_read_shard, hx_set_prefiltered per rank as usual
 *   hx_edge_merge          COLLECTIVE - every rank's thread calls it once per pass, after hx_chain_reads on its read shard: emits the shard's
 *                          edge-support records, all-gathers the packed records (ONE ncclAllGather, padded to the largest shard; counts are
 *                          exchanged through the process's memory), imports the concatenation (rank order = read order), sorts and segments:
 *                          `out` like hx_edge_support, identical on every rank (bbg_build_graph's multiset, Backbone_graph.cpp:148-171).
 *                          The ranks agree on failure before the collective: if one fails there (emission, export, buffers), all return an error, none
 *                          hangs. A failure INSIDE the collective (ncclAllGather returning an error on one rank, a fault on its stream, or no
 *                          completion within hx_group_set_timeout - 300 s by default) raises the group's abort flag: every rank polls its stream
 *                          instead of blocking on it, aborts its communicator (ncclCommAbort) when it sees the flag, and all ranks return the
 *                          error; the group refuses further exchanges.
 *   hx_group_backend_fill  the rank's hx_backend table: its own chain / coordinate / consensus operators, hx_edge_merge as edge_support
 *   hx_group_exchange_stats  bytes of records exchanged by the last hx_edge_merge and its wall time in ms */
typedef struct hx_group hx_group;
int hx_group_create(int n_ranks, const int* devices, const char* transport, hx_group** out);
void hx_group_set_timeout(hx_group*, double seconds);
void hx_group_inject_fault(hx_group*, int rank); /* testing: this rank's next all-gather "returns an error" (-1: none) */
void hx_group_destroy(hx_group*);
int hx_group_size(const hx_group*);
hx_ctx* hx_group_ctx(hx_group*, int rank);
const char* hx_group_transport(const hx_group*); /* "rccl" or "host" */
int hx_edge_merge(hx_group*, int rank, const hx_params*, hx_edges_out* out);
int hx_group_backend_fill(hx_group*, int rank, void* backend_table);
void hx_group_exchange_stats(const hx_group*, uint64_t* bytes, double* ms);
int hx_group_rccl_ranks(const hx_group*, int* ranks_per_communicator /* hx_group_size entries */); /* ncclCommCount of every rank's communicator (0s: host transport); returns 1 over RCCL, 0 over host memory */

/* kernel timing measured with hipEvents on the context's stream, accumulated per kernel family since the
 * last reset: 0 chain, 1 edges (emit+sort+segment), 2 coords, 3 poa. ms[] and launches[] have 4 entries. */
void hx_timing_reset(hx_ctx*);
void hx_timing_get(hx_ctx*, double* ms, uint64_t* launches);
/* diagnostics of the last hx_poa_batch: shader-clock cycles spent by lane 0 per phase [decode, dp, traceback,
 * graph update + consensus, toposort, csr], summed over edges (sum6) and for the slowest edge (max6); returns #edges.
 * With option debug set it also prints the per-class row statistics of the call to stderr. */
uint32_t hx_poa_phase_cycles(hx_ctx*, uint64_t* sum6, uint64_t* max6);
/* bytes of POA workspace (all pools) the largest hx_poa_batch of this context has needed: (resident work-groups) x (largest edge of a launch
 * class) + the shared edges' own slots - not the sum over the edges of the call */
uint64_t hx_poa_workspace_bytes(const hx_ctx*);
/* memory of the consensus stage: device memory free when the context first ran a consensus (the budget is 90 % of it unless option
 * poa_workspace_gb says otherwise), the budget, and the workspace the LAST call settled on (slot counts are scaled down until it fits) */
/* frees the consensus workspace (the pools grow to the largest call and otherwise live as long as the context) and forgets the memory budget: the next
 * consensus call takes it again from what is free then, and allocates anew */
int hx_poa_release_workspace(hx_ctx*);
/* The consensus workspace is ONE device allocation (an arena every pool of a batch is carved from), made by the first hx_poa_batch that needs it - or
 * ahead of it by hx_poa_reserve(bytes): a one-shot program (the reference is one, main.cpp:28-228: every stage runs exactly once, the consensus at :207)
 * calls it on a thread of its own while it still parses its text inputs, so that the allocation of 10^2 GB is not part of its consensus stage. At most
 * 80 % of the device memory that is free at the time is taken (and no more than option poa_workspace_gb allows, when it is set); hx_upload gives the arena back if the inputs do not fit beside it; a call that needs more
 * than was reserved allocates again. Thread-safe against the operators of the same context. The first reservation (or, without one, the first consensus call) of a
 * process also takes the hardware queues of the launch streams to the scratch size the largest kernel instance asks for - one wave each, once per process and
 * device (option poa_scratch_warm=0: not): queues that grow their scratch in the middle of a call hold some of its launches back.
 *   hx_poa_host_times   host wall time (ms) of the LAST consensus call, by part: [0] plan, [1] workspace (arena allocation + carving), [2] enqueue
 *                       (tables to the device, launches), [3] waiting for the device, [4] collection (status + consensus strings), [5] results
 *                       assembled, [6] unused, [7] the whole call
 *   hx_poa_arena_stats  bytes the arena holds, device allocations made for it so far, and their wall time */
int hx_poa_reserve(hx_ctx*, uint64_t bytes);
void hx_poa_host_times(const hx_ctx*, double* ms8);
void hx_poa_arena_stats(const hx_ctx*, uint64_t* capacity, uint64_t* allocations, double* alloc_ms);
void hx_poa_memory_stats(const hx_ctx*, uint64_t* free_at_first_call, uint64_t* budget, uint64_t* last_call_workspace);
/* pruning statistics of the last hx_poa_batch, summed over the pruned launches: [wave-rows, wave-rows skipped, attempts repeated, alignments with a threshold] */
void hx_poa_prune_stats(const hx_ctx*, uint64_t* out4);
/* POA work-group size: 0 = automatic (64..256 lanes per edge, ~8 DP columns per lane; gaps > 2047 columns are shared by several
 * work-groups), or force one work-group of 64/128/256/512/1024 lanes per edge (gaps up to 32767 bases) */
void hx_set_poa_block(hx_ctx*, int threads);
/* traceback source: 1 (default) = 1-byte direction codes written by the DP (an edge with a node of more than 16 in-edges is redone with 0), 0 = always re-derive the
 * moves from the int32 score matrix like the reference engine does (diagnostics / A-B comparison; results are identical) */
void hx_set_poa_traceback(hx_ctx*, int use_direction_bytes);

/* fill the host pipeline's backend table with the entry points above (struct hx_backend of haslr_host.h,
 * passed as void* to keep this header free of host-side types) */
void hx_backend_fill(hx_ctx*, void* backend_table);

#ifdef __cplusplus
}
#endif
#endif
