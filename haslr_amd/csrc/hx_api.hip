// hx_api.hip — implementation of the C-ABI in include/haslr_hip.h: device context, resident inputs,
// the four hot-path operators (kernel orchestration + result download), multi-GPU record exchange, timing.
// There is no CPU fallback here: without a usable HIP device every entry point fails with an error.
#include <atomic>
#include <thread>
#include <memory>
#include <chrono>
#include <cmath>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>
#include <array>
#include <mutex>

#include "../../include/haslr_hip.h"
#include "host/haslr_host.h"
#include "kernels/kernels.h"

namespace {

thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return -1; }

#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
    } while (0)

template <class T>
struct DV {   // device vector (capacity grows, never shrinks)
    T* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap && p) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = std::max<size_t>(n, 1);
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    ~DV() { release(); }
};

struct DevSideBuf {
    DV<uint32_t> qs, qe, ts, te, skf, skb;
    DV<uint8_t> rev;
    DV<uint64_t> cb, ce;
    hipError_t reserve(size_t n) {
        hipError_t e;
        if ((e = qs.reserve(n)) || (e = qe.reserve(n)) || (e = ts.reserve(n)) || (e = te.reserve(n)) || (e = skf.reserve(n)) ||
            (e = skb.reserve(n)) || (e = rev.reserve(n)) || (e = cb.reserve(n)) || (e = ce.reserve(n))) return e;
        return hipSuccess;
    }
    DevSide view() { return DevSide{qs.p, qe.p, ts.p, te.p, rev.p, cb.p, ce.p, skf.p, skb.p}; }
};

struct RecBuf {
    DV<uint64_t> key;
    DV<uint32_t> lr, ch, ct;
    DevSideBuf head, tail;
    hipError_t reserve(size_t n) {
        hipError_t e;
        if ((e = key.reserve(n)) || (e = lr.reserve(n)) || (e = ch.reserve(n)) || (e = ct.reserve(n)) || (e = head.reserve(n)) || (e = tail.reserve(n))) return e;
        return hipSuccess;
    }
    hxk::EdgeRecs view() { return hxk::EdgeRecs{key.p, lr.p, ch.p, ct.p, head.view(), tail.view()}; }
};

template <class T> T* host_copy(const T* d, size_t n) {
    T* h = (T*)malloc(std::max<size_t>(1, n) * sizeof(T));
    if (n) (void)hipMemcpy(h, d, n * sizeof(T), hipMemcpyDeviceToHost);
    return h;
}

struct Timer {
    hipEvent_t a = nullptr, b = nullptr;
    double ms[4] = {0, 0, 0, 0};
    uint64_t launches[4] = {0, 0, 0, 0};
};

}  // namespace

namespace {
struct PoaPlan {
    std::vector<hxk::PoaSeq> seqs;
    std::vector<hxk::PoaEdge> edges;
    std::vector<uint64_t> sumL;
    std::vector<uint32_t> nseq;
};

// The POA workspace is ONE device allocation (round 6): an arena that every pool of a batch is carved out of. Forty pools used to be forty synchronous
// hipMalloc calls inside the first consensus call of a context - seconds of a one-shot run at 140 Mb (215 GB), against a 0.5 s hot path. The arena can be
// reserved ahead of the first call (hx_poa_reserve: the CLI does it on a thread of its own while the text inputs are parsed), grows when a batch needs
// more (never shrinks), and is carved anew for every batch: nothing in it outlives a batch.
template <class T> struct AP { T* p = nullptr; size_t off = 0; };   // a pool: pointer into the arena, byte offset of the current carving
struct PoaPoolBufs {
    AP<uint8_t> code, n_aligned, mark, check, row_code, row_sink, seq;
    AP<uint32_t> aligned, in_head, in_tail, out_head, out_tail, rank2node, node2rank, stack, row_pred_off, pred_rank, e_from, e_to, e_next_in, e_next_out;
    AP<int32_t> score, pred, e_w, aln_node, aln_pos, H, pred_w;
    AP<uint32_t> row_meta, row_pred0, row_pred1;
    AP<uint4> nrec, nrec2;
    AP<uint8_t> dir, dirw; AP<uint32_t> wslot;
    AP<unsigned long long> mbox; AP<int32_t> sinkbuf; AP<uint32_t> csync; AP<uint16_t> row_al;   // cluster mode (edges shared by several workgroups)
    AP<char> cns;                                                                                  // consensus strings as the kernels leave them (capacity = node estimate per edge)
};
struct PoaArena {
    uint8_t* p = nullptr;
    size_t cap = 0;
    uint64_t n_alloc = 0;       // device allocations made for it so far
    double alloc_ms = 0;        // ... and the wall time they took
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    hipError_t ensure(size_t bytes) {   // at least `bytes`; the contents are not kept
        if (bytes <= cap && p) return hipSuccess;
        const auto t0 = std::chrono::steady_clock::now();
        release();
        const hipError_t e = hipMalloc((void**)&p, std::max<size_t>(bytes, 256));
        if (e == hipSuccess) cap = std::max<size_t>(bytes, 256); else p = nullptr;
        n_alloc++; alloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return e;
    }
    ~PoaArena() { release(); }
};
}  // namespace

// ---- tuning and test switches of a context. They used to be HX_* environment variables read inside the library on every call; now they are
// state of the context, set through hx_set_option (include/haslr_hip.h) - by the applications (the CLI and haslr_amd/hip.py copy the HX_*
// variables of their environment in, once, when they create a context) and by the tests. Defaults in the table below; -1 = automatic.
namespace {
struct HxOptions {
    int debug = 0;                 // progress and statistics of every consensus call on stderr
    int prof = 0;                  // 1 / 2 / 3: how hx_poa_phase_cycles reads the phase words of a build with -DHX_DP_PROF / PROF2 / PROF3 (development)
    double poa_workspace_gb = 0;   // cap of the POA workspace in GB (0: 90 % of the memory that was free at the context's first consensus call)
    int poa_poll_limit = 1 << 24;  // polls before a wave gives up waiting for another member (testing: forces the unshared retry)
    int poa_max_indeg = 16;        // in-degree the direction bytes hold (testing: forces the score-matrix retry earlier)
    int poa_member_lanes = 256, poa_cluster_min = 2048, poa_cluster_max = -1, poa_cluster_topk = -1, poa_wide_members = -1, poa_cluster_cols = -1;
    int poa_cols2_top = -1;        // the costliest shared edges of a call whose members take 2 columns per lane (twice the members, a shorter row): how many (-1: 4 in a few-edge call, else none)
    int poa_node_est_pct = 100, poa_far_rows = -1;
    int poa_far_shift = 3;         // rings of 4 kept rows (the many-edge regime): rows of H (rows read back from HBM) per edge = nodes >> this, + 256; an edge that needs more is redone with 4 x the room
    int poa_wave_max = 512, poa_cols = -1, poa_ring_kb = -1, poa_ring_zero = 0;
    int poa_balance = 1, poa_balance_pct = 125, poa_balance_lanes = 512;
    int poa_slots_pct = 100, poa_slots = 0, poa_batches = 0, poa_force_cm = 0, poa_no_xcd_map = 0, poa_streams = 8, poa_wide_delay_us = 60;
    int poa_prune = -1;            // exact score-bound pruning of the DP: -1 automatic (calls of thousands of edges), 0 never, else the threshold's percentage of the previous alignment's score per base
    int poa_pass_lanes = -1;       // column passes: unshared multi-wave edges run in workgroups of this many lanes, their DP columns in windows taken one after the other (-1 automatic: by
                                   // estimated chain length, where the rows are pruned; 0 never)
    int poa_bucket_half_octaves = 1;   // need buckets of the persistent launches half an octave apart (0: an octave, as until round 6)
    int poa_own_bucket_first = 1;  // a persistent workgroup takes the edges of its OWN need bucket before those of the smaller buckets it can also serve (0: whichever next edge has the longest chain, as until round 6 - see k_poa)
    int poa_resident_first = 0;    // bit 0: the shared edges' launch of a many-edge call, bit 1: the wide persistent launches (512 lanes and more) - the next launch leaves when their workgroups have all begun (each adds itself to a word in host memory), not after a fixed delay. Measured at 140 Mb: the 37 workgroups of the 512-lane launch have all begun 40 us after it (the fixed delay is 60), and the one pass in five that took 550-620 ms was not about arrival at all (poa_own_bucket_first); 0 stays the default, the best pass is 0.459 against 0.480 s
    int poa_slots_by_work = 1;     // many-edge calls: the slots of an instance's need buckets in proportion to the buckets' estimated work (0: from the largest need down, as until round 6)
    int poa_order_by_cells = 0;    // few-edge calls: the launch lists in the order of the edges' DP cells (until round 6) instead of the rows of their chains
    int poa_big_first = 1;         // few-edge calls: the unshared classes of 512 lanes and more leave before the shared edges' 256-lane members (0: behind them, as before round 5)
    int poa_scratch_warm = 1;      // the streams' hardware queues are taken to the largest scratch size any POA instance needs before the first launches of a process (launch_batch)
    int poa_chain_pct = 70;        // the automatic chain cap: the smallest one that is at least this percentage of the call's estimated wave-slot time over the waves resident (size_edges; 60 until the persistent workgroups took their own bucket first)
    int poa_chain_ms = -1;         // ... the automatic choice: the narrowest workgroup whose estimated chain (size_edges: DP rows x what a row costs at that width and number of
                                   // windows) stays below this many milliseconds; -1: the cap that balances the longest chain against the call's wave-slot time
    int poa_prune_shared = 0;      // ... of the edges shared by several workgroups (round 6: their members take DP ATTEMPTS, not sequences, so a missed threshold is repeated by all of
                                   // them): 0 never (the default), else the percentage. Measured at 12 Mb / 4.6 Mb with 95: 65 % of the wave-rows skipped, same consensus - and the longest
                                   // chain 154 -> 207 ms / 106 -> 142 ms: in a pipeline of waves every row is live in SOME wave, which sets the pace of that row for all of them;
                                   // what a skipped wave-row frees is issue slots, and a lone chain is not short of those
    int poa_prune_lazy = 1;        // ... a wave that skipped a whole batch of rows polls for the next one rarely (0: like any wave)
    int poa_prune_lanes = 128;     // ... in launches of workgroups of at least this many lanes (a one-wave workgroup has no block to skip)
    int coords_lds_supp = -1;      // supports per edge the coordinate kernel sorts in LDS (testing: 0 sends every edge through the global scratch)
};
struct OptDesc { const char* name; int HxOptions::*ip; double HxOptions::*dp; };
const OptDesc kOptions[] = {
    {"debug", &HxOptions::debug, nullptr}, {"prof", &HxOptions::prof, nullptr}, {"poa_workspace_gb", nullptr, &HxOptions::poa_workspace_gb},
    {"poa_poll_limit", &HxOptions::poa_poll_limit, nullptr}, {"poa_max_indeg", &HxOptions::poa_max_indeg, nullptr}, {"poa_member_lanes", &HxOptions::poa_member_lanes, nullptr},
    {"poa_cluster_min", &HxOptions::poa_cluster_min, nullptr}, {"poa_cluster_max", &HxOptions::poa_cluster_max, nullptr}, {"poa_cluster_topk", &HxOptions::poa_cluster_topk, nullptr},
    {"poa_wide_members", &HxOptions::poa_wide_members, nullptr}, {"poa_cluster_cols", &HxOptions::poa_cluster_cols, nullptr}, {"poa_cols2_top", &HxOptions::poa_cols2_top, nullptr}, {"poa_node_est_pct", &HxOptions::poa_node_est_pct, nullptr},
    {"poa_far_rows", &HxOptions::poa_far_rows, nullptr}, {"poa_far_shift", &HxOptions::poa_far_shift, nullptr}, {"poa_wave_max", &HxOptions::poa_wave_max, nullptr}, {"poa_cols", &HxOptions::poa_cols, nullptr},
    {"poa_ring_kb", &HxOptions::poa_ring_kb, nullptr}, {"poa_ring_zero", &HxOptions::poa_ring_zero, nullptr}, {"poa_balance", &HxOptions::poa_balance, nullptr},
    {"poa_balance_pct", &HxOptions::poa_balance_pct, nullptr}, {"poa_balance_lanes", &HxOptions::poa_balance_lanes, nullptr}, {"poa_slots_pct", &HxOptions::poa_slots_pct, nullptr},
    {"poa_slots", &HxOptions::poa_slots, nullptr}, {"poa_batches", &HxOptions::poa_batches, nullptr}, {"poa_force_cm", &HxOptions::poa_force_cm, nullptr},
    {"poa_no_xcd_map", &HxOptions::poa_no_xcd_map, nullptr}, {"poa_streams", &HxOptions::poa_streams, nullptr}, {"poa_wide_delay_us", &HxOptions::poa_wide_delay_us, nullptr},
    {"poa_prune", &HxOptions::poa_prune, nullptr}, {"poa_prune_lanes", &HxOptions::poa_prune_lanes, nullptr}, {"poa_prune_lazy", &HxOptions::poa_prune_lazy, nullptr}, {"poa_prune_shared", &HxOptions::poa_prune_shared, nullptr}, {"poa_pass_lanes", &HxOptions::poa_pass_lanes, nullptr}, {"poa_chain_ms", &HxOptions::poa_chain_ms, nullptr}, {"poa_chain_pct", &HxOptions::poa_chain_pct, nullptr}, {"poa_scratch_warm", &HxOptions::poa_scratch_warm, nullptr}, {"poa_big_first", &HxOptions::poa_big_first, nullptr}, {"poa_order_by_cells", &HxOptions::poa_order_by_cells, nullptr}, {"poa_slots_by_work", &HxOptions::poa_slots_by_work, nullptr}, {"poa_resident_first", &HxOptions::poa_resident_first, nullptr}, {"poa_own_bucket_first", &HxOptions::poa_own_bucket_first, nullptr}, {"poa_bucket_half_octaves", &HxOptions::poa_bucket_half_octaves, nullptr}, {"coords_lds_supp", &HxOptions::coords_lds_supp, nullptr},
};
}  // namespace

struct hx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // resident inputs
    uint32_t n_contigs = 0, n_reads = 0;
    uint64_t n_hits = 0, n_ops = 0;
    DV<double> km;
    DV<uint32_t> clen;
    DV<uint8_t> cls;
    DV<uint32_t> rlen;
    DV<uint64_t> roff;
    DV<uint8_t> packed;
    DV<uint32_t> q_id, q_start, q_end, t_id, t_len, t_start, t_end, n_match, n_block, cg_ops;
    DV<uint8_t> is_rev, mapq;
    DV<uint64_t> cg_off, rho;
    std::vector<uint32_t> h_rlen;
    std::vector<uint64_t> h_rho;
    uint32_t lr_begin = 0, lr_end = 0;
    bool prefiltered = false;   // the resident records are the filtered set of an index.longread
    DV<uint32_t> err;
    // chain results
    DV<uint32_t> c_hit, c_qs, c_qe, c_ts, c_te, c_nm, c_nb, c_skf, c_skb, c_cmp;
    DV<uint64_t> c_cb, c_ce, aln_off, cmp_off;
    uint64_t n_aln = 0, n_cmp = 0;
    bool have_chain = false;
    // edge records
    RecBuf rec_un, rec;   // unsorted (emission order) and sorted
    uint64_t n_rec_un = 0, n_rec = 0, n_edge = 0;
    DV<uint64_t> edge_key, edge_off;
    std::vector<uint64_t> h_edge_key, h_edge_off;
    bool have_edges = false;
    // coords results
    DV<uint32_t> k_head_end, k_tail_beg, k_supp_lr, k_spos, k_epos;
    std::vector<uint64_t> h_supp_off;
    std::vector<uint32_t> h_supp_lr, h_spos, h_epos;
    uint32_t n_sel = 0;
    bool have_coords = false;
    uint32_t dbg_slowest = 0;
    std::vector<uint32_t> dbg_lmax, dbg_nseq;
    std::vector<uint8_t> dbg_cls; uint32_t dbg_ring[11] = {};
    std::vector<uint32_t> dbg_shape;   // per edge: lanes of its workgroup | column passes << 16 | members << 24
    bool poa_no_dir = false;   // diagnostics: force the score-matrix traceback
    int poa_block = 0;   // 0 = automatic (lanes per edge chosen from the gap length)
    hipStream_t poa_streams[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint32_t* poa_started = nullptr;    // 16 words of mapped host memory: workgroups that have begun, per launch of a batch (kernels/poa.hip k_poa)
    hipEvent_t poa_ev[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    Timer tm;
    // scratch of the chain / edge / coordinate operators lives as long as the context too (grows, never shrinks): no allocation, free or
    // synchronisation for temporaries in a call once the sizes have been seen
    hxk::Workspace ws;
    struct {
        DV<uint32_t> hit, qs, qe, ts, te, nm, nb, skf, skb, dp, cmp, naln, ncmp;
        DV<uint64_t> cb, ce;
        DV<int32_t> from;
    } sc_chain;
    struct { DV<uint32_t> npairs, perm, perm_tmp, flag; DV<uint64_t> pair_off, key_tmp, fscan; } sc_edges;
    struct { DV<uint32_t> sel, nsupp, t_lr, t_sp, t_ep, best_list; DV<uint64_t> cap, out_off, b1, e1, b2, e2; DV<uint8_t> cur; } sc_coords;
    // POA workspace lives as long as the context: allocating tens of GB per call costs more than the kernel
    PoaPoolBufs poa_pools;
    PoaArena poa_arena;
    std::mutex poa_arena_mu;            // hx_poa_reserve may run on a thread of its own beside the upload and the first stages
    double poa_host_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // host wall time of the last consensus call: plan, workspace, enqueue, device wait, collect, finish, (unused), total
    uint64_t poa_budget = 0;
    DV<hxk::PoaEdge> poa_edges;
    DV<hxk::PoaSeq> poa_seqs;
    DV<uint32_t> poa_order, poa_len, poa_status, poa_counters, poa_btab;
    DV<hxk::PoaSlot> poa_slots;
    uint64_t poa_workspace_bytes = 0;   // largest POA workspace (pools) a call of this context has used
    uint64_t poa_last_workspace_bytes = 0, poa_free_at_first_call = 0;   // ... the last call's; free device memory when the budget was taken
    HxOptions opt;
    DV<uint32_t> poa_gather;            // collection: (source offset lo / hi, destination offset lo / hi, length) of every finished edge's consensus
    DV<char> poa_cns_dense;             // ... the strings side by side, as they are downloaded
    DV<unsigned long long> poa_phase_d, poa_cells_d;
    std::vector<unsigned long long> poa_phase;   // per edge x 6, cycles of the last hx_poa_batch

    DevHits hits_view() const {
        return DevHits{n_hits, q_id.p, q_start.p, q_end.p, t_id.p, t_len.p, t_start.p, t_end.p, n_match.p, n_block.p, is_rev.p, mapq.p, cg_off.p, cg_ops.p};
    }
    hxk::ChainFinal chain_view() { return hxk::ChainFinal{c_hit.p, c_qs.p, c_qe.p, c_ts.p, c_te.p, c_nm.p, c_nb.p, c_skf.p, c_skb.p, c_cb.p, c_ce.p, c_cmp.p}; }
    void tick() { (void)hipEventRecord(tm.a, stream); }
    void tock(int k) {
        (void)hipEventRecord(tm.b, stream);
        (void)hipEventSynchronize(tm.b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, tm.a, tm.b);
        tm.ms[k] += ms; tm.launches[k]++;
    }
};

// Every k_poa instance uses private memory (288 to 928 bytes per lane: spills, a by-value argument), and a hardware queue grows its scratch when a
// dispatch asks for more per wave than the queue has had - a trip through the runtime (an allocation of device memory: slow while the driver is still
// wiping what another process freed) that holds THAT launch back. In a process that has run the few-edge instances, the first many-edge call then had
// some of its launches held and others not, they reached the CUs in another order, and the call took 600-640 ms instead of 415-440 (tools/dev_cold.py:
// a 12 Mb context, then the 140 Mb one; bench.py's configs[3] leg: five runs of five). Once per process and device, every stream of the pool runs one
// wave that asks for the most: from hx_poa_reserve (beside the parse) or, without a reservation, before the first launches.
static int scratch_warm_once(hx_ctx* c) {
    static std::mutex warm_mu;
    static std::vector<char> warmed;
    std::lock_guard<std::mutex> lk(warm_mu);
    if ((int)warmed.size() <= c->device) warmed.resize((size_t)c->device + 1, 0);
    if (!c->opt.poa_scratch_warm || warmed[(size_t)c->device]) return 0;
    for (int i = 0; i < 8; i++) hxk::scratch_warm(c->poa_streams[i]);
    for (int i = 0; i < 8; i++) HIPCHK(hipStreamSynchronize(c->poa_streams[i]));
    warmed[(size_t)c->device] = 1;
    return 0;
}

extern "C" const char* hx_last_error(void) { return g_err.c_str(); }

extern "C" int hx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return fail("hipGetDeviceCount failed: no usable HIP device (the HIP path has no CPU fallback)");
    return n;
}

extern "C" int hx_ctx_create(int device, void* stream, hx_ctx** out) {
    *out = nullptr;
    // (the POA launch classes go to separate streams and overlap only with enough hardware queues: the APPLICATION sets GPU_MAX_HW_QUEUES >= 8
    //  before HIP initialises - haslr_assemble, haslr_amd/hip.py and bench.py do; the library does not touch the process environment)
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail("hx_ctx_create: no HIP device available (libhaslr_hip.so has no CPU fallback)");
    if (device < 0 || device >= n) return fail("hx_ctx_create: device index out of range");
    HIPCHK(hipSetDevice(device));
    hx_ctx* c = new hx_ctx;
    c->device = device;
    if (stream) c->stream = (hipStream_t)stream;
    else { HIPCHK(hipStreamCreate(&c->stream)); c->own_stream = true; }
    HIPCHK(hipEventCreate(&c->tm.a));
    HIPCHK(hipEventCreate(&c->tm.b));
    {   // distinct priorities map to distinct hardware queues, so the lane-count classes really run side by side.
        // Round 6: the eight streams of the launch classes are the PROCESS's, one set per device, created by the first context on it and handed to every later
        // one (never destroyed). A second context of a process used to run the 12 Mb step 13 % slower, whichever path it was - streams created when
        // the first context's had already taken the hardware queues do not get queues of their own, and destroying the first context did not give them back.
        static std::mutex pool_mu;
        static std::vector<std::array<hipStream_t, 8>> pool;   // by device
        std::lock_guard<std::mutex> lk(pool_mu);
        if ((int)pool.size() <= device) pool.resize((size_t)device + 1, std::array<hipStream_t, 8>{});
        if (!pool[(size_t)device][0]) {
            int lo = 0, hi = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));   // lo = least priority (largest number), hi = greatest
            for (int i = 0; i < 8; i++) {
                int pr = hi + i; if (pr > lo) pr = lo;
                HIPCHK(hipStreamCreateWithPriority(&pool[(size_t)device][(size_t)i], hipStreamNonBlocking, pr));
            }
        }
        for (int i = 0; i < 8; i++) c->poa_streams[i] = pool[(size_t)device][(size_t)i];
    }
    for (int i = 0; i < 9; i++) HIPCHK(hipEventCreateWithFlags(&c->poa_ev[i], hipEventDisableTiming));
    HIPCHK(hipHostMalloc((void**)&c->poa_started, 16 * sizeof(uint32_t), hipHostMallocMapped));
    HIPCHK(c->err.reserve(1));
    *out = c;
    return 0;
}

extern "C" void hx_ctx_destroy(hx_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->tm.a) (void)hipEventDestroy(c->tm.a);
    if (c->tm.b) (void)hipEventDestroy(c->tm.b);
    for (int i = 0; i < 9; i++) if (c->poa_ev[i]) (void)hipEventDestroy(c->poa_ev[i]);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    if (c->poa_started) (void)hipHostFree(c->poa_started);
    delete c;
}

template <class T>
static int up(DV<T>& d, const T* h, size_t n) {
    HIPCHK(d.reserve(n));
    if (n) HIPCHK(hipMemcpy(d.p, h, n * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

static int upload_inputs(hx_ctx* c, const hx_contigs* ctg, const hx_reads* rd, const hx_hits* h, const uint64_t* rho);
extern "C" int hx_upload(hx_ctx* c, const hx_contigs* ctg, const hx_reads* rd, const hx_hits* h, const uint64_t* rho) {
    if (upload_inputs(c, ctg, rd, h, rho) == 0) return 0;
    // a consensus arena reserved ahead of the inputs (hx_poa_reserve) must never be what keeps them out: give it back and try once more
    {
        std::lock_guard<std::mutex> lk(c->poa_arena_mu);
        if (!c->poa_arena.cap) return -1;
        (void)hipGetLastError();
        c->poa_arena.release();
    }
    return upload_inputs(c, ctg, rd, h, rho);
}
static int upload_inputs(hx_ctx* c, const hx_contigs* ctg, const hx_reads* rd, const hx_hits* h, const uint64_t* rho) {
    HIPCHK(hipSetDevice(c->device));
    c->n_contigs = ctg->n; c->n_reads = rd->n; c->n_hits = h->n; c->n_ops = h->cg_off[h->n];
    if (up(c->km, ctg->mean_kmer, ctg->n) || up(c->clen, ctg->len, ctg->n)) return -1;
    HIPCHK(c->cls.reserve(ctg->n));
    if (up(c->rlen, rd->len, rd->n) || up(c->roff, rd->off, (size_t)rd->n + 1) || up(c->packed, rd->packed, (size_t)rd->off[rd->n])) return -1;
    size_t n = h->n;
    if (up(c->q_id, h->q_id, n) || up(c->q_start, h->q_start, n) || up(c->q_end, h->q_end, n) || up(c->t_id, h->t_id, n) || up(c->t_len, h->t_len, n) ||
        up(c->t_start, h->t_start, n) || up(c->t_end, h->t_end, n) || up(c->n_match, h->n_match, n) || up(c->n_block, h->n_block, n) ||
        up(c->is_rev, h->is_rev, n) || up(c->mapq, h->mapq, n) || up(c->cg_off, h->cg_off, n + 1) || up(c->cg_ops, h->cg_ops, (size_t)c->n_ops) ||
        up(c->rho, rho, (size_t)rd->n + 1)) return -1;
    c->h_rlen.assign(rd->len, rd->len + rd->n);
    c->h_rho.assign(rho, rho + rd->n + 1);
    c->lr_begin = 0; c->lr_end = rd->n;
    c->prefiltered = false;
    c->have_chain = c->have_edges = c->have_coords = false;
    return 0;
}

extern "C" void hx_set_prefiltered(hx_ctx* c, int on) { c->prefiltered = on != 0; }

// option names: lower case, as in kOptions; the old environment variable spellings (HX_POA_SLOTS, HX_DEBUG ...) are accepted too
static std::string opt_key(const char* name) {
    std::string k(name ? name : "");
    for (char& ch : k) ch = (char)tolower((unsigned char)ch);
    if (k.rfind("hx_", 0) == 0) k = k.substr(3);
    if (k == "prof1" || k == "prof2" || k == "prof3") return k;
    return k;
}
extern "C" int hx_set_option(hx_ctx* c, const char* name, const char* value) {
    const std::string k = opt_key(name);
    const HxOptions dflt;
    const bool reset = !value || !*value;
    if (k == "prof1" || k == "prof2" || k == "prof3") { c->opt.prof = reset ? 0 : k[4] - '0'; return 0; }   // (HX_PROF1 / 2 / 3 of the development builds)
    for (const OptDesc& d : kOptions)
        if (k == d.name) {
            char* end = nullptr;
            if (d.ip) { const long v = reset ? dflt.*(d.ip) : strtol(value, &end, 10); if (!reset && (end == value || *end)) return fail(std::string("hx_set_option: ") + d.name + " takes an integer, not '" + value + "'"); c->opt.*(d.ip) = (int)v; }
            else { const double v = reset ? dflt.*(d.dp) : strtod(value, &end); if (!reset && (end == value || *end)) return fail(std::string("hx_set_option: ") + d.name + " takes a number, not '" + value + "'"); c->opt.*(d.dp) = v; }
            return 0;
        }
    return fail("hx_set_option: unknown option '" + std::string(name ? name : "") + "' (hx_option_names lists them)");
}
extern "C" const char* hx_option_names(void) {
    static const std::string names = [] { std::string n; for (const OptDesc& d : kOptions) { if (!n.empty()) n += ","; n += d.name; } return n; }();
    return names.c_str();
}
extern "C" int hx_get_option(const hx_ctx* c, const char* name, double* value) {
    const std::string k = opt_key(name);
    for (const OptDesc& d : kOptions) if (k == d.name) { *value = d.ip ? (double)(c->opt.*(d.ip)) : c->opt.*(d.dp); return 0; }
    return fail("hx_get_option: unknown option '" + std::string(name ? name : "") + "'");
}

extern "C" int hx_set_read_shard(hx_ctx* c, uint32_t b, uint32_t e) {
    if (b > e || e > c->n_reads) return fail("hx_set_read_shard: bad range");
    c->lr_begin = b; c->lr_end = e;
    return 0;
}

// the scan / sort launchers skip their work when their scratch cannot be allocated (Workspace::oom): nothing they were to write may be read
static int ws_ok(hx_ctx* c, const char* who) {
    if (!c->ws.oom) return 0;
    c->ws.oom = false;
    return fail(std::string(who) + ": out of device memory for scan / sort scratch");
}

static int check_err(hx_ctx* c, const char* who) {
    if (ws_ok(c, who)) return -1;
    uint32_t e = 0;
    HIPCHK(hipMemcpy(&e, c->err.p, 4, hipMemcpyDeviceToHost));
    if (!e) return 0;
    std::string m = std::string(who) + ":";
    if (e & HXE_TRIM_NO_M) m += " overlap trim ran off an alignment without M;";
    if (e & HXE_CHAIN_TOO_MANY) m += " more than 10000 chainable hits on one read (reference limit, Longread.cpp:529);";
    if (e & HXE_SPOS_RANGE) m += " consensus support starts beyond its read;";
    if (e & HXE_BAD_TID) m += " contig id out of range;";
    return fail(m);
}

// ================================================================================================ K1-K3
extern "C" int hx_chain_reads(hx_ctx* c, const hx_params* prm, hx_chain_out* out) {
    memset(out, 0, sizeof(*out));
    HIPCHK(hipSetDevice(c->device));
    const uint32_t nr = c->lr_end - c->lr_begin;
    const uint64_t nraw = c->h_rho[c->lr_end] - c->h_rho[c->lr_begin];
    hipStream_t s = c->stream;
    HIPCHK(hipMemsetAsync(c->err.p, 0, 4, s));
    const double thr_load = prm->uniq_freq * (3 + prm->max_uniq_dev), thr_uniq = prm->uniq_freq * (1 + prm->max_uniq_dev);
    // scratch at raw-hit granularity
    auto& S = c->sc_chain;
    DV<uint32_t>&s_hit = S.hit, &s_qs = S.qs, &s_qe = S.qe, &s_ts = S.ts, &s_te = S.te, &s_nm = S.nm, &s_nb = S.nb, &s_skf = S.skf, &s_skb = S.skb, &s_dp = S.dp, &s_cmp = S.cmp,
                &s_naln = S.naln, &s_ncmp = S.ncmp;
    DV<uint64_t>&s_cb = S.cb, &s_ce = S.ce;
    DV<int32_t>& s_from = S.from;
    c->ws.reset(s);
    HIPCHK(s_hit.reserve(nraw)); HIPCHK(s_qs.reserve(nraw)); HIPCHK(s_qe.reserve(nraw)); HIPCHK(s_ts.reserve(nraw)); HIPCHK(s_te.reserve(nraw));
    HIPCHK(s_nm.reserve(nraw)); HIPCHK(s_nb.reserve(nraw)); HIPCHK(s_skf.reserve(nraw)); HIPCHK(s_skb.reserve(nraw)); HIPCHK(s_dp.reserve(nraw));
    HIPCHK(s_cmp.reserve(nraw)); HIPCHK(s_cb.reserve(nraw)); HIPCHK(s_ce.reserve(nraw)); HIPCHK(s_from.reserve(nraw));
    HIPCHK(s_naln.reserve(nr)); HIPCHK(s_ncmp.reserve(nr));
    hxk::ChainScratch sc{s_hit.p, s_qs.p, s_qe.p, s_ts.p, s_te.p, s_nm.p, s_nb.p, s_skf.p, s_skb.p, s_cb.p, s_ce.p, s_dp.p, s_from.p, s_cmp.p, s_naln.p, s_ncmp.p};
    HIPCHK(c->aln_off.reserve((size_t)nr + 1)); HIPCHK(c->cmp_off.reserve((size_t)nr + 1));
    c->tick();
    hxk::contig_class(c->km.p, c->n_contigs, thr_load, thr_uniq, c->cls.p, s);
    hxk::chain_reads(c->hits_view(), c->rho.p, c->cls.p, c->n_contigs, c->lr_begin, c->lr_end, prm->min_aln_block, prm->min_aln_sim, prm->min_aln_mapq, sc, c->err.p, c->prefiltered, s);
    hxk::exclusive_scan_u32(s_naln.p, c->aln_off.p, nr, s, c->ws);
    hxk::exclusive_scan_u32(s_ncmp.p, c->cmp_off.p, nr, s, c->ws);
    if (ws_ok(c, "hx_chain_reads")) return -1;
    uint64_t tot[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(&tot[0], c->aln_off.p + nr, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&tot[1], c->cmp_off.p + nr, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    c->n_aln = tot[0]; c->n_cmp = tot[1];
    if (c->n_aln >= 0xffffffffULL) return fail("hx_chain_reads: more than 2^32-1 alignments in one shard");
    HIPCHK(c->c_hit.reserve(c->n_aln)); HIPCHK(c->c_qs.reserve(c->n_aln)); HIPCHK(c->c_qe.reserve(c->n_aln)); HIPCHK(c->c_ts.reserve(c->n_aln));
    HIPCHK(c->c_te.reserve(c->n_aln)); HIPCHK(c->c_nm.reserve(c->n_aln)); HIPCHK(c->c_nb.reserve(c->n_aln)); HIPCHK(c->c_skf.reserve(c->n_aln));
    HIPCHK(c->c_skb.reserve(c->n_aln)); HIPCHK(c->c_cb.reserve(c->n_aln)); HIPCHK(c->c_ce.reserve(c->n_aln)); HIPCHK(c->c_cmp.reserve(c->n_cmp));
    hxk::chain_compact(sc, c->rho.p, c->lr_begin, c->lr_end, c->aln_off.p, c->cmp_off.p, c->chain_view(), s);
    c->tock(0);
    HIPCHK(hipGetLastError());
    if (check_err(c, "hx_chain_reads")) return -1;
    c->have_chain = true; c->have_edges = c->have_coords = false;
    // download
    out->n_aln = c->n_aln; out->n_reads = nr; out->n_cmp = c->n_cmp;
    out->hit = host_copy(c->c_hit.p, c->n_aln); out->q_start = host_copy(c->c_qs.p, c->n_aln); out->q_end = host_copy(c->c_qe.p, c->n_aln);
    out->t_start = host_copy(c->c_ts.p, c->n_aln); out->t_end = host_copy(c->c_te.p, c->n_aln); out->n_match = host_copy(c->c_nm.p, c->n_aln);
    out->n_block = host_copy(c->c_nb.p, c->n_aln); out->cg_begin = host_copy(c->c_cb.p, c->n_aln); out->cg_end = host_copy(c->c_ce.p, c->n_aln);
    out->cg_skip_front = host_copy(c->c_skf.p, c->n_aln); out->cg_skip_back = host_copy(c->c_skb.p, c->n_aln);
    out->read_off = host_copy(c->aln_off.p, (size_t)nr + 1); out->cmp_off = host_copy(c->cmp_off.p, (size_t)nr + 1); out->cmp_aln = host_copy(c->c_cmp.p, c->n_cmp);
    return 0;
}

extern "C" void hx_free_chain(hx_ctx*, hx_chain_out* o) {
    free(o->hit); free(o->q_start); free(o->q_end); free(o->t_start); free(o->t_end); free(o->n_match); free(o->n_block);
    free(o->cg_begin); free(o->cg_end); free(o->cg_skip_front); free(o->cg_skip_back); free(o->read_off); free(o->cmp_off); free(o->cmp_aln);
    memset(o, 0, sizeof(*o));
}

// ================================================================================================ K4
extern "C" int hx_edge_emit(hx_ctx* c, const hx_params*, uint64_t* n_records) {
    if (!c->have_chain) return fail("hx_edge_emit: hx_chain_reads has not run");
    HIPCHK(hipSetDevice(c->device));
    const uint32_t nr = c->lr_end - c->lr_begin;
    hipStream_t s = c->stream;
    DV<uint32_t>& npairs = c->sc_edges.npairs;
    DV<uint64_t>& pair_off = c->sc_edges.pair_off;
    c->ws.reset(s);
    HIPCHK(npairs.reserve(nr)); HIPCHK(pair_off.reserve((size_t)nr + 1));
    c->tick();
    hxk::edge_count(c->hits_view(), c->cls.p, c->chain_view(), c->cmp_off.p, c->lr_begin, c->lr_end, npairs.p, s);
    hxk::exclusive_scan_u32(npairs.p, pair_off.p, nr, s, c->ws);
    if (ws_ok(c, "hx_edge_emit")) return -1;
    uint64_t tot = 0;
    HIPCHK(hipMemcpyAsync(&tot, pair_off.p + nr, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    c->n_rec_un = 2 * tot;
    HIPCHK(c->rec_un.reserve(c->n_rec_un));
    hxk::edge_emit(c->hits_view(), c->cls.p, c->chain_view(), c->cmp_off.p, c->lr_begin, c->lr_end, pair_off.p, c->rec_un.view(), s);
    c->tock(1);
    HIPCHK(hipGetLastError());
    if (n_records) *n_records = c->n_rec_un;
    return 0;
}

static void download_side(hx_rec_side& h, DevSideBuf& d, size_t n) {
    h.q_start = host_copy(d.qs.p, n); h.q_end = host_copy(d.qe.p, n); h.t_start = host_copy(d.ts.p, n); h.t_end = host_copy(d.te.p, n);
    h.is_rev = host_copy(d.rev.p, n); h.cg_begin = host_copy(d.cb.p, n); h.cg_end = host_copy(d.ce.p, n);
    h.cg_skip_front = host_copy(d.skf.p, n); h.cg_skip_back = host_copy(d.skb.p, n);
}

// sort the n records sitting in rec_un by key (stable), segment into edges, download
static int finish_edges(hx_ctx* c, uint64_t n, hx_edges_out* out) {
    hipStream_t s = c->stream;
    if (n >= 0xffffffffULL) return fail("hx_edge_support: more than 2^32-1 edge-support records");
    int bits = 1;
    while (bits < 32 && (1ull << bits) < 2ull * std::max<uint32_t>(1, c->n_contigs)) bits++;
    DV<uint32_t>&perm = c->sc_edges.perm, &perm_tmp = c->sc_edges.perm_tmp, &flag = c->sc_edges.flag;
    DV<uint64_t>&key_tmp = c->sc_edges.key_tmp, &fscan = c->sc_edges.fscan;
    c->ws.reset(s);
    HIPCHK(perm.reserve(n)); HIPCHK(perm_tmp.reserve(n)); HIPCHK(flag.reserve(n)); HIPCHK(key_tmp.reserve(n)); HIPCHK(fscan.reserve(n + 1));
    HIPCHK(c->rec.reserve(n));
    c->tick();
    hxk::iota_u32(perm.p, n, s);
    hxk::radix_sort_pairs(c->rec_un.key.p, perm.p, key_tmp.p, perm_tmp.p, n, bits, bits, s, c->ws);
    if (n) HIPCHK(hipMemcpyAsync(c->rec.key.p, c->rec_un.key.p, n * 8, hipMemcpyDeviceToDevice, s));
    hxk::edge_gather(c->rec_un.view(), perm.p, n, c->rec.view(), s);
    hxk::segment_flags(c->rec.key.p, n, flag.p, s);
    hxk::exclusive_scan_u32(flag.p, fscan.p, n, s, c->ws);
    if (ws_ok(c, "hx_edge_support")) return -1;
    uint64_t ne = 0;
    HIPCHK(hipMemcpyAsync(&ne, fscan.p + n, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(c->edge_key.reserve(ne)); HIPCHK(c->edge_off.reserve(ne + 1));
    if (n == 0) HIPCHK(hipMemsetAsync(c->edge_off.p, 0, 8, s));
    hxk::segment_scatter(c->rec.key.p, fscan.p, n, c->edge_key.p, c->edge_off.p, s);
    c->tock(1);
    HIPCHK(hipGetLastError());
    c->n_rec = n; c->n_edge = ne;
    c->h_edge_key.resize(ne); c->h_edge_off.resize(ne + 1);
    if (ne) HIPCHK(hipMemcpy(c->h_edge_key.data(), c->edge_key.p, ne * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(c->h_edge_off.data(), c->edge_off.p, (ne + 1) * 8, hipMemcpyDeviceToHost));
    c->have_edges = true; c->have_coords = false;
    memset(out, 0, sizeof(*out));
    out->n_rec = n; out->n_edge = ne;
    out->key = host_copy(c->rec.key.p, n); out->lr = host_copy(c->rec.lr.p, n); out->cmp_head = host_copy(c->rec.ch.p, n); out->cmp_tail = host_copy(c->rec.ct.p, n);
    download_side(out->head, c->rec.head, n); download_side(out->tail, c->rec.tail, n);
    out->edge_key = (uint64_t*)malloc(std::max<size_t>(1, ne) * 8); memcpy(out->edge_key, c->h_edge_key.data(), ne * 8);
    out->edge_off = (uint64_t*)malloc((ne + 1) * 8); memcpy(out->edge_off, c->h_edge_off.data(), (ne + 1) * 8);
    return 0;
}

extern "C" int hx_edge_support(hx_ctx* c, const hx_params* prm, hx_edges_out* out) {
    memset(out, 0, sizeof(*out));
    uint64_t n = 0;
    if (hx_edge_emit(c, prm, &n)) return -1;
    return finish_edges(c, n, out);
}

extern "C" uint32_t hx_edge_records_bytes(void) { return hxk::EDGE_REC_WORDS * 4; }

extern "C" int hx_edge_records_export(hx_ctx* c, void* dst, uint64_t cap) {
    if (cap < c->n_rec_un) return fail("hx_edge_records_export: destination too small");
    HIPCHK(hipSetDevice(c->device));
    hxk::edge_pack(c->rec_un.view(), c->n_rec_un, (uint32_t*)dst, c->stream);
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int hx_edge_records_import(hx_ctx* c, const void* src, uint64_t n, hx_edges_out* out) {
    memset(out, 0, sizeof(*out));
    if (n & 1) return fail("hx_edge_records_import: records come in (forward, twin) pairs, the count must be even");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(c->rec_un.reserve(n));
    hxk::edge_unpack((const uint32_t*)src, n, c->rec_un.view(), c->stream);
    c->n_rec_un = n;
    return finish_edges(c, n, out);
}

static void free_side(hx_rec_side& s) {
    free(s.q_start); free(s.q_end); free(s.t_start); free(s.t_end); free(s.is_rev); free(s.cg_begin); free(s.cg_end); free(s.cg_skip_front); free(s.cg_skip_back);
}
extern "C" void hx_free_edges(hx_ctx*, hx_edges_out* o) {
    free(o->key); free(o->lr); free(o->cmp_head); free(o->cmp_tail); free_side(o->head); free_side(o->tail); free(o->edge_key); free(o->edge_off);
    memset(o, 0, sizeof(*o));
}

// ================================================================================================ K5
extern "C" int hx_edge_coords(hx_ctx* c, uint32_t n_sel, const uint32_t* sel, hx_coords_out* out) {
    memset(out, 0, sizeof(*out));
    if (!c->have_edges) return fail("hx_edge_coords: hx_edge_support has not run");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    std::vector<uint64_t> cap_off((size_t)n_sel + 1, 0);
    for (uint32_t i = 0; i < n_sel; i++) {
        if (sel[i] >= c->n_edge) return fail("hx_edge_coords: edge index out of range");
        uint64_t key = c->h_edge_key[sel[i]], n = c->h_edge_off[sel[i] + 1] - c->h_edge_off[sel[i]];
        bool hairpin = (((uint32_t)key) ^ 1u) == (uint32_t)(key >> 32);
        cap_off[i + 1] = cap_off[i] + (hairpin ? 2 * n : n);
    }
    const uint64_t cap = cap_off[n_sel];
    auto& K = c->sc_coords;
    DV<uint32_t>&d_sel = K.sel, &d_nsupp = K.nsupp, &t_lr = K.t_lr, &t_sp = K.t_sp, &t_ep = K.t_ep, &best_list = K.best_list;
    DV<uint64_t>&d_cap = K.cap, &d_out_off = K.out_off, &b1 = K.b1, &e1 = K.e1, &b2 = K.b2, &e2 = K.e2;
    DV<uint8_t>& cur = K.cur;
    c->ws.reset(s);
    HIPCHK(d_sel.reserve(n_sel)); HIPCHK(d_nsupp.reserve(n_sel)); HIPCHK(t_lr.reserve(cap)); HIPCHK(t_sp.reserve(cap)); HIPCHK(t_ep.reserve(cap));
    HIPCHK(best_list.reserve(cap)); HIPCHK(d_cap.reserve((size_t)n_sel + 1)); HIPCHK(d_out_off.reserve((size_t)n_sel + 1));
    HIPCHK(b1.reserve(cap)); HIPCHK(e1.reserve(cap)); HIPCHK(b2.reserve(cap)); HIPCHK(e2.reserve(cap)); HIPCHK(cur.reserve(cap));
    HIPCHK(c->k_head_end.reserve(n_sel)); HIPCHK(c->k_tail_beg.reserve(n_sel));
    if (n_sel) HIPCHK(hipMemcpyAsync(d_sel.p, sel, (size_t)n_sel * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_cap.p, cap_off.data(), ((size_t)n_sel + 1) * 8, hipMemcpyHostToDevice, s));
    hxk::CoordsScratch sc{b1.p, e1.p, b2.p, e2.p, cur.p, nullptr, nullptr, best_list.p};
    c->tick();
    hxk::edge_coords(c->rec.view(), c->edge_key.p, c->edge_off.p, c->cg_ops.p, c->clen.p, c->rlen.p, n_sel, d_sel.p, d_cap.p, sc,
                     c->k_head_end.p, c->k_tail_beg.p, d_nsupp.p, t_lr.p, t_sp.p, t_ep.p, c->opt.coords_lds_supp, s);
    hxk::exclusive_scan_u32(d_nsupp.p, d_out_off.p, n_sel, s, c->ws);
    if (ws_ok(c, "hx_edge_coords")) return -1;
    uint64_t tot = 0;
    HIPCHK(hipMemcpyAsync(&tot, d_out_off.p + n_sel, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(c->k_supp_lr.reserve(tot)); HIPCHK(c->k_spos.reserve(tot)); HIPCHK(c->k_epos.reserve(tot));
    hxk::coords_compact(d_cap.p, d_out_off.p, n_sel, t_lr.p, t_sp.p, t_ep.p, c->k_supp_lr.p, c->k_spos.p, c->k_epos.p, s);
    c->tock(2);
    HIPCHK(hipGetLastError());
    c->n_sel = n_sel;
    c->h_supp_off.resize((size_t)n_sel + 1); c->h_supp_lr.resize(tot); c->h_spos.resize(tot); c->h_epos.resize(tot);
    HIPCHK(hipMemcpy(c->h_supp_off.data(), d_out_off.p, ((size_t)n_sel + 1) * 8, hipMemcpyDeviceToHost));
    if (tot) {
        HIPCHK(hipMemcpy(c->h_supp_lr.data(), c->k_supp_lr.p, tot * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(c->h_spos.data(), c->k_spos.p, tot * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(c->h_epos.data(), c->k_epos.p, tot * 4, hipMemcpyDeviceToHost));
    }
    c->have_coords = true;
    out->n_edge = n_sel;
    out->head_end = host_copy(c->k_head_end.p, n_sel); out->tail_beg = host_copy(c->k_tail_beg.p, n_sel);
    out->supp_off = (uint64_t*)malloc(((size_t)n_sel + 1) * 8); memcpy(out->supp_off, c->h_supp_off.data(), ((size_t)n_sel + 1) * 8);
    out->supp_lr = (uint32_t*)malloc(std::max<size_t>(1, tot) * 4); memcpy(out->supp_lr, c->h_supp_lr.data(), tot * 4);
    out->spos = (uint32_t*)malloc(std::max<size_t>(1, tot) * 4); memcpy(out->spos, c->h_spos.data(), tot * 4);
    out->epos = (uint32_t*)malloc(std::max<size_t>(1, tot) * 4); memcpy(out->epos, c->h_epos.data(), tot * 4);
    return 0;
}

extern "C" void hx_free_coords(hx_ctx*, hx_coords_out* o) {
    free(o->head_end); free(o->tail_beg); free(o->supp_off); free(o->supp_lr); free(o->spos); free(o->epos);
    memset(o, 0, sizeof(*o));
}

// ================================================================================================ K6

namespace {
// what the POA stage reads: the supports of every edge (the cns_supp lists of Assemble.cpp:503-543) and the read set they point into
struct PoaInput {
    uint32_t n_edge;
    const uint64_t* supp_off;
    const uint32_t *supp_lr, *spos, *epos;
    const uint32_t* h_rlen;       // host copy of the read lengths
    const uint8_t* d_packed;      // device: 2-bit reads, their byte offsets and lengths
    const uint64_t* d_roff;
    const uint32_t* d_rlen;
};

struct Need { uint64_t nn = 0, ec = 0, hc = 0, dc = 0, wc = 0, lm = 0, st = 0, al = 0, mb = 0; };
inline void need_max(Need& a, const Need& b) {
    a.nn = std::max(a.nn, b.nn); a.ec = std::max(a.ec, b.ec); a.hc = std::max(a.hc, b.hc); a.dc = std::max(a.dc, b.dc); a.wc = std::max(a.wc, b.wc);
    a.lm = std::max(a.lm, b.lm); a.st = std::max(a.st, b.st); a.al = std::max(a.al, b.al); a.mb = std::max(a.mb, b.mb);
}
inline uint64_t need_bytes(const Need& n) { return n.nn * 106 + n.ec * 28 + n.hc * 4 + n.dc + n.wc + n.lm + n.st * 4 + n.al * 8 + n.mb * 8; }

// launch classes: (shared?, lanes per workgroup, columns per lane, traceback flavour) - one kernel instance each, so that every
// launch runs with the registers ITS row loop needs (kernels/poa.hip)
struct Cls {
    bool shared; uint32_t nt, cm; bool dir;
    uint32_t dpl = 0;   // lanes in the DP when the workgroups are wider (wide cluster members), else 0
    uint32_t pb = 0;    // unshared edges of a call with column passes: bucket of their workspace need (log2 of the megabytes) - one slot size per bucket, all buckets of a kernel instance in ONE launch
    bool pk = false;    // ... the pruned instance whatever the lanes (edges that take their columns in several passes are among the class's)
    std::vector<uint32_t> edges;
    size_t blocks = 0, order_at = 0, slot_at = 0, n_slots = 0;
    Need need{};
    bool persistent = false;
    double share = 0;   // of the batch's wave-slot time: DP rows x lanes reserved
};

constexpr int NCLS = 11;
constexpr uint64_t kPoaLdsMax = 140 * 1024;   // dynamic LDS of a POA workgroup at most (160 KB per CU less the 1024-lane kernel's static 16.5 KB: sink lists, wave mailboxes)
const int kClassNT[NCLS] = {0, 1024, 512, 256, 128, 64, 1024, 512, 256, 128, 64};
constexpr size_t kManyEdges = 3000;

// One consensus call: the PLAN (sub-sequences, per-edge capacities, launch classes, workspace slots and batches against the memory budget), the
// LAUNCH of a batch, and the COLLECTION of its results with the verdict on every edge (done / again with more room / again another way).
struct PoaCall {
    hx_ctx* c;
    const PoaInput& in;
    const hx_poa_params* pp;
    const HxOptions& o;
    const uint32_t ne;
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    PoaPlan P;
    uint64_t seq_bases = 0, n_aligned = 0, budget = 0;
    struct CnsView { const char* p = nullptr; size_t n = 0; const char* data() const { return p; } size_t size() const { return n; } };
    std::vector<CnsView> cns;                            // per edge: where its finished consensus lies in ...
    std::vector<std::unique_ptr<char[]>> cns_blocks;     // ... the download of its batch (kept to the end of the call: no copy per edge, no zero fill)
    std::vector<uint8_t> grow;         // times an edge's graph outgrew its workspace: the node estimate doubles each time
    std::vector<uint8_t> force_nodir;  // edges whose in-degrees outgrew the direction bytes
    std::vector<uint8_t> full_h;       // edges that run with the score-matrix traceback
    std::vector<uint8_t> wide_grow;    // times an edge had more rows with over 4 predecessors than its wide-row pool: the estimate quadruples each time
    std::vector<uint8_t> no_share;     // edges whose members did not get through together: one workgroup from now on
    std::vector<uint8_t> many_sinks;   // edges with more sink rows than the smaller kernels keep in LDS: one 1024-lane workgroup
    std::vector<uint8_t> far_full;     // times an edge's far rows outgrew the estimate: four times the room each time
    std::vector<uint8_t> ecols;        // shared edges: columns per lane their members aim at (cl_cols, or 2 for the costliest: option poa_cols2_top)
    std::vector<uint32_t> mlanes;      // shared edges: lanes per member (the option's, or 1024 where the gap needs them to fit at all)
    // knobs of this round (the option, or what the number of edges in the call asks for)
    bool many_edges = false, balanced = false;
    uint32_t cl_lanes = 256, cl_min = 2048, cl_max = 16, cl_pref = 16, cl_topk = 192, wide_k = 0, cl_cols = 4, cols_per_lane = 4, wave_max = 512, prune_pct = 0, prune_shared_pct = 0, pass_lanes = 0;
    bool pass_on = false;
    std::vector<uint16_t> plane;       // unshared edges with column passes: lanes of their workgroup
    std::vector<uint8_t> batch_by_work; // per batch of the current plan: the slot policy plan_batches settled on
    std::vector<float> chain_ms;       // estimated duration of the edge's chain (size_edges): the order of the launch lists
    uint64_t ring_kb_wave = 0;
    double balance_f = 1.25;
    uint32_t balance_nt = 512;
    mutable bool by_work = false;      // slots of the need buckets in proportion to their work (arrange): chosen per batch by plan_batches - where the memory budget binds
    uint64_t score_abs_max = 8;        // largest |match|, |mismatch|, |gap| of the call: |score| <= that x (nodes + columns) must fit the keys

    PoaCall(hx_ctx* c_, const PoaInput& in_, const hx_poa_params* pp_) : c(c_), in(in_), pp(pp_), o(c_->opt), ne(in_.n_edge) {}
    double ms_since_start() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

    // ---- plan, part 1: the sub-sequence rule of Assemble.cpp:530-537 (u32 wrap + substr clamp; empty ones skipped)
    int plan_input(std::vector<uint32_t>& todo) {
        P.edges.resize(ne); P.sumL.assign(ne, 0); P.nseq.assign(ne, 0);
        // (330 000 sequences of 13 000 edges at 140 Mb, a read-length lookup each: counted and filled by a few threads, each its range of the edges - 3.6 ms on one)
        const uint64_t n_supp = ne ? in.supp_off[ne] - in.supp_off[0] : 0;
        const uint32_t nt = n_supp > 50000 ? 8u : 1u;
        std::vector<uint64_t> t_seqs(nt + 1, 0), t_bases(nt, 0);
        std::atomic<int> bad{0};
        auto range = [&](uint32_t t, bool fill) {
            const uint32_t e0 = (uint32_t)((uint64_t)ne * t / nt), e1 = (uint32_t)((uint64_t)ne * (t + 1) / nt);
            uint64_t at = fill ? t_seqs[t] : 0, bases = 0;
            for (uint32_t e = e0; e < e1; e++) {
                hxk::PoaEdge& E = P.edges[e];
                if (fill) { memset(&E, 0, sizeof(E)); E.seq_begin = (uint32_t)at; }
                uint32_t cnt = 0, lmax = 0; uint64_t sum = 0;
                for (uint64_t k = in.supp_off[e]; k < in.supp_off[e + 1]; k++) {
                    const uint32_t rid = in.supp_lr[k] & 0x7fffffffu, strand = in.supp_lr[k] >> 31;
                    const uint32_t rl = in.h_rlen[rid], sp = in.spos[k], ep = in.epos[k];
                    if (sp > rl) { bad = 1; return; }
                    const uint32_t want = ep - sp + 1, n = std::min(want, rl - sp);
                    if (n == 0) continue;
                    if (fill) P.seqs[at] = hxk::PoaSeq{rid, strand, sp, n};
                    at++; cnt++; sum += n; lmax = std::max(lmax, n);
                }
                if (fill) { E.seq_end = (uint32_t)at; E.lmax = lmax; P.sumL[e] = sum; P.nseq[e] = cnt; bases += sum; }
            }
            if (fill) t_bases[t] = bases; else t_seqs[t + 1] = at;   // (counting pass: `at` started at 0 - the range's own count)
        };
        auto all = [&](bool fill) {
            std::vector<std::thread> th;
            for (uint32_t t = 1; t < nt; t++) th.emplace_back(range, t, fill);
            range(0, fill);
            for (std::thread& x : th) x.join();
        };
        all(false);
        if (bad) return fail("hx_poa_batch: consensus support starts beyond its read (the reference would throw std::out_of_range, Assemble.cpp:530)");
        {   // the ranges' counts -> where each range's sequences begin
            uint64_t run = P.seqs.size();
            for (uint32_t t = 0; t < nt; t++) { const uint64_t n = t_seqs[t + 1]; t_seqs[t] = run; run += n; }
            t_seqs[nt] = run;
            P.seqs.resize(run);
        }
        all(true);
        for (uint32_t t = 0; t < nt; t++) seq_bases += t_bases[t];
        n_aligned += t_seqs[nt] - t_seqs[0];
        for (uint32_t e = 0; e < ne; e++) if (P.nseq[e]) todo.push_back(e);
        cns.assign(ne, CnsView{});
        grow.assign(ne, 0); force_nodir.assign(ne, 0); full_h.assign(ne, 0); wide_grow.assign(ne, 0); no_share.assign(ne, 0); many_sinks.assign(ne, 0); far_full.assign(ne, 0);
        mlanes.assign(ne, 0); plane.assign(ne, 0); chain_ms.assign(ne, 0.f); ecols.assign(ne, 4);
        return 0;
    }

    // ---- plan, part 2: the knobs of a round. Sharing an edge buys latency for that edge and costs throughput. Hundreds of edges (the longest is the
    // step): up to 16 members, the 192 costliest shared. Thousands (every CU busy anyway): 8 members (more only where a gap needs them to fit at
    // all), the 32 costliest - measured on 13 262 edges: 2.10 s with 16 x 192, 1.98 s with 8 x 32, 2.29 s without sharing (the largest edges then
    // run on after everything else has finished). (The two launch shapes cross between 2 200 and 3 300 edges: 292 against 313 ms at 2 214 edges,
    // 390-400 against 374 ms at 3 294.)
    int knobs(size_t n_todo) {
        const bool many_in = ne > kManyEdges;
        many_edges = n_todo > kManyEdges;
        cl_lanes = (uint32_t)o.poa_member_lanes; cl_min = (uint32_t)o.poa_cluster_min;
        cl_max = o.poa_cluster_max >= 0 ? (uint32_t)o.poa_cluster_max : 16;           // members per edge at most
        cl_pref = o.poa_cluster_max >= 0 ? cl_max : many_in ? 8 : 16;                 // ... unless the gap needs more to fit at all
        cl_topk = o.poa_cluster_topk >= 0 ? (uint32_t)o.poa_cluster_topk : many_in ? 32 : 192;   // shared edges per call at most (the costliest)
        wide_k = o.poa_wide_members >= 0 ? (uint32_t)o.poa_wide_members : 0;         // shared edges per call (the costliest) whose members are 1024-lane workgroups (default: size_edges)
        // columns per lane a member aims at: 4 while the longest edges set the duration; 8 in calls of thousands of edges - the 4-column instances take 145-158
        // registers, and ONE such wave on a SIMD leaves room for two waves of the 128-register instances instead of three: the 32 shared edges' 896 waves
        // held the whole chip at 3 200 resident waves of 4 096 while they ran (tools/dev_r05.sh edgedump: 3 870 with 8 columns, the call 705 -> 657 ms)
        cl_cols = o.poa_cluster_cols > 0 ? (uint32_t)o.poa_cluster_cols : many_in ? 8u : 4u;
        wave_max = (uint32_t)o.poa_wave_max;                                          // columns handled by ONE wavefront per edge
        // columns per lane of the multi-wave classes: 4 while edges are few (more lanes = a shorter row for the edges that set the step time),
        // 8 when thousands of edges keep every CU busy anyway (a row then costs fewer instructions in total: the per-row overhead is per wave).
        cols_per_lane = o.poa_cols > 0 ? (uint32_t)o.poa_cols : many_edges ? 8 : 4;
        // LDS of the kept-row ring. Few edges (their longest sets the duration): as many kept rows as fit, so that hardly any row is read back
        // from HBM. Thousands of edges (every CU busy): what counts is waves per SIMD - each wave spends most of its time waiting for its own
        // dependent instructions - so the ring is cut to `poa_ring_kb` per wave and several workgroups share a CU.
        ring_kb_wave = o.poa_ring_kb > 0 ? (uint64_t)o.poa_ring_kb : many_edges ? 11 : 0;   // 0 = no cut
        // Balanced launch for calls of thousands of edges: see build_classes / slots_wanted
        balanced = many_edges && o.poa_balance != 0;
        balance_f = std::max(10, o.poa_balance_pct) / 100.0;
        balance_nt = (uint32_t)o.poa_balance_lanes;                                   // classes of at least this many lanes per workgroup get a share
        // Exact score-bound pruning (kernels/poa.hip PRUNE) skips the (row, wave) blocks that cannot reach the alignment's score: work saved in the
        // multi-wave launches of a call whose CUs are all busy; a chain-bound call (hundreds of edges, the longest one is the step) gains nothing
        // from it - a row stays a row - so there the full-matrix instances run. poa_prune: -1 automatic, 0 never, else the percentage.
        // (round 6: automatic also in a few-edge call, for its unshared multi-wave classes and without the column passes - the step is the shared edges' and neither gains nor
        // loses, 164.98 against 165.54 ms at 12 Mb, but the dead wave-rows' nibble rows are not written)
        prune_pct = o.poa_prune < 0 ? 95u : (uint32_t)o.poa_prune;
        // The bound U = H + match x (columns left) and the lane test behind it are exact for scores of the usual signs only: gap <= 0, mismatch <= match,
        // gap <= match (hx_poa_sequences and spoa_hx.hpp take any int8 triple). Anything else runs the full-matrix instances.
        prune_shared_pct = (uint32_t)std::max(0, o.poa_prune_shared);
        if (!(pp->gap <= 0 && pp->mismatch <= pp->match && pp->gap <= pp->match && pp->match >= 0)) prune_pct = prune_shared_pct = 0;
        score_abs_max = std::max<uint64_t>({1, (uint64_t)std::abs((int)pp->match), (uint64_t)std::abs((int)pp->mismatch), (uint64_t)std::abs((int)pp->gap)});
        // Column passes (kernels/poa.hip): with the rows pruned, an edge's wave slots are mostly held by waves that skip - so the unshared multi-wave edges run
        // in workgroups of `pass_lanes` lanes and take their columns window by window. The call is bound by wave-slot time (thousands of edges, every slot
        // taken): an edge of 8 000 columns holds 4 waves instead of 16 for little more than the same time.
        pass_on = prune_pct != 0 && cols_per_lane <= 8 && o.poa_pass_lanes != 0 && (many_edges || o.poa_prune >= 0 || o.poa_pass_lanes > 0);   // (the automatic pruning of a few-edge call: without passes)
        pass_lanes = !pass_on || o.poa_pass_lanes < 0 ? 0u : (uint32_t)o.poa_pass_lanes;   // (0 with pass_on: by gap length, size_edges)
        if (pass_lanes != 0 && pass_lanes != 64 && pass_lanes != 128 && pass_lanes != 256 && pass_lanes != 512 && pass_lanes != 1024) return fail("option poa_pass_lanes must be 0, 64, 128, 256, 512 or 1024");
        if (cl_lanes != 64 && cl_lanes != 128 && cl_lanes != 256 && cl_lanes != 512 && cl_lanes != 1024) return fail("option poa_member_lanes must be 64, 128, 256, 512 or 1024");
        return 0;
    }

    // lanes per edge. Gaps up to 2047 bases: ONE wavefront per edge (row in registers, no barriers, many edges per CU).
    // Longer gaps: a multi-wave workgroup with ~8 columns per lane (256..1024 lanes). One launch per class, classes run concurrently.
    // Class 0 = edges shared by several workgroups (cluster members of cl_lanes lanes); classes 1..5 = one workgroup per edge.
    // Classes 6..10 = classes 1..5 for the edges that need the score-matrix traceback (rare: an in-degree
    // the direction bytes cannot hold, or the test switch), launched after their direction-byte twins on the same streams.
    int class_of(uint32_t e) const {   // launch class of an edge that is not shared (members == 1), direction-byte flavour
        static const uint32_t kMaxCm[6] = {0, 8, 16, 32, 32, 32};   // columns per lane each kernel variant keeps in registers
        const uint32_t ncol = P.edges[e].lmax + 1;
        int k = 5;
        if (many_sinks[e]) return 1;   // (the 1024-lane kernel keeps the full sink list)
        if (c->poa_block) { for (k = 1; k < 5 && kClassNT[k] > c->poa_block; k++) {} }
        else if (ncol > wave_max) { k = 4; while (k > 1 && (uint64_t)kClassNT[k] * cols_per_lane < ncol) k--; }
        if (P.edges[e].passes > 1) { for (int q = 1; q <= 5; q++) if ((uint32_t)kClassNT[q] == plane[e]) return q; }   // (column passes: a narrow workgroup, whatever the gap length)
        while (k > 1 && (uint64_t)kClassNT[k] * kMaxCm[k] < ncol) k--;
        return k;
    }
    uint32_t lanes_of(uint32_t e) const { return P.edges[e].members > 1 ? mlanes[e] : (uint32_t)kClassNT[class_of(e)]; }   // lanes of the edge's workgroup(s)
    static uint32_t cm_round(uint32_t ncol, uint32_t lanes, uint32_t r = 4) { const uint32_t cm = (ncol + lanes - 1) / lanes; while (r < cm) r <<= 1; return r; }   // (r: the narrowest instance that exists for the launch)
    // kept rows the LDS ring holds; row_bytes returns the LDS bytes of the ring
    uint32_t ring_rows_of(uint32_t nt, uint32_t cm, uint64_t& row_bytes) const {
        row_bytes = (uint64_t)cm * (nt / 64) * 65 * 4;   // planes of 65 words per wave
        uint64_t lds_budget = nt >= 1024 ? 128 * 1024 : nt == 64 ? 32 * 1024 : 64 * 1024 * (nt / 128 > 2 ? 2 : 1);
        if (ring_kb_wave) lds_budget = std::min<uint64_t>(lds_budget, std::max<uint64_t>(ring_kb_wave * 1024 * (nt / 64), 2 * row_bytes));
        if (2 * row_bytes > lds_budget) lds_budget = kPoaLdsMax;   // wide rows: whatever the CU has
        const uint64_t rows_fit = std::min<uint64_t>(lds_budget, kPoaLdsMax) / row_bytes;
        uint32_t R = rows_fit >= 8 ? 8 : rows_fit >= 4 ? 4 : rows_fit >= 2 ? 2 : 0;   // kept rows: a power of two (slot = kept-row counter & (R-1)); 0 = every kept row goes through HBM
        if (o.poa_ring_zero) R = 0;                                    // (testing: the ring-less mode that otherwise only gaps above 16 383 columns in ONE workgroup reach)
        row_bytes = row_bytes * std::max<uint32_t>(R, 1);              // -> LDS bytes (at least one row's worth: the kernel's other phases use the space too)
        return R;
    }
    // DP work of an edge ~ sum over its sequences of (nodes so far) x (length): with nodes growing linearly that is about half of
    // (final nodes) x (longest sequence) x (sequences). vcap < 2^21, lmax < 2^20, nseq < 2^24: no overflow
    uint64_t edge_cost(uint32_t e) const { return (uint64_t)P.edges[e].vcap * P.edges[e].lmax * std::max<uint32_t>(1, P.nseq[e]); }

    // DP rows of an edge's serial chain ~ the nodes of its graph before each sequence, summed (the model of the column passes: measured / model 1.10 .. 1.23). A call of
    // hundreds of edges ends when its last CHAIN ends, and a chain's duration goes with its rows, not with its cells: the order of the launch lists in such a call.
    double chain_rows(uint32_t e) const { const double S = std::max<uint32_t>(1, P.nseq[e]); return (double)P.edges[e].lmax * (S - 1.0) * (1.0 + 0.0275 * S) + (double)P.edges[e].lmax; }

    // ---- plan, part 3: per-edge capacities, members, far / wide row estimates; orders `todo` costliest first
    int size_edges(std::vector<uint32_t>& todo) {
        const uint64_t est_pct = (uint64_t)std::max(1, o.poa_node_est_pct);   // (testing: scales the node estimate)
        // ---- workspace sizes. Nodes of the finished graph: measured (nodes - L) / (L x sequences) on 13 %-error PacBio-like and 12 %-error
        // Nanopore-like reads is 0.05-0.06 (median), 0.07-0.08 (99th percentile, small edges). The estimate allows 0.09 plus a fifth of L
        // (a tighter one - 0.07 plus a twelfth - sent 7 of 13 230 edges of the 140 Mb data into a second attempt, which cost more than the memory was worth)
        // and doubles when a graph outgrows it, up to the proven bound (every base a node of its own).
        for (uint32_t e : todo) {
            hxk::PoaEdge& E = P.edges[e];
            if (E.lmax + 1 >= (1u << 20)) return fail("hx_poa_batch: a gap sub-sequence of " + std::to_string(E.lmax) + " bases is longer than the POA kernel's score keys hold (1 048 574)");
            const uint64_t est = std::max<uint64_t>(1, (((uint64_t)E.lmax * (120 + 9 * (uint64_t)P.nseq[e])) / 100 + 1024) * est_pct / 100) << std::min<uint32_t>(grow[e], 20);
            const uint64_t vc = std::max<uint64_t>(std::min<uint64_t>(P.sumL[e], est), E.lmax);   // (never below one sequence: per-base scratch shares the node pools)
            if (vc >= 0x7fffffffULL) return fail("hx_poa_batch: POA graph too large");
            // DP cells are keys = 64 x score + 6 tie-break bits in an int32: |score| <= 8 * (nodes + columns) must stay below 2^24
            if ((vc + E.lmax + 2) * score_abs_max >= (1ull << 24)) return fail("hx_poa_batch: POA graph of an edge exceeds 2^24 / " + std::to_string(score_abs_max) + " nodes + columns (score keys would overflow)");
            E.vcap = (uint32_t)vc; E.ecap = (uint32_t)(P.sumL[e] + P.nseq[e] + 1);
            // rows of H. The score-matrix traceback keeps every row; with direction bytes only rows that a successor reads after they left
            // the LDS ring go to HBM (about 1 row in 1000 on PacBio-like data): a sixteenth of the rows is the estimate, all of them the retry
            full_h[e] = c->poa_no_dir || force_nodir[e];   // (any number of sequences: the kernel reports an in-degree the direction bytes cannot hold, see max_indeg)
        }
        // (option poa_cols2_top, few-edge calls: the costliest edges that will be shared get members of 2 columns per lane - the kernel instances of the 256-lane
        // members and the 1024-lane wide members exist with 2 columns)
        for (uint32_t e : todo) ecols[e] = (uint8_t)cl_cols;
        // Measured (round 6, A/B in one GPU call): the longest 12 Mb edge's chain 159.6 -> 150 ms with 2 columns per lane (its row is ~28 instructions shorter, its
        // members twice as many); as the shape of ALL 192 shared edges the step got worse, 0.164 -> 0.195 s (twice the member waves crowd the chip: edges start
        // late); for the 4 costliest - the ones that get wide members - 0.164 -> 0.159 s (8: 0.161, 16: 0.164, 32: 0.175), 4.6 Mb 0.113 -> 0.110 s.
        const int cols2_top = o.poa_cols2_top >= 0 ? o.poa_cols2_top : (ne > kManyEdges ? 0 : 4);
        if (cols2_top > 0 && cl_lanes == 256) {
            std::vector<uint32_t> cand;
            for (uint32_t e : todo) if (P.edges[e].lmax + 1 > cl_min && !c->poa_block && !c->poa_no_dir && !force_nodir[e] && !many_sinks[e] && !no_share[e]) cand.push_back(e);
            std::sort(cand.begin(), cand.end(), [&](uint32_t a, uint32_t b) { const double ca = chain_rows(a), cb = chain_rows(b); return ca != cb ? ca > cb : a < b; });   // (the longest chains)
            for (size_t q = 0; q < cand.size() && q < (size_t)cols2_top; q++) ecols[cand[q]] = (uint8_t)(cl_cols >= 8 ? cl_cols / 2 : 2);   // (a many-edge call, whose members aim at 8 columns, when the option asks for it there: 4)
        }
        for (uint32_t e : todo) {
            hxk::PoaEdge& E = P.edges[e];
            // long gaps: the DP columns of the edge are shared by several workgroups (one CU each). Members of cl_lanes lanes x up to 32 columns per
            // lane x up to cl_max members hold 131 071 columns by default; a longer gap sub-sequence (the u32 wrap of Assemble.cpp:530 makes "the whole
            // tail of a read" a real case) gets 1024-lane members, 16 of which hold 524 287 columns.
            E.members = 1; mlanes[e] = cl_lanes;
            const uint32_t ncol = E.lmax + 1;
            const bool may_share = !c->poa_block && !c->poa_no_dir && !force_nodir[e] && !many_sinks[e] && !no_share[e];
            if (may_share && ncol > cl_min) {
                const uint64_t ecl = ecols[e], pref = ecl < cl_cols ? cl_max : cl_pref;   // (the longest chains, fewer columns per lane: as many members as that takes)
                auto members_for = [&](uint32_t lanes) -> uint64_t {
                    return std::min<uint64_t>(cl_max, std::max<uint64_t>(std::min<uint64_t>(pref, (ncol + (uint64_t)lanes * ecl - 1) / ((uint64_t)lanes * ecl)), (ncol + (uint64_t)lanes * 32 - 1) / ((uint64_t)lanes * 32)));
                };
                uint64_t mb = members_for(cl_lanes);
                if (((uint64_t)ncol + mb * cl_lanes - 1) / (mb * cl_lanes) > 32 && cl_lanes < 1024) { mlanes[e] = 1024; mb = members_for(1024); }   // (the gap does not fit the configured members)
                E.members = (uint32_t)mb;
            }
            if (E.members < 2 || ((uint64_t)ncol + (uint64_t)E.members * mlanes[e] - 1) / ((uint64_t)E.members * mlanes[e]) > 32) E.members = 1;   // (members too small for this gap: one workgroup)
            E.passes = 1;
            if (E.members == 1 && ncol > 1024u * (uint32_t)hxk::poa_kernel_max_cm(1024))
                return fail("hx_poa_batch: a gap sub-sequence of " + std::to_string(ncol - 1) + " bases needs the shared (cluster) mode - direction-byte traceback, automatic block size - with " +
                            std::to_string((ncol + 1024 * 32 - 1) / (1024 * 32)) + " members of 1024 lanes (option poa_cluster_max: " + std::to_string(cl_max) + ")");
        }
        // Sharing an edge among several CUs buys latency for the edge and costs throughput (the other members idle while member 0 walks
        // back and updates the graph). It pays while large edges are few; with many of them only the costliest keep their members.
        {
            std::vector<uint32_t> sh;
            for (uint32_t e : todo) if (P.edges[e].members > 1) sh.push_back(e);
            if (sh.size() > cl_topk) {
                std::sort(sh.begin(), sh.end(), [&](uint32_t a, uint32_t b) { const uint64_t ca = edge_cost(a), cb = edge_cost(b); return ca != cb ? ca > cb : a < b; });
                for (size_t q = cl_topk; q < sh.size(); q++) if (P.edges[sh[q]].lmax + 1 <= 8192) P.edges[sh[q]].members = 1;   // (longer gaps than a 1024-lane workgroup holds with its ring stay shared)
            }
        }
        // Column passes for the multi-wave edges that run unshared (decided here, after the costliest have kept their members). With the rows pruned a call of
        // thousands of edges is bound by WAVE-SLOT TIME: a 256-lane workgroup holds four wave slots of which the live band of the matrix keeps one or two
        // busy, and all four sit through the serial phases (traceback, graph update, CSR rebuild). Measured per DP row of an edge, under the load of such a
        // call (tools/dev_r05.sh edgedump, profiles/r05_edge_model.txt): a workgroup of 64 / 128 / 256 lanes takes 1.5 / 1.4 / 1.1 us with one window, 2.6 /
        // 1.9 / 1.4 us with two, 3.3 / 2.5 / 1.9 us with four - the narrowest workgroup is always the cheapest in wave-slot time (1.6 against 2.7 against 4.5
        // slot-us per row) and always the longest chain. So every edge gets the NARROWEST workgroup whose estimated chain stays below a cap, and the cap
        // is the one that balances the longest chain against the call's wave-slot time over the chip's slots (list scheduling: costliest first).
        if (pass_on) {
            // cycles per DP row at 2.4 GHz: the DP with 1 .. 4 windows (then per further window), everything else of the chain
            static const uint32_t kLanes[5] = {64, 128, 256, 512, 1024};
            // (the first dump's figures, 3 200 waves resident: they rank the widths as the final build's do - whose own table, taken where the long chains had been
            // given the wide workgroups, puts 128 lanes above 64 and moved the choices to 256 lanes: 0.58-0.68 s against 0.55 s - and overestimate its chains by a third)
            static const double kDp[5][4] = {{1470, 2900, 3800, 4435}, {1862, 2685, 3273, 3797}, {1741, 2430, 2900, 3150}, {1900, 2000, 2300, 2600}, {1850, 2000, 2200, 2400}};
            static const double kDpMore[5] = {500, 450, 250, 250, 200}, kRest[5] = {3300, 1900, 1000, 930, 900};
            struct Opt { double ms[5]; uint32_t np[5]; int first, last; };
            std::vector<uint32_t> ord;
            std::vector<Opt> opts;
            double fixed_slot_ms = 0;   // wave-slot time of the edges that have no choice (shared edges, one-wave gaps, score-matrix retries)
            for (uint32_t e : todo) {
                const hxk::PoaEdge& E = P.edges[e];
                const uint32_t ncol = E.lmax + 1, S = std::max<uint32_t>(1, P.nseq[e]);
                const double rows = 1.18 * (double)E.lmax * (double)(S - 1) * (1.0 + 0.0275 * S);   // nodes of the graph before each sequence, summed (measured / model: 1.10 .. 1.23)
                if (E.members > 1) { chain_ms[e] = (float)(rows * (1000 + 1000) / 2.4e6); fixed_slot_ms += chain_ms[e] * E.members * (mlanes[e] / 64); continue; }
                if (ncol <= wave_max || c->poa_block || full_h[e] || many_sinks[e]) { chain_ms[e] = (float)(rows * (1470 + 2200) / 2.4e6); fixed_slot_ms += chain_ms[e]; continue; }
                Opt q{}; q.first = -1; q.last = -1;
                for (int k = 0; k < 5; k++) {
                    const uint64_t win = (uint64_t)kLanes[k] * cols_per_lane;
                    const uint32_t np = (uint32_t)((ncol + win - 1) / win);
                    if (np > 64 || (pass_lanes && kLanes[k] != pass_lanes && np > 1)) continue;        // (option poa_pass_lanes: that width or the one that holds the gap)
                    q.np[k] = np; q.ms[k] = rows * ((np <= 4 ? kDp[k][np - 1] : kDp[k][3] + kDpMore[k] * (np - 4)) + kRest[k]) / 2.4e6;
                    if (q.first < 0) q.first = k;
                    q.last = k;
                    if (np == 1) break;                                                                // (wider than the gap: 4 columns per lane - not this model's)
                }
                if (q.first < 0) { chain_ms[e] = (float)(rows * 3000 / 2.4e6); continue; }
                ord.push_back(e); opts.push_back(q);
            }
            auto pick = [&](const Opt& q, double cap) { int k = q.first; while (k < q.last && (q.np[k] == 0 || q.ms[k] > cap)) k++; while (q.np[k] == 0) k--; return k; };
            double cap = o.poa_chain_ms > 0 ? (double)o.poa_chain_ms : 0;
            if (cap == 0) {
                // the smallest cap that is at least 0.65 of what the call then takes - its wave-slot time over the ~3 800 waves resident. (0.55 until the dead rows of
                // a window left in runs: the chains of the narrow workgroups - many windows, most of them dead - got shorter than this table says, and the sweep
                // moved: caps of 283 (= 0.55) / 310 / 330 / 360 / 400 / 450 ms -> 0.543-0.564 / 0.517-0.565 / 0.530-0.536 / 0.542-0.568 / 0.577 / 0.612 s. Before: measured on the 140 Mb
                // data, 13 197 edges: caps of 220 / 300 / 350 / 400 ms -> 0.78 / 0.74 / 0.71 / 0.80 s before the graph phases were rebuilt, 300 -> 0.55 s
                // after; below the balance the wide workgroups cost slots, above it the call waits for its last chains.)
                cap = 3200;
                for (double cq = 100; cq <= 3200; cq *= 1.0905) {   // (an eighth of an octave apart)
                    double slot = fixed_slot_ms;
                    for (const Opt& q : opts) { const int k = pick(q, cq); slot += q.ms[k] * (kLanes[k] / 64); }
                    // (round 6, once the persistent workgroups took their own bucket first - the slow passes of the earlier sweeps were that, not the cap: 60 / 70 / 80 / 90 / 100 %
                    // = caps of 308 / 366 / 399 / 435 / 475 ms at 140 Mb: 0.461 / 0.434 / 0.426 / 0.426 / 0.438 s, five passes each; at 400 Mb, 40 / 50 / 60 / 70 / 80 / 100 % =
                    // 872 / 1037 / 1234 / 1467 / 1744 / 2074 ms: 1.84 / 1.81 / 1.77 / 1.76 / 1.89-2.26 / 2.10 s. 70: profiles/r06_chain_cap_sweep.txt)
                    if (o.debug > 1) fprintf(stderr, "[hx] chain cap %.0f ms: wave-slot time over 3 800 waves %.0f ms\n", cq, slot / 3800.0);
                    if (cq >= std::max(10, o.poa_chain_pct) / 100.0 * slot / 3800.0) { cap = cq; break; }   // (round 6, same data, caps of 260 / 290 / 315 / 336 = 0.65 / 340 / 370 ms: 0.597 / 0.480 / 0.476 / 0.496-0.509 / 0.493 / 0.508 s)
                }
            }
            size_t hist[5] = {};
            for (size_t i = 0; i < ord.size(); i++) {
                const int k = pick(opts[i], cap);
                hist[k]++;
                chain_ms[ord[i]] = (float)opts[i].ms[k];
                if (opts[i].np[k] > 1) { P.edges[ord[i]].passes = opts[i].np[k]; plane[ord[i]] = (uint16_t)kLanes[k]; }
            }
            if (o.debug) fprintf(stderr, "[hx] column passes: chain cap %.0f ms; %zu / %zu / %zu / %zu / %zu unshared multi-wave edges in workgroups of 64 / 128 / 256 / 512 / 1024 lanes\n", cap, hist[0], hist[1], hist[2], hist[3], hist[4]);
        }
        // Wide members (build_classes) pay when ONE edge's serial chain is what the call waits for, and cost when the chip is busy anyway (every wide
        // workgroup has a CU to itself): time of the longest chain ~ its DP rows (nodes x sequences) x ~1 750 cycles, time of everything ~ DP cells
        // / throughput. Measured (Nanopore-like 25x): 4.6 Mb / 12 Mb genomes (423 / 1 079 edges) 0.190 -> 0.177 s and 0.248 -> 0.230 s with them, but 20 Mb
        // (1 864 edges, a chain 1.75 x longer) 0.426 -> 0.454 s and 30 Mb (2 737 edges) 0.327 -> 0.351 s: from ~1 500 edges on the chip is busy whatever the
        // longest chain does, so the edge count decides and the chain / work ratio only keeps calls without a dominant edge out.
        if (o.poa_wide_members < 0) {
            uint64_t top_rows = 0, sum_cost = 0;
            for (uint32_t e : todo) { top_rows = std::max<uint64_t>(top_rows, (uint64_t)P.edges[e].vcap * std::max<uint32_t>(1, P.nseq[e])); sum_cost += edge_cost(e); }
            wide_k = ne <= 1500 && (double)top_rows * 5e5 > (double)sum_cost ? 4 : 0;
            if (o.debug) fprintf(stderr, "[hx] wide members: longest chain %.3g node-sequences, all edges %.3g cost units -> %u\n", (double)top_rows, (double)sum_cost, wide_k);
        }
        // rows of H (see full_h above): how many rows leave the LDS ring before their last reader depends on how many the ring holds
        for (uint32_t e : todo) {
            hxk::PoaEdge& E = P.edges[e];
            const uint32_t ncol = E.lmax + 1, nt = lanes_of(e);
            uint64_t rb;
            const uint32_t cmq = cm_round(ncol, E.members > 1 ? E.members * nt : nt * std::max<uint32_t>(1, E.passes), E.members > 1 && ecols[e] < 4 ? 2u : 4u);
            const uint32_t Rp = ring_rows_of(nt, cmq, rb);
            // measured on PacBio-like data, rows read back from HBM per DP row: 0.15-0.4 % with 8 ring rows, 3-5 % with 4, 16-25 % on average
            // with 2 (single edges: up to every kept row, ~60 % of the rows). Graphs fill ~70 % of the node estimate these are fractions of.
            uint32_t est = o.poa_far_rows >= 0 ? (uint32_t)o.poa_far_rows : Rp >= 8 ? E.vcap / 32 + 256 : Rp >= 4 ? (E.vcap >> std::min(8, std::max(0, o.poa_far_shift))) + 256 : Rp >= 2 ? E.vcap / 2 + 256 : E.vcap + 1;
            E.hrows = full_h[e] || far_full[e] >= 3 ? E.vcap + 1 : (uint32_t)std::min<uint64_t>((uint64_t)E.vcap + 1, far_full[e] ? (uint64_t)std::max<uint32_t>(est, 256) << (2 * far_full[e]) : est);   // (a fourth attempt gets a row per node)
            // rows with more than 4 predecessors (a move byte per cell instead of a nibble): 1-2 % of the rows the DPs of 25- to 45-fold edges run
            // over, up to ~10 % of a finished deep graph; a 16th of the node estimate (graphs fill about a third of it) is the room, four times
            // more after every overflow
            E.wrows = wide_grow[e] >= 3 ? E.vcap + 1 : (uint32_t)std::min<uint64_t>((uint64_t)E.vcap + 1, ((uint64_t)E.vcap / 16 + 64) << (2 * wide_grow[e]));
        }
        // largest first (block scheduling is in grid order): cost ~ rows x columns x sequences; with column passes: the longest estimated chain first
        if (pass_on) std::sort(todo.begin(), todo.end(), [&](uint32_t a, uint32_t b) { return chain_ms[a] != chain_ms[b] ? chain_ms[a] > chain_ms[b] : a < b; });
        // (round 6, few-edge calls: by the rows of the chain, not by the cells. Per-edge timeline of the 12 Mb call, HX_DEBUG=2: with the longest chain at 150 ms the call
        // ended at 156 ms - with an edge of 2 088 columns x 21 reads that BEGAN at 109 ms and one of 2 470 x 25, 71 ms of chain, that began at 66 ms: in cell order
        // they stood behind wide gaps aligned by a few reads, whose chains are short)
        else if (!many_edges && !o.poa_order_by_cells) std::sort(todo.begin(), todo.end(), [&](uint32_t a, uint32_t b) { const double ca = chain_rows(a), cb = chain_rows(b); return ca != cb ? ca > cb : a < b; });
        else std::sort(todo.begin(), todo.end(), [&](uint32_t a, uint32_t b) { const uint64_t ca = edge_cost(a), cb = edge_cost(b); return ca != cb ? ca > cb : a < b; });
        return 0;
    }

    // ---- workspace. An edge that is shared by several workgroups owns a workspace slot for the call; every other launch class is PERSISTENT:
    // a number of slots, each sized for the class's largest edge, each owned by one workgroup that pulls edges (costliest first) from the class's
    // list (kernels/poa.hip). The workspace of a call is slots x largest edge, not the sum over its edges: 140 Mb on one GPU took 241 GB per
    // edge, a 400 Mb genome three batches. When even that does not fit the budget the slot counts are halved (fewer workgroups in flight); when a
    // class's slots alone do not fit, the edges are dealt to several batches in cost order as before.
    Need need_of(uint32_t e) const {
        const hxk::PoaEdge& E = P.edges[e];
        const uint64_t rw = ((uint64_t)E.lmax + 1 + 31) & ~31ull;                 // rows padded to 32 columns (the widest lane chunk)
        const uint64_t waves = (uint64_t)E.members * std::max<uint32_t>(1, E.passes) * (lanes_of(e) / 64);
        const uint64_t rwh = rw + (waves > 1 ? (waves + 3) & ~3ull : 0);       // rows of H end with one word per wave of the edge's pipeline
        Need n;
        n.nn = (uint64_t)E.vcap + 1; n.ec = E.ecap; n.dc = full_h[e] ? 0 : n.nn * (rw / 2); n.hc = (uint64_t)E.hrows * rwh; n.wc = full_h[e] ? 0 : (uint64_t)E.wrows * rw;
        n.lm = E.lmax; n.st = 4 * n.nn + E.ecap; n.al = n.nn + E.lmax + 2 + 64;   // (+ 64: the traceback's guard against a walk that does not end looks once per tile)
        n.mb = E.passes > 1 ? (uint64_t)E.passes * n.nn : 0;   // the carries handed from one column pass to the next
        return n;
    }
    // Thousands of edges: every launch class is persistent and would, on its own, ask for the whole chip (4096 waves) - six classes oversubscribe it six
    // times and the dispatcher deals the wave slots out as it pleases. A 1024-lane workgroup (16 waves: an EMPTY CU) can only be placed where nothing
    // else sits, so the 1024-lane class ran on the CUs it had grabbed in the first microseconds until everything else had finished: measured at 140 Mb
    // (profiles/r04_v1_fly_*), the classes ended at 920 / 1 130 / 1 440 / 1 730 / 1 900 ms - a tail of 0.8 s with the chip emptier and emptier.
    // Balanced launch (option poa_balance=0 switches it off): the classes of 512- and 1024-lane workgroups get workgroups for THEIR SHARE of the call's
    // wave-slot time (rows x lanes reserved: a workgroup holds its lanes whether or not a gap uses them all) x poa_balance_pct / 100, and a head start
    // (poa_wide_delay_us) so that they are resident before the small workgroups fragment the CUs; the small classes keep asking for the whole
    // chip and fill what is left - and what a large class that ends early leaves. Order: the 1024-lane class, then the shared edges (their waves are the
    // OLDEST on their SIMDs and win the issue arbitration: launched behind the 512-lane class as well, their chain - the longest of the call - took
    // 3.6 times as long), then the rest by size. (Measured on the way: 91 workgroups of 1024 lanes launched BEHIND the shared edges end at 2 230 ms,
    // 87 launched first at 1 540 ms - residency is the whole point.)
    int build_classes(const std::vector<uint32_t>& batch, std::vector<Cls>& classes) const {
        classes.clear();
        auto cls_of = [&](bool shared, uint32_t nt, uint32_t cm, bool dir, uint32_t dpl = 0, uint32_t pb = 0, bool pk = false) -> Cls& {
            for (Cls& q : classes) if (q.shared == shared && q.nt == nt && q.cm == cm && q.dir == dir && q.dpl == dpl && q.pb == pb && q.pk == pk) return q;
            classes.push_back(Cls{shared, nt, cm, dir, dpl, pb, pk, {}});
            return classes.back();
        };
        uint32_t n_wide = 0;
        for (uint32_t e : batch) {
            const uint32_t ncol = P.edges[e].lmax + 1;
            if (P.edges[e].members > 1) {
                const uint32_t ml = mlanes[e];
                uint32_t cmr = cm_round(ncol, P.edges[e].members * ml, ecols[e] < 4 ? 2u : 4u);
                if (cmr < 4 && hxk::poa_kernel_min_cm(n_wide < wide_k && ml < 1024 ? 1024 : (int)ml, true, true) > 2) cmr = 4;
                if (cmr > (uint32_t)hxk::poa_kernel_max_cm((int)ml)) return fail("hx_poa_batch: gap too long for the configured cluster size (raise option poa_cluster_max)");
                // the costliest shared edges run with WIDE members: workgroups of 1024 lanes of which the first cl_lanes take part in the DP (one
                // wave per SIMD, as before) and all sixteen waves in the graph phases of member 0 (graph update, CSR build, orders: latency-bound
                // loops over the nodes that want lanes). Such a workgroup has a CU to itself, so only a few edges get them.
                if (n_wide < wide_k && ml < 1024 && cmr <= 8) { n_wide++; cls_of(true, 1024, cmr, true, ml).edges.push_back(e); continue; }
                cls_of(true, ml, cmr, true).edges.push_back(e);   // batch is cost-sorted, so every class list is too
                continue;
            }
            const uint32_t nt = (uint32_t)kClassNT[class_of(e)];
            uint32_t cmq = cm_round(ncol, nt * std::max<uint32_t>(1, P.edges[e].passes));
            // (calls with column passes: the gaps of up to 255 bases run in the 8-column instance too - the 4-column instances take 143-158 registers, three waves
            // per SIMD, and one such wave on a SIMD leaves room for two of the 128-register ones instead of three: CUs sat at 12 waves of their 16)
            if (pass_on && !full_h[e] && cmq < 8) cmq = 8;
            if (o.poa_force_cm > 0) cmq = std::max<uint32_t>(cmq, std::min<uint32_t>((uint32_t)o.poa_force_cm, (uint32_t)hxk::poa_kernel_max_cm((int)nt)));   // (testing: a wider kernel instance than the gap needs)
            // (calls with column passes: a class per power of two of workspace need - a persistent workgroup's slot is sized for the largest edge of its
            // class, and a narrow workgroup may now hold a gap of any length; the classes of one kernel instance leave in one launch: launch_batch)
            uint32_t pb = 0;
            // (round 6: buckets HALF an octave apart where the memory budget binds - a slot holds the largest edge of its bucket, and with buckets an octave apart a
            // quarter of the slots' memory is slack on average. Where everything fits the octave stays: 140 Mb, alternating, five passes each: best 0.473 / 0.473 s in
            // 216 GB against 0.488 / 0.501 / 0.512 s in 183 GB)
            if (pass_on && !full_h[e]) {
                const uint64_t nb_ = need_bytes(need_of(e));
                if (o.poa_bucket_half_octaves && by_work) { const double l = std::log2((double)(nb_ >> 10) + 1.0) - 10.0; pb = l <= 0 ? 0u : (uint32_t)std::ceil(l * 2.0); }
                else { const uint64_t mb = nb_ >> 20; while ((1ull << pb) <= mb) pb++; }
            }
            cls_of(false, nt, cmq, !full_h[e], 0, pb, pass_on && !full_h[e] && cmq > 4).edges.push_back(e);   // (pk: with column passes every 8-column launch is the pruned instance - one launch per width)
        }
        // order of the launches: shared edges first (they set the duration), then by lanes; score-matrix launches after their direction-byte twins
        const bool bal = balanced, bigf = !many_edges && o.poa_big_first != 0;
        std::stable_sort(classes.begin(), classes.end(), [bal, bigf](const Cls& a, const Cls& b) {
            if (a.dir != b.dir) return a.dir;
            // (few-edge calls: the 900 member workgroups of the shared edges used to go out first and fill every CU's LDS; the 512-lane workgroups of the unshared
            // edges - chains of up to 100 ms of a 163 ms call - then began when two members on some CU had ended, 65-85 ms into the call, and most passes
            // took 182 ms instead of 163: tools/dev_r05_ab.py, 12 Mb, five passes each way)
            if (bigf) {   // wide members, then the large unshared workgroups, then the 256-lane members, then the rest
                auto grp = [](const Cls& q) { return q.shared && q.nt >= 1024 ? 0 : !q.shared && q.nt >= 512 ? 1 : q.shared ? 2 : 3; };
                if (grp(a) != grp(b)) return grp(a) < grp(b);
            }
            // (balanced launch: the 1024-lane workgroups - a whole CU each - go out before anything else sits anywhere; they share no SIMD with
            // the shared edges' members, which stay the oldest waves wherever they land)
            if (bal && (a.nt >= 1024 && !a.shared) != (b.nt >= 1024 && !b.shared)) return a.nt >= 1024 && !a.shared;
            if (a.shared != b.shared) return a.shared;
            if (a.nt != b.nt) return a.nt > b.nt;
            if (a.cm != b.cm) return a.cm > b.cm;
            if (a.pk != b.pk) return a.pk;
            return a.pb > b.pb;
        });
        double total_cost = 0;
        for (Cls& q : classes) {
            q.need = Need{};
            for (uint32_t e : q.edges) {
                need_max(q.need, need_of(e));
                q.share += (double)P.edges[e].vcap * std::max<uint32_t>(1, P.nseq[e]) * (q.shared ? (double)P.edges[e].members * mlanes[e] : (double)q.nt);   // DP rows x lanes reserved
            }
            total_cost += q.share;
        }
        for (Cls& q : classes) q.share = total_cost > 0 ? q.share / total_cost : 0;
        // The need buckets of one kernel instance share a launch, and a workgroup serves its own bucket AND every smaller one out of the slot it owns: the slot
        // must hold the largest of every component - nodes, edges, H rows, wide rows ... - over all those buckets, not only over its own. A bucket is a power of two
        // of the TOTAL bytes, and the components usually grow together; an edge that is redone with sixteen times the H rows (far rows outgrew the estimate) or a
        // larger wide-row pool is small in all and large in one - and ran, in the slot of a larger bucket, over that slot's share of the pool (round 5's fuzz: a GPU
        // memory access fault in the retries after "rows read back from HBM outgrew H"; once a consensus that differed from the oracle's).
        if (pass_on)
            for (size_t k = classes.size(); k-- > 1;) {
                Cls& a = classes[k - 1];
                const Cls& b = classes[k];
                if (!a.shared && !b.shared && a.nt == b.nt && a.cm == b.cm && a.dir == b.dir && a.dpl == b.dpl && a.pk == b.pk) need_max(a.need, b.need);
            }
        return 0;
    }
    // Slots of a persistent class: as many workgroups as the chip holds of that size at 16 waves per CU (all classes share the CUs, but when the
    // others have finished, what is left of this one still finds the whole chip: measured at 140 Mb, 1.95 s against 2.12 s with slots in
    // proportion to the classes' shares), at most one per edge. `shrink` scales the number down (memory budget).
    size_t slots_wanted(const Cls& q, uint32_t shrink, size_t cu_reserved) const {
        if (q.shared) return q.edges.size();
        size_t cap = std::max<size_t>(1, ((size_t)4096 / (q.nt / 64)) * (size_t)std::max(1, o.poa_slots_pct) / 100 * shrink / 1000);   // (`shrink`: per mille of the full count)
        if (balanced && q.dir && q.nt >= balance_nt) {
            cap = std::max<size_t>(1, std::min<size_t>(cap, (size_t)((double)cap * q.share * balance_f + 0.999)));   // the class's share of the chip
            // the members of shared edges must be resident TOGETHER (a member that waits for a CU stalls its edge: HXE_POA_STALLED and an unshared
            // redo): a wide class that holds most of the call's cost would otherwise take every CU before they are placed
            if (cu_reserved && q.nt >= 1024) cap = std::max<size_t>(1, std::min<size_t>(cap, 256 > cu_reserved ? 256 - cu_reserved : 1));
            cap = std::max<size_t>(cap, std::min<size_t>(4, q.edges.size()));   // (a floor: the share is a crude model and must not starve a class down to one workgroup)
        }
        if (o.poa_slots > 0) cap = (size_t)o.poa_slots;                        // (testing: workgroups per class, many edges each)
        return std::min(q.edges.size(), cap);
    }
    // A class runs persistent when it has more edges than slots: its list stays in DP-cost order (costliest first, taken by whoever is free) and
    // every slot is sized for the class's largest edge. (Tried: the slots' first edges = the edges with the largest workspace need, slot b sized
    // for its own first edge and the largest of the rest - 148 GB instead of 257 GB at 140 Mb, but 2.32-2.42 s against 2.03-2.09 s in the same
    // call: need and cost do not agree well enough - a gap aligned by 60 reads costs 20 times one aligned by 3 at the same need - and the
    // costliest edges then start late. Memory is saved by halving the slot counts instead: option poa_workspace_gb.)
    void arrange(std::vector<Cls>& classes, uint32_t shrink) const {
        size_t cu_reserved = 0;   // CUs the shared edges' member workgroups need (a 256-lane member: a quarter of a CU's wave slots, a wide one: a CU)
        for (const Cls& q : classes) if (q.shared) for (uint32_t e : q.edges) cu_reserved += ((size_t)P.edges[e].members * q.nt + 1023) / 1024;
        cu_reserved = std::min<size_t>(cu_reserved, 192);
        for (Cls& q : classes) {
            q.n_slots = slots_wanted(q, shrink, cu_reserved);
            q.persistent = !q.shared && (q.n_slots < q.edges.size() || pass_on) && hxk::poa_persistent_ok(q.dir);   // (pass_on: the need buckets of an instance share a launch)
            if (!q.persistent) q.n_slots = q.edges.size();
        }
        // The need buckets of one kernel instance share a launch and the chip: workgroups for 5/4 of what the chip holds of that width in all (a workgroup serves its
        // bucket and every smaller one, not the other way round; a bucket keeps a few workgroups of its own). Round 6: dealt IN PROPORTION TO THE BUCKETS' WORK (the
        // estimated chain time of their edges), not from the largest need down. Dealt top-down, the buckets of large need took a slot per edge and the memory with
        // them: a 400 Mb genome (37 936 edges, HX_DEBUG=1) ran with 1 208 slots of 137 MB for one bucket's 1 771 edges, EIGHT slots each for the 33 000 edges of
        // 37 MB and less, and 1 558 one-wave workgroups resident in all where the chip holds 4 096 - 258 GB of workspace and a chip at 40 %. With every bucket
        // finishing at about the same time, the same memory buys several times the workgroups (option poa_slots_by_work=0: as before).
        if (pass_on)
            for (size_t i = 0; i < classes.size();) {
                size_t j = i + 1;
                while (j < classes.size() && same_instance(classes[i], classes[j])) j++;
                if (classes[i].persistent && j - i > 1 && !o.poa_slots) {
                    size_t left = std::max<size_t>(1, ((size_t)4096 / (classes[i].nt / 64)) * 5 / 4 * shrink / 1000);
                    if (by_work) {
                        std::vector<double> w(j - i, 0.0);
                        double w_left = 0;
                        for (size_t k = i; k < j; k++) { for (uint32_t e : classes[k].edges) w[k - i] += std::max(1e-3, (double)chain_ms[e]); w_left += w[k - i]; }
                        for (size_t k = i; k < j; k++) {
                            Cls& q = classes[k];
                            const size_t floor_k = std::min<size_t>(q.edges.size(), 8);
                            const size_t share = w_left > 0 ? (size_t)((double)left * w[k - i] / w_left + 0.999) : 0;
                            q.n_slots = std::min(q.edges.size(), std::max(floor_k, std::min(share, left)));
                            left -= std::min(left, q.n_slots);
                            w_left -= w[k - i];
                        }
                    } else
                        for (size_t k = i; k < j; k++) {
                            Cls& q = classes[k];
                            q.n_slots = std::min(q.n_slots, std::max<size_t>(left, std::min<size_t>(q.edges.size(), 8)));
                            left -= std::min(left, q.n_slots);
                        }
                }
                i = j;
            }
    }
    static bool same_instance(const Cls& a, const Cls& b) { return a.persistent && b.persistent && a.nt == b.nt && a.cm == b.cm && a.dir == b.dir && a.dpl == b.dpl && a.pk == b.pk; }
    Need slot_need(const Cls& q, size_t b) const { return q.persistent ? q.need : need_of(q.edges[b]); }   // per slot: the edge's own need, or (persistent) the largest of the class
    uint64_t total_bytes(std::vector<Cls>& classes, uint32_t shrink) const {
        uint64_t t = 0;
        arrange(classes, shrink);
        for (Cls& q : classes) {
            for (size_t b = 0; b < q.n_slots; b++) t += need_bytes(slot_need(q, b));
            for (uint32_t e : q.edges) t += P.edges[e].vcap + (q.shared ? (uint64_t)P.edges[e].members * ((uint64_t)P.edges[e].vcap + 1) * 8 : 0);   // consensus output, cluster mailboxes
        }
        return t;
    }
    // ---- plan, part 4: batches and slot counts against the budget
    int plan_batches(const std::vector<uint32_t>& todo, std::vector<std::vector<uint32_t>>& batches, std::vector<uint32_t>& batch_shrink) {
        const size_t forced = o.poa_batches > 0 ? (size_t)o.poa_batches : 0;   // (testing)
        for (size_t nb = std::max<size_t>(1, forced);; nb++) {
            nb = std::min(nb, std::max<size_t>(1, todo.size()));
            batches.assign(nb, {}); batch_shrink.assign(nb, 1000); batch_by_work.assign(nb, 0);
            for (size_t i = 0; i < todo.size(); i++) batches[i % nb].push_back(todo[i]);   // dealt in cost order: every batch has its share of the large edges
            bool fits = true;
            for (size_t bi = 0; bi < nb && fits; bi++) {
                std::vector<Cls> cl;
                by_work = false;
                if (build_classes(batches[bi], cl)) return -1;
                uint32_t sh = 1000;   // per mille of the full slot counts: the largest that fits (down to 1 %: below that, more batches)
                // (the slots of the need buckets: from the largest need down while everything fits - at 140 Mb, 215 GB of a 257 GB budget, that is 3 % faster: the
                // long chains of the large buckets all start at once, 0.499 against 0.515 s - and in proportion to the buckets' work as soon as the budget binds:
                // 0.595 against 0.731 s under 140 GB, and one rank's 400 Mb share of configs[4] 2.00 against 2.95 s in its 260 GB)
                if (o.poa_slots_by_work && total_bytes(cl, sh) > budget) { by_work = true; if (build_classes(batches[bi], cl)) return -1; }   // (... and the buckets half an octave apart)
                batch_by_work[bi] = by_work;
                if (total_bytes(cl, sh) > budget) {
                    uint32_t lo = 10, hi = 1000;
                    while (hi - lo > 10) { const uint32_t mid = (lo + hi) / 2; if (total_bytes(cl, mid) <= budget) lo = mid; else hi = mid; }
                    sh = lo;
                }
                batch_shrink[bi] = sh;
                fits = total_bytes(cl, sh) <= budget;
            }
            if (fits) break;
            if (nb >= todo.size()) return fail("hx_poa_batch: a single edge needs more POA workspace than the device has free");
        }
        return 0;
    }

    // the pruned instance: unshared edges, direction bytes, 4 or 8 columns per lane, a workgroup of several waves - or of any width when its edges take their
    // columns in passes (a one-wave workgroup that holds its gap has nothing to skip: its rows are whole rows)
    bool launch_pruned(const Cls& q) const {
        if (q.shared) return hxk::poa_prune_ok(q.dir, (int)q.cm) && prune_shared_pct != 0;
        return hxk::poa_prune_ok(q.dir, (int)q.cm) && (q.nt >= (uint32_t)o.poa_prune_lanes || q.pk) && prune_pct != 0;
    }
    // ---- launch of one batch; what the collection needs afterwards
    struct Launched { std::vector<uint32_t> edges; uint64_t cns_bytes = 0, bytes = 0; std::vector<Cls> classes; };
    int launch_batch(const std::vector<uint32_t>& batch, uint32_t shrink, Launched& lb) {
        hipStream_t s = c->stream;
        PoaPoolBufs& B = c->poa_pools;
        lb.edges = batch;
        std::vector<Cls>& classes = lb.classes;
        if (build_classes(batch, classes)) return -1;
        // ---- slots and their offsets into the pools
        std::vector<hxk::PoaSlot> h_slots;
        uint64_t no = 0, eo = 0, ho = 0, dro = 0, wo = 0, so = 0, co = 0, sto = 0, ao = 0, clo = 0;
        auto add_slot = [&](const Need& n) {
            h_slots.push_back(hxk::PoaSlot{no, eo, ho, dro, wo, so, sto, ao, clo});
            no += n.nn; eo += n.ec; ho += n.hc; dro += n.dc; wo += n.wc; so += n.lm; sto += n.st; ao += n.al; clo += n.mb;
        };
        arrange(classes, shrink);
        for (Cls& q : classes) {
            q.slot_at = h_slots.size();
            for (size_t k = 0; k < q.n_slots; k++) {
                if (!q.persistent) P.edges[q.edges[k]].slot = (uint32_t)h_slots.size();   // one workgroup (or cluster) per edge: the edge's own slot
                add_slot(slot_need(q, k));
            }
            for (uint32_t e : q.edges) {
                P.edges[e].cns_off = co; co += P.edges[e].vcap;
                if (q.shared) { P.edges[e].cl_off = clo; clo += (uint64_t)P.edges[e].members * ((uint64_t)P.edges[e].vcap + 1); }
            }
        }
        lb.cns_bytes = co;
        const uint64_t bytes = no * 106 + eo * 28 + ho * 4 + dro + wo + so + sto * 4 + ao * 8 + clo * 8 + co;
        lb.bytes = bytes;
        // The pools of the batch, carved out of the context's arena (256-byte aligned). The arena grows when a batch needs more than it holds - by an eighth
        // more than asked, so that the retries of a call (a few edges with more room) do not each allocate again - and never shrinks.
        {
            const auto tw0 = std::chrono::steady_clock::now();
            size_t at = 0;
            auto place = [&at](auto& buf, uint64_t n) { buf.off = at; at += (std::max<uint64_t>(1, n) * sizeof(*buf.p) + 255) & ~(size_t)255; };
            auto bind = [this](auto& buf, uint64_t) { buf.p = reinterpret_cast<decltype(buf.p)>(c->poa_arena.p + buf.off); };
#define HX_POOLS(F) \
            F(B.H, ho); F(B.dir, dro); F(B.dirw, wo); F(B.wslot, no); F(B.code, no); F(B.n_aligned, no); F(B.mark, no); F(B.check, no); F(B.row_code, no); F(B.row_sink, no); \
            F(B.row_al, no); F(B.aligned, 3 * no); F(B.in_head, no); F(B.in_tail, no); F(B.out_head, no); F(B.out_tail, no); F(B.rank2node, no); F(B.node2rank, no); \
            F(B.row_pred_off, no); F(B.score, no); F(B.pred, no); F(B.pred_rank, eo); F(B.pred_w, eo); F(B.e_from, eo); F(B.e_to, eo); F(B.e_next_in, eo); F(B.e_next_out, eo); \
            F(B.e_w, eo); F(B.stack, sto); F(B.aln_node, ao); F(B.aln_pos, ao); F(B.row_meta, no); F(B.row_pred0, no); F(B.row_pred1, no); F(B.nrec, no); F(B.nrec2, no); \
            F(B.seq, so); F(B.cns, co); F(B.mbox, clo); F(B.csync, (uint64_t)ne * 8); F(B.sinkbuf, (uint64_t)ne * (1 + 2 * 1024));
            HX_POOLS(place)
            std::lock_guard<std::mutex> lk(c->poa_arena_mu);
            if (at > c->poa_arena.cap) {
                HIPCHK(hipStreamSynchronize(s));   // (nothing of an earlier batch is in flight: collect_batch has read its results)
                hipError_t e = c->poa_arena.ensure(std::min<size_t>(at + at / 8, std::max<size_t>(at, (size_t)budget + (size_t)ne * 8400)));
                if (e != hipSuccess) { (void)hipGetLastError(); e = c->poa_arena.ensure(at); }
                if (e != hipSuccess) {
                    // the budget was taken from what hipMemGetInfo called free - which somebody else (another context on this device: ranks that share a GPU, another
                    // process) has taken since. The caller looks again and plans anew with what is there now.
                    (void)hipGetLastError();
                    g_err = "hx_poa_batch: cannot allocate " + std::to_string(at >> 20) + " MB of POA workspace: " + hipGetErrorString(e);
                    return 1;
                }
            }
            HX_POOLS(bind)
#undef HX_POOLS
            c->poa_host_ms[1] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
        }
        c->poa_workspace_bytes = std::max<uint64_t>(c->poa_workspace_bytes, bytes);
        c->poa_last_workspace_bytes = std::max<uint64_t>(c->poa_last_workspace_bytes, bytes);
        const auto te0 = std::chrono::steady_clock::now();
        HIPCHK(hipMemsetAsync(B.csync.p, 0, (size_t)ne * 8 * 4, s));
        if (clo) HIPCHK(hipMemsetAsync(B.mbox.p, 0, clo * 8, s));   // tag 0 = nothing published
        HIPCHK(c->poa_edges.reserve(ne)); HIPCHK(c->poa_len.reserve(ne)); HIPCHK(c->poa_status.reserve(ne));
        std::vector<uint32_t> order_all;   // shared launches: one entry per workgroup (edge | member << 24); persistent launches: the class's edges, costliest first
        for (Cls& q : classes) {
            q.order_at = order_all.size();
            if (q.shared && !o.poa_no_xcd_map) {
                // Workgroups are handed to the 8 XCDs round-robin by index: put the members of one edge 8 indices apart so that they share an
                // XCD (one L2 for the carries, the handshakes and the direction bytes member 0 walks back over). Holes are no-op workgroups.
                for (size_t g0 = 0; g0 < q.edges.size(); g0 += 8) {
                    const size_t g1 = std::min(q.edges.size(), g0 + 8);
                    uint32_t gmax = 0;
                    for (size_t j = g0; j < g1; j++) gmax = std::max(gmax, P.edges[q.edges[j]].members);
                    for (uint32_t m = 0; m < gmax; m++)
                        for (size_t j = g0; j < g0 + 8; j++)
                            order_all.push_back(j < g1 && m < P.edges[q.edges[j]].members ? (q.edges[j] | (m << 24)) : 0x00ffffffu);
                }
            } else if (q.shared)
                for (uint32_t e : q.edges) for (uint32_t m = 0; m < P.edges[e].members; m++) order_all.push_back(e | (m << 24));
            else
                for (uint32_t e : q.edges) order_all.push_back(e);
            q.blocks = q.shared ? order_all.size() - q.order_at : q.n_slots;   // (not shared: one workgroup per slot - per edge unless persistent)
        }
        if (ne >= (1u << 24)) return fail("hx_poa_batch: more than 2^24 edges in one call");
        HIPCHK(hipMemcpyAsync(c->poa_edges.p, P.edges.data(), (size_t)ne * sizeof(hxk::PoaEdge), hipMemcpyHostToDevice, s));
        HIPCHK(c->poa_order.reserve(order_all.size()));
        HIPCHK(hipMemcpyAsync(c->poa_order.p, order_all.data(), order_all.size() * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c->poa_slots.reserve(h_slots.size()));
        HIPCHK(hipMemcpyAsync(c->poa_slots.p, h_slots.data(), h_slots.size() * sizeof(hxk::PoaSlot), hipMemcpyHostToDevice, s));
        HIPCHK(c->poa_counters.reserve(classes.size()));
        HIPCHK(hipMemsetAsync(c->poa_counters.p, 0, classes.size() * 4, s));
        hxk::PoaPools pools{B.code.p, B.n_aligned.p, B.aligned.p, B.in_head.p, B.in_tail.p, B.out_head.p, B.out_tail.p, B.rank2node.p, B.node2rank.p,
                            B.mark.p, B.check.p, B.stack.p, B.score.p, B.pred.p, B.row_code.p, B.row_sink.p, B.row_pred_off.p, B.pred_rank.p, B.row_meta.p, B.row_pred0.p, B.row_pred1.p, B.nrec.p, B.nrec2.p,
                            B.e_from.p, B.e_to.p, B.e_next_in.p, B.e_next_out.p, B.e_w.p, B.aln_node.p, B.aln_pos.p, B.H.p, B.dir.p, B.dirw.p, B.wslot.p, B.seq.p,
                            B.mbox.p, B.csync.p, B.sinkbuf.p, B.row_al.p, B.pred_w.p};
        const size_t n_streams = (size_t)std::min(8, std::max(1, o.poa_streams));   // (8: a stream per launch class of a 140 Mb call - with 6, the two one-wave classes waited 130 / 300 ms behind the shared edges)
        size_t wg_total = 0;
        for (const Cls& q : classes) wg_total += q.blocks;
        c->tick();
        // The launches. Persistent classes that differ only in their need bucket (build_classes) leave in ONE launch: their slots, lists and counters
        // lie side by side in class order (largest need first), `btab` tells a workgroup which bucket its slot belongs to (kernels/poa.hip k_poa).
        std::vector<uint32_t> h_btab;
        std::vector<std::array<size_t, 3>> groups;   // first class, one past the last, offset of the group's table in h_btab
        for (size_t i = 0; i < classes.size();) {
            size_t j = i + 1;
            const Cls& a = classes[i];
            while (j < classes.size() && same_instance(a, classes[j])) j++;
            groups.push_back({i, j, h_btab.size()});
            if (a.persistent) {
                h_btab.push_back((uint32_t)(j - i) | (o.poa_own_bucket_first ? 1u << 16 : 0u));
                uint32_t se = 0, ib = 0;
                for (size_t k = i; k < j; k++) { se += (uint32_t)classes[k].blocks; h_btab.push_back(se); }
                for (size_t k = i; k < j; k++) { h_btab.push_back(ib); ib += (uint32_t)classes[k].edges.size(); }
                h_btab.push_back(ib);
                for (size_t k = i; k < j; k++) for (uint32_t e : classes[k].edges) h_btab.push_back((uint32_t)std::min(4.0e9, (double)chain_ms[e] * 1000.0));   // est[]: microseconds
            }
            i = j;
        }
        HIPCHK(c->poa_btab.reserve(std::max<size_t>(1, h_btab.size())));
        if (!h_btab.empty()) HIPCHK(hipMemcpyAsync(c->poa_btab.p, h_btab.data(), h_btab.size() * 4, hipMemcpyHostToDevice, s));
        if (scratch_warm_once(c)) return -1;
        HIPCHK(hipEventRecord(c->poa_ev[8], s));
        size_t gi = 0;
        for (const auto& grp : groups) {
            const size_t ci = grp[0];
            const Cls& q = classes[ci];
            size_t g_blocks = 0, g_items = 0;
            for (size_t k = grp[0]; k < grp[1]; k++) { g_blocks += classes[k].blocks; g_items += classes[k].edges.size(); }
            const int sk = (int)(gi % n_streams);   // stream / event of the launch (launches that share a stream run one after the other)
            // LDS of the launch: the ring its row width allows, a power of two of kept rows
            uint64_t ring_need = 0;
            const uint32_t dp_nt = q.dpl ? q.dpl : q.nt;   // lanes in the DP
            const uint32_t R = ring_rows_of(dp_nt, q.cm, ring_need);
            // few edges: ask for enough LDS per workgroup that the dispatcher cannot stack them on a handful of CUs while others idle
            // (a lone wave runs at twice the speed of two waves sharing a SIMD); many edges: request only what the ring needs
            uint64_t lds_bytes = ring_need;
            {
                const uint64_t per_cu = (wg_total + 255) / 256;
                if (per_cu < 8) lds_bytes = std::max<uint64_t>(lds_bytes, std::min<uint64_t>(kPoaLdsMax, (158 * 1024) / per_cu - 18 * 1024));
                // hundreds of edges: the longest ones set the duration, and their waves run faster with two neighbours on a SIMD than with
                // three - 10 KB of LDS per wave keeps a CU at 12 waves (thousands of edges: 16, the ring alone is 8.3 KB per wave)
                if (!many_edges) lds_bytes = std::max<uint64_t>(lds_bytes, std::min<uint64_t>(kPoaLdsMax, 10 * 1024 * (uint64_t)(dp_nt / 64)));
                if (o.poa_ring_zero) lds_bytes = ring_need;   // (one row's worth: the kernel then finds room for no kept row either)
            }
            const int dcls = q.shared ? 0 : q.nt >= 1024 ? 1 : q.nt >= 512 ? 2 : q.nt >= 256 ? 3 : q.nt >= 128 ? 4 : 5;
            for (size_t k = grp[0]; k < grp[1]; k++)
                for (uint32_t e : classes[k].edges) { c->dbg_cls[e] = (uint8_t)(dcls + (q.dir ? 0 : 5)); c->dbg_shape[e] = q.nt | std::min<uint32_t>(255, P.edges[e].passes) << 16 | std::min<uint32_t>(255, P.edges[e].members) << 24; }
            c->dbg_ring[dcls + (q.dir ? 0 : 5)] = R;
            HIPCHK(hipStreamWaitEvent(c->poa_streams[sk], c->poa_ev[8], 0));
            hxk::PoaLaunch L{};
            L.edges = c->poa_edges.p; L.order = c->poa_order.p + q.order_at; L.n_items = q.persistent ? (uint32_t)g_items : (uint32_t)q.blocks;
            L.slots = c->poa_slots.p + (q.persistent ? q.slot_at : 0); L.counter = q.persistent ? c->poa_counters.p + ci : nullptr; L.n_blocks = (uint32_t)g_blocks;
            L.btab = q.persistent ? c->poa_btab.p + grp[2] : nullptr;
            L.seqs = c->poa_seqs.p; L.packed = in.d_packed; L.read_off = in.d_roff; L.read_len = in.d_rlen; L.pools = pools;
            L.match = pp->match; L.mismatch = pp->mismatch; L.gap = pp->gap; L.cns = B.cns.p; L.cns_len = c->poa_len.p; L.status = c->poa_status.p;
            L.cells = c->poa_cells_d.p; L.phase = c->poa_phase_d.p; L.block_threads = (int)q.nt; L.cm = (int)q.cm; L.poll_limit = (uint32_t)o.poa_poll_limit; L.ring_bytes = (uint32_t)lds_bytes;
            L.use_dir = q.dir; L.max_indeg = (uint32_t)std::min(16, std::max(1, o.poa_max_indeg)); L.dp_lanes = q.dpl;
            const bool wide_q = q.dpl || (balanced && q.nt >= balance_nt && q.nt >= 512 && q.persistent), shared_first = many_edges && q.shared && (o.poa_resident_first & 1);
            L.started = ((wide_q && (o.poa_resident_first & 2)) || shared_first) && gi < 16 ? c->poa_started + gi : nullptr;
            if (L.started) *(volatile uint32_t*)L.started = 0u;
            L.prune_pct = launch_pruned(q) ? (std::min<uint32_t>(q.shared ? prune_shared_pct : prune_pct, 1000u) | (o.poa_prune_lazy ? 1u << 16 : 0u)) : 0u;
            if (o.debug) { int occ = 0; L.occupancy = &occ; hxk::poa_run(L, c->poa_streams[sk]); L.occupancy = nullptr; fprintf(stderr, "[hx] launch %zu: %zu workgroups of %u lanes, %.1f KB of ring: %d workgroups per CU\n", gi, g_blocks, q.nt, lds_bytes / 1024.0, occ); }
            hxk::poa_run(L, c->poa_streams[sk]);
            HIPCHK(hipEventRecord(c->poa_ev[sk], c->poa_streams[sk]));
            HIPCHK(hipStreamWaitEvent(s, c->poa_ev[sk], 0));
            if (wide_q || shared_first) {
                // a 1024-lane workgroup needs an EMPTY CU: give the dispatcher a head start before the other launches fill the chip with small
                // workgroups (once they have, a CU only empties when its longest resident workgroup ends)
                HIPCHK(hipEventSynchronize(c->poa_ev[8]));   // (what precedes the launches on `s` is done: the wide launch is starting)
                const auto tw = std::chrono::steady_clock::now();
                if (L.started) {
                    // (round 6: not a fixed delay but the launch's own word - every workgroup adds itself when it begins. One pass in five of the 140 Mb call took 610-650 ms
                    // instead of 440-470: no edge redone, the same launches - in another order of arrival on the CUs. The shared edges' members and the wide classes
                    // must be the oldest waves where they sit; the next launch leaves when they have all begun, or after 2 ms)
                    volatile uint32_t* w = L.started;
                    while (*w < L.n_blocks && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw).count() < 2000.0) { }
                    if (o.debug) fprintf(stderr, "[hx] launch %zu: %u of %u workgroups had begun %.0f us after the launch\n", gi, (unsigned)*w, L.n_blocks, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw).count());
                } else
                    while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw).count() < o.poa_wide_delay_us) { }
            }
            gi++;
        }
        const auto te1 = std::chrono::steady_clock::now();
        c->tock(3);
        c->poa_host_ms[2] += std::chrono::duration<double, std::milli>(te1 - te0).count();
        c->poa_host_ms[3] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - te1).count();
        HIPCHK(hipGetLastError());
        if (o.debug) {
            HIPCHK(hipStreamSynchronize(s));
            fprintf(stderr, "[hx] POA batch: %zu edges, %.2f GB workspace, workgroups", batch.size(), bytes / 1e9);
            for (const Cls& q : classes) fprintf(stderr, " %s%s%s%s%ux%u:%zu(%zu edges, largest %.1f MB)", q.shared ? "shared/" : "", q.persistent ? "persistent/" : "", q.dir ? "" : "matrix/",
                                                 launch_pruned(q) ? (q.pk ? "pruned/passes/" : "pruned/") : "", q.nt, q.cm, q.blocks, q.edges.size(), need_bytes(q.need) / 1e6);
            fprintf(stderr, ", %.1f ms since the call began\n", ms_since_start());
        }
        return 0;
    }

    // ---- collection: consensus strings of the edges that are done; the others go to `retry` (worst-case workspace next) / `retry_same` (another way)
    int collect_batch(const Launched& lb, std::vector<uint32_t>& retry, std::vector<uint32_t>& retry_same) {
        const auto tc0 = std::chrono::steady_clock::now();
        hipStream_t s = c->stream;
        std::vector<uint32_t> h_len(ne), h_status(ne);
        HIPCHK(hipMemcpy(h_len.data(), c->poa_len.p, (size_t)ne * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(h_status.data(), c->poa_status.p, (size_t)ne * 4, hipMemcpyDeviceToHost));
        // the finished strings, moved side by side on the device before the download: the buffer the kernels write into is sized by the node estimates
        // (100 MB for the 13 000 edges of a 140 Mb genome, of which 30 MB are consensus)
        std::vector<uint32_t> desc;
        std::vector<uint64_t> dense_off(lb.edges.size() + 1, 0);
        desc.reserve(lb.edges.size() * 5);
        for (size_t i = 0; i < lb.edges.size(); i++) {
            const uint32_t e = lb.edges[i];
            const uint32_t n = h_status[e] ? 0u : std::min<uint32_t>(h_len[e], P.edges[e].vcap);
            dense_off[i + 1] = dense_off[i] + n;
            if (!n) continue;
            const uint64_t so = P.edges[e].cns_off, to = dense_off[i];
            desc.insert(desc.end(), {(uint32_t)so, (uint32_t)(so >> 32), (uint32_t)to, (uint32_t)(to >> 32), n});
        }
        cns_blocks.emplace_back(new char[std::max<uint64_t>(1, dense_off.back())]);
        const char* h_cns = cns_blocks.back().get();
        if (!desc.empty()) {
            HIPCHK(c->poa_gather.reserve(desc.size())); HIPCHK(c->poa_cns_dense.reserve(dense_off.back()));
            HIPCHK(hipMemcpyAsync(c->poa_gather.p, desc.data(), desc.size() * 4, hipMemcpyHostToDevice, s));
            hxk::gather_bytes(c->poa_pools.cns.p, c->poa_gather.p, (uint32_t)(desc.size() / 5), c->poa_cns_dense.p, s);
            HIPCHK(hipMemcpyAsync(cns_blocks.back().get(), c->poa_cns_dense.p, dense_off.back(), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
        }
        struct Lap { double& ms; std::chrono::steady_clock::time_point t0; ~Lap() { ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } lap{c->poa_host_ms[4], tc0};
        if (o.debug) {
            size_t n_far = 0, n_nodir = 0, n_over = 0, n_wide = 0, n_sinks = 0, n_stall = 0;
            for (uint32_t e : lb.edges) { n_far += !!(h_status[e] & HXE_POA_FARROWS); n_nodir += !!(h_status[e] & HXE_POA_NODIR); n_over += !!(h_status[e] & HXE_POA_OVERFLOW); n_wide += !!(h_status[e] & HXE_POA_WIDEROWS); n_sinks += !!(h_status[e] & HXE_POA_SINKS); n_stall += !!(h_status[e] & HXE_POA_STALLED); }
            if (n_far + n_nodir + n_over + n_wide + n_sinks + n_stall) fprintf(stderr, "[hx] POA batch: to be redone: %zu (rows read back from HBM outgrew H), %zu (in-degree above the direction bytes' limit), %zu (graph outgrew its workspace), %zu (rows with more than 4 predecessors outgrew the wide-row pool), %zu (more sink rows than the launch keeps), %zu (members of a shared edge not resident together%s: unshared next)\n",
                                                                       n_far, n_nodir, n_over, n_wide, n_sinks, n_stall, balanced ? ", in a balanced launch" : "");
        }
        for (size_t i = 0; i < lb.edges.size(); i++) {
            const uint32_t e = lb.edges[i];
            if (h_status[e] & HXE_POA_FARROWS) { if (P.edges[e].hrows >= P.edges[e].vcap + 1) return fail("hx_poa_batch: internal error (far-row retry)"); far_full[e]++; retry_same.push_back(e); continue; }
            if (h_status[e] & HXE_POA_STALLED) {
                if (P.edges[e].members < 2) return fail("hx_poa_batch: internal error (a wave of an unshared edge gave up waiting)");
                no_share[e] = 1; retry_same.push_back(e); continue;
            }
            if (h_status[e] & HXE_POA_WIDEROWS) { if (P.edges[e].wrows >= P.edges[e].vcap + 1) return fail("hx_poa_batch: internal error (wide-row retry)"); wide_grow[e]++; retry_same.push_back(e); continue; }
            if (h_status[e] & HXE_POA_SINKS) { if (many_sinks[e]) return fail("hx_poa_batch: internal error (sink-list retry)"); many_sinks[e] = 1; retry_same.push_back(e); continue; }
            if (h_status[e] & HXE_POA_NODIR) { if (force_nodir[e]) return fail("hx_poa_batch: internal error (direction-byte retry)"); force_nodir[e] = 1; retry_same.push_back(e); continue; }
            if (h_status[e] & ~(uint32_t)HXE_POA_OVERFLOW) return fail("hx_poa_batch: internal error (kernel variant / column count mismatch)");
            if (h_status[e] & HXE_POA_OVERFLOW) {
                if (P.edges[e].vcap >= P.sumL[e]) return fail("hx_poa_batch: POA workspace overflow at worst-case size (internal error)");
                grow[e]++;
                retry.push_back(e);
            } else cns[e] = CnsView{h_cns + dense_off[i], (size_t)(dense_off[i + 1] - dense_off[i])};
        }
        return 0;
    }
};
}  // namespace

static int poa_consensus(hx_ctx* c, const PoaInput& in, const hx_poa_params* pp, hx_cns_out* out) {
    memset(out, 0, sizeof(*out));
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    PoaCall K(c, in, pp);
    const uint32_t ne = K.ne;
    std::vector<uint32_t> todo;
    for (double& v : c->poa_host_ms) v = 0;
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    if (K.plan_input(todo)) return -1;
    c->poa_host_ms[0] += K.ms_since_start();
    if (c->opt.debug) fprintf(stderr, "[hx] POA call: %u edges prepared in %.1f ms\n", (unsigned)ne, K.ms_since_start());
    c->dbg_cls.assign(ne, 11); for (int k = 0; k < 11; k++) c->dbg_ring[k] = 0;
    c->dbg_shape.assign(ne, 0);
    c->dbg_nseq = K.P.nseq; c->dbg_lmax.resize(ne); for (uint32_t e = 0; e < ne; e++) c->dbg_lmax[e] = K.P.edges[e].lmax;
    HIPCHK(c->poa_seqs.reserve(K.P.seqs.size()));
    if (!K.P.seqs.empty()) HIPCHK(hipMemcpyAsync(c->poa_seqs.p, K.P.seqs.data(), K.P.seqs.size() * sizeof(hxk::PoaSeq), hipMemcpyHostToDevice, s));
    HIPCHK(c->poa_cells_d.reserve(1));
    HIPCHK(hipMemsetAsync(c->poa_cells_d.p, 0, 8, s));
    if (!c->poa_budget) {   // measured once: later calls would count the context's own (persistent) workspace as used
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        { std::lock_guard<std::mutex> lk(c->poa_arena_mu); free_b += c->poa_arena.cap; }   // (an arena reserved ahead - hx_poa_reserve - is the workspace's own)
        c->poa_budget = (uint64_t)(free_b * 0.9);
        c->poa_free_at_first_call = free_b;
    }
    // option poa_workspace_gb: cap of the POA workspace (default: 90 % of what was free when the context first ran a consensus). The workgroups in
    // flight per launch class are scaled down until the slots fit. Measured at 140 Mb (13 230 edges): 257 GB 2.0-2.1 s, 138 GB 2.10-2.13 s (and
    // the first call, which allocates the pools, 4.1 instead of 5-7.6 s), 39 GB 4.6 s, 22 GB 8.5 s; a 400 Mb genome (37 608 edges): 148 GB 6.7 s.
    // (Round 6, measured on the 140 Mb data set, 13 197 edges: 0.50 s with the 215 GB the plan takes of a 257 GB budget, 0.75-0.90 s under a cap of 140 GB, 1.04 s
    // under 100 GB - the slot counts of the one-wave classes are what shrinks. No cap of its own, then: 90 % of what is free.)
    K.budget = c->opt.poa_workspace_gb > 0 ? (uint64_t)(c->opt.poa_workspace_gb * 1e9) : c->poa_budget;
    c->poa_last_workspace_bytes = 0;
    HIPCHK(c->poa_phase_d.reserve((size_t)ne * hxk::POA_PHASE_WORDS));
    HIPCHK(hipMemsetAsync(c->poa_phase_d.p, 0, std::max<size_t>(1, (size_t)ne * hxk::POA_PHASE_WORDS) * 8, s));
    while (!todo.empty()) {
        const auto tp0 = std::chrono::steady_clock::now();
        if (K.knobs(todo.size()) || K.size_edges(todo)) return -1;
        const double t_size = since(tp0);
        std::vector<std::vector<uint32_t>> batches;
        std::vector<uint32_t> batch_shrink;
        if (K.plan_batches(todo, batches, batch_shrink)) return -1;
        c->poa_host_ms[0] += since(tp0);
        if (c->opt.debug) fprintf(stderr, "[hx] POA plan: widths and rooms of %zu edges %.2f ms, batches and slots %.2f ms\n", todo.size(), t_size, since(tp0) - t_size);
        std::vector<uint32_t> retry, retry_same;   // retry with the worst-case workspace / with the score-matrix traceback
        for (size_t bi = 0; bi < batches.size(); bi++) {
            if (batches[bi].empty()) continue;
            PoaCall::Launched lb;
            K.by_work = bi < K.batch_by_work.size() && K.batch_by_work[bi] != 0;
            const int rc = K.launch_batch(batches[bi], batch_shrink[bi], lb);
            if (rc == 1) {   // the arena could not be had at the planned size: the budget again from what is free NOW, the rest of the round planned anew
                size_t free_b = 0, total_b = 0;
                HIPCHK(hipMemGetInfo(&free_b, &total_b));
                uint64_t now_b;
                { std::lock_guard<std::mutex> lk(c->poa_arena_mu); now_b = (uint64_t)((double)(free_b + c->poa_arena.cap) * 0.9); }
                if (now_b + (now_b >> 6) >= K.budget) return -1;   // (nothing changed: the error stands)
                if (c->opt.debug) fprintf(stderr, "[hx] POA workspace: %.1f GB could not be allocated; %.1f GB are free now, budget %.1f -> %.1f GB\n", lb.bytes / 1e9, free_b / 1e9, K.budget / 1e9, now_b / 1e9);
                K.budget = now_b;
                if (c->opt.poa_workspace_gb <= 0) c->poa_budget = now_b;
                for (size_t bj = bi; bj < batches.size(); bj++) retry_same.insert(retry_same.end(), batches[bj].begin(), batches[bj].end());
                break;
            }
            if (rc || K.collect_batch(lb, retry, retry_same)) return -1;
        }
        todo.swap(retry);
        todo.insert(todo.end(), retry_same.begin(), retry_same.end());
    }
    const auto tf0 = std::chrono::steady_clock::now();
    unsigned long long cells = 0;
    HIPCHK(hipMemcpy(&cells, c->poa_cells_d.p, 8, hipMemcpyDeviceToHost));
    c->poa_phase.resize((size_t)ne * hxk::POA_PHASE_WORDS);
    if (ne) HIPCHK(hipMemcpy(c->poa_phase.data(), c->poa_phase_d.p, (size_t)ne * hxk::POA_PHASE_WORDS * 8, hipMemcpyDeviceToHost));
    std::vector<uint64_t> off((size_t)ne + 1, 0);
    for (uint32_t e = 0; e < ne; e++) off[e + 1] = off[e] + K.cns[e].size();
    out->n_edge = ne;
    out->cns_off = (uint64_t*)malloc(((size_t)ne + 1) * 8); memcpy(out->cns_off, off.data(), ((size_t)ne + 1) * 8);
    out->cns = (char*)malloc(std::max<uint64_t>(1, off[ne]));
    {   // (30 MB of strings at 140 Mb: a few threads, each its range of the edges)
        const uint32_t nt = off[ne] > (4u << 20) ? 4u : 1u;
        auto part = [&](uint32_t t) { for (uint32_t e = (uint32_t)((uint64_t)ne * t / nt); e < (uint32_t)((uint64_t)ne * (t + 1) / nt); e++) if (K.cns[e].size()) memcpy(out->cns + off[e], K.cns[e].data(), K.cns[e].size()); };
        std::vector<std::thread> th;
        for (uint32_t t = 1; t < nt; t++) th.emplace_back(part, t);
        part(0);
        for (std::thread& x : th) x.join();
    }
    out->dp_cells = cells; out->seq_bases = K.seq_bases; out->n_aligned = K.n_aligned;
    c->poa_host_ms[5] = since(tf0); c->poa_host_ms[7] = K.ms_since_start();
    if (c->opt.debug) fprintf(stderr, "[hx] POA call, host wall time: plan %.1f ms, workspace %.1f ms (%llu device allocations so far, %.0f ms), enqueue %.1f ms, device %.1f ms, collect %.1f ms, finish %.1f ms, total %.1f ms\n",
                              c->poa_host_ms[0], c->poa_host_ms[1], (unsigned long long)c->poa_arena.n_alloc, c->poa_arena.alloc_ms, c->poa_host_ms[2], c->poa_host_ms[3], c->poa_host_ms[4], c->poa_host_ms[5], c->poa_host_ms[7]);
    return 0;
}

extern "C" int hx_poa_batch(hx_ctx* c, const hx_poa_params* pp, hx_cns_out* out) {
    memset(out, 0, sizeof(*out));
    if (!c->have_coords) return fail("hx_poa_batch: hx_edge_coords has not run");
    const PoaInput in{c->n_sel, c->h_supp_off.data(), c->h_supp_lr.data(), c->h_spos.data(), c->h_epos.data(), c->h_rlen.data(), c->packed.p, c->roff.p, c->rlen.p};
    return poa_consensus(c, in, pp, out);
}

extern "C" int hx_poa_supports(hx_ctx* c, const hx_coords_out* sup, const hx_poa_params* pp, hx_cns_out* out) {
    memset(out, 0, sizeof(*out));
    if (!c->n_reads) return fail("hx_poa_supports: no reads are resident (hx_upload)");
    for (uint64_t k = 0; k < sup->supp_off[sup->n_edge]; k++)
        if ((sup->supp_lr[k] & 0x7fffffffu) >= c->n_reads) return fail("hx_poa_supports: long-read id out of range");
    const PoaInput in{sup->n_edge, sup->supp_off, sup->supp_lr, sup->spos, sup->epos, c->h_rlen.data(), c->packed.p, c->roff.p, c->rlen.p};
    return poa_consensus(c, in, pp, out);
}

extern "C" int hx_poa_sequences(hx_ctx* c, uint32_t n_sets, const uint64_t* set_off, const uint64_t* seq_off, const char* bases, const hx_poa_params* pp, hx_cns_out* out) {
    memset(out, 0, sizeof(*out));
    HIPCHK(hipSetDevice(c->device));
    const uint64_t nseq = set_off[n_sets];
    if (nseq >= 0x7fffffffULL) return fail("hx_poa_sequences: too many sequences");
    // pack like the long reads (2 bits, A0 C1 G2 T3, anything else A; every sequence on a 4-byte boundary) and align them whole, forward
    std::vector<uint32_t> len(nseq), lr(nseq), sp(nseq, 0), ep(nseq);
    std::vector<uint64_t> off(nseq + 1, 0);
    for (uint64_t i = 0; i < nseq; i++) {
        const uint64_t L = seq_off[i + 1] - seq_off[i];
        if (L >= 0xffffffffULL) return fail("hx_poa_sequences: sequence too long");
        len[i] = (uint32_t)L; lr[i] = (uint32_t)i; ep[i] = (uint32_t)L - 1;   // an empty sequence gives epos = spos - 1: skipped, as in the reference (Assemble.cpp:537)
        off[i + 1] = off[i] + ((L + 15) / 16) * 4;
    }
    std::vector<uint8_t> packed(std::max<uint64_t>(4, off[nseq]), 0);
    for (uint64_t i = 0; i < nseq; i++)
        for (uint32_t j = 0; j < len[i]; j++) {
            const char ch = bases[seq_off[i] + j];
            const uint8_t code = ch == 'C' || ch == 'c' ? 1 : ch == 'G' || ch == 'g' ? 2 : ch == 'T' || ch == 't' ? 3 : 0;
            packed[off[i] + (j >> 2)] |= (uint8_t)(code << ((j & 3) * 2));
        }
    DV<uint8_t> d_packed; DV<uint64_t> d_off; DV<uint32_t> d_len;
    if (up(d_packed, packed.data(), packed.size()) || up(d_off, off.data(), off.size()) || up(d_len, len.data(), std::max<size_t>(1, len.size()))) return -1;
    const PoaInput in{n_sets, set_off, lr.data(), sp.data(), ep.data(), len.data(), d_packed.p, d_off.p, d_len.p};
    const int rc = poa_consensus(c, in, pp, out);
    HIPCHK(hipStreamSynchronize(c->stream));
    return rc;
}

extern "C" void hx_free_cns(hx_ctx*, hx_cns_out* o) { free(o->cns_off); free(o->cns); memset(o, 0, sizeof(*o)); }

// ================================================================================================ misc
extern "C" void hx_timing_reset(hx_ctx* c) { for (int i = 0; i < 4; i++) { c->tm.ms[i] = 0; c->tm.launches[i] = 0; } }
extern "C" void hx_timing_get(hx_ctx* c, double* ms, uint64_t* launches) { for (int i = 0; i < 4; i++) { ms[i] = c->tm.ms[i]; launches[i] = c->tm.launches[i]; } }
extern "C" uint32_t hx_poa_phase_cycles(hx_ctx* c, uint64_t* sum6, uint64_t* max6) {
    // lane-0 cycle counters of the last hx_poa_batch: [decode, dp, traceback, graph update+consensus, toposort, csr];
    // sum over edges and the breakdown of the edge with the largest total (the critical path)
    constexpr size_t PW_ = hxk::POA_PHASE_WORDS;
    for (int k = 0; k < 6; k++) { sum6[k] = 0; max6[k] = 0; }
    unsigned long long best = 0;
    size_t ne = c->poa_phase.size() / PW_;
    for (size_t e = 0; e < ne; e++) for (int k = 0; k < 6; k++) if ((long long)c->poa_phase[e * PW_ + k] < 0) c->poa_phase[e * PW_ + k] = 0;   // (a phase that began and ended on different waves' clocks)
    for (size_t e = 0; e < ne; e++) {
        unsigned long long t = 0;
        for (int k = 0; k < 6; k++) { sum6[k] += c->poa_phase[e * PW_ + k]; t += c->poa_phase[e * PW_ + k]; }
        if (t > best) { best = t; for (int k = 0; k < 6; k++) max6[k] = c->poa_phase[e * PW_ + k]; c->dbg_slowest = (uint32_t)e; }
    }
    if (c->opt.debug && ne) {
        const unsigned long long* q = &c->poa_phase[(size_t)c->dbg_slowest * PW_];
        if (c->opt.prof == 1) {   // (a build with -DHX_DP_PROF: where the rows of the first wave of every workgroup spend their cycles, per launch class)
            static const char* seg[6] = {"decode", "predecessors + cells + chain", "wave scan", "carry", "carry applied + ring", "stores"};
            unsigned long long cs[12][7] = {};
            for (size_t e = 0; e < ne; e++) { const int k = e < c->dbg_cls.size() ? c->dbg_cls[e] : 11; for (int j = 0; j < 6; j++) cs[k][j] += c->poa_phase[e * PW_ + 6 + j]; cs[k][6] += c->poa_phase[e * PW_ + 1]; }
            for (int k = 0; k < 12; k++) {
                unsigned long long t = 0; for (int j = 0; j < 6; j++) t += cs[k][j];
                if (!t) continue;
                fprintf(stderr, "[hx] prof1 class %d: row segments of wave 0, %.3g cycles (DP phase %.3g):", k, (double)t, (double)cs[k][6]);
                for (int j = 0; j < 6; j++) fprintf(stderr, " %s %.1f %%%s", seg[j], 100.0 * (double)cs[k][j] / (double)t, j < 5 ? "," : "\n");
            }
            return (uint32_t)ne;
        }
        if (c->opt.prof == 2) {   // (a build with -DHX_DP_PROF -DHX_DP_PROF2: where member 0's DP phase goes, for the five longest edges)
            std::vector<std::pair<unsigned long long, uint32_t>> tt;
            for (size_t e = 0; e < ne; e++) { unsigned long long t = 0; for (int k = 0; k < 6; k++) t += c->poa_phase[e * PW_ + k]; tt.push_back({t, (uint32_t)e}); }
            std::sort(tt.rbegin(), tt.rend());
            for (size_t k = 0; k < std::min<size_t>(5, tt.size()); k++) {
                const unsigned long long* q2 = &c->poa_phase[(size_t)tt[k].second * PW_];
                fprintf(stderr, "[hx] prof2 edge %u lmax=%u nseq=%u dp phase %llu: publish %llu own columns %llu wait members %llu end node %llu (ties sorted %llu, toposort %llu)\n", tt[k].second, c->dbg_lmax[tt[k].second],
                        c->dbg_nseq[tt[k].second], q2[1], q2[6], q2[7], q2[8], q2[9], q2[10], q2[11]);
            }
            return (uint32_t)ne;
        }
        if (c->opt.prof == 3) {   // (a build with -DHX_DP_PROF3: per member of the five longest edges, kilocycles inside the DP and of them waiting for carries)
            std::vector<std::pair<unsigned long long, uint32_t>> tt;
            for (size_t e = 0; e < ne; e++) { unsigned long long t = 0; for (int k = 0; k < 6; k++) t += c->poa_phase[e * PW_ + k]; tt.push_back({t, (uint32_t)e}); }
            std::sort(tt.rbegin(), tt.rend());
            for (size_t k = 0; k < std::min<size_t>(5, tt.size()); k++) {
                const unsigned long long* q2 = &c->poa_phase[(size_t)tt[k].second * PW_];
                fprintf(stderr, "[hx] prof3 edge %u lmax=%u nseq=%u dp %llu:", tt[k].second, c->dbg_lmax[tt[k].second], c->dbg_nseq[tt[k].second], q2[1]);
                for (int m = 0; m < 6; m++) fprintf(stderr, " m%d dp %lluk wait %lluk", m, q2[6 + m] & 0xffffffffull, q2[6 + m] >> 32);
                fprintf(stderr, "\n");
            }
            return (uint32_t)ne;
        }
        const unsigned long long M40 = (1ull << 40) - 1;
        if (c->opt.debug >= 2) {   // every edge: shape of its launch, begin and end on the 100 MHz wall clock (relative to the call's first edge), phase cycles, DP rows
            unsigned long long t0 = ~0ull;
            const unsigned long long M44 = (1ull << 44) - 1;
            for (size_t e = 0; e < ne; e++) if (c->poa_phase[e * PW_ + 16]) t0 = std::min(t0, c->poa_phase[e * PW_ + 16] & M44);
            for (size_t e = 0; e < ne; e++) {
                const unsigned long long* q2 = &c->poa_phase[e * PW_];
                if (!q2[16]) continue;
                const uint32_t sh = e < c->dbg_shape.size() ? c->dbg_shape[e] : 0;
                fprintf(stderr, "[hx-edge] %zu lmax %u nseq %u cls %d lanes %u passes %u members %u hw %u begin_us %.1f end_us %.1f decode %llu dp %llu tb %llu graph %llu order %llu csr %llu rows %llu wrows %llu wskip %llu wbulk %llu cns %llu refcns %llu\n", e, c->dbg_lmax[e], c->dbg_nseq[e],
                        e < c->dbg_cls.size() ? c->dbg_cls[e] : 11, sh & 0xffffu, (sh >> 16) & 255u, sh >> 24, (unsigned)(q2[16] >> 44), (double)((q2[16] & M44) - t0) * 0.01, (double)(q2[17] - t0) * 0.01, q2[0], q2[1], q2[2], q2[3], q2[4], q2[5], q2[6], q2[12], q2[13], q2[20], q2[18], q2[19]);
            }
        }
        fprintf(stderr, "[hx] slowest edge %u: lmax=%u nseq=%u | DP rows %llu (multi-pred %llu, ring refs %llu, far refs %llu, kept %llu, more than 4 predecessors %llu, fifth-and-later entries %llu) over %llu sequences\n", c->dbg_slowest,
                c->dbg_lmax[c->dbg_slowest], c->dbg_nseq[c->dbg_slowest], q[6], q[7], q[8] & M40, q[9] & M40, q[10], q[9] >> 40, q[8] >> 40, q[11] & 0xffffffffull);
        {   // the five longest edges (critical-path candidates)
            std::vector<std::pair<unsigned long long, uint32_t>> tt;
            for (size_t e = 0; e < ne; e++) { unsigned long long t = 0; for (int k = 0; k < 6; k++) t += c->poa_phase[e * PW_ + k]; tt.push_back({t, (uint32_t)e}); }
            std::sort(tt.rbegin(), tt.rend());
            for (size_t k = 0; k < std::min<size_t>(5, tt.size()); k++) {
                const unsigned long long* q2 = &c->poa_phase[(size_t)tt[k].second * PW_];
                fprintf(stderr, "[hx] top edge %u: lmax=%u nseq=%u cycles=%llu (dp %llu tb %llu graph %llu order %llu csr %llu) rows %llu multi %llu ring %llu far %llu kept %llu wide %llu fifth+ %llu\n", tt[k].second, c->dbg_lmax[tt[k].second], c->dbg_nseq[tt[k].second],
                        tt[k].first, q2[1], q2[2], q2[3], q2[4], q2[5], q2[6], q2[7], q2[8] & ((1ull << 40) - 1), q2[9] & ((1ull << 40) - 1), q2[10], q2[9] >> 40, q2[8] >> 40);
            }
        }
        {   // finished graphs against the workspace estimate: nodes per base of the longest sequence, as a + b x sequences
            std::vector<double> grow, fill;
            for (size_t e = 0; e < ne; e++) {
                const double V = (double)(c->poa_phase[e * PW_ + 11] >> 32), L = c->dbg_lmax[e], S = c->dbg_nseq[e];
                if (V <= 0 || L <= 0 || S <= 0) continue;
                grow.push_back((V - L) / (L * S));
                fill.push_back(V / (L * (3 + S / 10) + 1024));
            }
            std::sort(grow.begin(), grow.end()); std::sort(fill.begin(), fill.end());
            auto pc = [](const std::vector<double>& v, double q) { return v.empty() ? 0.0 : v[std::min(v.size() - 1, (size_t)(q * v.size()))]; };
            fprintf(stderr, "[hx] graph growth (nodes - L) / (L x sequences): median %.3f  p90 %.3f  p99 %.3f  max %.3f | nodes / estimate: median %.2f  p99 %.2f  max %.2f\n",
                    pc(grow, 0.5), pc(grow, 0.9), pc(grow, 0.99), pc(grow, 1.0), pc(fill, 0.5), pc(fill, 0.99), pc(fill, 1.0));
        }
        {   // per launch class: how often a row is read back from the LDS ring / from HBM
            unsigned long long cr[12][4] = {};
            for (size_t e = 0; e < ne; e++) { const int k = e < c->dbg_cls.size() ? c->dbg_cls[e] : 11; const unsigned long long* q3 = &c->poa_phase[e * PW_]; cr[k][0] += q3[6]; cr[k][1] += q3[10]; cr[k][2] += q3[8] & ((1ull << 40) - 1); cr[k][3] += q3[9] & ((1ull << 40) - 1); }
            for (int k = 0; k < 12; k++) if (cr[k][0]) fprintf(stderr, "[hx] class %d (ring %u): DP rows %llu, kept %.1f %%, ring refs %.1f %%, far refs %.2f %%\n", k, k < 11 ? c->dbg_ring[k] : 0, cr[k][0], 100.0 * cr[k][1] / cr[k][0], 100.0 * cr[k][2] / cr[k][0], 100.0 * cr[k][3] / cr[k][0]);
            unsigned long long cy[12][4] = {};   // edges, all cycles, DP cycles, longest edge
            for (size_t e = 0; e < ne; e++) {
                const int k = e < c->dbg_cls.size() ? c->dbg_cls[e] : 11; const unsigned long long* q3 = &c->poa_phase[e * PW_];
                unsigned long long t = 0; for (int j = 0; j < 6; j++) t += q3[j];
                cy[k][0]++; cy[k][1] += t; cy[k][2] += q3[1]; cy[k][3] = std::max(cy[k][3], t);
            }
            for (int k = 0; k < 12; k++) if (cy[k][0]) fprintf(stderr, "[hx] class %d: %llu workgroups, %.3e cycles in all (DP %.0f %%), longest %.3e, DP cycles per row %.0f\n", k, cy[k][0], (double)cy[k][1], 100.0 * cy[k][2] / cy[k][1], (double)cy[k][3], cr[k][0] ? (double)cy[k][2] / cr[k][0] : 0.0);
        }
        {   // the pruning (kernels/poa.hip PRUNE): wave-rows of the pruned launches, those skipped, attempts repeated, per launch class
            unsigned long long pr[12][4] = {};
            for (size_t e = 0; e < ne; e++) { const int k = e < c->dbg_cls.size() ? c->dbg_cls[e] : 11; for (int j = 0; j < 4; j++) pr[k][j] += c->poa_phase[e * PW_ + 12 + j]; }
            for (int k = 0; k < 12; k++) if (pr[k][0]) fprintf(stderr, "[hx] class %d pruning: %.4g wave-rows, %.1f %% skipped, %llu alignments with a threshold, %llu repeated\n", k, (double)pr[k][0], 100.0 * pr[k][1] / pr[k][0], pr[k][3], pr[k][2]);
        }
        unsigned long long tot[6] = {0, 0, 0, 0, 0, 0};
        for (size_t e = 0; e < ne; e++) for (int k = 0; k < 6; k++) tot[k] += k == 5 ? (c->poa_phase[e * PW_ + 11] & 0xffffffffull) : (k == 2 || k == 3 ? c->poa_phase[e * PW_ + 6 + k] & ((1ull << 40) - 1) : c->poa_phase[e * PW_ + 6 + k]);
        fprintf(stderr, "[hx] all edges: DP rows %llu (multi-pred %llu, ring refs %llu, far refs %llu, kept %llu) over %llu sequences\n", tot[0], tot[1], tot[2], tot[3], tot[4], tot[5]);
    }
    return (uint32_t)ne;
}
extern "C" uint64_t hx_poa_workspace_bytes(const hx_ctx* c) { return c->poa_workspace_bytes; }
extern "C" int hx_poa_release_workspace(hx_ctx* c) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    { std::lock_guard<std::mutex> lk(c->poa_arena_mu); c->poa_arena.release(); }
    c->poa_budget = 0;   // taken again, from what is free then, by the next consensus call
    return 0;
}
extern "C" int hx_poa_reserve(hx_ctx* c, uint64_t bytes) {
    // the arena of the consensus workspace, ahead of the first call (the CLI: on a thread of its own, beside the parse of the text inputs): at most half of
    // what is free now, so that the inputs still fit beside it whatever the caller guessed; a later call that needs more allocates again
    HIPCHK(hipSetDevice(c->device));
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    std::lock_guard<std::mutex> lk(c->poa_arena_mu);
    uint64_t cap_b = (uint64_t)((double)(free_b + c->poa_arena.cap) * 0.8);                             // (the inputs go beside it: hx_upload gives the arena back if they do not fit)
    if (c->opt.poa_workspace_gb > 0) cap_b = std::min<uint64_t>(cap_b, (uint64_t)(c->opt.poa_workspace_gb * 1.02e9) + (64ull << 20));   // (option poa_workspace_gb: no call will take more)
    const size_t want = (size_t)std::min<uint64_t>(bytes, cap_b);
    if (want <= c->poa_arena.cap) return 0;
    const hipError_t e = c->poa_arena.ensure(want);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(std::string("hx_poa_reserve: ") + hipGetErrorString(e)); }
    if (scratch_warm_once(c)) return -1;
    return 0;
}
extern "C" void hx_poa_host_times(const hx_ctx* c, double* ms8) { for (int k = 0; k < 8; k++) ms8[k] = c->poa_host_ms[k]; }
extern "C" void hx_poa_arena_stats(const hx_ctx* c, uint64_t* capacity, uint64_t* allocations, double* alloc_ms) { *capacity = c->poa_arena.cap; *allocations = c->poa_arena.n_alloc; *alloc_ms = c->poa_arena.alloc_ms; }
extern "C" void hx_poa_memory_stats(const hx_ctx* c, uint64_t* free_at_first_call, uint64_t* budget, uint64_t* last_call_workspace) {
    *free_at_first_call = c->poa_free_at_first_call; *budget = c->poa_budget; *last_call_workspace = c->poa_last_workspace_bytes;
}
extern "C" void hx_poa_prune_stats(const hx_ctx* c, uint64_t* out4) {
    for (int j = 0; j < 4; j++) out4[j] = 0;
    const size_t PW_ = hxk::POA_PHASE_WORDS, ne = c->poa_phase.size() / PW_;
    for (size_t e = 0; e < ne; e++) for (int j = 0; j < 4; j++) out4[j] += c->poa_phase[e * PW_ + 12 + j];
}
extern "C" void hx_set_poa_traceback(hx_ctx* c, int use_direction_bytes) { c->poa_no_dir = !use_direction_bytes; }
extern "C" void hx_set_poa_block(hx_ctx* c, int t) { c->poa_block = t <= 0 ? 0 : t >= 1024 ? 1024 : t >= 512 ? 512 : t >= 256 ? 256 : t >= 128 ? 128 : 64; }

static int be_chain(void* p, const hx_params* a, hx_chain_out* o) { return hx_chain_reads((hx_ctx*)p, a, o); }
static int be_edges(void* p, const hx_params* a, hx_edges_out* o) { return hx_edge_support((hx_ctx*)p, a, o); }
static int be_coords(void* p, uint32_t n, const uint32_t* s, hx_coords_out* o) { return hx_edge_coords((hx_ctx*)p, n, s, o); }
static int be_poa(void* p, const hx_poa_params* a, hx_cns_out* o) { return hx_poa_batch((hx_ctx*)p, a, o); }
static void be_fc(void* p, hx_chain_out* o) { hx_free_chain((hx_ctx*)p, o); }
static void be_fe(void* p, hx_edges_out* o) { hx_free_edges((hx_ctx*)p, o); }
static void be_fk(void* p, hx_coords_out* o) { hx_free_coords((hx_ctx*)p, o); }
static void be_fn(void* p, hx_cns_out* o) { hx_free_cns((hx_ctx*)p, o); }

extern "C" void hx_backend_fill(hx_ctx* c, void* table) {
    hx_backend* b = (hx_backend*)table;
    b->ctx = c; b->chain_reads = be_chain; b->edge_support = be_edges; b->edge_coords = be_coords; b->poa_batch = be_poa;
    b->free_chain = be_fc; b->free_edges = be_fe; b->free_coords = be_fk; b->free_cns = be_fn; b->last_error = hx_last_error;
}

// ================================================================================================ multi-GPU inside one process
// What asm_calc_edge_coordinates_MT / asm_cal_cns_seq_MT (Assemble.cpp:453-477, :580-605; called from main.cpp:203-208) are to the reference -
// a fan-out of the per-read / per-edge work over the threads of ONE process - this is to the GPUs of one node: a group of contexts (one per
// device, one host thread each) with one RCCL communicator each (ncclCommInitAll), and ONE collective on the data path: the all-gather of
// the packed edge-support records between the chain stage and the key sort (hx_edge_merge). librccl is looked up at run time (it is half a
// gigabyte: a single-GPU run never maps it). HASLR_GROUP_TRANSPORT=host stages the exchange through host memory instead, which also allows
// several ranks on one device (rehearsal of the multi-GPU logic on a one-GPU box).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <thread>
#include <memory>
#include <mutex>

namespace {
struct GroupRank { hx_group* g; int rank; };
}
struct hx_group {
    int n = 0;
    std::vector<hx_ctx*> ctx;
    std::vector<int> dev;
    std::vector<GroupRank> self;             // opaque `ctx` of the ranks' backend tables
    bool rccl = false;
    void* lib = nullptr;
    std::vector<ncclComm_t> comm;
    ncclResult_t (*p_init_all)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*p_all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*p_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*p_abort)(ncclComm_t) = nullptr;
    ncclResult_t (*p_count)(const ncclComm_t, int*) = nullptr;
    const char* (*p_errstr)(ncclResult_t) = nullptr;
    std::atomic<int> abort_flag{0};               // a rank failed inside the collective: the ranks still waiting on their streams abort their communicators
    std::atomic<bool> broken{false};              // ... after which the group refuses further exchanges (written and read by the rank threads)
    double timeout_s = 300;                       // bound of the wait for the collective (hx_group_set_timeout)
    int fault_rank = -1;                          // (testing, hx_group_inject_fault: this rank's all-gather "returns an error")
    // rendezvous of the rank threads: everybody arrives with a status, everybody leaves with the worst one (so that no rank enters a
    // collective the others will never join)
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0, worst = 0, agreed = 0;
    uint64_t generation = 0;
    std::vector<uint64_t> counts;
    std::vector<std::unique_ptr<DV<uint8_t>>> sendb, recvb, merged;
    std::vector<std::vector<uint8_t>> stage;      // host transport
    uint64_t last_bytes = 0;
    double last_ms = 0;

    int rendezvous(int status) {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t gen = generation;
        worst = std::max(worst, status);
        if (++arrived == n) { agreed = worst; worst = 0; arrived = 0; generation++; cv.notify_all(); return agreed; }
        cv.wait(lk, [&] { return generation != gen; });
        return agreed;
    }
};

extern "C" int hx_group_create(int n, const int* devices, const char* transport, hx_group** out) {
    *out = nullptr;
    int ndev = 0;
    if (n < 1) return fail("hx_group_create: at least one rank");
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("hx_group_create: no HIP device available (no CPU fallback)");
    std::unique_ptr<hx_group> g(new hx_group);
    g->n = n;
    const char* tr = transport && *transport ? transport : nullptr;   // (the applications pass what their HASLR_GROUP_TRANSPORT says: the library reads no environment)
    bool distinct = true;
    for (int r = 0; r < n; r++) {
        const int d = devices ? devices[r] : (tr && !strcmp(tr, "host") ? r % ndev : r);
        if (d < 0 || d >= ndev) return fail("hx_group_create: rank " + std::to_string(r) + " asks for device " + std::to_string(d) + " of " + std::to_string(ndev) +
                                            " (one device per rank over RCCL; transport \"host\" lets ranks share devices)");
        for (int q : g->dev) distinct = distinct && q != d;
        g->dev.push_back(d);
    }
    if (tr && strcmp(tr, "host") && strcmp(tr, "rccl")) return fail("hx_group_create: transport must be \"rccl\" or \"host\" (or NULL: automatic)");
    g->rccl = tr ? !strcmp(tr, "rccl") : distinct;
    if (g->rccl && !distinct) return fail("hx_group_create: RCCL needs one device per rank");
    g->ctx.assign(n, nullptr);
    for (int r = 0; r < n; r++)
        if (hx_ctx_create(g->dev[r], nullptr, &g->ctx[r]) != 0) { for (hx_ctx* c : g->ctx) hx_ctx_destroy(c); return -1; }
    g->self.resize(n);
    for (int r = 0; r < n; r++) g->self[r] = GroupRank{g.get(), r};
    g->counts.assign(n, 0); g->stage.resize(n);
    for (int r = 0; r < n; r++) { g->sendb.emplace_back(new DV<uint8_t>); g->recvb.emplace_back(new DV<uint8_t>); g->merged.emplace_back(new DV<uint8_t>); }
    if (g->rccl) {
        g->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!g->lib) g->lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!g->lib) g->lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!g->lib) { const std::string m = std::string("hx_group_create: cannot load librccl: ") + dlerror(); for (hx_ctx* c : g->ctx) hx_ctx_destroy(c); return fail(m); }
        g->p_init_all = (decltype(g->p_init_all))dlsym(g->lib, "ncclCommInitAll");
        g->p_all_gather = (decltype(g->p_all_gather))dlsym(g->lib, "ncclAllGather");
        g->p_destroy = (decltype(g->p_destroy))dlsym(g->lib, "ncclCommDestroy");
        g->p_errstr = (decltype(g->p_errstr))dlsym(g->lib, "ncclGetErrorString");
        g->p_abort = (decltype(g->p_abort))dlsym(g->lib, "ncclCommAbort");
        g->p_count = (decltype(g->p_count))dlsym(g->lib, "ncclCommCount");
        if (!g->p_init_all || !g->p_all_gather || !g->p_destroy || !g->p_errstr) { for (hx_ctx* c : g->ctx) hx_ctx_destroy(c); return fail("hx_group_create: librccl lacks ncclCommInitAll / ncclAllGather"); }
        g->comm.assign(n, nullptr);
        const ncclResult_t rc = g->p_init_all(g->comm.data(), n, g->dev.data());
        if (rc != ncclSuccess) { const std::string m = std::string("ncclCommInitAll: ") + g->p_errstr(rc); for (hx_ctx* c : g->ctx) hx_ctx_destroy(c); return fail(m); }
    }
    *out = g.release();
    return 0;
}

extern "C" void hx_group_destroy(hx_group* g) {
    if (!g) return;
    // (a group whose collective failed has had its communicators aborted by their own ranks - hx_edge_merge - and whatever is left of it is aborted too:
    // ncclCommDestroy on a communicator whose peers are gone may wait for them)
    if (g->rccl) for (int r = 0; r < g->n; r++) if (g->comm[r]) { (void)hipSetDevice(g->dev[r]); if (g->broken.load() && g->p_abort) (void)g->p_abort(g->comm[r]); else (void)g->p_destroy(g->comm[r]); g->comm[r] = nullptr; }
    for (int r = 0; r < g->n; r++) { (void)hipSetDevice(g->dev[r]); g->sendb[r]->release(); g->recvb[r]->release(); g->merged[r]->release(); }
    for (hx_ctx* c : g->ctx) hx_ctx_destroy(c);
    // (librccl stays mapped: unloading it while the HIP runtime is alive buys nothing)
    delete g;
}
extern "C" int hx_group_size(const hx_group* g) { return g->n; }
extern "C" void hx_group_inject_fault(hx_group* g, int rank) { g->fault_rank = rank; }
extern "C" void hx_group_set_timeout(hx_group* g, double seconds) { g->timeout_s = seconds > 0 ? seconds : 300; }
extern "C" hx_ctx* hx_group_ctx(hx_group* g, int rank) { return rank >= 0 && rank < g->n ? g->ctx[rank] : nullptr; }
extern "C" const char* hx_group_transport(const hx_group* g) { return g->rccl ? "rccl" : "host"; }
extern "C" int hx_group_rccl_ranks(const hx_group* g, int* out) {   // what ncclCommCount says on every rank's communicator (0 for every rank: host transport)
    for (int r = 0; r < g->n; r++) {
        out[r] = 0;
        if (g->rccl && g->p_count && g->comm[r] && g->p_count(g->comm[r], &out[r]) != ncclSuccess) out[r] = -1;
    }
    return g->rccl ? 1 : 0;
}
extern "C" void hx_group_exchange_stats(const hx_group* g, uint64_t* bytes, double* ms) { *bytes = g->last_bytes; *ms = g->last_ms; }

extern "C" int hx_edge_merge(hx_group* g, int rank, const hx_params* prm, hx_edges_out* out) {
    memset(out, 0, sizeof(*out));
    if (rank < 0 || rank >= g->n) return fail("hx_edge_merge: rank out of range");
    if (g->broken.load()) return fail("hx_edge_merge: the group's collective failed earlier (communicators aborted): create a new group");
    hx_ctx* c = g->ctx[rank];
    const uint32_t rb = hx_edge_records_bytes();
    uint64_t n = 0;
    int rc = hx_edge_emit(c, prm, &n);
    g->counts[rank] = rc == 0 ? n : 0;
    std::string own_err = rc ? g_err : std::string();
    if (g->rendezvous(rc != 0)) return fail(rc ? own_err : "hx_edge_merge: another rank failed to emit its edge records");
    uint64_t cap_rec = 1, total = 0;
    bool equal = true;
    for (int r = 0; r < g->n; r++) { cap_rec = std::max(cap_rec, g->counts[r]); total += g->counts[r]; equal = equal && g->counts[r] == g->counts[0]; }
    const uint64_t cap = cap_rec * rb;
    DV<uint8_t>&sb = *g->sendb[rank], &rv = *g->recvb[rank];
    rc = 0;
    if (hipSetDevice(c->device) != hipSuccess || sb.reserve(cap) != hipSuccess || rv.reserve(cap * g->n) != hipSuccess) { rc = -1; own_err = "hx_edge_merge: out of device memory for the exchange buffers"; }
    if (!rc && hx_edge_records_export(c, sb.p, cap_rec) != 0) { rc = -1; own_err = g_err; }
    if (g->rendezvous(rc != 0)) return fail(rc ? own_err : "hx_edge_merge: another rank failed before the exchange");
    const auto t0 = std::chrono::steady_clock::now();
    if (g->rccl) {
        // THE collective of the path: every rank contributes its packed records padded to the largest shard (counts travelled through the
        // process's memory above: the ranks are threads of one process)
        // A failure INSIDE the collective must not leave the other ranks parked on their streams: the wait is a bounded poll of the stream; a rank
        // whose ncclAllGather returns an error (or whose stream faults, or whose wait runs out) raises the group's abort flag, every rank that sees it
        // aborts its communicator (ncclCommAbort ends the kernels of the collective on its device) and all of them meet at the rendezvous below with
        // the failure. The group is unusable afterwards (hx_edge_merge refuses).
        const bool injected = g->fault_rank == rank;
        if (injected) g->fault_rank = -1;   // (one shot)
        const ncclResult_t nr = injected ? ncclInternalError : g->p_all_gather(sb.p, rv.p, cap, ncclUint8, g->comm[rank], c->stream);
        if (nr != ncclSuccess) { rc = -1; own_err = std::string("ncclAllGather: ") + g->p_errstr(nr); g->abort_flag.store(1); }
        else {
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(g->timeout_s);
            for (;;) {
                const hipError_t q = hipStreamQuery(c->stream);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) { rc = -1; own_err = std::string("hx_edge_merge: all-gather failed on the stream: ") + hipGetErrorString(q); g->abort_flag.store(1); break; }
                const bool late = std::chrono::steady_clock::now() > deadline;
                if (g->abort_flag.load() || late) {
                    rc = -1; own_err = late ? "hx_edge_merge: the all-gather did not finish within " + std::to_string((int)g->timeout_s) + " s (hx_group_set_timeout)" : "hx_edge_merge: another rank failed inside the all-gather";
                    g->abort_flag.store(1);
                    if (g->p_abort && g->comm[rank]) { (void)g->p_abort(g->comm[rank]); g->comm[rank] = nullptr; }
                    (void)hipStreamSynchronize(c->stream);
                    break;
                }
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        }
        } else {
        g->stage[rank].resize(cap);
        if (hipMemcpy(g->stage[rank].data(), sb.p, cap, hipMemcpyDeviceToHost) != hipSuccess) { rc = -1; own_err = "hx_edge_merge: copy to the host staging buffer failed"; }
        if (g->rendezvous(rc != 0)) return fail(rc ? own_err : "hx_edge_merge: another rank failed in the exchange");
        for (int r = 0; r < g->n && !rc; r++)
            if (hipMemcpy(rv.p + (uint64_t)r * cap, g->stage[r].data(), cap, hipMemcpyHostToDevice) != hipSuccess) { rc = -1; own_err = "hx_edge_merge: copy from the host staging buffer failed"; }
    }
    if (g->rendezvous(rc != 0)) {
        if (g->rccl) {
            // EVERY rank leaves a failed collective with its own communicator aborted - the rank whose call returned the error and the ranks whose part
            // had already completed included (their peers are gone: ncclCommDestroy on such a communicator may wait for them) - and the group refuses
            // further exchanges
            g->broken.store(true);
            if (g->p_abort && g->comm[rank]) { (void)g->p_abort(g->comm[rank]); g->comm[rank] = nullptr; }
            (void)hipStreamSynchronize(c->stream);
        }
        return fail(rc ? own_err : "hx_edge_merge: another rank failed in the exchange");
    }
    if (rank == 0) { g->last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); g->last_bytes = total * rb; }
    const uint8_t* src = rv.p;
    if (!equal) {   // cut the padding out: rank order = ascending read ids, which the stable key sort relies on
        DV<uint8_t>& mg = *g->merged[rank];
        if (mg.reserve(std::max<uint64_t>(1, total * rb)) != hipSuccess) { rc = -1; own_err = "hx_edge_merge: out of device memory for the merged records"; }
        uint64_t off = 0;
        for (int r = 0; r < g->n && !rc; r++) {
            if (g->counts[r] && hipMemcpyAsync(mg.p + off, rv.p + (uint64_t)r * cap, g->counts[r] * rb, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) { rc = -1; own_err = "hx_edge_merge: compaction failed"; }
            off += g->counts[r] * rb;
        }
        if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) { rc = -1; own_err = "hx_edge_merge: compaction failed"; }
        src = mg.p;
    }
    if (!rc && hx_edge_records_import(c, src, total, out) != 0) { rc = -1; own_err = g_err; }
    if (g->rendezvous(rc != 0)) { if (!rc) hx_free_edges(c, out); return fail(rc ? own_err : "hx_edge_merge: another rank failed to import the merged records"); }
    return 0;
}

static int gb_chain(void* p, const hx_params* a, hx_chain_out* o) { GroupRank* q = (GroupRank*)p; return hx_chain_reads(q->g->ctx[q->rank], a, o); }
static int gb_edges(void* p, const hx_params* a, hx_edges_out* o) { GroupRank* q = (GroupRank*)p; return hx_edge_merge(q->g, q->rank, a, o); }
static int gb_coords(void* p, uint32_t n, const uint32_t* s, hx_coords_out* o) { GroupRank* q = (GroupRank*)p; return hx_edge_coords(q->g->ctx[q->rank], n, s, o); }
static int gb_poa(void* p, const hx_poa_params* a, hx_cns_out* o) { GroupRank* q = (GroupRank*)p; return hx_poa_batch(q->g->ctx[q->rank], a, o); }
static void gb_fc(void* p, hx_chain_out* o) { GroupRank* q = (GroupRank*)p; hx_free_chain(q->g->ctx[q->rank], o); }
static void gb_fe(void* p, hx_edges_out* o) { GroupRank* q = (GroupRank*)p; hx_free_edges(q->g->ctx[q->rank], o); }
static void gb_fk(void* p, hx_coords_out* o) { GroupRank* q = (GroupRank*)p; hx_free_coords(q->g->ctx[q->rank], o); }
static void gb_fn(void* p, hx_cns_out* o) { GroupRank* q = (GroupRank*)p; hx_free_cns(q->g->ctx[q->rank], o); }

extern "C" int hx_group_backend_fill(hx_group* g, int rank, void* table) {
    if (rank < 0 || rank >= g->n) return fail("hx_group_backend_fill: rank out of range");
    hx_backend* b = (hx_backend*)table;
    b->ctx = &g->self[rank]; b->chain_reads = gb_chain; b->edge_support = gb_edges; b->edge_coords = gb_coords; b->poa_batch = gb_poa;
    b->free_chain = gb_fc; b->free_edges = gb_fe; b->free_coords = gb_fk; b->free_cns = gb_fn; b->last_error = hx_last_error;
    return 0;
}
