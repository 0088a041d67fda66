// poa.hip — K6: per-edge partial-order-alignment consensus.
//
// Replaces the SPOA calls of asm_calc_single_cns_seq (Assemble.cpp:499-554): for every backbone edge the gap
// sub-sequences of its supporting long reads are aligned one after the other (global NW, linear gap,
// +5/-4/-8) to a growing partial-order graph, and the heaviest-bundle path is the consensus. Semantics follow
// the published rvaser/spoa 1.1.3 algorithm as restated in oracle/oracle.cpp (same recurrences, same
// tie-breaking, same graph update and topological order), so results are bit-identical to the oracle.
//
// Mapping (details in DESIGN.md "K6 in detail"): one workgroup per edge, or 2-16 cooperating workgroups ("members", one CU each) for
// gaps above 2047 columns - for the few costliest edges of a small call WIDE members: 1024 lanes of which the first 256 run the DP, all of
// them the graph phases, one spare wave relays the carries arriving through HBM; sequences of an edge are aligned one after the other,
// edges run concurrently (persistent workgroups pulling edges off a counter when a launch class has more edges than workspace slots).
//   * sequence k is decoded from the 2-bit packed read arena into a byte row
//   * DP (dp_rows): rows = graph nodes in topological order, columns = sequence positions, CM contiguous columns per lane kept in
//     registers. Cells are keys (64 x DE-RAMPED score, H - gap x column, + 6 tie-break bits), so one max() per decision reproduces the
//     reference's tie rules and the low bits are the traceback's direction nibble. Horizontal recurrence = a plain prefix maximum: one DPP
//     scan of the chunks' largest keys, then one lane-serial pass from the finished key on the left; the waves of an edge form a pipeline
//     (tagged mailboxes in LDS between waves, one tagged word per row in HBM between members, no barrier inside the DP). Rows needed
//     later as non-adjacent predecessors live in an LDS ring, the overflow in HBM. The row loop is written against what ONE wave alone on
//     its SIMD pays per instruction kind (tools/dev_lonebench.hip): scalar flags, one test for the rare cases, buffer-resource stores.
//   * traceback: the first wavefront walks 32x16 tiles of direction nibbles with v_readlane; a second wavefront, where the workgroup has
//     one, touches the lines the walk reaches next (cache warm-up only)
//   * graph update (spoa add_alignment), order update and the rank-ordered CSR rebuild run on all lanes (prefix sums for the ids the
//     serial walk would hand out); the reference's DFS topological sort runs only for end-node ties that a column cannot decide and
//     for heaviest bundles that do not end in a unique sink (one wavefront in lock step, on ranks).
#include <type_traits>

#include "kernels.h"

namespace hxk {

namespace {

#include "poa_graph.inl"    // the graph of an edge, the reference's topological order, heaviest bundle
#include "poa_dp.inl"       // scans, stores, the wave pipeline's mailboxes, the row loop
#include "poa_update.inl"   // CSR rebuild, graph update, order update

// One kernel per (largest workgroup, columns per lane, traceback flavour): the register budget of a launch is that of ITS row loop, so the
// many short gaps (one wavefront, 4-8 columns per lane) run with a fraction of the registers - and several times the waves per SIMD - of
// the few long ones; sequences shorter than the edge's longest leave the upper lanes / waves of the pipeline idle.
template <int MAXNT, int CM, bool DIR, bool PRUNE>
__device__ __forceinline__ void poa_edge(const uint32_t eidx, const uint32_t mem, const PoaSlot SL, const PoaEdge* __restrict__ edges,
                                         const PoaSeq* __restrict__ seqs, const uint8_t* __restrict__ packed, const uint64_t* __restrict__ read_off,
                                         const uint32_t* __restrict__ read_len, const PoaPools& P, int32_t match, int32_t mismatch, int32_t gap,
                                         char* cns, uint32_t* cns_len, uint32_t* status, unsigned long long* cells, unsigned long long* phase,
                                         uint32_t poll_limit, uint32_t lds_bytes, uint32_t max_indeg, uint32_t dp_lanes, uint32_t prune_pct) {
    __shared__ unsigned long long ph[POA_PHASE_WORDS];   // lane-0 cycle counts: decode, dp, traceback, graph update, toposort, csr; then row statistics (6-11) and the pruning's (12-15)
    __shared__ long long tc;
    if (threadIdx.x == 0) { for (int k = 0; k < POA_PHASE_WORDS; k++) ph[k] = 0; tc = clock64(); ph[16] = (wall_clock64() & ((1ull << 44) - 1)) | ((unsigned long long)((__builtin_amdgcn_s_getreg(63492) & 0x7fffu) | ((__builtin_amdgcn_s_getreg(63508) & 15u) << 15)) << 44); }   // (begin; where: HW_ID bits 0-14 = wave, SIMD, pipe, CU, SH, SE and the XCC)
#define PHASE(k) do { if (tid == 0) { long long _n = clock64(); ph[k] += (unsigned long long)(_n - tc); tc = _n; } } while (0)
#ifdef HX_GU_PROF   // development: where the graph update (slots 6-10) and the CSR rebuild (slot 11: its first half) spend their cycles - printed by HX_PROF2 (its labels are the DP's)
#define GU_T0() do { __syncthreads(); if (tid == 0) tg = clock64(); } while (0)
#define GU_T(k) do { __syncthreads(); if (tid == 0) { long long _n = clock64(); ph[k] += (unsigned long long)(_n - tg); tg = _n; } } while (0)
    long long tg = 0;
#else
#define GU_T0() do { } while (0)
#define GU_T(k) do { } while (0)
#endif
#if defined(HX_DP_PROF2) && !defined(HX_DP_PROF3)
    __shared__ long long tc2;
#define SUBT(k) do { if (tid == 0) { long long _n = clock64(); ph[k] += (unsigned long long)(_n - tc2); tc2 = _n; } } while (0)
#define SUBT0() do { if (tid == 0) tc2 = clock64(); } while (0)
#else
#define SUBT(k) do { } while (0)
#define SUBT0() do { } while (0)
#endif
    const PoaEdge ED = edges[eidx];
    const uint32_t tid = threadIdx.x, NT = blockDim.x;
    const uint32_t DL = dp_lanes ? dp_lanes : NT;   // lanes in the DP: the whole workgroup, or the first waves of a "wide" cluster member (one wave per SIMD in the DP, sixteen in the graph phases)
    extern __shared__ int32_t ring[];   // (no alignment attribute: the packed rows' table reads are split by the compiler; with aligned(16) the int32 row loop of this build came out 10 % slower - same instructions, another layout)
    G g;
    {
        const uint64_t no = SL.node_off, eo = SL.edge_off;
        g.code = P.code + no; g.n_aligned = P.n_aligned + no; g.aligned = P.aligned + 3 * no;
        g.in_head = P.in_head + no; g.in_tail = P.in_tail + no; g.out_head = P.out_head + no; g.out_tail = P.out_tail + no;
        g.rank2node = P.rank2node + no; g.node2rank = P.node2rank + no; g.mark = P.mark + no; g.check = P.check + no;
        g.stack = P.stack + SL.stack_off; g.score = P.score + no; g.pred = P.pred + no;
        g.row_code = P.row_code + no; g.row_sink = P.row_sink + no; g.row_pred_off = P.row_pred_off + no; g.pred_rank = P.pred_rank + eo; g.pred_w = P.pred_w + eo;
        g.row_meta = P.row_meta + no; g.row_pred0 = P.row_pred0 + no; g.row_pred1 = P.row_pred1 + no; g.nrec = P.nrec + no; g.nrec2 = P.nrec2 + no; g.row_al = P.row_al + no; g.wslot = P.wslot + no;
        g.e_from = P.e_from + eo; g.e_to = P.e_to + eo; g.e_next_in = P.e_next_in + eo; g.e_next_out = P.e_next_out + eo; g.e_w = P.e_w + eo;
        g.aln_node = P.aln_node + SL.aln_off; g.aln_pos = P.aln_pos + SL.aln_off;
        g.vcap = ED.vcap; g.ecap = ED.ecap;
    }
    int32_t* H = P.H + SL.h_off;
    uint8_t* Dm = DIR ? P.dir + SL.d_off : nullptr;   // direction nibbles: (vcap + 1) rows of W / 2 bytes; with them H holds only ED.hrows far-read rows
    uint8_t* Dw = DIR ? P.dirw + SL.w_off : nullptr;  // direction bytes of the rows with more than 4 predecessors: ED.wrows rows of W
    // ring geometry is a property of the edge (its longest sequence) and of the launch
    const uint32_t GM = ED.members;                  // workgroups sharing this edge's DP columns
    const uint32_t ring_w = CM * (DL >> 6) * 65u;    // planes of 65 words per wave: one per column (dp_rows)
    // kept rows the LDS ring holds for THIS edge: what fits the launch's LDS at the edge's own row width (a launch serves edges of several
    // widths; the host sizes the LDS for the widest), a power of two (slot = kept-row counter & (R - 1)). Only rows with a non-adjacent
    // reader go there (the previous row is read from registers), so every slot holds a kept row.
    uint32_t R = 0;
    {
        const uint32_t fit = lds_bytes / (ring_w * 4u);
        R = fit >= 8 ? 8 : fit >= 4 ? 4 : fit >= 2 ? 2 : 0;   // (0: rows too wide for two of them - every kept row is read back from HBM)
        if (fit < 1) R = 0xffffffffu;
    }
    uint8_t* seq = P.seq + SL.seq_off;
    const uint32_t W = (ED.lmax + 1 + 31) & ~31u;   // row stride: a multiple of the widest lane chunk (32 columns), so chunks are vector-aligned and stay inside their row
    // Column passes (round 5): an unshared edge whose sequences are wider than its workgroup takes the DP columns in NP windows of DL lanes x CM columns, one
    // after the other - the same pipeline as NP cluster members (member p = window p: the carries of a window's last column travel through the HBM
    // mailbox, complete before the next window starts), run by ONE workgroup. With the pruned rows most of a window's rows are skipped in bulk, so a
    // pass costs little more than the rows its window shares with the live band, and the edge holds NP times fewer wave slots while it runs.
    const uint32_t NP = GM == 1 && ED.passes > 1 ? ED.passes : 1u;
    const uint32_t WT = GM * NP * (DL >> 6);             // waves of the edge's whole pipeline
    const uint32_t WH = W + (WT > 1 ? (WT + 3u) & ~3u : 0u);   // rows of H end with one word per wave of the edge's pipeline (dp_rows)

    // static LDS is kept small for the launches that can share a CU: sink rows kept in LDS (an alignment ends in at most one sink per sequence
    // aligned so far; an edge with more than the launch keeps is redone by the 1024-lane kernel), wave mailboxes for the waves the launch can have
    // (128 entries up to 256 lanes: with the 8.3 KB ring of a many-edge call a one-wave workgroup then takes 10 128 bytes of LDS - sixteen of them on a CU, where
    // 256 entries left room for fourteen; round 5)
    constexpr uint32_t SINK_LDS = MAXNT <= 256 ? 128 : MAXNT < 1024 ? 256 : SINK_CAP;
    __shared__ WaveMailT<MAXNT / 64> wmail;
    __shared__ uint32_t lds_u[16];
    __shared__ uint32_t sV, sE, sNaln, sOk, sNsink, sNcand, sBestKey, sWalkN, sWalkI, sWalkJ;   // (sWalk*: entries of the traceback walk still in (rank, column) form, and where the walk stopped)
    __shared__ int sBestI;
    __shared__ uint32_t sink_row[SINK_LDS];
    __shared__ int sink_score[SINK_LDS];
    __shared__ uint32_t sCtl;
    __shared__ unsigned long long sCells;   // DP cells of this edge (reported only when the edge completes: retried edges count once)
    __shared__ int sPrevScore, sNewT; __shared__ uint32_t sPrevLen, sRetry;   // PRUNE: score and length of the edge's previous alignment (the source of the threshold), the verdict on an attempt
    if (tid == 0) { sV = 0; sE = 0; sOk = R != 0xffffffffu ? 1 : 2; sCells = 0; sPrevScore = 0; sPrevLen = 0; sRetry = 0; sNewT = PRUNE_OFF; }   // (the host gives every launch LDS for at least the latest row)
    if constexpr (MAXNT > 64) {
        for (uint32_t q = tid; q < (MAXNT / 64 - 1) * WAVE_MBOX; q += NT) wmail.box[q] = 0ull;   // tag 0 = nothing published
        if (tid < MAXNT / 64) wmail.consumed[tid] = 0u;
    }
    __syncthreads();
    uint32_t* csy = P.csync + (uint64_t)eidx * 8;                 // go, done, V, L, error
    int32_t* sinkbuf = P.sinkbuf + (uint64_t)eidx * (1 + 2 * SINK_CAP);
    DpCl cl;
    cl.mem = mem; cl.members = GM; cl.stride = ED.vcap + 1; cl.tag0 = 0;
    cl.mbox = P.mbox + (NP > 1 ? SL.mbox_off : ED.cl_off); cl.err = csy + 4; cl.poll_limit = poll_limit; cl.lanes = DL;
    constexpr uint32_t CL_ABORT = 0xffffffffu;
#define HX_DP_DISPATCH(Lq, Vq, nsq, Tq) do { \
        if (((Lq) + 1 + GM * NP * DL - 1) / (GM * NP * DL) <= (uint32_t)CM) {    /* the host puts an edge into a launch whose columns per lane hold its longest sequence */ \
            dp_rows<CM, DIR, PRUNE, MAXNT == 64>(g, H, Dm, Dw, W, WH, seq, Lq, Vq, ring, R, ring_w, match, mismatch, gap, wmail.box, wmail.consumed, sink_row, sink_score, SINK_LDS, nsq, cl, ph + 6, Tq, (prune_pct >> 16) & 1u, ph + 12, ED.hrows); \
        } else sOk = 2; } while (0)
    // The reference's topological order (spoa's DFS, inherently serial) is needed in two places only: to break ties between equally scored
    // end nodes of an alignment, and for the heaviest-bundle traversal of the finished graph. The DP itself runs on a cheaper order that
    // is maintained incrementally (see "order update" below): row values do not depend on which valid topological order is used.
    // Geometry of the wave's DFS (toposort_rank) in the LDS the ring leaves: a stack window and a cache of 16-record lines, as large as the launch's LDS allows (a
    // many-edge call gives a one-wave workgroup 8.3 KB), then the state bytes of the ranks - in LDS when they fit, else in global memory (one more round trip per
    // visit: still three to four times fewer than the one-lane walk over the node lists, which only a launch with less than 3 KB of LDS falls back to. Round 5: 12 %
    // of the edges of a 13 000-edge call need the reference's order for their consensus, and with the full geometry only - 21 KB + a byte per node - they almost
    // all took the one-lane walk: up to 370 M cycles at the end of the longest chains, 6 % of the call's wave cycles).
    const uint32_t topo_lcap = lds_bytes >= 24 * 1024 ? 1024u : lds_bytes >= 8 * 1024 ? 256u : 128u;
    const uint32_t topo_lines = lds_bytes >= 24 * 1024 ? 64u : lds_bytes >= 8 * 1024 ? 16u : 8u;
    const uint32_t topo_fixed = topo_lcap * 4 + topo_lines * 256 + topo_lines * 4;
    uint32_t* t_stack = reinterpret_cast<uint32_t*>(ring);
    uint4* t_cache = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(ring) + topo_lcap * 4);
    uint32_t* t_tags = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(ring) + topo_lcap * 4 + topo_lines * 256);
    uint8_t* st_lds = reinterpret_cast<uint8_t*>(ring) + topo_fixed;
    uint32_t* tmp_u32 = reinterpret_cast<uint32_t*>(g.pred);   // vcap+1 words of scratch (heaviest-bundle scratch, free until the end)
    auto exact_order = [&](uint32_t Vn, uint32_t* out) {   // all lanes; leaves spoa's rank->node order of the current graph in out[]
        // (needs the rank-ordered rows of the CURRENT graph: they are rebuilt after every sequence)
        const bool wave_dfs = topo_fixed + 16 <= lds_bytes;
        const bool st_in_lds = wave_dfs && (uint64_t)Vn + topo_fixed + 16 <= lds_bytes;
        uint8_t* st = st_in_lds ? st_lds : g.mark;
        if (wave_dfs) { for (uint32_t i = tid; i < Vn; i += NT) st[i] = 4u; }                      // mark 0, check 1
        else { for (uint32_t i = tid; i < Vn; i += NT) { g.mark[i] = 0; g.check[i] = 1; } }
        __threadfence_block();
        __syncthreads();
        if (wave_dfs) {
            uint32_t* ranks = reinterpret_cast<uint32_t*>(g.score);   // free between the CSR build and the graph update
#if defined(HX_DP_PROF2) && !defined(HX_DP_PROF3)
            long long tq0 = clock64();
#endif
            if (tid < 64) toposort_rank(g, Vn, st, t_stack, t_cache, t_tags, ranks, topo_lcap, topo_lines);   // wave 0, 64 lanes in lock step
#if defined(HX_DP_PROF2) && !defined(HX_DP_PROF3)
            if (tid == 0) ph[11] += (unsigned long long)(clock64() - tq0);
#endif
            __threadfence_block();
            __syncthreads();
            for (uint32_t i = tid; i < Vn; i += NT) tmp_u32[i] = g.rank2node[ranks[i]];
            __syncthreads();
            if (out != tmp_u32) { for (uint32_t i = tid; i < Vn; i += NT) out[i] = tmp_u32[i]; }
        } else if (tid == 0) toposort(g, Vn, out);
        __syncthreads();
    };

    // Cluster protocol (round 6: DP ATTEMPTS, not sequences). Member 0 publishes every DP it wants - a new sequence, or the same one again under another threshold
    // (PRUNE) - as {V, L, threshold} and a rising attempt number in csy[0]; the other members run one DP per attempt, whatever it is, add themselves to csy[1], and wait
    // for the next number or for the release (CL_ABORT, written by member 0 when it leaves the loop for whatever reason). They do not count sequences.
    uint32_t att = 0;   // DP attempts of this edge so far (uniform; member 0 publishes att, the others wait for att + 1)
    for (uint32_t k = ED.seq_begin; mem > 0 || k < ED.seq_end; k += (mem == 0 ? 1u : 0u)) {
        uint32_t L, V;
        if (mem == 0 && sOk != 1) break;
        if (mem == 0) {
            const PoaSeq q = seqs[k];
            L = q.len;
            // ---- decode the gap sub-sequence (forward: read[spos+j]; reverse strand: complement of read[rlen-1-(spos+j)])
            {
                const uint8_t* rp = packed + read_off[q.rid];
                const uint32_t rlen = read_len[q.rid];
                for (uint32_t j = tid; j < L; j += NT) {
                    uint32_t p = q.strand == 0 ? q.spos + j : rlen - 1 - (q.spos + j);
                    uint8_t b = (rp[p >> 2] >> ((p & 3) * 2)) & 3;
                    seq[j] = q.strand == 0 ? b : (uint8_t)(3 - b);
                }
            }
            __syncthreads();
            PHASE(0);
            V = sV;
            SUBT0();
        } else {
            // ---- other member of a cluster: only the DP, over its own columns; everything else happens in member 0
            if (tid == 0) {
                uint32_t v = 0;
                for (uint32_t spin = 0;; spin++) {
                    v = ld_dev(csy + 0);
                    if (v == CL_ABORT || v >= att + 1) break;
                    if (spin > poll_limit) { st_dev(csy + 4, 1u); v = CL_ABORT; break; }
                    __builtin_amdgcn_s_sleep(32);
                }
                sCtl = v;
            }
            __syncthreads();
            if (sCtl == CL_ABORT) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // graph rows, sequence and counts written by member 0
            V = ld_dev(csy + 2); L = ld_dev(csy + 3);
        }
        // =================================================== DP over (rank, column): every member, its own columns
        if (mem == 0) SUBT(6);   // publish
        // PRUNE: the threshold of this alignment = what the previous one of the edge scored per base, on this length, x prune_pct / 100 (scores per
        // base rise as the graph turns into a consensus, so this errs low). Too high an estimate costs a second attempt, never a wrong result: the
        // best sink of a pruned matrix is a real path's score, and an attempt that stays below its threshold is repeated with exactly that score.
        int thrT = PRUNE_OFF;
        if constexpr (PRUNE) {
            if (mem == 0 && (prune_pct & 0xffffu) != 0u && sPrevLen != 0u && V < (1u << 20)) {   // (2^20 rows: "nothing" keys lose at most a vertical move per row and must not wrap)
                const float f = (float)(prune_pct & 0xffffu) * 0.01f;
                float e = (float)sPrevScore * (float)L / (float)sPrevLen;
                e = e >= 0.f ? e * f : e * (2.f - f);
                thrT = (int)fmaxf((float)PRUNE_OFF, floorf(e));
            }
        }
    redo_dp:
        att++;
        if (GM > 1) {
            if (mem == 0) {   // publish this attempt to the other members: graph rows (CSR), decoded sequence, V, L, the threshold
                __threadfence();
                __syncthreads();
                if (tid == 0) { st_dev(csy + 2, V); st_dev(csy + 3, L); st_dev(csy + 5, (uint32_t)thrT); __threadfence(); st_dev(csy + 0, att); }
            } else if constexpr (PRUNE) thrT = (int)ld_dev(csy + 5);
        }
        if (V > 0) {
            uint32_t ns = 0xffffffffu;
#ifdef HX_DP_PROF3
            const long long td0 = clock64();
            if (tid == 0) ph[6] = 0;
#endif
            if (NP > 1) {
                for (uint32_t pass = 0; pass < NP && (uint64_t)pass * DL * CM <= L; pass++) {   // (a window beyond the sequence has nothing to do)
                    cl.mem = pass; cl.members = NP;
                    HX_DP_DISPATCH(L, V, ns, thrT);
                    __threadfence();          // the carries of this window's last column, its far rows: visible to the next window's waves
                    __syncthreads();
                }
                cl.mem = 0; cl.members = 1;
            } else
            HX_DP_DISPATCH(L, V, ns, thrT);
#ifdef HX_DP_PROF3
            if (tid == 0 && phase) atomicAdd(&phase[(uint64_t)eidx * POA_PHASE_WORDS + 6 + min(mem, 5u)], ((ph[6] >> 10) << 32) | ((unsigned long long)(clock64() - td0) >> 10));
#endif
            if (ns != 0xffffffffu) sNsink = ns;   // written by the lane that owns column L
            cl.tag0 += V;
        }
        if (mem > 0) {
            __syncthreads();
            if (V > 0 && L / (DL * (uint32_t)CM) == mem) {   // this member owns the last column: hand the sink rows to member 0
                const uint32_t nsk = min(sNsink, SINK_LDS);
                if (tid == 0) sinkbuf[0] = (int32_t)sNsink;
                for (uint32_t q = tid; q < nsk; q += NT) { sinkbuf[1 + q] = (int32_t)sink_row[q]; sinkbuf[1 + SINK_CAP + q] = sink_score[q]; }
            }
            __threadfence();                                       // direction bytes, sink rows: visible before "done"
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(csy + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        SUBT(7);   // own columns
        if (V > 0) {
            {
                if (GM > 1) {   // wait for the other members' columns (direction bytes, sinks)
                    __syncthreads();
                    if (tid == 0) {
                        const uint32_t need = (GM - 1) * att;
                        for (uint32_t spin = 0;; spin++) {
                            if (ld_dev(csy + 1) >= need) break;
                            if (spin > poll_limit) { st_dev(csy + 4, 1u); break; }
                            __builtin_amdgcn_s_sleep(32);
                        }
                    }
                    __syncthreads();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    if (L / (DL * (uint32_t)CM) != 0) {   // the last column lives in another member: fetch its sink rows
                        const uint32_t nsk_all = (uint32_t)sinkbuf[0], nsk = min(nsk_all, SINK_LDS);
                        for (uint32_t q = tid; q < nsk; q += NT) { sink_row[q] = (uint32_t)sinkbuf[1 + q]; sink_score[q] = sinkbuf[1 + SINK_CAP + q]; }
                        if (tid == 0) sNsink = nsk_all;
                    }
                }
            }
            __syncthreads();
            if (tid == 0 && ld_dev(csy + 4) && sOk == 1) sOk = ld_dev(csy + 4) == 2u ? 2 : 8;   // a wave gave up waiting for a carry (never seen within one workgroup; between workgroups
                                                                    // when the members of an edge are not resident together): the host redoes the edge unshared
            __syncthreads();
            if (sOk != 1) break;                                    // (the matrix of this sequence is not to be walked)
            SUBT(8);   // wait for the other members, sinks
            // ---- end node of the global alignment: the best-scoring sink; ties go to the smallest rank in the REFERENCE's order
            if (tid == 0) {
                if (sNsink > SINK_LDS) sOk = 6;   // more sink rows than the launch keeps: the host redoes the edge in a launch with the full list
                int best = INT32_MIN + 1024; uint32_t ncand = 0, first = 0;
                const uint32_t nsk = min(sNsink, SINK_LDS);
                for (uint32_t q = 0; q < nsk; q++) if (sink_score[q] > best) best = sink_score[q];
                for (uint32_t q = 0; q < nsk; q++) if (sink_score[q] == best) { if (!ncand) first = q; sink_row[ncand++] = sink_row[q]; }   // compact candidates to the front
                (void)first;
                if constexpr (PRUNE) {   // did the alignment reach its threshold? (else: again, with the score it did reach - a real path's - or, no sink computed at all, unpruned)
                    sRetry = thrT > PRUNE_OFF && sOk == 1 && best < thrT;
                    if (sRetry) { sNewT = nsk ? max(best, PRUNE_OFF) : PRUNE_OFF; ph[14]++; }
                    else { sPrevScore = best; sPrevLen = L; ph[15] += thrT > PRUNE_OFF; }
                }
                sNcand = ncand; sBestI = ncand ? (int)sink_row[0] : -1; sBestKey = 0xffffffffu;
                lds_u[8] = 0; lds_u[9] = 0; lds_u[10] = 0;   // (traceback helper: nowhere yet, not done)
                // Ties (a quarter of all alignments have one): the winner is the candidate the REFERENCE's topological order lists first. That order is a DFS
                // over in-edges and aligned nodes with the roots taken in node-id order (toposort above) - the whole graph, serial, 1 500 cycles per node
                // when it has to run (36 M cycles per tie on a 20 000-node graph: 22 ties were HALF the chain of the longest edge of a 3 316-edge call).
                // It need not run: let U be the candidates' columns (aligned groups) and everything downstream of them, closed under out-edges and
                // aligned mates. No node outside U has a node of U among its ancestors, so neither a root outside U nor the predecessors a node of U
                // has outside U ever visit a node of U: the order in which the DFS visits, finishes and emits the nodes of U is that of the same DFS run
                // on U alone (roots = U in id order, predecessors outside U taken as finished). Candidates are sinks at the end of the graph: U is the
                // few letters seen at the end of the gap (at most 8 nodes in 99.7 %, never above 32 on the three committed SPOA input sets: 1 267 ties,
                // every one decided like the full sort - the oracle's ORC_POA_TIES statistic runs the same simulation on the CPU).
                // One lane; ids, marks and the stack sit in the LDS the ring has left. More than 32 nodes / 8 candidates: the full sort below.
                if (ncand > 1 && ncand <= 8 && lds_bytes >= 640u && !(PRUNE && sRetry)) {
                    constexpr uint32_t UCAP = 32, SCAP = 256;
                    uint32_t* U = reinterpret_cast<uint32_t*>(ring);
                    uint8_t* mk = reinterpret_cast<uint8_t*>(U + UCAP);
                    uint8_t* ck = mk + UCAP;
                    uint8_t* stk = ck + UCAP;
                    uint32_t nu = 0, cn[8];
                    bool ok = true;
                    auto find = [&](uint32_t x) -> uint32_t { for (uint32_t q = 0; q < nu; q++) if (U[q] == x) return q; return NONE; };
                    auto addcol = [&](uint32_t x) {   // a column enters U whole (its members list each other: one member in U <=> all of them)
                        if (find(x) != NONE) return;
                        const uint32_t na = g.n_aligned[x];
                        if (nu + 1 + na > UCAP) { ok = false; return; }
                        U[nu++] = x;
                        for (uint32_t k = 0; k < na; k++) U[nu++] = g.aligned[3 * x + k];
                    };
                    for (uint32_t c = 0; c < ncand; c++) { cn[c] = g.rank2node[sink_row[c] - 1]; if (ok) addcol(cn[c]); }
                    for (uint32_t q = 0; q < nu && ok; q++)
                        for (uint32_t e = g.out_head[U[q]]; e != NONE && ok; e = g.e_next_out[e]) addcol(g.e_to[e]);
                    if (ok) {
                        for (uint32_t q = 1; q < nu; q++) {   // roots are taken in id order
                            const uint32_t x = U[q]; uint32_t r = q;
                            while (r > 0 && U[r - 1] > x) { U[r] = U[r - 1]; r--; }
                            U[r] = x;
                        }
                        for (uint32_t q = 0; q < nu; q++) { mk[q] = 0; ck[q] = 1; }
                        auto cand_of = [&](uint32_t x) -> uint32_t { for (uint32_t c = 0; c < ncand; c++) if (cn[c] == x) return c; return NONE; };
                        uint32_t win = NONE;
                        for (uint32_t root = 0; root < nu && win == NONE && ok; root++) {
                            if (mk[root]) continue;
                            uint32_t sp = 0;
                            stk[sp++] = (uint8_t)root;
                            while (sp && win == NONE && ok) {
                                const uint32_t q = stk[sp - 1], n = U[q];
                                bool valid = true;
                                if (mk[q] != 2) {
                                    for (uint32_t e = g.in_head[n]; e != NONE; e = g.e_next_in[e]) {
                                        const uint32_t f = find(g.e_from[e]);
                                        if (f != NONE && mk[f] != 2) { if (sp < SCAP) stk[sp++] = (uint8_t)f; else ok = false; valid = false; }
                                    }
                                    const uint32_t na = g.n_aligned[n];
                                    if (ck[q])
                                        for (uint32_t k = 0; k < na; k++) {
                                            const uint32_t a = find(g.aligned[3 * n + k]);
                                            if (mk[a] != 2) { if (sp < SCAP) stk[sp++] = (uint8_t)a; else ok = false; ck[a] = 0; valid = false; }
                                        }
                                    if (valid) {
                                        mk[q] = 2;
                                        if (ck[q]) {   // emitted: the node, then its aligned list in list order
                                            win = cand_of(n);
                                            for (uint32_t k = 0; k < na && win == NONE; k++) win = cand_of(g.aligned[3 * n + k]);
                                        }
                                    } else mk[q] = 1;
                                }
                                if (valid) sp--;
                            }
                        }
                        if (ok && win != NONE) { sBestI = (int)sink_row[win]; sNcand = 1; }
                    }
                }
            }
            __syncthreads();
            if constexpr (PRUNE) {
                if (sRetry) {   // (uniform: every thread reads the same word after the barrier)
                    thrT = sNewT;
                    __syncthreads();
                    if (tid == 0) sRetry = 0;
                    goto redo_dp;
                }
            }
            if (sNcand > 1 && sOk == 1) {
                if (tid == 0) ph[10] += 1;
                exact_order(V, tmp_u32);
                const uint32_t nc = sNcand;
                for (uint32_t r = tid; r < V; r += NT) {
                    const uint32_t n = tmp_u32[r];
                    for (uint32_t q = 0; q < nc; q++)
                        if (g.rank2node[sink_row[q] - 1] == n) atomicMin(&sBestKey, (r << 10) | q);   // nc <= 1024 candidates
                }
                __syncthreads();
                if (tid == 0) { sBestI = (int)sink_row[sBestKey & 1023u]; atomicAdd(&ph[4], 0ull); }
            }
            __syncthreads();
            SUBT(9);   // end node (ties: reference order)
            PHASE(1);
            // =================================================== traceback, stored reversed
            if (DIR) {
                // Direction bytes: the first wavefront walks the path together. A tile of 64 rows x 32 columns of direction nibbles (four registers)
                // and the records of those 32 rows are fetched with one round of loads; the walk inside the tile runs on v_readlane, i.e. one
                // memory round trip per ~12 steps instead of 3-4 dependent ones per step. Ranks are turned into node ids by all lanes afterwards.
                if (tid < 64) {
                    const uint32_t ln = tid;
                    if (ln == 0) sCells += (unsigned long long)V * L;
                    uint32_t i = (uint32_t)__builtin_amdgcn_readfirstlane(sBestI), j = L, na = 0;
                    // The walk records the cell it stands on, (rank, column), before every move: entry n in lane n % 64 of two registers
                    // (v_writelane), 64 entries leave with one coalesced store each. What the alignment wants - "node or nothing, position or
                    // nothing" - is a comparison of neighbouring entries and is made by all lanes afterwards.
                    int pn = 0, pp = 0;
                    uint32_t nwalk = 0, iend = 0, jend = 0;   // entries of the walk proper, and where it stopped
                    bool tail = false;
                    while (!(i == 0 && j == 0)) {
                        // (every step leaves a row or a column behind: a walk of more entries than rows + columns is walking a matrix that is not one - a bug somewhere
                        // else. It ends here, the edge comes back with an internal error and the call fails loudly, instead of a kernel that never ends; the entry
                        // arrays have 64 entries of slack for the tile that runs over: need_of)
                        if (__builtin_expect(na > V + L + 2u, 0)) { if (ln == 0) sOk = 2; break; }
                        if (i == 0) {   // only horizontal moves are left in the virtual row: written in their final form
                            if (na & 63u) { const uint32_t base = na & ~63u; if (ln < na - base) { g.aln_node[base + ln] = pn; g.aln_pos[base + ln] = pp; } }
                            tail = true;
                            nwalk = na; iend = 0; jend = j;
                            for (uint32_t q = ln; q < j; q += 64) { g.aln_node[na + q] = -1; g.aln_pos[na + q] = (int32_t)(j - 1 - q); }
                            na += j; j = 0;
                            break;
                        }
                        // tile: 64 rows (ranks ti .. ti-63) x 32 columns (ct-31 .. ct, ct = tj made odd: whole bytes of two nibbles); lane = row: the 16 bytes =
                        // 32 columns of one row in four registers. (32 rows x 16 columns, lane = (row, half), until round 6. A tile is a round trip to memory - 2 000
                        // cycles with the helper wave ahead of the walk, 30 % of a walk's time - and graphs have ~2 ranks per column: the walk left a tile of 16
                        // columns through its columns after 16.5 steps on average, whatever its rows; 64 x 32 is the shape it leaves through both at once.)
                        const uint32_t ti = i, ct = j | 1u;
                        if (NT > 64 && ln == 0) { st_wg(&lds_u[8], ti); st_wg(&lds_u[9], ct); }   // (where the walk is: the helper wave fetches ahead of it)
                        const int32_t bs = ((int32_t)ct - 31) >> 1;   // first byte of the tile in its rows (ct < 31: negative - bytes before the row, never looked at)
                        uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, mt = 0, pv0 = 0, pv1 = 0, qo = 0, qw = 0;
                        if (ti > ln) {
                            // four unaligned dword loads per lane (bytes bs .. bs + 15 of the row; columns below 0 read the end of the previous row - row 0 exists)
                            const uint8_t* rowp = Dm + (uint64_t)(ti - ln) * (W >> 1) + bs;
                            __builtin_memcpy(&w0, rowp, 4); __builtin_memcpy(&w1, rowp + 4, 4); __builtin_memcpy(&w2, rowp + 8, 4); __builtin_memcpy(&w3, rowp + 12, 4);
                            // the records of the tile's rows: where a move into the first / second predecessor leads
                            const uint32_t rr = ti - 1 - ln;
                            mt = g.row_meta[rr]; qo = g.row_pred_off[rr]; if (mt & 32u) qw = g.wslot[rr];
                            pv0 = (mt >> META_NP) == 0 ? 0u : (g.row_pred0[rr] & 0x0fffffffu) + 1;   // (a source node continues in the virtual row 0)
                            pv1 = (g.row_pred1[rr] & 0x0fffffffu) + 1;
                        }
                        uint32_t ilo = ti > 63u ? ti - 63u : 1u;                       // the walk goes on while i >= ilo (rows of the tile, never row 0) ...
                        asm volatile("" : "+s"(ilo));                                // (... as ONE subtraction per step: the compiler otherwise takes the constant apart again)
                        const int32_t jb0 = 2 * bs;                                  // ... and j >= first column of the tile
                        // (bit dr: the tile's row dr has more than 4 predecessors - a scalar bit test per step instead of a readlane of the row's record)
                        const unsigned long long wmask = __ballot((mt & 32u) != 0);
                        if (mt & 32u) { w0 = 0x88888888u; w1 = 0x88888888u; w2 = 0x88888888u; w3 = 0x88888888u; }   // a row with more than 4 predecessors: every cell reads as code 8, one of the rare moves - its real move is in the wide-row pool
                        for (;;) {
                            const uint32_t dr = ti - i, jb = (uint32_t)((int32_t)j - jb0);   // row and column inside the tile (0..63, 0..31)
                            const uint32_t wsa = (uint32_t)__builtin_amdgcn_readlane((int)w0, (int)dr), wsb = (uint32_t)__builtin_amdgcn_readlane((int)w1, (int)dr);
                            const uint32_t wsc = (uint32_t)__builtin_amdgcn_readlane((int)w2, (int)dr), wsd = (uint32_t)__builtin_amdgcn_readlane((int)w3, (int)dr);
                            const uint32_t wsh = ((jb & 16u) ? ((jb & 8u) ? wsd : wsc) : ((jb & 8u) ? wsb : wsa)) >> (4 * (jb & 7u));   // the cell's code in bits 0-3
                            const uint32_t n4 = wsh & 15u;
                            // move code: type (3 diagonal / 2 vertical / 1 horizontal) * 4 + 3 - predecessor slot; a row with more than 4 predecessors
                            // keeps type * 16 + 15 - slot in the wide-row pool
                            // (a move into the third or a later predecessor - codes 8, 9, 12, 13 of a 4-bit row - has to fetch that predecessor's rank)
                            const uint32_t pva = (uint32_t)__builtin_amdgcn_readlane((int)pv0, (int)dr), pvb = (uint32_t)__builtin_amdgcn_readlane((int)pv1, (int)dr);
                            // What the step waits for is its chain of DEPENDENT instructions from one (row, column) to the next, ~20 cycles each for a lone wave
                            // (docs/poa_kernel_notes.md); the compiler spends 7 on "which row next" and 5 on "which column next". Spelled out, 4 and 3:
                            //   row:    not slot 0 (a low bit of the code clear)? second predecessor : first; bit 3 (types 2 and 3: the move leaves the row)? that : this row
                            //   column: type 2 (bits 3-2 = 10: vertical)? this column : one to the left
                            uint32_t ni, nj;
                            asm("s_andn2_b32 %0, 3, %2\n\ts_cselect_b32 %0, %4, %3\n\ts_bitcmp1_b32 %2, 3\n\ts_cselect_b32 %0, %0, %5\n\t"
                                "s_and_b32 %1, %2, 12\n\ts_cmp_lg_u32 %1, 8\n\ts_subb_u32 %1, %6, 0"
                                : "=&s"(ni), "=&s"(nj) : "s"(wsh), "s"(pva), "s"(pvb), "s"(i), "s"(j) : "scc");
                            if (__builtin_expect((n4 & 10u) == 8u, 0)) {   // ONE test for the rare moves: codes 8, 9, 12, 13 = a third or later predecessor, or (code 8 by the line above the loop) a wide row
                                uint32_t type = n4 >> 2, slot = 3u - (n4 & 3u), later = 1u, pv = slot == 0 ? pva : pvb;
                                if ((uint32_t)(wmask >> dr) & 1u) {
                                    const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)Dw[(uint64_t)__builtin_amdgcn_readlane((int)qw, (int)dr) * W + j]);
                                    type = d >> 4; slot = 15u - (d & 15u); later = slot >= 2 && type > 1u;
                                    pv = slot == 0 ? pva : pvb;
                                }
                                if (later != 0)
                                    pv = ((uint32_t)__builtin_amdgcn_readfirstlane((int)g.pred_rank[(uint32_t)__builtin_amdgcn_readlane((int)qo, (int)dr) + slot]) & 0x0fffffffu) + 1;
                                ni = type > 1u ? pv : i;          // (a horizontal move is type 1 from the int32 rows, type 0 from the packed ones)
                                nj = j - (uint32_t)(type != 2u);  // (horizontal and diagonal moves go a column to the left, a vertical one does not)
                            }
                            {
                                const int el = (int)(na & 63u);
                                // (M0 is saved and restored around the two v_writelane: the compiler reserves it - an earlier version that listed it as a
                                //  clobber broke the one kernel instance that spills heavily, k_poa<1024, 32, true>: memory faults / wrong consensus for
                                //  gaps above 16 383 columns in ONE workgroup, found by the round-3 fuzz)
                                int m0_keep;
                                asm volatile("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %4, m0\n\tv_writelane_b32 %1, %5, m0\n\ts_mov_b32 m0, %2"
                                             : "+v"(pn), "+v"(pp), "=&s"(m0_keep) : "s"(el), "s"(i), "s"(j));
                            }
                            if (__builtin_expect((na & 63u) == 63u, 0)) {   // 64 entries: out they go (a store per step cost more than the step itself)
                                asm volatile("" ::: "memory");
                                g.aln_node[na - 63 + ln] = pn; g.aln_pos[na - 63 + ln] = pp;
                            }
                            na++;
                            i = ni; j = nj;
                            if ((int32_t)((i - ilo) | (uint32_t)((int32_t)j - jb0)) < 0) break;   // left the tile's rows or columns
                        }
                    }
                    if (!tail) {
                        if (na & 63u) { const uint32_t base = na & ~63u; if (ln < na - base) { g.aln_node[base + ln] = pn; g.aln_pos[base + ln] = pp; } }
                        nwalk = na; iend = 0; jend = 0;
                    }
                    if (ln == 0) { sNaln = na; lds_u[0] = nwalk; lds_u[1] = iend; lds_u[2] = jend; st_wg(&lds_u[10], 1u); }
                } else if (tid < 128) {
                    // Helper wavefront of the walk (any workgroup with a second wave; it sits on another SIMD): touches what the walk reaches in the
                    // next one to three tiles - the nibble rows around the path's expected column (graphs have ~2 ranks per column), the rows'
                    // records, the later predecessor entries of rows with more than two and the move bytes of wide rows - so that the walk's
                    // tile fetches and its rare dependent loads find their lines in the CU's vector cache instead of the L2. Nothing it loads is used.
                    // (measured on the longest 12 Mb edge, cycles of the traceback phase: no helper 92 M, 64 rows ahead 76 M, 128 rows 70-72 M, 192 rows 74 M,
                    //  256 rows 78 M; two or three helper waves 83-87 M; polling four times as often 80 M)
                    const uint32_t hl = tid - 64u;
                    uint32_t last_i = 0xffffffffu, sink = 0;
                    for (uint32_t spin = 0; spin < (1u << 26); spin++) {
                        if (ld_wg(&lds_u[10])) break;
                        const uint32_t pi_ = ld_wg(&lds_u[8]), pc = ld_wg(&lds_u[9]);
                        if (pi_ == last_i) { __builtin_amdgcn_s_sleep(8); continue; }
                        last_i = pi_;
                        for (uint32_t hk = 0; hk < 2; hk++) {                        // rows pi_ - 64 .. pi_ - 191, 64 at a time
                        const uint32_t d = 64u + hl + 64u * hk;
                        if (pi_ > d) {
                            const uint32_t r = pi_ - d, rr = r - 1;
                            const uint8_t* rowp = Dm + (uint64_t)r * (W >> 1);
                            const uint32_t c_lo = pc > d ? pc - d : 0u, c_hi = pc > d / 3u ? pc - d / 3u : 0u;
                            uint32_t a0, a1;
                            __builtin_memcpy(&a0, rowp + ((c_lo >> 1) & ~3u), 4); __builtin_memcpy(&a1, rowp + ((c_hi >> 1) & ~3u), 4);
                            const uint32_t mt = g.row_meta[rr], qo = g.row_pred_off[rr];
                            sink ^= a0 ^ a1 ^ g.row_pred0[rr] ^ g.row_pred1[rr];
                            if ((mt >> META_NP) > 2u) sink ^= g.pred_rank[qo + 2] ^ g.pred_rank[qo + (mt >> META_NP) - 1];
                            if (mt & 32u) { const uint8_t* wp = Dw + (uint64_t)g.wslot[rr] * W; sink ^= wp[c_lo] ^ wp[(c_lo + c_hi) >> 1] ^ wp[c_hi]; }
                        }
                        }
                    }
                    asm volatile("" :: "v"(sink));
                }
                __syncthreads();
                // (the walk's entries - the cell it stood on before every move - become alignment entries in graph_update: "the node of its row unless the move
                // stayed in the row, its column unless the move stayed in the column" is read off neighbouring entries there, on the way to the bases)
                if (tid == 0) { sWalkN = lds_u[0]; sWalkI = lds_u[1]; sWalkJ = lds_u[2]; }
                __syncthreads();
            } else if (tid == 0) {
                sCells += (unsigned long long)V * L;
                uint32_t i = (uint32_t)sBestI, j = L, na = 0;
                while (!DIR && !(i == 0 && j == 0)) {
                    const int hij = H[(uint64_t)i * WH + j];
                    uint32_t pi_ = i, pj_ = j;
                    bool found = false;
                    uint32_t po = 0, pe = 0;
                    if (i != 0) { po = g.row_pred_off[i - 1]; pe = g.row_pred_off[i]; }
                    if (i != 0 && j != 0) {
                        const int mc = seq[j - 1] == (uint8_t)(g.row_meta[i - 1] & 3u) ? match : mismatch;
                        if (po == pe) { if (hij == H[j - 1] + mc) { pi_ = 0; pj_ = j - 1; found = true; } }
                        else for (uint32_t p = po; p < pe && !found; p++) {
                            uint32_t pr = (g.pred_rank[p] & 0x0fffffffu) + 1;
                            if (hij == H[(uint64_t)pr * WH + j - 1] + mc) { pi_ = pr; pj_ = j - 1; found = true; }
                        }
                    }
                    if (!found && i != 0) {
                        if (po == pe) { if (hij == H[j] + gap) { pi_ = 0; pj_ = j; found = true; } }
                        else for (uint32_t p = po; p < pe && !found; p++) {
                            uint32_t pr = (g.pred_rank[p] & 0x0fffffffu) + 1;
                            if (hij == H[(uint64_t)pr * WH + j] + gap) { pi_ = pr; pj_ = j; found = true; }
                        }
                    }
                    if (!found) { pi_ = i; pj_ = j - 1; }
                    g.aln_node[na] = i == pi_ ? -1 : (int32_t)g.rank2node[i - 1];
                    g.aln_pos[na] = j == pj_ ? -1 : (int32_t)(j - 1);
                    na++;
                    i = pi_; j = pj_;
                }
                sNaln = na; sWalkN = 0;
            }
        } else {
            if (tid == 0) { sNaln = 0; sWalkN = 0; }
            if (GM > 1) {   // nothing to align against yet: the other members only count the sequence (and must have read V = 0 before it changes)
                __syncthreads();
                if (tid == 0) {
                    const uint32_t need = (GM - 1) * att;
                    for (uint32_t spin = 0;; spin++) {
                        if (ld_dev(csy + 1) >= need) break;
                        if (spin > poll_limit) { st_dev(csy + 4, 1u); break; }
                        __builtin_amdgcn_s_sleep(32);
                    }
                }
                __syncthreads();
            }
        }
        __syncthreads();
        if (sOk != 1) break;                                        // (a walk that did not end: see there)
        PHASE(2);
        // =================================================== graph update + order update (all lanes): graph_update / order_update above
        const uint32_t V_old = sV;
        graph_update(g, seq, L, sNaln, sWalkN, sWalkI, sWalkJ, lds_u, &sV, &sE, &sNcand, &sOk);
        PHASE(3);
        __syncthreads();
        if (sOk != 1) break;
        order_update(g, V_old, sV, L, lds_u);
        PHASE(4);
        __syncthreads();
        // =================================================== rank-order CSR for the next DP (all lanes): csr_rebuild above
        csr_rebuild<MAXNT, DIR>(g, sV, R, max_indeg, ED.hrows, ED.wrows, lds_u, &sOk, ph, phase != nullptr, eidx, k == ED.seq_begin, k + 1 == ED.seq_end);
        __syncthreads();
        PHASE(5);
    }
    if (mem > 0) return;
    if (GM > 1 && tid == 0) st_dev(csy + 0, CL_ABORT);   // release the other members (they wait for the next attempt: done or not, there is none)
    long long t_cns = 0;
    if (tid == 0) t_cns = clock64();
    if (sOk == 1 && sV) {
        // heaviest bundle: first on the maintained order (exact whenever the heaviest node is unique and a sink); else on the reference's
        // topological order of the finished graph
        if (tid < 64) { const uint32_t cl_ = consensus_fast_wave(g, sV, cns + ED.cns_off); if (tid == 0) sCtl = cl_; }
        __syncthreads();
        if (sCtl == NONE) {
            if (tid == 0) ph[19] = 1;
            exact_order(sV, g.rank2node);
            for (uint32_t r = tid; r < sV; r += NT) g.node2rank[g.rank2node[r]] = r;
            __syncthreads();
            // the rank-ordered in-edge rows (predecessor ranks, weights, letter, sink flag) of THIS order: the CSR rebuild again (its ring slots and far / wide rows
            // mean nothing here and must not fail the edge: no limits)
            csr_rebuild<MAXNT, DIR>(g, sV, R, 0xffffffffu, 0xffffffffu, 0xffffffffu, lds_u, &sOk, ph, false, eidx, false, false);
            __threadfence_block();
            __syncthreads();
            if (tid < 64) { const uint32_t cl_ = consensus_wave(g, sV, cns + ED.cns_off); if (tid == 0) sCtl = cl_; }
            __syncthreads();
        }
    }
    if (tid == 0) {
        if (sOk == 2) { status[eidx] = HXE_SPOS_RANGE << 8; cns_len[eidx] = 0; }   // internal: kernel variant cannot hold this many columns per lane
        else if (!sOk) { status[eidx] = HXE_POA_OVERFLOW; cns_len[eidx] = 0; }
        else if (sOk == 4) { status[eidx] = HXE_POA_NODIR; cns_len[eidx] = 0; }
        else if (sOk == 5) { status[eidx] = HXE_POA_FARROWS; cns_len[eidx] = 0; }
        else if (sOk == 6) { status[eidx] = HXE_POA_SINKS; cns_len[eidx] = 0; }
        else if (sOk == 7) { status[eidx] = HXE_POA_WIDEROWS; cns_len[eidx] = 0; }
        else if (sOk == 8) { status[eidx] = HXE_POA_STALLED; cns_len[eidx] = 0; }
        else {
            status[eidx] = 0; cns_len[eidx] = !sV ? 0 : sCtl; atomicAdd(cells, sCells);
#if !defined(HX_DP_PROF) && !defined(HX_GU_PROF)
            if (phase) atomicAdd(&ph[11], (unsigned long long)sV << 32);   // statistics: nodes of the finished graph (high word)
#endif
        }
        ph[18] = (unsigned long long)(clock64() - t_cns);
        PHASE(3);
        ph[17] = wall_clock64() & ((1ull << 44) - 1);
#ifdef HX_DP_PROF3
        if (phase) for (int k = 0; k < 6; k++) phase[(uint64_t)eidx * POA_PHASE_WORDS + k] = ph[k];
#else
        if (phase) for (int k = 0; k < POA_PHASE_WORDS; k++) phase[(uint64_t)eidx * POA_PHASE_WORDS + k] = ph[k];
#endif
    }
}

// The launch. An edge that is shared by several workgroups gets one workgroup per member and a workspace slot of its own (`order` entry =
// edge | member << 24, the slot is the edge's PoaEdge::slot). Everything else runs PERSISTENT: the grid is a number of workspace slots, each
// workgroup owns slot blockIdx.x - sized for the largest edge of its BUCKET of the launch (buckets of workspace need: round 5) - and works
// through the bucket's list (costliest first) off an atomic counter, then through the smaller buckets'. The workspace of a call is
// (workgroups in flight) x (largest edge of their bucket), not the sum over all its edges.
// (two instances per shape: the plain one keeps the register allocation of a kernel that runs one edge - the loop of the persistent one costs
// 8-12 VGPRs, which takes the 4-column kernels from 4 to 3 waves per SIMD - and is what the few-edge regime launches)
template <int MAXNT, int CM, bool DIR, bool PERSIST, bool PRUNE>
__global__ void __launch_bounds__(MAXNT, (CM == 8 && DIR ? 4 : 1)) k_poa(const PoaEdge* __restrict__ edges, const uint32_t* __restrict__ order, uint32_t n_items,
                                            const PoaSlot* __restrict__ slots, uint32_t* __restrict__ counter /* null: one workgroup per entry of `order` */,
                                            const uint32_t* __restrict__ btab /* persistent: buckets, slot ends, item begins (PoaLaunch) */, const PoaSeq* __restrict__ seqs, const uint8_t* __restrict__ packed, const uint64_t* __restrict__ read_off,
                                            const uint32_t* __restrict__ read_len, PoaPools P, int32_t match, int32_t mismatch, int32_t gap,
                                            char* cns, uint32_t* cns_len, uint32_t* status, unsigned long long* cells, unsigned long long* phase,
                                            uint32_t poll_limit, uint32_t lds_bytes, uint32_t max_indeg, uint32_t dp_lanes, uint32_t prune_pct,
                                            uint32_t* __restrict__ started /* null, or a word in host memory: every workgroup adds itself when it begins (the host waits for the launches that must be resident first) */) {
    __shared__ uint32_t sNext;
    if (started && threadIdx.x == 0) __hip_atomic_fetch_add(started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // persistent launch: the bucket whose slot this workgroup owns (slots are laid out bucket by bucket, largest workspace need first)
    uint32_t kb = 0, nbk = 0;
    bool own_first = false;
    if (PERSIST) { nbk = btab[0] & 0xffffu; own_first = (btab[0] >> 16) & 1u; while (kb + 1 < nbk && blockIdx.x >= btab[1 + kb]) kb++; }
    for (uint32_t round = 0;; round++) {   // (one call site of the edge body for both kinds of launch)
        uint32_t eidx, mem = 0;
        PoaSlot SL;
        if (!PERSIST) {
            eidx = order[blockIdx.x] & 0x00ffffffu; mem = order[blockIdx.x] >> 24;   // edge, member of its cluster
            if (eidx == 0x00ffffffu) return;              // hole in the XCD-aligned cluster grid
            SL = slots[edges[eidx].slot];
        } else {
            // The launch's edges come in BUCKETS of workspace need (a power of two each), every bucket's list ordered by estimated chain time, longest
            // first, behind a counter of its own. A workgroup's slot is sized for the largest edge of ITS bucket, so it can take the edges of that
            // bucket and of every bucket of smaller need - whoever is free takes ...
            if (round) __syncthreads();                   // (the previous edge has left the LDS)
            if (threadIdx.x == 0) {
                // ... of the buckets it can serve, the one whose next edge has the longest estimated chain (est[]: beside the lists): the call ends when its
                // longest chains end, and a long chain of little need must not wait behind its bucket's thousands.
                // Round 6 (`own_first`, bit 16 of btab[0]): its OWN bucket first while that has edges left - they are the edges the fewest workgroups can take
                // (least flexible job first). By the longest chain alone, a workgroup of a large bucket took a longer edge of a smaller bucket, the smaller bucket's
                // workgroup then found its list empty and LEFT, and the large bucket's own edge waited for one of the few slots that hold it: one pass in five of the
                // 140 Mb call ran 37 edges of the 512-lane class on 31 of its 37 workgroups, the last one third in line - 615 ms instead of 450.
                const uint32_t* ib = btab + 1 + nbk;
                const uint32_t* est = ib + nbk + 1;
                uint32_t idx = 0xffffffffu;
                for (uint32_t tries = 0; tries < 64 && idx == 0xffffffffu; tries++) {
                    uint32_t best = 0xffffffffu, bv = 0;
                    for (uint32_t k = kb; k < nbk; k++) {
                        const uint32_t b0 = ib[k], b1 = ib[k + 1], cur = __hip_atomic_load(counter + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (cur >= b1 - b0) continue;
                        const uint32_t v = est[b0 + cur];
                        if (best == 0xffffffffu || v > bv) { best = k; bv = v; }
                        if (own_first && k == kb) break;
                    }
                    if (best == 0xffffffffu) break;       // every bucket this workgroup can serve is empty
                    const uint32_t t = ib[best] + atomicAdd(counter + best, 1u);
                    if (t < ib[best + 1]) idx = t;        // (else: others emptied the bucket meanwhile - look again)
                }
                sNext = idx;
            }
            __syncthreads();
            const uint32_t idx = sNext;
            if (idx == 0xffffffffu) return;
            eidx = order[idx] & 0x00ffffffu;
            SL = slots[blockIdx.x];
        }
        poa_edge<MAXNT, CM, DIR, PRUNE>(eidx, mem, SL, edges, seqs, packed, read_off, read_len, P, match, mismatch, gap, cns, cns_len, status, cells, phase, poll_limit, lds_bytes, max_indeg, dp_lanes, prune_pct);
        if (!PERSIST) return;
    }
}

}  // namespace

// The instances are compiled in four translation units, one per largest workgroup (kernels/poa_part{64,256,512,1024}.hip: `#define HX_POA_PART n` and
// this file) - a build of all 61 instances in one unit takes three and a half minutes, the four side by side one; a unit without HX_POA_PART holds them all.
#if !defined(HX_POA_PART) || HX_POA_PART == 64
#define HX_POA_HAS_64 1
#endif
#if !defined(HX_POA_PART) || HX_POA_PART == 256
#define HX_POA_HAS_256 1
#endif
#if !defined(HX_POA_PART) || HX_POA_PART == 512
#define HX_POA_HAS_512 1
#endif
#if !defined(HX_POA_PART) || HX_POA_PART == 1024
#define HX_POA_HAS_1024 1
#endif
#define HX_LAUNCH(MNT, CMV, DIRV, PERS, PRN) do { \
        (void)hipFuncSetAttribute((const void*)k_poa<MNT, CMV, DIRV, PERS, PRN>, hipFuncAttributeMaxDynamicSharedMemorySize, 142 * 1024); \
        if (q.occupancy) { (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(q.occupancy, (const void*)k_poa<MNT, CMV, DIRV, PERS, PRN>, q.block_threads, q.ring_bytes); break; } \
        k_poa<MNT, CMV, DIRV, PERS, PRN><<<q.n_blocks, q.block_threads, q.ring_bytes, s>>>(q.edges, q.order, q.n_items, q.slots, q.counter, q.btab, q.seqs, q.packed, q.read_off, q.read_len, q.pools, q.match, q.mismatch, q.gap, \
                                                                       q.cns, q.cns_len, q.status, q.cells, q.phase, q.poll_limit, q.ring_bytes, q.max_indeg, q.dp_lanes, q.prune_pct, q.started); } while (0)
// (persistent instances exist for the direction-byte flavour only: poa_persistent_ok; pruned ones for it with 4 or 8 columns per lane: poa_prune_ok)
#define HX_LAUNCH_CM(MNT, CMV) do { if (q.use_dir && q.counter) HX_LAUNCH(MNT, CMV, true, true, false); else if (q.use_dir) HX_LAUNCH(MNT, CMV, true, false, false); else HX_LAUNCH(MNT, CMV, false, false, false); } while (0)
#define HX_LAUNCH_PR(MNT, CMV) do { if (q.counter) HX_LAUNCH(MNT, CMV, true, true, true); else HX_LAUNCH(MNT, CMV, true, false, true); } while (0)
// the instances the host's launch classes use (poa_kernel_lanes): workgroups up to 64 / 256 / 512 / 1024 lanes x 4, 8, 16 or 32 columns per lane
#define HX_POA_RUN_PART(MNT, LAST_CM) \
    void poa_run_##MNT(const PoaLaunch& q, hipStream_t s, bool prune) { \
        const int cm = q.cm; \
        if constexpr (MNT == 256 || MNT == 1024) { if (prune && cm <= 2 && !q.counter) { HX_LAUNCH(MNT, 2, true, false, true); return; } } \
        if (prune) { if (cm <= 4) HX_LAUNCH_PR(MNT, 4); else HX_LAUNCH_PR(MNT, 8); return; } \
        if constexpr (MNT == 256 || MNT == 1024) { if (cm <= 2 && q.use_dir && !q.counter) { HX_LAUNCH(MNT, 2, true, false, false); return; } }   /* (2 columns per lane: the members of shared edges, poa_kernel_min_cm) */ \
        if (cm <= 4) HX_LAUNCH_CM(MNT, 4); else if (cm <= 8) HX_LAUNCH_CM(MNT, 8); else if (cm <= 16 || LAST_CM == 16) HX_LAUNCH_CM(MNT, 16); else HX_LAUNCH_CM(MNT, LAST_CM); \
    }
#ifdef HX_POA_HAS_64
HX_POA_RUN_PART(64, 32)
#endif
#ifdef HX_POA_HAS_256
HX_POA_RUN_PART(256, 32)
#endif
#ifdef HX_POA_HAS_512
HX_POA_RUN_PART(512, 16)
#endif
#ifdef HX_POA_HAS_1024
HX_POA_RUN_PART(1024, 32)   // (1024 x 32: one workgroup for a gap of 8192..32767 bases: register spills, rare)
#endif
#undef HX_POA_RUN_PART
#undef HX_LAUNCH_CM
#undef HX_LAUNCH_PR
#undef HX_LAUNCH

#if !defined(HX_POA_PART) || HX_POA_PART == 64   // the dispatcher lives with the first part
void poa_run_64(const PoaLaunch&, hipStream_t, bool);
void poa_run_256(const PoaLaunch&, hipStream_t, bool);
void poa_run_512(const PoaLaunch&, hipStream_t, bool);
void poa_run_1024(const PoaLaunch&, hipStream_t, bool);
void poa_run(const PoaLaunch& q, hipStream_t s) {
    if (!q.n_blocks || !q.n_items) return;
    if (q.counter && !q.use_dir) return;   // (the host never asks for it: poa_persistent_ok)
    const bool prune = poa_prune_ok(q.use_dir, q.cm) && q.prune_pct != 0u;
    const int mnt = poa_kernel_lanes(q.block_threads);
    if (mnt == 64) poa_run_64(q, s, prune);
    else if (mnt == 256) poa_run_256(q, s, prune);
    else if (mnt == 512) poa_run_512(q, s, prune);
    else poa_run_1024(q, s, prune);
}
#endif

}  // namespace hxk
