// poa_update.inl - part of kernels/poa.hip: what follows an alignment, on all lanes - the rank-ordered CSR rebuild for the next DP, the graph update (spoa's add_alignment from the
// walk's entries) and the incremental order update.
// =================================================== rank-order CSR for the next DP (all lanes)
// Round 5: lane = rank. The first version dealt the ranks out in contiguous chunks per thread (a thread's prefix sums were then its own running
// counts) and ran seven passes of dependent list walks over them: every load of a wave touched 64 different cache lines, and the rebuild took ~2 000
// cycles PER ROW of a one-wave workgroup under load - a fifth of all wave cycles of a 13 000-edge call (tools/dev_r05.sh edgedump). Now the ranks
// are taken NT at a time, lane t = rank base + t: the rank-indexed arrays are read and written coalesced, the node-indexed ones nearly so (node ids
// rise with the ranks), offsets and ring slots come from a scan per block of ranks with a running base, and what used to need a pass of its own is
// read where it already is: the first two in-edge sources and the aligned ids from the node's 16-byte record, "kept" (a successor that is not the
// next row) from the node's own out-list instead of atomics from its successors. Four passes, one dependent chain of 3-4 loads each.
// (What the registers of this code cost the rest of the kernel, measured on the way: with four ranks per lane and the views read through the reference - every
// pointer in two VECTOR registers, 21 of them - the 8-column instances, capped at 128 registers, spilled 50-60 bytes more, and a 13 000-edge call took 0.617 s
// instead of 0.557 s; as a real function call the kernel takes the callee's registers as its own (132 > 128: three waves per SIMD) and the row loop of the
// 4-column instances got 6 % slower (970 -> 1 030 cycles per row at 12 Mb) with the call ABI's scalar registers. Inlined, with scalar pointers and U = 2: both fine.)
template <int MAXNT, bool DIR>
__device__ __forceinline__ void csr_rebuild(const G& g_in, const uint32_t V2, const uint32_t R, const uint32_t max_indeg, const uint32_t hrows, const uint32_t wrows, uint32_t* lds_u,
                                                      uint32_t* sOk, unsigned long long* ph, const bool stats, const uint32_t eidx, const bool first_seq, const bool last_seq) {
    // (the views arrive through memory: as they are, every pointer would sit in two VECTOR registers - the compiler cannot know them uniform - and 21 of them
    // are used here; read through readfirstlane they are scalars)
    G g = g_in;
    g.rank2node = uptr(g.rank2node); g.node2rank = uptr(g.node2rank); g.nrec = uptr(g.nrec); g.nrec2 = uptr(g.nrec2); g.in_head = uptr(g.in_head); g.code = uptr(g.code);
    g.e_w = uptr(g.e_w); g.e_next_in = uptr(g.e_next_in); g.e_next_out = uptr(g.e_next_out); g.e_to = uptr(g.e_to); g.e_from = uptr(g.e_from); g.row_pred_off = uptr(g.row_pred_off);
    g.row_al = uptr(g.row_al); g.score = uptr(g.score); g.pred_rank = uptr(g.pred_rank); g.pred_w = uptr(g.pred_w); g.row_meta = uptr(g.row_meta); g.row_pred0 = uptr(g.row_pred0);
    g.row_pred1 = uptr(g.row_pred1); g.pred = uptr(g.pred); g.wslot = uptr(g.wslot);
        // Every access below is a round trip to a memory that 3 800 other waves are using (~1 us under the load of such a call), and a pass is as long as its
        // chain of DEPENDENT round trips times its iterations: so each lane takes U ranks per iteration (their loads are issued together), and a row's
        // in-edges and out-edges are read off the node's two records (the first two of each: link_edge) instead of walked - the lists only for the rare
        // node with more. Pass B: rank -> node -> records -> ranks of the neighbours / weights: three round trips for U x NT rows.
        constexpr uint32_t U = 2;
        const uint32_t tid = threadIdx.x, NT = blockDim.x;
#ifdef HX_CSR_PROF   // development: cycles of the rebuild's stages on lane 0 (printed for every 500th edge at its end)
        __shared__ unsigned long long cp[8];
        if (tid == 0 && first_seq) for (int q = 0; q < 8; q++) cp[q] = 0;
        long long ct = clock64();
#define CSR_T(q, reg) do { asm volatile("" :: "v"(reg)); if (tid == 0) { const long long _n = clock64(); cp[q] += (unsigned long long)(_n - ct); ct = _n; } } while (0)
#else
#define CSR_T(q, reg) do { } while (0)
#endif
        for (uint32_t base = 0; base < V2; base += NT * U) {
            uint32_t nn_[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) { const uint32_t r = base + u * NT + tid; nn_[u] = r < V2 ? g.rank2node[r] : NONE; }
#pragma unroll
            for (uint32_t u = 0; u < U; u++) { const uint32_t r = base + u * NT + tid; if (nn_[u] != NONE) g.node2rank[nn_[u]] = r; }
        }
        __syncthreads();
        CSR_T(0, V2);
        uint32_t off_base = 0, kept_base = 0;
        uint32_t st_multi = 0, st_ring = 0, st_far = 0, st_wide = 0, st_fifth = 0;
        for (uint32_t base = 0; base < V2; base += NT * U) {       // ---- pass B: everything a row knows about itself
            uint32_t n[U], cd[U], np[U], kept[U], alp[U], pf0[U], pf1[U], e0[U], ei[U], eo[U]; int32_t w0[U], w1[U];
            uint4 A[U], B[U];
            bool on[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) { const uint32_t r = base + u * NT + tid; on[u] = r < V2; n[u] = g.rank2node[on[u] ? r : 0u]; }
            CSR_T(1, n[U - 1]);
#pragma unroll
            for (uint32_t u = 0; u < U; u++) { A[u] = g.nrec[n[u]]; B[u] = g.nrec2[n[u]]; e0[u] = g.in_head[n[u]]; cd[u] = g.code[n[u]]; }
            CSR_T(2, cd[U - 1]);   // {f0, f1, aligned ids + 1 (3 x 21 bits) | bit 63: more in-edges}, {2nd out-edge, 2nd in-edge, t0 | bit 31: more out-edges, t1}
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t r = base + u * NT + tid;
                const uint32_t f0 = A[u].x, f1 = A[u].y;
                np[u] = !on[u] || f0 == NONE ? 0u : f1 == NONE ? 1u : 2u;
                pf0[u] = g.node2rank[np[u] >= 1 ? f0 : 0u]; pf1[u] = g.node2rank[np[u] >= 2 ? f1 : 0u];
                w0[u] = g.e_w[np[u] >= 1 ? e0[u] : 0u]; w1[u] = g.e_w[np[u] >= 2 ? B[u].y : 0u];
                const bool o0 = B[u].z != NONE, o1 = B[u].w != NONE;
                const uint32_t rt0 = g.node2rank[o0 ? B[u].z & 0x7fffffffu : 0u], rt1 = g.node2rank[o1 ? B[u].w : 0u];
                kept[u] = (uint32_t)(on[u] && ((o0 && rt0 - r >= 2u) || (o1 && rt1 - r >= 2u)));   // a successor that is not the next row reads this one from the ring / HBM
                const unsigned long long al = ((unsigned long long)A[u].z | ((unsigned long long)A[u].w << 32)) & 0x7fffffffffffffffULL;
                const uint32_t a0 = (uint32_t)al & 0x1fffffu, a1 = (uint32_t)(al >> 21) & 0x1fffffu, a2 = (uint32_t)(al >> 42) & 0x1fffffu;
                const uint32_t ra0 = g.node2rank[a0 ? a0 - 1 : 0u], ra1 = g.node2rank[a1 ? a1 - 1 : 0u], ra2 = g.node2rank[a2 ? a2 - 1 : 0u];
                // the column's other members in list order, as rank deltas (a column is contiguous in this order; the list has no holes)
                alp[u] = (a0 ? ((ra0 - r + 4u) & 7u) : 0u) | (a1 ? ((ra1 - r + 4u) & 7u) << 3 : 0u) | (a2 ? ((ra2 - r + 4u) & 7u) << 6 : 0u);
                // the few nodes with more than two in-edges / out-edges: their lists from the second entry on - the U chains of a lane, and the lanes of the
                // wave, step TOGETHER below (every wave has such nodes among its 64 x U, and one chain after the other was most of this pass)
                ei[u] = on[u] && (A[u].w & 0x80000000u) ? B[u].y : NONE;
                eo[u] = on[u] && B[u].z != NONE && (B[u].z & 0x80000000u) ? B[u].x : NONE;
            }
            CSR_T(3, alp[U - 1]);
            uint32_t e3[U];                                        // third in-edge of the nodes that have one
            {
                bool first = true, more = false;
#pragma unroll
                for (uint32_t u = 0; u < U; u++) { more |= ei[u] != NONE || eo[u] != NONE; e3[u] = NONE; }
                while (more) {
                    uint32_t ni[U], no_[U], to[U];
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) { ni[u] = g.e_next_in[ei[u] != NONE ? ei[u] : 0u]; no_[u] = g.e_next_out[eo[u] != NONE ? eo[u] : 0u]; }
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) to[u] = g.e_to[eo[u] != NONE && no_[u] != NONE ? no_[u] : 0u];
                    more = false;
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) {
                        const uint32_t r = base + u * NT + tid;
                        if (ei[u] != NONE) { ei[u] = ni[u]; if (ni[u] != NONE) np[u]++; if (first) e3[u] = ni[u]; }
                        if (eo[u] != NONE) { eo[u] = no_[u]; if (no_[u] != NONE) kept[u] |= (uint32_t)(g.node2rank[to[u]] - r >= 2u); }
                        more |= ei[u] != NONE || eo[u] != NONE;
                    }
                    first = false;
                }
            }
            uint32_t off_[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t r = base + u * NT + tid;
                uint32_t tot_np, tot_k;
                const uint32_t ex_np = block_excl_scan_add(np[u], lds_u, &tot_np);
                const uint32_t ex_k = block_excl_scan_add(kept[u], lds_u, &tot_k);
                off_[u] = off_base + ex_np;
                if (on[u]) {
                    const uint32_t off = off_[u], kx = kept_base + ex_k, sink = B[u].z == NONE;
                    g.row_pred_off[r] = off; g.row_al[r] = (uint16_t)alp[u];   // (the row's letter and sink flag: bits 0-1 and 2 of its record)
                    g.score[r] = (int32_t)kx;                      // kept rows before r
                    if (np[u] >= 1) { g.pred_rank[off] = pf0[u]; g.pred_w[off] = w0[u]; }
                    if (np[u] >= 2) { g.pred_rank[off + 1] = pf1[u]; g.pred_w[off + 1] = w1[u]; }
                    g.row_meta[r] = cd[u] | (sink << 2) | (kept[u] << 4) | (np[u] > 4u ? 32u : 0u) | ((kept[u] && R ? (kx & (R - 1)) : 15u) << META_SLOT) | (np[u] << META_NP);   // slot 15: no non-adjacent reader (or no ring at all) - the row is not written to the ring
                    if (DIR && np[u] > max_indeg) *sOk = 4;         // the direction bytes hold a 4-bit predecessor slot (max_indeg <= 16)
                    g.row_pred0[r] = pf0[u]; g.row_pred1[r] = pf1[u];
                    st_multi += np[u] >= 2; st_wide += np[u] > 4; st_fifth += np[u] > 4 ? np[u] - 4 : 0;
                }
                off_base += tot_np; kept_base += tot_k;
            }
            {   // the third and later in-edges (same stepping: entry k of every chain that has one)
                uint32_t mx = 0;
#pragma unroll
                for (uint32_t u = 0; u < U; u++) mx = max(mx, np[u]);
                for (uint32_t k = 2; k < mx; k++) {
                    uint32_t f[U], nx[U]; int32_t w[U];
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) { const uint32_t e = k < np[u] ? e3[u] : 0u; f[u] = g.e_from[e]; w[u] = g.e_w[e]; nx[u] = g.e_next_in[e]; }
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) if (k < np[u]) { g.pred_rank[off_[u] + k] = g.node2rank[f[u]]; g.pred_w[off_[u] + k] = w[u]; e3[u] = nx[u]; }
                }
            }
#ifdef HX_CSR_PROF
            CSR_T(4, off_base);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            CSR_T(7, off_base);   // (the iteration's stores acknowledged)
#endif
        }
        if (tid == 0) g.row_pred_off[V2] = off_base;
        const uint32_t ktot = kept_base;
        __syncthreads();
        CSR_T(4, off_base);
        // ---- pass C: where the DP will find each predecessor row: 1..R ring slot + 1, 13 the previous row (registers), 14 the virtual row 0, 15 HBM
        // (a kept row that has left the ring by then: it is marked as read back from HBM). Two round trips: the row's entries, their rows' kept counts.
        for (uint32_t base = 0; base < V2; base += NT * U) {
            uint32_t po[U], np[U], kr[U], p0[U], p1[U], k0[U], k1[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t r = base + u * NT + tid, rc = r < V2 ? r : 0u;
                po[u] = g.row_pred_off[rc]; np[u] = r < V2 ? g.row_meta[rc] >> META_NP : 0u; kr[u] = (uint32_t)g.score[rc];
                p0[u] = g.row_pred0[rc]; p1[u] = g.row_pred1[rc];   // (pass B left the first two predecessor ranks here)
            }
#pragma unroll
            for (uint32_t u = 0; u < U; u++) { k0[u] = (uint32_t)g.score[np[u] >= 1 ? p0[u] : 0u]; k1[u] = (uint32_t)g.score[np[u] >= 2 ? p1[u] : 0u]; }
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t r = base + u * NT + tid;
                if (r >= V2) continue;
                auto place = [&](const uint32_t pr, const uint32_t kp) -> uint32_t {
                    uint32_t loc;
                    if (r - pr == 1) loc = 13;
                    else {
                        const uint32_t live = kr[u] - kp;          // kept rows produced in [pr, r), pr included
                        if (live <= R) loc = 1 + (kp & (R - 1));
                        else { loc = 15; atomicOr(&g.row_meta[pr], 8u); }
                    }
                    st_ring += r - pr >= 2 && loc != 15; st_far += loc == 15;
                    return pr | (loc << 28);
                };
                if (np[u] >= 1) { const uint32_t ent = place(p0[u], k0[u]); g.pred_rank[po[u]] = ent; g.row_pred0[r] = ent; }
                else g.row_pred0[r] = 14u << 28;                   // a source node: the virtual row 0
                if (np[u] >= 2) { const uint32_t ent = place(p1[u], k1[u]); g.pred_rank[po[u] + 1] = ent; g.row_pred1[r] = ent; }
            }
            {   // the third and later entries, entry q of every row that has one at a time
                uint32_t mx = 0;
#pragma unroll
                for (uint32_t u = 0; u < U; u++) mx = max(mx, np[u]);
                for (uint32_t q = 2; q < mx; q++) {
                    uint32_t pr[U], kp[U];
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) pr[u] = g.pred_rank[q < np[u] ? po[u] + q : 0u];
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) kp[u] = (uint32_t)g.score[q < np[u] ? pr[u] & 0x0fffffffu : 0u];
#pragma unroll
                    for (uint32_t u = 0; u < U; u++) {
                        const uint32_t r = base + u * NT + tid;
                        if (q >= np[u]) continue;
                        uint32_t loc;
                        if (r - pr[u] == 1) loc = 13;
                        else {
                            const uint32_t live = kr[u] - kp[u];
                            if (live <= R) loc = 1 + (kp[u] & (R - 1));
                            else { loc = 15; atomicOr(&g.row_meta[pr[u]], 8u); }
                        }
                        st_ring += r - pr[u] >= 2 && loc != 15; st_far += loc == 15;
                        g.pred_rank[po[u] + q] = pr[u] | (loc << 28);
                    }
                }
            }
        }
        if (DIR) {   // ---- pass D: rows that a far successor reads back from HBM get consecutive rows of H (the score matrix is not kept with direction bytes);
                     // rows with more than 4 predecessors a row of the wide-row pool (a direction byte per cell)
            __syncthreads();
            CSR_T(5, st_far);
            uint32_t* farslot = reinterpret_cast<uint32_t*>(g.pred);
            uint32_t far_base = 0, wide_base = 0;
            for (uint32_t base = 0; base < V2; base += NT * U) {
                uint32_t mt[U];
#pragma unroll
                for (uint32_t u = 0; u < U; u++) { const uint32_t r = base + u * NT + tid; mt[u] = r < V2 ? g.row_meta[r] : 0u; }
#pragma unroll
                for (uint32_t u = 0; u < U; u++) {
                    const uint32_t r = base + u * NT + tid;
                    uint32_t tf, tw;
                    const uint32_t exf = block_excl_scan_add((mt[u] >> 3) & 1u, lds_u, &tf);
                    const uint32_t exw = block_excl_scan_add((mt[u] >> 5) & 1u, lds_u, &tw);
                    if (mt[u] & 8u) farslot[r] = far_base + exf;
                    if (mt[u] & 32u) g.wslot[r] = wide_base + exw;
                    far_base += tf; wide_base += tw;
                }
            }
            if (tid == 0 && far_base > hrows && *sOk == 1) *sOk = 5;    // more far rows than the estimate: the host retries with a row per node
            if (tid == 0 && wide_base > wrows && *sOk == 1) *sOk = 7;   // more wide rows than the estimate: the host retries with more
            CSR_T(6, wide_base);
        }
#ifdef HX_CSR_PROF
        if (tid == 0 && last_seq && eidx % 500 == 0)
            printf("[csrprof] edge %u lanes %u V %u seqs %u: scatter %llu | B: ranks %llu records %llu neighbours %llu scans+stores %llu store drain %llu | C %llu | D %llu\n", eidx, NT, V2, 0u, cp[0], cp[1], cp[2], cp[3], cp[4], cp[7], cp[5], cp[6]);
#endif
#if !defined(HX_DP_PROF) && !defined(HX_GU_PROF)
        if (stats) {   // statistics of the rows the next DP will run over
            if (st_multi) atomicAdd(&ph[7], (unsigned long long)st_multi);
            if (st_ring | st_fifth) atomicAdd(&ph[8], (unsigned long long)st_ring | ((unsigned long long)st_fifth << 40));   // (high bits: fifth-and-later predecessor entries, fetched inside the row)
            if (st_far | st_wide) atomicAdd(&ph[9], (unsigned long long)st_far | ((unsigned long long)st_wide << 40));      // (high bits: rows with more than 4 predecessors)
            if (tid == 0) { atomicAdd(&ph[6], (unsigned long long)V2); atomicAdd(&ph[10], (unsigned long long)ktot); atomicAdd(&ph[11], 1ull); }
        }
#endif
    }

// spoa Graph::add_alignment by all lanes (poa_edge: "graph update"); the views' pointers as scalars, like the CSR rebuild
__device__ __forceinline__ void graph_update(const G& g_in, const uint8_t* seq_, const uint32_t L, const uint32_t na, const uint32_t nw /* leading entries in the traceback walk's (rank, column) form */,
                                             const uint32_t w_ie, const uint32_t w_je /* where the walk stopped */, uint32_t* lds_u, uint32_t* sV_, uint32_t* sE_, uint32_t* sNcand_, uint32_t* sOk_) {
    G g = g_in;
    g.stack = uptr(g.stack); g.aln_pos = uptr(g.aln_pos); g.aln_node = uptr(g.aln_node); g.code = uptr(g.code); g.n_aligned = uptr(g.n_aligned); g.aligned = uptr(g.aligned);
    g.score = uptr(g.score); g.row_pred1 = uptr(g.row_pred1); g.nrec = uptr(g.nrec); g.nrec2 = uptr(g.nrec2); g.out_head = uptr(g.out_head); g.out_tail = uptr(g.out_tail);
    g.in_head = uptr(g.in_head); g.in_tail = uptr(g.in_tail); g.e_next_out = uptr(g.e_next_out); g.e_next_in = uptr(g.e_next_in); g.e_to = uptr(g.e_to); g.e_from = uptr(g.e_from); g.e_w = uptr(g.e_w);
    const uint8_t* seq = uptr(seq_);
    const uint32_t tid = threadIdx.x, NT = blockDim.x;
    uint32_t* path = reinterpret_cast<uint32_t*>(g.score);   // node of every base of this sequence (vcap+1 words, free until the CSR build)
    uint32_t* colref = g.row_pred1;                          // column reference of every base (free until the CSR build)
    // spoa Graph::add_alignment, all lanes. A global alignment consumes every base exactly once and visits every aligned group
    // ("column") at most once, so bases are independent: base p looks at the node it was aligned to (reuse it, reuse a same-letter
    // member of its column, or open a new node that joins the column), and the edge (node of base p-1 -> node of base p) either
    // exists (weight += 2) or is appended. New node / edge ids are prefix sums in base order — the ids the serial walk hands out —
    // and every node gains at most one in-edge and one out-edge per sequence, so list appends never collide.
    const uint32_t V0 = *sV_, E0 = *sE_;
    int32_t* anode = reinterpret_cast<int32_t*>(g.stack);   // node aligned to base p, -1 = none (horizontal move)
    const bool room = E0 + L + 1 <= g.ecap && L <= g.vcap;   // edges: worst case (every base a new edge); per-base scratch lives in node pools; nodes are counted exactly below
    if (!room) { if (tid == 0) (*sOk_) = 0; }
    else {
        if (tid == 0) (*sNcand_) = 0;
        for (uint32_t p = tid; p < L; p += NT) anode[p] = -2;
        __syncthreads();
        // the alignment scattered to the bases, four entries per lane and iteration (their loads together: an iteration is two round trips). Entry k of the walk
        // = the cell it stood on before move k: the node of its row unless the move stayed in the row, its column unless the move stayed in the column.
        uint32_t nv = 0;
        g.rank2node = uptr(g.rank2node);
        for (uint32_t base = 0; base < na; base += 4 * NT) {
            int32_t r[4], c[4], r2[4], c2[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t k = base + u * NT + tid, kc = k < na ? k : 0u, kn = k + 1 < nw ? k + 1 : kc;
                r[u] = g.aln_node[kc]; c[u] = g.aln_pos[kc]; r2[u] = g.aln_node[kn]; c2[u] = g.aln_pos[kn];
                if (k + 1 >= nw) { r2[u] = (int32_t)w_ie; c2[u] = (int32_t)w_je; }
            }
            int32_t nd[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) { const uint32_t k = base + u * NT + tid; nd[u] = (int32_t)g.rank2node[k < nw && r[u] != r2[u] ? (uint32_t)(r[u] - 1) : 0u]; }
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t k = base + u * NT + tid;
                if (k >= na) continue;
                int32_t node = r[u], pos = c[u];
                if (k < nw) { node = r[u] == r2[u] ? -1 : nd[u]; pos = c[u] == c2[u] ? -1 : c[u] - 1; }
                if (pos != -1) { anode[pos] = node; nv++; }
            }
        }
        if (nv) atomicAdd(&(*sNcand_), nv);
        __syncthreads();
        const bool chain = na == 0;                 // empty graph: the sequence becomes a chain
        const bool par = chain || (*sNcand_) == L;      // always true for a global alignment
        if (!par) {                                 // (kept for safety: the serial walk handles any alignment shape)
            if (tid == 0) {
                for (uint32_t k = 0; k < nw; k++) {   // the walk's entries into the alignment's form, in place (entry k + 1 is read before it is rewritten)
                    const int32_t r = g.aln_node[k], c = g.aln_pos[k], r2 = k + 1 < nw ? g.aln_node[k + 1] : (int32_t)w_ie, c2 = k + 1 < nw ? g.aln_pos[k + 1] : (int32_t)w_je;
                    g.aln_node[k] = r == r2 ? -1 : (int32_t)g.rank2node[r - 1]; g.aln_pos[k] = c == c2 ? -1 : c - 1;
                }
                uint32_t V2 = V0, E2 = E0; if (!add_alignment(g, V2, E2, na, seq, L, path, colref)) (*sOk_) = 0; else { (*sV_) = V2; (*sE_) = E2; }
            }
        } else {
            // (round 5: lane = base. Bases are taken NT at a time - coalesced accesses by position, one scan per block with a running base for the ids -
            // where round 1 dealt them out in contiguous chunks per thread: every load of a wave then touched 64 different cache lines)
            uint32_t nbase = 0;
            bool ovf = false;
            for (uint32_t base = 0; base < L; base += NT) {
                const uint32_t p = base + tid;
                const bool on = p < L;
                uint32_t tgt = NONE;                // NONE = new node
                int32_t an = -1;
                uint8_t c = 0;
                if (on) {
                    c = seq[p];
                    an = chain ? -1 : anode[p];
                    if (an >= 0) {
                        if (g.code[an] == c) tgt = (uint32_t)an;
                        else for (uint32_t q = 0, nq = g.n_aligned[an]; q < nq; q++) { const uint32_t a = g.aligned[3 * an + q]; if (g.code[a] == c) { tgt = a; break; } }
                    }
                }
                const bool isnew = on && tgt == NONE;
                uint32_t tot;
                const uint32_t ex = block_excl_scan_add((uint32_t)isnew, lds_u, &tot);
                if (V0 + nbase + tot > g.vcap) { ovf = true; break; }   // the graph outgrows its workspace (same verdict on every lane): the host retries with more
                if (isnew) {
                    uint32_t vv = V0 + nbase + ex;
                    const uint32_t nn = add_node(g, vv, c);
                    if (an >= 0) {                  // joins the column of the node it was aligned to
                        for (uint32_t q = 0, nq = g.n_aligned[an]; q < nq; q++) { const uint32_t a = g.aligned[3 * an + q]; push_aligned(g, nn, a); push_aligned(g, a, nn); }
                        push_aligned(g, nn, (uint32_t)an); push_aligned(g, (uint32_t)an, nn);
                    }
                    tgt = nn;
                }
                if (on) { path[p] = tgt; colref[p] = an >= 0 ? (uint32_t)an : NONE; }
                nbase += tot;
            }
            const uint32_t newV = nbase;
            if (ovf) { if (tid == 0) (*sOk_) = 0; }
            else {
            __syncthreads();
            uint32_t ebase = 0;
            for (uint32_t base = 0; base < L; base += NT) {
                const uint32_t p = base + tid;
                const bool on = p >= 1 && p < L;
                uint32_t f = NONE, t = NONE, hit = NONE;
                if (on) {
                    f = path[p - 1]; t = path[p];
                    if (f < V0 && t < V0) for (uint32_t e = g.out_head[f]; e != NONE; e = g.e_next_out[e]) if (g.e_to[e] == t) { hit = e; break; }
                    if (hit != NONE) g.e_w[hit] += 2;
                }
                const bool isnew = on && hit == NONE;
                uint32_t tot;
                const uint32_t ex = block_excl_scan_add((uint32_t)isnew, lds_u, &tot);
                if (isnew) {
                    const uint32_t e = E0 + ebase + ex;
                    g.e_from[e] = f; g.e_to[e] = t; g.e_w[e] = 2; g.e_next_in[e] = NONE; g.e_next_out[e] = NONE;
                    link_edge(g, e, f, t);   // (a node gains at most one in-edge and one out-edge per sequence: different words of its records)
                }
                ebase += tot;
            }
            const uint32_t newE = ebase;
            if (tid == 0) { (*sV_) = V0 + newV; (*sE_) = E0 + newE; }
            }
        }
    }
}

// The order update of poa_edge
__device__ __forceinline__ void order_update(const G& g_in, const uint32_t V_old, const uint32_t V2, const uint32_t L, uint32_t* lds_u) {
    G g = g_in;
    g.stack = uptr(g.stack); g.row_pred0 = uptr(g.row_pred0); g.row_pred1 = uptr(g.row_pred1); g.node2rank = uptr(g.node2rank); g.n_aligned = uptr(g.n_aligned); g.aligned = uptr(g.aligned);
    g.score = uptr(g.score); g.rank2node = uptr(g.rank2node); g.pred = uptr(g.pred);
    const uint32_t tid = threadIdx.x, NT = blockDim.x;
    uint32_t* path = reinterpret_cast<uint32_t*>(g.score);
    uint32_t* colref = g.row_pred1;
    uint32_t* tmp_u32 = reinterpret_cast<uint32_t*>(g.pred);
    // Order update. Ranks keep every aligned group ("column") contiguous, like the reference's order does: a later sequence may
    // enter a column through one member and continue from another, so edges must run from earlier columns to later ones.
    // The new sequence's path visits existing columns in increasing rank. Each new node gets an insertion point X in the OLD order:
    //   new mismatch node (joins the column of the old node it was aligned to)  -> X = last rank of that column + 1
    //   new unaligned node (a new column)                                       -> X = first rank of the next existing column on the path (or the end)
    // Nodes with the same X keep path order (X never decreases along the path). New rank of an old node = old rank + #new nodes
    // with X <= old rank: one prefix sum over the old order instead of a serial DFS over the whole graph.
    
    uint32_t* ins = g.stack;          // V_old+1 counters, then their exclusive prefix
    uint32_t* xq = g.row_pred0;       // insertion point of every new node, by sequence position (free until the CSR build)
    if (V_old == 0) {
        for (uint32_t r = tid; r < V2; r += NT) g.rank2node[r] = r;
    } else {
        uint32_t* firstidx = g.stack + (V_old + 1);   // per insertion point: the first new node (in path order) that goes there
        for (uint32_t r = tid; r <= V_old; r += NT) { ins[r] = 0; firstidx[r] = NONE; }
        __syncthreads();
        {
            // every base on its own: new node ids are consecutive in path order, so "position among the new nodes" = id - V_old
            auto col_first = [&](uint32_t n) { uint32_t f = g.node2rank[n]; for (uint32_t k = 0, na = g.n_aligned[n]; k < na; k++) { const uint32_t a = g.aligned[3 * n + k]; if (a < V_old) f = min(f, g.node2rank[a]); } return f; };
            auto col_last = [&](uint32_t n) { uint32_t f = g.node2rank[n]; for (uint32_t k = 0, na = g.n_aligned[n]; k < na; k++) { const uint32_t a = g.aligned[3 * n + k]; if (a < V_old) f = max(f, g.node2rank[a]); } return f; };
            for (uint32_t q = tid; q < L; q += NT) {
                const uint32_t n = path[q];
                if (n < V_old) continue;
                uint32_t X;
                if (colref[q] != NONE) X = col_last(colref[q]) + 1;
                else {                                                // unaligned new node: the next existing column on the path
                    uint32_t q2 = q + 1;
                    while (q2 < L && path[q2] >= V_old && colref[q2] == NONE) q2++;
                    X = q2 < L ? col_first(path[q2] < V_old ? path[q2] : colref[q2]) : V_old;
                }
                xq[q] = X;
                atomicAdd(&ins[X], 1u);
                atomicMin(&firstidx[X], n - V_old);
            }
        }
        __syncthreads();
        uint32_t ibase = 0;
        for (uint32_t base = 0; base <= V_old; base += NT) {      // (lane = old rank, NT at a time: one scan per block with a running base)
            const uint32_t r = base + tid;
            const uint32_t c = r <= V_old ? ins[r] : 0u;
            uint32_t tot;
            const uint32_t ex = ibase + block_excl_scan_add(c, lds_u, &tot);
            if (r <= V_old) {
                ins[r] = ex;                                        // new nodes with X == r start at r + ex
                if (r < V_old) tmp_u32[r + ex + c] = g.rank2node[r];    // the old node itself moves behind them
            }
            ibase += tot;
        }
        __syncthreads();
        for (uint32_t q = tid; q < L; q += NT) {   // nodes with the same insertion point keep path order
            const uint32_t n = path[q];
            if (n < V_old) continue;
            const uint32_t X = xq[q];
            tmp_u32[X + ins[X] + (n - V_old - firstidx[X])] = n;
        }
        __syncthreads();
        for (uint32_t r = tid; r < V2; r += NT) g.rank2node[r] = tmp_u32[r];
    }
}

