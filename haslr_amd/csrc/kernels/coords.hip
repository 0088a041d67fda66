// coords.hip — K5: per backbone edge, the read coordinates of the gap between its two anchor contigs.
//
// Replaces asm_calc_single_edge_coordinates (Assemble.cpp:157-363) with its helpers
// asm_best_supported_interval_contig1/2 (:24-126) and asm_find_lr_pos (:129-155).
// One 64-lane workgroup per edge:
//   1. rank-sort the (t_start,i)/(t_end,i) lists of both anchors (lanes = supports)
//   2. lane 0 sweeps each pair of lists (a 2n-step merge) and records the merged-order position of every
//      open/close event and the position of the best open; set membership at that moment is then evaluated
//      by all lanes (equivalent to the reference's std::set copy at each new best)
//   3. ascending intersection of the two sets (ballot compaction)
//   4. lanes = surviving supports: two run-length CIGAR walks each (no per-base expansion, no sscanf),
//      ordered compaction of the valid ones into the edge's output slice
#include "kernels.h"

namespace hxk {

namespace {

__device__ long long find_lr_pos(const CgView& v, bool reversed, uint32_t lr, uint32_t c, int lstep, int cstep, uint32_t contig_pos) {
    if ((cstep > 0 && c > contig_pos) || (cstep < 0 && c < contig_pos)) return -1;
    // Eight op words are fetched at a time (independent addresses, one memory round trip), then walked from registers: the walk's exit
    // depends on the data, so one-op-at-a-time costs a full memory latency per op.
    const uint64_t n = v.e - v.b;
    bool stop = false;
    for (uint64_t k0 = 0; k0 < n && !stop; k0 += 8) {
        uint32_t w[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint64_t kk = k0 + u < n ? k0 + u : n - 1; w[u] = v.ops[reversed ? v.e - 1 - kk : v.b + kk]; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (stop || k0 + u >= n) continue;
            const uint64_t g = reversed ? v.e - 1 - (k0 + u) : v.b + (k0 + u);
            uint32_t len = HX_CG_LEN(w[u]);
            if (g == v.b) len -= v.skf;
            if (g + 1 == v.e) len -= v.skb;
            if (len == 0) continue;
            const uint32_t code = HX_CG_OP(w[u]);
            const uint32_t d = cstep > 0 ? contig_pos - c : c - contig_pos;
            if (code == HX_CG_I) {
                if (d == 0) { stop = true; continue; }
                lr += len * lstep;
            } else {
                if (d < len) {
                    if (code == HX_CG_M) lr += d * lstep;
                    stop = true; continue;
                }
                if (code == HX_CG_M) lr += len * lstep;
                c += len * cstep;
            }
        }
    }
    return (long long)lr;
}

constexpr uint32_t NEVER = 0xffffffffu;

// lane 0: the merge sweep of Assemble.cpp:39-71 / :91-123. Records event positions instead of copying sets.
__device__ void sweep(const uint64_t* beg, const uint64_t* end, uint32_t n, bool last_max, uint32_t* open_step, uint32_t* close_step,
                      uint32_t& beg_best, uint32_t& end_best, uint32_t& t_best) {
    int curr = 0, best = 0;
    uint32_t i = 0, j = 0, step = 0;
    bool started = false;
    beg_best = end_best = 0; t_best = NEVER;
    while (i < n && j < n) {
        uint32_t bv = (uint32_t)(beg[i] >> 32), ev = (uint32_t)(end[j] >> 32);
        if (bv < ev) {
            curr++;
            open_step[(uint32_t)beg[i]] = step;
            if (last_max ? curr >= best : curr > best) { best = curr; beg_best = bv; t_best = step; started = true; }
            i++;
        } else {
            if (started) { end_best = ev; started = false; }
            curr--;
            close_step[(uint32_t)end[j]] = step;
            j++;
        }
        step++;
    }
    if (started && j < n) end_best = (uint32_t)(end[j] >> 32);
}

__global__ void __launch_bounds__(64) k_edge_coords(EdgeRecs R, const uint64_t* __restrict__ edge_key, const uint64_t* __restrict__ edge_off,
                                                    const uint32_t* __restrict__ cg_ops, const uint32_t* __restrict__ contig_len,
                                                    const uint32_t* __restrict__ read_len, uint32_t n_sel, const uint32_t* __restrict__ sel_edge,
                                                    const uint64_t* __restrict__ cap_off, CoordsScratch sc, uint32_t* head_end, uint32_t* tail_beg,
                                                    uint32_t* n_supp, uint32_t* supp_lr, uint32_t* spos, uint32_t* epos) {
    const uint32_t s = blockIdx.x, lane = threadIdx.x;
    const uint32_t e = sel_edge[s];
    const uint64_t key = edge_key[e];
    const uint32_t v1 = (uint32_t)(key >> 32), to = (uint32_t)key;
    const uint32_t node1 = v1 >> 1, rev1 = v1 & 1, node2 = to >> 1, rev2 = to & 1;
    const uint64_t b = edge_off[e];
    const uint32_t n = (uint32_t)(edge_off[e + 1] - b);
    const bool hairpin = (to ^ 1u) == v1;
    const uint64_t so = cap_off[s];
    __shared__ uint32_t sh[8];

    // ---- 1. rank sort (pairs compare as (value, index), the reference's std::sort on pair<uint32,uint32>)
    for (uint32_t i = lane; i < n; i += 64) {
        uint64_t x1 = ((uint64_t)R.head.t_start[b + i] << 32) | i, y1 = ((uint64_t)R.head.t_end[b + i] << 32) | i;
        uint64_t x2 = ((uint64_t)R.tail.t_start[b + i] << 32) | i, y2 = ((uint64_t)R.tail.t_end[b + i] << 32) | i;
        uint32_t r1 = 0, q1 = 0, r2 = 0, q2 = 0;
        for (uint32_t j = 0; j < n; j++) {
            r1 += ((((uint64_t)R.head.t_start[b + j] << 32) | j) < x1);
            q1 += ((((uint64_t)R.head.t_end[b + j] << 32) | j) < y1);
            r2 += ((((uint64_t)R.tail.t_start[b + j] << 32) | j) < x2);
            q2 += ((((uint64_t)R.tail.t_end[b + j] << 32) | j) < y2);
        }
        sc.beg1[so + r1] = x1; sc.end1[so + q1] = y1; sc.beg2[so + r2] = x2; sc.end2[so + q2] = y2;
    }
    // step records: reuse the output slice as scratch (4 x uint32 per support are needed; cur/best give 3 bytes, so use
    // the u64 arrays' neighbours): open/close steps live in best_list's slice pairs
    uint32_t* open1 = supp_lr + so;      // output slices are written only in phase 4, after these are dead
    uint32_t* close1 = spos + so;
    uint32_t* open2 = epos + so;
    uint32_t* close2 = sc.best_list + so;
    for (uint32_t i = lane; i < n; i += 64) { open1[i] = NEVER; close1[i] = NEVER; open2[i] = NEVER; close2[i] = NEVER; }
    __syncthreads();
    // ---- 2. sweeps
    if (lane == 0) {
        uint32_t bb1, eb1, t1, bb2, eb2, t2;
        sweep(sc.beg1 + so, sc.end1 + so, n, true, open1, close1, bb1, eb1, t1);
        sweep(sc.beg2 + so, sc.end2 + so, n, false, open2, close2, bb2, eb2, t2);
        sh[0] = rev1 == 0 ? eb1 - 1 : bb1;    // contig1_pos  (Assemble.cpp:228-231)
        sh[1] = rev2 == 0 ? bb2 : eb2 - 1;    // contig2_pos  (:232-235)
        sh[2] = t1; sh[3] = t2;
    }
    __syncthreads();
    const uint32_t c1pos = sh[0], c2pos = sh[1], t1 = sh[2], t2 = sh[3];
    // membership at the best open: opened at or before it, and not closed in between
    for (uint32_t i = lane; i < n; i += 64) {
        bool in1 = t1 != NEVER && open1[i] <= t1 && !(close1[i] != NEVER && close1[i] < t1 && close1[i] > open1[i]);
        bool in2 = t2 != NEVER && open2[i] <= t2 && !(close2[i] != NEVER && close2[i] < t2 && close2[i] > open2[i]);
        sc.cur[so + i] = (in1 && in2) ? 1 : 0;
    }
    __syncthreads();
    // ---- 3. ascending intersection list (into beg1's slice, which is dead now)
    uint32_t* best = reinterpret_cast<uint32_t*>(sc.beg1 + so);
    uint32_t nbest = 0;
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (uint32_t base = 0; base < n; base += 64) {
        uint32_t i = base + lane;
        bool f = i < n && sc.cur[so + i];
        uint64_t m = __ballot(f);
        if (f) best[nbest + __popcll(m & lt)] = i;
        nbest += __popcll(m);
    }
    __syncthreads();
    // ---- 4. CIGAR walks, lanes = supports of the intersection, ordered compaction
    uint32_t nout = 0;
    for (uint32_t base = 0; base < nbest; base += 64) {
        uint32_t k = base + lane;
        bool valid = false;
        uint32_t o_lr = 0, o_sp = 0, o_ep = 0, m_lr = 0, m_sp = 0, m_ep = 0;
        if (k < nbest) {
            uint64_t x = b + best[k];
            uint32_t rid = R.lr[x] & 0x7fffffffu, rlen = read_len[rid];
            uint32_t rstrand = (rev1 == R.head.is_rev[x]) ? 0 : 1;   // :262
            CgView vh{cg_ops, R.head.cg_begin[x], R.head.cg_end[x], R.head.cg_skip_front[x], R.head.cg_skip_back[x]};
            CgView vt{cg_ops, R.tail.cg_begin[x], R.tail.cg_end[x], R.tail.cg_skip_front[x], R.tail.cg_skip_back[x]};
            uint32_t hqs = R.head.q_start[x], hqe = R.head.q_end[x], hts = R.head.t_start[x], hte = R.head.t_end[x];
            uint32_t tqs = R.tail.q_start[x], tqe = R.tail.q_end[x], tts = R.tail.t_start[x], tte = R.tail.t_end[x];
            long long ls, le;
            if (rstrand == 0) {   // cases 1-4
                ls = rev1 == 0 ? find_lr_pos(vh, false, hqs, hts, +1, +1, c1pos) : find_lr_pos(vh, true, hqs, hte - 1, +1, -1, c1pos);
                le = rev2 == 0 ? find_lr_pos(vt, true, tqe - 1, tte - 1, -1, -1, c2pos) : find_lr_pos(vt, false, tqe - 1, tts, -1, +1, c2pos);
            } else {              // cases 5-8
                ls = rev1 == 0 ? find_lr_pos(vh, false, rlen - hqe, hts, +1, +1, c1pos) : find_lr_pos(vh, true, rlen - hqe, hte - 1, +1, -1, c1pos);
                le = rev2 == 0 ? find_lr_pos(vt, true, rlen - tqs - 1, tte - 1, -1, -1, c2pos) : find_lr_pos(vt, false, rlen - tqs - 1, tts, -1, +1, c2pos);
            }
            if (ls != -1 && le != -1) {
                valid = true;
                o_lr = rid | (rstrand << 31); o_sp = (uint32_t)(ls + 1); o_ep = (uint32_t)(le - 1);
                m_lr = rid | ((1 - rstrand) << 31); m_sp = (uint32_t)(rlen - (le - 1) - 1); m_ep = (uint32_t)(rlen - (ls + 1) - 1);
            }
        }
        __syncthreads();   // phase-2 scratch that aliases the output slice is dead from here on
        uint64_t m = __ballot(valid);
        if (valid) {
            uint32_t p = nout + __popcll(m & lt);
            if (!hairpin) { supp_lr[so + p] = o_lr; spos[so + p] = o_sp; epos[so + p] = o_ep; }
            else {   // edge == twin: the mirrored entry goes into the same list (Assemble.cpp:330-331)
                supp_lr[so + 2 * p] = o_lr; spos[so + 2 * p] = o_sp; epos[so + 2 * p] = o_ep;
                supp_lr[so + 2 * p + 1] = m_lr; spos[so + 2 * p + 1] = m_sp; epos[so + 2 * p + 1] = m_ep;
            }
        }
        nout += __popcll(m);
    }
    if (lane == 0) {
        uint32_t total = hairpin ? 2 * nout : nout;
        n_supp[s] = total;
        if (total > 0) { head_end[s] = c1pos; tail_beg[s] = c2pos; }
        else { head_end[s] = rev1 == 0 ? contig_len[node1] - 1 : 0; tail_beg[s] = rev2 == 0 ? 0 : contig_len[node2] - 1; }
    }
}

__global__ void k_coords_compact(const uint64_t* __restrict__ cap_off, const uint64_t* __restrict__ out_off, uint32_t n_sel,
                                 const uint32_t* lr_in, const uint32_t* sp_in, const uint32_t* ep_in, uint32_t* lr_out, uint32_t* sp_out, uint32_t* ep_out) {
    uint32_t s = blockIdx.x;
    if (s >= n_sel) return;
    uint64_t si = cap_off[s], so = out_off[s], n = out_off[s + 1] - so;
    for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) { lr_out[so + i] = lr_in[si + i]; sp_out[so + i] = sp_in[si + i]; ep_out[so + i] = ep_in[si + i]; }
}

}  // namespace

void edge_coords(const EdgeRecs& recs, const uint64_t* edge_key, const uint64_t* edge_off, const uint32_t* cg_ops, const uint32_t* contig_len,
                 const uint32_t* read_len, uint32_t n_sel, const uint32_t* sel_edge, const uint64_t* sel_rec_off, const CoordsScratch& sc,
                 uint32_t* head_end, uint32_t* tail_beg, uint32_t* n_supp, uint32_t* supp_lr, uint32_t* spos, uint32_t* epos, hipStream_t s) {
    if (n_sel) k_edge_coords<<<n_sel, 64, 0, s>>>(recs, edge_key, edge_off, cg_ops, contig_len, read_len, n_sel, sel_edge, sel_rec_off, sc,
                                                 head_end, tail_beg, n_supp, supp_lr, spos, epos);
}

void coords_compact(const uint64_t* cap_off, const uint64_t* out_off, uint32_t n_sel, const uint32_t* lr_in, const uint32_t* sp_in,
                    const uint32_t* ep_in, uint32_t* lr_out, uint32_t* sp_out, uint32_t* ep_out, hipStream_t s) {
    if (n_sel) k_coords_compact<<<n_sel, 64, 0, s>>>(cap_off, out_off, n_sel, lr_in, sp_in, ep_in, lr_out, sp_out, ep_out);
}

}  // namespace hxk
