// coords.hip — K5: per backbone edge, the read coordinates of the gap between its two anchor contigs.
//
// Replaces asm_calc_single_edge_coordinates (Assemble.cpp:157-363) with its helpers
// asm_best_supported_interval_contig1/2 (:24-126) and asm_find_lr_pos (:129-155).
// One workgroup of four wavefronts per edge:
//   1. rank-sort the (t_start,i)/(t_end,i) lists of both anchors (threads = supports); the sorted lists and the event records of
//      edges with at most LDS_SUPP supports stay in LDS
//   2. one lane of wave 0 and one of wave 1 sweep the two pairs of lists (a 2n-step merge each) and record the merged-order position
//      of every open/close event and the position of the best open; set membership at that moment is then evaluated by all threads
//      (equivalent to the reference's std::set copy at each new best)
//   3. ascending intersection of the two sets (ballot compaction)
//   4. the run-length CIGAR walks (two per surviving support; no per-base expansion, no sscanf) are dealt to the four wavefronts, and
//      every walk to the 64 lanes of its wavefront (round 4; K2's formulation: a lane sums its slice of the ops, a wave prefix sum gives
//      each slice its start state, one ballot finds the op where the walk stops) - a 50 kb anchor alignment is ~15 000 ops, which one
//      lane per support walked through ~1 900 dependent memory round trips
//   5. ordered compaction of the valid supports into the edge's output slice
#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace hxk {

namespace {

__device__ __forceinline__ uint32_t wave_excl_add(uint32_t v, uint32_t lane) {
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(x, d, 64); if ((int)lane >= d) x += o; }
    return x - v;
}

// asm_find_lr_pos (Assemble.cpp:129-155) with the ops of the view dealt to the 64 lanes of a wavefront; every lane returns the same value.
// The serial walk: skip ops of effective length 0; d = contig bases to go; an insertion stops the walk at d == 0 and otherwise advances the
// read; any other op stops it when d < len (a match advances the read by d first) and otherwise advances the contig (and, a match, the read).
// 32-bit wrap-around arithmetic throughout, as in the reference.
__device__ long long find_lr_pos_wave(const CgView& v, bool reversed, uint32_t lr0, uint32_t c0, int lstep, int cstep, uint32_t contig_pos, uint32_t lane) {
    if ((cstep > 0 && c0 > contig_pos) || (cstep < 0 && c0 < contig_pos)) return -1;
    const uint32_t D0 = cstep > 0 ? contig_pos - c0 : c0 - contig_pos;   // contig bases to go at the start
    const uint64_t n = v.e - v.b;
    const uint64_t C = (n + 63) / 64, k0 = n < (uint64_t)lane * C ? n : (uint64_t)lane * C, k1 = n < k0 + C ? n : k0 + C;
    // pass 1: what the lane's slice adds to the contig and read counters
    uint32_t sCon = 0, sRead = 0;
    for (uint64_t k = k0; k < k1; k++) {
        const uint64_t g = reversed ? v.e - 1 - k : v.b + k;
        const uint32_t len = v.eff(g), code = HX_CG_OP(v.ops[g]);
        if (code == HX_CG_I) sRead += len;
        else { sCon += len; if (code == HX_CG_M) sRead += len; }
    }
    const uint32_t bCon = wave_excl_add(sCon, lane), bRead = wave_excl_add(sRead, lane);
    // pass 2: the first op of the slice at which the serial walk stops (slices behind the stopping one start from counters the serial walk
    // never reaches: the first stopping lane is the answer)
    uint32_t cc = bCon, cr = bRead, res = 0;
    bool stop = false;
    if (bCon <= D0) {
        for (uint64_t k = k0; k < k1; k++) {
            const uint64_t g = reversed ? v.e - 1 - k : v.b + k;
            const uint32_t len = v.eff(g);
            if (len == 0) continue;
            const uint32_t code = HX_CG_OP(v.ops[g]);
            const uint32_t d = D0 - cc;
            if (code == HX_CG_I) {
                if (d == 0) { stop = true; res = cr; break; }
                cr += len;
            } else {
                if (d < len) { stop = true; res = code == HX_CG_M ? cr + d : cr; break; }
                if (code == HX_CG_M) cr += len;
                cc += len;
            }
        }
    }
    const unsigned long long stopMask = __ballot(stop);
    const uint32_t adv = stopMask ? __shfl(res, __builtin_ctzll(stopMask), 64) : __shfl(bRead + sRead, 63, 64);   // (the ops ran out: everything counted)
    return (long long)(uint32_t)(lr0 + adv * (uint32_t)lstep);
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

constexpr uint32_t K5_NT = 256;        // threads per edge (four wavefronts)
constexpr uint32_t LDS_SUPP = 384;     // supports per edge up to which the sorted lists and the event records stay in LDS (20 B per support and list pair)

__global__ void __launch_bounds__(K5_NT) k_edge_coords(EdgeRecs R, const uint64_t* __restrict__ edge_key, const uint64_t* __restrict__ edge_off,
                                                    const uint32_t* __restrict__ cg_ops, const uint32_t* __restrict__ contig_len,
                                                    const uint32_t* __restrict__ read_len, uint32_t n_sel, const uint32_t* __restrict__ sel_edge,
                                                    const uint64_t* __restrict__ cap_off, CoordsScratch sc, uint32_t* head_end, uint32_t* tail_beg,
                                                    uint32_t* n_supp, uint32_t* supp_lr, uint32_t* spos, uint32_t* epos, uint32_t lds_supp) {
    const uint32_t s = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t e = sel_edge[s];
    const uint64_t key = edge_key[e];
    const uint32_t v1 = (uint32_t)(key >> 32), to = (uint32_t)key;
    const uint32_t node1 = v1 >> 1, rev1 = v1 & 1, node2 = to >> 1, rev2 = to & 1;
    const uint64_t b = edge_off[e];
    const uint32_t n = (uint32_t)(edge_off[e + 1] - b);
    const bool hairpin = (to ^ 1u) == v1;
    const uint64_t so = cap_off[s];
    __shared__ uint32_t sh[8];
    __shared__ uint64_t l_sorted[4 * LDS_SUPP];
    __shared__ uint32_t l_step[4 * LDS_SUPP];
    const bool in_lds = n <= lds_supp;   // (LDS_SUPP; option coords_lds_supp=0 sends every edge through the global scratch: testing)
    // (generic pointers: the sweep's dependent loads are LDS round trips for nearly every edge, HBM ones only for an edge with hundreds of supports)
    uint64_t* beg1 = in_lds ? l_sorted : sc.beg1 + so;
    uint64_t* end1 = in_lds ? l_sorted + LDS_SUPP : sc.end1 + so;
    uint64_t* beg2 = in_lds ? l_sorted + 2 * LDS_SUPP : sc.beg2 + so;
    uint64_t* end2 = in_lds ? l_sorted + 3 * LDS_SUPP : sc.end2 + so;
    // step records of the global path: the output slices are written only in phase 5, after these are dead
    uint32_t* open1 = in_lds ? l_step : supp_lr + so;
    uint32_t* close1 = in_lds ? l_step + LDS_SUPP : spos + so;
    uint32_t* open2 = in_lds ? l_step + 2 * LDS_SUPP : epos + so;
    uint32_t* close2 = in_lds ? l_step + 3 * LDS_SUPP : sc.best_list + so;

    // ---- 1. rank sort (pairs compare as (value, index), the reference's std::sort on pair<uint32,uint32>)
    for (uint32_t i = tid; i < n; i += K5_NT) {
        uint64_t x1 = ((uint64_t)R.head.t_start[b + i] << 32) | i, y1 = ((uint64_t)R.head.t_end[b + i] << 32) | i;
        uint64_t x2 = ((uint64_t)R.tail.t_start[b + i] << 32) | i, y2 = ((uint64_t)R.tail.t_end[b + i] << 32) | i;
        uint32_t r1 = 0, q1 = 0, r2 = 0, q2 = 0;
        for (uint32_t j = 0; j < n; j++) {
            r1 += ((((uint64_t)R.head.t_start[b + j] << 32) | j) < x1);
            q1 += ((((uint64_t)R.head.t_end[b + j] << 32) | j) < y1);
            r2 += ((((uint64_t)R.tail.t_start[b + j] << 32) | j) < x2);
            q2 += ((((uint64_t)R.tail.t_end[b + j] << 32) | j) < y2);
        }
        beg1[r1] = x1; end1[q1] = y1; beg2[r2] = x2; end2[q2] = y2;
        open1[i] = NEVER; close1[i] = NEVER; open2[i] = NEVER; close2[i] = NEVER;
    }
    __syncthreads();
    // ---- 2. sweeps: anchor 1 on wave 0, anchor 2 on wave 1
    if (tid == 0) {
        uint32_t bb1, eb1, t1;
        sweep(beg1, end1, n, true, open1, close1, bb1, eb1, t1);
        sh[0] = rev1 == 0 ? eb1 - 1 : bb1;    // contig1_pos  (Assemble.cpp:228-231)
        sh[2] = t1;
    } else if (tid == 64) {
        uint32_t bb2, eb2, t2;
        sweep(beg2, end2, n, false, open2, close2, bb2, eb2, t2);
        sh[1] = rev2 == 0 ? bb2 : eb2 - 1;    // contig2_pos  (:232-235)
        sh[3] = t2;
    }
    __syncthreads();
    const uint32_t c1pos = sh[0], c2pos = sh[1], t1 = sh[2], t2 = sh[3];
    // membership at the best open: opened at or before it, and not closed in between
    for (uint32_t i = tid; i < n; i += K5_NT) {
        bool in1 = t1 != NEVER && open1[i] <= t1 && !(close1[i] != NEVER && close1[i] < t1 && close1[i] > open1[i]);
        bool in2 = t2 != NEVER && open2[i] <= t2 && !(close2[i] != NEVER && close2[i] < t2 && close2[i] > open2[i]);
        sc.cur[so + i] = (in1 && in2) ? 1 : 0;
    }
    __syncthreads();
    // ---- 3. ascending intersection list (wave 0; into the global beg1 slice, which the LDS path never used and the global path is done with)
    uint32_t* best = reinterpret_cast<uint32_t*>(sc.beg1 + so);
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if (wv == 0) {
        uint32_t nb = 0;
        for (uint32_t base = 0; base < n; base += 64) {
            uint32_t i = base + lane;
            bool f = i < n && sc.cur[so + i];
            uint64_t m = __ballot(f);
            if (f) best[nb + __popcll(m & lt)] = i;
            nb += __popcll(m);
        }
        if (lane == 0) sh[4] = nb;
    }
    __syncthreads();
    const uint32_t nbest = sh[4];
    // ---- 4. CIGAR walks: item 2k = the head anchor of surviving support k, item 2k + 1 = its tail anchor; a wavefront per item
    long long* const pos = reinterpret_cast<long long*>(sc.end1 + so);     // [2 * nbest] (end1 and end2 are adjacent slices of n words each in
    long long* const pos2 = reinterpret_cast<long long*>(sc.end2 + so);    //  separate arrays: heads go to end1's, tails to end2's)
    for (uint32_t it = wv; it < 2 * nbest; it += K5_NT / 64) {
        const uint32_t k = it >> 1;
        const uint64_t x = b + best[k];
        const uint32_t rid = R.lr[x] & 0x7fffffffu, rlen = read_len[rid];
        const uint32_t rstrand = (rev1 == R.head.is_rev[x]) ? 0 : 1;   // :262
        long long r;
        if ((it & 1u) == 0) {
            const CgView vh{cg_ops, R.head.cg_begin[x], R.head.cg_end[x], R.head.cg_skip_front[x], R.head.cg_skip_back[x]};
            const uint32_t hqs = R.head.q_start[x], hqe = R.head.q_end[x], hts = R.head.t_start[x], hte = R.head.t_end[x];
            const uint32_t q0 = rstrand == 0 ? hqs : rlen - hqe;       // cases 1-4 / 5-8
            r = rev1 == 0 ? find_lr_pos_wave(vh, false, q0, hts, +1, +1, c1pos, lane) : find_lr_pos_wave(vh, true, q0, hte - 1, +1, -1, c1pos, lane);
            if (lane == 0) pos[k] = r;
        } else {
            const CgView vt{cg_ops, R.tail.cg_begin[x], R.tail.cg_end[x], R.tail.cg_skip_front[x], R.tail.cg_skip_back[x]};
            const uint32_t tqs = R.tail.q_start[x], tqe = R.tail.q_end[x], tts = R.tail.t_start[x], tte = R.tail.t_end[x];
            const uint32_t q0 = rstrand == 0 ? tqe - 1 : rlen - tqs - 1;
            r = rev2 == 0 ? find_lr_pos_wave(vt, true, q0, tte - 1, -1, -1, c2pos, lane) : find_lr_pos_wave(vt, false, q0, tts, -1, +1, c2pos, lane);
            if (lane == 0) pos2[k] = r;
        }
    }
    __syncthreads();   // (also: the phase-2 scratch that aliases the output slice is dead from here on)
    // ---- 5. ordered compaction of the valid supports (wave 0)
    if (wv != 0) return;
    uint32_t nout = 0;
    for (uint32_t base = 0; base < nbest; base += 64) {
        uint32_t k = base + lane;
        bool valid = false;
        uint32_t o_lr = 0, o_sp = 0, o_ep = 0, m_lr = 0, m_sp = 0, m_ep = 0;
        if (k < nbest) {
            const uint64_t x = b + best[k];
            const uint32_t rid = R.lr[x] & 0x7fffffffu, rlen = read_len[rid];
            const uint32_t rstrand = (rev1 == R.head.is_rev[x]) ? 0 : 1;
            const long long ls = pos[k], le = pos2[k];
            if (ls != -1 && le != -1) {
                valid = true;
                o_lr = rid | (rstrand << 31); o_sp = (uint32_t)(ls + 1); o_ep = (uint32_t)(le - 1);
                m_lr = rid | ((1 - rstrand) << 31); m_sp = (uint32_t)(rlen - (le - 1) - 1); m_ep = (uint32_t)(rlen - (ls + 1) - 1);
            }
        }
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
                 uint32_t* head_end, uint32_t* tail_beg, uint32_t* n_supp, uint32_t* supp_lr, uint32_t* spos, uint32_t* epos, int lds_supp_opt, hipStream_t s) {
    const uint32_t lds_supp = lds_supp_opt >= 0 ? std::min<uint32_t>(LDS_SUPP, (uint32_t)lds_supp_opt) : LDS_SUPP;   // (option coords_lds_supp; 0 sends every edge through the global scratch: testing)
    if (n_sel) k_edge_coords<<<n_sel, K5_NT, 0, s>>>(recs, edge_key, edge_off, cg_ops, contig_len, read_len, n_sel, sel_edge, sel_rec_off, sc,
                                                 head_end, tail_beg, n_supp, supp_lr, spos, epos, lds_supp);
}

void coords_compact(const uint64_t* cap_off, const uint64_t* out_off, uint32_t n_sel, const uint32_t* lr_in, const uint32_t* sp_in,
                    const uint32_t* ep_in, uint32_t* lr_out, uint32_t* sp_out, uint32_t* ep_out, hipStream_t s) {
    if (n_sel) k_coords_compact<<<n_sel, 64, 0, s>>>(cap_off, out_off, n_sel, lr_in, sp_in, ep_in, lr_out, sp_out, ep_out);
}

}  // namespace hxk
