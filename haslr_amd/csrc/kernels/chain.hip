// chain.hip — K0..K3: per-long-read contig-hit filter / sort / dedup / overlap-trim / chain.
//
// Replaces the reference's serial loops over reads (paths under /root/reference/src/haslr_assemble/src/):
//   filters 1-4            Longread.cpp:262-272      sort by (q_end,q_start)   :52-55,:256
//   process_lr_alignment_group :182-232 (<=1 hit dropped, palindrome truncation, filter 5)
//   fix_overlapping_alignments :430-512 + find_contig_pos :375-420 (on run-length ops, no per-base strings)
//   find_best_scheduling   :524-610 (weighted interval scheduling, strict '>' tie rule)
//
// Mapping (round 1): one lane per read. A read's raw hits are contiguous (PAF grouped by query), so lane r
// streams hits [read_hit_off[r], read_hit_off[r+1]) and neighbouring lanes touch neighbouring memory.
// Scratch for a read lives at the same offsets as its raw hits (a read never has more survivors than raw
// hits); a scan over the per-read counts followed by chain_compact produces the dense tables.
#include "kernels.h"

namespace hxk {

__global__ void k_contig_class(const double* km, uint32_t n, double thr_load, double thr_uniq, uint8_t* cls) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = km[i];
    uint8_t c = 0;
    if (v > thr_load) c |= HXC_DROP_LOAD;
    if (v < thr_uniq) c |= HXC_UNIQUE;
    if (v > thr_uniq) c |= HXC_DROP_CHAIN;
    if (v <= thr_uniq) c |= HXC_EDGE_OK;
    cls[i] = c;
}

void contig_class(const double* mean_kmer, uint32_t n, double thr_load, double thr_uniq, uint8_t* cls, hipStream_t s) {
    if (n) k_contig_class<<<(n + 255) / 256, 256, 0, s>>>(mean_kmer, n, thr_load, thr_uniq, cls);
}

namespace {

struct TrimRes {
    bool ok;
    uint32_t lr, c, kept, nmatch;
    uint64_t last_run;
    uint32_t kept_in_last;
};

// find_contig_pos (Longread.cpp:375-420) on run-length ops; see oracle/oracle.cpp for the derivation of the
// run-length form. Stop BEFORE the first per-base op at which lr == lr_pos; if that op is not M fall back
// to the last M before it (undoing M/I/D but not other ops, as the reference does).
__device__ TrimRes trim_walk(const CgView& v, bool reversed, uint32_t lr, uint32_t c, int lstep, int cstep, uint32_t lr_pos) {
    TrimRes r;
    r.ok = false; r.lr = r.c = r.kept = r.nmatch = r.kept_in_last = 0; r.last_run = 0;
    bool haveM = false;
    uint64_t m_g = 0;
    uint32_t m_len = 0, m_lr = 0, m_c = 0, m_idx = 0, m_m = 0;
    uint32_t idx = 0, mcount = 0, other_extra = 0;
    const uint64_t n = v.e - v.b;
    for (uint64_t k = 0; k < n; k++) {
        uint64_t g = reversed ? v.e - 1 - k : v.b + k;
        uint32_t len = v.eff(g);
        if (len == 0) continue;
        uint32_t code = HX_CG_OP(v.ops[g]);
        uint32_t d = lstep > 0 ? lr_pos - lr : lr - lr_pos;
        if (code == HX_CG_M || code == HX_CG_I) {
            if (d < len) {
                if (code == HX_CG_M) {
                    r.ok = true;
                    r.lr = lr + d * lstep; r.c = c + d * cstep;
                    r.kept = idx + d + 1; r.nmatch = mcount + d + 1;
                    r.last_run = g; r.kept_in_last = d + 1;
                    return r;
                }
                break;
            }
            if (code == HX_CG_M) {
                haveM = true; m_g = g; m_len = len; m_lr = lr; m_c = c; m_idx = idx; m_m = mcount;
                other_extra = 0;
                c += len * cstep; mcount += len;
            }
            lr += len * lstep;
        } else {
            if (d == 0) break;
            c += len * cstep;
            if (code == HX_CG_OTHER) other_extra += len;
        }
        idx += len;
    }
    if (!haveM) return r;
    r.ok = true;
    r.lr = m_lr + (m_len - 1) * lstep;
    r.c = m_c + (m_len - 1) * cstep + other_extra * cstep;
    r.kept = m_idx + m_len; r.nmatch = m_m + m_len;
    r.last_run = m_g; r.kept_in_last = m_len;
    return r;
}

// One read, start to end, on ONE lane (the round-1 mapping): kept for the reads with more raw hits than a wavefront has lanes.
__device__ void chain_read_serial(const DevHits& h, const uint64_t* __restrict__ rho, const uint8_t* __restrict__ cls, uint32_t n_contigs,
                                  uint32_t lr_begin, uint32_t r, uint32_t min_block, double min_sim, uint32_t min_mapq,
                                  const ChainScratch& sc, uint32_t* err, const bool prefiltered) {
    const uint64_t raw_b = rho[r], raw_e = rho[r + 1];
    const uint64_t base = raw_b - rho[lr_begin];
    uint32_t* L = sc.hit + base;   // the read's working list of raw-hit indices
    // ---- filters 1-4
    uint32_t n = 0;
    for (uint64_t i = raw_b; i < raw_e; i++) {
        uint32_t tid = h.t_id[i];
        if (tid >= n_contigs) { atomicOr(err, (uint32_t)HXE_BAD_TID); continue; }
        if (prefiltered) { L[n++] = (uint32_t)i; continue; }   // the filtered set of an index.longread: taken as it is (main.cpp:90-116)
        if (h.n_block[i] < min_block) continue;
        if ((double)h.n_match[i] / (double)h.n_block[i] < min_sim) continue;
        if (h.mapq[i] < min_mapq) continue;
        if (cls[tid] & HXC_DROP_LOAD) continue;
        L[n++] = (uint32_t)i;
    }
    // ---- stable insertion sort by (q_end, q_start); ties keep PAF order
    for (uint32_t i = 1; i < n && !prefiltered; i++) {
        uint32_t x = L[i];
        uint32_t xe = h.q_end[x], xs = h.q_start[x];
        uint32_t j = i;
        while (j > 0) {
            uint32_t y = L[j - 1];
            uint32_t ye = h.q_end[y], ys = h.q_start[y];
            if (ye < xe || (ye == xe && ys <= xs)) break;
            L[j] = y;
            j--;
        }
        L[j] = x;
    }
    uint32_t n_aln = 0, n_cmp = 0;
    if (n > (prefiltered ? 0u : 1u)) {
        // ---- palindrome rule: truncate at the second hit of a unique contig
        uint32_t keep = n;
        for (uint32_t i = 0; i < keep && !prefiltered; i++) {
            uint32_t tid = h.t_id[L[i]];
            if (!(cls[tid] & HXC_UNIQUE)) continue;
            for (uint32_t k = 0; k < i; k++)
                if (h.t_id[L[k]] == tid) { keep = i; break; }   // every earlier hit of a unique tid was itself recorded
        }
        // ---- filter 5 (interior hits covering < 0.8 of the contig) + materialise the alignment rows in place
        for (uint32_t i = 0; i < keep; i++) {
            uint32_t x = L[i];
            if (!prefiltered && i > 0 && i + 1 < keep && (h.t_end[x] - h.t_start[x]) / (double)h.t_len[x] < 0.8) continue;
            uint64_t o = base + n_aln;
            sc.hit[o] = x;   // o <= base+i: never overwrites an unread entry of L
            sc.qs[o] = h.q_start[x]; sc.qe[o] = h.q_end[x]; sc.ts[o] = h.t_start[x]; sc.te[o] = h.t_end[x];
            sc.nm[o] = h.n_match[x]; sc.nb[o] = h.n_block[x];
            sc.cb[o] = h.cg_off[x]; sc.ce[o] = h.cg_off[x + 1]; sc.skf[o] = 0; sc.skb[o] = 0;
            n_aln++;
        }
        // ---- overlap trim, left to right
        for (uint32_t i = 0; i + 1 < n_aln; i++) {
            uint64_t a = base + i, b = a + 1;
            if (!(sc.qe[a] > sc.qs[b])) continue;
            long long ov = (long long)sc.qe[a] - (long long)sc.qs[b];
            {
                CgView v{h.cg_ops, sc.cb[a], sc.ce[a], sc.skf[a], sc.skb[a]};
                bool rev = h.is_rev[sc.hit[a]];
                uint32_t target = (uint32_t)((long long)sc.qe[a] - ov / 2 - 1);
                TrimRes t = rev ? trim_walk(v, true, sc.qs[a], sc.te[a] - 1, +1, -1, target)
                                : trim_walk(v, false, sc.qs[a], sc.ts[a], +1, +1, target);
                if (!t.ok) { atomicOr(err, (uint32_t)HXE_TRIM_NO_M); continue; }
                sc.qe[a] = t.lr + 1;
                if (rev) sc.ts[a] = t.c; else sc.te[a] = t.c + 1;
                sc.nb[a] = t.kept; sc.nm[a] = t.nmatch;
                uint32_t raw = HX_CG_LEN(h.cg_ops[t.last_run]);
                if (!rev) { sc.skb[a] = raw - (t.last_run == sc.cb[a] ? sc.skf[a] : 0) - t.kept_in_last; sc.ce[a] = t.last_run + 1; }
                else      { sc.skf[a] = raw - (t.last_run + 1 == sc.ce[a] ? sc.skb[a] : 0) - t.kept_in_last; sc.cb[a] = t.last_run; }
            }
            {
                CgView v{h.cg_ops, sc.cb[b], sc.ce[b], sc.skf[b], sc.skb[b]};
                bool rev = h.is_rev[sc.hit[b]];
                uint32_t target = (uint32_t)((long long)sc.qs[b] + (ov - ov / 2));
                TrimRes t = rev ? trim_walk(v, false, sc.qe[b] - 1, sc.ts[b], -1, +1, target)
                                : trim_walk(v, true, sc.qe[b] - 1, sc.te[b] - 1, -1, -1, target);
                if (!t.ok) { atomicOr(err, (uint32_t)HXE_TRIM_NO_M); continue; }
                sc.qs[b] = t.lr;
                if (rev) sc.te[b] = t.c + 1; else sc.ts[b] = t.c;
                sc.nb[b] = t.kept; sc.nm[b] = t.nmatch;
                uint32_t raw = HX_CG_LEN(h.cg_ops[t.last_run]);
                if (rev) { sc.skb[b] = raw - (t.last_run == sc.cb[b] ? sc.skf[b] : 0) - t.kept_in_last; sc.ce[b] = t.last_run + 1; }
                else     { sc.skf[b] = raw - (t.last_run + 1 == sc.ce[b] ? sc.skb[b] : 0) - t.kept_in_last; sc.cb[b] = t.last_run; }
            }
        }
        // ---- chaining: weighted interval scheduling over the hits that pass :535 and :539
        uint32_t* U = sc.cmp + base;    // candidate list (local alignment indices), later overwritten by the solution
        uint32_t* dp = sc.dp + base;
        int32_t* from = sc.from + base;
        uint32_t nu = 0;
        for (uint32_t i = 0; i < n_aln; i++) {
            uint64_t a = base + i;
            if (sc.nb[a] < min_block) continue;
            if (cls[h.t_id[sc.hit[a]]] & HXC_DROP_CHAIN) continue;
            U[nu++] = i;
        }
        if (nu > 10000) { atomicOr(err, (uint32_t)HXE_CHAIN_TOO_MANY); nu = 0; }
        if (nu > 0) {
            dp[0] = sc.nm[base + U[0]]; from[0] = -1;
            for (uint32_t i = 1; i < nu; i++) {
                int32_t j = -1;
                uint32_t qs_i = sc.qs[base + U[i]];
                for (int32_t k = (int32_t)i - 1; k >= 0; k--) if (sc.qe[base + U[k]] <= qs_i) { j = k; break; }
                uint32_t w = sc.nm[base + U[i]] + (j >= 0 ? dp[j] : 0);
                if (w > dp[i - 1]) { dp[i] = w; from[i] = j; }
                else { dp[i] = dp[i - 1]; from[i] = -2; }
            }
            // walk back; solution size first, then fill from the end (reuses dp[] as the output staging)
            uint32_t cnt = 0;
            for (int32_t i = (int32_t)nu - 1; i >= 0;) { if (from[i] == -2) { i--; continue; } cnt++; i = from[i]; }
            uint32_t w = cnt;
            for (int32_t i = (int32_t)nu - 1; i >= 0;) { if (from[i] == -2) { i--; continue; } dp[--w] = U[i]; i = from[i]; }
            for (uint32_t i = 0; i < cnt; i++) U[i] = dp[i];
            n_cmp = cnt;
        }
    }
    sc.n_aln[r - lr_begin] = n_aln;
    sc.n_cmp[r - lr_begin] = n_cmp;
}

// ---------------------------------------------------------------------------------------------------
// One WAVEFRONT per read (reads with at most 64 raw hits - in practice all of them).
//   * lane i loads raw hit i of the read: 64 consecutive records of every column, coalesced; the filters are one ballot
//   * sort by (q_end, q_start): every surviving lane counts the survivors that precede it (a loop over the set bits of the ballot with
//     v_readlane broadcasts), the rank is its place; ties keep PAF order like the serial insertion sort
//   * palindrome rule and filter 5: ballots over the sorted lanes; rows move to their final places through ds_bpermute
//   * overlap trim: pairs left to right (a hit can be cut on both sides by two successive pairs), but the CIGAR walk of a cut - thousands of
//     run-length ops for a 10 kb alignment, the serial kernel's whole cost - is shared by the 64 lanes: every lane sums a slice of the ops,
//     a wave prefix sum gives each slice its start state, one ballot finds the op where the walk stops (trim_walk_wave)
//   * chaining: the latest compatible predecessor of every candidate in parallel, the O(n) recurrence itself on wave-uniform values
// Rows of the read live in LDS while they change and are written once, coalesced.
// ---------------------------------------------------------------------------------------------------
struct WaveRows {
    uint32_t hit[64], qs[64], qe[64], ts[64], te[64], nm[64], nb[64], skf[64], skb[64], perm[64], cand[64], jprev[64], sol[64];
    uint64_t cb[64], ce[64];
};

__device__ __forceinline__ uint32_t wave_excl_add(uint32_t v, uint32_t lane) {
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(x, d, 64); if ((int)lane >= d) x += o; }
    return x - v;
}

// trim_walk with the ops of the view dealt to the 64 lanes; every lane returns the same result
__device__ TrimRes trim_walk_wave(const CgView& v, bool reversed, uint32_t lr0, uint32_t c0, int lstep, int cstep, uint32_t lr_pos, uint32_t lane) {
    TrimRes r;
    r.ok = false; r.lr = r.c = r.kept = r.nmatch = r.kept_in_last = 0; r.last_run = 0;
    const uint64_t n = v.e - v.b;
    const uint64_t C = (n + 63) / 64, k0 = std::min<uint64_t>(n, (uint64_t)lane * C), k1 = std::min<uint64_t>(n, k0 + C);
    const uint32_t D0 = lstep > 0 ? lr_pos - lr0 : lr0 - lr_pos;   // read bases to go (32-bit like the serial walk: a target behind the start is never reached)
    // pass 1: what the lane's slice adds to the walk's counters
    uint32_t sRead = 0, sM = 0, sIdx = 0, sCon = 0, sOth = 0;
    for (uint64_t k = k0; k < k1; k++) {
        const uint64_t g = reversed ? v.e - 1 - k : v.b + k;
        const uint32_t len = v.eff(g), code = HX_CG_OP(v.ops[g]);
        sIdx += len;
        if (code == HX_CG_M) { sRead += len; sM += len; sCon += len; }
        else if (code == HX_CG_I) sRead += len;
        else { sCon += len; if (code == HX_CG_OTHER) sOth += len; }
    }
    const uint32_t bRead = wave_excl_add(sRead, lane), bM = wave_excl_add(sM, lane), bIdx = wave_excl_add(sIdx, lane), bCon = wave_excl_add(sCon, lane),
                   bOth = wave_excl_add(sOth, lane);
    // pass 2: the first op of the slice at which the serial walk would stop, and the last M run before it
    uint32_t cr = bRead, cm = bM, ci = bIdx, cc = bCon, co = bOth;
    bool stop = false, stopM = false;
    uint32_t st_d = 0, st_read = 0, st_m = 0, st_idx = 0, st_con = 0, st_oth = 0; uint64_t st_g = 0;
    bool haveM = false;
    uint32_t m_read = 0, m_m = 0, m_idx = 0, m_con = 0, m_len = 0, m_othAfter = 0; uint64_t m_g = 0;
    for (uint64_t k = k0; k < k1 && !stop; k++) {
        const uint64_t g = reversed ? v.e - 1 - k : v.b + k;
        const uint32_t len = v.eff(g);
        if (len == 0) continue;
        const uint32_t code = HX_CG_OP(v.ops[g]);
        const uint32_t d = D0 - cr;
        if (code == HX_CG_M || code == HX_CG_I) {
            if (d < len) { stop = true; stopM = code == HX_CG_M; st_d = d; st_read = cr; st_m = cm; st_idx = ci; st_con = cc; st_oth = co; st_g = g; break; }
            if (code == HX_CG_M) { haveM = true; m_read = cr; m_m = cm; m_idx = ci; m_con = cc; m_len = len; m_g = g; m_othAfter = co; cm += len; cc += len; }
            cr += len;
        } else {
            if (d == 0) { stop = true; stopM = false; st_d = 0; st_read = cr; st_m = cm; st_idx = ci; st_con = cc; st_oth = co; st_g = g; break; }
            cc += len;
            if (code == HX_CG_OTHER) co += len;
        }
        ci += len;
    }
    const unsigned long long stopMask = __ballot(stop);
    const int fl = stopMask ? __builtin_ctzll(stopMask) : 64;                 // the lane whose slice holds the stop (64: the ops ran out)
    if (fl < 64 && __shfl((int)stopM, fl, 64)) {                              // stopped inside an M run: the cut
        const uint32_t d = __shfl(st_d, fl, 64), rd = __shfl(st_read, fl, 64), mm = __shfl(st_m, fl, 64), ix = __shfl(st_idx, fl, 64), cn = __shfl(st_con, fl, 64);
        const uint32_t glo = __shfl((uint32_t)st_g, fl, 64), ghi = __shfl((uint32_t)(st_g >> 32), fl, 64);
        r.ok = true;
        r.lr = lr0 + (rd + d) * (uint32_t)lstep; r.c = c0 + (cn + d) * (uint32_t)cstep;
        r.kept = ix + d + 1; r.nmatch = mm + d + 1;
        r.last_run = (uint64_t)glo | ((uint64_t)ghi << 32); r.kept_in_last = d + 1;
        return r;
    }
    // not on an M: back to the last M run before the stop (lanes beyond the stopping one do not count)
    const unsigned long long mMask = __ballot(haveM && (int)lane <= fl);
    if (!mMask) return r;
    const int ml = 63 - __builtin_clzll(mMask);
    const uint32_t othStop = fl < 64 ? __shfl(st_oth, fl, 64) : __shfl(bOth + sOth, 63, 64);   // OTHER bases before the stop
    const uint32_t rd = __shfl(m_read, ml, 64), mm = __shfl(m_m, ml, 64), ix = __shfl(m_idx, ml, 64), cn = __shfl(m_con, ml, 64), ln = __shfl(m_len, ml, 64),
                   oa = __shfl(m_othAfter, ml, 64);
    const uint32_t glo = __shfl((uint32_t)m_g, ml, 64), ghi = __shfl((uint32_t)(m_g >> 32), ml, 64);
    const uint32_t other_extra = othStop - oa;
    r.ok = true;
    r.lr = lr0 + (rd + ln - 1) * (uint32_t)lstep;
    r.c = c0 + (cn + ln - 1) * (uint32_t)cstep + other_extra * (uint32_t)cstep;
    r.kept = ix + ln; r.nmatch = mm + ln;
    r.last_run = (uint64_t)glo | ((uint64_t)ghi << 32); r.kept_in_last = ln;
    return r;
}

__global__ void __launch_bounds__(256) k_chain_reads_wave(DevHits h, const uint64_t* __restrict__ rho, const uint8_t* __restrict__ cls, uint32_t n_contigs,
                                                           uint32_t lr_begin, uint32_t lr_end, uint32_t min_block, double min_sim, uint32_t min_mapq,
                                                           ChainScratch sc, uint32_t* err, const bool prefiltered) {
    __shared__ WaveRows rows_all[4];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t r = lr_begin + blockIdx.x * 4u + wv;
    if (r >= lr_end) return;
    const uint64_t raw_b = rho[r], raw_e = rho[r + 1];
    const uint32_t nraw = (uint32_t)(raw_e - raw_b);
    if (raw_e - raw_b > 64) {   // more hits than lanes: the one-lane path
        if (lane == 0) chain_read_serial(h, rho, cls, n_contigs, lr_begin, r, min_block, min_sim, min_mapq, sc, err, prefiltered);
        return;
    }
    WaveRows& R = rows_all[wv];
    const uint64_t base = raw_b - rho[lr_begin];
    // ---- lane i = raw hit i: filters 1-4
    const uint64_t x = raw_b + lane;
    uint32_t tid = 0, qs = 0, qe = 0, ts = 0, te = 0, nm = 0, nb = 0, tl = 1;
    bool pass = false;
    if (lane < nraw) {
        tid = h.t_id[x]; qs = h.q_start[x]; qe = h.q_end[x]; ts = h.t_start[x]; te = h.t_end[x]; nm = h.n_match[x]; nb = h.n_block[x]; tl = h.t_len[x];
        if (tid >= n_contigs) atomicOr(err, (uint32_t)HXE_BAD_TID);
        else pass = prefiltered || (nb >= min_block && !((double)nm / (double)nb < min_sim) && h.mapq[x] >= min_mapq && !(cls[tid] & HXC_DROP_LOAD));
    }
    const unsigned long long mpass = __ballot(pass);
    uint32_t n = (uint32_t)__popcll(mpass);
    uint32_t n_aln = 0, n_cmp = 0;
    if (n > (prefiltered ? 0u : 1u)) {
        // ---- rank among the survivors by (q_end, q_start), ties in PAF order (= lane order); an index.longread is in that order already
        uint32_t rank = (uint32_t)__popcll(mpass & ((1ull << lane) - 1ull));
        if (!prefiltered) {
            rank = 0;
            for (unsigned long long m = mpass; m; m &= m - 1) {
                const int j = __builtin_ctzll(m);
                const uint32_t je = (uint32_t)__builtin_amdgcn_readlane((int)qe, j), js = (uint32_t)__builtin_amdgcn_readlane((int)qs, j);
                rank += (je < qe || (je == qe && (js < qs || (js == qs && (uint32_t)j < lane)))) ? 1u : 0u;
            }
        }
        if (pass) R.perm[rank] = lane;
        __builtin_amdgcn_wave_barrier();
        // sorted lane p takes the row of the lane that ranked p
        const int src = lane < n ? (int)R.perm[lane] : 0;
        const uint32_t hx = (uint32_t)raw_b + (uint32_t)src;   // (low word; the row keeps the full raw index below)
        tid = __shfl(tid, src, 64); qs = __shfl(qs, src, 64); qe = __shfl(qe, src, 64); ts = __shfl(ts, src, 64); te = __shfl(te, src, 64);
        nm = __shfl(nm, src, 64); nb = __shfl(nb, src, 64); tl = __shfl(tl, src, 64);
        (void)hx;
        const bool in = lane < n;
        // ---- palindrome rule: cut at the second hit of a unique contig
        uint32_t keep = n;
        if (!prefiltered) {
            const bool uniq = in && (cls[tid] & HXC_UNIQUE);
            bool dup = false;
            for (uint32_t k = 0; k + 1 < n; k++) { const uint32_t tk = (uint32_t)__builtin_amdgcn_readlane((int)tid, (int)k); dup = dup || (uniq && lane > k && tk == tid); }
            const unsigned long long dm = __ballot(dup);
            if (dm) keep = (uint32_t)__builtin_ctzll(dm);
        }
        // ---- filter 5: interior hits that cover less than 0.8 of their contig
        const bool kept_row = lane < keep && !(!prefiltered && lane > 0 && lane + 1 < keep && (te - ts) / (double)tl < 0.8);
        const unsigned long long km = __ballot(kept_row);
        n_aln = (uint32_t)__popcll(km);
        if (kept_row) {
            const uint32_t o = (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
            const uint64_t raw = raw_b + (uint32_t)src;
            R.hit[o] = (uint32_t)raw; R.qs[o] = qs; R.qe[o] = qe; R.ts[o] = ts; R.te[o] = te; R.nm[o] = nm; R.nb[o] = nb;
            R.cb[o] = h.cg_off[raw]; R.ce[o] = h.cg_off[raw + 1]; R.skf[o] = 0; R.skb[o] = 0;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- overlap trim, left to right: both walks of a pair on all lanes, the row updates on lane 0
        for (uint32_t i = 0; i + 1 < n_aln; i++) {
            const uint32_t a = i, b = i + 1;
            const uint32_t qe_a = R.qe[a], qs_b = R.qs[b];
            if (!(qe_a > qs_b)) continue;
            const long long ov = (long long)qe_a - (long long)qs_b;
            {
                const CgView v{h.cg_ops, R.cb[a], R.ce[a], R.skf[a], R.skb[a]};
                const bool rev = h.is_rev[R.hit[a]];
                const uint32_t target = (uint32_t)((long long)qe_a - ov / 2 - 1);
                const TrimRes t = rev ? trim_walk_wave(v, true, R.qs[a], R.te[a] - 1, +1, -1, target, lane) : trim_walk_wave(v, false, R.qs[a], R.ts[a], +1, +1, target, lane);
                if (!t.ok) { if (lane == 0) atomicOr(err, (uint32_t)HXE_TRIM_NO_M); continue; }
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {
                    R.qe[a] = t.lr + 1;
                    if (rev) R.ts[a] = t.c; else R.te[a] = t.c + 1;
                    R.nb[a] = t.kept; R.nm[a] = t.nmatch;
                    const uint32_t raw = HX_CG_LEN(h.cg_ops[t.last_run]);
                    if (!rev) { R.skb[a] = raw - (t.last_run == R.cb[a] ? R.skf[a] : 0) - t.kept_in_last; R.ce[a] = t.last_run + 1; }
                    else      { R.skf[a] = raw - (t.last_run + 1 == R.ce[a] ? R.skb[a] : 0) - t.kept_in_last; R.cb[a] = t.last_run; }
                }
                __builtin_amdgcn_wave_barrier();
            }
            {
                const CgView v{h.cg_ops, R.cb[b], R.ce[b], R.skf[b], R.skb[b]};
                const bool rev = h.is_rev[R.hit[b]];
                const uint32_t target = (uint32_t)((long long)qs_b + (ov - ov / 2));
                const TrimRes t = rev ? trim_walk_wave(v, false, R.qe[b] - 1, R.ts[b], -1, +1, target, lane) : trim_walk_wave(v, true, R.qe[b] - 1, R.te[b] - 1, -1, -1, target, lane);
                if (!t.ok) { if (lane == 0) atomicOr(err, (uint32_t)HXE_TRIM_NO_M); continue; }
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {
                    R.qs[b] = t.lr;
                    if (rev) R.te[b] = t.c + 1; else R.ts[b] = t.c;
                    R.nb[b] = t.kept; R.nm[b] = t.nmatch;
                    const uint32_t raw = HX_CG_LEN(h.cg_ops[t.last_run]);
                    if (rev) { R.skb[b] = raw - (t.last_run == R.cb[b] ? R.skf[b] : 0) - t.kept_in_last; R.ce[b] = t.last_run + 1; }
                    else     { R.skf[b] = raw - (t.last_run + 1 == R.ce[b] ? R.skb[b] : 0) - t.kept_in_last; R.cb[b] = t.last_run; }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- chaining: weighted interval scheduling over the hits that pass Longread.cpp:535 and :539
        const bool is_c = lane < n_aln && R.nb[lane] >= min_block && !(cls[h.t_id[R.hit[lane]]] & HXC_DROP_CHAIN);
        const unsigned long long cmask = __ballot(is_c);
        const uint32_t nu = (uint32_t)__popcll(cmask);
        if (is_c) R.cand[__popcll(cmask & ((1ull << lane) - 1ull))] = lane;
        __builtin_amdgcn_wave_barrier();
        if (nu > 0) {
            // candidate u = lane u: its row, and the latest candidate k < u that ends where u starts or before
            const uint32_t row = lane < nu ? R.cand[lane] : 0;
            const uint32_t uqs = R.qs[row], uqe = R.qe[row], unm = R.nm[row];
            int32_t jp = -1;
            for (uint32_t k = 0; k + 1 < nu; k++) { const uint32_t ke = (uint32_t)__builtin_amdgcn_readlane((int)uqe, (int)k); if (lane > k && ke <= uqs) jp = (int32_t)k; }
            // the recurrence, on wave-uniform values: take u iff its weight plus the best before its predecessor beats the best without it (strict)
            uint32_t dpv = 0;     // lane u: dp[u]
            uint32_t take = 0;    // lane u: 1 = u is in the solution that ends at or before u
            uint32_t best = 0;
            for (uint32_t u = 0; u < nu; u++) {
                const int32_t j = __builtin_amdgcn_readlane(jp, (int)u);
                const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)unm, (int)u) + (j >= 0 ? (uint32_t)__builtin_amdgcn_readlane((int)dpv, j) : 0u);
                const bool tk = u == 0 || w > best;
                if (tk) best = w;
                if (lane == u) { dpv = best; take = tk ? 1u : 0u; }
            }
            // walk back from the last candidate
            uint32_t cnt = 0;
            for (int32_t u = (int32_t)nu - 1; u >= 0;) {
                if (!__builtin_amdgcn_readlane((int)take, u)) { u--; continue; }
                const uint32_t row_u = (uint32_t)__builtin_amdgcn_readlane((int)row, u);   // (cross-lane reads stay outside divergent blocks)
                if (lane == 0) R.sol[cnt] = row_u;
                cnt++;
                u = __builtin_amdgcn_readlane(jp, u);
            }
            n_cmp = cnt;
            __builtin_amdgcn_wave_barrier();
            if (lane < cnt) sc.cmp[base + lane] = R.sol[cnt - 1 - lane];   // (found last to first)
        }
        // ---- the rows, once, coalesced
        if (lane < n_aln) {
            const uint64_t o = base + lane;
            sc.hit[o] = R.hit[lane]; sc.qs[o] = R.qs[lane]; sc.qe[o] = R.qe[lane]; sc.ts[o] = R.ts[lane]; sc.te[o] = R.te[lane];
            sc.nm[o] = R.nm[lane]; sc.nb[o] = R.nb[lane]; sc.cb[o] = R.cb[lane]; sc.ce[o] = R.ce[lane]; sc.skf[o] = R.skf[lane]; sc.skb[o] = R.skb[lane];
        }
    }
    if (lane == 0) { sc.n_aln[r - lr_begin] = n_aln; sc.n_cmp[r - lr_begin] = n_cmp; }
}

__global__ void k_chain_compact(ChainScratch sc, const uint64_t* __restrict__ rho, uint32_t lr_begin, uint32_t lr_end,
                                const uint64_t* __restrict__ aln_off, const uint64_t* __restrict__ cmp_off, ChainFinal out) {
    uint32_t r = lr_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= lr_end) return;
    uint64_t base = rho[r] - rho[lr_begin];
    uint64_t ao = aln_off[r - lr_begin], na = aln_off[r - lr_begin + 1] - ao;
    for (uint64_t i = 0; i < na; i++) {
        uint64_t s = base + i, d = ao + i;
        out.hit[d] = sc.hit[s]; out.qs[d] = sc.qs[s]; out.qe[d] = sc.qe[s]; out.ts[d] = sc.ts[s]; out.te[d] = sc.te[s];
        out.nm[d] = sc.nm[s]; out.nb[d] = sc.nb[s]; out.cb[d] = sc.cb[s]; out.ce[d] = sc.ce[s]; out.skf[d] = sc.skf[s]; out.skb[d] = sc.skb[s];
    }
    uint64_t co = cmp_off[r - lr_begin], nc = cmp_off[r - lr_begin + 1] - co;
    for (uint64_t i = 0; i < nc; i++) out.cmp_aln[co + i] = (uint32_t)(ao + sc.cmp[base + i]);
}

}  // namespace

void chain_reads(const DevHits& h, const uint64_t* rho, const uint8_t* cls, uint32_t n_contigs, uint32_t lr_begin, uint32_t lr_end,
                 uint32_t min_aln_block, double min_aln_sim, uint32_t min_mapq, const ChainScratch& sc, uint32_t* err, bool prefiltered, hipStream_t s) {
    uint32_t n = lr_end - lr_begin;
    if (n) k_chain_reads_wave<<<(n + 3) / 4, 256, 0, s>>>(h, rho, cls, n_contigs, lr_begin, lr_end, min_aln_block, min_aln_sim, min_mapq, sc, err, prefiltered);
}

void chain_compact(const ChainScratch& sc, const uint64_t* rho, uint32_t lr_begin, uint32_t lr_end, const uint64_t* aln_off,
                   const uint64_t* cmp_off, const ChainFinal& out, hipStream_t s) {
    uint32_t n = lr_end - lr_begin;
    if (n) k_chain_compact<<<(n + 63) / 64, 64, 0, s>>>(sc, rho, lr_begin, lr_end, aln_off, cmp_off, out);
}

}  // namespace hxk
