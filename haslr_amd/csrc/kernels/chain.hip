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

__global__ void k_chain_reads(DevHits h, const uint64_t* __restrict__ rho, const uint8_t* __restrict__ cls, uint32_t n_contigs,
                              uint32_t lr_begin, uint32_t lr_end, uint32_t min_block, double min_sim, uint32_t min_mapq,
                              ChainScratch sc, uint32_t* err, uint32_t spread, const bool prefiltered) {
    // one lane per read; with few reads only every `spread`-th lane works, so that the reads are spread over more wavefronts (a wave takes
    // as long as its slowest read, and 13 k reads on 64 per wave would leave four fifths of the SIMDs idle)
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    if (gt % spread) return;
    uint32_t r = lr_begin + gt / spread;
    if (r >= lr_end) return;
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
    const uint32_t spread = n <= 16384 ? 4 : n <= 65536 ? 2 : 1;
    if (n) k_chain_reads<<<(uint32_t)(((uint64_t)n * spread + 63) / 64), 64, 0, s>>>(h, rho, cls, n_contigs, lr_begin, lr_end, min_aln_block, min_aln_sim, min_mapq, sc, err, spread, prefiltered);
}

void chain_compact(const ChainScratch& sc, const uint64_t* rho, uint32_t lr_begin, uint32_t lr_end, const uint64_t* aln_off,
                   const uint64_t* cmp_off, const ChainFinal& out, hipStream_t s) {
    uint32_t n = lr_end - lr_begin;
    if (n) k_chain_compact<<<(n + 63) / 64, 64, 0, s>>>(sc, rho, lr_begin, lr_end, aln_off, cmp_off, out);
}

}  // namespace hxk
