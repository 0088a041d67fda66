// oracle.cpp — TEST INFRASTRUCTURE: plain, scalar CPU restatement of the reference hot path.
// See oracle.h for who may use it and for the pinning status of each function.
// Every function cites the reference lines it follows (paths under
// /root/reference/src/haslr_assemble/src/). No reference text is copied: the reference works on
// per-base expanded CIGAR strings and pointer arenas; this works on run-length op words and SoA
// index arrays, and is checked against the compiled reference by tests/test_oracle_vs_ref.py.
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <thread>
#include <vector>
#include <map>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return -1; }

template <class T> T* dup(const std::vector<T>& v) {
    T* p = (T*)malloc(std::max<size_t>(1, v.size()) * sizeof(T));
    if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

// ------------------------------------------------------------------ CIGAR views
struct CgView {
    const uint32_t* ops;
    uint64_t b, e;
    uint32_t skf, skb;
    uint32_t eff(uint64_t k) const {
        uint32_t l = HX_CG_LEN(ops[k]);
        if (k == b) l -= skf;
        if (k + 1 == e) l -= skb;
        return l;
    }
};

struct TrimRes {
    bool ok = false;
    uint32_t lr = 0, c = 0, kept = 0, nmatch = 0;
    uint64_t last_run = 0;
    uint32_t kept_in_last = 0;
};

// find_contig_pos (Longread.cpp:375-420) on run-length ops. Walk the (optionally reversed) CIGAR from
// (lr,c); stop BEFORE consuming the first per-base op at which lr == lr_pos (:382); M moves both, I the
// read, anything else the contig (:384-396). If that op is not M, undo ops backwards until the op at the
// cursor is an M (:399-415; only literal M/I/D are undone). Ops [0..cursor] are kept (:417-418).
TrimRes trim_walk(const CgView& v, bool reversed, uint32_t lr, uint32_t c, int lstep, int cstep, uint32_t lr_pos) {
    TrimRes r;
    struct { bool have = false; uint64_t g = 0; uint32_t len = 0, lr_s = 0, c_s = 0, idx_s = 0, m_s = 0; } lastM;
    uint32_t idx = 0, mcount = 0, other_extra = 0;
    const uint64_t n = v.e - v.b;
    bool broke_on_M = false;
    for (uint64_t k = 0; k < n; k++) {
        uint64_t g = reversed ? v.e - 1 - k : v.b + k;
        uint32_t len = v.eff(g);
        if (len == 0) continue;
        uint32_t code = HX_CG_OP(v.ops[g]);
        uint32_t d = lstep > 0 ? lr_pos - lr : lr - lr_pos;  // per-base steps until lr == lr_pos (mod 2^32)
        if (code == HX_CG_M || code == HX_CG_I) {
            if (d < len) {  // stop inside this run, d ops of it consumed
                if (code == HX_CG_M) {
                    r.ok = true;
                    r.lr = lr + d * lstep; r.c = c + d * cstep;
                    r.kept = idx + d + 1; r.nmatch = mcount + d + 1;
                    r.last_run = g; r.kept_in_last = d + 1;
                    broke_on_M = true;
                }
                break;  // I run: everything consumed of it is undone again by the back-off
            }
            if (code == HX_CG_M) {
                lastM.have = true; lastM.g = g; lastM.len = len; lastM.lr_s = lr; lastM.c_s = c; lastM.idx_s = idx; lastM.m_s = mcount;
                other_extra = 0;
                c += len * cstep; mcount += len;
            }
            lr += len * lstep;
        } else {
            if (d == 0) break;  // lr is constant over a contig-only run: stop at its first op or not at all
            c += len * cstep;
            if (code == HX_CG_OTHER) other_extra += len;  // never undone by the reference (:401-413)
        }
        idx += len;
    }
    if (broke_on_M) return r;
    // stopped on a non-M op, or ran off the end (the reference then reads the NUL terminator, :399)
    if (!lastM.have) return r;  // reference would index before the string: undefined there, an error here
    r.ok = true;
    r.lr = lastM.lr_s + (lastM.len - 1) * lstep;
    r.c = lastM.c_s + (lastM.len - 1) * cstep + other_extra * cstep;
    r.kept = lastM.idx_s + lastM.len; r.nmatch = lastM.m_s + lastM.len;
    r.last_run = lastM.g; r.kept_in_last = lastM.len;
    return r;
}

// asm_find_lr_pos (Assemble.cpp:129-155): returns -1 only if the walk starts beyond the target (:132-133);
// stops before the first op at which c == contig_pos (:137); exhausting the CIGAR is not a failure.
long long find_lr_pos(const CgView& v, bool reversed, uint32_t lr, uint32_t c, int lstep, int cstep, uint32_t contig_pos) {
    if ((cstep > 0 && c > contig_pos) || (cstep < 0 && c < contig_pos)) return -1;
    const uint64_t n = v.e - v.b;
    for (uint64_t k = 0; k < n; k++) {
        uint64_t g = reversed ? v.e - 1 - k : v.b + k;
        uint32_t len = v.eff(g);
        if (len == 0) continue;
        uint32_t code = HX_CG_OP(v.ops[g]);
        uint32_t d = cstep > 0 ? contig_pos - c : c - contig_pos;
        if (code == HX_CG_I) {
            if (d == 0) break;
            lr += len * lstep;
        } else {
            if (d < len) {
                if (code == HX_CG_M) lr += d * lstep;
                c += d * cstep;
                break;
            }
            if (code == HX_CG_M) lr += len * lstep;
            c += len * cstep;
        }
    }
    return (long long)lr;
}

struct Aln {
    uint32_t hit, qs, qe, ts, te, nm, nb;
    uint64_t cb, ce;
    uint32_t skf, skb;
};

}  // namespace

extern "C" const char* orc_last_error(void) { return g_err.c_str(); }

// =====================================================================================================
// a1-a5: filters, per-read sort, palindrome rule, filter 5, overlap trim, chaining
// =====================================================================================================
extern "C" int orc_chain_reads_ex(const hx_contigs* ctg, const hx_hits* h, const uint64_t* read_hit_off,
                                  uint32_t n_reads, const hx_params* prm, int prefiltered, hx_chain_out* out);
extern "C" int orc_chain_reads(const hx_contigs* ctg, const hx_hits* h, const uint64_t* read_hit_off,
                               uint32_t n_reads, const hx_params* prm, hx_chain_out* out) {
    return orc_chain_reads_ex(ctg, h, read_hit_off, n_reads, prm, 0, out);
}
// prefiltered: the records are the reference's filtered set read back from index.longread (Longread.cpp:341-372): main() goes straight to
// fix_alignments with them (main.cpp:90-116) - no filters, no sort, no group rule
extern "C" int orc_chain_reads_ex(const hx_contigs* ctg, const hx_hits* h, const uint64_t* read_hit_off,
                                  uint32_t n_reads, const hx_params* prm, int prefiltered, hx_chain_out* out) {
    memset(out, 0, sizeof(*out));
    const double thr_load = prm->uniq_freq * (3 + prm->max_uniq_dev);   // Longread.cpp:272
    const double thr_uniq = prm->uniq_freq * (1 + prm->max_uniq_dev);   // Longread.cpp:191, :539 (copy_count = 1)
    std::vector<Aln> alns;
    std::vector<uint64_t> read_off(n_reads + 1, 0), cmp_off(n_reads + 1, 0);
    std::vector<uint32_t> cmp_aln;
    std::vector<uint32_t> grp, seen_tid;

    for (uint32_t r = 0; r < n_reads; r++) {
        read_off[r] = alns.size();
        cmp_off[r] = cmp_aln.size();
        // ---- filters 1-4 (Longread.cpp:262-272) on raw PAF records of this read
        grp.clear();
        for (uint64_t i = read_hit_off[r]; i < read_hit_off[r + 1]; i++) {
            if (h->q_id[i] != r) return fail("orc_chain_reads: PAF not grouped by ascending query id");
            if (h->t_id[i] >= ctg->n) return fail("orc_chain_reads: contig id out of range");
            if (prefiltered) { grp.push_back((uint32_t)i); continue; }
            if (h->n_block[i] < prm->min_aln_block) continue;
            if ((double)h->n_match[i] / (double)h->n_block[i] < prm->min_aln_sim) continue;
            if (h->mapq[i] < prm->min_aln_mapq) continue;
            if (ctg->mean_kmer[h->t_id[i]] > thr_load) continue;
            grp.push_back((uint32_t)i);
        }
        // ---- sort by (q_end, q_start) (Longread.cpp:52-55,256). The reference's std::sort leaves ties on
        // both keys in implementation order; the restatement fixes them to PAF order (stable).
        if (!prefiltered) std::stable_sort(grp.begin(), grp.end(), [&](uint32_t a, uint32_t b) {
            return h->q_end[a] < h->q_end[b] || (h->q_end[a] == h->q_end[b] && h->q_start[a] < h->q_start[b]);
        });
        if (!prefiltered && grp.size() <= 1) continue;  // Longread.cpp:184
        // ---- palindrome rule (Longread.cpp:187-202): truncate at the second hit of a unique contig
        seen_tid.clear();
        size_t keep = grp.size();
        for (size_t i = 0; i < keep && !prefiltered; i++) {
            uint32_t tid = h->t_id[grp[i]];
            if (ctg->mean_kmer[tid] < thr_uniq) {
                if (std::find(seen_tid.begin(), seen_tid.end(), tid) != seen_tid.end()) keep = i;
                else seen_tid.push_back(tid);
            }
        }
        grp.resize(keep);
        // ---- filter 5 (Longread.cpp:207) + append (:216-230)
        for (size_t i = 0; i < grp.size(); i++) {
            uint32_t x = grp[i];
            if (!prefiltered && i > 0 && i + 1 < grp.size() && (h->t_end[x] - h->t_start[x]) / (double)h->t_len[x] < 0.8) continue;
            alns.push_back({x, h->q_start[x], h->q_end[x], h->t_start[x], h->t_end[x], h->n_match[x], h->n_block[x],
                            h->cg_off[x], h->cg_off[x + 1], 0, 0});
        }
        // ---- overlap trim (fix_overlapping_alignments, Longread.cpp:430-512), left to right
        Aln* a = alns.data() + read_off[r];
        int num = (int)(alns.size() - read_off[r]);
        for (int i = 0; i + 1 < num; i++) {
            if (!(a[i].qe > a[i + 1].qs)) continue;
            long long ov = (long long)a[i].qe - (long long)a[i + 1].qs;
            {   // first alignment: cut ov/2 off its read-space tail (:445-473)
                Aln& x = a[i];
                CgView v{h->cg_ops, x.cb, x.ce, x.skf, x.skb};
                bool rev = h->is_rev[x.hit];
                uint32_t target = (uint32_t)((long long)x.qe - ov / 2 - 1);
                TrimRes t = rev ? trim_walk(v, true, x.qs, x.te - 1, +1, -1, target)
                                : trim_walk(v, false, x.qs, x.ts, +1, +1, target);
                if (!t.ok) return fail("orc_chain_reads: overlap trim ran off an alignment without M");
                x.qe = t.lr + 1;
                if (rev) x.ts = t.c; else x.te = t.c + 1;
                x.nb = t.kept; x.nm = t.nmatch;
                uint32_t raw = HX_CG_LEN(h->cg_ops[t.last_run]);
                if (!rev) { x.skb = raw - (t.last_run == x.cb ? x.skf : 0) - t.kept_in_last; x.ce = t.last_run + 1; }
                else      { x.skf = raw - (t.last_run + 1 == x.ce ? x.skb : 0) - t.kept_in_last; x.cb = t.last_run; }
            }
            {   // second alignment: cut ov - ov/2 off its read-space head (:478-505)
                Aln& y = a[i + 1];
                CgView v{h->cg_ops, y.cb, y.ce, y.skf, y.skb};
                bool rev = h->is_rev[y.hit];
                uint32_t target = (uint32_t)((long long)y.qs + (ov - ov / 2));
                TrimRes t = rev ? trim_walk(v, false, y.qe - 1, y.ts, -1, +1, target)
                                : trim_walk(v, true, y.qe - 1, y.te - 1, -1, -1, target);
                if (!t.ok) return fail("orc_chain_reads: overlap trim ran off an alignment without M");
                y.qs = t.lr;
                if (rev) y.te = t.c + 1; else y.ts = t.c;
                y.nb = t.kept; y.nm = t.nmatch;
                uint32_t raw = HX_CG_LEN(h->cg_ops[t.last_run]);
                if (rev) { y.skb = raw - (t.last_run == y.cb ? y.skf : 0) - t.kept_in_last; y.ce = t.last_run + 1; }
                else     { y.skf = raw - (t.last_run + 1 == y.ce ? y.skb : 0) - t.kept_in_last; y.cb = t.last_run; }
            }
        }
        // ---- chaining (find_best_scheduling, Longread.cpp:524-610): weighted interval scheduling
        std::vector<int> u;  // indices (within the read) that pass the two filters (:535,:539)
        for (int i = 0; i < num; i++) {
            if (a[i].nb < prm->min_aln_block) continue;
            if (ctg->mean_kmer[h->t_id[a[i].hit]] > thr_uniq) continue;
            u.push_back(i);
        }
        if (u.empty()) continue;
        if (u.size() > 10000) return fail("orc_chain_reads: more than 10000 chainable hits on one read (reference limit, Longread.cpp:529)");
        int n = (int)u.size();
        std::vector<uint32_t> dp(n);
        std::vector<int> from(n);  // >=0: took i with predecessor solution `from`; -1: took i alone; -2: inherited i-1
        dp[0] = a[u[0]].nm; from[0] = -1;
        for (int i = 1; i < n; i++) {
            int j = -1;
            for (int k = i - 1; k >= 0; k--) if (a[u[k]].qe <= a[u[i]].qs) { j = k; break; }  // latest_compatible :514-522
            uint32_t w = a[u[i]].nm + (j >= 0 ? dp[j] : 0);
            if (w > dp[i - 1]) { dp[i] = w; from[i] = j >= 0 ? j : -1; }   // strict > (:576,:590)
            else { dp[i] = dp[i - 1]; from[i] = -2; }
        }
        std::vector<int> sol;
        for (int i = n - 1; i >= 0;) {
            if (from[i] == -2) { i--; continue; }
            sol.push_back(i);
            i = from[i] >= 0 ? from[i] : -1;
        }
        for (auto it = sol.rbegin(); it != sol.rend(); ++it) cmp_aln.push_back((uint32_t)(read_off[r] + u[*it]));
    }
    read_off[n_reads] = alns.size();
    cmp_off[n_reads] = cmp_aln.size();

    out->n_aln = alns.size(); out->n_reads = n_reads; out->n_cmp = cmp_aln.size();
    size_t n = std::max<size_t>(1, alns.size());
#define COL(T, name, expr) { out->name = (T*)malloc(n * sizeof(T)); for (size_t i = 0; i < alns.size(); i++) out->name[i] = alns[i].expr; }
    COL(uint32_t, hit, hit) COL(uint32_t, q_start, qs) COL(uint32_t, q_end, qe) COL(uint32_t, t_start, ts) COL(uint32_t, t_end, te)
    COL(uint32_t, n_match, nm) COL(uint32_t, n_block, nb) COL(uint64_t, cg_begin, cb) COL(uint64_t, cg_end, ce)
    COL(uint32_t, cg_skip_front, skf) COL(uint32_t, cg_skip_back, skb)
#undef COL
    out->read_off = dup(read_off); out->cmp_off = dup(cmp_off); out->cmp_aln = dup(cmp_aln);
    return 0;
}

extern "C" void orc_free_chain(hx_chain_out* o) {
    free(o->hit); free(o->q_start); free(o->q_end); free(o->t_start); free(o->t_end); free(o->n_match); free(o->n_block);
    free(o->cg_begin); free(o->cg_end); free(o->cg_skip_front); free(o->cg_skip_back); free(o->read_off); free(o->cmp_off); free(o->cmp_aln);
    memset(o, 0, sizeof(*o));
}

// =====================================================================================================
// a6: edge-support multiset (bbg_build_graph / bbg_add_edge, Backbone_graph.cpp:148-171, :10-25)
// =====================================================================================================
namespace {
struct Rec {
    uint64_t key;
    uint32_t lr, ch, ct;
    uint32_t ah, at;  // global alignment index of head / tail
};
void alloc_side(hx_rec_side& s, size_t n) {
    n = std::max<size_t>(1, n);
    s.q_start = (uint32_t*)malloc(n * 4); s.q_end = (uint32_t*)malloc(n * 4); s.t_start = (uint32_t*)malloc(n * 4); s.t_end = (uint32_t*)malloc(n * 4);
    s.is_rev = (uint8_t*)malloc(n); s.cg_begin = (uint64_t*)malloc(n * 8); s.cg_end = (uint64_t*)malloc(n * 8);
    s.cg_skip_front = (uint32_t*)malloc(n * 4); s.cg_skip_back = (uint32_t*)malloc(n * 4);
}
void fill_side(hx_rec_side& s, size_t i, const hx_chain_out* c, const hx_hits* h, uint32_t a) {
    s.q_start[i] = c->q_start[a]; s.q_end[i] = c->q_end[a]; s.t_start[i] = c->t_start[a]; s.t_end[i] = c->t_end[a];
    s.is_rev[i] = h->is_rev[c->hit[a]]; s.cg_begin[i] = c->cg_begin[a]; s.cg_end[i] = c->cg_end[a];
    s.cg_skip_front[i] = c->cg_skip_front[a]; s.cg_skip_back[i] = c->cg_skip_back[a];
}
void free_side(hx_rec_side& s) {
    free(s.q_start); free(s.q_end); free(s.t_start); free(s.t_end); free(s.is_rev); free(s.cg_begin); free(s.cg_end); free(s.cg_skip_front); free(s.cg_skip_back);
}
}  // namespace

extern "C" int orc_edge_support(const hx_contigs* ctg, const hx_hits* h, const hx_params* prm, const hx_chain_out* c,
                                uint32_t lr_begin, uint32_t lr_end, hx_edges_out* out) {
    memset(out, 0, sizeof(*out));
    const double thr_edge = prm->uniq_freq * (1 + prm->max_uniq_dev);  // Backbone_graph.cpp:160 (<=)
    std::vector<Rec> recs;
    std::vector<uint32_t> sel;
    for (uint32_t r = lr_begin; r < lr_end; r++) {
        uint64_t b = c->cmp_off[r], e = c->cmp_off[r + 1];
        if (e - b <= 1) continue;  // :153
        sel.clear();
        for (uint64_t j = b; j < e; j++)
            if (ctg->mean_kmer[h->t_id[c->hit[c->cmp_aln[j]]]] <= thr_edge) sel.push_back((uint32_t)(j - b));
        for (size_t k = 0; k + 1 < sel.size(); k++) {
            uint32_t i1 = sel[k], i2 = sel[k + 1];
            uint32_t a1 = c->cmp_aln[b + i1], a2 = c->cmp_aln[b + i2];
            uint32_t n1 = h->t_id[c->hit[a1]], r1 = h->is_rev[c->hit[a1]];
            uint32_t n2 = h->t_id[c->hit[a2]], r2 = h->is_rev[c->hit[a2]];
            uint64_t kf = ((uint64_t)((n1 << 1) | r1) << 32) | ((n2 << 1) | r2);
            uint64_t kt = ((uint64_t)((n2 << 1) | (1 - r2)) << 32) | ((n1 << 1) | (1 - r1));
            recs.push_back({kf, r, i1, i2, a1, a2});                 // forward first (:23)
            recs.push_back({kt, r | 0x80000000u, i2, i1, a2, a1});   // then the twin (:24)
        }
    }
    std::stable_sort(recs.begin(), recs.end(), [](const Rec& x, const Rec& y) { return x.key < y.key; });
    size_t n = recs.size();
    out->n_rec = n;
    out->key = (uint64_t*)malloc(std::max<size_t>(1, n) * 8);
    out->lr = (uint32_t*)malloc(std::max<size_t>(1, n) * 4);
    out->cmp_head = (uint32_t*)malloc(std::max<size_t>(1, n) * 4);
    out->cmp_tail = (uint32_t*)malloc(std::max<size_t>(1, n) * 4);
    alloc_side(out->head, n); alloc_side(out->tail, n);
    std::vector<uint64_t> ekey, eoff;
    for (size_t i = 0; i < n; i++) {
        out->key[i] = recs[i].key; out->lr[i] = recs[i].lr; out->cmp_head[i] = recs[i].ch; out->cmp_tail[i] = recs[i].ct;
        fill_side(out->head, i, c, h, recs[i].ah); fill_side(out->tail, i, c, h, recs[i].at);
        if (i == 0 || recs[i].key != recs[i - 1].key) { ekey.push_back(recs[i].key); eoff.push_back(i); }
    }
    eoff.push_back(n);
    out->n_edge = ekey.size(); out->edge_key = dup(ekey); out->edge_off = dup(eoff);
    return 0;
}

extern "C" void orc_free_edges(hx_edges_out* o) {
    free(o->key); free(o->lr); free(o->cmp_head); free(o->cmp_tail); free_side(o->head); free_side(o->tail);
    free(o->edge_key); free(o->edge_off);
    memset(o, 0, sizeof(*o));
}

// =====================================================================================================
// a8: edge coordinates (asm_calc_single_edge_coordinates, Assemble.cpp:157-363)   [PARITY UNPINNED]
// =====================================================================================================
namespace {
// asm_best_supported_interval_contig1/2 (Assemble.cpp:24-126): sweep over sorted (t_start,i) / (t_end,i);
// `last_max` selects >= (contig1, :45) or > (contig2, :97).
void best_interval(std::vector<std::pair<uint32_t, uint32_t>> beg, std::vector<std::pair<uint32_t, uint32_t>> end,
                   bool last_max, uint32_t& beg_best, uint32_t& end_best, std::set<uint32_t>& best_lrs) {
    std::sort(beg.begin(), beg.end());
    std::sort(end.begin(), end.end());
    int curr = 0, best = 0, i = 0, j = 0, len = (int)beg.size();
    bool started = false;
    std::set<uint32_t> cur;
    beg_best = end_best = 0;
    while (i < len && j < len) {
        if (beg[i].first < end[j].first) {
            curr++;
            cur.insert(beg[i].second);
            if (last_max ? curr >= best : curr > best) { best = curr; beg_best = beg[i].first; best_lrs = cur; started = true; }
            i++;
        } else {
            if (started) { end_best = end[j].first; started = false; }
            curr--;
            cur.erase(end[j].second);
            j++;
        }
    }
    if (started && j < len) end_best = end[j].first;
}
}  // namespace

extern "C" int orc_edge_coords(const hx_contigs* ctg, const uint32_t* read_len, const hx_hits* h, const hx_edges_out* ed,
                               uint32_t n_sel, const uint32_t* sel_edge, hx_coords_out* out) {
    memset(out, 0, sizeof(*out));
    std::vector<uint32_t> head_end(n_sel), tail_beg(n_sel), slr, spos, epos;
    std::vector<uint64_t> soff(n_sel + 1, 0);
    for (uint32_t s = 0; s < n_sel; s++) {
        uint32_t e = sel_edge[s];
        if (e >= ed->n_edge) return fail("orc_edge_coords: edge index out of range");
        uint64_t key = ed->edge_key[e];
        uint32_t v1 = (uint32_t)(key >> 32), to = (uint32_t)key;
        uint32_t node1 = v1 >> 1, rev1 = v1 & 1, node2 = to >> 1, rev2 = to & 1;
        uint64_t b = ed->edge_off[e], n = ed->edge_off[e + 1] - b;
        const bool hairpin = (to ^ 1u) == v1;   // the arc is its own twin (node2 == node1, rev2 == 1 - rev1)
        soff[s] = slr.size();
        std::vector<std::pair<uint32_t, uint32_t>> b1(n), e1(n), b2(n), e2(n);
        for (uint64_t i = 0; i < n; i++) {
            b1[i] = {ed->head.t_start[b + i], (uint32_t)i}; e1[i] = {ed->head.t_end[b + i], (uint32_t)i};
            b2[i] = {ed->tail.t_start[b + i], (uint32_t)i}; e2[i] = {ed->tail.t_end[b + i], (uint32_t)i};
        }
        uint32_t bb1, eb1, bb2, eb2;
        std::set<uint32_t> l1, l2;
        best_interval(b1, e1, true, bb1, eb1, l1);
        best_interval(b2, e2, false, bb2, eb2, l2);
        uint32_t c1pos = rev1 == 0 ? eb1 - 1 : bb1;   // :228-231
        uint32_t c2pos = rev2 == 0 ? bb2 : eb2 - 1;   // :232-235
        std::vector<uint32_t> best;
        std::set_intersection(l1.begin(), l1.end(), l2.begin(), l2.end(), std::back_inserter(best));  // :238
        size_t n_before = slr.size();
        for (uint32_t bi : best) {
            uint64_t x = b + bi;
            uint32_t rid = ed->lr[x] & 0x7fffffffu;
            uint32_t rlen = read_len[rid];
            const hx_rec_side &H = ed->head, &T = ed->tail;
            uint32_t rstrand = (rev1 == H.is_rev[x]) ? 0 : 1;   // :262
            CgView vh{h->cg_ops, H.cg_begin[x], H.cg_end[x], H.cg_skip_front[x], H.cg_skip_back[x]};
            CgView vt{h->cg_ops, T.cg_begin[x], T.cg_end[x], T.cg_skip_front[x], T.cg_skip_back[x]};
            long long ls, le;
            if (rstrand == 0) {   // cases 1-4 (:269-295)
                ls = rev1 == 0 ? find_lr_pos(vh, false, H.q_start[x], H.t_start[x], +1, +1, c1pos)
                               : find_lr_pos(vh, true, H.q_start[x], H.t_end[x] - 1, +1, -1, c1pos);
                le = rev2 == 0 ? find_lr_pos(vt, true, T.q_end[x] - 1, T.t_end[x] - 1, -1, -1, c2pos)
                               : find_lr_pos(vt, false, T.q_end[x] - 1, T.t_start[x], -1, +1, c2pos);
            } else {              // cases 5-8 (:297-324)
                ls = rev1 == 0 ? find_lr_pos(vh, false, rlen - H.q_end[x], H.t_start[x], +1, +1, c1pos)
                               : find_lr_pos(vh, true, rlen - H.q_end[x], H.t_end[x] - 1, +1, -1, c1pos);
                le = rev2 == 0 ? find_lr_pos(vt, true, rlen - T.q_start[x] - 1, T.t_end[x] - 1, -1, -1, c2pos)
                               : find_lr_pos(vt, false, rlen - T.q_start[x] - 1, T.t_start[x], -1, +1, c2pos);
            }
            if (ls != -1 && le != -1) {   // :326-331
                slr.push_back(rid | (rstrand << 31));
                spos.push_back((uint32_t)(ls + 1));
                epos.push_back((uint32_t)(le - 1));
                if (hairpin) {   // edge and twin are the same object: the mirrored entry lands in the same vector (:331)
                    slr.push_back(rid | ((1 - rstrand) << 31));
                    spos.push_back((uint32_t)(rlen - (le - 1) - 1));
                    epos.push_back((uint32_t)(rlen - (ls + 1) - 1));
                }
            }
        }
        if (slr.size() > n_before) { head_end[s] = c1pos; tail_beg[s] = c2pos; }   // :351-352
        else {                                                                      // :244-251, :354-361
            head_end[s] = rev1 == 0 ? ctg->len[node1] - 1 : 0;
            tail_beg[s] = rev2 == 0 ? 0 : ctg->len[node2] - 1;
        }
    }
    soff[n_sel] = slr.size();
    out->n_edge = n_sel; out->head_end = dup(head_end); out->tail_beg = dup(tail_beg); out->supp_off = dup(soff);
    out->supp_lr = dup(slr); out->spos = dup(spos); out->epos = dup(epos);
    return 0;
}

extern "C" void orc_free_coords(hx_coords_out* o) {
    free(o->head_end); free(o->tail_beg); free(o->supp_off); free(o->supp_lr); free(o->spos); free(o->epos);
    memset(o, 0, sizeof(*o));
}

// =====================================================================================================
// a9: partial-order alignment consensus                                              [PARITY UNPINNED]
// Restatement of the published algorithm of rvaser/spoa tag 1.1.3 as used by Assemble.cpp:499-554:
// createAlignmentEngine(kNW, 5, -4, -8), createGraph(), then per sequence align_sequence_with_graph +
// add_alignment(weight 1), finally generate_consensus(). The SIMD and SISD engines of spoa are meant to
// return the same alignment; this follows the SISD formulation (full (V+1)x(L+1) int32 matrix).
// =====================================================================================================
namespace {

struct PoaGraph {
    struct Edge { uint32_t from, to; int64_t w; };
    std::vector<uint8_t> code;                     // node base, 0..3
    std::vector<std::vector<uint32_t>> in, outs;   // edge ids, insertion order
    std::vector<std::vector<uint32_t>> aligned;    // aligned node ids, insertion order
    std::vector<Edge> edges;
    std::vector<uint32_t> rank2node;

    uint32_t add_node(uint8_t c) { code.push_back(c); in.emplace_back(); outs.emplace_back(); aligned.emplace_back(); return (uint32_t)code.size() - 1; }
    // spoa Graph::add_edge: an existing (from,to) edge gains the weight, else a new edge is appended
    void add_edge(uint32_t f, uint32_t t, int64_t w) {
        for (uint32_t e : outs[f]) if (edges[e].to == t) { edges[e].w += w; return; }
        edges.push_back({f, t, w});
        outs[f].push_back((uint32_t)edges.size() - 1);
        in[t].push_back((uint32_t)edges.size() - 1);
    }
    // spoa Graph::add_sequence: a fresh chain for seq[b,e); unit weights => every edge weight 2
    int32_t add_chain(const uint8_t* seq, uint32_t b, uint32_t e) {
        if (b == e) return -1;
        uint32_t first = add_node(seq[b]);
        for (uint32_t i = b + 1; i < e; i++) { uint32_t n = add_node(seq[i]); add_edge(n - 1, n, 2); }
        return (int32_t)first;
    }
    // spoa Graph::topological_sort: iterative DFS over in-edges and aligned nodes; a node is emitted
    // together with its aligned nodes, which are never emitted on their own.
    void toposort() {
        size_t V = code.size();
        rank2node.clear();
        std::vector<uint8_t> mark(V, 0);
        std::vector<char> check(V, 1);
        std::vector<uint32_t> st;
        for (uint32_t i = 0; i < V; i++) {
            if (mark[i]) continue;
            st.push_back(i);
            while (!st.empty()) {
                uint32_t n = st.back();
                bool valid = true;
                if (mark[n] != 2) {
                    for (uint32_t e : in[n]) if (mark[edges[e].from] != 2) { st.push_back(edges[e].from); valid = false; }
                    if (check[n]) for (uint32_t a : aligned[n]) if (mark[a] != 2) { st.push_back(a); check[a] = 0; valid = false; }
                    if (valid) {
                        mark[n] = 2;
                        if (check[n]) { rank2node.push_back(n); for (uint32_t a : aligned[n]) rank2node.push_back(a); }
                    } else mark[n] = 1;
                }
                if (valid) st.pop_back();
            }
        }
    }
    // spoa Graph::add_alignment with unit weights
    void add_alignment(const std::vector<std::pair<int32_t, int32_t>>& aln, const uint8_t* seq, uint32_t len) {
        if (len == 0) return;
        if (aln.empty()) { add_chain(seq, 0, len); toposort(); return; }
        std::vector<uint32_t> valid;
        for (auto& p : aln) if (p.second != -1) valid.push_back((uint32_t)p.second);
        uint32_t before = (uint32_t)code.size();
        add_chain(seq, 0, valid.front());
        int32_t head = before == code.size() ? -1 : (int32_t)code.size() - 1;
        int32_t tail = add_chain(seq, valid.back() + 1, len);
        int32_t nn = -1;
        for (auto& p : aln) {
            if (p.second == -1) continue;
            uint8_t c = seq[p.second];
            if (p.first == -1) nn = (int32_t)add_node(c);
            else if (code[p.first] == c) nn = p.first;
            else {
                int32_t hit = -1;
                for (uint32_t a : aligned[p.first]) if (code[a] == c) { hit = (int32_t)a; break; }
                if (hit == -1) {
                    nn = (int32_t)add_node(c);
                    for (uint32_t a : aligned[p.first]) { aligned[nn].push_back(a); aligned[a].push_back(nn); }
                    aligned[nn].push_back(p.first);
                    aligned[p.first].push_back(nn);
                } else nn = hit;
            }
            if (head != -1) add_edge(head, nn, 2);
            head = nn;
        }
        if (tail != -1) add_edge(head, tail, 2);
        toposort();
    }
    // spoa Graph::traverse_heaviest_bundle + branch_completion
    std::vector<uint32_t> consensus() const {
        size_t V = code.size();
        std::vector<int32_t> pred(V, -1);
        std::vector<int64_t> score(V, -1);
        uint32_t best = 0;
        auto relax = [&](uint32_t n, bool skip_dead) {
            for (uint32_t e : in[n]) {
                uint32_t f = edges[e].from;
                if (skip_dead && score[f] == -1) continue;
                if (score[n] < edges[e].w || (score[n] == edges[e].w && score[pred[n]] <= score[f])) { score[n] = edges[e].w; pred[n] = (int32_t)f; }
            }
            if (pred[n] != -1) score[n] += score[pred[n]];
        };
        for (uint32_t n : rank2node) { relax(n, false); if (score[best] < score[n]) best = n; }
        if (!outs[best].empty()) {
            std::vector<uint32_t> rank(V, 0);
            for (uint32_t i = 0; i < rank2node.size(); i++) rank[rank2node[i]] = i;
            while (!outs[best].empty()) {
                uint32_t n0 = best;
                for (uint32_t e : outs[n0]) for (uint32_t oe : in[edges[e].to]) if (edges[oe].from != n0) score[edges[oe].from] = -1;
                int64_t mx = 0; uint32_t mxid = 0;
                for (uint32_t i = rank[n0] + 1; i < rank2node.size(); i++) {
                    uint32_t n = rank2node[i];
                    score[n] = -1; pred[n] = -1;
                    relax(n, true);
                    if (mx < score[n]) { mx = score[n]; mxid = n; }
                }
                best = mxid;
            }
        }
        std::vector<uint32_t> path;
        while (pred[best] != -1) { path.push_back(best); best = (uint32_t)pred[best]; }
        path.push_back(best);
        std::reverse(path.begin(), path.end());
        return path;
    }
};

// ---- row kernels of the aligner. The scalar forms are the restatement; the AVX2 forms compute the same integers 8 columns at a
// time (like spoa's SIMD engine does for this recurrence: element-wise maxima over the predecessor rows, then the horizontal
// recurrence H[j] = max(T[j], H[j-1] + g) as a prefix-max scan inside a vector with a carry between vectors). They exist so that
// bench.py's cpu_baseline is not a scalar straw man; selected at run time (the library is built for plain x86-64-v2).
struct RowKernels {
    // row[j] = max(pw[j-1] + pr[j], pw[j] + g)            for j in [1, W)   (first predecessor)
    void (*first)(int32_t* row, const int32_t* pw, const int32_t* pr, int32_t g, size_t W);
    // row[j] = max(row[j], pw[j-1] + pr[j], pw[j] + g)    for j in [1, W)   (further predecessors)
    void (*more)(int32_t* row, const int32_t* pw, const int32_t* pr, int32_t g, size_t W);
    // row[j] = max(row[j-1] + g, row[j])                  for j in [1, W)   (row[0] is final)
    void (*horiz)(int32_t* row, int32_t g, size_t W);
    // `first` and `horiz` in one pass (rows with a single predecessor)
    void (*first_horiz)(int32_t* row, const int32_t* pw, const int32_t* pr, int32_t g, size_t W);
    const char* name;
};

void row_first_scalar(int32_t* row, const int32_t* pw, const int32_t* pr, int32_t g, size_t W) {
    for (size_t j = 1; j < W; j++) row[j] = std::max(pw[j - 1] + pr[j], pw[j] + g);
}
void row_more_scalar(int32_t* row, const int32_t* pw, const int32_t* pr, int32_t g, size_t W) {
    for (size_t j = 1; j < W; j++) row[j] = std::max(pw[j - 1] + pr[j], std::max(row[j], pw[j] + g));
}
void row_horiz_scalar(int32_t* row, int32_t g, size_t W) {
    for (size_t j = 1; j < W; j++) row[j] = std::max(row[j - 1] + g, row[j]);
}

void row_first_horiz_scalar(int32_t* row, const int32_t* pw, const int32_t* pr, int32_t g, size_t W) {
    for (size_t j = 1; j < W; j++) row[j] = std::max(row[j - 1] + g, std::max(pw[j - 1] + pr[j], pw[j] + g));
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) void row_first_avx2(int32_t* row, const int32_t* pw, const int32_t* pr, int32_t g, size_t W) {
    const __m256i vg = _mm256_set1_epi32(g);
    size_t j = 1;
    for (; j + 8 <= W; j += 8) {
        const __m256i d = _mm256_add_epi32(_mm256_loadu_si256((const __m256i*)(pw + j - 1)), _mm256_loadu_si256((const __m256i*)(pr + j)));
        const __m256i v = _mm256_add_epi32(_mm256_loadu_si256((const __m256i*)(pw + j)), vg);
        _mm256_storeu_si256((__m256i*)(row + j), _mm256_max_epi32(d, v));
    }
    for (; j < W; j++) row[j] = std::max(pw[j - 1] + pr[j], pw[j] + g);
}
__attribute__((target("avx2"))) void row_more_avx2(int32_t* row, const int32_t* pw, const int32_t* pr, int32_t g, size_t W) {
    const __m256i vg = _mm256_set1_epi32(g);
    size_t j = 1;
    for (; j + 8 <= W; j += 8) {
        const __m256i d = _mm256_add_epi32(_mm256_loadu_si256((const __m256i*)(pw + j - 1)), _mm256_loadu_si256((const __m256i*)(pr + j)));
        const __m256i v = _mm256_add_epi32(_mm256_loadu_si256((const __m256i*)(pw + j)), vg);
        _mm256_storeu_si256((__m256i*)(row + j), _mm256_max_epi32(_mm256_loadu_si256((const __m256i*)(row + j)), _mm256_max_epi32(d, v)));
    }
    for (; j < W; j++) row[j] = std::max(pw[j - 1] + pr[j], std::max(row[j], pw[j] + g));
}
__attribute__((target("avx2"))) void row_horiz_avx2(int32_t* row, int32_t g, size_t W) {
    // inside a vector: y[k] = max over i <= k of x[i] + (k - i) g  (three shift-and-max steps), then the carry of the columns before it
    const __m256i neg = _mm256_set1_epi32(INT32_MIN / 2);
    const __m256i s1 = _mm256_setr_epi32(0, 0, 1, 2, 3, 4, 5, 6), s2 = _mm256_setr_epi32(0, 0, 0, 1, 2, 3, 4, 5), s4 = _mm256_setr_epi32(0, 0, 0, 0, 0, 1, 2, 3);
    const __m256i g1 = _mm256_set1_epi32(g), g2 = _mm256_set1_epi32(2 * g), g4 = _mm256_set1_epi32(4 * g);
    const __m256i ramp = _mm256_mullo_epi32(_mm256_setr_epi32(1, 2, 3, 4, 5, 6, 7, 8), g1), last = _mm256_set1_epi32(7);
    __m256i carry = _mm256_set1_epi32(row[0]);
    size_t j = 1;
    for (; j + 8 <= W; j += 8) {
        __m256i x = _mm256_loadu_si256((const __m256i*)(row + j));
        x = _mm256_max_epi32(x, _mm256_add_epi32(_mm256_blend_epi32(_mm256_permutevar8x32_epi32(x, s1), neg, 0x01), g1));
        x = _mm256_max_epi32(x, _mm256_add_epi32(_mm256_blend_epi32(_mm256_permutevar8x32_epi32(x, s2), neg, 0x03), g2));
        x = _mm256_max_epi32(x, _mm256_add_epi32(_mm256_blend_epi32(_mm256_permutevar8x32_epi32(x, s4), neg, 0x0f), g4));
        x = _mm256_max_epi32(x, _mm256_add_epi32(carry, ramp));
        _mm256_storeu_si256((__m256i*)(row + j), x);
        carry = _mm256_permutevar8x32_epi32(x, last);
    }
    for (; j < W; j++) row[j] = std::max(row[j - 1] + g, row[j]);
}
__attribute__((target("avx2"))) void row_first_horiz_avx2(int32_t* row, const int32_t* pw, const int32_t* pr, int32_t g, size_t W) {
    const __m256i neg = _mm256_set1_epi32(INT32_MIN / 2);
    const __m256i s1 = _mm256_setr_epi32(0, 0, 1, 2, 3, 4, 5, 6), s2 = _mm256_setr_epi32(0, 0, 0, 1, 2, 3, 4, 5), s4 = _mm256_setr_epi32(0, 0, 0, 0, 0, 1, 2, 3);
    const __m256i g1 = _mm256_set1_epi32(g), g2 = _mm256_set1_epi32(2 * g), g4 = _mm256_set1_epi32(4 * g);
    const __m256i ramp = _mm256_mullo_epi32(_mm256_setr_epi32(1, 2, 3, 4, 5, 6, 7, 8), g1), last = _mm256_set1_epi32(7);
    __m256i carry = _mm256_set1_epi32(row[0]);
    size_t j = 1;
    for (; j + 8 <= W; j += 8) {
        const __m256i d = _mm256_add_epi32(_mm256_loadu_si256((const __m256i*)(pw + j - 1)), _mm256_loadu_si256((const __m256i*)(pr + j)));
        __m256i x = _mm256_max_epi32(d, _mm256_add_epi32(_mm256_loadu_si256((const __m256i*)(pw + j)), g1));
        x = _mm256_max_epi32(x, _mm256_add_epi32(_mm256_blend_epi32(_mm256_permutevar8x32_epi32(x, s1), neg, 0x01), g1));
        x = _mm256_max_epi32(x, _mm256_add_epi32(_mm256_blend_epi32(_mm256_permutevar8x32_epi32(x, s2), neg, 0x03), g2));
        x = _mm256_max_epi32(x, _mm256_add_epi32(_mm256_blend_epi32(_mm256_permutevar8x32_epi32(x, s4), neg, 0x0f), g4));
        x = _mm256_max_epi32(x, _mm256_add_epi32(carry, ramp));
        _mm256_storeu_si256((__m256i*)(row + j), x);
        carry = _mm256_permutevar8x32_epi32(x, last);
    }
    for (; j < W; j++) row[j] = std::max(row[j - 1] + g, std::max(pw[j - 1] + pr[j], pw[j] + g));
}
#endif

// ---- the same four kernels on int16 cells, 16 per AVX2 vector: what spoa's SIMD engine runs when the scores fit 16 bits (its dispatch:
// max penalty x (nodes + columns) below the int16 range; here 8 (V + L) + 256 < 32767, so that no real value ever saturates and the integers
// are those of the int32 forms). Half the memory traffic per cell, twice the cells per instruction: bench.py's cpu_baseline is then the engine
// width the reference would run at on short gaps, not the slower one. Saturating adds keep the "minus infinity" filler (-32768) from wrapping.
struct RowKernels16 {
    void (*first)(int16_t* row, const int16_t* pw, const int16_t* pr, int16_t g, size_t W);
    void (*more)(int16_t* row, const int16_t* pw, const int16_t* pr, int16_t g, size_t W);
    void (*horiz)(int16_t* row, int16_t g, size_t W);
    void (*first_horiz)(int16_t* row, const int16_t* pw, const int16_t* pr, int16_t g, size_t W);
    const char* name;
};
void row16_first_scalar(int16_t* row, const int16_t* pw, const int16_t* pr, int16_t g, size_t W) { for (size_t j = 1; j < W; j++) row[j] = (int16_t)std::max(pw[j - 1] + pr[j], pw[j] + g); }
void row16_more_scalar(int16_t* row, const int16_t* pw, const int16_t* pr, int16_t g, size_t W) { for (size_t j = 1; j < W; j++) row[j] = (int16_t)std::max(pw[j - 1] + pr[j], std::max<int>(row[j], pw[j] + g)); }
void row16_horiz_scalar(int16_t* row, int16_t g, size_t W) { for (size_t j = 1; j < W; j++) row[j] = (int16_t)std::max<int>(row[j - 1] + g, row[j]); }
void row16_first_horiz_scalar(int16_t* row, const int16_t* pw, const int16_t* pr, int16_t g, size_t W) { for (size_t j = 1; j < W; j++) row[j] = (int16_t)std::max(row[j - 1] + g, std::max(pw[j - 1] + pr[j], pw[j] + g)); }
#if defined(__x86_64__)
// x shifted up by K 16-bit lanes (K = 1, 2, 4, 8) with "minus infinity" in the K lanes that open, plus K gaps: the step of the in-vector prefix scan
template <int K> __attribute__((target("avx2"))) inline __m256i scan16_step(__m256i x, __m256i negk, __m256i gk) {
    const __m256i t = _mm256_permute2x128_si256(x, x, 0x08);                       // [0, x.low]
    const __m256i sh = K == 8 ? t : _mm256_alignr_epi8(x, t, (16 - 2 * K) & 15);   // lanes move up by K words across the 128-bit halves
    return _mm256_max_epi16(x, _mm256_adds_epi16(_mm256_or_si256(sh, negk), gk));
}
struct Scan16 {
    __m256i n1, n2, n4, n8, g1, g2, g4, g8, ramp;
    __attribute__((target("avx2"))) explicit Scan16(int16_t g) {
        alignas(32) int16_t a[16];
        __m256i* const out[4] = {&n1, &n2, &n4, &n8};
        for (int q = 0; q < 4; q++) { for (int i = 0; i < 16; i++) a[i] = i < (1 << q) ? (int16_t)0x8000 : (int16_t)0; *out[q] = _mm256_load_si256((const __m256i*)a); }
        g1 = _mm256_set1_epi16(g); g2 = _mm256_set1_epi16((int16_t)(2 * g)); g4 = _mm256_set1_epi16((int16_t)(4 * g)); g8 = _mm256_set1_epi16((int16_t)(8 * g));
        for (int i = 0; i < 16; i++) a[i] = (int16_t)((i + 1) * g);
        ramp = _mm256_load_si256((const __m256i*)a);
    }
    __attribute__((target("avx2"))) inline __m256i run(__m256i x, int16_t& carry) const {
        x = scan16_step<1>(x, n1, g1); x = scan16_step<2>(x, n2, g2); x = scan16_step<4>(x, n4, g4); x = scan16_step<8>(x, n8, g8);
        x = _mm256_max_epi16(x, _mm256_adds_epi16(_mm256_set1_epi16(carry), ramp));
        carry = (int16_t)_mm256_extract_epi16(x, 15);
        return x;
    }
};
__attribute__((target("avx2"))) void row16_first_avx2(int16_t* row, const int16_t* pw, const int16_t* pr, int16_t g, size_t W) {
    const __m256i vg = _mm256_set1_epi16(g);
    size_t j = 1;
    for (; j + 16 <= W; j += 16) {
        const __m256i d = _mm256_adds_epi16(_mm256_loadu_si256((const __m256i*)(pw + j - 1)), _mm256_loadu_si256((const __m256i*)(pr + j)));
        const __m256i v = _mm256_adds_epi16(_mm256_loadu_si256((const __m256i*)(pw + j)), vg);
        _mm256_storeu_si256((__m256i*)(row + j), _mm256_max_epi16(d, v));
    }
    for (; j < W; j++) row[j] = (int16_t)std::max(pw[j - 1] + pr[j], pw[j] + g);
}
__attribute__((target("avx2"))) void row16_more_avx2(int16_t* row, const int16_t* pw, const int16_t* pr, int16_t g, size_t W) {
    const __m256i vg = _mm256_set1_epi16(g);
    size_t j = 1;
    for (; j + 16 <= W; j += 16) {
        const __m256i d = _mm256_adds_epi16(_mm256_loadu_si256((const __m256i*)(pw + j - 1)), _mm256_loadu_si256((const __m256i*)(pr + j)));
        const __m256i v = _mm256_adds_epi16(_mm256_loadu_si256((const __m256i*)(pw + j)), vg);
        _mm256_storeu_si256((__m256i*)(row + j), _mm256_max_epi16(_mm256_loadu_si256((const __m256i*)(row + j)), _mm256_max_epi16(d, v)));
    }
    for (; j < W; j++) row[j] = (int16_t)std::max(pw[j - 1] + pr[j], std::max<int>(row[j], pw[j] + g));
}
__attribute__((target("avx2"))) void row16_horiz_avx2(int16_t* row, int16_t g, size_t W) {
    const Scan16 S(g);
    int16_t carry = row[0];
    size_t j = 1;
    for (; j + 16 <= W; j += 16) _mm256_storeu_si256((__m256i*)(row + j), S.run(_mm256_loadu_si256((const __m256i*)(row + j)), carry));
    for (; j < W; j++) row[j] = (int16_t)std::max<int>(row[j - 1] + g, row[j]);
}
__attribute__((target("avx2"))) void row16_first_horiz_avx2(int16_t* row, const int16_t* pw, const int16_t* pr, int16_t g, size_t W) {
    const Scan16 S(g);
    const __m256i vg = _mm256_set1_epi16(g);
    int16_t carry = row[0];
    size_t j = 1;
    for (; j + 16 <= W; j += 16) {
        const __m256i d = _mm256_adds_epi16(_mm256_loadu_si256((const __m256i*)(pw + j - 1)), _mm256_loadu_si256((const __m256i*)(pr + j)));
        const __m256i v = _mm256_adds_epi16(_mm256_loadu_si256((const __m256i*)(pw + j)), vg);
        _mm256_storeu_si256((__m256i*)(row + j), S.run(_mm256_max_epi16(d, v), carry));
    }
    for (; j < W; j++) row[j] = (int16_t)std::max(row[j - 1] + g, std::max(pw[j - 1] + pr[j], pw[j] + g));
}
#endif
const RowKernels16* row_kernels16() {   // nullptr: the int16 path is off (ORC_POA_INT16=0)
    static const RowKernels16 k = []() {
        RowKernels16 r{row16_first_scalar, row16_more_scalar, row16_horiz_scalar, row16_first_horiz_scalar, "scalar int16"};
#if defined(__x86_64__)
        const char* force = getenv("ORC_POA_SCALAR");
        if (!(force && force[0] == '1') && __builtin_cpu_supports("avx2")) r = RowKernels16{row16_first_avx2, row16_more_avx2, row16_horiz_avx2, row16_first_horiz_avx2, "avx2 (16 x int16)"};
#endif
        return r;
    }();
    static const bool on = !(getenv("ORC_POA_INT16") && getenv("ORC_POA_INT16")[0] == '0');
    return on ? &k : nullptr;
}

const RowKernels& row_kernels() {
    static const RowKernels k = []() {
        RowKernels r{row_first_scalar, row_more_scalar, row_horiz_scalar, row_first_horiz_scalar, "scalar"};
#if defined(__x86_64__)
        const char* force = getenv("ORC_POA_SCALAR");
        if (!(force && force[0] == '1') && __builtin_cpu_supports("avx2")) r = RowKernels{row_first_avx2, row_more_avx2, row_horiz_avx2, row_first_horiz_avx2, "avx2 (8 x int32)"};
#endif
        return r;
    }();
    return k;
}

struct PoaAligner {
    int32_t m, x, g;
    std::vector<int32_t> H, prof;
    std::vector<int16_t> H16, prof16;
    std::vector<uint32_t> node2rank;
    int64_t prev_score = 0; uint32_t prev_len = 0;   // (ORC_POA_PRUNE_SIM only: the previous alignment of this edge, the source of the pruning threshold)
    // spoa SisdAlignmentEngine::align (kNW, linear gap). Returns (node|-1, seq pos|-1) pairs.
    std::vector<std::pair<int32_t, int32_t>> align(const PoaGraph& G, const uint8_t* seq, uint32_t len, uint64_t* cells) {
        std::vector<std::pair<int32_t, int32_t>> aln;
        size_t V = G.code.size();
        if (V == 0 || len == 0) return aln;
        const size_t W = (size_t)len + 1;
        *cells += (uint64_t)V * len;
        {   // int16 cells where the scores fit (what spoa's SIMD engine does): same integers, see RowKernels16. The development statistics read the int32 matrix.
            static const bool stats = getenv("ORC_POA_TIES") || getenv("ORC_POA_PRUNE") || getenv("ORC_POA_TB") || getenv("ORC_POA_PRUNE_SIM");
            const int64_t pen = std::max<int64_t>(std::max<int64_t>(std::llabs(m), std::llabs(x)), std::llabs(g));
            const RowKernels16* K16 = stats ? nullptr : row_kernels16();
            if (K16 && pen * (int64_t)(V + len) + 32 * pen < 32767) return align16(G, seq, len, *K16);
        }
        H.resize((V + 1) * W);
        prof.resize(4 * W);
        for (int c = 0; c < 4; c++) { prof[c * W] = 0; for (uint32_t j = 0; j < len; j++) prof[c * W + j + 1] = seq[j] == c ? m : x; }
        node2rank.resize(V);
        for (uint32_t i = 0; i < V; i++) node2rank[G.rank2node[i]] = i;
        H[0] = 0;
        for (size_t j = 1; j < W; j++) H[j] = (int32_t)j * g;
        for (size_t i = 1; i <= V; i++) {   // first column
            uint32_t n = G.rank2node[i - 1];
            if (G.in[n].empty()) H[i * W] = g;
            else {
                int32_t pen = INT32_MIN + 1024;
                for (uint32_t e : G.in[n]) pen = std::max(pen, H[(size_t)(node2rank[G.edges[e].from] + 1) * W]);
                H[i * W] = pen + g;
            }
        }
        int32_t max_score = INT32_MIN + 1024; int64_t max_i = -1;
        const RowKernels& K = row_kernels();
        for (size_t i = 1; i <= V; i++) {
            uint32_t n = G.rank2node[i - 1];
            const int32_t* pr = &prof[(size_t)G.code[n] * W];
            int32_t* row = &H[i * W];
            size_t pi = G.in[n].empty() ? 0 : node2rank[G.edges[G.in[n][0]].from] + 1;
            const int32_t* pw = &H[pi * W];
            if (G.in[n].size() <= 1) K.first_horiz(row, pw, pr, g, W);
            else {
                K.first(row, pw, pr, g, W);
                for (size_t p = 1; p < G.in[n].size(); p++) K.more(row, &H[(size_t)(node2rank[G.edges[G.in[n][p]].from] + 1) * W], pr, g, W);
                K.horiz(row, g, W);
            }
            if (G.outs[n].empty() && max_score < row[W - 1]) { max_score = row[W - 1]; max_i = (int64_t)i; }   // first max in rank order
        }
        if (getenv("ORC_POA_TIES")) {   // development statistics: how end-node ties look (what the kernel's tie shortcuts have to decide without the DFS order)
            std::vector<size_t> cand;
            for (size_t r = 1; r <= V; r++) if (G.outs[G.rank2node[r - 1]].empty() && H[r * W + W - 1] == max_score) cand.push_back(r);
            if (cand.size() > 1) {
                auto same_col = [&](uint32_t a, uint32_t b) { if (a == b) return true; for (uint32_t x : G.aligned[a]) if (x == b) return true; return false; };
                const uint32_t n0 = G.rank2node[cand[0] - 1];
                bool onecol = true, allsink = G.outs[n0].empty();
                for (size_t c : cand) onecol = onecol && same_col(n0, G.rank2node[c - 1]);
                for (uint32_t a : G.aligned[n0]) allsink = allsink && G.outs[a].empty();
                uint32_t minid = 0xffffffffu; for (size_t c : cand) minid = std::min(minid, G.rank2node[c - 1]);
                // the column's first emitted member (the one the DFS reached with check = 1) is the first of the column in rank order
                size_t r0 = cand[0]; while (r0 > 1 && same_col(n0, G.rank2node[r0 - 2])) r0--;
                const uint32_t lead = G.rank2node[r0 - 1];
                uint32_t colmin = lead; for (uint32_t a : G.aligned[lead]) colmin = std::min(colmin, a);
                // the local simulation the kernel uses (poa.hip, "local order"): U = the candidates' columns and everything downstream of them (closed under
                // out-edges and aligned mates); the reference's DFS restricted to U, roots in id order, predecessors outside U taken as finished
                {
                    std::vector<uint32_t> U;
                    auto inU = [&](uint32_t x) { return std::find(U.begin(), U.end(), x) != U.end(); };
                    auto addcol = [&](uint32_t x) { if (inU(x)) return; U.push_back(x); for (uint32_t a : G.aligned[x]) if (!inU(a)) U.push_back(a); };
                    for (size_t c : cand) addcol(G.rank2node[c - 1]);
                    for (size_t q = 0; q < U.size() && U.size() <= 4096; q++) for (uint32_t e : G.outs[U[q]]) addcol(G.edges[e].to);
                    std::sort(U.begin(), U.end());
                    std::map<uint32_t, int> mk, ck;
                    for (uint32_t u : U) { mk[u] = 0; ck[u] = 1; }
                    std::vector<uint32_t> st; int64_t win = -1;
                    auto is_cand = [&](uint32_t x) { for (size_t c : cand) if (G.rank2node[c - 1] == x) return true; return false; };
                    for (uint32_t r : U) {
                        if (win >= 0) break;
                        if (mk[r]) continue;
                        st.push_back(r);
                        while (!st.empty() && win < 0) {
                            uint32_t n = st.back(); bool valid = true;
                            if (mk[n] != 2) {
                                for (uint32_t e : G.in[n]) { uint32_t f = G.edges[e].from; if (mk.count(f) && mk[f] != 2) { st.push_back(f); valid = false; } }
                                if (ck[n]) for (uint32_t a : G.aligned[n]) if (mk[a] != 2) { st.push_back(a); ck[a] = 0; valid = false; }
                                if (valid) {
                                    mk[n] = 2;
                                    if (ck[n]) { if (is_cand(n)) win = n; else for (uint32_t a : G.aligned[n]) if (is_cand(a)) { win = a; break; } }
                                } else mk[n] = 1;
                            }
                            if (valid) st.pop_back();
                        }
                    }
                    fprintf(stderr, "POALOCAL U=%zu ok=%d\n", U.size(), (int)(win == (int64_t)G.rank2node[cand[0] - 1]));
                }
                fprintf(stderr, "POATIE V=%zu ncand=%zu onecol=%d allsink=%d span=%zu winner_is_min_id=%d colsize=%zu lead_is_col_min_id=%d lead_is_sink=%d winner_is_lead=%d\n", V, cand.size(), (int)onecol, (int)allsink,
                        cand.back() - cand[0], (int)(G.rank2node[cand[0] - 1] == minid), G.aligned[n0].size() + 1, (int)(lead == colmin), (int)G.outs[lead].empty(), (int)(lead == n0));
            }
        }
        if (getenv("ORC_POA_PRUNE")) {   // development statistics: which cells an exact score-bound pruning would have to compute. A cell is LIVE when
            // U = H[i][j] + match * (columns to go) reaches a threshold T <= the final score (no path through a dead cell reaches T: U never
            // grows along a path); a (row, block of BW columns) must be computed when one of its inputs is live: the predecessor rows' cells of
            // the block and of the column on its left, the row's own block on the left (the carry).
            static std::atomic<uint64_t> tot{0}, liveT[2] = {{0}, {0}}, blk[2][3] = {{{0}, {0}, {0}}, {{0}, {0}, {0}}};
            static struct Pr { ~Pr() { fprintf(stderr, "POAPRUNE cells %.4g | T=opt: live %.3f, blocks of 64/256/512 columns %.3f %.3f %.3f | T=0.9 opt: live %.3f, blocks %.3f %.3f %.3f\n", (double)tot.load(),
                (double)liveT[0] / tot, (double)blk[0][0] / tot, (double)blk[0][1] / tot, (double)blk[0][2] / tot, (double)liveT[1] / tot, (double)blk[1][0] / tot, (double)blk[1][1] / tot, (double)blk[1][2] / tot); } } printer;
            tot += (uint64_t)V * W;
            const int BWs[3] = {64, 256, 512};
            for (int ti = 0; ti < 2; ti++) {
                const int64_t T = ti == 0 ? max_score : (max_score > 0 ? (int64_t)(0.9 * max_score) : (int64_t)(1.1 * max_score));
                uint64_t lv = 0;
                for (int bi = 0; bi < 3; bi++) {
                    const size_t BW = BWs[bi], NB = (W + BW - 1) / BW;
                    std::vector<uint8_t> lb((V + 1) * NB, 0);   // block (row, w) holds a live cell
                    for (size_t r = 0; r <= V; r++) for (size_t jj = 0; jj < W; jj++)
                        if ((int64_t)H[r * W + jj] + (int64_t)m * (int64_t)(W - 1 - jj) >= T) { lb[r * NB + jj / BW] = 1; if (bi == 0) lv++; }
                    uint64_t need = 0;
                    for (size_t r = 1; r <= V; r++) {
                        const uint32_t n = G.rank2node[r - 1];
                        for (size_t w = 0; w < NB; w++) {
                            bool in = w > 0 && lb[r * NB + w - 1];
                            auto pred = [&](size_t p) { in = in || lb[p * NB + w] || (w > 0 && lb[p * NB + w - 1]); };
                            if (G.in[n].empty()) pred(0); else for (uint32_t e : G.in[n]) pred(node2rank[G.edges[e].from] + 1);
                            if (in || lb[r * NB + w]) need += std::min(BW, W - w * BW);
                        }
                    }
                    blk[ti][bi] += need;
                }
                liveT[ti] += lv;
            }
        }
        // traceback: diagonal (in-edge order), then vertical (in-edge order), then horizontal
        size_t i = (size_t)max_i, j = W - 1;
        const bool tbstat = getenv("ORC_POA_TB") != nullptr;   // development statistics: how the traceback's steps look (runs a wave could take at once)
        uint64_t tb_steps = 0, tb_diag = 0, tb_diag0 = 0, tb_diag0_lag1 = 0, tb_diag0_lag2 = 0, tb_runs1 = 0, tb_runs12 = 0; bool in1 = false, in12 = false;
        while (!(i == 0 && j == 0)) {
            int32_t hij = H[i * W + j];
            bool found = false;
            size_t pi_ = 0, pj_ = 0;
            if (i != 0 && j != 0) {
                uint32_t n = G.rank2node[i - 1];
                int32_t mc = prof[(size_t)G.code[n] * W + j];
                size_t np = G.in[n].size();
                for (size_t p = 0; p < std::max<size_t>(1, np) && !found; p++) {
                    size_t pi = np == 0 ? 0 : node2rank[G.edges[G.in[n][p]].from] + 1;
                    if (hij == H[pi * W + j - 1] + mc) { pi_ = pi; pj_ = j - 1; found = true; }
                }
            }
            if (!found && i != 0) {
                uint32_t n = G.rank2node[i - 1];
                size_t np = G.in[n].size();
                for (size_t p = 0; p < std::max<size_t>(1, np) && !found; p++) {
                    size_t pi = np == 0 ? 0 : node2rank[G.edges[G.in[n][p]].from] + 1;
                    if (hij == H[pi * W + j] + g) { pi_ = pi; pj_ = j; found = true; }
                }
            }
            if (!found) { pi_ = i; pj_ = j - 1; found = true; }   // horizontal: the only remaining source of H[i][j]
            aln.emplace_back(i == pi_ ? -1 : (int32_t)G.rank2node[i - 1], j == pj_ ? -1 : (int32_t)(j - 1));
            if (tbstat) {
                tb_steps++;
                const bool diag = pi_ != i && pj_ != j;
                bool d0 = false;
                if (diag && i != 0) { uint32_t n = G.rank2node[i - 1]; size_t first = G.in[n].empty() ? 0 : node2rank[G.edges[G.in[n][0]].from] + 1; d0 = first == pi_; }
                tb_diag += diag; tb_diag0 += d0;
                const bool l1 = d0 && pi_ + 1 == i, l12 = d0 && (pi_ + 1 == i || pi_ + 2 == i);
                tb_diag0_lag1 += l1; tb_diag0_lag2 += d0 && pi_ + 2 == i;
                if (l1 && !in1) tb_runs1++;
                if (l12 && !in12) tb_runs12++;
                in1 = l1; in12 = l12;
            }
            i = pi_; j = pj_;
        }
        if (tbstat) fprintf(stderr, "POATB steps=%lu diag=%lu diag_slot0=%lu slot0_lag1=%lu slot0_lag2=%lu runs_lag1=%lu runs_lag12=%lu\n", (unsigned long)tb_steps, (unsigned long)tb_diag, (unsigned long)tb_diag0,
                            (unsigned long)tb_diag0_lag1, (unsigned long)tb_diag0_lag2, (unsigned long)tb_runs1, (unsigned long)tb_runs12);
        std::reverse(aln.begin(), aln.end());
        if (const char* ps = getenv("ORC_POA_PRUNE_SIM")) prune_sim(G, seq, len, atoi(ps) > 0 ? atoi(ps) : 8, max_score, aln);
        prev_score = max_score; prev_len = len;
        return aln;
    }

    // the same alignment on int16 cells (every real value fits: the caller checked); fill and traceback follow align() statement by statement
    std::vector<std::pair<int32_t, int32_t>> align16(const PoaGraph& G, const uint8_t* seq, uint32_t len, const RowKernels16& K) {
        std::vector<std::pair<int32_t, int32_t>> aln;
        const size_t V = G.code.size(), W = (size_t)len + 1;
        std::vector<int16_t>& Hm = H16;
        Hm.resize((V + 1) * W);
        prof16.resize(4 * W);
        const int16_t g16 = (int16_t)g;
        for (int c = 0; c < 4; c++) { prof16[c * W] = 0; for (uint32_t j = 0; j < len; j++) prof16[c * W + j + 1] = (int16_t)(seq[j] == c ? m : x); }
        node2rank.resize(V);
        for (uint32_t i = 0; i < V; i++) node2rank[G.rank2node[i]] = i;
        Hm[0] = 0;
        for (size_t j = 1; j < W; j++) Hm[j] = (int16_t)((int32_t)j * g);
        for (size_t i = 1; i <= V; i++) {   // first column
            const uint32_t n = G.rank2node[i - 1];
            if (G.in[n].empty()) Hm[i * W] = g16;
            else {
                int32_t pen = INT32_MIN + 1024;
                for (uint32_t e : G.in[n]) pen = std::max<int32_t>(pen, Hm[(size_t)(node2rank[G.edges[e].from] + 1) * W]);
                Hm[i * W] = (int16_t)(pen + g);
            }
        }
        int32_t max_score = INT32_MIN + 1024; int64_t max_i = -1;
        for (size_t i = 1; i <= V; i++) {
            const uint32_t n = G.rank2node[i - 1];
            const int16_t* pr = &prof16[(size_t)G.code[n] * W];
            int16_t* row = &Hm[i * W];
            const size_t pi = G.in[n].empty() ? 0 : node2rank[G.edges[G.in[n][0]].from] + 1;
            const int16_t* pw = &Hm[pi * W];
            if (G.in[n].size() <= 1) K.first_horiz(row, pw, pr, g16, W);
            else {
                K.first(row, pw, pr, g16, W);
                for (size_t p = 1; p < G.in[n].size(); p++) K.more(row, &Hm[(size_t)(node2rank[G.edges[G.in[n][p]].from] + 1) * W], pr, g16, W);
                K.horiz(row, g16, W);
            }
            if (G.outs[n].empty() && max_score < row[W - 1]) { max_score = row[W - 1]; max_i = (int64_t)i; }   // first max in rank order
        }
        size_t i = (size_t)max_i, j = W - 1;
        while (!(i == 0 && j == 0)) {   // traceback: diagonal (in-edge order), then vertical (in-edge order), then horizontal
            const int32_t hij = Hm[i * W + j];
            bool found = false;
            size_t pi_ = 0, pj_ = 0;
            if (i != 0 && j != 0) {
                const uint32_t n = G.rank2node[i - 1];
                const int32_t mc = prof16[(size_t)G.code[n] * W + j];
                const size_t np = G.in[n].size();
                for (size_t p = 0; p < std::max<size_t>(1, np) && !found; p++) {
                    const size_t pi = np == 0 ? 0 : node2rank[G.edges[G.in[n][p]].from] + 1;
                    if (hij == Hm[pi * W + j - 1] + mc) { pi_ = pi; pj_ = j - 1; found = true; }
                }
            }
            if (!found && i != 0) {
                const uint32_t n = G.rank2node[i - 1];
                const size_t np = G.in[n].size();
                for (size_t p = 0; p < std::max<size_t>(1, np) && !found; p++) {
                    const size_t pi = np == 0 ? 0 : node2rank[G.edges[G.in[n][p]].from] + 1;
                    if (hij == Hm[pi * W + j] + g) { pi_ = pi; pj_ = j; found = true; }
                }
            }
            if (!found) { pi_ = i; pj_ = j - 1; }
            aln.emplace_back(i == pi_ ? -1 : (int32_t)G.rank2node[i - 1], j == pj_ ? -1 : (int32_t)(j - 1));
            i = pi_; j = pj_;
        }
        std::reverse(aln.begin(), aln.end());
        prev_score = max_score; prev_len = len;
        return aln;
    }

    // ---- ORC_POA_PRUNE_SIM=<columns per lane>: development model of K6's exact score-bound pruning (kernels/poa.hip, "Pruning"), a STATISTIC - the
    // alignment returned above is the unpruned one, always. The model runs the kernel's rule on the CPU: the DP columns of a row are dealt to
    // wavefronts of 64 lanes x CM columns; with U(i, j) = H[i][j] + match x (L - j) an upper bound of any path through the cell (it never grows
    // along a path) and a threshold T, the block (row i, wave w) is COMPUTED only when one of its inputs can still reach T - a predecessor row's
    // block of the same wave that was flagged, or the carry entering from the wave on the left; everything else counts as minus infinity. A
    // computed block is FLAGGED for its successors when one of its lanes passes the per-lane test (the finished key of the lane's last column
    // with the bias of its first), or its carry-in is live. T comes from the previous alignment of the edge (score per base x this length x
    // ORC_POA_PRUNE_F); an attempt whose best sink stays below T is repeated with T = that score (a real path's, so the repeat cannot fail).
    // Checked here: the pruned matrix gives the same end node and the same traceback as the full one, every time.
    void prune_sim(const PoaGraph& G, const uint8_t* seq, uint32_t len, int CM, int32_t full_score, const std::vector<std::pair<int32_t, int32_t>>& full_aln) {
        static std::atomic<uint64_t> n_aln{0}, n_retry{0}, n_unpruned{0}, n_bad{0}, rows_all{0}, rows_done{0}, rows_retry{0}, mw_all{0}, mw_done{0}, mw_retry{0}, lane_live{0}, lane_all{0};
        static struct Pr { ~Pr() {
            fprintf(stderr, "POAPRUNESIM alignments %lu (first of an edge / no estimate: %lu unpruned), retries %lu, WRONG %lu | wave-rows %.4g computed %.3f (+ retries %.3f) | multi-wave alignments only: wave-rows %.4g computed %.3f (+ retries %.3f)\n",
                    (unsigned long)n_aln.load(), (unsigned long)n_unpruned.load(), (unsigned long)n_retry.load(), (unsigned long)n_bad.load(), (double)rows_all.load(), (double)rows_done.load() / std::max<double>(1, (double)rows_all.load()),
                    (double)rows_retry.load() / std::max<double>(1, (double)rows_all.load()), (double)mw_all.load(), (double)mw_done.load() / std::max<double>(1, (double)mw_all.load()), (double)mw_retry.load() / std::max<double>(1, (double)mw_all.load())); } } printer;
        const size_t V = G.code.size(), W = (size_t)len + 1, BW = 64 * (size_t)CM, NB = (W + BW - 1) / BW;
        const int64_t NEGV = -(1ll << 40), L = len;
        const double f = getenv("ORC_POA_PRUNE_F") ? atof(getenv("ORC_POA_PRUNE_F")) : 0.9;
        n_aln++;
        rows_all += (uint64_t)V * NB; if (NB > 1) mw_all += (uint64_t)V * NB;
        std::vector<int64_t> Hp((V + 1) * W), CI((V + 1) * (NB + 1));
        std::vector<uint8_t> F((V + 1) * NB);
        auto attempt = [&](int64_t T, uint64_t& blocks, int64_t& S, int64_t& Si) {
            std::fill(Hp.begin(), Hp.end(), NEGV); std::fill(F.begin(), F.end(), 0); std::fill(CI.begin(), CI.end(), NEGV);
            blocks = 0;
            for (size_t j = 0; j < W; j++) Hp[j] = (int64_t)j * g;
            for (size_t w = 0; w < NB; w++) {
                const int64_t c0 = (int64_t)(w * BW), cl = std::max<int64_t>(c0 - 1, 0);
                F[w] = (int64_t)m * L - (int64_t)(m - g) * cl >= T;
                CI[w] = w ? (c0 - 1) * g : NEGV;
            }
            S = NEGV; Si = -1;
            for (size_t i = 1; i <= V; i++) {
                const uint32_t n = G.rank2node[i - 1];
                const int32_t* pr = &prof[(size_t)G.code[n] * W];
                const size_t np = G.in[n].size();
                for (size_t w = 0; w < NB; w++) {
                    const int64_t c0 = (int64_t)(w * BW), c1 = std::min<int64_t>((int64_t)W, c0 + (int64_t)BW);
                    const int64_t cin = w ? CI[i * (NB + 1) + w] : NEGV;   // H of column c0 - 1 in this row, as the wave on the left handed it over
                    const bool cin_live = w && cin > NEGV / 2 && (cin - (c0 - 1) * g) + (int64_t)m * L - (int64_t)(m - g) * (c0 - 1) >= T;
                    bool act = cin_live || np > 4;
                    auto rank_of = [&](size_t p) -> size_t { return np == 0 ? 0 : node2rank[G.edges[G.in[n][p]].from] + 1; };
                    for (size_t p = 0; p < std::max<size_t>(1, np) && !act; p++) act = F[rank_of(p) * NB + w];
                    if (!act) { CI[i * (NB + 1) + w + 1] = cin > NEGV / 2 ? cin + g * (c1 - c0) : NEGV; continue; }
                    blocks++;
                    for (int64_t j = c0; j < c1; j++) {
                        int64_t v = NEGV;
                        for (size_t p = 0; p < std::max<size_t>(1, np); p++) {
                            const size_t pi = rank_of(p);
                            if (!F[pi * NB + w]) continue;   // an unflagged predecessor block is not read (it may never have been written)
                            const int64_t left = j == 0 ? NEGV : j == c0 ? CI[pi * (NB + 1) + w] : Hp[pi * W + j - 1];
                            if (j > 0 && left > NEGV / 2) v = std::max(v, left + pr[j]);
                            if (Hp[pi * W + j] > NEGV / 2) v = std::max(v, Hp[pi * W + j] + g);
                        }
                        const int64_t hl = j == c0 ? cin : Hp[i * W + j - 1];
                        if (hl > NEGV / 2) v = std::max(v, hl + g);
                        Hp[i * W + j] = v;
                    }
                    CI[i * (NB + 1) + w + 1] = Hp[i * W + c1 - 1];
                    bool fl = cin_live;
                    for (int64_t j0 = c0; j0 < c1 && !fl; j0 += CM) {
                        const int64_t jl = std::min<int64_t>(j0 + CM - 1, (int64_t)W - 1), h = Hp[i * W + jl];
                        fl = h > NEGV / 2 && (h - jl * g) + (int64_t)m * L - (int64_t)(m - g) * j0 >= T;
                    }
                    F[i * NB + w] = fl;
                }
                if (G.outs[n].empty() && Hp[i * W + W - 1] > S) { S = Hp[i * W + W - 1]; Si = (int64_t)i; }
            }
        };
        uint64_t blocks = 0; int64_t S = 0, Si = -1;
        int64_t T = NEGV;
        const bool est = prev_len > 0;
        if (est) { const double e = (double)prev_score * (double)len / (double)prev_len; T = (int64_t)std::floor(e >= 0 ? e * f : e * (2 - f)); } else n_unpruned++;
        attempt(T, blocks, S, Si);
        rows_done += blocks; if (NB > 1) mw_done += blocks;
        if (S < T) {
            n_retry++;
            attempt(Si >= 0 ? S : NEGV, blocks, S, Si);
            rows_retry += blocks; if (NB > 1) mw_retry += blocks;
        }
        // the pruned matrix must give the reference's alignment: same end node, same walk
        bool ok = S == full_score;
        std::vector<std::pair<int32_t, int32_t>> a2;
        if (ok) {
            size_t i = (size_t)Si, j = W - 1;
            while (!(i == 0 && j == 0)) {
                const int64_t hij = Hp[i * W + j];
                bool found = false; size_t pi_ = 0, pj_ = 0;
                const size_t wj = j / BW;
                auto val = [&](size_t r, size_t c, size_t wave_of_reader) -> int64_t {   // what the walk reads: a block that was never computed holds nothing
                    (void)wave_of_reader; return Hp[r * W + c]; };
                if (i != 0 && j != 0) {
                    const uint32_t n = G.rank2node[i - 1]; const int32_t mc = prof[(size_t)G.code[n] * W + j]; const size_t np = G.in[n].size();
                    for (size_t p = 0; p < std::max<size_t>(1, np) && !found; p++) { const size_t pi = np == 0 ? 0 : node2rank[G.edges[G.in[n][p]].from] + 1; if (hij == val(pi, j - 1, wj) + mc) { pi_ = pi; pj_ = j - 1; found = true; } }
                }
                if (!found && i != 0) {
                    const uint32_t n = G.rank2node[i - 1]; const size_t np = G.in[n].size();
                    for (size_t p = 0; p < std::max<size_t>(1, np) && !found; p++) { const size_t pi = np == 0 ? 0 : node2rank[G.edges[G.in[n][p]].from] + 1; if (hij == val(pi, j, wj) + g) { pi_ = pi; pj_ = j; found = true; } }
                }
                if (!found) { pi_ = i; pj_ = j - 1; }
                a2.emplace_back(i == pi_ ? -1 : (int32_t)G.rank2node[i - 1], j == pj_ ? -1 : (int32_t)(j - 1));
                i = pi_; j = pj_;
            }
            std::reverse(a2.begin(), a2.end());
            ok = a2 == full_aln;
        }
        if (!ok) { n_bad++; fprintf(stderr, "POAPRUNESIM MISMATCH V=%zu L=%u T=%lld S=%lld full=%d\n", V, len, (long long)T, (long long)S, full_score); }
    }
};

inline uint8_t base_at(const hx_reads* R, uint32_t rid, uint32_t i) {
    return (R->packed[R->off[rid] + (i >> 2)] >> ((i & 3) * 2)) & 3;
}

// Assemble.cpp:503-543 for one edge: substring rule of A.8, then POA in stored support order.
int poa_edge(const hx_reads* R, const hx_coords_out* C, uint32_t s, const hx_poa_params* pp, std::string& cns,
             uint64_t* cells, uint64_t* bases, uint64_t* naln) {
    PoaGraph G;
    PoaAligner A{pp->match, pp->mismatch, pp->gap, {}, {}, {}};
    std::vector<uint8_t> seq;
    uint32_t non_empty = 0;
    for (uint64_t k = C->supp_off[s]; k < C->supp_off[s + 1]; k++) {
        uint32_t rid = C->supp_lr[k] & 0x7fffffffu, strand = C->supp_lr[k] >> 31;
        uint32_t rlen = R->len[rid], sp = C->spos[k], ep = C->epos[k];
        if (sp > rlen) return -2;   // std::string::substr would throw here (Assemble.cpp:530)
        uint32_t want = ep - sp + 1;  // u32 arithmetic: wraps when epos+1 < spos, substr then clamps (:530-532)
        uint32_t n = std::min(want, rlen - sp);
        if (n == 0) continue;       // :537
        seq.resize(n);
        for (uint32_t i = 0; i < n; i++)
            seq[i] = strand == 0 ? base_at(R, rid, sp + i) : (uint8_t)(3 - base_at(R, rid, rlen - 1 - (sp + i)));
        auto aln = A.align(G, seq.data(), n, cells);
        G.add_alignment(aln, seq.data(), n);
        non_empty++; *bases += n; (*naln)++;
    }
    if (getenv("ORC_POA_STATS")) {
        uint32_t lmax = 0; uint64_t sum = 0; uint32_t ns = 0;
        for (uint64_t k = C->supp_off[s]; k < C->supp_off[s + 1]; k++) { uint32_t rl = R->len[C->supp_lr[k] & 0x7fffffffu]; uint32_t w = C->epos[k] - C->spos[k] + 1; uint32_t n = std::min(w, rl - C->spos[k]); lmax = std::max(lmax, n); sum += n; ns++; }
        size_t maxin = 0; for (auto& v : G.in) maxin = std::max(maxin, v.size());
        // columns (aligned groups) and predecessor lags in columns: what an anti-diagonal schedule of the DP would see (tools/dev_skewbench.hip)
        size_t ncol = 0, lag1 = 0, lag6 = 0, lagfar = 0, np2 = 0, np3 = 0, np5 = 0;
        {
            G.toposort();
            std::vector<uint32_t> colof(G.code.size(), 0);
            uint32_t c = 0;
            std::vector<uint8_t> seen(G.code.size(), 0);
            for (uint32_t n : G.rank2node) {
                if (seen[n]) continue;
                seen[n] = 1; colof[n] = c;
                for (uint32_t a : G.aligned[n]) { seen[a] = 1; colof[a] = c; }
                c++;
            }
            ncol = c;
            for (size_t n = 0; n < G.code.size(); n++) {
                np2 += G.in[n].size() >= 2; np3 += G.in[n].size() >= 3; np5 += G.in[n].size() >= 5;
                for (uint32_t e : G.in[n]) { const uint32_t d = colof[n] - colof[G.edges[e].from]; lag1 += d == 1; lag6 += d <= 6; lagfar += d > 6; }
            }
        }
        fprintf(stderr, "POASTAT nseq=%u lmax=%u sumL=%lu V=%zu E=%zu maxin=%zu cols=%zu lag1=%zu lag<=6=%zu lag>6=%zu np>=2=%zu np>=3=%zu np>=5=%zu\n", ns, lmax, (unsigned long)sum, G.code.size(), G.edges.size(), maxin,
                ncol, lag1, lag6, lagfar, np2, np3, np5);
    }
    cns.clear();
    if (non_empty == 0) return 0;   // :544-551
    for (uint32_t n : G.consensus()) cns.push_back("ACGT"[G.code[n]]);
    return 0;
}

}  // namespace

extern "C" int orc_poa_batch(const hx_reads* R, const hx_coords_out* C, const hx_poa_params* pp, int n_threads, hx_cns_out* out) {
    memset(out, 0, sizeof(*out));
    uint32_t n = C->n_edge;
    std::vector<std::string> cns(n);
    std::atomic<uint32_t> next{0};
    std::atomic<int> err{0};
    std::atomic<uint64_t> cells{0}, bases{0}, naln{0};
    // edges are handed to the threads costliest first (supports x longest gap squared), so that the longest ones do not start last
    std::vector<uint32_t> order(n);
    std::vector<uint64_t> cost(n, 0);
    for (uint32_t i = 0; i < n; i++) {
        order[i] = i;
        uint64_t lmax = 0, ns = C->supp_off[i + 1] - C->supp_off[i];
        for (uint64_t k = C->supp_off[i]; k < C->supp_off[i + 1]; k++) lmax = std::max<uint64_t>(lmax, (uint32_t)(C->epos[k] - C->spos[k] + 1) & 0xffffffu);
        cost[i] = ns * lmax * lmax;
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
    auto work = [&]() {
        uint64_t lc = 0, lb = 0, la = 0;
        for (;;) {
            uint32_t q = next.fetch_add(1);
            if (q >= n) break;
            const uint32_t s = order[q];
            if (poa_edge(R, C, s, pp, cns[s], &lc, &lb, &la) != 0) err = 1;
        }
        cells += lc; bases += lb; naln += la;
    };
    if (n_threads <= 1) work();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; t++) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    if (err) return fail("orc_poa_batch: spos beyond read length (the reference would throw std::out_of_range)");
    std::vector<uint64_t> off(n + 1, 0);
    for (uint32_t i = 0; i < n; i++) off[i + 1] = off[i] + cns[i].size();
    out->n_edge = n; out->cns_off = dup(off);
    out->cns = (char*)malloc(std::max<uint64_t>(1, off[n]));
    for (uint32_t i = 0; i < n; i++) memcpy(out->cns + off[i], cns[i].data(), cns[i].size());
    out->dp_cells = cells; out->seq_bases = bases; out->n_aligned = naln;
    return 0;
}

extern "C" const char* orc_poa_kernel_name(void) {
    static const std::string n = std::string(row_kernels().name) + (row_kernels16() ? std::string(", ") + row_kernels16()->name + " where 8 (nodes + columns) fits 16 bits" : std::string());
    return n.c_str();
}

extern "C" void orc_free_cns(hx_cns_out* o) { free(o->cns_off); free(o->cns); memset(o, 0, sizeof(*o)); }

extern "C" char* orc_poa_consensus(const char* const* seqs, uint32_t n, const hx_poa_params* pp) {
    PoaGraph G;
    PoaAligner A{pp->match, pp->mismatch, pp->gap, {}, {}, {}};
    uint64_t cells = 0;
    uint32_t non_empty = 0;
    for (uint32_t k = 0; k < n; k++) {
        size_t L = strlen(seqs[k]);
        if (L == 0) continue;
        std::vector<uint8_t> s(L);
        for (size_t i = 0; i < L; i++) { const char* p = strchr("ACGT", seqs[k][i]); s[i] = p ? (uint8_t)(p - "ACGT") : 0; }
        auto aln = A.align(G, s.data(), (uint32_t)L, &cells);
        G.add_alignment(aln, s.data(), (uint32_t)L);
        non_empty++;
    }
    std::string c;
    if (non_empty) for (uint32_t v : G.consensus()) c.push_back("ACGT"[G.code[v]]);
    char* r = (char*)malloc(c.size() + 1);
    memcpy(r, c.c_str(), c.size() + 1);
    return r;
}
extern "C" void orc_free_str(char* p) { free(p); }
