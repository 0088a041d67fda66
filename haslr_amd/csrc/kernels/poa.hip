// poa.hip — K6: per-edge partial-order-alignment consensus.
//
// Replaces the SPOA calls of asm_calc_single_cns_seq (Assemble.cpp:499-554): for every backbone edge the gap
// sub-sequences of its supporting long reads are aligned one after the other (global NW, linear gap,
// +5/-4/-8) to a growing partial-order graph, and the heaviest-bundle path is the consensus. Semantics follow
// the published rvaser/spoa 1.1.3 algorithm as restated in oracle/oracle.cpp (same recurrences, same
// tie-breaking, same graph update and topological order), so results are bit-identical to the oracle.
//
// Round-1 mapping: one workgroup per edge (edges are independent; thousands are in flight).
//   * sequence k is decoded from the 2-bit packed read arena straight into a byte row (coalesced dword loads)
//   * DP: rows = graph nodes in topological order (sequential, data-dependent), columns = sequence positions
//     split into one contiguous chunk per lane. A row is T[j] = max over predecessor rows p of
//     (H[p][j-1]+s, H[p][j]+g) followed by the horizontal recurrence H[j] = max(T[j], H[j-1]+g), which is a
//     prefix-max of T[k]-k*g: lanes scan their chunk serially, chunk ends are combined with a wavefront
//     prefix scan (shuffles) and an LDS exchange between waves, then the carry is applied.
//   * traceback, graph update, topological sort and heaviest bundle are O(V+L) pointer work done by lane 0;
//     the rank-ordered CSR that the DP reads (row code, predecessor ranks, sink flag) is rebuilt by all lanes.
// Full (V+1)x(L+1) int32 score matrix in HBM, as in the reference's engine; banding and LDS-resident row rings
// are the next optimisation steps (DESIGN.md "K6 roadmap").
#include "kernels.h"

namespace hxk {

namespace {

constexpr uint32_t NONE = 0xffffffffu;
constexpr int32_t NEG = -(1 << 29);

struct G {   // per-edge views into the pools
    uint8_t *code, *n_aligned; uint32_t* aligned;
    uint32_t *in_head, *in_tail, *out_head, *out_tail, *rank2node, *node2rank;
    uint8_t *mark, *check; uint32_t* stack;
    int32_t *score, *pred;
    uint8_t *row_code, *row_sink; uint32_t *row_pred_off, *pred_rank;
    uint32_t *e_from, *e_to, *e_next_in, *e_next_out; int32_t* e_w;
    int32_t *aln_node, *aln_pos;
    uint32_t vcap, ecap;
};

__device__ __forceinline__ uint32_t add_node(G& g, uint32_t& V, uint8_t c) {
    uint32_t n = V++;
    g.code[n] = c; g.n_aligned[n] = 0;
    g.in_head[n] = g.in_tail[n] = g.out_head[n] = g.out_tail[n] = NONE;
    return n;
}

// spoa Graph::add_edge: an existing (from,to) edge gains the weight, else a new edge is appended to both lists
__device__ void add_edge(G& g, uint32_t& E, uint32_t f, uint32_t t, int32_t w) {
    for (uint32_t e = g.out_head[f]; e != NONE; e = g.e_next_out[e])
        if (g.e_to[e] == t) { g.e_w[e] += w; return; }
    uint32_t e = E++;
    g.e_from[e] = f; g.e_to[e] = t; g.e_w[e] = w; g.e_next_in[e] = NONE; g.e_next_out[e] = NONE;
    if (g.out_tail[f] == NONE) g.out_head[f] = e; else g.e_next_out[g.out_tail[f]] = e;
    g.out_tail[f] = e;
    if (g.in_tail[t] == NONE) g.in_head[t] = e; else g.e_next_in[g.in_tail[t]] = e;
    g.in_tail[t] = e;
}

// spoa Graph::add_sequence for seq[b,e): returns first node or NONE
__device__ uint32_t add_chain(G& g, uint32_t& V, uint32_t& E, const uint8_t* seq, uint32_t b, uint32_t e) {
    if (b == e) return NONE;
    uint32_t first = add_node(g, V, seq[b]);
    for (uint32_t i = b + 1; i < e; i++) { uint32_t n = add_node(g, V, seq[i]); add_edge(g, E, n - 1, n, 2); }
    return first;
}

// spoa Graph::topological_sort (iterative DFS over in-edges and aligned nodes); lane 0 only
__device__ void toposort(G& g, uint32_t V) {
    uint32_t sp = 0, nr = 0;
    for (uint32_t i = 0; i < V; i++) {
        if (g.mark[i]) continue;
        g.stack[sp++] = i;
        while (sp) {
            uint32_t n = g.stack[sp - 1];
            bool valid = true;
            if (g.mark[n] != 2) {
                for (uint32_t e = g.in_head[n]; e != NONE; e = g.e_next_in[e]) {
                    uint32_t f = g.e_from[e];
                    if (g.mark[f] != 2) { g.stack[sp++] = f; valid = false; }
                }
                if (g.check[n]) {
                    uint32_t na = g.n_aligned[n];
                    for (uint32_t k = 0; k < na; k++) {
                        uint32_t a = g.aligned[3 * n + k];
                        if (g.mark[a] != 2) { g.stack[sp++] = a; g.check[a] = 0; valid = false; }
                    }
                }
                if (valid) {
                    g.mark[n] = 2;
                    if (g.check[n]) {
                        g.rank2node[nr++] = n;
                        uint32_t na = g.n_aligned[n];
                        for (uint32_t k = 0; k < na; k++) g.rank2node[nr++] = g.aligned[3 * n + k];
                    }
                } else g.mark[n] = 1;
            }
            if (valid) sp--;
        }
    }
}

// spoa Graph::add_alignment with unit weights; alignment is stored REVERSED (index n_aln-1 is the first pair). lane 0 only.
// returns false on workspace overflow
__device__ bool add_alignment(G& g, uint32_t& V, uint32_t& E, uint32_t n_aln, const uint8_t* seq, uint32_t len) {
    if (V + len > g.vcap || E + len + 1 > g.ecap) return false;   // worst case: every base a new node / edge
    if (n_aln == 0) { add_chain(g, V, E, seq, 0, len); return true; }
    int32_t first_valid = -1, last_valid = -1;
    for (int32_t k = (int32_t)n_aln - 1; k >= 0; k--) if (g.aln_pos[k] != -1) { first_valid = g.aln_pos[k]; break; }
    for (uint32_t k = 0; k < n_aln; k++) if (g.aln_pos[k] != -1) { last_valid = g.aln_pos[k]; break; }
    uint32_t before = V;
    add_chain(g, V, E, seq, 0, (uint32_t)first_valid);
    uint32_t head = before == V ? NONE : V - 1;
    uint32_t tail = add_chain(g, V, E, seq, (uint32_t)last_valid + 1, len);
    for (int32_t k = (int32_t)n_aln - 1; k >= 0; k--) {
        int32_t pos = g.aln_pos[k];
        if (pos == -1) continue;
        int32_t an = g.aln_node[k];
        uint8_t c = seq[pos];
        uint32_t nn;
        if (an == -1) nn = add_node(g, V, c);
        else if (g.code[an] == c) nn = (uint32_t)an;
        else {
            uint32_t hit = NONE, na = g.n_aligned[an];
            for (uint32_t q = 0; q < na; q++) { uint32_t a = g.aligned[3 * an + q]; if (g.code[a] == c) { hit = a; break; } }
            if (hit == NONE) {
                nn = add_node(g, V, c);
                for (uint32_t q = 0; q < na; q++) {
                    uint32_t a = g.aligned[3 * an + q];
                    g.aligned[3 * nn + g.n_aligned[nn]++] = a;
                    g.aligned[3 * a + g.n_aligned[a]++] = nn;
                }
                g.aligned[3 * nn + g.n_aligned[nn]++] = (uint32_t)an;
                g.aligned[3 * an + g.n_aligned[an]++] = nn;
            } else nn = hit;
        }
        if (head != NONE) add_edge(g, E, head, nn, 2);
        head = nn;
    }
    if (tail != NONE) add_edge(g, E, head, tail, 2);
    return true;
}

// spoa Graph::traverse_heaviest_bundle + branch_completion; lane 0 only. Writes the consensus, returns its length.
__device__ uint32_t consensus(G& g, uint32_t V, char* out) {
    for (uint32_t i = 0; i < V; i++) { g.pred[i] = -1; g.score[i] = -1; }
    uint32_t best = 0;
    for (uint32_t r = 0; r < V; r++) {
        uint32_t n = g.rank2node[r];
        for (uint32_t e = g.in_head[n]; e != NONE; e = g.e_next_in[e]) {
            uint32_t f = g.e_from[e]; int32_t w = g.e_w[e];
            if (g.score[n] < w || (g.score[n] == w && g.score[g.pred[n]] <= g.score[f])) { g.score[n] = w; g.pred[n] = (int32_t)f; }
        }
        if (g.pred[n] != -1) g.score[n] += g.score[g.pred[n]];
        if (g.score[best] < g.score[n]) best = n;
    }
    while (g.out_head[best] != NONE) {   // branch completion
        uint32_t n0 = best;
        for (uint32_t e = g.out_head[n0]; e != NONE; e = g.e_next_out[e])
            for (uint32_t oe = g.in_head[g.e_to[e]]; oe != NONE; oe = g.e_next_in[oe])
                if (g.e_from[oe] != n0) g.score[g.e_from[oe]] = -1;
        int32_t mx = 0; uint32_t mxid = 0;
        for (uint32_t r = g.node2rank[n0] + 1; r < V; r++) {
            uint32_t n = g.rank2node[r];
            g.score[n] = -1; g.pred[n] = -1;
            for (uint32_t e = g.in_head[n]; e != NONE; e = g.e_next_in[e]) {
                uint32_t f = g.e_from[e]; int32_t w = g.e_w[e];
                if (g.score[f] == -1) continue;
                if (g.score[n] < w || (g.score[n] == w && g.score[g.pred[n]] <= g.score[f])) { g.score[n] = w; g.pred[n] = (int32_t)f; }
            }
            if (g.pred[n] != -1) g.score[n] += g.score[g.pred[n]];
            if (mx < g.score[n]) { mx = g.score[n]; mxid = n; }
        }
        best = mxid;
    }
    uint32_t len = 0;
    for (uint32_t n = best;; n = (uint32_t)g.pred[n]) { len++; if (g.pred[n] == -1) break; }
    uint32_t w = len;
    for (uint32_t n = best;; n = (uint32_t)g.pred[n]) { out[--w] = "ACGT"[g.code[n]]; if (g.pred[n] == -1) break; }
    return len;
}

template <int NT>
__device__ __forceinline__ int block_excl_scan_max(int v, int* lds /* NT/64 + 1 */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = wave_scan_max(v);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    int base = NEG;
    for (int i = 0; i < w; i++) base = max(base, lds[i]);
    int prev = __shfl_up(inc, 1, 64);
    int ex = lane == 0 ? NEG : prev;
    return max(base, ex);
}

template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan_add(uint32_t v, uint32_t* lds /* NT/64 */, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_scan_add(v);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int i = 0; i < NT / 64; i++) { if (i < w) base += lds[i]; tot += lds[i]; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

template <int NT>
__global__ void __launch_bounds__(NT) k_poa(const PoaEdge* __restrict__ edges, const uint32_t* __restrict__ order, uint32_t n_edges,
                                            const PoaSeq* __restrict__ seqs, const uint8_t* __restrict__ packed, const uint64_t* __restrict__ read_off,
                                            const uint32_t* __restrict__ read_len, PoaPools P, int32_t match, int32_t mismatch, int32_t gap,
                                            char* cns, uint32_t* cns_len, uint32_t* status, unsigned long long* cells) {
    const uint32_t eidx = order[blockIdx.x];
    const PoaEdge ED = edges[eidx];
    const uint32_t tid = threadIdx.x;
    G g;
    {
        const uint64_t no = ED.node_off, eo = ED.edge_off;
        g.code = P.code + no; g.n_aligned = P.n_aligned + no; g.aligned = P.aligned + 3 * no;
        g.in_head = P.in_head + no; g.in_tail = P.in_tail + no; g.out_head = P.out_head + no; g.out_tail = P.out_tail + no;
        g.rank2node = P.rank2node + no; g.node2rank = P.node2rank + no; g.mark = P.mark + no; g.check = P.check + no;
        g.stack = P.stack + ED.stack_off; g.score = P.score + no; g.pred = P.pred + no;
        g.row_code = P.row_code + no; g.row_sink = P.row_sink + no; g.row_pred_off = P.row_pred_off + no; g.pred_rank = P.pred_rank + eo;
        g.e_from = P.e_from + eo; g.e_to = P.e_to + eo; g.e_next_in = P.e_next_in + eo; g.e_next_out = P.e_next_out + eo; g.e_w = P.e_w + eo;
        g.aln_node = P.aln_node + ED.aln_off; g.aln_pos = P.aln_pos + ED.aln_off;
        g.vcap = ED.vcap; g.ecap = ED.ecap;
    }
    int32_t* H = P.H + ED.h_off;
    uint8_t* seq = P.seq + ED.seq_off;
    const uint32_t W = ED.lmax + 1;

    __shared__ int lds_i[NT / 64 + 1];
    __shared__ uint32_t lds_u[NT / 64];
    __shared__ uint32_t sV, sE, sNaln, sOk;
    __shared__ int sBestScore, sBestI;
    if (tid == 0) { sV = 0; sE = 0; sOk = 1; }
    __syncthreads();

    for (uint32_t k = ED.seq_begin; k < ED.seq_end; k++) {
        const PoaSeq q = seqs[k];
        const uint32_t L = q.len;
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
        const uint32_t V = sV;
        if (V > 0) {
            // =================================================== DP over (rank, column)
            const uint32_t ncol = L + 1;
            const uint32_t C = (ncol + NT - 1) / NT;
            const uint32_t j0 = tid * C, j1 = min(j0 + C, ncol);
            for (uint32_t j = j0; j < j1; j++) H[j] = (int32_t)j * gap;   // row 0
            int bestScore = INT32_MIN + 1024, bestI = -1;                // tracked by the lane owning column L
            __syncthreads();
            for (uint32_t i = 1; i <= V; i++) {
                const uint32_t rc = g.row_code[i - 1];
                const uint32_t po = g.row_pred_off[i - 1], pe = g.row_pred_off[i];
                int32_t* row = H + (uint64_t)i * W;
                int run = NEG, endv = NEG;
                if (j0 < j1) {
                    // pass 1: T[j] and the chunk-local horizontal recurrence
                    for (uint32_t j = j0; j < j1; j++) {
                        int t;
                        if (j == 0) {
                            if (po == pe) t = gap;
                            else { t = NEG; for (uint32_t p = po; p < pe; p++) t = max(t, H[(uint64_t)(g.pred_rank[p] + 1) * W]); t += gap; }
                            run = t;
                        } else {
                            const int sc = seq[j - 1] == rc ? match : mismatch;
                            if (po == pe) t = max(H[j - 1] + sc, H[j] + gap);
                            else {
                                t = NEG;
                                for (uint32_t p = po; p < pe; p++) {
                                    const int32_t* pw = H + (uint64_t)(g.pred_rank[p] + 1) * W;
                                    t = max(t, max(pw[j - 1] + sc, pw[j] + gap));
                                }
                            }
                            run = max(t, run + gap);
                        }
                        row[j] = run;
                    }
                    endv = run - (int)(j1 - 1) * gap;
                }
                // pass 2: carry from the columns to the left (prefix max of T[k]-k*g)
                int ex = block_excl_scan_max<NT>(endv, lds_i);
                if (j0 < j1 && j0 > 0 && ex > NEG / 2) {
                    for (uint32_t j = j0; j < j1; j++) {
                        int viaLeft = ex + (int)j * gap;
                        if (viaLeft > row[j]) row[j] = viaLeft; else break;   // once the chunk-local value wins it wins for the rest
                    }
                }
                if (j1 == ncol && j0 < j1 && g.row_sink[i - 1]) {
                    int v = row[L];
                    if (bestScore < v) { bestScore = v; bestI = (int)i; }
                }
                __syncthreads();   // row i complete and visible; lds_i free again
            }
            if (j1 == ncol && j0 < j1) { sBestScore = bestScore; sBestI = bestI; }
            __syncthreads();
            // =================================================== traceback (lane 0), stored reversed
            if (tid == 0) {
                atomicAdd(cells, (unsigned long long)V * L);
                uint32_t i = (uint32_t)sBestI, j = L, na = 0;
                while (!(i == 0 && j == 0)) {
                    const int hij = H[(uint64_t)i * W + j];
                    uint32_t pi_ = i, pj_ = j;
                    bool found = false;
                    uint32_t po = 0, pe = 0;
                    if (i != 0) { po = g.row_pred_off[i - 1]; pe = g.row_pred_off[i]; }
                    if (i != 0 && j != 0) {
                        const int mc = seq[j - 1] == g.row_code[i - 1] ? match : mismatch;
                        if (po == pe) { if (hij == H[j - 1] + mc) { pi_ = 0; pj_ = j - 1; found = true; } }
                        else for (uint32_t p = po; p < pe && !found; p++) {
                            uint32_t pr = g.pred_rank[p] + 1;
                            if (hij == H[(uint64_t)pr * W + j - 1] + mc) { pi_ = pr; pj_ = j - 1; found = true; }
                        }
                    }
                    if (!found && i != 0) {
                        if (po == pe) { if (hij == H[j] + gap) { pi_ = 0; pj_ = j; found = true; } }
                        else for (uint32_t p = po; p < pe && !found; p++) {
                            uint32_t pr = g.pred_rank[p] + 1;
                            if (hij == H[(uint64_t)pr * W + j] + gap) { pi_ = pr; pj_ = j; found = true; }
                        }
                    }
                    if (!found) { pi_ = i; pj_ = j - 1; }
                    g.aln_node[na] = i == pi_ ? -1 : (int32_t)g.rank2node[i - 1];
                    g.aln_pos[na] = j == pj_ ? -1 : (int32_t)(j - 1);
                    na++;
                    i = pi_; j = pj_;
                }
                sNaln = na;
            }
        } else if (tid == 0) sNaln = 0;
        __syncthreads();
        // =================================================== graph update + topological sort
        {
            uint32_t Vn = sV;   // marks are cleared for the node count AFTER the update; clear generously up to V+L
            uint32_t lim = min(Vn + L, g.vcap);
            for (uint32_t i = tid; i < lim; i += NT) { g.mark[i] = 0; g.check[i] = 1; }
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t V2 = sV, E2 = sE;
            if (!add_alignment(g, V2, E2, sNaln, seq, L)) sOk = 0;
            else { toposort(g, V2); sV = V2; sE = E2; }
        }
        __syncthreads();
        if (!sOk) break;
        // =================================================== rank-order CSR for the next DP (all lanes)
        {
            const uint32_t V2 = sV;
            for (uint32_t r = tid; r < V2; r += NT) g.node2rank[g.rank2node[r]] = r;
            __syncthreads();
            const uint32_t CH = (V2 + NT - 1) / NT;
            const uint32_t r0 = min(tid * CH, V2), r1 = min(r0 + CH, V2);
            uint32_t cnt = 0;
            for (uint32_t r = r0; r < r1; r++) {
                uint32_t n = g.rank2node[r];
                for (uint32_t e = g.in_head[n]; e != NONE; e = g.e_next_in[e]) cnt++;
            }
            uint32_t tot;
            uint32_t off = block_excl_scan_add<NT>(cnt, lds_u, &tot);
            for (uint32_t r = r0; r < r1; r++) {
                uint32_t n = g.rank2node[r];
                g.row_pred_off[r] = off;
                g.row_code[r] = g.code[n];
                g.row_sink[r] = g.out_head[n] == NONE;
                for (uint32_t e = g.in_head[n]; e != NONE; e = g.e_next_in[e]) g.pred_rank[off++] = g.node2rank[g.e_from[e]];
            }
            if (tid == NT - 1) g.row_pred_off[V2] = tot;
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (!sOk) { status[eidx] = HXE_POA_OVERFLOW; cns_len[eidx] = 0; }
        else { status[eidx] = 0; cns_len[eidx] = sV ? consensus(g, sV, cns + ED.cns_off) : 0; }
    }
}

}  // namespace

void poa_run(const PoaEdge* edges, const uint32_t* order, uint32_t n_edges, const PoaSeq* seqs, const uint8_t* packed, const uint64_t* read_off,
             const uint32_t* read_len, PoaPools pools, uint64_t, int32_t match, int32_t mismatch, int32_t gap, char* cns, uint32_t* cns_len,
             uint32_t* status, unsigned long long* cells, int block_threads, hipStream_t s) {
    if (!n_edges) return;
    if (block_threads >= 512)
        k_poa<512><<<n_edges, 512, 0, s>>>(edges, order, n_edges, seqs, packed, read_off, read_len, pools, match, mismatch, gap, cns, cns_len, status, cells);
    else if (block_threads >= 256)
        k_poa<256><<<n_edges, 256, 0, s>>>(edges, order, n_edges, seqs, packed, read_off, read_len, pools, match, mismatch, gap, cns, cns_len, status, cells);
    else
        k_poa<64><<<n_edges, 64, 0, s>>>(edges, order, n_edges, seqs, packed, read_off, read_len, pools, match, mismatch, gap, cns, cns_len, status, cells);
}

}  // namespace hxk
