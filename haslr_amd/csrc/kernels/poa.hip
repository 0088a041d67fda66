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

constexpr uint32_t NONE = 0xffffffffu;
constexpr int32_t NEG = -(1 << 29);

constexpr uint32_t META_SLOT = 8, META_NP = 12;   // row record: ring slot the row is written to (4 bits; 15 = not written), number of predecessors
struct G {   // per-edge views into the pools
    uint8_t *code, *n_aligned; uint32_t* aligned;
    uint32_t *in_head, *in_tail, *out_head, *out_tail, *rank2node, *node2rank;
    uint8_t *mark, *check; uint32_t* stack;
    int32_t *score, *pred;
    uint8_t *row_code, *row_sink; uint32_t *row_pred_off, *pred_rank;
    uint32_t *row_meta, *row_pred0, *row_pred1;   // per rank: code | sink<<2 | far<<3 | kept<<4 | wide<<5 | own ring slot<<8 | npred<<12 (META_SLOT, META_NP) ; ranks of the first two predecessors
    uint16_t* row_al;                             // per rank: aligned nodes in list order as rank deltas (3 x 3 bits, delta + 4, 0 = none)
    uint32_t* wslot;                              // per rank: row of the wide-row pool (rows with more than 4 predecessors: a direction byte per cell)
    int32_t* pred_w;                              // per entry of pred_rank: weight of that in-edge
    uint4* nrec;   // per node, one 16-byte record for the serial graph walks: {1st in-edge source, 2nd in-edge source, 3 aligned ids (+1) x 21 bit, bit 63: more in-edges}
    uint4* nrec2;  // ... and a second one (round 5, the CSR rebuild): {2nd out-edge, 2nd in-edge (edge ids: where a walk of the third and later list entries starts), 1st out-edge target | bit 31: more than two out-edges, 2nd out-edge target}; NONE = no entry
    uint32_t *e_from, *e_to, *e_next_in, *e_next_out; int32_t* e_w;
    int32_t *aln_node, *aln_pos;
    uint32_t vcap, ecap;
};

__device__ __forceinline__ uint32_t add_node(G& g, uint32_t& V, uint8_t c) {
    uint32_t n = V++;
    g.code[n] = c; g.n_aligned[n] = 0;
    g.in_head[n] = g.in_tail[n] = g.out_head[n] = g.out_tail[n] = NONE;
    g.nrec[n] = make_uint4(NONE, NONE, 0u, 0u);
    g.nrec2[n] = make_uint4(NONE, NONE, NONE, NONE);
    return n;
}

// a new edge e = (f -> t) joins f's out-list and t's in-list, and the two nodes' records (the lists' first two entries)
__device__ __forceinline__ void link_edge(G& g, const uint32_t e, const uint32_t f, const uint32_t t) {
    uint32_t* rf = reinterpret_cast<uint32_t*>(&g.nrec2[f]);
    if (g.out_tail[f] == NONE) { g.out_head[f] = e; rf[2] = t; }
    else { g.e_next_out[g.out_tail[f]] = e; if (rf[3] == NONE) { rf[3] = t; rf[0] = e; } else rf[2] |= 0x80000000u; }   // third and later out-edges: walk the list (from the second edge on)
    g.out_tail[f] = e;
    uint32_t* r = reinterpret_cast<uint32_t*>(&g.nrec[t]);
    uint32_t* r2 = reinterpret_cast<uint32_t*>(&g.nrec2[t]);
    if (g.in_tail[t] == NONE) { g.in_head[t] = e; r[0] = f; }
    else { g.e_next_in[g.in_tail[t]] = e; if (r[1] == NONE) { r[1] = f; r2[1] = e; } else r[3] |= 0x80000000u; }   // third and later in-edges: walk the list
    g.in_tail[t] = e;
}

// spoa Graph::add_edge: an existing (from,to) edge gains the weight, else a new edge is appended to both lists
__device__ void add_edge(G& g, uint32_t& E, uint32_t f, uint32_t t, int32_t w) {
    for (uint32_t e = g.out_head[f]; e != NONE; e = g.e_next_out[e])
        if (g.e_to[e] == t) { g.e_w[e] += w; return; }
    uint32_t e = E++;
    g.e_from[e] = f; g.e_to[e] = t; g.e_w[e] = w; g.e_next_in[e] = NONE; g.e_next_out[e] = NONE;
    link_edge(g, e, f, t);
}

// append node `a` to node n's aligned list (array form + the packed copy in the node record)
__device__ __forceinline__ void push_aligned(G& g, uint32_t n, uint32_t a) {
    const uint32_t k = g.n_aligned[n]++;
    g.aligned[3 * n + k] = a;
    uint32_t* r = reinterpret_cast<uint32_t*>(&g.nrec[n]);
    unsigned long long packed = (unsigned long long)r[2] | ((unsigned long long)r[3] << 32);
    packed |= (unsigned long long)(a + 1) << (21 * k);   // ids are stored +1 so that 0 means "no entry"
    r[2] = (uint32_t)packed; r[3] = (uint32_t)(packed >> 32);
}

// spoa Graph::add_sequence for seq[b,e): returns first node or NONE
__device__ uint32_t add_chain(G& g, uint32_t& V, uint32_t& E, const uint8_t* seq, uint32_t b, uint32_t e, uint32_t* path, uint32_t* colref) {
    if (b == e) return NONE;
    uint32_t first = add_node(g, V, seq[b]);
    path[b] = first; colref[b] = NONE;
    for (uint32_t i = b + 1; i < e; i++) { uint32_t n = add_node(g, V, seq[i]); path[i] = n; colref[i] = NONE; add_edge(g, E, n - 1, n, 2); }
    return first;
}

// spoa Graph::topological_sort (iterative DFS over in-edges and aligned nodes); lane 0 only
__device__ void toposort(G& g, uint32_t V, uint32_t* out) {
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
                        out[nr++] = n;
                        uint32_t na = g.n_aligned[n];
                        for (uint32_t k = 0; k < na; k++) out[nr++] = g.aligned[3 * n + k];
                    }
                } else g.mark[n] = 1;
            }
            if (valid) sp--;
        }
    }
}


constexpr uint32_t SINK_CAP = 1024;      // sink rows whose end score is kept per alignment (more: error)
// spoa's traversal (toposort above) by one whole wavefront in lock step - every lane computes the same scalars, the idle ones fetch records cooperatively
// through a direct-mapped LDS cache of 16-record lines; the mark / check bits of every node sit in one LDS byte, the top of the DFS stack in an LDS window
// that spills to the HBM stack - on RANKS of the order the DP maintains (any valid topological order with contiguous columns): predecessors have
// smaller, nearby ranks, so the 16-rank record lines (row_meta, first two predecessor ranks, aligned-rank deltas) hit the LDS cache
// almost always — node ids are visited in a scattered order, ranks are not. Roots are still taken in node-id order (that is what
// fixes the reference's result); out[] receives ranks, the caller maps them back to node ids.
__device__ void toposort_rank(G& g, const uint32_t V, uint8_t* st /* by rank; LDS, or global memory when the graph is larger than the LDS left */, uint32_t* lstack, uint4* cache, uint32_t* tags, uint32_t* out,
                              const uint32_t LCAP /* entries of the stack window */, const uint32_t LINES /* lines of the record cache: powers of two both */) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = lane; i < LINES; i += 64) tags[i] = NONE;
    uint32_t sp = 0, nr = 0, base = 0;
    uint32_t rootV = 0;
    for (uint32_t i = 0; i < V; i++) {
        if ((i & 63u) == 0) rootV = i + lane < V ? g.node2rank[i + lane] : 0;
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)rootV, (int)(i & 63u));
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)st[r0]) & 3u) continue;
        if (lane == 0) lstack[sp & (LCAP - 1)] = r0;
        sp++;
        while (sp) {
            if (sp == base) { base--; if (lane == 0) lstack[base & (LCAP - 1)] = g.stack[base]; }
            const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)lstack[(sp - 1) & (LCAP - 1)]);
            const uint32_t line = n >> 4, slot = line & (LINES - 1);
            const uint32_t tg = (uint32_t)__builtin_amdgcn_readfirstlane((int)tags[slot]);
            const uint32_t sn = (uint32_t)__builtin_amdgcn_readfirstlane((int)st[n]);
            if ((sn & 3u) == 2u) { sp--; continue; }     // pushed more than once, finished meanwhile
            if (tg != line) {
                if (lane < 16) {
                    const uint32_t id = (line << 4) + lane;
                    cache[slot * 16 + lane] = id < V ? make_uint4(g.row_meta[id], g.row_pred0[id], g.row_pred1[id], (uint32_t)g.row_al[id]) : make_uint4(0u, 0u, 0u, 0u);
                }
                if (lane == 0) tags[slot] = line;
            }
            const uint4 rv = cache[slot * 16 + (n & 15u)];
            const uint32_t npred = (uint32_t)__builtin_amdgcn_readfirstlane((int)rv.x) >> META_NP;
            const uint32_t alp = (uint32_t)__builtin_amdgcn_readfirstlane((int)rv.w);
            const bool chk = sn & 4u;
            const uint32_t spb = sp;
            const bool again = (sn & 3u) == 1u;   // second visit: everything this node pushed has been finished (LIFO, no cycles), nothing to check
            if (!again && npred > 2) {   // three or more in-edges: the list (rare)
                const uint32_t po = g.row_pred_off[n];
                for (uint32_t p = 0; p < npred; p++) {
                    const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.pred_rank[po + p]) & 0x0fffffffu;
                    if (((uint32_t)__builtin_amdgcn_readfirstlane((int)st[f]) & 3u) != 2u) {
                        if (sp - base == LCAP) { if (lane == 0) g.stack[base] = lstack[base & (LCAP - 1)]; base++; }
                        if (lane == 0) lstack[sp & (LCAP - 1)] = f;
                        sp++;
                    }
                }
            }
            // candidates of the lanes: 0/1 the first two in-edge sources (unless the list was walked), 2..4 the aligned ranks (only if the node still checks its column)
            uint32_t cand = NONE;
            if (lane == 0 && npred >= 1 && npred <= 2) cand = rv.y & 0x0fffffffu;
            else if (lane == 1 && npred == 2) cand = rv.z & 0x0fffffffu;
            else if (lane >= 2 && lane < 5 && chk) { const uint32_t d = (alp >> (3 * (lane - 2))) & 7u; cand = d ? n + d - 4u : NONE; }
            const bool todo = !again && cand != NONE && (st[cand] & 3u) != 2u;
            const unsigned long long tm = again ? 0ull : __ballot(todo);
            const uint32_t npush = (uint32_t)__popcll(tm);
            if (npush) {
                while (sp + npush - base > LCAP) { if (lane == 0) g.stack[base] = lstack[base & (LCAP - 1)]; base++; }   // make room in the LDS window
                if (todo) {
                    const uint32_t pos = sp + (uint32_t)__popcll(tm & ((1ull << lane) - 1));
                    lstack[pos & (LCAP - 1)] = cand;
                    if (lane >= 2) st[cand] &= (uint8_t)~4u;   // an aligned node reached from its column does not check the column again
                }
                sp += npush;
            }
            if (sp == spb) {   // every predecessor and column member is final: so is this node
                if (lane == 0) st[n] = (uint8_t)((sn & ~3u) | 2u);
                if (chk) {
                    if (lane == 0) out[nr] = n;
                    if (lane >= 2 && lane < 5) { const uint32_t d = (alp >> (3 * (lane - 2))) & 7u; if (d) out[nr + lane - 1] = n + d - 4u; }
                    nr += 1 + ((alp & 7u) != 0) + ((alp & 0x38u) != 0) + ((alp & 0x1c0u) != 0);
                }
                sp--;
            } else if (lane == 0) st[n] = (uint8_t)((sn & ~3u) | 1u);
        }
    }
}

// spoa Graph::add_alignment with unit weights; alignment is stored REVERSED (index n_aln-1 is the first pair). lane 0 only.
// returns false on workspace overflow
__device__ bool add_alignment(G& g, uint32_t& V, uint32_t& E, uint32_t n_aln, const uint8_t* seq, uint32_t len, uint32_t* path /* node of every base */,
                              uint32_t* colref /* an OLD node of the aligned column the base went to, NONE for an unaligned base */) {
    if (V + len > g.vcap || E + len + 1 > g.ecap) return false;   // worst case: every base a new node / edge
    if (n_aln == 0) { add_chain(g, V, E, seq, 0, len, path, colref); return true; }
    int32_t first_valid = -1, last_valid = -1;
    for (int32_t k = (int32_t)n_aln - 1; k >= 0; k--) if (g.aln_pos[k] != -1) { first_valid = g.aln_pos[k]; break; }
    for (uint32_t k = 0; k < n_aln; k++) if (g.aln_pos[k] != -1) { last_valid = g.aln_pos[k]; break; }
    uint32_t before = V;
    add_chain(g, V, E, seq, 0, (uint32_t)first_valid, path, colref);
    uint32_t head = before == V ? NONE : V - 1;
    uint32_t tail = add_chain(g, V, E, seq, (uint32_t)last_valid + 1, len, path, colref);
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
                    push_aligned(g, nn, a);
                    push_aligned(g, a, nn);
                }
                push_aligned(g, nn, (uint32_t)an);
                push_aligned(g, (uint32_t)an, nn);
            } else nn = hit;
        }
        path[pos] = nn; colref[pos] = an == -1 ? NONE : (uint32_t)an;
        if (head != NONE) add_edge(g, E, head, nn, 2);
        head = nn;
    }
    if (tail != NONE) add_edge(g, E, head, tail, 2);
    return true;
}

// The forward pass of the heaviest bundle on ANY valid topological order gives the same scores and predecessors (a node looks only at its
// in-edges, in in-edge order). The order matters in two places: which of several equally heavy nodes is taken as the end ("first in rank order"),
// and the branch completion that follows when that node is not a sink. So: run the pass on the order the DP maintains; if the heaviest node is
// unique and a sink, the walk back from it IS the reference's consensus. Otherwise return NONE and let the caller sort the graph the reference's way.
// It is made by one whole wavefront on the rank-ordered rows of the last CSR build (row_pred_off / row_pred0 / row_pred1 / pred_rank, pred_w):
// a single lane walking the node lists pays 5-6 dependent HBM round trips per node (25-50 M cycles on a 20 000-node graph: 4-8 % of the longest edges). Here 64
// consecutive ranks are taken at a time: every lane fetches its row and folds the predecessors that lie BEFORE the chunk (their scores are final:
// independent loads, one round trip for the chunk), then the chunk is finished rank by rank with the scores of the predecessors inside it read from
// a register (ds_bpermute). The fold "take the edge if it is heavier, or as heavy and its source scores at least as much" (spoa's <=: the later
// in-edge wins a tie) is the maximum of (weight, source score, position in the in-edge list), so the two halves can be folded in any order.
// Returns the consensus length, or NONE when the heaviest node is not a unique sink (the caller then sorts the graph the reference's way).
// One forward pass over the ranks [r_begin, V) of the rank-ordered rows, by one wavefront. `restricted` = the pass of spoa's branch completion:
// in-edges from nodes whose score is -1 do not count. Returns through best / nbest the first rank whose score exceeds `floor_score` and every
// later maximum (strictly greater moves it, equal counts it).
__device__ void bundle_pass(G& g, const uint32_t V, const uint32_t r_begin, const bool restricted, const int32_t floor_score, uint32_t& best, uint32_t& nbest) {
    const uint32_t lane = threadIdx.x & 63u;
    int32_t* sc_r = g.score;          // by rank
    int32_t* pr_r = g.pred;           // by rank: rank of the chosen predecessor, -1 = none
    best = NONE; nbest = 0;
    int32_t bscore = floor_score;
    for (uint32_t r0 = r_begin & ~63u; r0 < V; r0 += 64) {
        const uint32_t r = r0 + lane;
        const uint32_t lim = max(r0, r_begin);                  // predecessors below this rank are final: their scores come from memory
        const bool valid = r < V && r >= r_begin;
        const uint32_t np = valid ? g.row_meta[r] >> META_NP : 0u, off = valid ? g.row_pred_off[r] : 0u;
        // the first four in-edges in registers (rank, weight); more than four: the list is walked again where needed (rare)
        uint32_t ep[4]; int32_t ew[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { ep[k] = (uint32_t)k < np ? g.pred_rank[off + k] & 0x0fffffffu : NONE; ew[k] = (uint32_t)k < np ? g.pred_w[off + k] : 0; }
        int32_t bw = -1, bs = 0, bp = -1; uint32_t bk = 0;     // best candidate so far: weight, its source's score, its source's rank, its position
        auto take = [&](int32_t w, int32_t s, uint32_t p, uint32_t k) {
            if (restricted && s == -1) return;
            if (w > bw || (w == bw && (s > bs || (s == bs && k >= bk)))) { bw = w; bs = s; bp = (int32_t)p; bk = k; }
        };
        for (uint32_t k = 0; k < np; k++) {                     // predecessors before the chunk
            const uint32_t p = k < 4 ? ep[k] : g.pred_rank[off + k] & 0x0fffffffu;
            if (p < lim) take(k < 4 ? ew[k] : g.pred_w[off + k], sc_r[p], p, k);
        }
        int32_t sc = -1;
        const uint32_t l0 = lim - r0, nv = min(64u, V - r0);
        for (uint32_t l = l0; l < nv; l++) {                    // the chunk, rank by rank (wave-uniform loop; lane l is the one that finishes)
            const uint32_t npl = (uint32_t)__builtin_amdgcn_readlane((int)np, (int)l);
            for (uint32_t k = 0; k < npl; k++) {
                uint32_t p = NONE; int32_t w = 0;
                if (lane == l) { p = k < 4 ? ep[k < 4 ? k : 0] : g.pred_rank[off + k] & 0x0fffffffu; w = k < 4 ? ew[k < 4 ? k : 0] : g.pred_w[off + k]; }
                const bool inside = lane == l && p >= lim && p != NONE;
                const int32_t s = __shfl(sc, inside ? (int)(p - r0) : 0);   // (every lane takes part in the exchange)
                if (inside) take(w, s, p, k);
            }
            if (lane == l) sc = bp == -1 ? -1 : bw + bs;
            const int32_t sl = __builtin_amdgcn_readlane(sc, (int)l);
            if (sl > bscore) { best = r0 + l; bscore = sl; nbest = 1; }
            else if (sl == bscore) nbest++;
        }
        if (valid) { sc_r[r] = sc; pr_r[r] = bp; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// the walk back from rank `best` along the chosen predecessors; lane 0 writes, every lane gets the length
__device__ uint32_t bundle_backtrack(G& g, const uint32_t best, char* out) {
    const int32_t* pr_r = g.pred;
    uint32_t len = 0;
    if ((threadIdx.x & 63u) == 0) {
        for (int32_t r = (int32_t)best; r != -1; r = pr_r[r]) len++;
        uint32_t w = len;
        for (int32_t r = (int32_t)best; r != -1; r = pr_r[r]) out[--w] = "ACGT"[g.row_meta[r] & 3u];
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)len);
}

__device__ uint32_t consensus_fast_wave(G& g, const uint32_t V, char* out) {
    uint32_t best, nbest;
    bundle_pass(g, V, 0, false, -2, best, nbest);               // (every score is >= -1: the first rank opens the maximum)
    if (best == NONE || nbest != 1 || !(g.row_meta[best] & 4u)) return NONE;    // (bit 2 of a row record: the node has no out-edge)
    return bundle_backtrack(g, best, out);
}

// spoa Graph::traverse_heaviest_bundle + branch_completion on the REFERENCE's topological order (rank2node / node2rank hold it, the rank-ordered
// rows have been rebuilt for it: k_poa's bundle_rows), by one wavefront. The reference starts with best = node 0 and moves it to every node that
// scores strictly more, in rank order; its branch completion does the same from (0, node 0).
__device__ uint32_t consensus_wave(G& g, const uint32_t V, char* out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t best, nbest;
    bundle_pass(g, V, 0, false, -1, best, nbest);
    if (best == NONE) best = g.node2rank[0];
    for (uint32_t round = 0; !(g.row_meta[best] & 4u) && round <= V; round++) {   // branch completion (the bound only guards against a cycle the reference would hang in)
        const uint32_t n0 = g.rank2node[best];
        if (lane == 0)
            for (uint32_t e = g.out_head[n0]; e != NONE; e = g.e_next_out[e])
                for (uint32_t oe = g.in_head[g.e_to[e]]; oe != NONE; oe = g.e_next_in[oe])
                    if (g.e_from[oe] != n0) g.score[g.node2rank[g.e_from[oe]]] = -1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint32_t nb;
        bundle_pass(g, V, best + 1, true, 0, nb, nbest);
        best = nb == NONE ? g.node2rank[0] : nb;
    }
    return bundle_backtrack(g, best, out);
}

__device__ __forceinline__ int block_excl_scan_max(int v, int* lds /* blockDim/64 */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = wave_scan_max(v);
    if (lane == 63) lds[w] = inc;
    barrier_lds_only();
    // wave totals (<= 16): one LDS read per lane, a 16-lane DPP row scan, and a scalar read of entry w-1
    const int nw = blockDim.x >> 6;
    int tot = (lane & 15) < nw ? lds[lane & 15] : NEG;
    int x = tot;
    x = max(x, __builtin_amdgcn_update_dpp(NEG, tot, 0x111, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(NEG, tot, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(NEG, tot, 0x113, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(NEG, x, 0x114, 0xf, 0xe, false));
    x = max(x, __builtin_amdgcn_update_dpp(NEG, x, 0x118, 0xf, 0xc, false));
    const int base = w == 0 ? NEG : __builtin_amdgcn_readlane(x, w - 1);
    return max(base, wave_shift_up1(inc, NEG));
}

// a pointer every lane holds the same value of, as a scalar
template <class T> __device__ __forceinline__ T* uptr(T* p) {
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return reinterpret_cast<T*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
}
// inclusive prefix sum over the 64 lanes of a wave: the DPP sequence of wave_scan_max with an addition (no LDS round trips)
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v) {
    uint32_t x = v;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, false);   // row_shr:3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xe, false);   // row_shr:4 bank_mask:0xe
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xc, false);   // row_shr:8 bank_mask:0xc
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 row_mask:0xa
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 row_mask:0xc
    return x;
}
// Exclusive prefix sum over the workgroup. Its barriers wait for LDS only: the loops of the graph phases call it once per block of ranks / bases, between
// their stores - with __syncthreads (which drains the wave's outstanding global stores first, ~2 us under load) the scans WERE those loops' time.
// A one-wave workgroup meets no barrier at all. Nothing here orders global memory: callers that hand data to other lanes through it synchronise themselves.
__device__ __forceinline__ uint32_t block_excl_scan_add(uint32_t v, uint32_t* lds /* blockDim/64 */, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t inc = wave_incl_add(v);
    if (nw == 1) { *total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63); return inc - v; }
    if (lane == 63) lds[w] = inc;
    barrier_lds_only();
    uint32_t base = 0, tot = 0;
    for (uint32_t i = 0; i < nw; i++) { const uint32_t x = lds[i]; if (i < w) base += x; tot += x; }
    barrier_lds_only();
    *total = tot;
    return base + inc - v;
}



// contiguous per-lane chunk stores/loads as single wide memory instructions (rows are padded to 16 columns, chunks are CM-aligned)
template <int CM> __device__ __forceinline__ void store_chunk_i32(int32_t* p, const int (&v)[CM]) {
    if constexpr (CM >= 4) {
#pragma unroll
        for (int q = 0; q < CM / 4; q++) reinterpret_cast<int4*>(p)[q] = make_int4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else if constexpr (CM == 2) *reinterpret_cast<int2*>(p) = make_int2(v[0], v[1]);
    else p[0] = v[0];
}
template <int CM> __device__ __forceinline__ void load_chunk_i32(const int32_t* p, int (&v)[CM]) {
    if constexpr (CM >= 4) {
#pragma unroll
        for (int q = 0; q < CM / 4; q++) { const int4 x = reinterpret_cast<const int4*>(p)[q]; v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w; }
    } else if constexpr (CM == 2) { const int2 x = *reinterpret_cast<const int2*>(p); v[0] = x.x; v[1] = x.y; }
    else v[0] = p[0];
}
template <int CM> __device__ __forceinline__ void store_chunk_u8(uint8_t* p, const uint32_t (&v)[CM]) {
    if constexpr (CM >= 4) {
        uint32_t w[CM / 4];
#pragma unroll
        for (int q = 0; q < CM / 4; q++) w[q] = (v[4 * q] & 0xffu) | ((v[4 * q + 1] & 0xffu) << 8) | ((v[4 * q + 2] & 0xffu) << 16) | (v[4 * q + 3] << 24);
        if constexpr (CM == 4) *reinterpret_cast<uint32_t*>(p) = w[0];
        else if constexpr (CM == 8) *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
        else {
#pragma unroll
            for (int q = 0; q < CM / 16; q++) reinterpret_cast<uint4*>(p)[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
        }
    } else if constexpr (CM == 2) *reinterpret_cast<uint16_t*>(p) = (uint16_t)((v[0] & 0xffu) | (v[1] << 8));
    else p[0] = (uint8_t)v[0];
}

// ---------------------------------------------------------------------------------------------------
// DP over (rank, column) for one sequence against the current graph — rows live in registers, the waves of an edge form a pipeline.
//
// Lane t of wave w owns the CM contiguous columns [(64 w + t) CM, (64 w + t + 1) CM) (w counts through all workgroups that share
// the edge) and keeps the CURRENT row there. The common predecessor of row i is row i-1: the lane's own registers, plus the value left
// of its first column, which falls out of the prefix scan that row i-1 needed anyway. So the usual row costs no LDS row traffic and
// exactly ONE cross-lane operation: the 64-lane DPP prefix-max scan of (chunk end - column*gap) that resolves the horizontal
// recurrence H[j] = max(T[j], H[j-1]+g) inside the wave.
//
// What crosses a wave boundary is one number per row: the prefix maximum through the wave's last column. Round 1 exchanged it with a
// workgroup barrier per row (all waves in lock step: 65 % of the wave cycles were spent parked). Now every wave runs at its own pace
// and is a stage of a pipeline: it publishes the carry of every finished row to a mailbox — a small tagged ring in LDS towards the
// next wave of the workgroup, a tagged word per row in HBM towards the first wave of the next workgroup ("member") — and takes its
// own carries from the wave on its left, 32 rows at a time (one coalesced read, lane r = row r of the batch, broadcast per row with
// v_readlane like the row records). A wave therefore runs one batch behind its left neighbour, polls once per 32 rows and never meets
// a barrier inside the DP. Tags (a row counter that runs through all DPs of the edge) make every entry self-validating; the consumer
// reports how far it has read so that the producer never laps it.
// Nothing else is shared: the LDS ring of kept rows and the rows kept in HBM are private to the wave (each with a copy of the value
// left of its first column), so a wave that is ahead can never pull a row from under one that is behind.
//
// Cells are "keys": 64 x score + 6 low bits = move type * 16 + 15 - predecessor slot. The low bits make one max() do the
// reference's tie-breaking: type 3 diagonal > 2 vertical > 1 horizontal (its traceback tries them in this order and takes a
// horizontal move only when nothing else reaches the score), and among moves of one type the first predecessor in in-edge order
// wins. Scores stay below 2^24 in magnitude (8*(V+L) with V+L < 2^21, checked by the host), so keys fit 32 bits. The 6 bits of the
// winning move ARE the direction byte written to HBM for the traceback (in-degrees above 16 send the edge back to the host, which
// retries it with the score-matrix traceback).
// Round 4: the score in a key is DE-RAMPED, X[i][j] = H[i][j] - gap * j. A horizontal move then keeps the key's score, so the row's
// horizontal recurrence is a plain prefix maximum: the scan input is the chunk's largest key, nothing is subtracted before the scan or
// added after it, and the chunk's own recurrence runs once, after the carry is known - each finished key (score x 64 + KHC, the one
// horizontal code of every row format) is the next column's horizontal candidate and the row as later rows read it, in one register.
// A diagonal move adds (substitution score - gap), a vertical one gap; only the sink scores and the score-matrix flavour's HBM rows
// (plain scores for its traceback) put the ramp back.
//
// Rows that a later row needs as a NON-adjacent predecessor are flagged by the CSR build ("kept") and copied to an LDS
// ring in the order they are produced (per wave: CM planes of 65 words, column t*CM+k at word 65*CM*w + 65*k + 1 + t, conflict-free; word 0
// of the wave's LAST plane holds the value left of the wave's first column, so "the column left of my chunk" is one load
// at lane offset 0 for EVERY lane), or to HBM when the ring has wrapped; predecessor references carry that location (0 registers,
// 1..14 ring slot+1, 15 HBM). Row metadata travels in registers: every wave loads the records of 64 rows with one coalesced load (one
// batch ahead) and broadcasts the current row's words with v_readlane.
// Columns beyond L are computed like real ones and never read by a real column, so the loop has no column predicates.
// ---------------------------------------------------------------------------------------------------
constexpr int32_t NEGK = -(1 << 30);   // "minus infinity" key
constexpr int KD = 63, KV = 47;           // low 6 bits of a key of a wide row = move type * 16 + 15 - predecessor slot: diagonal 3, vertical 2
constexpr int KHC = 4;                    // ... and the horizontal move's code in EVERY row format (4-bit rows: type 1 * 4 + 3 - 3; wide rows: below every other code, the
                                          // traceback takes type 0 and 1 alike): a finished key (score x 64 + KHC) is the horizontal candidate of the next column as it is

constexpr uint32_t CARRY_BATCH = 32;       // rows whose carries a wave takes at a time at most
constexpr uint32_t CARRY_MIN = 4;          // ... and at least (int32 rows; = how far a wave that keeps up runs behind its left neighbour)
constexpr uint32_t WAVE_MBOX = 64;         // entries of the LDS mailbox between two waves of a workgroup (a power of two >= 2 * CARRY_BATCH)
constexpr uint32_t MAX_WAVES = 16;         // waves per workgroup at most
constexpr uint32_t WG_POLL_LIMIT = 1u << 24;   // polls of a wave for another wave of its own workgroup (resident by construction) before it flags an internal error

template <int NWAVES> struct WaveMailT {   // LDS
    unsigned long long box[(NWAVES > 1 ? NWAVES - 1 : 1) * WAVE_MBOX];   // boundary b (between waves b and b + 1): entry of row i at [b][i % WAVE_MBOX] = {tag, carry}
    uint32_t consumed[NWAVES];                                           // boundary b: tag up to which wave b + 1 has taken the carries
};
using WaveMail = WaveMailT<MAX_WAVES>;     // (dp_rows addresses box[] and consumed[] through their own pointers: the layout of the largest serves all)

// ---- an edge shared by several workgroups ("members", one CU each): member m owns the waves [m*NW, (m+1)*NW) of the pipeline; the carry
// of its last wave travels through a tagged 64-bit word per row in HBM (relaxed device-scope atomics; the tag validates the word, no
// fences inside the DP).
struct DpCl {
    uint32_t mem, members;            // this member / members of the edge (1: no cluster)
    uint32_t stride;                  // rows per member in mbox (vcap + 1)
    uint32_t tag0;                    // tag of row i = tag0 + i (rows of all DPs of the edge numbered consecutively)
    unsigned long long* mbox;         // edge base
    uint32_t* err;                    // device-visible error word of the edge (set when a poll gives up)
    uint32_t poll_limit;              // polls before a waiter gives up and flags the edge instead of hanging the GPU (the host then redoes it unshared)
    uint32_t lanes;                   // lanes of the workgroup that take part in the DP (a "wide" member has more: they work in the graph phases only)
};
__device__ __forceinline__ uint32_t ld_dev(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_dev64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_wg64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_wg64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t ld_wg(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_wg(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// inclusive prefix maximum over the 64 lanes of a wave: the classic DPP sequence with the max fused into the DPP instruction
// (lanes without a source keep their value). s_nop 1 = the two wait states a DPP read needs after a VALU write of its source.
__device__ __forceinline__ int wave_incl_max(int v) {
    int x;
    asm volatile(
        "v_mov_b32 %0, %1\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "=&v"(x) : "v"(v));
    return x;
}
// The same scan with the wait states a DPP read needs (two after a VALU write of its source) spent on scalar work of the row instead of s_nop - a lone
// wave pays ~8 cycles per `s_nop 1` (tools/dev_lonebench.hip). Scalar outputs: the LDS address of mailbox entry i and its tag, the row's ring slot (4 bits of
// its record) and the slot's byte offset, the record's rare-case bits; the nibble row pointer (dlo, dhi) moves on by dstep.
__device__ __forceinline__ int wave_incl_max_fill(int v, uint32_t i, uint32_t tag0, uint32_t mb_lds, uint32_t meta, uint32_t ring_w4, uint32_t dstep,
                                                  uint32_t& mb_addr, uint32_t& mb_tag, uint32_t& slot, uint32_t& rare, uint32_t& roff, uint32_t& dlo, uint32_t& dhi) {
    int x;
    // (every scalar operand through readfirstlane: a no-op for a value that already sits in a scalar register, and the only way to tell the compiler so)
    i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i); tag0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tag0); mb_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb_lds);
    meta = (uint32_t)__builtin_amdgcn_readfirstlane((int)meta); ring_w4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ring_w4); dstep = (uint32_t)__builtin_amdgcn_readfirstlane((int)dstep);
    dlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)dlo); dhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)dhi);
    asm volatile(
        "v_mov_b32 %[x], %[v]\n\t"
        "s_and_b32 %[a], %[i], 63\n\t"
        "s_lshl_b32 %[a], %[a], 3\n\t"
        "v_max_i32_dpp %[x], %[v], %[v] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %[x], %[v], %[x] row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %[x], %[v], %[x] row_shr:3 row_mask:0xf bank_mask:0xf\n\t"
        "s_add_i32 %[a], %[a], %[mb]\n\t"
        "s_add_i32 %[t], %[i], %[tag0]\n\t"
        "v_max_i32_dpp %[x], %[x], %[x] row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
        "s_bfe_u32 %[slot], %[meta], 0x40008\n\t"
        "s_and_b32 %[rare], %[meta], 44\n\t"
        "v_max_i32_dpp %[x], %[x], %[x] row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_mul_i32 %[roff], %[slot], %[rw]\n\t"
        "s_add_u32 %[dlo], %[dlo], %[dstep]\n\t"
        "v_max_i32_dpp %[x], %[x], %[x] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_addc_u32 %[dhi], %[dhi], 0\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %[x], %[x], %[x] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : [x] "=&v"(x), [a] "=&s"(mb_addr), [t] "=&s"(mb_tag), [slot] "=&s"(slot), [rare] "=&s"(rare), [roff] "=&s"(roff), [dlo] "+s"(dlo), [dhi] "+s"(dhi)
        : [v] "v"(v), [i] "s"(i), [tag0] "s"(tag0), [mb] "s"(mb_lds), [meta] "s"(meta), [rw] "s"(ring_w4), [dstep] "s"(dstep) : "scc");
    // (... and back: the compiler takes what an asm statement writes for divergent - a v_cmp for every test of it, a waterfall loop around the buffer store)
    mb_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb_addr); mb_tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb_tag); slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
    rare = (uint32_t)__builtin_amdgcn_readfirstlane((int)rare); roff = (uint32_t)__builtin_amdgcn_readfirstlane((int)roff);
    dlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)dlo); dhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)dhi);
    return x;
}
// 1 if a >= b, else 0, both wave-uniform: a scalar compare and select
__device__ __forceinline__ uint32_t s_ge_i32(int a, int b) {
    uint32_t r;
    asm("s_cmp_ge_i32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(a), "s"(b) : "scc");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
}
// 1 if the lane mask has a bit set, else 0
__device__ __forceinline__ uint32_t s_nz_u64(unsigned long long m) {
    uint32_t r;
    asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(m) : "scc");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
}
// byte 0 of four registers -> one dword
__device__ __forceinline__ uint32_t pack_b0(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint32_t ab = __builtin_amdgcn_perm(b, a, 0x0c0c0400u), cd = __builtin_amdgcn_perm(d, c, 0x0c0c0400u);
    return __builtin_amdgcn_perm(cd, ab, 0x05040100u);
}
template <int CM> __device__ __forceinline__ void store_dirs(uint8_t* p, const uint32_t (&v)[CM], uint32_t keep) {
    if constexpr (CM >= 4) {
        uint32_t w[CM / 4];
#pragma unroll
        for (int q = 0; q < CM / 4; q++) w[q] = pack_b0(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]) & keep;
        if constexpr (CM == 4) *reinterpret_cast<uint32_t*>(p) = w[0];
        else if constexpr (CM == 8) *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
        else {
#pragma unroll
            for (int q = 0; q < CM / 16; q++) reinterpret_cast<uint4*>(p)[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < CM; k++) p[k] = (uint8_t)(v[k] & keep);
    }
}

// low nibble of CM registers -> CM / 2 bytes (cell k in the low half of byte k / 2 for even k, the high half for odd k)
template <int CM> __device__ __forceinline__ void store_nibbles(uint8_t* p, const uint32_t (&v)[CM]) {
    static_assert(CM >= 4 && CM % 4 == 0, "4, 8, 16 or 32 columns per lane");
    uint32_t b[CM / 2];   // byte 0 of b[q] = the two cells 2q, 2q + 1 (higher bits are dropped by the byte packing)
#pragma unroll
    for (int q = 0; q < CM / 2; q++) b[q] = (v[2 * q] & 15u) | (v[2 * q + 1] << 4);
    if constexpr (CM == 4) *reinterpret_cast<uint16_t*>(p) = (uint16_t)__builtin_amdgcn_perm(b[1], b[0], 0x0c0c0400u);
    else {
        uint32_t w[CM / 8];
#pragma unroll
        for (int q = 0; q < CM / 8; q++) w[q] = pack_b0(b[4 * q], b[4 * q + 1], b[4 * q + 2], b[4 * q + 3]);
        if constexpr (CM == 8) *reinterpret_cast<uint32_t*>(p) = w[0];
        else if constexpr (CM == 16) *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
        else *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// the same nibbles through a raw buffer resource (base = the row, range = its bytes): offsets beyond the range are not written.
// Word 3 of the resource (0x00020000: raw, dword data format) and the rule the row loop relies on - "an access whose offset lies beyond num_records
// is dropped" - are the gfx9 family's; gfx10+ / gfx12 encode the word differently and check ranges per format. This file is written for gfx950 only:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "kernels/poa.hip stores its nibble rows through a gfx9 raw buffer resource (written for gfx950): another target needs the predicated store_nibbles()"
#endif
// (the row's resource covers W / 2 bytes = exactly the pitch of the nibble matrix: dp_rows steps its row pointer by the same W >> 1)
template <int CM> __device__ __forceinline__ void store_nibbles_buf(__amdgpu_buffer_rsrc_t r, const uint32_t off, const uint32_t (&v)[CM]) {
    static_assert(CM == 2 || (CM >= 4 && CM % 4 == 0), "2, 4, 8, 16 or 32 columns per lane");
    uint32_t b[CM / 2];
#pragma unroll
    for (int q = 0; q < CM / 2; q++) b[q] = (v[2 * q] & 15u) | (v[2 * q + 1] << 4);
    if constexpr (CM == 2) __builtin_amdgcn_raw_buffer_store_b8((char)b[0], r, (int)off, 0, 0);   // (two columns per lane - the members of the few-edge regime's shared edges: a byte per lane)
    else if constexpr (CM == 4) __builtin_amdgcn_raw_buffer_store_b16((short)__builtin_amdgcn_perm(b[1], b[0], 0x0c0c0400u), r, (int)off, 0, 0);
    else {
        uint32_t w[CM / 8];
#pragma unroll
        for (int q = 0; q < CM / 8; q++) w[q] = pack_b0(b[4 * q], b[4 * q + 1], b[4 * q + 2], b[4 * q + 3]);
        // (8 columns per lane = the instances of the many-edge regime: the nibble rows leave as NON-TEMPORAL stores. A 13 000-edge call writes 0.45 TB of them, of
        // which the traceback reads back one byte in a few thousand; streamed past the L2 they leave it to the graph arrays and the far rows, whose round trips
        // are what the serial phases are made of - wave cycles of all workgroups of the 140 Mb call: -2.8 %. aux 2 = nt on gfx940/gfx950.)
#ifndef HX_NIB_AUX
#define HX_NIB_AUX 2
#endif
        if constexpr (CM == 8) __builtin_amdgcn_raw_buffer_store_b32(w[0], r, (int)off, 0, HX_NIB_AUX);
        else if constexpr (CM == 16) { typedef uint32_t u32x2 __attribute__((ext_vector_type(2))); __builtin_amdgcn_raw_buffer_store_b64((u32x2){w[0], w[1]}, r, (int)off, 0, 0); }
        else { typedef uint32_t u32x4 __attribute__((ext_vector_type(4))); __builtin_amdgcn_raw_buffer_store_b128((u32x4){w[0], w[1], w[2], w[3]}, r, (int)off, 0, 0); }
    }
}

#ifdef HX_DP_PROF3   // development: per member of a shared edge, cycles inside the DP and cycles of them spent waiting for carries (phase slots 6 + member)
#define HX_DP_PROF
#define HX_DP_PROF2
#endif
#if defined(HX_DP_PROF) && !defined(HX_DP_PROF2)
#define DP_T(k) do { if (tid == 0) { const long long _n = clock64(); prof[k] += (unsigned long long)(_n - tprev); tprev = _n; } } while (0)
#else
#define DP_T(k) do { } while (0)
#endif
// The row loop is written for a lone wavefront's latency: on this hardware a VALU instruction costs ~5 cycles whether or not it depends
// on its predecessor, a taken scalar branch ~35, an LDS round trip ~75, the DPP scan ~90. So a row is ONE dispatch on where its
// predecessor lives (registers / LDS ring / anything else), then straight-line code: rare events (row spilled to HBM, sink row) share
// one not-taken branch, only rows with a non-adjacent reader are copied to the LDS ring (slot from the row's record), selects are
// arithmetic.
#ifdef HX_FARREAD_STORE   // (development: dead far-read rows store "nothing" and need a look, flags or not - as before the sticky far bit)
#define HX_FARREAD_RISKY(fb) true
#else
#define HX_FARREAD_RISKY(fb) (!(fb))
#endif
template <int CM, bool DIR, bool PRUNE, bool ONEW /* the workgroup is one wave (the 64-lane instances): no LDS mailbox on either side, no relay - known at compile time, the row loses its tests of them */>
__device__ __forceinline__ void dp_rows(const G& g, int32_t* __restrict__ H, uint8_t* __restrict__ D, uint8_t* __restrict__ Dwide, const uint32_t W, const uint32_t WH, const uint8_t* __restrict__ seq,
                        const uint32_t L_, const uint32_t V_, int32_t* ring, const uint32_t R_, const uint32_t ring_w_, const int match, const int mismatch, const int gap,
                        unsigned long long* wm_box, uint32_t* wm_cons, uint32_t* sink_row, int* sink_score, const uint32_t sink_cap, uint32_t& nSinkOut, const DpCl& cl, unsigned long long* prof,
                        const int thrT /* PRUNE: score threshold T of this alignment (PRUNE_OFF: nothing real is below it) */, const uint32_t lazy_on /* PRUNE: skipped waves poll rarely */, unsigned long long* pstat /* PRUNE: wave-rows, wave-rows skipped */,
                        const uint32_t far_n /* PRUNE: rows of H (far-read rows) of the edge */) {
    static_assert(!PRUNE || DIR, "pruned rows: direction-byte flavour only");
#if defined(HX_DP_PROF) && !defined(HX_DP_PROF2)
    long long tprev = clock64();
#endif
    // wave-uniform values the compiler cannot know to be uniform (they come through LDS / integer division / the thread index): in scalar
    // registers they turn the loop control, the ring slot arithmetic and the carry hand-over into scalar instructions and branches
    const uint32_t L = (uint32_t)__builtin_amdgcn_readfirstlane((int)L_), V = (uint32_t)__builtin_amdgcn_readfirstlane((int)V_);
    const uint32_t R = (uint32_t)__builtin_amdgcn_readfirstlane((int)R_), ring_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)ring_w_);
    const uint32_t tid = threadIdx.x, NT = ONEW ? 64u : blockDim.x, lane = tid & 63u, wv = ONEW ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), NW = ONEW ? 1u : cl.lanes >> 6;
    const uint32_t ncol = L + 1;
    // A wide member (1024 lanes, NW of its 16 waves in the DP) other than the first lets a spare wave RELAY the carries that arrive through HBM
    // into an LDS mailbox: its first DP wave then takes them like any wave takes its left neighbour's (an LDS round trip per batch of rows
    // instead of a device-scope load, ~1.7 us, that it would sit through - and the pipeline runs at the speed of its slowest wave).
    constexpr uint32_t RELAY_BOX = MAX_WAVES - 2;                             // mailbox / consumed word of the relay (the 1024-lane instances have them; boundaries 0 .. NW-2 are the DP's)
    const bool relay_mode = !ONEW && NT == 1024u && NW + 2u <= 16u && cl.mem > 0;
    if (wv >= NW) {                                                           // (a wave of a wide member that sits the DP out)
        if (relay_mode && wv == NW + 1u && (uint64_t)(cl.mem * NW) * 64u * CM < ncol) {   // (wave NW + 1: not the SIMD of the wave it feeds)
            const unsigned long long* src = cl.mbox + (uint64_t)(cl.mem - 1) * cl.stride;
            unsigned long long* dst = wm_box + (size_t)RELAY_BOX * WAVE_MBOX;
            const uint32_t* cons = wm_cons + RELAY_BOX;
            bool dead = false;
            for (uint32_t ib = 0; ib < V; ib += 64) {
                const uint32_t ie = min(64u, V - ib);
                uint32_t nb = 0;
                for (uint32_t rb = 0; rb < ie; rb += nb) {
                    const uint32_t want = min(CARRY_BATCH, ie - rb), i0 = ib + rb + 1;   // (as many rows as have arrived, at least CARRY_MIN: like the DP waves)
                    nb = want;
                    unsigned long long v = (unsigned long long)(cl.tag0 + i0 + lane);   // (a relay that gave up still hands out tagged entries: the edge is flagged and redone)
                    for (uint32_t spin = 0; !dead; spin++) {
                        bool ok = true;
                        if (lane < want) { v = ld_dev64(src + i0 + lane); ok = (uint32_t)v == cl.tag0 + i0 + lane; }
                        const unsigned long long okm = __ballot(ok);
                        const uint32_t run = okm == ~0ull ? want : (uint32_t)__builtin_ctzll(~okm);
                        if (run >= min(want, CARRY_MIN)) { nb = run; break; }
                        if (spin > cl.poll_limit) { if (lane == 0) st_dev(cl.err, 1u); dead = true; v = (unsigned long long)(cl.tag0 + i0 + lane); break; }
                        __builtin_amdgcn_s_sleep(8);
                    }
                    const uint32_t need = i0 + nb - 1 > WAVE_MBOX ? cl.tag0 + i0 + nb - 1 - WAVE_MBOX : 0;   // the entries overwritten must have been taken
                    for (uint32_t spin = 0; need; spin++) {
                        const uint32_t got = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld_wg(cons));
                        if ((int32_t)(got - need) >= 0) break;
                        if (spin > WG_POLL_LIMIT) { if (lane == 0) st_dev(cl.err, 2u); break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (lane < nb) st_wg64(dst + ((i0 + lane) & (WAVE_MBOX - 1)), v);
                }
            }
        }
        return;
    }
    const uint32_t gw = cl.mem * NW + wv;                                     // this wave's place in the edge's pipeline
    if ((uint64_t)gw * 64u * CM >= ncol) return;                              // the wave owns no real column of this sequence
    const uint32_t gt = gw * 64u + lane;                                      // lane index over all waves
    const bool has_in = gw > 0;                                               // a wave on the left feeds the horizontal carry ...
    const bool in_lds = !ONEW && (wv > 0 || relay_mode);                                 // ... through the workgroup's LDS mailbox, or (first wave of a member without a relay wave) through HBM
    const bool has_out = (uint64_t)(gw + 1) * 64u * CM < ncol;                // a wave on the right owns real columns (the host sized the pipeline for the longest sequence)
    const bool out_lds = !ONEW && wv + 1 < NW;
    // (wave-uniform tests of the row loop as 32-bit scalars: a test of a lane-mask boolean is `s_andn2 vcc` + a vcc branch, ~32 cycles for a lone
    // wave against ~15 for `s_cmp` + an scc branch: tools/dev_lonebench.hip)
    const uint32_t out_l = (uint32_t)__builtin_amdgcn_readfirstlane((int)(has_out && out_lds)), out_h = (uint32_t)__builtin_amdgcn_readfirstlane((int)(has_out && !out_lds));
    const unsigned long long* mb_in_h = cl.mbox + (uint64_t)(cl.mem ? cl.mem - 1 : 0) * cl.stride;
    unsigned long long* mb_out_h = cl.mbox + (uint64_t)cl.mem * cl.stride;
    const unsigned long long* mb_in_l = wm_box + (size_t)(wv ? wv - 1 : relay_mode ? RELAY_BOX : 0) * WAVE_MBOX;
    unsigned long long* mb_out_l = wm_box + (size_t)wv * WAVE_MBOX;
    uint32_t* cons_in = wm_cons + (wv ? wv - 1 : relay_mode ? RELAY_BOX : 0);   // what this wave has taken from the boundary on its left
    const uint32_t* cons_out = wm_cons + wv;                                  // what the wave on the right has taken from this wave's mailbox
    if (has_in && in_lds && lane == 0) st_wg(cons_in, cl.tag0);               // everything of earlier DPs counts as taken (a wave may have sat out a short sequence)
    const uint32_t* farslot = reinterpret_cast<const uint32_t*>(g.pred);      // per rank: row of H that holds the far-read row (consensus scratch, free during the DP)
    const uint32_t* wideslot = g.wslot;                                       // per rank: row of the wide-row pool (rows with more than 4 predecessors)
    const uint32_t j0 = gt * CM;
    const bool live = j0 <= L;                       // the chunk holds at least one real column: only such chunks touch HBM
    const bool owns_last = live && L < j0 + CM;
    const uint32_t klast = owns_last ? L - j0 : 0;
    const uint32_t hleft = W + gw;                   // H rows end with one word per wave: the value left of the wave's first column (its own copy)
    // bases under the lane's columns, 2 bits per column (bit pair k); columns without a base (column 0, padding) never match
    using mask_t = typename std::conditional<(CM <= 16), uint32_t, unsigned long long>::type;
    static_assert(CM <= 32, "at most 32 columns per lane");
    mask_t bases = 0, nobase = 0;
    uint32_t onehot = 0;   // up to 8 columns per lane: bit 4 k + letter of the base under column k - a row's match bits are ONE shift by its letter (no base: no bit)
#pragma unroll
    for (int k = 0; k < CM; k++) {
        const uint32_t j = j0 + k;
        if (j >= 1 && j < ncol) { bases |= (mask_t)seq[j - 1] << (2 * k); if constexpr (CM <= 8) onehot |= 1u << (4 * k + seq[j - 1]); } else nobase |= (mask_t)1 << (2 * k);
    }
    const int mm64 = mismatch * 64, g64 = gap * 64, m64 = match * 64;
    const int jg0 = (int)j0 * g64;
    const int mdN = m64 - g64 + 15 - KHC, mdW = m64 - g64 + KD - KHC, gvN = g64 + 11 - KHC, gvW = g64 + KV - KHC;   // diagonal (match) / vertical constants of 4-bit and wide rows
    // The previous row is still in the registers of the lanes that own its columns (tp, lnp): a successor that follows it immediately
    // reads it there - no LDS round trip on the most common dependency. Every row a NON-adjacent successor reads ("kept") lives in the LDS
    // ring, R slots in the order they are produced; rows nobody else reads are not written at all. A predecessor reference is a code from
    // the CSR build (1 + ring slot; 13 = the previous row; 14 = the virtual row 0 that source nodes start from; 15 = a kept row that left
    // the ring: HBM), the row's own slot sits in its record. Per wave: CM planes of 65 words, column t*CM+k at word 65*CM*wv + 65*k + 1 + t;
    // word 0 of the wave's LAST plane holds the value left of the wave's first column, so "the column left of my chunk" is word
    // 65*(CM-1) + t for EVERY lane: one load, no select. Plane offsets are instruction offsets of ONE address register per slot.
    constexpr uint32_t PW = 65u;
    int32_t* const ring_me = ring + wv * (65u * CM) + lane;
    if (!DIR && live) {                                 // the score-matrix traceback reads row 0 like any other row
        int pl[CM];
#pragma unroll
        for (int k = 0; k < CM; k++) pl[k] = (jg0 + k * g64) >> 6;
        store_chunk_i32<CM>(H + j0, pl);
    }
    uint32_t nsink = 0;
    // row records of 64 rows per register: the current batch (C), the next one (N, complete with the third and fourth predecessor entries
    // of the rows that have them - a gather that needs the records first), and the one after it (F) in flight
    uint32_t mC = 0, aC = 0, bC = 0, oC = 0, cC = 0, dC = 0, fC = 0, mN = 0, aN = 0, bN = 0, oN = 0, cN = 0, dN = 0, fN = 0, mF = 0, aF = 0, bF = 0, oF = 0;
    auto fetch = [&](uint32_t base, uint32_t& m, uint32_t& a, uint32_t& b, uint32_t& o) {
        const uint32_t r = base + lane;
        if (r < V) { m = g.row_meta[r]; a = g.row_pred0[r]; b = g.row_pred1[r]; o = g.row_pred_off[r]; }
    };
    // Second stage, a batch ahead of its use. Everything that would otherwise be a DEPENDENT load on the row's own path is gathered here:
    // the third and fourth predecessor entries, and (direction-byte flavour) the H slots of far rows - a far first / second predecessor
    // entry gets its slot in place of the rank (the row loop never needs the rank), a row that is stored for a far reader its own slot.
    auto fetch_more = [&](uint32_t base, uint32_t m, uint32_t o, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d, uint32_t& f) {
        if (base + lane < V) {
            if ((m >> META_NP) > 2u) c = g.pred_rank[o + 2];
            if ((m >> META_NP) > 3u) d = g.pred_rank[o + 3];
            if (DIR) {
                if ((a >> 28) == 15u) a = 0xf0000000u | farslot[a & 0x0fffffffu];
                if ((b >> 28) == 15u && (m >> META_NP) > 1u) b = 0xf0000000u | farslot[b & 0x0fffffffu];
                if (m & 8u) f = farslot[base + lane];
            }
        }
    };
    fetch(0, mN, aN, bN, oN);
    fetch(64, mF, aF, bF, oF);
    fetch_more(0, mN, oN, aN, bN, cN, dN, fN);
    int32_t* hrow = H;
    uint32_t dlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)D), dhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)D >> 32));   // the row of direction nibbles (a scalar pointer in two halves: wave_incl_max_fill moves it)
    const uint32_t mb_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)mb_out_l);   // LDS byte address of this wave's mailbox towards the right
    const uint32_t ring_w4 = ring_w * 4u, tag0_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)cl.tag0), dstep_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(DIR ? W >> 1 : 0u));
    // predecessor row `ent` (slot << 28 | rank): its columns under this lane and the value left of them
    int tp[CM], lnp = NEGK;                  // the previous row's finished keys under this lane, and the key left of the wave's first column (lane 0's is used)
#pragma unroll
    for (int k = 0; k < CM; k++) tp[k] = NEGK;
    // ---- PRUNE: exact score-bound pruning at the granularity this pipeline works at, (row, wave). U(i, j) = H[i][j] + match x (L - j) bounds the
    // final score of every path through cell (i, j) and never grows along a path, so with a threshold T <= the final score S no cell with U < T
    // lies on an optimal path, every cell of an optimal path keeps its exact value whatever stands in the dead cells (anything <= their true
    // value), and the traceback - which compares the candidates of optimal cells only - is unchanged. In de-ramped keys (X = H - gap j, never
    // decreasing along a row) U = X + match L - (match - gap) j: a lane's columns are all dead when the key of its LAST column, taken at its
    // FIRST, is below T (`thr_lane`), the carry entering the wave is dead below `thr_cin`. A wave SKIPS a row (no predecessor reads, no cells, no
    // scan, no ring copy, no nibbles: it forwards the incoming carry under the row's tag) when the carry is dead and none of its predecessor
    // rows was FLAGGED by this wave; a computed row is flagged when one of its lanes or its carry-in is live. The flags of the rows a successor
    // can name live in one scalar word FM, bit = the location code of a predecessor entry: 1 + ring slot, 13 the previous row, 14 the virtual
    // row 0, 15 a row read back from HBM (always set: a skipped row with a far reader stores "nothing" there, so it may be read). Predecessors
    // whose flag is clear are not read at all (their registers / ring slot hold an older row). T is the caller's: poa_edge checks S >= T
    // afterwards and repeats the alignment otherwise. (kernels.h: PRUNE_OFF; oracle.cpp prune_sim = this rule on the CPU, a statistic.)
    int thr_lane = 0; uint32_t FM = 0xffffu, n_dead = 0, n_bulk = 0, lazy = 0; int thr_cin = 0;
    if constexpr (PRUNE) {
        const int mg = match - gap, thr_base = thrT - match * (int)L;
        const int c0 = (int)(gw * 64u * CM);
        // (padding lanes - no real column: the last wave's - never flag a row. They read the FIRST chunk of a far row, another wave's, and what stands there when
        // that wave had no reason to store the row is whatever the slot held before: harmless for the cells, but a flag from it made the pruning counters differ
        // from run to run)
        thr_lane = live ? (thr_base + mg * (int)j0) * 64 : INT32_MAX;
        thr_cin = __builtin_amdgcn_readfirstlane((thr_base + mg * (c0 - 1)) * 64);
        const uint32_t f0 = (uint32_t)(match * (int)L - mg * max(c0 - 1, 0) >= thrT);
        // (bit 15, "a row in HBM": set for good where the far rows have no flags of their own; else it is raised by the first far row this wave stores LIVE - until
        // then every far row it could name is dead, and above the band of the matrix a record that names one is as dead as its neighbours)
        FM = (uint32_t)__builtin_amdgcn_readfirstlane((int)((far_n <= 2048u ? 0u : 0x8000u) | (f0 << 14)));   // nothing in the ring, no previous row yet
    }
    // PRUNE: the flags of the rows kept in HBM ("far" rows, code 15), one bit per row of H in ONE register of the wave (lane = slot / 32: 2 048 slots; an edge with
    // more has every bit set for good - its far rows are read as they always were). A far row whose flag is clear is not fetched: its reader takes "nothing",
    // like the readers of an unflagged ring row do - in a dead region of the matrix that was an HBM round trip (~2 us under load) on the path of a row that came
    // out dead anyway, one row in forty.
    const bool far_bits = PRUNE && far_n <= 2048u;
    uint32_t farbits = far_bits ? 0u : 0xffffffffu;
    auto far_set = [&](const uint32_t slot, const uint32_t fl) {   // (slot, fl: wave-uniform)
        if (far_bits && lane == ((slot >> 5) & 63u)) farbits = (farbits & ~(1u << (slot & 31u))) | (fl << (slot & 31u));
    };
    auto pred_row = [&](const uint32_t ent, int (&hp)[CM], int& left, const bool slot_known) {
        const uint32_t loc = ent >> 28;
        if (__builtin_expect(loc == 13u, 1)) {   // the previous row: registers (the likely case falls through: a taken scalar branch costs a lone wave ~35 cycles)
#pragma unroll
            for (int k = 0; k < CM; k++) hp[k] = tp[k];
            left = wave_shift_up1(tp[CM - 1], lnp);
        } else if (__builtin_expect(loc < 13u, 1)) {    // in the LDS ring
            const int32_t* S = ring_me + (size_t)(loc - 1) * ring_w;
#pragma unroll
            for (int k = 0; k < CM; k++) hp[k] = S[k * PW + 1];
            left = S[(CM - 1) * PW];
        } else if (loc == 14u) {                 // a source node starts from the virtual row 0
#pragma unroll
            for (int k = 0; k < CM; k++) hp[k] = KHC;   // (row 0 is the gap ramp itself)
            left = gt > 0 ? KHC : NEGK;
        } else {                                 // kept row that fell out of the ring: HBM
            // (no divergent branch in here: with one, the compiler structurises the whole dispatch and every row pays a flag test. Padding lanes
            // - columns beyond the sequence, last wave only - read the row's first chunk instead: their keys reach no real column)
            // with direction bytes only the rows a far successor reads are in HBM, in the slots the CSR build gave them
            const uint32_t hr = DIR ? (slot_known ? ent & 0x0fffffffu : farslot[ent & 0x0fffffffu]) : (ent & 0x0fffffffu) + 1;
            if constexpr (PRUNE) {
                const uint32_t hs = (uint32_t)__builtin_amdgcn_readfirstlane((int)hr);
                if ((((uint32_t)__builtin_amdgcn_readlane((int)farbits, (int)((hs >> 5) & 63u)) >> (hs & 31u)) & 1u) == 0u) {   // an unflagged far row: not read
#pragma unroll
                    for (int k = 0; k < CM; k++) hp[k] = NEGK;
                    left = NEGK;
                    return;
                }
            }
            const int32_t* Grow = H + (uint64_t)hr * WH;
            const uint32_t jl = live ? j0 : 0u;
            const int32_t* Gp = Grow + jl;
            load_chunk_i32<CM>(Gp, hp);
            // the key left of the chunk: the neighbour's last column - lane 0: the wave's own copy (the column belongs to a wave that may be far ahead)
            const int32_t* lp = (lane > 0 && live) ? Gp - 1 : has_in ? Grow + hleft : Grow;
            left = *lp;
            if (!DIR) {                          // the score matrix holds plain scores
#pragma unroll
                for (int k = 0; k < CM; k++) hp[k] = (hp[k] << 6) - (jg0 + k * g64) + KHC;
                left = (left << 6) - (jg0 - g64) + KHC;
            }
            left = gt > 0 ? left : NEGK;
            // the loaded values are consumed HERE: otherwise the wait for them is placed where the three sources of a predecessor row
            // join - on the path of every row - and waits for the previous rows' direction stores as well (vmcnt counts them)
#pragma unroll
            for (int k = 0; k < CM; k++) asm volatile("" : "+v"(hp[k]));
            asm volatile("" : "+v"(left));
        }
    };
    // (a one-wave workgroup has no LDS mailbox on either side - its carries come from and go to HBM, window by window - and takes the 64 rows of a record batch
    // at once: half as many round trips for the carries, which is what a dead batch costs now that its rows leave in runs)
    const uint32_t cbatch = (uint32_t)__builtin_amdgcn_readfirstlane((int)(NW == 1u ? 64u : CARRY_BATCH));
    for (uint32_t ib = 0; ib < V; ib += 64) {
        // the batches move up (the only waits for these loads: everything was requested at least 64 rows ago), another one goes in flight
        mC = mN; aC = aN; bC = bN; oC = oN; cC = cN; dC = dN; fC = fN;
        mN = mF; aN = aF; bN = bF; oN = oF;
        fetch(ib + 128, mF, aF, bF, oF);
        fetch_more(ib + 64, mN, oN, aN, bN, cN, dN, fN);
        const uint32_t ie = min(64u, V - ib);
        uint32_t nb = 0;
        for (uint32_t rb = 0; rb < ie; rb += nb) {
            // rows i0 .. i0 + nb - 1: up to CARRY_BATCH rows of the record batch - as many as have their carries in the mailbox, at least CARRY_MIN (a
            // wave follows its left neighbour at that distance when it keeps up, and the mailbox's 64 entries still absorb a neighbour's hiccup)
            const uint32_t want = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(cbatch, ie - rb)), i0 = ib + rb + 1;
            nb = want;
            int cinV = NEGK;     // lane r: carry into this wave for row i0 + r
            if (has_in) {
#ifdef HX_DP_PROF3
                const long long tw0 = clock64();
#endif
                for (uint32_t spin = 0;; spin++) {
                    unsigned long long v = 0;
                    bool ok = true;
                    if (lane < want) {
                        v = in_lds ? ld_wg64(mb_in_l + ((i0 + lane) & (WAVE_MBOX - 1))) : ld_dev64(mb_in_h + i0 + lane);
                        ok = (uint32_t)v == cl.tag0 + i0 + lane;
                    }
                    const unsigned long long okm = __ballot(ok);
                    const uint32_t run = okm == ~0ull ? want : (uint32_t)__builtin_ctzll(~okm);   // leading rows whose carries have arrived
                    // (PRUNE: a wave whose last batch was skipped whole is AHEAD of the band - it is not what its edge waits for, but a poll every ~130 cycles
                    // takes issue slots from the waves that are: it waits for whole batches and sleeps 16 times as long between polls)
                    if (run >= min(want, PRUNE && lazy ? want : CARRY_MIN)) { nb = run; cinV = (int)(uint32_t)(v >> 32); break; }
                    if (spin > (in_lds ? WG_POLL_LIMIT : cl.poll_limit)) { if (lane == 0) st_dev(cl.err, in_lds ? 2u : 1u); break; }
                    if (PRUNE && lazy) __builtin_amdgcn_s_sleep(32);
                    else if (in_lds) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(8);
                }
                if (in_lds && lane == 0) st_wg(cons_in, cl.tag0 + i0 + nb - 1);   // the entries of these rows may be written again
#ifdef HX_DP_PROF3
                if (tid == 0) prof[0] += (unsigned long long)(clock64() - tw0);
#endif
            }
            if (has_out && out_lds) {   // the rows of this batch overwrite the entries of the rows WAVE_MBOX earlier: the wave on the right must have taken those
                const uint32_t need = i0 + nb - 1 > WAVE_MBOX ? cl.tag0 + i0 + nb - 1 - WAVE_MBOX : 0;
                for (uint32_t spin = 0; need; spin++) {
                    const uint32_t got = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld_wg(cons_out));
                    if ((int32_t)(got - need) >= 0) break;
                    if (spin > WG_POLL_LIMIT) { if (lane == 0) st_dev(cl.err, 2u); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            // (a row's record words are read out of their lane during the row BEFORE: a scalar instruction that consumes a readlane's result at once
            // waits ~14 cycles for it)
            // PRUNE: rows skipped in RUNS. `bad` = the rows of the batch that cannot be skipped without a look at them (bit r = row i0 + r): a live carry, or
            // a record that names a predecessor outside the ring and the previous row (the virtual row 0 where this wave has it flagged, a row in HBM, a
            // fifth predecessor) or that is itself read back from HBM (it has to store "nothing" there) - all of it read off the 64 row records in their
            // lanes, once per batch. Wherever nothing in the ring or the previous row is flagged (FM), the rows up to the next bad one are dead and leave
            // together: their carries go out as one vector store. (Round 5, second half: the first version skipped whole batches only, and one risky row in
            // fourteen left 45 % of the dead rows to the row-by-row path at ~750 cycles each under load.)
            unsigned long long bad = 0, farref = 0;
            if constexpr (PRUNE) {
                const uint32_t np_l = mC >> META_NP;
                // (codes 14 and 15 - the virtual row 0, a row in HBM - name something live only where this wave has their bit of FM set. A row that is itself read back
                // from HBM needs a look only where the far rows have no flags: with them its bit is clear until somebody stores it live, and nobody fetches it)
                auto outside = [&](const uint32_t ent) -> bool { const uint32_t c = ent >> 28; return c >= 14u && ((FM >> c) & 1u) != 0u; };
                const bool risky = outside(aC) || (np_l > 1u && outside(bC)) || (np_l > 2u && outside(cC)) || (np_l > 3u && outside(dC)) || np_l > 4u || (HX_FARREAD_RISKY(far_bits) && (mC & 8u) != 0u);
                const bool clive = lane < nb && cinV >= thr_cin;
                bad = (__builtin_amdgcn_ballot_w64(risky) >> rb) | __builtin_amdgcn_ballot_w64(clive);
                if ((FM & 0x8000u) == 0u) {   // the rows that name a far row: they need a look from the moment one is stored live (below)
                    const bool refs_far = (aC >> 28) == 15u || (np_l > 1u && (bC >> 28) == 15u) || (np_l > 2u && (cC >> 28) == 15u) || (np_l > 3u && (dC >> 28) == 15u);
                    farref = __builtin_amdgcn_ballot_w64(refs_far) >> rb;
                }
                if (nb < 64u) { bad &= (1ull << nb) - 1ull; farref &= (1ull << nb) - 1ull; }   // (a bit beyond the batch would send a run past its end)
            }
            // (the run is looked for where one can begin - at the batch's first row and behind a row that was skipped by itself - not on the path of a live row: as a
            // test at the head of every row it cost ten scalar instructions, and a 13 000-edge call 4 %)
            auto skip_run = [&](const uint32_t from) -> uint32_t {
                if ((FM & 0x3ffeu) != 0u || from >= nb) return 0u;
                const unsigned long long rest = bad >> from;
                const uint32_t run = rest ? (uint32_t)__builtin_ctzll(rest) : nb - from;
                if (run != 0u) {
                    if (lane >= from && lane < from + run) {
                        const unsigned long long ent = (unsigned long long)(tag0_s + i0 + lane) | ((unsigned long long)(uint32_t)cinV << 32);
                        if (out_l != 0u) st_wg64(mb_out_l + ((i0 + lane) & (WAVE_MBOX - 1)), ent);
                        if (out_h != 0u) st_dev64(mb_out_h + i0 + lane, ent);
                    }
                    n_dead += run; n_bulk += run;
                    const unsigned long long dp_ = (((unsigned long long)dhi << 32) | dlo) + (unsigned long long)dstep_s * run;
                    dlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)dp_); dhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(dp_ >> 32));
                }
                return run;
            };
            const uint32_t dead_before = n_dead;
            uint32_t rj0 = 0;
            if constexpr (PRUNE) rj0 = skip_run(0u);
            uint32_t meta_nx = __builtin_amdgcn_readlane(mC, (rb + rj0) & 63u), p0_nx = __builtin_amdgcn_readlane(aC, (rb + rj0) & 63u);
            // The row, in two forms of one body. FAST (direction bytes, no pruning: the instances of the few-edge regime, where ONE wave's instruction count per row is
            // what the call waits for) = a row whose record names ONE predecessor, the previous row: three rows in five. Such rows run in a loop of their own
            // (below): the cells come straight out of the previous row's registers (as one of three sources joined in one set of registers the compiler copies them:
            // six v_mov), the row format's constants are the 4-bit ones, and neither the dispatch on the first predecessor's location nor the branch around the
            // later predecessors exists. (Round 5 tried the same cells as a block inside the one loop: + 14 %, through the two taken branches around it.)
            uint32_t rj = rj0;
            auto row = [&](auto fast_tag) __attribute__((always_inline)) {
                constexpr bool FAST = decltype(fast_tag)::value;
                const uint32_t ri = rb + rj, i = ib + ri + 1;
                const uint32_t meta = meta_nx, p0 = p0_nx;
                const uint32_t npred = meta >> META_NP;
                uint32_t cin_live = 0, fl0 = 1, flB = 1, flC = 1, flD = 1;
                if constexpr (PRUNE) {
                    // does anything this wave can read for the row still reach T? (all scalar: flags of the predecessor entries, the carry's test)
                    const int cin_e = __builtin_amdgcn_readlane(cinV, rj);
                    cin_live = s_ge_i32(cin_e, thr_cin);   // (as a C comparison the flag became a lane mask and the whole test vector code: v_cndmask, v_or, v_cmp_ne, a vcc branch)
                    fl0 = (FM >> (p0 >> 28)) & 1u;
                    uint32_t act = cin_live | fl0;
                    if (npred > 1) {
                        flB = (FM >> ((uint32_t)__builtin_amdgcn_readlane(bC, ri) >> 28)) & 1u; act |= flB;
                        if (npred > 2) {
                            flC = (FM >> ((uint32_t)__builtin_amdgcn_readlane(cC, ri) >> 28)) & 1u; act |= flC;
                            if (npred > 3) { flD = (FM >> ((uint32_t)__builtin_amdgcn_readlane(dC, ri) >> 28)) & 1u; act |= flD | ((npred + 3u) >> 3); }   // (non-zero for a fifth predecessor)
                        }
                    }
                    if (act == 0u) {
                        // ---- a skipped row: the carry passes through, the row's flags are cleared, a far reader finds "nothing"
                        n_dead++;
                        const uint32_t slot_d = (meta >> META_SLOT) & 15u;
                        FM &= ~(0x2000u | (2u << slot_d));   // (slot 15 = not kept: bit 16, which nobody reads)
                        meta_nx = __builtin_amdgcn_readlane(mC, (ri + 1) & 63u); p0_nx = __builtin_amdgcn_readlane(aC, (ri + 1) & 63u);
                        {
                            const unsigned long long ent = (unsigned long long)(tag0_s + i) | ((unsigned long long)(uint32_t)cin_e << 32);
                            if (out_l != 0u) { if (lane == 63) *(volatile __attribute__((address_space(3))) unsigned long long*)(uintptr_t)(mb_lds + ((i & (WAVE_MBOX - 1)) << 3)) = ent; }
                            if (out_h != 0u) { if (lane == 63) st_dev64(mb_out_h + i, ent); }
                        }
                        {   // the nibble row pointer moves on (live rows: inside the scan)
                            const unsigned long long dp_ = (((unsigned long long)dhi << 32) | dlo) + dstep_s;
                            dlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)dp_); dhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(dp_ >> 32));
                        }
                        if (__builtin_expect((meta & 8u) != 0u, 0)) {   // a far successor will read this row from HBM: keys of "nothing" (its flag is always taken for set)
                            const uint32_t fslot = __builtin_amdgcn_readlane(fC, ri);
                            if (HX_FARREAD_RISKY(far_bits) && live) {   // (with flags: the row's bit is clear - every DP begins with none set - and nobody fetches an unflagged row)
                                int32_t* F = H + (uint64_t)fslot * WH;
                                int ng[CM];
#pragma unroll
                                for (int k = 0; k < CM; k++) ng[k] = NEGK;
                                store_chunk_i32<CM>(F + j0, ng);
                                if (lane == 0 && has_in) F[hleft] = NEGK;
                            }
                        }
                        {   // the rows behind this one, up to the next that needs a look
                            const uint32_t run = skip_run(rj + 1u);
                            if (run != 0u) { rj += run; meta_nx = __builtin_amdgcn_readlane(mC, (rb + rj + 1u) & 63u); p0_nx = __builtin_amdgcn_readlane(aC, (rb + rj + 1u) & 63u); }
                        }
                        return;
                    }
                }
                // mismatch bits of this row: bit 2k set <=> the base under column k differs from the row's letter
                mask_t mis = 0;
                uint32_t hit = 0;     // (CM <= 8) bit 4 k set <=> the base under column k is the row's letter
                if constexpr (CM <= 8) hit = onehot >> (meta & 3u);
                else {
                    const mask_t x = bases ^ ((mask_t)(meta & 3u) * (mask_t)0x5555555555555555ull);
                    mis = x | (x >> 1) | nobase;
                }
                // Move codes (the low 6 bits of a key while a row is computed; they are masked off before the row is used as a predecessor, so the
                // format is the row's own). A row with at most 4 predecessors uses 4 bits - type * 4 + 3 - predecessor slot - which are its
                // traceback nibble as they are; a "wide" row (rare) uses type * 16 + 15 - slot and stores a byte per cell in a side pool.
                const bool wide = !FAST && (!DIR || (meta & 32u));   // (a fast row has one predecessor: never the wide format)
                // (a diagonal move leaves the ramp of column j - 1 for that of column j; a finished key carries KHC; both formats' constants wait in
                // scalar registers: one bit test and three selects per row)
                const int md = wide ? mdW : mdN, gv = wide ? gvW : gvN, mmd = wide ? mdW + (mm64 - m64) : mdN + (mm64 - m64);
                auto score_of = [&](int k) -> int {   // 64 x substitution score of column k + the diagonal move code
                    if constexpr (CM <= 8 && FAST) {   // (the bit field through an asm statement: with constant terms around it the compiler turns one of the cells into v_and + v_cmp + two v_mov + v_cndmask)
                        int bit;
                        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(bit) : "v"(hit), "n"((4 * k) & 31));
                        return mmd + ((m64 - mm64) & bit);
                    }
                    if constexpr (CM <= 8) return mmd + ((m64 - mm64) & __builtin_amdgcn_sbfe((int)hit, (4 * k) & 31, 1));   // (-1 on a match)
                    int neg;   // -1 on a mismatch, 0 on a match
                    if constexpr (CM <= 16) neg = __builtin_amdgcn_sbfe((int)mis, 2 * k, 1);
                    else neg = 2 * k < 32 ? __builtin_amdgcn_sbfe((int)(uint32_t)mis, (2 * k) & 31, 1) : __builtin_amdgcn_sbfe((int)(uint32_t)(mis >> 32), (2 * k) & 31, 1);
                    return md + ((mm64 - m64) & neg);
                };
                hrow += WH;
                DP_T(0);   // row decode
                int m[CM];
                if constexpr (FAST) {
                    // (see the loops below: ONE predecessor, the previous row - its cells straight from the registers they are in, the 4-bit row format's constants,
                    // no dispatch, nothing behind the cells to skip)
                    const int left = wave_shift_up1(tp[CM - 1], lnp);
#pragma unroll
                    for (int k = 0; k < CM; k++) m[k] = max((k == 0 ? left : tp[k - 1]) + score_of(k), tp[k] + gv);
                } else
                if (!PRUNE || __builtin_expect(fl0 != 0u, 1)) {   // the first predecessor (or row 0): diagonal and vertical move
#ifndef HX_NO_PREV_DIRECT
                    if (PRUNE && __builtin_expect((p0 >> 28) == 13u, 1)) {   // (PRUNE = the instances of the many-edge regime; the row of the 4-column instances a lone wave runs got 14 % SLOWER with this block: 292 -> 332 M cycles on the longest 12 Mb edge)
                        // the previous row: its cells straight from the registers they are in. (Through pred_row the three sources of a predecessor row join
                        // in ONE set of registers and the compiler copies the previous row into them - ten v_mov per row of the 8-column instances. The
                        // statement at the end keeps this block from being merged with the general one below.)
                        const int left = wave_shift_up1(tp[CM - 1], lnp);
#pragma unroll
                        for (int k = 0; k < CM; k++) m[k] = max((k == 0 ? left : tp[k - 1]) + score_of(k), tp[k] + gv);
                        asm volatile("" : "+v"(m[CM - 1]));
                    } else
#endif
                    {
                        int hp[CM], left;
                        pred_row(p0, hp, left, true);
#pragma unroll
                        for (int k = 0; k < CM; k++) m[k] = max((k == 0 ? left : hp[k - 1]) + score_of(k), hp[k] + gv);
                    }
                } else {                                          // (PRUNE: a skipped row is not read)
#pragma unroll
                    for (int k = 0; k < CM; k++) m[k] = NEGK;
                }
                if (!FAST && npred > 1) {   // (two rows in five at 25-45x)
                    // the maximum over the predecessors: the low bits carry the move type and 15 - p, so ONE running maximum does it all
                    // (a diagonal beats a vertical move of the same score, the first predecessor in in-edge order beats the later ones).
                    // The second, third and fourth predecessor are spelled out - their entries come with the row records, one readlane each, and most of
                    // them live in the LDS ring, which is read here without the general dispatch; a loop that picks the entry by its index and then
                    // dispatches compiles into a dozen flag tests per predecessor (~300 cycles for a lone wave).
                    auto more = [&](const uint32_t ent, const int ps, const bool slot_known) {
                        int hp[CM], left;
                        const uint32_t loc = ent >> 28;
                        const bool inring = loc - 1u < 12u;
                        // the ring row is requested FIRST, whatever the entry is (another kind asks for slot 0 and is overwritten below): one not-taken
                        // test on the usual path instead of an if / else whose join the compiler guards with a flag test
                        const int32_t* S = ring_me + (size_t)(inring ? loc - 1u : 0u) * ring_w;
#pragma unroll
                        for (int k = 0; k < CM; k++) hp[k] = S[k * PW + 1];
                        left = S[(CM - 1) * PW];
                        if (__builtin_expect(!inring, 0)) pred_row(ent, hp, left, slot_known);
                        // (diagonal: ONE three-operand add per cell - predecessor key + the cell's substitution term + the slot's code offset)
                        const int mps = -ps, gvp = gv - ps;
#pragma unroll
                        for (int k = 0; k < CM; k++) {
                            int dg;
                            asm("v_add3_u32 %0, %1, %2, %3" : "=v"(dg) : "v"(k == 0 ? left : hp[k - 1]), "v"(score_of(k)), "s"(mps));
                            m[k] = max(m[k], max(dg, hp[k] + gvp));
                        }
                    };
                    if (!PRUNE || flB != 0u) more(__builtin_amdgcn_readlane(bC, ri), DIR ? 1 : 0, true);
                    if (npred > 2) {
                        if (!PRUNE || flC != 0u) more(__builtin_amdgcn_readlane(cC, ri), DIR ? 2 : 0, false);
                        if (npred > 3) {
                            if (!PRUNE || flD != 0u) more(__builtin_amdgcn_readlane(dC, ri), DIR ? 3 : 0, false);
                            if (__builtin_expect(npred > 4, 0)) {   // a fifth and later ones are fetched here (direction bytes exist only while in-degrees stay <= 16: the CSR build checks)
                                const uint32_t po = __builtin_amdgcn_readlane(oC, ri);
                                for (uint32_t p = 4; p < npred; p++) {
                                    const uint32_t ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.pred_rank[po + p]);
                                    if (!PRUNE || ((FM >> (ent >> 28)) & 1u) != 0u) more(ent, DIR ? (int)p : 0, false);
                                }
                            }
                        }
                    }
                }
                // what this chunk hands to the right whatever comes in from the left: its largest key (the horizontal move of de-ramped keys is a
                // plain prefix maximum, so the chunk's own recurrence can wait for the carry and run ONCE, after the scan)
                int lm = m[0];
#pragma unroll
                for (int k = 1; k < CM; k++) lm = max(lm, m[k]);
                DP_T(1);   // predecessor rows + cells
                // prefix maximum over the lanes to the left; the wait states of its DPP steps do the row's scalar chores (mailbox entry address and tag,
                // ring slot and its offset, the test word of the rare cases, the step of the nibble row pointer)
                uint32_t mb_addr, mb_tag, slot, rare, roff;
                const int inc = wave_incl_max_fill((lm & ~63) | KHC, (uint32_t)__builtin_amdgcn_readfirstlane((int)i), tag0_s, mb_lds, meta, ring_w4, dstep_s, mb_addr, mb_tag, slot, rare, roff, dlo, dhi);
                int ex = wave_shift_up1(inc, NEGK);
                DP_T(2);   // wave scan
                const int cin = __builtin_amdgcn_readlane(cinV, rj);   // NEGK without a wave on the left
                meta_nx = __builtin_amdgcn_readlane(mC, (ri + 1) & 63u);   // (the next row's record; beyond the batch: unused)
                if constexpr (!FAST) p0_nx = __builtin_amdgcn_readlane(aC, (ri + 1) & 63u);   // (inside a run of fast rows the first predecessor is known; read again behind the run)
                // the carry of this row for the wave on the right: the prefix maximum through this wave's last column (lane 63 holds it)
                if (out_l != 0u) { if (lane == 63) *(volatile __attribute__((address_space(3))) unsigned long long*)(uintptr_t)mb_addr = (unsigned long long)mb_tag | ((unsigned long long)(uint32_t)max(cin, inc) << 32); }
                if (out_h != 0u) { if (lane == 63) st_dev64(mb_out_h + i, (unsigned long long)mb_tag | ((unsigned long long)(uint32_t)max(cin, inc) << 32)); }
                ex = max(ex, cin);
                DP_T(3);   // carry in / out
                // the horizontal recurrence from the finished key left of this chunk (the exclusive prefix; it carries the horizontal code, which
                // loses every tie) through the chunk: each finished key t[k] is both the next column's horizontal candidate and the row as a predecessor
                m[0] = max(m[0], ex);
                int t[CM];
                t[0] = (m[0] & ~63) | KHC;
#pragma unroll
                for (int k = 1; k < CM; k++) { m[k] = max(m[k], t[k - 1]); t[k] = (m[k] & ~63) | KHC; }
                const int left_now = ex;              // key of column j0 - 1
                if (slot != 15u) {   // a kept row goes to its ring slot (the CSR build counted the kept rows)
                    int32_t* S = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ring_me) + roff);
#pragma unroll
                    for (int k = 0; k < CM; k++) S[k * PW + 1] = t[k];
                    if (lane == 0) S[(CM - 1) * PW] = left_now;   // (first wave of the edge: "minus infinity")
                }
#pragma unroll
                for (int k = 0; k < CM; k++) tp[k] = t[k];
                lnp = left_now;
                uint32_t row_fl = 1u;
                if constexpr (PRUNE) {   // the row's flag for its successors: a lane whose last key, taken at its first column, reaches T - or a live carry-in (the column left of the wave)
                    const uint32_t fl = s_nz_u64(__builtin_amdgcn_ballot_w64(t[CM - 1] >= thr_lane)) | cin_live;   // (as a C comparison: s_cselect_b64, v_cndmask, v_readfirstlane)
                    row_fl = fl;
                    { const uint32_t mk = 0x2000u | (2u << slot); FM = (FM & ~mk) | (mk * fl); }   // (the previous row's bit and the ring slot's: both the row's flag)
                }
                DP_T(4);   // carry applied, ring copy
                if (DIR) {
                    // the move code of every cell: type * 4 + 3 - predecessor slot. The row is stored through a buffer resource of ITS bytes: chunks
                    // beyond the row (the padding lanes of the last wave) fail the range check and are dropped - no exec mask, no branch
                    uint32_t dc[CM];
#pragma unroll
                    for (int k = 0; k < CM; k++) dc[k] = (uint32_t)m[k];
                    store_nibbles_buf<CM>(__builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t*>(((uintptr_t)dhi << 32) | dlo), 0, (int)(W >> 1), 0x00020000), j0 >> 1, dc);
                }
                if (!DIR && __builtin_expect(live, 1)) {
                    {
                        int pl[CM];
#pragma unroll
                        for (int k = 0; k < CM; k++) pl[k] = (t[k] + jg0 + k * g64) >> 6;
                        store_chunk_i32<CM>(hrow + j0, pl);
                        if (lane == 0 && has_in) hrow[hleft] = (left_now + jg0 - g64) >> 6;   // the wave's own copy of the column on its left
                    }
                }
                DP_T(5);   // stores
                if (__builtin_expect(rare != 0u, 0)) {   // ONE test for everything rare: wide row, far reader, sink (meta & 44)
                    if (DIR && (meta & 32u) && live) {   // wide row: type * 16 + 15 - slot, a byte per cell
                        uint32_t dc[CM];
#pragma unroll
                        for (int k = 0; k < CM; k++) dc[k] = (uint32_t)m[k];
                        store_dirs<CM>(Dwide + (uint64_t)wideslot[i - 1] * W + j0, dc, 0x3f3f3f3fu);
                    }
                    if (meta & 8u) {   // a far successor reads this row back from HBM (keys; with the score matrix it is there already)
                        // (the slot is read out of its lane HERE, where every lane is active: inside the divergent block below a register
                        // that was spilled is reloaded for the active lanes only, and lane ri need not be one of them)
                        const uint32_t fslot = DIR ? __builtin_amdgcn_readlane(fC, ri) : 0u;
                        if constexpr (PRUNE) { far_set(fslot, row_fl); if (row_fl != 0u && (FM & 0x8000u) == 0u) { FM |= 0x8000u; bad |= farref; } }
                        if (DIR && live
                            ) {
                            int32_t* F = H + (uint64_t)fslot * WH;
                            store_chunk_i32<CM>(F + j0, t);
                            if (lane == 0 && has_in) F[hleft] = left_now;
                        }
                        // (no wait: only this wave reads these words back, and a wave's memory instructions reach the cache in program order)
                    }
                    if (owns_last && (meta & 4u)) {   // sink node: candidate end of the global alignment
                        int v = NEGK;
#pragma unroll
                        for (int k = 0; k < CM; k++) if ((uint32_t)k == klast) v = t[k];
                        if (nsink < sink_cap) { sink_row[nsink] = i; sink_score[nsink] = (v >> 6) + (int)L * gap; }
                        nsink++;
                    }
                }
            };
            // Measured (round 6, 12 Mb, A/B of two builds in one GPU call): SLOWER - the longest edge's DP 284 -> 322 M cycles, the step 164 -> 180 ms. Its 303 916 rows
            // are 42 % rows with several predecessors, 30 % rows whose one predecessor sits in the ring (bubbles interleave in rank order), and only 28 % fast rows, in
            // runs of 1.3: what a run costs to set up (mask shift, count, the first predecessor's entry read again: ~100 cycles) and what every other row pays for the
            // test is more than the ~25 instructions a fast row saves. Kept behind -DHX_FAST_ROWS (off) as the measured alternative it is.
#ifdef HX_FAST_ROWS
            constexpr bool FAST_OK = DIR && !PRUNE;
#else
            constexpr bool FAST_OK = false;
#endif
            // fast <=> one predecessor (meta >> META_NP == 1) with the location code 13 (the previous row). Which rows of the batch are is read off the 64 records in
            // their lanes ONCE, as a mask (bit r = row i0 + r, nothing beyond the batch); a run of fast rows is then a counted loop - its back edge is s_sub + s_cmp +
            // one branch (as a test of the next row's record after every row it was a flag-guarded pair of branches and five scalar instructions).
            unsigned long long fastm = 0;
            if constexpr (FAST_OK) {
                fastm = __builtin_amdgcn_ballot_w64(((mC >> META_NP) << 4 | (aC >> 28)) == (1u << 4 | 13u)) >> rb;
                if (nb < 64u) fastm &= (1ull << nb) - 1ull;
            }
            while (rj < nb) {
                if constexpr (FAST_OK) {
                    // (bit nb - rj of the complement is set, so the run ends with the batch at the latest - except for a batch of 64 fast rows seen from its first
                    // row: the complement is zero there, and the count of trailing zeros of zero is not 64 but whatever the instruction leaves)
                    const unsigned long long inv = ~(fastm >> rj);
                    uint32_t run = inv ? (uint32_t)__builtin_ctzll(inv) : 64u;
                    if (__builtin_expect(run != 0u, 1)) {
                        do { row(std::true_type{}); rj++; } while (--run != 0u);
                        p0_nx = __builtin_amdgcn_readlane(aC, (rb + rj) & 63u);
                        continue;
                    }
                }
                row(std::false_type{});
                rj++;
            }
            if constexpr (PRUNE) lazy = (uint32_t)(n_dead - dead_before == nb) & lazy_on;
        }
    }
    if (owns_last) nSinkOut = nsink;
    if constexpr (PRUNE) { if (lane == 0 && pstat) { atomicAdd(&pstat[0], (unsigned long long)V); atomicAdd(&pstat[1], (unsigned long long)n_dead); atomicAdd(&pstat[8], (unsigned long long)n_bulk); } }
}

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
                // Direction bytes: the first wavefront walks the path together. A tile of 32 rows x 16 columns of direction bytes (two registers)
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
                        // tile: 32 rows (ranks ti .. ti-31) x 16 columns (ct-15 .. ct, ct = tj made odd: whole bytes of two nibbles); lane = (row, half):
                        // the 4 bytes = 8 columns of one row in one register
                        const uint32_t ti = i, ct = j | 1u, r0 = ln >> 1, hf = ln & 1u;
                        if (NT > 64 && ln == 0) { st_wg(&lds_u[8], ti); st_wg(&lds_u[9], ct); }   // (where the walk is: the helper wave fetches ahead of it)
                        const int32_t bs = ((int32_t)ct - 15) >> 1;   // first byte of the tile in its rows (ct < 15: negative - bytes before the row, never looked at)
                        uint32_t w0 = 0, mt = 0, pv0 = 0, pv1 = 0, qo = 0, qw = 0;
                        if (ti > r0) {
                            // one unaligned dword load per lane (bytes bs + 4 hf .. + 3 of the row; columns below 0 read the end of the previous row - row 0 exists)
                            const uint8_t* rowp = Dm + (uint64_t)(ti - r0) * (W >> 1) + bs + 4 * (int32_t)hf;
                            __builtin_memcpy(&w0, rowp, 4);
                        }
                        if (ln < 32 && ti > ln) {   // the records of the tile's rows: where a move into the first / second predecessor leads (lanes = rows)
                            const uint32_t rr = ti - 1 - ln;
                            mt = g.row_meta[rr]; qo = g.row_pred_off[rr]; if (mt & 32u) qw = g.wslot[rr];
                            pv0 = (mt >> META_NP) == 0 ? 0u : (g.row_pred0[rr] & 0x0fffffffu) + 1;   // (a source node continues in the virtual row 0)
                            pv1 = (g.row_pred1[rr] & 0x0fffffffu) + 1;
                        }
                        const uint32_t ilo = ti > 31u ? ti - 31u : 1u;                 // the walk goes on while i >= ilo (rows of the tile, never row 0) ...
                        const int32_t jb0 = 2 * bs;                                  // ... and j >= first column of the tile
                        // (bit dr: the tile's row dr has more than 4 predecessors - a scalar bit test per step instead of a readlane of the row's record)
                        const uint32_t wmask = (uint32_t)__ballot(ln < 32 && (mt & 32u) != 0);
                        for (;;) {
                            const uint32_t dr = ti - i, jb = (uint32_t)((int32_t)j - jb0);   // row and column inside the tile (0..31, 0..15)
                            const uint32_t wsel = (uint32_t)__builtin_amdgcn_readlane((int)w0, (int)(dr * 2 + (jb >> 3)));
                            const uint32_t n4 = (wsel >> (4 * (jb & 7u))) & 15u;
                            // move code: type (3 diagonal / 2 vertical / 1 horizontal) * 4 + 3 - predecessor slot; a row with more than 4 predecessors
                            // keeps type * 16 + 15 - slot in the wide-row pool
                            // (a move into the third or a later predecessor - codes 8, 9, 12, 13 of a 4-bit row - has to fetch that predecessor's rank)
                            uint32_t type = n4 >> 2, slot = 3u - (n4 & 3u);
                            // (both candidates read, the choice made by arithmetic: the compiler turns the obvious select into two branches and a flag test)
                            const uint32_t pva = (uint32_t)__builtin_amdgcn_readlane((int)pv0, (int)dr), pvb = (uint32_t)__builtin_amdgcn_readlane((int)pv1, (int)dr);
                            uint32_t pv = pvb ^ ((pva ^ pvb) & (0u - (uint32_t)(slot == 0)));
                            if (__builtin_expect((((0x3300u >> n4) | (wmask >> dr)) & 1u) != 0, 0)) {   // ONE test for the rare moves: a wide row, a third or later predecessor
                                uint32_t later = (0x3300u >> n4) & 1u;
                                if ((wmask >> dr) & 1u) {
                                    const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)Dw[(uint64_t)__builtin_amdgcn_readlane((int)qw, (int)dr) * W + j]);
                                    type = d >> 4; slot = 15u - (d & 15u); later = slot >= 2 && type > 1u;
                                    pv = slot == 0 ? (uint32_t)__builtin_amdgcn_readlane((int)pv0, (int)dr) : (uint32_t)__builtin_amdgcn_readlane((int)pv1, (int)dr);
                                }
                                if (later != 0)
                                    pv = ((uint32_t)__builtin_amdgcn_readfirstlane((int)g.pred_rank[(uint32_t)__builtin_amdgcn_readlane((int)qo, (int)dr) + slot]) & 0x0fffffffu) + 1;
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
                            na++;
                            if (type > 1u) i = pv;                // (a horizontal move is type 1 from the int32 rows, type 0 from the packed ones)
                            j -= type != 2u;                      // (horizontal and diagonal moves go a column to the left, a vertical one does not)
                            if (__builtin_expect((na & 63u) == 0, 0)) {   // 64 entries: out they go (a store per step cost more than the step itself)
                                asm volatile("" ::: "memory");
                                g.aln_node[na - 64 + ln] = pn; g.aln_pos[na - 64 + ln] = pp;
                            }
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
                        for (uint32_t hk = 0; hk < 2; hk++) {                        // rows pi_ - 32 .. pi_ - 159, 64 at a time
                        const uint32_t d = 32u + hl + 64u * hk;
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
                                            uint32_t poll_limit, uint32_t lds_bytes, uint32_t max_indeg, uint32_t dp_lanes, uint32_t prune_pct) {
    __shared__ uint32_t sNext;
    // persistent launch: the bucket whose slot this workgroup owns (slots are laid out bucket by bucket, largest workspace need first)
    uint32_t kb = 0, nbk = 0;
    if (PERSIST) { nbk = btab[0]; while (kb + 1 < nbk && blockIdx.x >= btab[1 + kb]) kb++; }
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
                // longest chains end, and a long chain of little need must not wait behind its bucket's thousands
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
                                                                       q.cns, q.cns_len, q.status, q.cells, q.phase, q.poll_limit, q.ring_bytes, q.max_indeg, q.dp_lanes, q.prune_pct); } while (0)
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
