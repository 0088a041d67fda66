// poa_graph.inl - part of kernels/poa.hip (included there, inside namespace hxk::<anonymous>): the partial-order graph of an edge - its views into the pools, nodes / edges /
// aligned groups (spoa's add_alignment), the reference's topological order (serial and as one wavefront in lock step on ranks), heaviest bundle + branch completion.
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

// The walk back from rank `best` along the chosen predecessors, by the whole wavefront: every lane gets the length. A lane on its own pays a dependent
// round trip per node, twice (length first, then the bases back to front): 4-5 M cycles of the 14 M the consensus of the longest 12 Mb edge took. Here the
// predecessors and bases of the 64 ranks around the walk are fetched at once, the walk inside them runs on v_readlane, the bases are kept back to front
// in the alignment-entry array (free by now, a word per node at least: need_of) and turned round by all lanes.
__device__ uint32_t bundle_backtrack(G& g, const uint32_t best, char* out) {
    const uint32_t lane = threadIdx.x & 63u;
    const int32_t* pr_r = g.pred;
    int32_t* rev = g.aln_node;
    int32_t r = __builtin_amdgcn_readfirstlane((int)best);
    uint32_t len = 0;
    int acc = 0;
    while (r != -1) {
        const uint32_t cb = (uint32_t)r & ~63u, idx = min(cb + lane, (uint32_t)r);   // (ranks above the walk are never looked at)
        const int p = pr_r[idx], b = (int)(g.row_meta[idx] & 3u);
        while (r >= (int32_t)cb) {
            const int l = r - (int32_t)cb;
            const int bl = __builtin_amdgcn_readlane(b, l);
            acc = lane == (len & 63u) ? bl : acc;
            len++;
            if ((len & 63u) == 0) rev[len - 64 + lane] = acc;
            r = __builtin_amdgcn_readlane(p, l);
        }
    }
    if (lane < (len & 63u)) rev[(len & ~63u) + lane] = acc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (uint32_t k = lane; k < len; k += 64) out[k] = "ACGT"[rev[len - 1 - k] & 3];
    return len;
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

