// dev_skewbench.hip — prototype of the anti-diagonal ("skew-tile") formulation of the POA DP (VERDICT round 2, item 4 iii), standalone:
// a synthetic partial-order graph in rank order, a CPU reference of the row recurrence (same keys, same move codes as kernels/poa.hip),
// the skewed kernel, a cell-by-cell comparison of the move codes and sink scores, and the time of a step.
//
// Formulation. lane = graph rank, step t = column tile + tau(rank), tau = index of the rank's aligned column (every edge of a POA graph runs
// to a later column, so tau(r) >= tau(p) + 1 for every predecessor p). At step t a lane computes the CM cells of tile c = t - tau of its row:
// the tile c of a predecessor with lag d = tau(r) - tau(p) was finished d steps earlier, and the horizontal dependency is the lane's own
// previous tile - no prefix scan, no carry mailbox. Finished tiles go to an LDS ring keyed by tile index (K tiles per lane); a workgroup
// ("band" of NT ranks) runs its waves in lock step (one LDS-only barrier per step). Predecessors outside the band, or further back than the
// ring reaches, come from HBM: a row somebody outside needs is "exported" (64-bit tagged words, whole row) and read back by the consumer
// lane itself; the tag makes every word self-validating, so bands of one graph run concurrently on different workgroups ("members").
//
// Development tool: hipcc --offload-arch=gfx950 -O3 -o tools/dev_skewbench tools/dev_skewbench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int32_t NEGK = -(1 << 30);
constexpr int MATCH = 5, MISMATCH = -4, GAP = -8;

struct SkewRow {        // per rank, 32 bytes
    uint32_t meta;      // code | sink << 2 | export << 3 | npred << 12
    uint32_t tau;
    uint32_t pref[4];   // kind << 30 | payload. kind 0: ring, payload = lag << 16 | lane in band; 1: exported row, payload = export slot; 2: virtual row 0
    uint32_t xslot;     // own export slot
    uint32_t sink_idx;
};
struct BandInfo { uint32_t t0, t1; uint64_t d_off; uint32_t nstage, pad; };   // first / last tau of the band's ranks; offset of its move codes (16-bit units); staging rows in use

struct StageRef;
struct SkewArgs {
    const SkewRow* rows; const BandInfo* bands; const StageRef* stage; uint32_t V, L, nbands;
    const uint8_t* seq;              // bases (0..3), L of them
    int32_t* xrows;                  // exported rows: slot * XW + column, keys with a 6-bit tag in the low bits
    uint32_t XW; uint32_t tagbase;
    uint16_t* dg;                    // move codes, skewed: band offset + (step * NT + lane)
    int32_t* sink_score;             // by sink index
    unsigned long long* prof;        // per workgroup: cycles, steps, retries
};

// ring geometry: K slots (tile index & (K - 1)) of NT + 1 tiles; tile NT of every slot stays "minus infinity" (a lane with fewer predecessors than
// its wave goes through points its spare slots there); the whole ring is reset to minus infinity at the start of a band, which is also what a
// predecessor's tile -1 must read as.
constexpr int NSTAGE = 64;    // staging rows per band: tiles of rows that live in HBM, kept a few tiles ahead of their reader by the band's helper wavefront (one lane per row)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
template <int CM> struct TileT;
template <> struct TileT<4> { using type = int4; using vec = v4i; };
template <> struct TileT<2> { using type = int2; using vec = v2i; };

template <int CM, int K, int NT, int S, bool BARRIER>
__device__ __forceinline__ void band_steps(const SkewArgs& a, const BandInfo bi, const uint32_t ntiles, int32_t* ring, const uint8_t* seq_lds, const bool valid, const SkewRow& row,
                                           const bool wave_src) {
    using tile_t = typename TileT<CM>::type;
    constexpr uint32_t TB = CM * 4, ROWB = (NT + 1 + NSTAGE) * TB;  // bytes of a tile, of a ring slot (NT lanes, the minus-infinity tile, NSTAGE staging rows)
    const uint32_t tid = threadIdx.x;
    const int m64 = MATCH * 64, mm64 = MISMATCH * 64, g64 = GAP * 64;
    constexpr int kd = 15, kv = 11, kh = 4;                          // 4-bit move codes: type * 4 + 3 - slot
    const uint32_t npred = valid ? (row.meta >> 12) : 0u, tau = row.tau;
    const uint32_t codepat = (row.meta & 3u) * 0x55u;
    const bool is_sink = row.meta & 4u, is_exp = row.meta & 8u;
    uint32_t pr[4] = {row.pref[0], row.pref[1], row.pref[2], row.pref[3]};
    // byte offset of every predecessor slot's lane inside a ring slot; anything that is not a ring reference reads the minus-infinity tile
    uint32_t aoff[S];
#pragma unroll
    for (int q = 0; q < S; q++) {
        const uint32_t kind = (uint32_t)q < npred ? pr[q] >> 30 : 3u;
        // ring reference: the predecessor's lane; exported row: this reference's staging row (bits 24..29 of the entry); anything else: the minus-infinity tile
        aoff[q] = (kind == 0u ? (pr[q] & (NT - 1)) : kind == 1u ? NT + 1 + ((pr[q] >> 24) & 63u) : (uint32_t)NT) * TB;
    }

    const uint32_t woff = tid * TB;
    const bool src = valid && npred == 0;
    int hprev = NEGK;
    const char* ringb = reinterpret_cast<const char*>(ring);
    char* ringw = reinterpret_cast<char*>(ring);
    uint16_t* dgp = a.dg + bi.d_off + tid;
    const uint32_t t_end = bi.t1 + ntiles;                             // exclusive
    uint32_t nibacc = 0;
    for (uint32_t t = bi.t0; t < t_end; t++) {
        const uint32_t c = t - tau;
        const bool act = valid && c < ntiles;
        const uint32_t sl = (c & (K - 1)) * ROWB, slp = ((c - 1) & (K - 1)) * ROWB;
        const uint32_t b = seq_lds[min(c, ntiles)];                    // (entry ntiles exists: inactive lanes read it)
        int P[S][CM], Lf[S];
#pragma unroll
        for (int q = 0; q < S; q++) {
            const tile_t v = *reinterpret_cast<const tile_t*>(ringb + sl + aoff[q]);
            if constexpr (CM == 4) { P[q][0] = v.x; P[q][1] = v.y; P[q][2] = v.z; P[q][3] = v.w; } else { P[q][0] = v.x; P[q][1] = v.y; }
            Lf[q] = *reinterpret_cast<const int32_t*>(ringb + slp + aoff[q] + (CM - 1) * 4);
        }
        const int j0 = (int)(c * CM);
        if (wave_src) {                                                // a source node starts from the virtual row 0 (rare: one branch per wave)
#pragma unroll
            for (int k = 0; k < CM; k++) P[0][k] = src ? (j0 + k) * g64 : P[0][k];
            Lf[0] = src ? (c ? (j0 - 1) * g64 : NEGK) : Lf[0];
        }
        // substitution score of the tile's columns: match unless the base under the column differs from the row's letter
        const uint32_t x = b ^ codepat, mis = x | (x >> 1);
        int m[CM], hm[CM];
#pragma unroll
        for (int k = 0; k < CM; k++) {
            const int dlt = __builtin_amdgcn_sbfe((int)mis, 2 * k, 1) & (mm64 - m64);          // 0 or (mismatch - match)
            // every candidate of the column in one chain of maxima: diagonal / vertical per predecessor slot (the slot lowers the move code), then the horizontal move
            int best = NEGK;
#pragma unroll
            for (int q = 0; q < S; q++) {
                const int dg = (k == 0 ? Lf[q] : P[q][k - 1]) + dlt + (m64 + kd - q);
                const int vt = P[q][k] + (g64 + kv - q);
                best = q == 0 ? max(dg, vt) : max(best, max(dg, vt));
            }
            const int hz = (k == 0 ? (hprev & ~63) : hm[k - 1]) + (g64 + kh);
            m[k] = max(best, hz);
            hm[k] = m[k] & ~63;
        }
        uint32_t nib = 0;
#pragma unroll
        for (int k = CM - 1; k >= 0; k--) nib = (nib << 4) | ((uint32_t)m[k] & 15u);
        if (act) {
            hprev = m[CM - 1];
            tile_t o;
            if constexpr (CM == 4) o = make_int4(hm[0], hm[1], hm[2], hm[3]); else o = make_int2(hm[0], hm[1]);
            *reinterpret_cast<tile_t*>(ringw + sl + woff) = o;
            if (is_exp) {   // somebody outside the ring's reach reads this row: the tile goes to HBM, every value tagged with (tile + salt) & 63 in its low bits
                const int tg = (int)((c + a.tagbase) & 63u);
                tile_t* X = reinterpret_cast<tile_t*>(a.xrows + (uint64_t)row.xslot * a.XW + (uint32_t)j0);
                typename TileT<CM>::vec ov;
                if constexpr (CM == 4) ov = v4i{hm[0] | tg, hm[1] | tg, hm[2] | tg, hm[3] | tg}; else ov = v2i{hm[0] | tg, hm[1] | tg};
                if constexpr (CM == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(X), "v"(ov) : "memory");
                else asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(X), "v"(ov) : "memory");
            }
            if (is_sink && (uint32_t)j0 <= a.L && a.L < (uint32_t)j0 + CM) {
                int v = NEGK;
#pragma unroll
                for (int k = 0; k < CM; k++) if ((uint32_t)(j0 + k) == a.L) v = m[k];
                a.sink_score[row.sink_idx] = v >> 6;
            }
        }
        if constexpr (CM == 4) dgp[(uint64_t)(t - bi.t0) * NT] = (uint16_t)nib;
        else {   // two steps per 16-bit word (low byte = the even step of the band)
            const uint32_t s = t - bi.t0;
            nibacc = (s & 1u) ? (nibacc | (nib << 8)) : nib;
            if ((s & 1u) || t + 1 == t_end) dgp[(uint64_t)(s >> 1) * NT] = (uint16_t)nibacc;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

// The band's helper wavefront: lane i keeps staging row i (tiles of one exported row, read by one lane of the band) a few tiles ahead of its reader.
// It takes part in the band's barrier per step and simply does not arrive while a tile the next step needs is missing - the compute waves wait for
// it there, without a single instruction of their own spent on rows that live in HBM.
struct StageRef { uint32_t xslot, tau; };             // exported row, tau of the reading lane
template <int CM, int K, int NT>
__device__ __forceinline__ void band_helper(const SkewArgs& a, const BandInfo bi, const uint32_t ntiles, int32_t* ring, const StageRef* refs, const uint32_t nrefs,
                                            unsigned long long& retries, bool& gave_up) {
    using tile_t = typename TileT<CM>::type;
    constexpr uint32_t TB = CM * 4, ROWB = (NT + 1 + NSTAGE) * TB;
    const uint32_t lane = threadIdx.x & 63u;
    const bool mine = lane < nrefs;
    StageRef ref{0, 0};
    if (mine) ref = refs[lane];
    char* ringw = reinterpret_cast<char*>(ring);
    const uint32_t aoff = (NT + 1 + lane) * TB;
    const int32_t* xrow = a.xrows + (uint64_t)ref.xslot * a.XW;
    uint32_t fnext = 0;                                 // first tile not staged yet
    const uint32_t t_end = bi.t1 + ntiles;
    for (uint32_t t = bi.t0; t <= t_end; t++) {         // iteration t prepares step t (the compute waves are in step t - 1 meanwhile); the last one only meets the barrier
        const int ci = (int)t - (int)ref.tau;           // the reader's tile in step t
        if (t < t_end) {
            for (uint32_t spin = 0;; spin++) {
                // room: the readers of step t - 1 still look at tiles ci - 2 and ci - 1; a batch [f, f + 4) overwrites the slots of f - 8 .. f - 5
                const bool fill = mine && fnext < ntiles && (int)fnext <= ci + 2;
                if (fill) {
                    const int32_t* X = xrow + (uint64_t)fnext * CM;
                    typename TileT<CM>::vec w[4];
                    if constexpr (CM == 4)
                        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\tglobal_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
                                     "global_load_dwordx4 %3, %4, off offset:48 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(X) : "memory");
                    else
                        asm volatile("global_load_dwordx2 %0, %4, off sc1\n\tglobal_load_dwordx2 %1, %4, off offset:8 sc1\n\tglobal_load_dwordx2 %2, %4, off offset:16 sc1\n\t"
                                     "global_load_dwordx2 %3, %4, off offset:24 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(X) : "memory");
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t tg = (fnext + i + a.tagbase) & 63u;
                        uint32_t bad = ((uint32_t)w[i].x ^ tg) | ((uint32_t)w[i].y ^ tg);
                        if constexpr (CM == 4) bad |= ((uint32_t)w[i].z ^ tg) | ((uint32_t)w[i].w ^ tg);
                        ok = ok && (fnext + i >= ntiles || (bad & 63u) == 0u);
                    }
                    if (ok) {
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            typename TileT<CM>::vec o = w[i] & ~63;
                            *reinterpret_cast<typename TileT<CM>::vec*>(ringw + ((fnext + i) & (K - 1)) * ROWB + aoff) = o;
                        }
                        fnext += 4;
                    }
                }
                const bool need = mine && ci >= 0 && (uint32_t)ci < ntiles && (uint32_t)ci >= fnext;   // the reader's tile of step t is not there yet
                if (__ballot(need) == 0ull) break;
                retries++;
                if (spin > (1u << 22)) { gave_up = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

template <int CM, int K, int NT>
__global__ void __launch_bounds__(NT + 64) k_skew(SkewArgs a, uint32_t members, uint32_t xcd_stride) {
    extern __shared__ int4 ring4[];                   // K * (NT + 1 + NSTAGE) tiles, then the sequence
    int32_t* ring = reinterpret_cast<int32_t*>(ring4);
    uint8_t* seq_lds = reinterpret_cast<uint8_t*>(ring + K * (NT + 1 + NSTAGE) * CM);   // packed: byte c = bases of the columns CM c .. CM c + CM - 1 (column j carries base j - 1)
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const bool helper = tid >= NT;                    // the last wavefront of the workgroup stages exported rows, the others compute
    // xcd_stride 8: workgroups go to the 8 XCDs round-robin by index, so only every 8th index works - all members of the edge share one XCD (one L2)
    if (blockIdx.x % xcd_stride) return;
    const uint32_t wg = blockIdx.x / xcd_stride, mem = wg % members;
    const uint32_t L = a.L, ntiles = (L + 1 + CM - 1) / CM;
    for (uint32_t c = tid; c <= ntiles; c += NT + 64) {
        uint32_t b = 0;
        for (int k = 0; k < CM; k++) { const uint32_t j = c * CM + k; const uint32_t base = (j >= 1 && j <= L) ? a.seq[j - 1] : 0u; b |= base << (2 * k); }
        seq_lds[c] = (uint8_t)b;
    }
    unsigned long long retries = 0, steps = 0;
    bool gave_up = false;
    const long long tc0 = clock64();
    for (uint32_t band = mem; band < a.nbands; band += members) {
        const BandInfo bi = a.bands[band];
        const uint32_t r = band * NT + tid;
        const bool valid = !helper && r < a.V;
        SkewRow row;
        if (valid) row = a.rows[r]; else { row.meta = 0; row.tau = 0; row.pref[0] = row.pref[1] = row.pref[2] = row.pref[3] = 2u << 30; row.xslot = 0; row.sink_idx = 0; }
        const uint32_t npred = valid ? (row.meta >> 12) : 0u;
        // wave-uniform number of predecessor slots to go through, and whether any lane is a source
        uint32_t maxnp = max(npred, 1u);
        for (int o = 32; o; o >>= 1) maxnp = max(maxnp, (uint32_t)__shfl_xor((int)maxnp, o));
        maxnp = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxnp);
        const bool wave_src = __ballot(valid && npred == 0) != 0ull;
        __syncthreads();                               // (everybody has left the previous band; seq_lds is ready)
        for (uint32_t i = tid; i < K * (NT + 1 + NSTAGE) * CM; i += NT + 64) ring[i] = NEGK;
        __syncthreads();
        if (helper) band_helper<CM, K, NT>(a, bi, ntiles, ring, a.stage + (uint64_t)band * NSTAGE, bi.nstage, retries, gave_up);
        else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();              // (the helper has staged what the first step reads)
            if (maxnp <= 1) band_steps<CM, K, NT, 1, true>(a, bi, ntiles, ring, seq_lds, valid, row, wave_src);
            else if (maxnp == 2) band_steps<CM, K, NT, 2, true>(a, bi, ntiles, ring, seq_lds, valid, row, wave_src);
            else if (maxnp == 3) band_steps<CM, K, NT, 3, true>(a, bi, ntiles, ring, seq_lds, valid, row, wave_src);
            else band_steps<CM, K, NT, 4, true>(a, bi, ntiles, ring, seq_lds, valid, row, wave_src);
        }
        steps += bi.t1 + ntiles - bi.t0;
    }
    if (tid == 0) { a.prof[wg * 3 + 0] = (unsigned long long)(clock64() - tc0); a.prof[wg * 3 + 1] = steps; }
    if (lane == 0 && retries) atomicAdd(&a.prof[wg * 3 + 2], retries);
    if (gave_up && lane == 0) atomicAdd(&a.prof[wg * 3 + 2], 1ull << 60);   // (shows as an absurd retry count)
}

// ------------------------------------------------------------------------------------------------ host
struct Graph {
    uint32_t V = 0;
    std::vector<uint8_t> code, sink;
    std::vector<uint32_t> col;
    std::vector<std::vector<uint32_t>> pred;
};

// a POA-like graph: columns of 1-3 aligned alternatives, edges from the previous column, deletion edges over 1-5 columns, rare far ones
static Graph make_graph(uint32_t ncols, std::mt19937& g, double far_p) {
    Graph G;
    std::uniform_real_distribution<double> U(0, 1);
    std::vector<std::vector<uint32_t>> cols;
    for (uint32_t c = 0; c < ncols; c++) {
        const double u = U(g);
        const uint32_t n = u < 0.62 ? 1 : u < 0.90 ? 2 : u < 0.98 ? 3 : 4;
        std::vector<uint32_t> ids;
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t r = G.V++;
            G.code.push_back((uint8_t)((g() >> 7) & 3)); G.sink.push_back(0); G.col.push_back(c); G.pred.emplace_back();
            ids.push_back(r);
            if (c > 0) {
                auto add = [&](uint32_t p) { auto& pv = G.pred[r]; if (pv.size() < 4 && std::find(pv.begin(), pv.end(), p) == pv.end()) pv.push_back(p); };
                const auto& prev = cols[c - 1];
                add(prev[g() % prev.size()]);
                if (U(g) < 0.25) add(prev[g() % prev.size()]);
                if (U(g) < 0.18 && c > 1) { const uint32_t back = 2 + g() % std::min<uint32_t>(5, c - 1); const auto& pc = cols[c - back]; add(pc[g() % pc.size()]); }
                if (U(g) < far_p && c > 8) { const uint32_t back = 7 + g() % std::min<uint32_t>(60, c - 7); const auto& pc = cols[c - back]; add(pc[g() % pc.size()]); }
                if (U(g) < 0.01) G.pred[r].clear();   // a source in the middle (a sequence that started late)
            }
        }
        cols.push_back(ids);
    }
    std::vector<uint8_t> has_succ(G.V, 0);
    for (uint32_t r = 0; r < G.V; r++) for (uint32_t p : G.pred[r]) has_succ[p] = 1;
    for (uint32_t r = 0; r < G.V; r++) G.sink[r] = !has_succ[r];
    return G;
}

// the row recurrence of kernels/poa.hip on keys (64 x score + move code): returns the 4-bit move code of every cell and the sink scores
static void cpu_reference(const Graph& G, const std::vector<uint8_t>& seq, std::vector<uint8_t>& nib, std::vector<int32_t>& sink_score) {
    const uint32_t L = (uint32_t)seq.size(), W = L + 1;
    std::vector<int32_t> H((size_t)G.V * W);
    nib.assign((size_t)G.V * W, 0);
    sink_score.clear();
    const int m64 = MATCH * 64, mm64 = MISMATCH * 64, g64 = GAP * 64;
    for (uint32_t r = 0; r < G.V; r++) {
        int32_t* h = &H[(size_t)r * W];
        for (uint32_t j = 0; j <= L; j++) {
            int best = NEGK;
            const size_t np = G.pred[r].size();
            for (size_t q = 0; q < std::max<size_t>(1, np); q++) {
                int pj, pjm1;
                if (np == 0) { pj = (int)j * g64; pjm1 = j ? ((int)j - 1) * g64 : NEGK; }
                else { const int32_t* hp = &H[(size_t)G.pred[r][q] * W]; pj = hp[j]; pjm1 = j ? hp[j - 1] : NEGK; }
                const int sd = ((j >= 1 && seq[j - 1] == G.code[r]) ? m64 : mm64) + 15 - (int)q;
                best = std::max(best, std::max(pjm1 + sd, pj + g64 + 11 - (int)q));
            }
            if (j) best = std::max(best, h[j - 1] + g64 + 4);
            nib[(size_t)r * W + j] = (uint8_t)(best & 15);
            h[j] = best & ~63;
        }
        if (G.sink[r]) sink_score.push_back(h[L] >> 6);
    }
}

int main(int argc, char** argv) {
    uint32_t ncols = 3000, L = 2000, members = 4, seed = 1, reps = 3, nedges = 1, xcd = 1, local_only = 0, nt = 256, cm = 4;
    double far_p = 0.003;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--cols")) ncols = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--len")) L = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--members")) members = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--seed")) seed = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--edges")) nedges = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--far")) far_p = atof(argv[i + 1]);
        else if (!strcmp(argv[i], "--xcd")) xcd = atoi(argv[i + 1]) ? 8 : 1;
        else if (!strcmp(argv[i], "--nt")) nt = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--cm")) cm = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--local")) local_only = atoi(argv[i + 1]);   // drop every predecessor the LDS ring cannot serve (the pure step cost)
    }
    constexpr int K = 8;
    const uint32_t NT = nt, CM = cm;
    std::mt19937 g(seed);
    Graph G = make_graph(ncols, g, far_p);
    std::vector<uint8_t> seq(L);
    for (auto& b : seq) b = (uint8_t)((g() >> 9) & 3);
    const uint32_t V = G.V, nbands = (V + NT - 1) / NT, ntiles = (L + 1 + CM - 1) / CM;
    if (local_only)
        for (uint32_t r = 0; r < V; r++) {
            auto& pv = G.pred[r];
            pv.erase(std::remove_if(pv.begin(), pv.end(), [&](uint32_t p) { return p / NT != r / NT || G.col[r] - G.col[p] > K - 2; }), pv.end());
        }
    {
        std::vector<uint8_t> has_succ(V, 0);
        for (uint32_t r = 0; r < V; r++) for (uint32_t p : G.pred[r]) has_succ[p] = 1;
        for (uint32_t r = 0; r < V; r++) G.sink[r] = !has_succ[r];
    }
    // ---- the skew plan (the CSR phase of the real kernel would build this on the device)
    std::vector<SkewRow> rows(V);
    std::vector<uint32_t> xslot(V, 0xffffffffu);
    uint32_t nx = 0, nremote = 0, nlocal = 0, nsink = 0;
    for (uint32_t r = 0; r < V; r++)
        for (uint32_t p : G.pred[r]) {
            const uint32_t d = G.col[r] - G.col[p];
            if (d < 1) { fprintf(stderr, "graph: edge inside a column\n"); return 1; }
            if (p / NT != r / NT || d > K - 2) { if (xslot[p] == 0xffffffffu) xslot[p] = nx++; nremote++; } else nlocal++;
        }
    std::vector<uint32_t> stage_used(nbands, 0);
    std::vector<StageRef> stage((size_t)nbands * NSTAGE, StageRef{0, 0});
    for (uint32_t r = 0; r < V; r++) {
        SkewRow& R = rows[r];
        R.meta = G.code[r] | (G.sink[r] ? 4u : 0u) | (xslot[r] != 0xffffffffu ? 8u : 0u) | ((uint32_t)G.pred[r].size() << 12);
        R.tau = G.col[r];
        for (uint32_t q = 0; q < 4; q++) {
            if (q >= G.pred[r].size()) { R.pref[q] = 2u << 30; continue; }
            const uint32_t p = G.pred[r][q], d = G.col[r] - G.col[p];
            if (p / NT != r / NT || d > K - 2) {
                const uint32_t sid = stage_used[r / NT]++;
                if (sid >= (uint32_t)NSTAGE || xslot[p] >= (1u << 24)) { fprintf(stderr, "band %u needs more than %d staging rows\n", r / NT, NSTAGE); return 1; }
                R.pref[q] = (1u << 30) | (sid << 24) | xslot[p];
                stage[(size_t)(r / NT) * NSTAGE + sid] = StageRef{xslot[p], G.col[r]};
            } else R.pref[q] = (d << 16) | (p % NT);
        }
        R.xslot = xslot[r] == 0xffffffffu ? 0 : xslot[r];
        R.sink_idx = G.sink[r] ? nsink++ : 0;
    }
    std::vector<BandInfo> bands(nbands);
    uint64_t dtotal = 0, steps_total = 0;
    for (uint32_t b = 0; b < nbands; b++) {
        uint32_t t0 = ~0u, t1 = 0;
        for (uint32_t r = b * NT; r < std::min(V, (b + 1) * NT); r++) { t0 = std::min(t0, G.col[r]); t1 = std::max(t1, G.col[r]); }
        bands[b] = {t0, t1, dtotal, stage_used[b], 0};
        dtotal += (uint64_t)((t1 - t0 + ntiles) * CM / 4 + 1) * NT;   // 16-bit words: CM = 4 one per step and lane, CM = 2 one per two steps
        steps_total += t1 - t0 + ntiles;
    }
    const uint32_t XW = ((ntiles * CM + 16) + 7) & ~7u;
    printf("graph: %u ranks in %u columns (%.2f per column), %u bands of %u; L = %u (%u tiles); predecessor refs: %u ring, %u exported (%u rows exported)\n", V, ncols, (double)V / ncols, nbands, NT,
           L, ntiles, nlocal, nremote, nx);
    printf("steps: sum over bands %llu; critical path with enough members: %u (columns + tiles); row kernel: %u rows\n", (unsigned long long)steps_total, ncols + ntiles, V);
    // ---- reference
    std::vector<uint8_t> nib_ref; std::vector<int32_t> sink_ref;
    cpu_reference(G, seq, nib_ref, sink_ref);
    // ---- device
    SkewRow* d_rows; BandInfo* d_bands; uint8_t* d_seq; unsigned long long *d_prof; int32_t* d_x; uint16_t* d_dg; int32_t* d_sink;
    CHK(hipMalloc(&d_rows, V * sizeof(SkewRow))); CHK(hipMalloc(&d_bands, nbands * sizeof(BandInfo))); CHK(hipMalloc(&d_seq, L));
    CHK(hipMalloc(&d_x, std::max<uint64_t>(1, (uint64_t)nx * XW + 8) * 4)); CHK(hipMalloc(&d_prof, (size_t)nedges * members * 3 * 8));
    CHK(hipMalloc(&d_dg, dtotal * 2 * nedges)); CHK(hipMalloc(&d_sink, std::max<uint32_t>(1, nsink) * 4 * nedges));
    CHK(hipMemcpy(d_rows, rows.data(), V * sizeof(SkewRow), hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_bands, bands.data(), nbands * sizeof(BandInfo), hipMemcpyHostToDevice));
    StageRef* d_stage; CHK(hipMalloc(&d_stage, stage.size() * sizeof(StageRef)));
    CHK(hipMemcpy(d_stage, stage.data(), stage.size() * sizeof(StageRef), hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_seq, seq.data(), L, hipMemcpyHostToDevice));
    CHK(hipMemset(d_x, 0xff, std::max<uint64_t>(1, (uint64_t)nx * XW + 8) * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (uint32_t rep = 0; rep < reps; rep++) {
        SkewArgs a{d_rows, d_bands, d_stage, V, L, nbands, d_seq, d_x, XW, (1u + rep) * 37u, d_dg, d_sink, d_prof};
        CHK(hipMemset(d_prof, 0, (size_t)nedges * members * 3 * 8));
        CHK(hipMemset(d_dg, 0xff, dtotal * 2));
        CHK(hipEventRecord(e0));
        const size_t lds = (size_t)K * (NT + 1 + NSTAGE) * CM * 4 + ntiles + 32;
#define LAUNCH(CMV, NTV) do { CHK(hipFuncSetAttribute((const void*)k_skew<CMV, K, NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
                         k_skew<CMV, K, NTV><<<nedges * members * xcd, NTV + 64, lds>>>(a, members, xcd); } while (0)
#define LAUNCH_NT(CMV) do { if (NT == 64) LAUNCH(CMV, 64); else if (NT == 256) LAUNCH(CMV, 256); else if (NT == 512) LAUNCH(CMV, 512); else { fprintf(stderr, "--nt 64|256|512\n"); return 2; } } while (0)
        if (CM == 4) LAUNCH_NT(4); else if (CM == 2) LAUNCH_NT(2); else { fprintf(stderr, "--cm 2|4\n"); return 2; }
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> prof((size_t)nedges * members * 3);
        CHK(hipMemcpy(prof.data(), d_prof, prof.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long cyc = 0, st = 0, rt = 0;
        for (uint32_t i = 0; i < nedges * members; i++) { cyc = std::max(cyc, prof[i * 3]); st = std::max(st, prof[i * 3 + 1]); rt += prof[i * 3 + 2]; }
        printf("rep %u: %.3f ms, longest workgroup %llu cycles for %llu steps = %.0f cycles per step; %.2f cycles per row of the graph; retries %llu; %.1f GCUPS (x%u edges)\n", rep, ms, cyc, st,
               (double)cyc / st, (double)cyc / V, rt, (double)V * (L + 1) * nedges / ms / 1e6, nedges);
    }
    // ---- compare (edge 0)
    std::vector<uint16_t> dg(dtotal); std::vector<int32_t> sink(std::max<uint32_t>(1, nsink));
    CHK(hipMemcpy(dg.data(), d_dg, dtotal * 2, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(sink.data(), d_sink, nsink * 4, hipMemcpyDeviceToHost));
    uint64_t bad = 0;
    for (uint32_t r = 0; r < V; r++) {
        const BandInfo& bi = bands[r / NT];
        for (uint32_t j = 0; j <= L; j++) {
            const uint32_t c = j / CM, k = j % CM, s = G.col[r] - bi.t0 + c;
            const uint32_t got = CM == 4 ? (dg[bi.d_off + (uint64_t)s * NT + r % NT] >> (4 * k)) & 15u : (dg[bi.d_off + (uint64_t)(s >> 1) * NT + r % NT] >> (8 * (s & 1) + 4 * k)) & 15u;
            if (got != nib_ref[(size_t)r * (L + 1) + j]) { if (bad < 5) printf("MISMATCH rank %u (col %u) column %u: got %u want %u\n", r, G.col[r], j, got, nib_ref[(size_t)r * (L + 1) + j]); bad++; }
        }
    }
    uint32_t sbad = 0;
    for (uint32_t i = 0; i < nsink; i++) sbad += sink[i] != sink_ref[i];
    printf("%s: %llu of %llu move codes differ, %u of %u sink scores differ\n", bad + sbad ? "FAIL" : "OK", (unsigned long long)bad, (unsigned long long)V * (L + 1), sbad, nsink);
    return bad + sbad ? 1 : 0;
}
