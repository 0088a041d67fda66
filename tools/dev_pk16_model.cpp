// dev_pk16_model.cpp — CPU model of the packed 16-bit POA row body (kernels/poa.hip, dp_rows16) on synthetic POA-shaped DAGs, checked cell by cell
// against the plain int32 recurrence with the reference's tie rules. Development tool: the arithmetic of the HIP kernel is written here first, instruction
// by instruction (every helper below is one VALU instruction on a 32-bit register holding two 16-bit columns), so that what the GPU run has to
// prove is the implementation, not the mathematics.
//
// The formulation. Lane t of wave w owns CM contiguous columns j0 .. j0 + CM - 1 (j0 = (64 w + t) CM); register q (0 <= q < h = CM / 2) holds column
// j0 + q in its low half and column j0 + q + h in its high half ("split" layout: the left neighbour of a register's two columns is the previous
// register - no per-register alignment instruction). A cell is a 16-bit key
//       K = 4 * Xr + type + OFF (unsigned),      Xr = H[i][j] - B[i] - G * (j - c0 + 1)
// with c0 the wave's first column, B[i] = H[i][c0 - 1] the finished score just LEFT of the wave in row i (what the wave on the left hands over
// anyway; the first wave uses B[i] = H[i][0] - G, a virtual column -1), G the gap score, type 3 diagonal / 2 vertical / 0 horizontal. Xr is the
// score relative to the wave's own left edge with the gap ramp taken out: 0 <= Xr <= (M - 2 G) * 64 CM for ANY gap length (a horizontal step
// changes H by at least G and at most M - G), so 4 Xr + type fits 16 bits for 64 CM <= 512 columns with the reference's scores - no gap-length
// limit, no fallback. The recurrences in this frame (dq = 4 (B[p] - B[i]) <= -4 G for every predecessor p; the constants carry +(-4 G) and
// (-4 G) - dq >= 0 is SUBTRACTED with unsigned saturation: exact wherever the result can matter, 0 = "nothing" below):
//       diagonal    K = (Kp[j-1] & ~3) + 4 (s - G) + 3 + dq
//       vertical    K = (Kp[j]   & ~3) + 4 G + 2 + dq
//       horizontal  K = (K[j-1]  & ~3)                      (type 0; Xr does not change along a horizontal move: a prefix maximum)
// One max() per decision keeps the reference's tie order diagonal > vertical > horizontal; the first predecessor in in-edge order wins among
// equals (a strict comparison moves the slot).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef uint32_t u32;
static inline int lo16(u32 a) { return (int)(a & 0xffff); }
static inline int hi16(u32 a) { return (int)(a >> 16); }
static inline u32 mk(int lo, int hi) { return ((u32)(uint16_t)lo) | ((u32)(uint16_t)hi << 16); }
static bool g_wrapped = false;
static inline int wrap16(int v) { if (v > 65535 || v < 0) g_wrapped = true; return v & 0xffff; }
// v_pk_add_u16 (wrapping: the model flags any wrap - none may happen), v_pk_sub_u16 clamp, v_pk_max_u16, v_pk_min_u16, v_pk_mul_lo_u16
static inline u32 pk_add(u32 a, u32 b) { return mk(wrap16(lo16(a) + lo16(b)), wrap16(hi16(a) + hi16(b))); }
static inline u32 pk_sub_sat(u32 a, u32 b) { return mk(std::max(0, lo16(a) - lo16(b)), std::max(0, hi16(a) - hi16(b))); }
static inline u32 pk_max(u32 a, u32 b) { return mk(std::max(lo16(a), lo16(b)), std::max(hi16(a), hi16(b))); }
static inline u32 pk_min(u32 a, u32 b) { return mk(std::min(lo16(a), lo16(b)), std::min(hi16(a), hi16(b))); }
static inline u32 pk_mul(u32 a, u32 b) { return mk((uint16_t)((a & 0xffff) * (b & 0xffff)), (uint16_t)((a >> 16) * (b >> 16))); }
static inline u32 alignbit16(u32 hi, u32 lo) { return (lo >> 16) | (hi << 16); }   // v_alignbit_b32 hi, lo, 16

constexpr int OFF = 512;   // key of Xr = 0, type 0 (unsigned 16-bit keys: everything a row keeps is >= this; 0 = "nothing")
constexpr int NEGI = -(1 << 29);

struct Dag {
    int V;
    std::vector<uint8_t> code;                 // letter per rank
    std::vector<std::vector<int>> pred;        // predecessors (ranks, in-edge order) per rank; empty = source (virtual row 0)
};

int main(int argc, char** argv) {
    const int n_cases = argc > 1 ? atoi(argv[1]) : 40;
    const int M = 5, X = -4, G = -8;
    std::mt19937 rng(12345);
    long long cells_checked = 0;
    int max_key_seen = 0;
    for (int cs = 0; cs < n_cases; cs++) {
        const int CM = (cs & 1) ? 8 : 4, h = CM / 2;
        const int L = 1 + (int)(rng() % (cs % 5 == 0 ? 3000 : cs % 7 == 3 ? 6000 : 900));
        const int V = 1 + (int)(rng() % (2 * L + 50));
        Dag d; d.V = V; d.code.resize(V); d.pred.resize(V);
        std::vector<uint8_t> seq(L);
        for (auto& b : seq) b = rng() & 3;
        // letters: mostly follow the sequence (so that good diagonals exist), graph shaped like a POA graph: chains with branches, merges, a few wide rows
        for (int r = 0; r < V; r++) {
            d.code[r] = (rng() % 100 < 70) ? seq[std::min(L - 1, (int)((long long)r * L / V))] : (rng() & 3);
            if (r == 0 || rng() % 400 == 0) continue;                                  // a source
            int np = 1; const int x = rng() % 100;
            if (x >= 58) np = 2; if (x >= 85) np = 3; if (x >= 93) np = 4; if (x >= 97) np = 5 + rng() % 6;
            np = std::min(np, r);
            std::vector<int> ps;
            for (int k = 0; k < np; k++) {
                int p;
                for (;;) {
                    const int y = rng() % 100;
                    p = y < 69 ? r - 1 : y < 95 ? r - 2 - (int)(rng() % 6) : (int)(rng() % r);
                    if (p < 0) p = 0;
                    if (std::find(ps.begin(), ps.end(), p) == ps.end()) break;
                    if ((int)ps.size() >= r) break;
                }
                if (std::find(ps.begin(), ps.end(), p) == ps.end()) ps.push_back(p);
            }
            std::shuffle(ps.begin(), ps.end(), rng);
            d.pred[r] = ps;
        }
        if (cs % 7 == 3 && V > 600) {   // long-range in-edges (a read with a long deletion): the far predecessor's B is tens of thousands below the row's
            for (int k = 0; k < 6; k++) {
                const int r = V / 2 + (int)(rng() % (V / 2)), p = (int)(rng() % 40);
                if (std::find(d.pred[r].begin(), d.pred[r].end(), p) == d.pred[r].end()) d.pred[r].insert(d.pred[r].begin() + (rng() % (d.pred[r].size() + 1)), p);
            }
        }
        // ---------------- reference: plain recurrence on (V + 1) x (L + 1), move = (type, slot) with the reference's tie rules
        const int W = L + 1;
        std::vector<int> Hm((size_t)(V + 1) * W);
        std::vector<uint8_t> Ty((size_t)(V + 1) * W), Sl((size_t)(V + 1) * W);
        for (int j = 0; j <= L; j++) Hm[j] = G * j;
        for (int i = 1; i <= V; i++) {
            const auto& ps = d.pred[i - 1];
            for (int j = 0; j <= L; j++) {
                int best = NEGI, ty = 0, sl = 0;
                const int np = std::max<int>(1, (int)ps.size());
                if (j >= 1) {
                    const int s = seq[j - 1] == d.code[i - 1] ? M : X;
                    for (int k = 0; k < np; k++) { const int pr = ps.empty() ? 0 : ps[k] + 1; const int v = Hm[(size_t)pr * W + j - 1] + s; if (v > best) { best = v; ty = 3; sl = k; } }
                }
                for (int k = 0; k < np; k++) { const int pr = ps.empty() ? 0 : ps[k] + 1; const int v = Hm[(size_t)pr * W + j] + G; if (v > best) { best = v; ty = 2; sl = k; } }
                if (j >= 1) { const int v = Hm[(size_t)i * W + j - 1] + G; if (v > best) { best = v; ty = 1; sl = 0; } }
                Hm[(size_t)i * W + j] = best; Ty[(size_t)i * W + j] = (uint8_t)ty; Sl[(size_t)i * W + j] = (uint8_t)sl;
            }
        }
        // ---------------- the packed model: waves of 64 lanes x CM columns, rows in rank order, every wave keeps all its rows (the ring / far rows
        // of the kernel are storage, not arithmetic) with their B
        const int ncol = L + 1, wcols = 64 * CM, NWv = (ncol + wcols - 1) / wcols;
        std::vector<std::vector<u32>> rows((size_t)NWv * (V + 1));      // [w][row] -> 64 * h registers (masked keys)
        std::vector<int> Bq((size_t)NWv * (V + 1));                     // 4 * B per (wave, row)
        const u32 MASKK = 0xfffcfffcu;
        for (int w = 0; w < NWv; w++) {                                 // virtual row 0: Xr = 0 everywhere
            rows[(size_t)w * (V + 1)].assign(64 * h, mk(OFF, OFF));
            Bq[(size_t)w * (V + 1)] = 4 * G * (w * wcols - 1);
        }
        const int c_match = 4 * (M - 2 * G) + 3, c_mis = 4 * (X - 2 * G) + 3, c_vert = 2;   // (each carries -4 G, taken off again with the frame shift)
        for (int i = 1; i <= V; i++) {
            const auto& ps = d.pred[i - 1];
            const int np = std::max<int>(1, (int)ps.size());
            int carry = 0;                                              // 4 * H[i][last column of the wave on the left]
            for (int w = 0; w < NWv; w++) {
                const int c0 = w * wcols;
                // B of this row in this wave
                int Bqi;
                if (w > 0) Bqi = carry;
                else { Bqi = NEGI; for (int k = 0; k < np; k++) { const int pr = ps.empty() ? 0 : ps[k] + 1; Bqi = std::max(Bqi, Bq[pr]); } Bqi += 4 * G; }
                const int fill = w == 0 ? 0 : OFF;                 // "the column left of lane 0": the wave's left edge (Xr = 0), or nothing at all for the first wave
                std::vector<u32> T(64 * h), SL(64 * h, 0);
                for (int k = 0; k < np; k++) {
                    const int pr = ps.empty() ? 0 : ps[k] + 1;
                    const std::vector<u32>& P = rows[(size_t)w * (V + 1) + pr];
                    int dq = Bq[(size_t)w * (V + 1) + pr] - Bqi;       // <= -4 G; clamped below (a predecessor far below this row loses everywhere)
                    if (dq > -4 * G) { printf("case %d: dq %d above -4G\n", cs, dq); return 1; }
                    const int sub = std::min(-4 * G - dq, 65535);       // >= 0
                    const u32 dpk = mk(sub, sub);
                    for (int t = 0; t < 64; t++) {
                        const u32* hp = &P[(size_t)t * h];
                        const u32 nb = t > 0 ? P[(size_t)(t - 1) * h + h - 1] : mk(0, fill);   // wave_shr:1 of the last register (lane 0: the fill in its high half)
                        for (int q = 0; q < h; q++) {
                            const u32 dsrc = q ? hp[q - 1] : alignbit16(hp[h - 1], nb);          // low: column j0 + q - 1 (q = 0: the neighbour's last column), high: column j0 + q + h - 1
                            const int jl = c0 + t * CM + q, jh = jl + h;
                            const int sl_ = (jl >= 1 && jl < ncol && seq[jl - 1] == d.code[i - 1]) ? c_match : c_mis;
                            const int sh_ = (jh >= 1 && jh < ncol && seq[jh - 1] == d.code[i - 1]) ? c_match : c_mis;
                            u32 cand = pk_max(pk_add(dsrc, mk(sl_, sh_)), pk_add(hp[q], mk(c_vert, c_vert)));
                            cand = pk_sub_sat(cand, dpk);
                            if (k == 0) T[(size_t)t * h + q] = cand;
                            else {
                                u32& m = T[(size_t)t * h + q];
                                const u32 gt = pk_min(pk_sub_sat(cand, m), mk(1, 1));
                                SL[(size_t)t * h + q] = pk_max(SL[(size_t)t * h + q], pk_mul(gt, mk(k, k)));
                                m = pk_max(m, cand);
                            }
                        }
                    }
                }
                // in-lane prefix maximum (split layout), wave scan of the chunk ends, the horizontal candidate, the final keys
                std::vector<u32> R(64 * h), K(64 * h);
                std::vector<int> xe(64);
                for (int t = 0; t < 64; t++) {
                    u32* r = &R[(size_t)t * h];
                    const u32* m = &T[(size_t)t * h];
                    r[0] = m[0];
                    for (int q = 1; q < h; q++) r[q] = pk_max(m[q], r[q - 1]);
                    const u32 c = r[h - 1] << 16;                                       // v_lshl_or_b32: the low chain's total into the high halves
                    for (int q = 0; q < h; q++) r[q] = pk_max(r[q], c);
                    xe[t] = (int)(r[h - 1] >> 16) & ~3;
                }
                int run = OFF;                                                                      // exclusive prefix maximum over the lanes, starting from the wave's left edge (Xr = 0)
                for (int t = 0; t < 64; t++) {
                    const int E = run;
                    run = std::max(run, xe[t]);
                    const u32 Epk = mk(E, E);
                    u32* r = &R[(size_t)t * h];
                    for (int q = 0; q < h; q++) {
                        const u32 sh = q ? r[q - 1] : ((r[h - 1] << 16) | (u32)(uint16_t)E);        // scores left of the register's two columns inside the lane
                        const u32 u = pk_max(sh, Epk);
                        const u32 hk = u & MASKK;                                               // horizontal: type 0
                        K[(size_t)t * h + q] = pk_max(T[(size_t)t * h + q], hk);
                    }
                }
                carry = Bqi + (run - OFF) + 4 * G * wcols;
                // store the row (masked) and check every real cell against the reference
                std::vector<u32>& out = rows[(size_t)w * (V + 1) + i];
                out.resize(64 * h);
                Bq[(size_t)w * (V + 1) + i] = Bqi;
                for (int t = 0; t < 64; t++)
                    for (int q = 0; q < h; q++) {
                        const u32 kk = K[(size_t)t * h + q];
                        out[(size_t)t * h + q] = kk & MASKK;
                        for (int half = 0; half < 2; half++) {
                            const int j = c0 + t * CM + q + half * h;
                            if (j >= ncol) continue;
                            const int key = half ? hi16(kk) : lo16(kk);
                            max_key_seen = std::max(max_key_seen, key);
                            const int ty = key & 3, xr = (key - OFF) >> 2;
                            const int Hval = xr + (Bqi >> 2) + G * (j - c0 + 1);
                            const int sl = (half ? (SL[(size_t)t * h + q] >> 16) : (SL[(size_t)t * h + q] & 0xffff));
                            const size_t at = (size_t)i * W + j;
                            cells_checked++;
                            if ((Bqi & 3) || Hval != Hm[at] || (ty ? ty : 1) != Ty[at] || (ty > 1 && sl != Sl[at]) || key < OFF) {
                                printf("case %d (CM %d L %d V %d): row %d col %d wave %d: H %d (ref %d) type %d (ref %d) slot %d (ref %d) key %d npred %d\n", cs, CM, L, V, i, j, w, Hval, Hm[at], ty, Ty[at], sl, Sl[at], key, np);
                                return 1;
                            }
                        }
                    }
            }
        }
        if (g_wrapped) { printf("case %d: a non-saturating packed add wrapped\n", cs); return 1; }
        printf("case %d ok: CM %d, L %d, V %d, %d waves\n", cs, CM, L, V, NWv);
    }
    printf("all %d cases ok, %lld cells, largest key %d\n", n_cases, cells_checked, max_key_seen);
    return 0;
}
