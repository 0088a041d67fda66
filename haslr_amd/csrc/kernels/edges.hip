// edges.hip — K4: edge-support records of the backbone graph.
//
// Replaces bbg_build_graph / bbg_add_edge (Backbone_graph.cpp:148-171, :10-25): for every compact long read,
// consecutive anchors on unique contigs emit one forward and one twin record. The reference appends them to
// per-edge vectors inside std::map nodes; here they are written once to a flat SoA multiset, sorted by the
// 64-bit edge key with a stable radix sort (emission order = read asc, pair asc, forward before twin, which
// is exactly the reference's push order), and segmented into edges. Records carry a copy of both anchor
// alignments so that they stay meaningful after the multi-GPU all-gather (the alignment table is per rank).
#include "kernels.h"

namespace hxk {

namespace {

__global__ void k_edge_count(DevHits h, const uint8_t* __restrict__ cls, ChainFinal c, const uint64_t* __restrict__ cmp_off,
                             uint32_t lr_begin, uint32_t lr_end, uint32_t* n_pairs) {
    uint32_t r = lr_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= lr_end) return;
    uint64_t b = cmp_off[r - lr_begin], e = cmp_off[r - lr_begin + 1];
    uint32_t nsel = 0;
    if (e - b > 1)
        for (uint64_t j = b; j < e; j++) nsel += (cls[h.t_id[c.hit[c.cmp_aln[j]]]] & HXC_EDGE_OK) ? 1 : 0;
    n_pairs[r - lr_begin] = nsel > 1 ? nsel - 1 : 0;
}

__device__ __forceinline__ void put_side(const DevSide& s, uint64_t i, const ChainFinal& c, const DevHits& h, uint32_t a) {
    s.q_start[i] = c.qs[a]; s.q_end[i] = c.qe[a]; s.t_start[i] = c.ts[a]; s.t_end[i] = c.te[a];
    s.is_rev[i] = h.is_rev[c.hit[a]]; s.cg_begin[i] = c.cb[a]; s.cg_end[i] = c.ce[a];
    s.cg_skip_front[i] = c.skf[a]; s.cg_skip_back[i] = c.skb[a];
}

__global__ void k_edge_emit(DevHits h, const uint8_t* __restrict__ cls, ChainFinal c, const uint64_t* __restrict__ cmp_off,
                            uint32_t lr_begin, uint32_t lr_end, const uint64_t* __restrict__ pair_off, EdgeRecs out) {
    uint32_t r = lr_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= lr_end) return;
    uint64_t b = cmp_off[r - lr_begin], e = cmp_off[r - lr_begin + 1];
    if (e - b <= 1) return;
    uint64_t o = 2 * pair_off[r - lr_begin];
    int64_t prev = -1;
    for (uint64_t j = b; j < e; j++) {
        uint32_t a2 = c.cmp_aln[j];
        if (!(cls[h.t_id[c.hit[a2]]] & HXC_EDGE_OK)) continue;
        if (prev >= 0) {
            uint32_t a1 = c.cmp_aln[prev];
            uint32_t i1 = (uint32_t)(prev - b), i2 = (uint32_t)(j - b);
            uint32_t n1 = h.t_id[c.hit[a1]], r1 = h.is_rev[c.hit[a1]];
            uint32_t n2 = h.t_id[c.hit[a2]], r2 = h.is_rev[c.hit[a2]];
            out.key[o] = ((uint64_t)((n1 << 1) | r1) << 32) | ((n2 << 1) | r2);
            out.lr[o] = r; out.cmp_head[o] = i1; out.cmp_tail[o] = i2;
            put_side(out.head, o, c, h, a1); put_side(out.tail, o, c, h, a2);
            o++;
            out.key[o] = ((uint64_t)((n2 << 1) | (1 - r2)) << 32) | ((n1 << 1) | (1 - r1));
            out.lr[o] = r | 0x80000000u; out.cmp_head[o] = i2; out.cmp_tail[o] = i1;
            put_side(out.head, o, c, h, a2); put_side(out.tail, o, c, h, a1);
            o++;
        }
        prev = (int64_t)j;
    }
}

__device__ __forceinline__ void copy_side(const DevSide& d, uint64_t i, const DevSide& s, uint64_t j) {
    d.q_start[i] = s.q_start[j]; d.q_end[i] = s.q_end[j]; d.t_start[i] = s.t_start[j]; d.t_end[i] = s.t_end[j];
    d.is_rev[i] = s.is_rev[j]; d.cg_begin[i] = s.cg_begin[j]; d.cg_end[i] = s.cg_end[j];
    d.cg_skip_front[i] = s.cg_skip_front[j]; d.cg_skip_back[i] = s.cg_skip_back[j];
}

__global__ void k_edge_gather(EdgeRecs in, const uint32_t* __restrict__ perm, uint64_t n, EdgeRecs out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t j = perm[i];
    // out.key is already sorted by the radix sort; the rest follows the permutation
    out.lr[i] = in.lr[j]; out.cmp_head[i] = in.cmp_head[j]; out.cmp_tail[i] = in.cmp_tail[j];
    copy_side(out.head, i, in.head, j);
    copy_side(out.tail, i, in.tail, j);
}

__global__ void k_iota(uint32_t* p, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (uint32_t)i;
}

__global__ void k_segment_flags(const uint64_t* __restrict__ key, uint64_t n, uint32_t* flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || key[i] != key[i - 1]) ? 1u : 0u;
}

__global__ void k_segment_scatter(const uint64_t* __restrict__ key, const uint64_t* __restrict__ fs, uint64_t n, uint64_t* edge_key, uint64_t* edge_off) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (fs[i + 1] != fs[i]) { edge_key[fs[i]] = key[i]; edge_off[fs[i]] = i; }
    if (i == n - 1) edge_off[fs[n]] = n;
}

// packed exchange layout (multi-GPU all-gather). Records are emitted in pairs (forward, twin) that hold the same two anchor alignments
// with head and tail swapped, so a pair travels as ONE unit of 22 dwords = 44 bytes per record (the 112-byte record of round 1 carried
// both copies, 64-bit CIGAR ends and a byte per dword):
//   0-1  forward key (low = vertex entered at the tail anchor, high = vertex left at the head anchor)
//   2    long-read id        3  compact indices, head | tail << 16 (a read has at most 10 000 chainable hits, Longread.cpp:529)
//   4-12 head anchor, 13-21 tail anchor: q_start, q_end, t_start, t_end, cg_begin (64 bit), cg_end - cg_begin, skip_front, skip_back
// is_rev of an anchor is the low bit of its key half. The twin's fields follow from the forward record's (Backbone_graph.cpp:10-25).
__device__ __forceinline__ void pack_side(uint32_t* w, const DevSide& s, uint64_t i) {
    w[0] = s.q_start[i]; w[1] = s.q_end[i]; w[2] = s.t_start[i]; w[3] = s.t_end[i];
    w[4] = (uint32_t)s.cg_begin[i]; w[5] = (uint32_t)(s.cg_begin[i] >> 32); w[6] = (uint32_t)(s.cg_end[i] - s.cg_begin[i]);
    w[7] = s.cg_skip_front[i]; w[8] = s.cg_skip_back[i];
}
__device__ __forceinline__ void unpack_side(const uint32_t* w, uint32_t is_rev, const DevSide& s, uint64_t i) {
    s.q_start[i] = w[0]; s.q_end[i] = w[1]; s.t_start[i] = w[2]; s.t_end[i] = w[3]; s.is_rev[i] = (uint8_t)is_rev;
    const uint64_t cb = w[4] | ((uint64_t)w[5] << 32);
    s.cg_begin[i] = cb; s.cg_end[i] = cb + w[6];
    s.cg_skip_front[i] = w[7]; s.cg_skip_back[i] = w[8];
}
__global__ void k_edge_pack(EdgeRecs r, uint64_t n_pairs, uint32_t* dst) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const uint64_t i = 2 * p;   // the forward record of the pair
    uint32_t* w = dst + p * (2 * EDGE_REC_WORDS);
    w[0] = (uint32_t)r.key[i]; w[1] = (uint32_t)(r.key[i] >> 32); w[2] = r.lr[i]; w[3] = r.cmp_head[i] | (r.cmp_tail[i] << 16);
    pack_side(w + 4, r.head, i); pack_side(w + 13, r.tail, i);
}
__global__ void k_edge_unpack(const uint32_t* src, uint64_t n_pairs, EdgeRecs r) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const uint32_t* w = src + p * (2 * EDGE_REC_WORDS);
    const uint32_t v2 = w[0], v1 = w[1], ih = w[3] & 0xffffu, it = w[3] >> 16;   // v1 = n1<<1|rev1 (head), v2 = n2<<1|rev2 (tail)
    uint64_t i = 2 * p;
    r.key[i] = ((uint64_t)v1 << 32) | v2; r.lr[i] = w[2]; r.cmp_head[i] = ih; r.cmp_tail[i] = it;
    unpack_side(w + 4, v1 & 1u, r.head, i); unpack_side(w + 13, v2 & 1u, r.tail, i);
    i++;                                                                          // the twin: leaves n2 by its other end, enters n1 reversed
    r.key[i] = ((uint64_t)(v2 ^ 1u) << 32) | (v1 ^ 1u); r.lr[i] = w[2] | 0x80000000u; r.cmp_head[i] = it; r.cmp_tail[i] = ih;
    unpack_side(w + 13, v2 & 1u, r.head, i); unpack_side(w + 4, v1 & 1u, r.tail, i);
}

inline unsigned grid_for(uint64_t n, int t) { return (unsigned)((n + t - 1) / t); }

}  // namespace

void edge_count(const DevHits& h, const uint8_t* cls, const ChainFinal& c, const uint64_t* cmp_off, uint32_t lr_begin, uint32_t lr_end, uint32_t* n_pairs, hipStream_t s) {
    uint32_t n = lr_end - lr_begin;
    if (n) k_edge_count<<<grid_for(n, 64), 64, 0, s>>>(h, cls, c, cmp_off, lr_begin, lr_end, n_pairs);
}
void edge_emit(const DevHits& h, const uint8_t* cls, const ChainFinal& c, const uint64_t* cmp_off, uint32_t lr_begin, uint32_t lr_end, const uint64_t* pair_off, const EdgeRecs& out, hipStream_t s) {
    uint32_t n = lr_end - lr_begin;
    if (n) k_edge_emit<<<grid_for(n, 64), 64, 0, s>>>(h, cls, c, cmp_off, lr_begin, lr_end, pair_off, out);
}
void edge_gather(const EdgeRecs& in, const uint32_t* perm, uint64_t n, const EdgeRecs& out, hipStream_t s) {
    if (n) k_edge_gather<<<grid_for(n, 256), 256, 0, s>>>(in, perm, n, out);
}
void edge_pack(const EdgeRecs& r, uint64_t n, uint32_t* dst, hipStream_t s) { if (n) k_edge_pack<<<grid_for(n / 2, 256), 256, 0, s>>>(r, n / 2, dst); }
void edge_unpack(const uint32_t* src, uint64_t n, const EdgeRecs& r, hipStream_t s) { if (n) k_edge_unpack<<<grid_for(n / 2, 256), 256, 0, s>>>(src, n / 2, r); }
void iota_u32(uint32_t* p, uint64_t n, hipStream_t s) { if (n) k_iota<<<grid_for(n, 256), 256, 0, s>>>(p, n); }
void segment_flags(const uint64_t* key, uint64_t n, uint32_t* flag, hipStream_t s) { if (n) k_segment_flags<<<grid_for(n, 256), 256, 0, s>>>(key, n, flag); }
void segment_scatter(const uint64_t* key, const uint64_t* fs, uint64_t n, uint64_t* edge_key, uint64_t* edge_off, hipStream_t s) {
    if (n) k_segment_scatter<<<grid_for(n, 256), 256, 0, s>>>(key, fs, n, edge_key, edge_off);
}

}  // namespace hxk
