// common.h — shared device helpers for the gfx950 kernels (wave = 64 lanes everywhere).
#ifndef HX_KERNELS_COMMON_H
#define HX_KERNELS_COMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "haslr_types.h"

#define HX_WAVE 64

// error bits reported through hx_ctx's device error word
enum {
    HXE_TRIM_NO_M = 1,        // overlap trim ran off an alignment without any M (undefined in the reference)
    HXE_CHAIN_TOO_MANY = 2,   // > 10000 chainable hits on one read (reference stack arrays, Longread.cpp:529)
    HXE_SPOS_RANGE = 4,       // consensus support starts beyond its read (std::out_of_range in the reference)
    HXE_POA_OVERFLOW = 8,     // POA graph outgrew its workspace (host retries with the worst-case size)
    HXE_BAD_TID = 16,
    HXE_POA_NODIR = 32,       // a POA node gained more than 16 in-edges: the 4-bit predecessor slot of the direction bytes is too small (host retries with the score-matrix traceback)
    HXE_POA_SINKS = 128,      // more sink rows than a one-wavefront launch keeps in LDS (host retries the edge with a multi-wave workgroup, which keeps 1024)
    HXE_POA_WIDEROWS = 256,   // more rows with over 4 predecessors than the wide-row pool has rows for (host retries this edge with more)
    HXE_POA_STALLED = 512,    // a wave of a shared edge gave up waiting for another workgroup's carry (host retries the edge with one workgroup)
    HXE_POA_FARROWS = 64,     // more rows left the LDS ring and were read back than H has rows for (host retries this edge with a row of H for every node)
};

// contig class bits, computed once per run from mean_kmer and the three thresholds of SURVEY.md A.1
enum {
    HXC_DROP_LOAD = 1,   // mean_kmer >  uniq*(3+dev)   Longread.cpp:272
    HXC_UNIQUE = 2,      // mean_kmer <  uniq*(1+dev)   Longread.cpp:191  (palindrome rule)
    HXC_DROP_CHAIN = 4,  // mean_kmer >  uniq*(1+dev)   Longread.cpp:539  (copy_count = 1)
    HXC_EDGE_OK = 8,     // mean_kmer <= uniq*(1+dev)   Backbone_graph.cpp:160
};

struct DevHits {
    uint64_t n;
    const uint32_t *q_id, *q_start, *q_end, *t_id, *t_len, *t_start, *t_end, *n_match, *n_block;
    const uint8_t *is_rev, *mapq;
    const uint64_t* cg_off;
    const uint32_t* cg_ops;
};

// one side of an edge-support record (device SoA, mirrors hx_rec_side)
struct DevSide {
    uint32_t *q_start, *q_end, *t_start, *t_end;
    uint8_t* is_rev;
    uint64_t *cg_begin, *cg_end;
    uint32_t *cg_skip_front, *cg_skip_back;
};

struct CgView {
    const uint32_t* ops;
    uint64_t b, e;
    uint32_t skf, skb;
    __device__ __forceinline__ uint32_t eff(uint64_t k) const {
        uint32_t l = HX_CG_LEN(ops[k]);
        if (k == b) l -= skf;
        if (k + 1 == e) l -= skb;
        return l;
    }
};

// wave-wide inclusive max-scan over 64 lanes with DPP (no LDS round trips): the classic GCN/CDNA sequence
// row_shr:1,2,3 of the original value, then row_shr:4 / row_shr:8 (bank-masked) inside each 16-lane row,
// then row_bcast:15 / row_bcast:31 across rows. Lanes without a source keep `ident` (bound_ctrl off).
__device__ __forceinline__ int wave_scan_max(int v) {
    constexpr int ident = -(1 << 30);
    int x = v;
    x = max(x, __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xf, 0xf, false));   // row_shr:1
    x = max(x, __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xf, 0xf, false));   // row_shr:2
    x = max(x, __builtin_amdgcn_update_dpp(ident, v, 0x113, 0xf, 0xf, false));   // row_shr:3
    x = max(x, __builtin_amdgcn_update_dpp(ident, x, 0x114, 0xf, 0xe, false));   // row_shr:4 bank_mask:0xe
    x = max(x, __builtin_amdgcn_update_dpp(ident, x, 0x118, 0xf, 0xc, false));   // row_shr:8 bank_mask:0xc
    x = max(x, __builtin_amdgcn_update_dpp(ident, x, 0x142, 0xa, 0xf, false));   // row_bcast:15 row_mask:0xa
    x = max(x, __builtin_amdgcn_update_dpp(ident, x, 0x143, 0xc, 0xf, false));   // row_bcast:31 row_mask:0xc
    return x;
}
// value of the previous lane (lane 0 gets `fill`): wave_shr:1
__device__ __forceinline__ int wave_shift_up1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
// workgroup barrier that only waits for LDS traffic (outstanding global stores keep flying)
__device__ __forceinline__ void barrier_lds_only() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
#pragma unroll
    for (int d = 1; d < HX_WAVE; d <<= 1) {
        uint32_t o = __shfl_up(v, d, HX_WAVE);
        if ((int)(threadIdx.x & (HX_WAVE - 1)) >= d) v += o;
    }
    return v;
}

#endif
