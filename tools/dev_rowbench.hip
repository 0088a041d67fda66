// dev_rowbench.hip — the POA row body (4 columns per lane, one predecessor from the LDS ring) as a synthetic loop, with pieces switched off one at
// a time: where do the cycles of a row go? Development tool (prints cycles per row for one wave alone and for 4 / 8 / 16 waves of one workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 8192
constexpr int CM = 4;
__device__ __forceinline__ int wave_incl_max(int v) {
    int x;
    asm volatile(
        "v_mov_b32 %0, %1\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
        : "=&v"(x) : "v"(v));
    return x;
}
__device__ __forceinline__ uint32_t pack_b0(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint32_t ab = __builtin_amdgcn_perm(b, a, 0x0c0c0400u), cd = __builtin_amdgcn_perm(d, c, 0x0c0c0400u);
    return __builtin_amdgcn_perm(cd, ab, 0x05040100u);
}
// skip bits: 1 ring read, 2 scan, 4 publish, 8 ring write, 16 dir store, 32 mismatch mask / scores (constant score), 64 horizontal chain
template <int SKIP>
__global__ void rowk(long long* out, const uint32_t* __restrict__ meta_in, uint8_t* D, int W, int which) {
    extern __shared__ int32_t ring[];
    __shared__ unsigned long long box[16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ring_w = CM * (blockDim.x >> 6) * 65;
    int32_t* ring_me = ring + wv * 65 * CM + lane;
    for (int s = 0; s < 5; s++) for (int k = 0; k < CM; k++) ring_me[s * ring_w + k * 65 + 1] = -(lane * CM + k) * 512;
    if (lane == 0) for (int s = 0; s < 5; s++) ring_me[s * ring_w + (CM - 1) * 65] = -(1 << 30);
    __syncthreads();
    const int g64 = -8 * 64, m64 = 5 * 64, mm64 = -4 * 64, KD = 63, KV = 47, KH = 16;
    const int j0 = tid * CM, jg0 = j0 * g64;
    uint32_t bases = 0x1b1b1b1bu >> (lane & 7), nobase = 0;
    uint32_t mC = meta_in[lane];
    uint32_t nkept = 0;
    uint8_t* drow = D;
    int cinV = -(1 << 30);
    uint32_t dacc[4] = {0, 0, 0, 0};
    long long t0 = clock64();
    for (int i = 1; i <= N; i++) {
        if ((i & 63) == 0) mC = meta_in[(i + lane) & 4095];
        const uint32_t meta = __builtin_amdgcn_readlane(mC, i & 63);
        uint32_t mis;
        if (SKIP & 32) mis = 0; else { const uint32_t x = bases ^ ((meta & 3u) * 0x55555555u); mis = x | (x >> 1) | nobase; }
        drow += W;
        int hp[CM], left;
        if (SKIP & 1) { for (int k = 0; k < CM; k++) hp[k] = jg0 + k * g64 + (int)i; left = jg0 - g64; }
        else {
            const uint32_t loc = (meta >> 12) & 7u;   // ring slot 0..4
            const int32_t* S = ring_me + loc * ring_w;
#pragma unroll
            for (int k = 0; k < CM; k++) hp[k] = S[k * 65 + 1];
            left = S[(CM - 1) * 65];
        }
        int m[CM];
#pragma unroll
        for (int k = 0; k < CM; k++) {
            const int neg = __builtin_amdgcn_sbfe((int)mis, 2 * k, 1);
            const int sc = (m64 + KD) + ((mm64 - m64) & neg);
            m[k] = max((k == 0 ? left : hp[k - 1]) + sc, hp[k] + (g64 + KV));
        }
        if (!(SKIP & 64)) {
#pragma unroll
            for (int k = 1; k < CM; k++) m[k] = max(m[k], (m[k - 1] & ~63) + (g64 + KH));
        }
        int inc = (m[CM - 1] & ~63) - (jg0 + (CM - 1) * g64), ex;
        if (SKIP & 2) ex = inc - 64; else { inc = wave_incl_max(inc); ex = __builtin_amdgcn_update_dpp(-(1 << 30), inc, 0x138, 0xf, 0xf, false); }
        const int cin = __builtin_amdgcn_readlane(cinV, i & 31);
        if (!(SKIP & 4)) { if (lane == 63) __hip_atomic_store(box + wv * 64 + (i & 63), (unsigned long long)(uint32_t)i | ((unsigned long long)(uint32_t)max(cin, inc) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        ex = max(ex, cin);
        const int base = ex + jg0;
#pragma unroll
        for (int k = 0; k < CM; k++) m[k] = max(m[k], base + (k * g64 + KH));
        int t[CM];
#pragma unroll
        for (int k = 0; k < CM; k++) t[k] = m[k] & ~63;
        if (!(SKIP & 8)) {
            const uint32_t kept = (meta >> 4) & 1u, slot = kept ? (nkept & 3u) : 4u;
            int32_t* S = ring_me + slot * ring_w;
#pragma unroll
            for (int k = 0; k < CM; k++) S[k * 65 + 1] = t[k];
            if (lane == 0) S[(CM - 1) * 65] = base - g64;
            nkept += kept;
        }
        if (SKIP & 128) {   // four rows' dwords kept in registers, one 16-byte store every fourth row
            dacc[i & 3] = pack_b0(m[0], m[1], m[2], m[3]) & 0x3f3f3f3fu;
            if ((i & 3) == 3) *reinterpret_cast<uint4*>(drow + (size_t)j0 * 4) = make_uint4(dacc[0], dacc[1], dacc[2], dacc[3]);
        } else if (SKIP & 1024) {   // four rows' dwords kept in registers, four 4-byte stores every fourth row
            dacc[i & 3] = pack_b0(m[0], m[1], m[2], m[3]) & 0x3f3f3f3fu;
            if ((i & 3) == 3) { for (int q = 0; q < 4; q++) *reinterpret_cast<uint32_t*>(drow - (3 - q) * W + j0) = dacc[q]; }
        } else if (SKIP & 256) {   // pack only
            cinV ^= (int)(pack_b0(m[0], m[1], m[2], m[3]) & 0x3f3f3f3fu);
        } else if (SKIP & 512) {   // store only
            *reinterpret_cast<uint32_t*>(drow + j0) = (uint32_t)m[0];
        } else
        if (!(SKIP & 16)) *reinterpret_cast<uint32_t*>(drow + j0) = pack_b0(m[0], m[1], m[2], m[3]) & 0x3f3f3f3fu;
        else if (which == 12345) D[tid] = (uint8_t)(m[0] + m[1] + m[2] + m[3]);
        if (SKIP & 9) cinV += t[0] + t[3];   // keep the values live
    }
    long long t1 = clock64();
    if (tid == 0) out[which] = t1 - t0;
    D[(size_t)W * (N + 2) + tid] = (uint8_t)(cinV + nkept);
}
int main() {
    long long* out; uint32_t* meta; uint8_t* D;
    const int W = 4096;
    hipMalloc(&out, 64 * 8); hipMalloc(&meta, 4096 * 4); hipMalloc(&D, (size_t)W * (N + 4) * 4);
    uint32_t h[4096];
    uint32_t x = 12345;
    for (int i = 0; i < 4096; i++) { x = x * 1664525u + 1013904223u; const uint32_t kept = (x >> 9) % 10 < 6, slot = kept ? (x >> 13) & 3 : 4; h[i] = ((x >> 20) & 3) | (kept << 4) | (slot << 12) | (1u << 8); }
    hipMemcpy(meta, h, sizeof(h), hipMemcpyHostToDevice);
    const char* names[] = {"full row", "x4 store every 4th row", "pack only", "store only", "- ring read", "- scan", "- publish", "- ring write", "- dir store", "- mismatch mask", "- chain", "- ring read/write + store", "- everything but cells", "4 stores every 4th row"};
    for (int nt : {64, 1024}) {
        printf("== %d waves in one workgroup (one per SIMD up to 4)\n", nt / 64);
        const size_t lds = (size_t)5 * CM * (nt / 64) * 65 * 4;
#define RUN(S, idx) do { hipFuncSetAttribute((const void*)rowk<S>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024); rowk<S><<<1, nt, lds>>>(out, meta, D, W, idx); hipDeviceSynchronize(); \
        long long c; hipMemcpy(&c, out + idx, 8, hipMemcpyDeviceToHost); printf("  %-28s %7.1f cycles/row\n", names[idx], (double)c / N); } while (0)
        RUN(0, 0); RUN(128, 1); RUN(1024, 13); RUN(256, 2); RUN(512, 3); RUN(1, 4); RUN(2, 5); RUN(4, 6); RUN(8, 7); RUN(16, 8); RUN(32, 9); RUN(64, 10); RUN(25, 11); RUN(127, 12);
    }
    return 0;
}
