// dev_ubench.hip — latency micro-benchmarks for the POA row loop building blocks (single wave unless stated). Development tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 4096
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
__global__ void k(long long* out, int* sink, int mode) {
    __shared__ int lds[4096];
    const int tid = threadIdx.x;
    int v = tid * 3 + sink[0], w = tid + 7, acc = 0;
    lds[tid] = tid; lds[tid + 1024] = 1;
    __syncthreads();
    long long t0 = clock64();
    if (mode == 0) { for (int i = 0; i < N; i++) { v = max(v + 3, w); asm volatile("" : "+v"(v)); } }                      // 2 dependent VALU
    else if (mode == 1) { for (int i = 0; i < N; i++) { v = wave_incl_max(v) + 1; } }                                          // DPP scan + 1
    else if (mode == 2) { for (int i = 0; i < N; i++) { v = lds[(v & 1023)] + 1; } }                                           // LDS round trip (dependent)
    else if (mode == 3) { for (int i = 0; i < N; i++) { __builtin_amdgcn_s_barrier(); v++; } }                                // barrier only
    else if (mode == 4) { for (int i = 0; i < N; i++) { int s = __builtin_amdgcn_readlane(v, i & 63); if (s & 1) v += 3; else v ^= 5; } }   // readlane -> scalar branch
    else if (mode == 5) { for (int i = 0; i < N; i++) { if (tid == 63) lds[2048 + (i & 1)] = v; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); v = max(v, lds[2048 + (i & 1)]) + 1; } }   // exchange
    else if (mode == 6) { for (int i = 0; i < N; i++) { v = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false) + 1; } }  // wave_shr + 1
    else if (mode == 7) { int a = v, b = w, c = v ^ w, d = v + w; for (int i = 0; i < N; i++) { a = max(a + 3, w); b = max(b + 5, w); c = max(c + 7, w); d = max(d + 9, w); asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); } v = a + b + c + d; }   // 4 independent chains of 2
    else if (mode == 8) { for (int i = 0; i < N; i++) { v = (v & ~63) + 513; v = max(v, w); asm volatile("" : "+v"(v)); } }    // and, add, max chain
    else if (mode == 9) { for (int i = 0; i < N; i++) { sink[1024 + tid] = v; v++; } }                                          // global store per iter
    else if (mode == 10) { for (int i = 0; i < N; i++) { v += __builtin_amdgcn_readlane(v, 5); } }                            // readlane -> VALU use
    else if (mode == 11) { for (int i = 0; i < N; i++) { int s = __builtin_amdgcn_readfirstlane(v); s = (s >> 3) & 7; if (s == 1) acc++; else if (s == 2) acc += 2; else if (s == 3) acc ^= 1; v += s; } }
    long long t1 = clock64();
    if (tid == 0) out[mode] = t1 - t0;
    sink[tid + 1] = v + acc;
}
int main() {
    long long* out; int* sink;
    hipMalloc(&out, 64 * 8); hipMalloc(&sink, 1 << 20); hipMemset(sink, 0, 1 << 20);
    const char* names[] = {"2 dependent VALU (add,max)", "DPP 64-lane max scan + add", "dependent LDS load + add", "s_barrier (NT waves)", "readlane->scalar branch", "LDS exchange + barrier", "wave_shr dpp + add", "4 independent chains x2 ops", "and,add,max chain (3 dep ops)", "global store + add", "readlane -> VALU add", "readfirstlane -> 3 scalar branches"};
    for (int nt : {64, 256, 1024}) {
        printf("== block %d threads\n", nt);
        for (int m = 0; m < 12; m++) {
            k<<<1, nt>>>(out, sink, m); hipDeviceSynchronize();
            long long h; hipMemcpy(&h, out + m, 8, hipMemcpyDeviceToHost);
            printf("  %-36s %8.1f cycles/iter\n", names[m], (double)h / N);
        }
    }
    return 0;
}
