// dev_lonebench.hip — what one wavefront alone on its SIMD pays per instruction KIND (the regime of the POA row loop of a chain-bound call):
// straight-line bodies of 64 "slots" built with .rept, looped 2000 times, timed with s_memtime. Development tool (round 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2000
// every body: 16 x { 4 x v_add (independent registers) + EXTRA }
#define BODY(EXTRA) asm volatile(".rept 16\n\tv_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t" EXTRA "\n\t.endr" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "scc", "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "memory")
__global__ void k(long long* out, int* sink, int mode) {
    __shared__ unsigned long long lds[256];
    const int tid = threadIdx.x;
    int a0 = tid + sink[0], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b0 = 3 + sink[1];
    lds[tid & 255] = tid;
    __syncthreads();
    const uint32_t la = (uint32_t)(uintptr_t)(&lds[tid & 63]);
    asm volatile("s_cmp_eq_u32 0, 1\n\ts_mov_b32 s44, 0\n\ts_mov_b32 s45, 0\n\ts_mov_b32 s46, 0\n\ts_mov_b32 s47, 0x00020000\n\ts_mov_b32 s42, 3\n\ts_mov_b32 s43, 7" ::: "scc", "s42", "s43", "s44", "s45", "s46", "s47");   // scc = 0; a null buffer resource
    long long t0 = clock64();
    for (int i = 0; i < ITER; i++) {
        switch (mode) {
        case 0: BODY(""); break;                                                     // 64 VALU
        case 1: BODY("s_add_u32 s40, s40, 1"); break;                                 // + 16 SALU
        case 2: BODY("s_cmp_eq_u32 0, 1\n\ts_cbranch_scc1 1f\n\t1:"); break;               // + 16 x (s_cmp + branch NOT taken)
        case 3: BODY("s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\t1:"); break;    // + 16 x (s_cmp + branch TAKEN over one s_nop)
        case 4: BODY("s_and_saveexec_b64 s[40:41], vcc\n\ts_or_b64 exec, exec, s[40:41]"); break;   // + 16 x exec save / restore (vcc: whatever)
        case 5: BODY("s_waitcnt lgkmcnt(0)"); break;                                  // + 16 x waitcnt with nothing outstanding
        case 6: BODY("s_nop 1"); break;
        case 7: BODY("v_readlane_b32 s40, %0, 5\n\ts_lshr_b32 s41, s40, 3"); break;   // + 16 x (readlane -> dependent SALU)
        case 8: BODY("v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf"); break;
        case 9: BODY("v_cmp_eq_u32 vcc, 0, %0\n\ts_cbranch_vccz 1f\n\t1:"); break;         // + 16 x (v_cmp + vcc branch, mostly taken: vcc = 0 ... either way)
        case 10: asm volatile(".rept 16\n\tv_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\tds_write_b64 %5, %[d]\n\t.endr"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(la), [d] "v"((unsigned long long)a0) : "memory"); break;   // + 16 x ds_write_b64 (no wait)
        case 11: BODY("v_max_i32_dpp %1, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf"); break;
        case 12: BODY("v_add3_u32 %1, %0, %4, %1"); break;                            // + 16 x VOP3 (8-byte encoding)
        case 13: BODY("s_andn2_b64 vcc, exec, s[42:43]\n\ts_cbranch_vccnz 1f\n\t1:"); break;   // + 16 x (s_andn2 -> vcc branch): the compiler's flag test
        case 14: BODY("s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 1f\n\t.rept 64\n\ts_nop 0\n\t.endr\n\t1:"); break;   // TAKEN over 64 instructions (256 bytes: new fetch lines)
        case 15: BODY("v_readlane_b32 s40, %0, 5"); break;                            // + 16 x readlane alone
        case 16: BODY("s_and_saveexec_b64 s[40:41], exec\n\ts_cbranch_execz 1f\n\t1:\n\ts_or_b64 exec, exec, s[40:41]"); break;   // + 16 x (saveexec, execz branch not taken, restore)
        case 17: BODY("v_readlane_b32 s40, %0, 5\n\tv_add_u32 %1, s40, %1"); break;     // + 16 x (readlane -> dependent VALU)
        case 18: BODY("s_bitcmp1_b32 s42, 5\n\ts_cselect_b32 s40, 63, 15\n\ts_cselect_b32 s41, 47, 11"); break;   // + 16 x 3 SALU
        case 19: BODY("s_mov_b32 m0, s42\n\ts_nop 0\n\tv_writelane_b32 %1, s43, m0"); break;   // + 16 x (m0, writelane)
        case 20: BODY("buffer_store_short %0, %4, s[44:47], 0 offen"); break;          // + 16 x buffer store whose offsets fail the range check (num_records 0)
        case 22: asm volatile(".rept 8\n\tv_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\t.endr"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0)); break;   // 64 v_add, eight independent chains
        case 23: asm volatile(".rept 64\n\tv_add_u32 %0, %0, %1\n\t.endr" : "+v"(a0) : "v"(b0)); break;                     // 64 v_add, ONE dependent chain
        case 24: asm volatile("s_mov_b32 s40, 16\n\t1:\n\tv_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\ts_sub_u32 s40, s40, 1\n\ts_cmp_lg_u32 s40, 0\n\ts_cbranch_scc1 1b"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "scc", "s40"); break;                           // the same 64 v_add as a loop of 4 (16 taken branches)
        case 25: asm volatile(".rept 32\n\tv_add_u32 %0, %0, %4\n\tv_max_i32 %1, %1, %0\n\t.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0)); break;   // 64 VALU: pairs, the second depends on the first
        case 21: BODY("s_cmp_eq_u32 0, 1\n\ts_cbranch_scc1 1f\n\t1:\n\ts_cmp_eq_u32 0, 1\n\ts_cbranch_scc1 2f\n\t2:"); break;   // + 16 x two not-taken scc branches back to back
        }
    }
    long long t1 = clock64();
    if (tid == 0) out[mode] = t1 - t0;
    sink[tid + 8] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    long long* out; int* sink;
    hipMalloc(&out, 64 * 8); hipMalloc(&sink, 1 << 20); hipMemset(sink, 0, 1 << 20);
    const char* names[] = {"64 v_add", "+16 s_add", "+16 (s_cmp, branch not taken)", "+16 (s_cmp, branch taken, s_nop)", "+16 (saveexec, s_or exec)", "+16 s_waitcnt lgkmcnt(0)", "+16 s_nop 1",
                           "+16 (readlane, dependent s_lshr)", "+16 v_mov_dpp wave_shr:1", "+16 (v_cmp, vcc branch)", "+16 ds_write_b64", "+16 v_max_dpp row_shr:1", "+16 v_add3 (8-byte)",
                           "+16 (s_andn2 vcc, vccnz branch)", "+16 (s_cmp, branch taken over 256 B)", "+16 readlane",
                           "+16 (saveexec, execz branch, s_or)", "+16 (readlane, dependent v_add)", "+16 (s_bitcmp, 2 s_cselect)", "+16 (s_mov m0, s_nop, v_writelane)", "+16 buffer_store (out of range)", "+16 x 2 branches not taken",
                           "64 v_add, 8 independent chains", "64 v_add, one dependent chain", "64 v_add as a loop of 4 x 16", "32 x (v_add, dependent v_max)"};
    for (int rep = 0; rep < 2; rep++) {
        long long base = 0;
        for (int m = 0; m < 26; m++) {
            k<<<1, 64>>>(out, sink, m); hipDeviceSynchronize();
            long long h; hipMemcpy(&h, out + m, 8, hipMemcpyDeviceToHost);
            const double per = (double)h / ITER;
            if (m == 0) base = h;
            printf("  %-40s %8.1f cycles per body, extra per item %6.1f\n", names[m], per, (double)(h - base) / ITER / 16);
        }
        printf("\n");
    }
    return 0;
}
