// dev_tpbench.hip — THROUGHPUT of VALU operations per SIMD with several waves resident (dev_pkbench measures one wave's latency: the oldest
// wave wins the issue arbitration, so its clock says nothing about saturation). Whole-chip launch, hipEvent time. Development tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 4096
#define BODY_IND(I0, I1, I2, I3, I4, I5, I6, I7) for (int i = 0; i < N; i++) { asm volatile(I0 "\n\t" I1 "\n\t" I2 "\n\t" I3 "\n\t" I4 "\n\t" I5 "\n\t" I6 "\n\t" I7 \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0)); }
#define IND8(OPC, TAIL) BODY_IND(OPC " %0, %0, %8" TAIL, OPC " %1, %1, %8" TAIL, OPC " %2, %2, %8" TAIL, OPC " %3, %3, %8" TAIL, OPC " %4, %4, %8" TAIL, OPC " %5, %5, %8" TAIL, OPC " %6, %6, %8" TAIL, OPC " %7, %7, %8" TAIL)
#define IND8_3(OPC) BODY_IND(OPC " %0, %0, %8, %9", OPC " %1, %1, %8, %9", OPC " %2, %2, %8, %9", OPC " %3, %3, %8, %9", OPC " %4, %4, %8, %9", OPC " %5, %5, %8, %9", OPC " %6, %6, %8, %9", OPC " %7, %7, %8, %9")
__global__ void __launch_bounds__(1024) k(int* sink, int mode) {
    const int tid = threadIdx.x;
    int a0 = tid + sink[0], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b0 = 0x00030001 + sink[1], c0 = 0x00010001 + sink[2];
    switch (mode) {
    case 0: IND8("v_and_b32", ""); break;
    case 1: IND8("v_or_b32", ""); break;
    case 2: IND8("v_xor_b32", ""); break;
    case 3: IND8("v_add_u32", ""); break;
    case 4: IND8("v_sub_u32", ""); break;
    case 5: IND8("v_lshlrev_b32", ""); break;
    case 6: IND8("v_lshrrev_b32", ""); break;
    case 7: IND8("v_ashrrev_i32", ""); break;
    case 8: IND8("v_max_i32", ""); break;
    case 9: IND8("v_max_u32", ""); break;
    case 10: IND8("v_min_i32", ""); break;
    case 11: IND8("v_min_u32", ""); break;
    case 12: IND8("v_max_f32", ""); break;
    case 13: IND8("v_min_f32", ""); break;
    case 14: IND8("v_add_f32", ""); break;
    case 15: IND8("v_mul_u32_u24", ""); break;
    case 16: IND8("v_max_i16", ""); break;
    case 17: IND8("v_max_u16", ""); break;
    case 18: IND8("v_add_u16", ""); break;
    case 19: IND8("v_pk_max_u16", ""); break;
    case 20: IND8("v_pk_max_i16", ""); break;
    case 21: IND8("v_pk_add_u16", ""); break;
    case 22: IND8("v_pk_max_f16", ""); break;
    case 23: IND8("v_pk_add_f16", ""); break;
    case 24: IND8_3("v_and_or_b32"); break;
    case 25: IND8_3("v_or3_b32"); break;
    case 26: IND8_3("v_add3_u32"); break;
    case 27: IND8_3("v_lshl_add_u32"); break;
    case 28: IND8_3("v_lshl_or_b32"); break;
    case 29: IND8_3("v_bfe_i32"); break;
    case 30: IND8_3("v_bfe_u32"); break;
    case 31: IND8_3("v_bfi_b32"); break;
    case 32: IND8_3("v_perm_b32"); break;
    case 33: IND8_3("v_max3_i32"); break;
    case 34: IND8_3("v_max3_f32"); break;
    case 35: IND8_3("v_med3_i32"); break;
    case 36: IND8_3("v_alignbit_b32"); break;
    case 37: IND8_3("v_mad_u32_u24"); break;
    case 38: IND8_3("v_fma_f32"); break;
    case 39: IND8_3("v_xad_u32"); break;
    case 40: IND8_3("v_add_lshl_u32"); break;
    case 41: IND8("v_cndmask_b32", ""); break;
    case 42: BODY_IND("v_mov_b32 %0, %8", "v_mov_b32 %1, %8", "v_mov_b32 %2, %8", "v_mov_b32 %3, %8", "v_mov_b32 %4, %8", "v_mov_b32 %5, %8", "v_mov_b32 %6, %8", "v_mov_b32 %7, %8"); break;
    }
    sink[8 + (blockIdx.x * blockDim.x + tid) % 4096] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    int* sink; (void)hipMalloc(&sink, 1 << 20); (void)hipMemset(sink, 0, 1 << 20);
    const char* names[] = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_max_i32", "v_max_u32", "v_min_i32", "v_min_u32", "v_max_f32", "v_min_f32", "v_add_f32", "v_mul_u32_u24", "v_max_i16", "v_max_u16", "v_add_u16", "v_pk_max_u16", "v_pk_max_i16", "v_pk_add_u16", "v_pk_max_f16", "v_pk_add_f16", "v_and_or_b32", "v_or3_b32", "v_add3_u32", "v_lshl_add_u32", "v_lshl_or_b32", "v_bfe_i32", "v_bfe_u32", "v_bfi_b32", "v_perm_b32", "v_max3_i32", "v_max3_f32", "v_med3_i32", "v_alignbit_b32", "v_mad_u32_u24", "v_fma_f32", "v_xad_u32", "v_add_lshl_u32", "v_cndmask_b32", "v_mov_b32"};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps : {1, 8}) {   // waves per SIMD
        const int nt = wps >= 4 ? 1024 : wps * 256, blocks = 256 * (wps == 8 ? 2 : 1);
        fflush(stdout); printf("== %d waves per SIMD (%d blocks x %d): SIMD cycles per wave-instruction at 2.4 GHz (8 per iteration; mode 16: 16)\n", wps, blocks, nt);
        for (int m = 0; m < 43; m++) {
            k<<<blocks, nt>>>(sink, m); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); k<<<blocks, nt>>>(sink, m); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double cyc = ms * 1e-3 * 2.4e9, per = cyc / ((double)N * 8 * wps);
            printf("  %-26s %7.3f ms  %6.2f cycles per instruction and SIMD\n", names[m], ms, per); fflush(stdout);
        }
    }
    return 0;
}
