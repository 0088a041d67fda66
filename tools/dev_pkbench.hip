// dev_pkbench.hip — issue rate and latency of the packed 16-bit VALU operations the int16 POA row body is built from, against their
// 32-bit counterparts (one wave alone, and 16 waves of one workgroup = 4 per SIMD). Development tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 2048
// OP(a, b): a = f(a, b). DEP: one chain of 8 per iteration; IND: 8 independent chains
#define BODY_DEP(INS) for (int i = 0; i < N; i++) { asm volatile(INS "\n\t" INS "\n\t" INS "\n\t" INS "\n\t" INS "\n\t" INS "\n\t" INS "\n\t" INS : "+v"(a0) : "v"(b0), "v"(c0)); }
#define BODY_IND(I0, I1, I2, I3, I4, I5, I6, I7) for (int i = 0; i < N; i++) { asm volatile(I0 "\n\t" I1 "\n\t" I2 "\n\t" I3 "\n\t" I4 "\n\t" I5 "\n\t" I6 "\n\t" I7 \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0)); }
#define IND8(OPC, TAIL) BODY_IND(OPC " %0, %0, %8" TAIL, OPC " %1, %1, %8" TAIL, OPC " %2, %2, %8" TAIL, OPC " %3, %3, %8" TAIL, OPC " %4, %4, %8" TAIL, OPC " %5, %5, %8" TAIL, OPC " %6, %6, %8" TAIL, OPC " %7, %7, %8" TAIL)
#define IND8_3(OPC) BODY_IND(OPC " %0, %0, %8, %9", OPC " %1, %1, %8, %9", OPC " %2, %2, %8, %9", OPC " %3, %3, %8, %9", OPC " %4, %4, %8, %9", OPC " %5, %5, %8, %9", OPC " %6, %6, %8, %9", OPC " %7, %7, %8, %9")
__global__ void k(long long* out, int* sink, int mode) {
    __shared__ uint4 lds[1024];
    const int tid = threadIdx.x;
    int a0 = tid + sink[0], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b0 = 0x00030001 + sink[1], c0 = 0x00010001 + sink[2];
    lds[tid & 1023] = make_uint4(tid, 1, 2, 3);
    __syncthreads();
    long long t0 = clock64();
    switch (mode) {
    case 0: IND8("v_add_u32", ""); break;
    case 1: IND8("v_max_i32", ""); break;
    case 2: IND8("v_pk_add_i16", ""); break;
    case 3: IND8("v_pk_add_i16", " clamp"); break;
    case 4: IND8("v_pk_max_i16", ""); break;
    case 5: IND8("v_pk_sub_u16", " clamp"); break;
    case 6: IND8_3("v_pk_mad_u16"); break;
    case 7: IND8_3("v_perm_b32"); break;
    case 8: IND8_3("v_and_or_b32"); break;
    case 9: IND8_3("v_alignbit_b32"); break;
    case 10: IND8_3("v_lshl_or_b32"); break;
    case 11: IND8("v_pk_min_u16", ""); break;
    case 12: BODY_DEP("v_add_u32 %0, %0, %1"); break;
    case 13: BODY_DEP("v_pk_add_i16 %0, %0, %1"); break;
    case 14: BODY_DEP("v_pk_max_i16 %0, %0, %1"); break;
    case 15: BODY_DEP("v_perm_b32 %0, %0, %1, %2"); break;
    case 16: BODY_DEP("v_pk_mad_u16 %0, %0, %1, %2"); break;
    case 17: for (int i = 0; i < N; i++) { uint4 v = lds[(a0 & 1023)]; a0 = (int)(v.x + v.y + v.z + v.w); } break;   // dependent ds_read_b128 (+3 adds)
    case 18: for (int i = 0; i < N; i++) { int s = __builtin_amdgcn_readlane(a0, 7); s = (s << 2) & 0xffff; s |= s << 16; a0 = a0 + s; asm volatile("" : "+v"(a0)); } break;   // readlane -> 3 SALU -> VALU
    case 19: IND8("v_pk_ashrrev_i16", ""); break;
    case 20: IND8("v_pk_lshlrev_b16", ""); break;
    case 21: IND8_3("v_max3_i32"); break;
    case 22: IND8_3("v_add3_u32"); break;
    case 23: IND8_3("v_bfe_i32"); break;
    }
    long long t1 = clock64();
    if (tid == 0) out[mode] = t1 - t0;
    sink[tid + 8] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    long long* out; int* sink;
    hipMalloc(&out, 64 * 8); hipMalloc(&sink, 1 << 20); hipMemset(sink, 0, 1 << 20);
    const char* names[] = {"v_add_u32 x8 ind", "v_max_i32 x8 ind", "v_pk_add_i16 x8 ind", "v_pk_add_i16 clamp x8 ind", "v_pk_max_i16 x8 ind", "v_pk_sub_u16 clamp x8 ind", "v_pk_mad_u16 x8 ind", "v_perm_b32 x8 ind",
                           "v_and_or_b32 x8 ind", "v_alignbit_b32 x8 ind", "v_lshl_or_b32 x8 ind", "v_pk_min_u16 x8 ind", "v_add_u32 x8 dep", "v_pk_add_i16 x8 dep", "v_pk_max_i16 x8 dep", "v_perm_b32 x8 dep", "v_pk_mad_u16 x8 dep",
                           "ds_read_b128 dep + 3 add", "readlane -> 3 salu -> valu", "v_pk_ashrrev_i16 x8 ind", "v_pk_lshlrev_b16 x8 ind", "v_max3_i32 x8 ind", "v_add3_u32 x8 ind", "v_bfe_i32 x8 ind"};
    for (int nt : {64, 1024}) {
        printf("== block %d threads: cycles per group of 8 instructions (mode 17/18: per iteration)\n", nt);
        for (int m = 0; m < 24; m++) {
            k<<<1, nt>>>(out, sink, m); hipDeviceSynchronize();
            long long h; hipMemcpy(&h, out + m, 8, hipMemcpyDeviceToHost);
            printf("  %-30s %8.1f\n", names[m], (double)h / N);
        }
    }
    return 0;
}
