// r05_valu_ubench.hip (round 5: the forms the cost scan and its row walk are made of)
// valu_ubench.hip -- issue cost of the VALU instructions the prefilter is made of (gfx950), cycles per wave64 instruction
// and SIMD at 4 waves per SIMD: 8 independent chains per wave, so dependency latency is hidden and the figure is the
// pipe's throughput.   hipcc --offload-arch=gfx950 -O3 valu_ubench.hip -o valu_ubench && ./valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define ITERS 4096

#define KERNEL(NAME, ASM)                                                                          \
__global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned seed) {                        \
    unsigned r0 = threadIdx.x + seed, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13,     \
             r6 = r0 * 17, r7 = r0 * 19, a = r0 ^ 0x5555, b = r0 | 0x33, sh = 3;                  \
    __shared__ unsigned lds[2048];                                                                  \
    lds[threadIdx.x] = r0; lds[threadIdx.x + 256] = r1; __syncthreads();                           \
    asm volatile("s_mov_b64 s[4:5], 0x5555\ns_mov_b32 s8, 0x0f0f" ::: "s4", "s5", "s8");                    \
    for (int i = 0; i < ITERS; ++i) {                                                               \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                        \
                     ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                        \
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                     : "v"(a), "v"(b), "v"(sh) : "vcc", "s4", "s5", "s6", "s7", "s8");                                            \
    }                                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;             \
}

#define A_0(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_1(i) "v_add_u32 %" #i ", %" #i ", %" #i "\n"
#define A_2(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define A_3(i) "v_lshrrev_b32 %" #i ", 31, %" #i "\n"
#define A_4(i) "v_ashrrev_i32 %" #i ", 31, %" #i "\n"
#define A_5(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define A_6(i) "v_bfi_b32 %" #i ", %" #i ", %8, %9\n"
#define A_7(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0xca\n"
#define A_8(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_9(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define A_10(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define A_11(i) "v_sub_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define A_12(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_13(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[4:5]\n"
#define A_14(i) "v_cmp_le_i32 vcc, %" #i ", %8\n"
#define A_15(i) "v_cmp_le_i32_e64 s[6:7], %" #i ", %8\n"
#define A_16(i) "v_max_i32 %" #i ", %" #i ", %8\n"
#define A_17(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define A_18(i) "v_max_u16 %" #i ", %" #i ", %8\n"
#define A_19(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define A_20(i) "v_ffbh_u32 %" #i ", %" #i "\n"
#define A_21(i) "v_ffbl_b32 %" #i ", %" #i "\n"
#define A_22(i) "v_bfe_u32 %" #i ", %" #i ", %10, 1\n"
#define A_23(i) "v_lshrrev_b32 %" #i ", %10, %" #i "\n"
#define A_24(i) "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
#define A_25(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_26(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_27(i) "v_add_u32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_28(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_29(i) "v_sub_u16 %" #i ", %" #i ", %8\n"
#define A_30(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define A_31(i) "v_pk_max_i16 %" #i ", %" #i ", %8\n"
#define A_32(i) "v_pk_min_u16 %" #i ", %" #i ", %8\n"
#define A_33(i) "v_and_b32 %" #i ", s8, %" #i "\n"
#define A_34(i) "v_bitop3_b32 %" #i ", %" #i ", s8, %9 bitop3:0xea\n"
#define A_35(i) "v_lshl_or_b32 %" #i ", %" #i ", 4, s8\n"
#define A_36(i) "v_or_b32 %" #i ", s8, %" #i "\n"
#define A_37(i) "v_add_u32 %" #i ", 0x1001ff, %" #i "\n"
#define A_38(i) "v_and_b32 %" #i ", 0xfff3ffff, %" #i "\n"
#define A_39(i) "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_40(i) "v_bfe_i32 %" #i ", %" #i ", 9, 1\n"

KERNEL(k_0, A_0) KERNEL(k_1, A_1) KERNEL(k_2, A_2) KERNEL(k_3, A_3) KERNEL(k_4, A_4) KERNEL(k_5, A_5) KERNEL(k_6, A_6) KERNEL(k_7, A_7) KERNEL(k_8, A_8) KERNEL(k_9, A_9) KERNEL(k_10, A_10) KERNEL(k_11, A_11) KERNEL(k_12, A_12) KERNEL(k_13, A_13) KERNEL(k_14, A_14) KERNEL(k_15, A_15) KERNEL(k_16, A_16) KERNEL(k_17, A_17) KERNEL(k_18, A_18) KERNEL(k_19, A_19) KERNEL(k_20, A_20) KERNEL(k_21, A_21) KERNEL(k_22, A_22) KERNEL(k_23, A_23) KERNEL(k_24, A_24) KERNEL(k_25, A_25) KERNEL(k_26, A_26) KERNEL(k_27, A_27) KERNEL(k_28, A_28) KERNEL(k_29, A_29) KERNEL(k_30, A_30) KERNEL(k_31, A_31) KERNEL(k_32, A_32) KERNEL(k_33, A_33) KERNEL(k_34, A_34) KERNEL(k_35, A_35) KERNEL(k_36, A_36) KERNEL(k_37, A_37) KERNEL(k_38, A_38) KERNEL(k_39, A_39) KERNEL(k_40, A_40)

int main() {
    unsigned* d;
    hipMalloc(&d, 256 * 4 * 256 * 64);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    struct K { const char* name; void (*fn)(unsigned*, unsigned); };
    K ks[] = {{"v_and_b32", k_0}, {"v_add_u32 x,x", k_1}, {"v_lshlrev_b32 1", k_2}, {"v_lshrrev_b32 31", k_3}, {"v_ashrrev_i32 31", k_4}, {"v_sub_u32", k_5}, {"v_bfi_b32", k_6}, {"v_bitop3 0xca", k_7}, {"v_add3_u32", k_8}, {"v_add_co_u32", k_9}, {"v_addc_co_u32", k_10}, {"v_sub_co_u32", k_11}, {"v_cndmask vcc", k_12}, {"v_cndmask e64 s[4:5]", k_13}, {"v_cmp_le_i32 vcc", k_14}, {"v_cmp_le_i32 e64 s[6:7]", k_15}, {"v_max_i32", k_16}, {"v_min_u32", k_17}, {"v_max_u16", k_18}, {"v_bcnt_u32_b32", k_19}, {"v_ffbh_u32", k_20}, {"v_ffbl_b32", k_21}, {"v_bfe_u32 reg", k_22}, {"v_lshrrev_b32 reg", k_23}, {"v_lshl_or_b32 1", k_24}, {"v_mul_lo_u32", k_25}, {"v_mov_b32 dpp row_shr:1", k_26}, {"v_add_u32 dpp row_shr:1", k_27}, {"v_xor_b32", k_28}, {"v_sub_u16", k_29}, {"v_pk_add_u16", k_30}, {"v_pk_max_i16", k_31}, {"v_pk_min_u16", k_32}, {"v_and_b32 sgpr", k_33}, {"v_bitop3 sgpr operand", k_34}, {"v_lshl_or sgpr operand", k_35}, {"v_or_b32 sgpr", k_36}, {"v_add_u32 literal", k_37}, {"v_and_b32 literal", k_38}, {"v_min3_u32", k_39}, {"v_bfe_i32 const", k_40}};
    for (int wps : {2, 4}) {
        printf("---- %d wave(s) per SIMD ----\n", wps);
        for (auto& k : ks) {
            const int blocks = cus * wps;            // 256 threads = 4 waves = one per SIMD
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, 1u);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, 2u);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double insts = (double)ITERS * 16 * wps;               // per SIMD
            printf("%-26s %8.3f ms  %6.2f ns per instruction and SIMD  (= %.2f cycles at 2.1 GHz)\n", k.name, ms,
                   ms * 1e6 / insts, ms * 1e6 / insts * 2.1);
        }
    }
    return 0;
}
