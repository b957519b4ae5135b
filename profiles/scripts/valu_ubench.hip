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
    for (int i = 0; i < ITERS; ++i) {                                                               \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                        \
                     ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                        \
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                     : "v"(a), "v"(b), "v"(sh) : "vcc");                                            \
    }                                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;             \
}

#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define A_BITOP(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x80\n"
#define A_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 4, %8\n"
#define A_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 4, %" #i "\n"
#define A_SDWA(i) "v_lshlrev_b32_sdwa %" #i ", %10, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CMP(i) "v_cmp_ne_u32 vcc, %" #i ", %8\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define A_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 8\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_DOT4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
#define A_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define A_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define A_ANDSDWA(i) "v_and_b32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define A_PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define A_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define A_SAD(i) "v_sad_u8 %" #i ", %" #i ", %8, %9\n"

KERNEL(k_and, A_AND) KERNEL(k_add, A_ADD) KERNEL(k_or3, A_OR3) KERNEL(k_bitop3, A_BITOP) KERNEL(k_lshl_or, A_LSHLOR)
KERNEL(k_and_or, A_ANDOR) KERNEL(k_lshl, A_LSHL) KERNEL(k_sdwa, A_SDWA) KERNEL(k_cndmask, A_CNDMASK) KERNEL(k_cmp, A_CMP)
KERNEL(k_perm, A_PERM) KERNEL(k_alignbit, A_ALIGNBIT) KERNEL(k_mov, A_MOV) KERNEL(k_xor, A_XOR) KERNEL(k_lshl_add, A_LSHLADD)
KERNEL(k_add3, A_ADD3) KERNEL(k_dot4, A_DOT4) KERNEL(k_mul24, A_MUL24) KERNEL(k_mad24, A_MAD24) KERNEL(k_bfe, A_BFE)
KERNEL(k_and_sdwa, A_ANDSDWA) KERNEL(k_pk_add, A_PKADD) KERNEL(k_xad, A_XAD) KERNEL(k_sad, A_SAD)

int main() {
    unsigned* d;
    hipMalloc(&d, 256 * 4 * 256 * 64);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    struct K { const char* name; void (*fn)(unsigned*, unsigned); };
    K ks[] = {{"v_and_b32", k_and}, {"v_add_u32", k_add}, {"v_xor_b32", k_xor}, {"v_mov_b32", k_mov}, {"v_or3_b32", k_or3},
              {"v_bitop3_b32", k_bitop3}, {"v_lshl_or_b32", k_lshl_or}, {"v_and_or_b32", k_and_or}, {"v_lshlrev_b32", k_lshl},
              {"v_lshlrev_b32_sdwa", k_sdwa}, {"v_and_b32_sdwa", k_and_sdwa}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_ne_u32", k_cmp},
              {"v_perm_b32", k_perm}, {"v_alignbit_b32", k_alignbit}, {"v_lshl_add_u32", k_lshl_add}, {"v_add3_u32", k_add3},
              {"v_dot4_u32_u8", k_dot4}, {"v_mul_u32_u24", k_mul24}, {"v_mad_u32_u24", k_mad24}, {"v_bfe_u32", k_bfe},
              {"v_pk_add_u16", k_pk_add}, {"v_xad_u32", k_xad}, {"v_sad_u8", k_sad}};
    for (int wps : {1, 2, 4}) {
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
            printf("%-22s %8.3f ms  %6.2f ns per instruction and SIMD  (= %.2f cycles at 2.1 GHz)\n", k.name, ms,
                   ms * 1e6 / insts, ms * 1e6 / insts * 2.1);
        }
    }
    return 0;
}
