// group_ubench.hip -- the prefilter's lead-word group (4 SDWA address extractions, 4 ds_read_b64 table lookups, two word
// steps) in isolation: cycles per group at 16 waves per CU, with the lookups one group ahead (as in k_filter_stream2).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void addr4(unsigned (&ad)[4], const unsigned w, const unsigned shv) {
    asm("v_lshlrev_b32_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_lshlrev_b32_sdwa %1, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
        "v_lshlrev_b32_sdwa %2, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
        "v_lshlrev_b32_sdwa %3, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3"
        : "=&v"(ad[0]), "=&v"(ad[1]), "=&v"(ad[2]), "=&v"(ad[3]) : "v"(shv), "v"(w));
}
template <int MODE>   // 0: full group, 1: no LDS reads, 2: no SDWA (addresses reused), 3: LDS reads only
__global__ __launch_bounds__(1024) void k_group(unsigned* out, int iters, unsigned S0, unsigned S1, unsigned F0, unsigned F1) {
    __shared__ __attribute__((aligned(16))) unsigned tab[4 * 256 * 2];
    for (int i = threadIdx.x; i < 4 * 256 * 2; i += blockDim.x) tab[i] = 0xFFFFFFFFu ^ (i * 2654435761u >> 28);
    __syncthreads();
    const unsigned char* T = reinterpret_cast<const unsigned char*>(tab);
    unsigned w = 0x41434754u ^ (threadIdx.x * 0x01000193u & 0x03030303u);   // bytes in 'A'..'W'
    unsigned shv = 3; asm volatile("" : "+v"(shv));
    unsigned R0 = threadIdx.x, R1 = threadIdx.x * 3, f = 0;
    unsigned la[4]; u32x2 m[2][4];
    addr4(la, w, shv);
    for (int i = 0; i < 4; ++i) m[0][i] = *reinterpret_cast<const u32x2*>(T + (3 - i) * 2048 + la[i]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (MODE != 2) addr4(la, w + g, shv);
            if (MODE != 1) { for (int i = 0; i < 4; ++i) m[(g + 1) & 1][i] = *reinterpret_cast<const u32x2*>(T + (3 - i) * 2048 + la[i]); }
            else { for (int i = 0; i < 4; ++i) m[(g + 1) & 1][i] = (u32x2){la[i], la[i] + 1}; }
            if (MODE != 3) {
                R0 = (R0 << 4) | S0;
                R0 = __builtin_amdgcn_bitop3_b32(R0, m[g & 1][0].x, m[g & 1][1].x, 0x80);
                R0 = __builtin_amdgcn_bitop3_b32(R0, m[g & 1][2].x, m[g & 1][3].x, 0x80);
                f = R0 & F0;
                R1 = (R1 << 4) | S1;
                R1 = __builtin_amdgcn_bitop3_b32(R1, m[g & 1][0].y, m[g & 1][1].y, 0x80);
                R1 = __builtin_amdgcn_bitop3_b32(R1, m[g & 1][2].y, m[g & 1][3].y, 0x80);
                f = __builtin_amdgcn_bitop3_b32(R1, F1, f, 0xea);
            } else {
                f ^= m[g & 1][0].x ^ m[g & 1][1].y ^ m[g & 1][2].x ^ m[g & 1][3].y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        w = (w + f) & 0x5F5F5F5Fu | 0x40404040u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = R0 ^ R1 ^ f;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 1024 * 4);
    hipDeviceProp_t p; hipGetDevicePropertiesR0600(&p, 0);
    const int iters = 20000;
    void (*ks[])(unsigned*, int, unsigned, unsigned, unsigned, unsigned) = {k_group<0>, k_group<1>, k_group<2>, k_group<3>};
    const char* names[] = {"full group", "no LDS reads", "no SDWA", "LDS reads only"};
    for (int wpb : {1024, 512, 256}) {
        for (int k = 0; k < 4; ++k) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(ks[k], dim3(p.multiProcessorCount), dim3(wpb), 0, 0, d, 100, 0x11u, 0x101u, 0x8000u, 0x80u);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(ks[k], dim3(p.multiProcessorCount), dim3(wpb), 0, 0, d, iters, 0x11u, 0x101u, 0x8000u, 0x80u);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%4d threads/CU  %-16s %7.3f ms  = %6.1f ns per group and wave = %6.1f cycles at 2.1 GHz\n", wpb, names[k], ms,
                   ms * 1e6 / (iters * 4.0), ms * 1e6 / (iters * 4.0) * 2.1);
        }
    }
    return 0;
}
