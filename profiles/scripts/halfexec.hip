// does gfx950 skip the idle half of a wave64 VALU instruction when EXEC[63:32] == 0 ?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int mode, unsigned* out, int iters) {
    unsigned lane = threadIdx.x & 63;
    bool on = mode == 0 ? true : mode == 1 ? lane < 32 : mode == 2 ? (lane & 1) == 0 : mode == 3 ? lane < 16 : lane >= 32;
    unsigned a = threadIdx.x, b = a * 3 + 1, c = a * 5 + 2, d = a * 7 + 3;
    if (on) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int t = 0; t < 16; t++) {
                asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_xor_b32 %3, %3, %0"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}
__global__ void k4(int mode, unsigned* out, int iters) {   // 4-cycle ops: v_min_u32 / v_cndmask
    unsigned lane = threadIdx.x & 63;
    bool on = mode == 0 ? true : mode == 1 ? lane < 32 : mode == 2 ? (lane & 1) == 0 : mode == 3 ? lane < 16 : lane >= 32;
    unsigned a = threadIdx.x, b = a * 3 + 1, c = a * 5 + 2, d = a * 7 + 3;
    if (on) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int t = 0; t < 16; t++) {
                asm volatile("v_min_u32 %0, %0, %1\n v_max_u32 %1, %1, %2\n v_min_u32 %2, %2, %3\n v_max_u32 %3, %3, %0"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}
int main() {
    unsigned* out; hipMalloc(&out, 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[5] = {"all 64 lanes", "lanes 0-31", "even lanes", "lanes 0-15", "lanes 32-63"};
    for (int which = 0; which < 2; which++)
    for (int mode = 0; mode < 5; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, mode, out, 4000);
            else hipLaunchKernelGGL(k4, dim3(2048), dim3(256), 0, 0, mode, out, 4000);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1) printf("%s %-14s %.3f ms\n", which == 0 ? "add/xor (2-cycle)" : "min/max (4-cycle)", names[mode], ms);
        }
    }
    return 0;
}
