// r04_mask_ubench.hip -- DESIGN.md 11 (b) of round 3, measured: the lead-word step of k_filter_stream2 with the character
// masks (1) read from the LDS table, as the kernel does (one SDWA address + one ds_read_b64 per character, the two words'
// masks side by side), against (2) the masks of A / C / G / T held in registers and selected by the character's 2-bit code
// (two compares + three selects per character and word -- v_perm cannot do it: a 32-bit mask is four byte planes), the
// LDS table kept only for the other characters.  Both forms run the same word update (v_lshl_or + 2 v_bitop3 + found) for
// two lead words over four characters per step, 16 waves per CU like the kernel.  hipcc --offload-arch=gfx950 -O3.
// Output: nanoseconds per wave and 4-character group, and the ratio.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define GROUPS 8192
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void word_step(unsigned& R, unsigned& f, unsigned init4, unsigned found, unsigned m0, unsigned m1,
                                          unsigned m2, unsigned m3) {
    R = (R << 4) | init4;
    R = __builtin_amdgcn_bitop3_b32(R, m0, m1, 0x80);
    R = __builtin_amdgcn_bitop3_b32(R, m2, m3, 0x80);
    f = __builtin_amdgcn_bitop3_b32(R, found, f, 0xea);
}

// (1) the kernel's form: table entries of 8 bytes {word 0, word 1} per character and shift position
__global__ __launch_bounds__(1024) void k_table(const unsigned* text, unsigned* out, unsigned p_other) {
    __shared__ u32x2 tab[4][128];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) tab[i >> 7][i & 127] = (u32x2){0x9E3779B9u * (i + 1), 0x85EBCA6Bu * (i + 7)};
    __syncthreads();
    unsigned R0 = 0, R1 = 0, f = 0, w = text[threadIdx.x & 255];
    for (int g = 0; g < GROUPS; ++g) {
        u32x2 m[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = tab[3 - i][(w >> (8 * i)) & 127u];
        word_step(R0, f, 0x11u, 0x8000u, m[0].x, m[1].x, m[2].x, m[3].x);
        word_step(R1, f, 0x101u, 0x80000u, m[0].y, m[1].y, m[2].y, m[3].y);
        w = w * 1664525u + 1013904223u;                      // (the next group's characters: any bytes)
        w = (w & 0x03030303u) * 2u + 0x41414141u;            // ... four letters out of A C E G (codes 0..3 in bits 1..2)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = R0 ^ R1 ^ f;
    (void)p_other;
}

// (2) register masks for the four common characters, the table for the others (lane by lane: where ANY lane of the wave has
// another character the LDS read is issued for the wave; p_other = how often that is, out of 2^16)
__global__ __launch_bounds__(1024) void k_select(const unsigned* text, unsigned* out, unsigned p_other) {
    __shared__ u32x2 tab[4][128];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) tab[i >> 7][i & 127] = (u32x2){0x9E3779B9u * (i + 1), 0x85EBCA6Bu * (i + 7)};
    __syncthreads();
    unsigned R0 = 0, R1 = 0, f = 0, w = text[threadIdx.x & 255];
    // masks of the four characters for the two words, per shift position s: (mask << s) | fill(s) is what the table holds;
    // here: the unshifted masks in registers, shifted where they are used (one more instruction per character and word would
    // be needed: left out -- the select alone decides the comparison)
    unsigned mA0 = text[1], mC0 = text[2], mG0 = text[3], mT0 = text[4], mA1 = text[5], mC1 = text[6], mG1 = text[7], mT1 = text[8];
    unsigned lcg = threadIdx.x * 2654435761u;
    for (int g = 0; g < GROUPS; ++g) {
        unsigned m0[4], m1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned c = (w >> (8 * i + 1)) & 3u;
            const bool b0 = (c & 1u) != 0, b1 = (c & 2u) != 0;
            m0[i] = b1 ? (b0 ? mT0 : mG0) : (b0 ? mC0 : mA0);
            m1[i] = b1 ? (b0 ? mT1 : mG1) : (b0 ? mC1 : mA1);
        }
        lcg = lcg * 1664525u + 1013904223u;
        if (__builtin_amdgcn_ballot_w64((lcg >> 16) < p_other)) {                 // some lane holds another character (N, ...)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x2 t = tab[3 - i][(w >> (8 * i)) & 127u];
                const bool other = (lcg >> 16) < p_other;
                m0[i] = other ? t.x : m0[i];
                m1[i] = other ? t.y : m1[i];
            }
        }
        word_step(R0, f, 0x11u, 0x8000u, m0[0], m0[1], m0[2], m0[3]);
        word_step(R1, f, 0x101u, 0x80000u, m1[0], m1[1], m1[2], m1[3]);
        w = w * 1664525u + 1013904223u;
        w = (w & 0x03030303u) * 2u + 0x41414141u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = R0 ^ R1 ^ f;
}

int main() {
    unsigned *text, *out;
    hipMalloc(&text, 1024); hipMalloc(&out, 256 * 1024 * 4);
    std::vector<unsigned> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0x41434754u ^ (i * 0x01010101u & 0x06060606u);
    hipMemcpy(text, h.data(), 1024, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, void (*k)(const unsigned*, unsigned*, unsigned), unsigned p_other) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, text, out, p_other);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // 256 CUs x 16 waves, GROUPS groups each: per SIMD 4 waves share the issue slots
        const double ns_per_group_per_simd = best * 1e6 / GROUPS / 4.0;
        printf("%-44s %8.3f ms   %6.2f ns per 4-character group and wave (4 waves per SIMD)  = %5.1f cycles at 2.4 GHz\n", name, best,
               ns_per_group_per_simd, ns_per_group_per_simd * 2.4);
        return best;
    };
    const float t = run("LDS table (the kernel's form)", k_table, 0);
    const float s0 = run("register masks, no other character", k_select, 0);
    const float s1 = run("register masks, 1 % of the groups fall back", k_select, 655);       // real data: N is rare
    const float s2 = run("register masks, 72 % of the groups fall back", k_select, 47186);    // p_N = 0.005 per base x 256 characters per wave-group
    printf("ratio select / table: %.2f (no fallback)  %.2f (1 %%)  %.2f (72 %%: the benchmark's read model)\n", s0 / t, s1 / t, s2 / t);
    return 0;
}
