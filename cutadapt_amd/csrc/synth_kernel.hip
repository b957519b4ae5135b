// synth_kernel.hip -- device-side generator of the synthetic benchmark reads
// (SURVEY.md section 8d).  One read per thread; read r is a pure function of
// (seed, first_index + r), so every rank can fill its own shard directly in HBM and the
// CPU twin (oracle/synth_reads.c) can regenerate any sub-range for parity checks.
// Benchmark utility, not part of the matching path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

#define GOLDEN 0x9E3779B97F4A7C15ULL

__device__ __forceinline__ uint64_t mix64(uint64_t z) {   // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t stream_key(uint64_t seed, uint64_t index, uint64_t stream) {
    return mix64(mix64(seed + GOLDEN * (stream + 1)) ^ (index * GOLDEN));
}
__device__ __forceinline__ uint64_t draw(uint64_t key, uint64_t counter) {
    return mix64(key + GOLDEN * (counter + 1));
}
__device__ __forceinline__ uint8_t base_char(unsigned b) {
    return (uint8_t)(0x54474341u >> (8 * (b & 3u)));      // "ACGT"
}

__global__ __launch_bounds__(256) void k_synth(uint64_t seed, int64_t first_index, int64_t n_reads,
                                               int32_t read_len, uint32_t p_adapter_u32,
                                               uint32_t p_edit_u32, uint32_t p_n_u16,
                                               const char* adapters, const int32_t* adapter_off,
                                               int32_t n_adapters, uint8_t* seqs, int64_t* offsets) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_reads; r += stride) {
        offsets[r] = r * (int64_t)read_len;
        if (r == n_reads) break;
        const uint64_t idx = (uint64_t)(first_index + r);
        uint8_t* out = seqs + r * (int64_t)read_len;
        // stream 0: bases, 32 per draw
        const uint64_t k0 = stream_key(seed, idx, 0);
        for (int j = 0; j < read_len; j += 32) {
            const uint64_t w = draw(k0, (uint64_t)(j >> 5));
            for (int t = 0; t < 32 && j + t < read_len; t++) out[j + t] = base_char((unsigned)(w >> (2 * t)));
        }
        // stream 1: adapter insertion with edits
        const uint64_t k1 = stream_key(seed, idx, 1);
        const uint64_t u = draw(k1, 0);
        if (n_adapters > 0 && (uint32_t)(u >> 32) < p_adapter_u32) {
            const int which = (int)((u & 0xFFFFFFFFULL) % (uint64_t)n_adapters);
            const char* ad = adapters + adapter_off[which];
            const int m = adapter_off[which + 1] - adapter_off[which];
            int pos = (int)(draw(k1, 1) % (uint64_t)(read_len + 1));
            for (int i = 0; i < m && pos < read_len; i++) {
                const uint64_t e = draw(k1, (uint64_t)(2 + i));
                const char c = ad[i];
                if ((uint32_t)(e & 0xFFFFFFFFULL) < p_edit_u32) {
                    const unsigned kind = (unsigned)(e >> 32) & 3u;
                    const unsigned rb = (unsigned)(e >> 34) & 3u;
                    if (kind < 2) {
                        const unsigned cur = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u;
                        const unsigned step = 1u + (unsigned)((e >> 36) % 3ULL);
                        out[pos++] = base_char(cur + step);
                    } else if (kind == 2) {
                        out[pos++] = base_char(rb);
                        if (pos < read_len) out[pos++] = (uint8_t)c;
                    }
                } else {
                    out[pos++] = (uint8_t)c;
                }
            }
        }
        // stream 2: N substitution
        if (p_n_u16) {
            const uint64_t k2 = stream_key(seed, idx, 2);
            for (int j = 0; j < read_len; j += 4) {
                const uint64_t w = draw(k2, (uint64_t)(j >> 2));
                for (int t = 0; t < 4 && j + t < read_len; t++)
                    if (((w >> (16 * t)) & 0xFFFFULL) < p_n_u16) out[j + t] = 'N';
            }
        }
    }
}

hipError_t launch_synth(uint64_t seed, int64_t first_index, int64_t n_reads, int32_t read_len,
                        uint32_t p_adapter_u32, uint32_t p_edit_u32, uint32_t p_n_u16,
                        const char* d_adapters, const int32_t* d_adapter_off, int32_t n_adapters,
                        uint8_t* d_seqs, int64_t* d_offsets, int n_cus, hipStream_t s) {
    int64_t need = (n_reads + 1 + 255) / 256;
    int64_t cap = (int64_t)n_cus * 16;
    const int grid = (int)(need < cap ? need : cap);
    hipLaunchKernelGGL(k_synth, dim3(grid), dim3(256), 0, s, seed, first_index, n_reads, read_len,
                       p_adapter_u32, p_edit_u32, p_n_u16, d_adapters, d_adapter_off, n_adapters, d_seqs,
                       d_offsets);
    return hipGetLastError();
}
