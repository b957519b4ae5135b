// bucket.hip -- ragged batches for plans of SEVERAL adapters: the reads sorted into buckets of one length each, so
// that every bucket is a batch of equally long reads and takes the streaming multi-adapter form (multi2.hip:
// k_multi_stream / k_multi_scan work on one read length per launch).
//
// The reference hands its aligner one str of any length at a time (adapters.py:815-832, MultipleAdapters.match_to
// :1265-1286); behind the quality trimmers a pipeline's reads are ragged (cli.py:938-954).  Per read nothing changes:
// a bucket's reads are COPIES of the views' characters, results are relative to the view's first character either
// way, and cah_scatter_results puts every row back at its read's index.
//
//   cah_length_histogram   hist[L] = reads of L characters (L <= max_len; longer ones in hist[max_len + 1])
//   cah_bucket_reads       slot = first[L] + (rank of the read inside its bucket, by atomic), perm[slot] = read,
//                          characters copied to dst[base[L] + rank * L ..)
//   cah_scatter_results    out6 / status / best_adapter rows of slot i go to read perm[i]
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cutadapt_hip.h"

extern int cah_set_error_(int code, const char* msg);   // api.cpp

namespace {

__global__ void k_length_histogram(const int32_t* lens, const int64_t* offsets, int64_t n, int32_t max_len, unsigned long long* hist) {
    extern __shared__ unsigned s_hist[];
    for (int i = threadIdx.x; i < max_len + 2; i += blockDim.x) s_hist[i] = 0u;
    __syncthreads();
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t L = lens ? (int64_t)lens[r] : offsets[r + 1] - offsets[r];
        atomicAdd(&s_hist[L < 0 ? 0 : (L > max_len ? max_len + 1 : (int)L)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < max_len + 2; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&hist[i], (unsigned long long)s_hist[i]);
}

// one wave per 64 reads: the lanes take their slots (one atomic per read on its bucket's cursor), then the wave copies
// read after read with 64 lanes side by side (coalesced on both sides)
__global__ void k_bucket_reads(const uint8_t* seqs, const int64_t* offsets, const int32_t* lens, int64_t n, int32_t max_len,
                               const int64_t* base, const int64_t* first, unsigned long long* cursor, uint8_t* dst, int32_t* perm) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r0 = wave * 64; r0 < n; r0 += waves * 64) {
        const int64_t r = r0 + lane;
        int64_t src = 0, to = 0;
        int L = 0;
        if (r < n) {
            src = offsets[r];
            const int64_t l64 = lens ? (int64_t)lens[r] : offsets[r + 1] - src;
            L = (int)(l64 < 0 ? 0 : (l64 > max_len ? max_len + 1 : l64));
            const int64_t rank = (int64_t)atomicAdd(&cursor[L], 1ull);
            perm[first[L] + rank] = (int32_t)r;
            to = base[L] + rank * L;
            if (L > max_len) L = 0;                                  // (longer reads are not copied: their bucket is served in place)
        }
        for (int t = 0; t < 64; ++t) {
            const int64_t s_t = __shfl(src, t, 64), d_t = __shfl(to, t, 64);
            const int L_t = __shfl(L, t, 64);
            for (int c = lane; c < L_t; c += 64) dst[d_t + c] = seqs[s_t + c];
        }
    }
}

__global__ void k_scatter_results(const int32_t* perm, int64_t n, const int32_t* t6, const uint8_t* tst, const int32_t* tbest,
                                  int32_t* out6, uint8_t* status, int32_t* best) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * 6; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t slot = i / 6;
        const int c = (int)(i - slot * 6);
        const int64_t r = perm[slot];
        out6[r * 6 + c] = t6[i];
        if (c == 0) {
            status[r] = tst[slot];
            if (best && tbest) best[r] = tbest[slot];
        }
    }
}

int grid_for(int64_t n, int per_block) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return (int)g;
}

}  // namespace

extern "C" {

int cah_length_histogram(const int64_t* d_offsets, const int32_t* d_lens, int64_t n_reads, int32_t max_len,
                         unsigned long long* d_hist, void* stream) {
    if (n_reads < 0 || max_len < 0 || max_len > 65534 || !d_hist || (n_reads > 0 && !d_offsets && !d_lens))
        return cah_set_error_(CAH_EINVAL, "cah_length_histogram: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(d_hist, 0, sizeof(unsigned long long) * ((size_t)max_len + 2), s) != hipSuccess)
        return cah_set_error_(CAH_EHIP, "cah_length_histogram: memset failed");
    if (n_reads == 0) return CAH_OK;
    hipLaunchKernelGGL(k_length_histogram, dim3(grid_for(n_reads, 256 * 64)), dim3(256), sizeof(unsigned) * ((size_t)max_len + 2), s,
                       d_lens, d_offsets, n_reads, max_len, d_hist);
    return hipGetLastError() == hipSuccess ? CAH_OK : cah_set_error_(CAH_EHIP, "cah_length_histogram: launch failed");
}

int cah_bucket_reads(const uint8_t* d_seqs, const int64_t* d_offsets, const int32_t* d_lens, int64_t n_reads, int32_t max_len,
                     const int64_t* d_base, const int64_t* d_first, unsigned long long* d_cursor, uint8_t* d_dst,
                     int32_t* d_perm, void* stream) {
    if (n_reads < 0 || n_reads > 2147483647LL || max_len < 0 || max_len > 65534 || !d_base || !d_first || !d_cursor || !d_perm ||
        (n_reads > 0 && (!d_seqs || !d_offsets || !d_dst)))
        return cah_set_error_(CAH_EINVAL, "cah_bucket_reads: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(d_cursor, 0, sizeof(unsigned long long) * ((size_t)max_len + 2), s) != hipSuccess)
        return cah_set_error_(CAH_EHIP, "cah_bucket_reads: memset failed");
    if (n_reads == 0) return CAH_OK;
    hipLaunchKernelGGL(k_bucket_reads, dim3(grid_for(n_reads, 256 * 4)), dim3(256), 0, s, d_seqs, d_offsets, d_lens, n_reads, max_len,
                       d_base, d_first, d_cursor, d_dst, d_perm);
    return hipGetLastError() == hipSuccess ? CAH_OK : cah_set_error_(CAH_EHIP, "cah_bucket_reads: launch failed");
}

int cah_scatter_results(const int32_t* d_perm, int64_t n_reads, const int32_t* d_tmp_out6, const uint8_t* d_tmp_status,
                        const int32_t* d_tmp_best, int32_t* d_out6, uint8_t* d_status, int32_t* d_best_adapter, void* stream) {
    if (n_reads < 0 || (n_reads > 0 && (!d_perm || !d_tmp_out6 || !d_tmp_status || !d_out6 || !d_status)))
        return cah_set_error_(CAH_EINVAL, "cah_scatter_results: bad argument");
    if (n_reads == 0) return CAH_OK;
    hipLaunchKernelGGL(k_scatter_results, dim3(grid_for(n_reads * 6, 256 * 8)), dim3(256), 0, (hipStream_t)stream, d_perm, n_reads,
                       d_tmp_out6, d_tmp_status, d_tmp_best, d_out6, d_status, d_best_adapter);
    return hipGetLastError() == hipSuccess ? CAH_OK : cah_set_error_(CAH_EHIP, "cah_scatter_results: launch failed");
}

}  // extern "C"
