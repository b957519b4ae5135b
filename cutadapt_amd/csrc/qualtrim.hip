// qualtrim.hip -- quality / NextSeq / poly-A trimming positions and expected errors for a batch
// (the O(n) per-read scans that run just before adapter matching,
// reference src/cutadapt/qualtrim.pyx:22-190, expected_errors.h:103-140, applied by
// modifiers.py:825-879).  One read per lane; the scans stop early exactly where the reference's
// loops break, so a lane usually touches only the first / last few characters.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdio>
#include <mutex>

#include "../../include/cutadapt_hip.h"
#include "revcomp.h"
#include "dev_common.h"

extern int cah_set_error_(int code, const char* msg);   // api.cpp

#define QT_TRY(expr)                                                                   \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess) {                                                       \
            char b__[256];                                                             \
            snprintf(b__, sizeof(b__), "%s failed: %s", #expr, hipGetErrorString(e__)); \
            return cah_set_error_(CAH_EHIP, b__);                                      \
        }                                                                              \
    } while (0)

namespace {

__device__ __forceinline__ void qt_extent(const int64_t* offsets, const int32_t* lens, int64_t r, int64_t& off, int& n) {
    off = offsets[r];
    const int64_t n64 = lens ? (int64_t)lens[r] : offsets[r + 1] - off;
    n = (int)(n64 > 0x7FFFFFFF ? 0x7FFFFFFF : n64);
}

// 16 characters per global load: reads are packed back to back, so a chunk starts at an arbitrary
// byte; gfx950 executes unaligned dwordx4 loads natively.  The caller guarantees p[0..15] is inside
// the read.
struct __attribute__((packed, aligned(1))) QtUnaligned16 { uint32_t w[4]; };
__device__ __forceinline__ QtUnaligned16 qt_load16(const uint8_t* p) { return *reinterpret_cast<const QtUnaligned16*>(p); }
__device__ __forceinline__ uint8_t qt_byte(const QtUnaligned16& c, int t) { return (uint8_t)(c.w[t >> 2] >> ((t & 3) * 8)); }

// quality_trim_index (qualtrim.pyx:22-70): BWA-style partial sums from both ends; the quality
// bytes are read as C `char` (signed), like the reference does.
__global__ __launch_bounds__(256) void k_quality_trim(const uint8_t* quals, const int64_t* offsets, const int32_t* lens,
                                                      int64_t n_reads, int cutoff_front, int cutoff_back, int base,
                                                      int32_t* start_stop) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    int64_t off; int n;
    qt_extent(offsets, lens, r, off, n);
    const signed char* q = reinterpret_cast<const signed char*>(quals + off);
    int start = 0, stop = n;
    int s = 0, max_qual = 0;
    const uint8_t* qb = quals + off;
    {                                                                // 5' end, :50-57
        auto step = [&](const signed char c, const int i) -> bool {
            s += cutoff_front - ((int)c - base);
            if (s < 0) return true;
            if (s > max_qual) { max_qual = s; start = i + 1; }
            return false;
        };
        int i = 0;
        bool done = false;
        for (; i + 16 <= n && !done; i += 16) {
            const QtUnaligned16 c = qt_load16(qb + i);
#pragma unroll
            for (int t = 0; t < 16; t++) if (!done) done = step((signed char)qt_byte(c, t), i + t);
        }
        for (; i < n && !done; i++) done = step(q[i], i);
    }
    max_qual = 0; s = 0;
    {                                                                // 3' end, :60-67
        auto step = [&](const signed char c, const int i) -> bool {
            s += cutoff_back - ((int)c - base);
            if (s < 0) return true;
            if (s > max_qual) { max_qual = s; stop = i; }
            return false;
        };
        int i = n;
        bool done = false;
        for (; i >= 16 && !done; i -= 16) {
            const QtUnaligned16 c = qt_load16(qb + i - 16);
#pragma unroll
            for (int t = 15; t >= 0; t--) if (!done) done = step((signed char)qt_byte(c, t), i - 16 + t);
        }
        for (i = i - 1; i >= 0 && !done; i--) done = step(q[i], i);
    }
    if (start >= stop) { start = 0; stop = 0; }                      // :68-69
    start_stop[2 * r] = start;
    start_stop[2 * r + 1] = stop;
}

// nextseq_trim_index (qualtrim.pyx:73-113): as above from the 3' end, 'G' counts as cutoff - 1
// (qual_offsets != NULL: the qualities of read r start at quals[qual_offsets[r]] -- sequences and qualities matched
// in place in a raw FASTQ chunk; NULL: they are packed like the sequences)
__global__ __launch_bounds__(256) void k_nextseq_trim(const uint8_t* seqs, const uint8_t* quals, const int64_t* offsets,
                                                      const int64_t* qual_offsets, const int32_t* lens, int64_t n_reads,
                                                      int cutoff, int base, int32_t* stop_out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    int64_t off; int n;
    qt_extent(offsets, lens, r, off, n);
    const uint8_t* b = seqs + off;
    if (qual_offsets) off = qual_offsets[r];
    const signed char* q = reinterpret_cast<const signed char*>(quals + off);
    int s = 0, max_qual = 0, max_i = n;
    auto step = [&](const signed char qc, const uint8_t bc, const int i) -> bool {
        int qv = (int)qc - base;
        if (bc == 'G') qv = cutoff - 1;
        s += cutoff - qv;
        if (s < 0) return true;
        if (s > max_qual) { max_qual = s; max_i = i; }
        return false;
    };
    int i = n;
    bool done = false;
    for (; i >= 16 && !done; i -= 16) {
        const QtUnaligned16 cq = qt_load16(quals + off + i - 16);
        const QtUnaligned16 cb = qt_load16(b + i - 16);
#pragma unroll
        for (int t = 15; t >= 0; t--) if (!done) done = step((signed char)qt_byte(cq, t), qt_byte(cb, t), i - 16 + t);
    }
    for (i = i - 1; i >= 0 && !done; i--) done = step(q[i], b[i], i);
    stop_out[r] = max_i;
}

// poly_a_trim_index (qualtrim.pyx:116-165): +1 per A (T with revcomp), -2 otherwise, error rate <= 0.2,
// tails shorter than 3 ignored.  No early exit in the reference: the whole read is scanned.
__global__ __launch_bounds__(256) void k_poly_a_trim(const uint8_t* seqs, const int64_t* offsets, const int32_t* lens,
                                                     int64_t n_reads, int revcomp, int32_t* index_out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    int64_t off; int n;
    qt_extent(offsets, lens, r, off, n);
    const uint8_t* b = seqs + off;
    int best_score = 0, score = 0, errors = 0, best_index;
    if (revcomp) {
        best_index = 0;
        auto step = [&](const uint8_t c, const int i) {
            if (c == 'T') score += 1; else { score -= 2; errors += 1; }
            if (score > best_score && errors * 5 <= i + 1) { best_score = score; best_index = i + 1; }
        };
        int i = 0;
        for (; i + 16 <= n; i += 16) {
            const QtUnaligned16 c = qt_load16(b + i);
#pragma unroll
            for (int t = 0; t < 16; t++) step(qt_byte(c, t), i + t);
        }
        for (; i < n; i++) step(b[i], i);
        if (best_index < 3) best_index = 0;
    } else {
        best_index = n;
        auto step = [&](const uint8_t c, const int i) {
            if (c == 'A') score += 1; else { score -= 2; errors += 1; }
            if (score > best_score && errors * 5 <= n - i) { best_score = score; best_index = i; }
        };
        int i = n;                                                   // positions i-1, i-2, ... are still to do
        for (; i >= 16; i -= 16) {
            const QtUnaligned16 c = qt_load16(b + i - 16);
#pragma unroll
            for (int t = 15; t >= 0; t--) step(qt_byte(c, t), i - 16 + t);
        }
        for (i = i - 1; i >= 0; i--) step(b[i], i);
        if (best_index > n - 3) best_index = n;
    }
    index_out[r] = best_index;
}

// expected_errors (qualtrim.pyx:168-190, expected_errors.h:103-140): sum of 10^(-q/10) with the
// reference's FOUR interleaved double accumulators and its final e0+e1+e2+e3, so the double result
// is bit-identical.  An invalid phred value (byte < base or > 126) gives status CAH_INVALID.
__global__ __launch_bounds__(256) void k_expected_errors(const uint8_t* quals, const int64_t* offsets, const int32_t* lens,
                                                         int64_t n_reads, int base, const double* table, double* out,
                                                         uint8_t* status) {
    __shared__ double s_tab[94];
    if (threadIdx.x < 94) s_tab[threadIdx.x] = table[threadIdx.x];
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    int64_t off; int n;
    qt_extent(offsets, lens, r, off, n);
    const uint8_t* q = quals + off;
    const uint8_t ubase = (uint8_t)base;
    const uint8_t max_phred = (uint8_t)(126 - ubase);
    double e0 = 0.0, e1 = 0.0, e2 = 0.0, e3 = 0.0;
    bool bad = false;
    int i = 0;
    // 16 characters = four of the reference's groups of four (accumulator = position mod 4)
    for (; i + 16 <= n && !bad; i += 16) {
        const QtUnaligned16 c = qt_load16(q + i);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint8_t p0 = (uint8_t)(qt_byte(c, 4 * g) - ubase), p1 = (uint8_t)(qt_byte(c, 4 * g + 1) - ubase);
            const uint8_t p2 = (uint8_t)(qt_byte(c, 4 * g + 2) - ubase), p3 = (uint8_t)(qt_byte(c, 4 * g + 3) - ubase);
            if (p0 > max_phred || p1 > max_phred || p2 > max_phred || p3 > max_phred) { bad = true; break; }
            e0 += s_tab[p0]; e1 += s_tab[p1]; e2 += s_tab[p2]; e3 += s_tab[p3];
        }
    }
    for (; i + 3 < n && !bad; i += 4) {
        const uint8_t p0 = (uint8_t)(q[i] - ubase), p1 = (uint8_t)(q[i + 1] - ubase);
        const uint8_t p2 = (uint8_t)(q[i + 2] - ubase), p3 = (uint8_t)(q[i + 3] - ubase);
        if (p0 > max_phred || p1 > max_phred || p2 > max_phred || p3 > max_phred) { bad = true; break; }
        e0 += s_tab[p0]; e1 += s_tab[p1]; e2 += s_tab[p2]; e3 += s_tab[p3];
    }
    if (!bad) {
        for (; i < n; i++) {
            const uint8_t p = (uint8_t)(q[i] - ubase);
            if (p > max_phred) { bad = true; break; }
            e0 += s_tab[p];
        }
    }
    out[r] = bad ? -1.0 : e0 + e1 + e2 + e3;
    if (status) status[r] = bad ? CAH_INVALID : CAH_MATCH;
}

// SCORE_TO_ERROR_RATE of expected_errors.h:6-101 is 10^(-q/10) in double; one copy per device
struct ErrTable {
    double* d[64] = {nullptr};
    std::mutex mu;                      // the pipeline calls in from several threads
};
ErrTable g_tab;

int error_table_on_device(const double** out) {
    int device = 0;
    QT_TRY(hipGetDevice(&device));
    if (device < 0 || device >= 64) return cah_set_error_(CAH_EUNSUPPORTED, "device index too large");
    std::lock_guard<std::mutex> lk(g_tab.mu);
    if (!g_tab.d[device]) {
        double h[94];
        for (int q = 0; q < 94; q++) h[q] = std::pow(10.0, -(double)q / 10.0);
        double* p = nullptr;
        QT_TRY(hipMalloc((void**)&p, sizeof(h)));
        QT_TRY(hipMemcpy(p, h, sizeof(h), hipMemcpyHostToDevice));
        g_tab.d[device] = p;
    }
    *out = g_tab.d[device];
    return CAH_OK;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace


// ---------------------------------------------------------------------------------------------
// k_reverse_reads: every read reversed into a second packed buffer.  Rightmost* adapters search the reversed read
// with the reversed adapter (reference adapters.py:766, :870: `sequence[::-1]`); one read per lane, 16 characters
// per load (the chunk that ENDS where the last one began), byte-swapped in registers, one 16-byte store; the last
// partial chunk byte by byte (a 16-byte store there would reach into the next read's slot).
// COMPLEMENT: the reverse COMPLEMENT (ReverseComplementer, reference modifiers.py:264-308; table in revcomp.h, held
// in 256 bytes of LDS).  select != NULL: reads with select[r] == 0 are copied as they are -- that merges "the
// orientation that matched better" of every read into one batch in one pass.
// ---------------------------------------------------------------------------------------------
template <bool COMPLEMENT>
__global__ __launch_bounds__(256) void k_reverse_reads(const uint8_t* seqs, const int64_t* offsets, const int32_t* lens,
                                                       int64_t n_reads, const int64_t* out_offsets, uint8_t* out,
                                                       const uint8_t* select) {
    __shared__ uint8_t comp[256];
    if (COMPLEMENT) {
        comp[threadIdx.x] = cah_complement((uint8_t)threadIdx.x);
        __syncthreads();
    }
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int64_t off = offsets[r];
    const int64_t n64 = lens ? (int64_t)lens[r] : offsets[r + 1] - off;
    const int n = (int)(n64 > CAH_MAX_READ_LEN ? CAH_MAX_READ_LEN : n64);
    const uint8_t* q = seqs + off;
    uint8_t* o = out + out_offsets[r];
    if (select && !select[r]) {                                  // this read keeps its orientation
        int done = 0;
        for (; done + 16 <= n; done += 16) {
            const Chunk c = load_chunk(q, done, n, done + 16);
            Unaligned16 u;
            u.w[0] = c.w[0]; u.w[1] = c.w[1]; u.w[2] = c.w[2]; u.w[3] = c.w[3];
            __builtin_memcpy(o + done, &u, 16);
        }
        for (int i = done; i < n; ++i) o[i] = q[i];
        return;
    }
    auto flip = [&](unsigned w) -> unsigned {                    // the four characters of a dword, reversed (+ complemented)
        if (!COMPLEMENT) return __builtin_bswap32(w);
        return ((unsigned)comp[w & 0xFFu] << 24) | ((unsigned)comp[(w >> 8) & 0xFFu] << 16)
             | ((unsigned)comp[(w >> 16) & 0xFFu] << 8) | (unsigned)comp[w >> 24];
    };
    int done = 0;
    for (; done + 16 <= n; done += 16) {
        const int pos = n - done - 16;                           // input characters [pos, pos + 16)
        const Chunk c = load_chunk(q, pos, n, pos + 16);
        Unaligned16 u;
        u.w[0] = flip(c.w[3]); u.w[1] = flip(c.w[2]); u.w[2] = flip(c.w[1]); u.w[3] = flip(c.w[0]);
        __builtin_memcpy(o + done, &u, 16);
    }
    for (int i = done; i < n; ++i) o[i] = COMPLEMENT ? comp[q[n - 1 - i]] : q[n - 1 - i];
}

// k_linked_views: where the second stage of a linked adapter looks -- view r = read[rstop_r:], rstop_r = query_stop of the
// front match or 0 (reference adapters.py:1222-1224).  One pass over the front stage's results instead of a chain of
// elementwise tensor operations (compare, gather, where, two casts, two adds and a subtraction over n reads each).
__global__ __launch_bounds__(256) void k_linked_views(const int32_t* out6, const uint8_t* status, const int64_t* offsets,
                                                      const int32_t* lens, const int32_t read_len, const int64_t n_reads,
                                                      int64_t* starts, int32_t* view_lens) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    int64_t off, n;
    if (read_len > 0) { off = r * (int64_t)read_len; n = read_len; }
    else { off = offsets[r]; n = lens ? (int64_t)lens[r] : offsets[r + 1] - off; }
    int64_t stop = status[r] == 1 ? (int64_t)out6[r * 6 + 3] : 0;
    stop = stop < 0 ? 0 : (stop > n ? n : stop);
    starts[r] = off + stop;
    view_lens[r] = (int32_t)(n - stop);
}

extern "C" {

int cah_quality_trim_batch(const uint8_t* d_quals, const int64_t* d_offsets, const int32_t* d_lens, int64_t n_reads,
                           int32_t cutoff_front, int32_t cutoff_back, int32_t base, int32_t* d_start_stop, void* stream) {
    if (n_reads < 0) return cah_set_error_(CAH_EINVAL, "n_reads < 0");
    if (n_reads == 0) return CAH_OK;
    if (!d_offsets || !d_start_stop) return cah_set_error_(CAH_EINVAL, "cah_quality_trim_batch: NULL argument");
    hipLaunchKernelGGL(k_quality_trim, dim3(blocks_for(n_reads)), dim3(256), 0, (hipStream_t)stream, d_quals, d_offsets,
                       d_lens, n_reads, cutoff_front, cutoff_back, base, d_start_stop);
    QT_TRY(hipGetLastError());
    return CAH_OK;
}

int cah_nextseq_trim_batch(const uint8_t* d_seqs, const uint8_t* d_quals, const int64_t* d_offsets, const int32_t* d_lens,
                           int64_t n_reads, int32_t cutoff, int32_t base, int32_t* d_stop, void* stream) {
    if (n_reads < 0) return cah_set_error_(CAH_EINVAL, "n_reads < 0");
    if (n_reads == 0) return CAH_OK;
    if (!d_offsets || !d_stop) return cah_set_error_(CAH_EINVAL, "cah_nextseq_trim_batch: NULL argument");
    hipLaunchKernelGGL(k_nextseq_trim, dim3(blocks_for(n_reads)), dim3(256), 0, (hipStream_t)stream, d_seqs, d_quals,
                       d_offsets, (const int64_t*)nullptr, d_lens, n_reads, cutoff, base, d_stop);
    QT_TRY(hipGetLastError());
    return CAH_OK;
}

// ... for reads whose qualities are not packed like their sequences: d_qual_offsets int64[n_reads] (a raw FASTQ chunk
// in HBM, indexed by cah_fastq_index_device: d_offsets into the sequence lines, d_qual_offsets into the quality lines)
int cah_nextseq_trim_batch_q(const uint8_t* d_seqs, const uint8_t* d_quals, const int64_t* d_offsets,
                             const int64_t* d_qual_offsets, const int32_t* d_lens, int64_t n_reads, int32_t cutoff,
                             int32_t base, int32_t* d_stop, void* stream) {
    if (n_reads < 0) return cah_set_error_(CAH_EINVAL, "n_reads < 0");
    if (n_reads == 0) return CAH_OK;
    if (!d_offsets || !d_qual_offsets || !d_lens || !d_stop)
        return cah_set_error_(CAH_EINVAL, "cah_nextseq_trim_batch_q: NULL argument");
    hipLaunchKernelGGL(k_nextseq_trim, dim3(blocks_for(n_reads)), dim3(256), 0, (hipStream_t)stream, d_seqs, d_quals,
                       d_offsets, d_qual_offsets, d_lens, n_reads, cutoff, base, d_stop);
    QT_TRY(hipGetLastError());
    return CAH_OK;
}

int cah_poly_a_trim_batch(const uint8_t* d_seqs, const int64_t* d_offsets, const int32_t* d_lens, int64_t n_reads,
                          int32_t revcomp, int32_t* d_index, void* stream) {
    if (n_reads < 0) return cah_set_error_(CAH_EINVAL, "n_reads < 0");
    if (n_reads == 0) return CAH_OK;
    if (!d_offsets || !d_index) return cah_set_error_(CAH_EINVAL, "cah_poly_a_trim_batch: NULL argument");
    hipLaunchKernelGGL(k_poly_a_trim, dim3(blocks_for(n_reads)), dim3(256), 0, (hipStream_t)stream, d_seqs, d_offsets,
                       d_lens, n_reads, revcomp, d_index);
    QT_TRY(hipGetLastError());
    return CAH_OK;
}

int cah_expected_errors_batch(const uint8_t* d_quals, const int64_t* d_offsets, const int32_t* d_lens, int64_t n_reads,
                              int32_t base, double* d_expected, uint8_t* d_status, void* stream) {
    if (n_reads < 0) return cah_set_error_(CAH_EINVAL, "n_reads < 0");
    if (base < 0 || base > 126) return cah_set_error_(CAH_EINVAL, "quality base out of range");
    if (n_reads == 0) return CAH_OK;
    if (!d_offsets || !d_expected) return cah_set_error_(CAH_EINVAL, "cah_expected_errors_batch: NULL argument");
    const double* tab = nullptr;
    int rc = error_table_on_device(&tab);
    if (rc) return rc;
    hipLaunchKernelGGL(k_expected_errors, dim3(blocks_for(n_reads)), dim3(256), 0, (hipStream_t)stream, d_quals, d_offsets,
                       d_lens, n_reads, base, tab, d_expected, d_status);
    QT_TRY(hipGetLastError());
    return CAH_OK;
}

// Reversed copy of every read (see k_reverse_reads).  d_out_offsets: int64[n_reads] start of each read in d_out
// (packed: the running sum of the lengths); all device pointers.
static int reverse_reads_impl(const char* who, const uint8_t* d_seqs, const int64_t* d_offsets, const int32_t* d_lens,
                              int64_t n_reads, const int64_t* d_out_offsets, uint8_t* d_out, int complement,
                              const uint8_t* d_select, void* stream) {
    if (n_reads < 0) return cah_set_error_(CAH_EINVAL, "n_reads < 0");
    if (n_reads == 0) return CAH_OK;
    if (!d_offsets || !d_out_offsets || !d_out) {
        char msg[96];
        snprintf(msg, sizeof msg, "%s: NULL argument", who);
        return cah_set_error_(CAH_EINVAL, msg);
    }
    const int64_t blocks = (n_reads + 255) / 256;
    if (complement)
        hipLaunchKernelGGL(k_reverse_reads<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_seqs, d_offsets,
                           d_lens, n_reads, d_out_offsets, d_out, d_select);
    else
        hipLaunchKernelGGL(k_reverse_reads<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_seqs, d_offsets,
                           d_lens, n_reads, d_out_offsets, d_out, d_select);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cah_set_error_(CAH_EHIP, hipGetErrorString(e));
    return CAH_OK;
}

// Views for the second stage of a linked adapter (see k_linked_views): d_starts int64[n_reads], d_view_lens
// int32[n_reads].  The reads are (d_offsets, d_lens) as in cah_match_batch, or -- read_len > 0 -- equally long reads back
// to back from byte 0 (d_offsets may be NULL).
int cah_linked_views(const int32_t* d_out6_front, const uint8_t* d_status_front, const int64_t* d_offsets,
                     const int32_t* d_lens, int32_t read_len, int64_t n_reads, int64_t* d_starts, int32_t* d_view_lens,
                     void* stream) {
    if (n_reads < 0) return cah_set_error_(CAH_EINVAL, "n_reads < 0");
    if (n_reads == 0) return CAH_OK;
    if (!d_out6_front || !d_status_front || !d_starts || !d_view_lens || (read_len <= 0 && !d_offsets))
        return cah_set_error_(CAH_EINVAL, "cah_linked_views: NULL argument");
    hipLaunchKernelGGL(k_linked_views, dim3(blocks_for(n_reads)), dim3(256), 0, (hipStream_t)stream, d_out6_front,
                       d_status_front, d_offsets, d_lens, read_len, n_reads, d_starts, d_view_lens);
    QT_TRY(hipGetLastError());
    return CAH_OK;
}

int cah_reverse_reads_batch(const uint8_t* d_seqs, const int64_t* d_offsets, const int32_t* d_lens, int64_t n_reads,
                            const int64_t* d_out_offsets, uint8_t* d_out, void* stream) {
    return reverse_reads_impl("cah_reverse_reads_batch", d_seqs, d_offsets, d_lens, n_reads, d_out_offsets, d_out, 0,
                              nullptr, stream);
}

// The reverse complement of every read (complement != 0; complement == 0 only reverses, which is what quality
// strings need), or -- with d_select: uint8[n_reads] -- of the reads with d_select[r] != 0 while the others are
// copied unchanged.  Replaces, per batch, dnaio's SequenceRecord.reverse_complement() as the reference's
// ReverseComplementer calls it (modifiers.py:280).
int cah_revcomp_reads_batch(const uint8_t* d_seqs, const int64_t* d_offsets, const int32_t* d_lens, int64_t n_reads,
                            const int64_t* d_out_offsets, uint8_t* d_out, int32_t complement, const uint8_t* d_select,
                            void* stream) {
    return reverse_reads_impl("cah_revcomp_reads_batch", d_seqs, d_offsets, d_lens, n_reads, d_out_offsets, d_out,
                              complement, d_select, stream);
}

}  // extern "C"
