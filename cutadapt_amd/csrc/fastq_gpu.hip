// fastq_gpu.hip -- FASTQ record indexing and trimmed-record formatting ON THE GPU (SURVEY.md section 8(f)
// rows 1-2: the data formats either side of the matching path).
//
// The host-side twins (fastq.cpp: cah_fastq_scan, cah_pack_sequences, cah_records_write) parse and format
// at a few hundred MB/s per core; with the matcher at several Greads/s they bound the end-to-end rate
// (round 1: 21 Mreads/s with 16 threads, GPU idle > 95 %).  Here the raw chunk goes to HBM as it is:
//   1. k_nl_count / k_tile_scan / k_nl_write   positions of all line feeds (16-byte loads, exact
//                                              zero-byte mask, tile counts + one-block scan)
//   2. k_records                               four lines = one record: offsets of name, sequence and
//                                              qualities ("\r\n" accepted), format checks of
//                                              dnaio / fastq.cpp (leading '@' and '+', equal lengths)
//   3. (matching works directly on the raw chunk: read r = buf[seq_off[r] : +seq_len[r]], the `lens` view
//      of the C ABI -- nothing is packed or copied)
//   4. k_out_len / scan / k_format             "@name\nSEQ[beg:end]\n+\nQUAL[beg:end]\n" of every kept
//                                              record, written at its exclusive-scan offset
// so that only raw FASTQ bytes go in over PCIe and only trimmed FASTQ bytes come out; the host cuts chunks
// at record starts (cah_record_boundary) and does nothing per read.
// Reference: dnaio.read_chunks + SequenceRecord parsing (reference src/cutadapt/files.py:108-114,
// runners.py:116-126), Match.trimmed() slicing (adapters.py:453-454, :486-487), dnaio's FASTQ writer.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>

#include "../../include/cutadapt_hip.h"
#include "revcomp.h"

extern int cah_set_error_(int code, const char* msg);   // api.cpp

namespace {

#define NL_TILE_BYTES 16384          // one 256-thread block: 64 bytes per thread
#define SCAN_ELEMS 2048              // elements per block of the generic int64 scan (8 per thread)

// exact mask (bit 7 of each byte) of the bytes of v that equal c
__device__ __forceinline__ unsigned eq_mask(unsigned v, unsigned c4) {
    const unsigned x = v ^ c4;
    const unsigned t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | x | 0x7F7F7F7Fu);
}

// the 64 bytes of thread t of tile `tile`, zero beyond len; w[16]
__device__ __forceinline__ void load64(const uint8_t* buf, int64_t len, int64_t base, unsigned (&w)[16]) {
    if (base + 64 <= len && ((uintptr_t)(buf + base) & 15) == 0) {
        const uint4* p = reinterpret_cast<const uint4*>(buf + base);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const uint4 v = p[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            unsigned v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int64_t q = base + 4 * i + b;
                if (q < len) v |= (unsigned)buf[q] << (8 * b);
            }
            w[i] = v;
        }
    }
}

__device__ __forceinline__ int block_reduce_sum(int v, int* s_red) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    const int total = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(256) void k_nl_count(const uint8_t* buf, int64_t len, int64_t n_tiles, int64_t* tile_counts) {
    __shared__ int s_red[4];
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        unsigned w[16];
        load64(buf, len, tile * NL_TILE_BYTES + (int64_t)threadIdx.x * 64, w);
        int c = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) c += __popc(eq_mask(w[i], 0x0A0A0A0Au));
        const int total = block_reduce_sum(c, s_red);
        if (threadIdx.x == 0) tile_counts[tile] = total;
    }
}

// exclusive scan of counts[0..n) in place, total to *total (one block; n is a few thousand)
__global__ __launch_bounds__(1024) void k_tile_scan(int64_t* counts, int64_t n, int64_t* total) {
    __shared__ int64_t s_part[1024];
    __shared__ int64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = i < n ? counts[i] : 0;
        s_part[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {                // Hillis-Steele inclusive scan
            const int64_t add = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0;
            __syncthreads();
            s_part[threadIdx.x] += add;
            __syncthreads();
        }
        const int64_t incl = s_part[threadIdx.x];
        if (i < n) counts[i] = s_carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = s_carry;
}

// positions of the line feeds, in order: nl_pos[k] = offset of the k-th '\n'
__global__ __launch_bounds__(256) void k_nl_write(const uint8_t* buf, int64_t len, int64_t n_tiles, const int64_t* tile_base,
                                                  int64_t* nl_pos, int64_t cap) {
    __shared__ int s_cnt[256];
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        unsigned w[16];
        const int64_t base = tile * NL_TILE_BYTES + (int64_t)threadIdx.x * 64;
        load64(buf, len, base, w);
        int c = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) c += __popc(eq_mask(w[i], 0x0A0A0A0Au));
        s_cnt[threadIdx.x] = c;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int add = threadIdx.x >= d ? s_cnt[threadIdx.x - d] : 0;
            __syncthreads();
            s_cnt[threadIdx.x] += add;
            __syncthreads();
        }
        int64_t k = tile_base[tile] + (s_cnt[threadIdx.x] - c);
        __syncthreads();
        if (c) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                unsigned m = eq_mask(w[i], 0x0A0A0A0Au);
                while (m) {
                    const int b = (__ffs((int)m) - 1) >> 3;
                    m &= m - 1;
                    if (k < cap) nl_pos[k] = base + 4 * i + b;
                    ++k;
                }
            }
        }
    }
}

// Record r = lines 4r .. 4r+3.  info[1] = first bad record + 1 (0: none), info[2] = its error code.
//   1: name line does not start with '@'   2: third line does not start with '+'   3: lengths differ
__global__ __launch_bounds__(256) void k_records(const uint8_t* buf, int64_t len, const int64_t* nl_pos, int64_t n_newlines,
                                                 int64_t n_records, int64_t* rec6, int64_t* seq_off, int32_t* seq_len,
                                                 unsigned long long* info) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_records; r += stride) {
        int64_t ls[4], le[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int64_t li = 4 * r + l;
            ls[l] = li == 0 ? 0 : nl_pos[li - 1] + 1;
            int64_t e = li < n_newlines ? nl_pos[li] : len;        // the last line of a file may lack its line feed
            if (e > ls[l] && buf[e - 1] == '\r') --e;
            le[l] = e;
        }
        int err = 0;
        if (le[0] == ls[0] || buf[ls[0]] != '@') err = 1;
        else if (le[2] == ls[2] || buf[ls[2]] != '+') err = 2;
        else if (le[1] - ls[1] != le[3] - ls[3]) err = 3;
        if (err) {
            const unsigned long long code = ((unsigned long long)(r + 1) << 8) | (unsigned)err;
            atomicMin(&info[1], code);
        }
        int64_t* o = rec6 + r * 6;
        o[0] = ls[0] + 1; o[1] = le[0]; o[2] = ls[1]; o[3] = le[1]; o[4] = ls[3]; o[5] = le[3];
        seq_off[r] = ls[1];
        seq_len[r] = (int32_t)(le[1] - ls[1]);
    }
}

// (flags / suffix_len: --revcomp, the records with flags[r] != 0 carry a suffix behind their name; flags NULL: none does)
__global__ __launch_bounds__(256) void k_out_len(const int64_t* rec6, int64_t n_records, const int32_t* beg, const int32_t* end,
                                                 const uint8_t* keep, int64_t* out_len, const uint8_t* flags = nullptr,
                                                 const int suffix_len = 0) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_records; r += stride) {
        int64_t v = 0;
        if (!keep || keep[r]) {
            const int64_t* o = rec6 + r * 6;
            const int64_t seq_len = o[3] - o[2];
            int64_t a = beg[r], b = end[r];
            if (a < 0) a = 0;
            if (b > seq_len) b = seq_len;
            if (b < a) b = a;
            v = 1 + (o[1] - o[0]) + 1 + (b - a) + 1 + 2 + (b - a) + 1;
            if (flags && flags[r]) v += suffix_len;
        }
        out_len[r] = v;
    }
}

// generic exclusive scan over int64, SCAN_ELEMS per block: block sums, one-block scan of the sums, apply
__global__ __launch_bounds__(256) void k_scan_sums(const int64_t* in, int64_t n, int64_t* block_sums) {
    __shared__ int64_t s_red[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_ELEMS;
    int64_t v = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ELEMS / 256; ++i) {
        const int64_t q = base + (int64_t)threadIdx.x * (SCAN_ELEMS / 256) + i;
        if (q < n) v += in[q];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

__global__ __launch_bounds__(256) void k_scan_apply(const int64_t* in, int64_t n, const int64_t* block_base, int64_t* out) {
    __shared__ int64_t s_part[256];
    const int64_t base = (int64_t)blockIdx.x * SCAN_ELEMS + (int64_t)threadIdx.x * (SCAN_ELEMS / 256);
    int64_t loc[SCAN_ELEMS / 256];
    int64_t sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ELEMS / 256; ++i) { loc[i] = base + i < n ? in[base + i] : 0; sum += loc[i]; }
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int64_t add = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0;
        __syncthreads();
        s_part[threadIdx.x] += add;
        __syncthreads();
    }
    int64_t run = block_base[blockIdx.x] + s_part[threadIdx.x] - sum;
#pragma unroll
    for (int i = 0; i < SCAN_ELEMS / 256; ++i) {
        if (base + i < n) out[base + i] = run;
        run += loc[i];
    }
}

struct NameSuffix { uint8_t c[CAH_MAX_NAME_SUFFIX]; int len; };

// one wave per record (round-robin inside a block): "@name\nSEQ[a:b]\n+\nQUAL[a:b]\n"; SUFFIX: "@name<suffix>\n..." for
// the records with flags[r] != 0 (ReverseComplementer's rc_suffix, reference modifiers.py:296-297)
template <bool SUFFIX>
__global__ __launch_bounds__(256) void k_format(const uint8_t* buf, const int64_t* rec6, int64_t n_records, const int32_t* beg,
                                                const int32_t* end, const uint8_t* keep, const int64_t* out_off,
                                                uint8_t* out, int64_t out_cap, const uint8_t* flags, const NameSuffix sfx) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < n_records; r += n_waves) {
        if (keep && !keep[r]) continue;
        const int64_t* o = rec6 + r * 6;
        const int64_t name_len = o[1] - o[0], seq_len = o[3] - o[2];
        int64_t a = beg[r], b = end[r];
        if (a < 0) a = 0;
        if (b > seq_len) b = seq_len;
        if (b < a) b = a;
        const int64_t body = b - a;
        int64_t pos = out_off[r];
        const int extra = SUFFIX && flags[r] ? sfx.len : 0;
        if (pos + 1 + name_len + extra + 1 + body + 3 + body + 1 > out_cap) continue;       // never: the caller sizes out for the input
        uint8_t* w = out + pos;
        if (lane == 0) w[0] = '@';
        for (int64_t k = lane; k < name_len; k += 64) w[1 + k] = buf[o[0] + k];
        w += 1 + name_len;
        if (SUFFIX) {
            if (lane < extra) w[lane] = sfx.c[lane];
            w += extra;
        }
        if (lane == 0) w[0] = '\n';
        for (int64_t k = lane; k < body; k += 64) w[1 + k] = buf[o[2] + a + k];
        w += 1 + body;
        if (lane == 0) { w[0] = '\n'; w[1] = '+'; w[2] = '\n'; }
        for (int64_t k = lane; k < body; k += 64) w[3 + k] = buf[o[4] + a + k];
        if (lane == 0) w[3 + body] = '\n';
    }
}

// --info-file rows on the device (reference steps.py:232-253, adapters.py:395-417), one line per record:
//   a read with a match   name[suffix] \t errors \t rstart \t rstop \t seq[:rstart] \t seq[rstart:rstop] \t seq[rstop:] \t adapter
//                         \t qual[:rstart] \t qual[rstart:rstop] \t qual[rstop:] \t rc \n
//                         (seq = the WHOLE read as it came in -- turned around if the reverse complement won --, cut at the
//                         coordinates of the match, which were found on what the modifiers in front of the adapter step left
//                         of it: steps.py:233-236 starts from info.original_read; rc = "", "0" or "1": RC_MAP, :224)
//   a read without one    name \t -1 \t seq[fb:fe] \t qual[fb:fe] \n      (the read as it is written, :248-251)
struct InfoArgs {
    const uint8_t* buf; const int64_t* rec6; int64_t n_records;
    // the adapter step's results, round after round (--times N): round k's rows of record r at [k * n_records + r]
    const int32_t* out6; const uint8_t* status; const int32_t* best; int rounds;
    const uint8_t* kinds;                                            // per adapter: 0 3' (keeps what is in front), 1 5', 2 anywhere
    const int32_t* final_beg; const int32_t* final_end;
    const uint8_t* names; const int32_t* name_off; int n_names;
    const uint8_t* is_rc;                                            // NULL: the rc column stays empty
    NameSuffix sfx;
};

__device__ __forceinline__ int n_digits(int v) {
    int d = 1;
    for (int t = v < 0 ? 0 : v; t >= 10; t /= 10) ++d;
    return d;
}

__device__ __forceinline__ void put_digits(uint8_t* w, int v, const int d) {
    if (v < 0) v = 0;
    for (int i = d - 1; i >= 0; --i) { w[i] = (uint8_t)('0' + v % 10); v /= 10; }
}

// One match row: the read as InfoFileWriter holds it at that round -- [qb, qe) of the read as it came in: steps.py:233-247
// starts from info.original_read and trims it the way every match trims (`current_read = match.trimmed(current_read)`) -- cut
// at the match's coordinates (found on what the earlier modifiers and rounds left; clamped to that read as the host writer
// clamps them, pipeline.py: _info_rows_on_original)
struct InfoRow { int err, rs, re, ni, nlen; int64_t qb, qe; };

__device__ __forceinline__ bool info_round(const InfoArgs& a, const int64_t r, const int k, int64_t& qb, int64_t& qe, InfoRow& x) {
    const int64_t at = (int64_t)k * a.n_records + r;
    if (a.status[at] != 1) return false;
    const int32_t* m = a.out6 + at * 6;
    int64_t re = m[3], rs = m[2];
    if (re > qe - qb) re = qe - qb;
    if (re < 0) re = 0;
    if (rs > re) rs = re;
    if (rs < 0) rs = 0;
    x.err = m[5]; x.rs = (int)rs; x.re = (int)re; x.qb = qb; x.qe = qe;
    x.ni = a.best[at];
    if (x.ni < 0 || x.ni >= a.n_names) x.ni = 0;
    x.nlen = a.n_names > 0 ? a.name_off[x.ni + 1] - a.name_off[x.ni] : 0;
    // what the match leaves (Match.trimmed: a 5' match keeps what follows it, a 3' match what is in front; an "anywhere"
    // adapter counts as 5' when the match starts at position 0: adapters.py:453-454, :486-487, :931)
    const int kind = a.kinds ? a.kinds[x.ni] : 0;
    const bool before = kind == 1 || (kind == 2 && m[2] == 0);
    if (before) qb = qb + re; else qe = qb + rs;
    return true;
}

__device__ __forceinline__ int64_t info_match_len(const InfoArgs& a, const InfoRow& x, const int64_t name_len, const int extra) {
    const int64_t shown = x.qe - x.qb;
    return name_len + extra + 1 + n_digits(x.err) + 1 + n_digits(x.rs) + 1 + n_digits(x.re) + 1 + shown + 3 + x.nlen + 1
         + shown + 2 + 1 + (a.is_rc ? 1 : 0) + 1;
}

__device__ __forceinline__ void info_final(const InfoArgs& a, const int64_t r, const int64_t seq_len, int64_t& fa, int64_t& fb) {
    fa = a.final_beg[r]; fb = a.final_end[r];
    if (fa < 0) fa = 0;
    if (fb > seq_len) fb = seq_len;
    if (fb < fa) fb = fa;
}

__global__ __launch_bounds__(256) void k_info_len(const InfoArgs a, int64_t* out_len) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_records; r += stride) {
        const int64_t* o = a.rec6 + r * 6;
        const int64_t name_len = o[1] - o[0], seq_len = o[3] - o[2];
        const int extra = a.is_rc && a.is_rc[r] ? a.sfx.len : 0;
        int64_t qb = 0, qe = seq_len, total = 0;
        InfoRow x;
        int rows = 0;
        for (int k = 0; k < a.rounds; ++k)
            if (info_round(a, r, k, qb, qe, x)) { total += info_match_len(a, x, name_len, extra); ++rows; }
        if (rows == 0) {
            int64_t fa, fb;
            info_final(a, r, seq_len, fa, fb);
            total = name_len + 4 + 2 * (fb - fa) + 2;
        }
        out_len[r] = total;
    }
}

// one wave per record
__global__ __launch_bounds__(256) void k_info_format(const InfoArgs a, const int64_t* out_off, uint8_t* out, int64_t out_cap) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < a.n_records; r += n_waves) {
        const int64_t* o = a.rec6 + r * 6;
        const int64_t name_len = o[1] - o[0], seq_len = o[3] - o[2];
        const int extra = a.is_rc && a.is_rc[r] ? a.sfx.len : 0;
        uint8_t* w = out + out_off[r];
        uint8_t* const cap = out + out_cap;
        auto copy = [&](const uint8_t* src, const int64_t len) {
            for (int64_t k = lane; k < len; k += 64) w[k] = src[k];
            w += len;
        };
        auto put = [&](const uint8_t c) { if (lane == 0) w[0] = c; ++w; };
        const uint8_t* seq = a.buf + o[2];
        const uint8_t* qual = a.buf + o[4];
        int64_t qb = 0, qe = seq_len;
        InfoRow x;
        int rows = 0;
        for (int k = 0; k < a.rounds; ++k) {
            if (!info_round(a, r, k, qb, qe, x)) continue;                   // (a linked adapter's optional part that is absent)
            ++rows;
            if (w + info_match_len(a, x, name_len, extra) > cap) break;     // never: the caller sizes out for the chunk
            copy(a.buf + o[0], name_len);
            if (lane < extra) w[lane] = a.sfx.c[lane];
            w += extra;
            put('\t');
            const int d0 = n_digits(x.err), d1 = n_digits(x.rs), d2 = n_digits(x.re);
            if (lane == 0) {
                put_digits(w, x.err, d0); w[d0] = '\t';
                put_digits(w + d0 + 1, x.rs, d1); w[d0 + 1 + d1] = '\t';
                put_digits(w + d0 + d1 + 2, x.re, d2); w[d0 + d1 + d2 + 2] = '\t';
            }
            w += d0 + d1 + d2 + 3;
            const int64_t a0 = x.qb + x.rs, a1 = x.qb + x.re;
            copy(seq + x.qb, x.rs); put('\t');
            copy(seq + a0, x.re - x.rs); put('\t');
            copy(seq + a1, x.qe - a1); put('\t');
            if (x.nlen) copy(a.names + a.name_off[x.ni], x.nlen);
            put('\t');
            copy(qual + x.qb, x.rs); put('\t');
            copy(qual + a0, x.re - x.rs); put('\t');
            copy(qual + a1, x.qe - a1); put('\t');
            if (a.is_rc) put(a.is_rc[r] ? '1' : '0');
            put('\n');
        }
        if (rows == 0) {
            int64_t fa, fb;
            info_final(a, r, seq_len, fa, fb);
            if (w + name_len + 4 + 2 * (fb - fa) + 2 > cap) continue;
            copy(a.buf + o[0], name_len);
            if (lane == 0) { w[0] = '\t'; w[1] = '-'; w[2] = '1'; w[3] = '\t'; }
            w += 4;
            copy(seq + fa, fb - fa);
            put('\t');
            copy(qual + fa, fb - fa);
            put('\n');
        }
    }
}

// ReverseComplementer's chosen orientation (reference modifiers.py:280-297), IN PLACE in the device's copy of the chunk: the
// window [win_beg[r], win_beg[r] + win_len[r]) of every record with flags[r] != 0 -- the read as the adapter step saw it
// -- becomes its reverse complement (revcomp.h) and the same window of the qualities is reversed.  One wave per record; a
// lane owns the characters k and n - 1 - k, loads both, stores both.
__global__ __launch_bounds__(256) void k_revcomp_in_place(uint8_t* buf, const int64_t* rec6, int64_t n_records,
                                                          const int32_t* win_beg, const int32_t* win_len, const uint8_t* flags) {
    __shared__ uint8_t comp[256];
    comp[threadIdx.x] = cah_complement((uint8_t)threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < n_records; r += n_waves) {
        if (!flags[r]) continue;
        const int64_t* o = rec6 + r * 6;
        const int64_t seq_len = o[3] - o[2];
        int64_t a = win_beg ? win_beg[r] : 0, n = win_len[r];
        if (a < 0) a = 0;
        if (a > seq_len) a = seq_len;
        if (n > seq_len - a) n = seq_len - a;
        uint8_t* s = buf + o[2] + a;
        uint8_t* q = buf + o[4] + a;
        for (int64_t k = lane; 2 * k < n; k += 64) {
            const int64_t m = n - 1 - k;
            const uint8_t s0 = s[k], s1 = s[m], q0 = q[k], q1 = q[m];
            s[k] = comp[s1]; s[m] = comp[s0];                // (k == m, the middle of an odd window: complemented once)
            q[k] = q1; q[m] = q0;
        }
    }
}

// AdapterCutter's marking actions (reference modifiers.py:170-198: masked_read / lowercased_read), IN PLACE in the device's
// copy of the chunk: of the characters [beg, end) of read r (the read as the adapter step saw it) those OUTSIDE [mark_beg,
// mark_end) -- what trimming would have removed -- become 'N' (mode 1: mask) or lower case, with the characters inside in
// upper case (mode 2: lowercase).  The modifiers behind the adapter step and the formatter then see the marked read, as
// the reference's do.  One wave per record.
__global__ __launch_bounds__(256) void k_mark_reads(uint8_t* buf, const int64_t* rec6, int64_t n_records, const int32_t* beg,
                                                    const int32_t* end, const int32_t* mark_beg, const int32_t* mark_end,
                                                    const int mode) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < n_records; r += n_waves) {
        // (a read without a match keeps its whole window: nothing to mask; --action=lowercase puts it in upper case like
        // every other read -- the reference upper-cases the read before it looks for adapters, modifiers.py:222-223)
        const int64_t* o = rec6 + r * 6;
        const int64_t seq_len = o[3] - o[2];
        int64_t a = beg[r], b = end[r];
        if (a < 0) a = 0;
        if (b > seq_len) b = seq_len;
        const int64_t mb = mark_beg[r], me = mark_end[r];
        for (int64_t k = a + lane; k < b; k += 64) {
            uint8_t c = buf[o[2] + k];
            const bool inside = k >= mb && k < me;
            if (mode == 1) c = inside ? c : (uint8_t)'N';
            else if (inside) c = (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c;           // str.upper() / str.lower() on ASCII
            else c = (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c;
            buf[o[2] + k] = c;
        }
    }
}

// What is left of every read after its best match, and whether the record is written: Match.trimmed() (read[rstop:]
// for a 5' match, read[:rstart] for a 3' match; an "anywhere" adapter counts as 5' when the match starts at
// position 0: reference adapters.py:453-454, :486-487, :931), then the filters in the reference's order
// (cli.py:735-912: too short, too long, --discard-trimmed / --discard-untrimmed).
// counters: [0] reads [1] with adapters [2] bp in [3] bp out (kept records) [4] too short [5] too long [6] invalid reads
// (win_beg / full_len, both or neither: the adapter step saw the window [win_beg[r], win_beg[r] + seq_len[r]) of a read of
// full_len[r] characters -- what the modifiers in front of it left, cli.py:938-954; beg / end are relative to the read)
__global__ __launch_bounds__(256) void k_trim_decide(const int32_t* out6, const uint8_t* status, const int32_t* best_adapter,
                                                     const int32_t* seq_len, const int32_t* win_beg, const int32_t* full_len,
                                                     int64_t n, const uint8_t* adapter_kind,
                                                     int32_t min_len, int32_t max_len, int32_t discard_trimmed,
                                                     int32_t discard_untrimmed, int32_t* beg, int32_t* end, uint8_t* keep,
                                                     unsigned long long* counters, const int32_t count_out = 1,
                                                     const int32_t action = 0) {
    __shared__ unsigned long long s_acc[7];
    if (threadIdx.x < 7) s_acc[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long acc[7] = {0, 0, 0, 0, 0, 0, 0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
        const int len = seq_len[r];
        const uint8_t st = status[r];
        const bool found = st == 1;
        int b = 0, e = len;
        if (found) {
            const int ad = best_adapter ? best_adapter[r] : 0;
            const uint8_t kind = adapter_kind[ad < 0 ? 0 : ad];       // 0: 3' (remove after), 1: 5' (remove before), 2: anywhere
            const int rstart = out6[r * 6 + 2], rstop = out6[r * 6 + 3];
            const bool before = kind == 1 || (kind == 2 && rstart == 0);
            // what the read keeps (reference modifiers.py:170-198, :225-251; Match.trimmed / retained_adapter_interval,
            // adapters.py:446-487): trim -- everything on the far side of the adapter; none -- everything; retain --
            // the same side INCLUDING the adapter; crop -- the adapter alone
            if (action == CAH_ACTION_TRIM) { if (before) b = rstop; else e = rstart; }
            else if (action == CAH_ACTION_RETAIN) { if (before) b = rstart; else e = rstop; }
            else if (action == CAH_ACTION_CROP) { b = rstart; e = rstop; }
        }
        const int out_len = e - b;
        bool k = true;
        if (min_len >= 0 && out_len < min_len) { acc[4] += 1; k = false; }
        if (k && max_len >= 0 && out_len > max_len) { acc[5] += 1; k = false; }
        if (discard_trimmed) k = k && !found;
        else if (discard_untrimmed) k = k && found;
        const int w0 = win_beg ? win_beg[r] : 0;
        beg[r] = w0 + b; end[r] = w0 + e; keep[r] = k ? 1 : 0;
        acc[0] += 1; acc[1] += found ? 1 : 0; acc[2] += (unsigned)(full_len ? full_len[r] : len); acc[3] += (k && count_out) ? (unsigned)out_len : 0u;
        acc[6] += st == 2 ? 1 : 0;
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) if (acc[i]) atomicAdd(&s_acc[i], acc[i]);
    __syncthreads();
    if (threadIdx.x < 7 && s_acc[threadIdx.x]) atomicAdd(&counters[threadIdx.x], s_acc[threadIdx.x]);
}

// The filters behind the modifiers, in the reference's order (cli.py:735-912: too short, too long, too many expected
// errors -- each counted on what the earlier ones left -- then --discard-trimmed / --discard-untrimmed), for reads
// whose kept interval [beg, end) is final.  counters (see k_trim_decide): [3] bp out, [4] too short, [5] too long,
// [7] too many expected errors.
__global__ __launch_bounds__(256) void k_trim_filter(const int32_t* beg, const int32_t* end, const uint8_t* status,
                                                     const double* ee, int64_t n, int32_t min_len, int32_t max_len,
                                                     double max_ee, int32_t discard_trimmed, int32_t discard_untrimmed,
                                                     uint8_t* keep, unsigned long long* counters) {
    __shared__ unsigned long long s_acc[4];
    if (threadIdx.x < 4) s_acc[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long acc[4] = {0, 0, 0, 0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
        const int out_len = end[r] - beg[r];
        const bool found = status[r] == 1;
        bool k = true;
        if (min_len >= 0 && out_len < min_len) { acc[1] += 1; k = false; }
        if (k && max_len >= 0 && out_len > max_len) { acc[2] += 1; k = false; }
        if (k && ee && max_ee >= 0.0 && ee[r] > max_ee) { acc[3] += 1; k = false; }
        if (discard_trimmed) k = k && !found;
        else if (discard_untrimmed) k = k && found;
        keep[r] = k ? 1 : 0;
        acc[0] += k ? (unsigned)out_len : 0u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) if (acc[i]) atomicAdd(&s_acc[i], acc[i]);
    __syncthreads();
    if (threadIdx.x < 4 && s_acc[threadIdx.x]) atomicAdd(&counters[threadIdx.x == 3 ? 7 : 3 + threadIdx.x], s_acc[threadIdx.x]);
}

int cus() {
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int c = 0;
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && c > 0) v = c;
    }
    return v;
}

int hip_fail(const char* what, hipError_t e) {
    char msg[256];
    snprintf(msg, sizeof(msg), "%s failed: %s", what, hipGetErrorString(e));
    return cah_set_error_(CAH_EHIP, msg);
}
#define GPU_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return hip_fail(#expr, e__); } while (0)

int64_t n_tiles_of(int64_t len) { return (len + NL_TILE_BYTES - 1) / NL_TILE_BYTES; }
int64_t n_scan_blocks(int64_t n) { return (n + SCAN_ELEMS - 1) / SCAN_ELEMS; }

}  // namespace

extern "C" {

size_t cah_fastq_device_scratch_bytes(int64_t chunk_bytes, int64_t max_records) {
    if (chunk_bytes < 0) chunk_bytes = 0;
    if (max_records < 0) max_records = 0;
    // [tile counts][line-feed positions 8 * (4 * records + 4)][per-record output lengths + offsets][scan block sums]
    return 256 + 8 * (size_t)(n_tiles_of(chunk_bytes) + 8) + 8 * (size_t)(4 * max_records + 8) + 16 * (size_t)(max_records + 8) +
           8 * (size_t)(n_scan_blocks(max_records) + 8) + 1024;
}

// Step 1: count the line feeds of the chunk.  d_info[0] = number of '\n' (read it after synchronising the
// stream: the host sizes the record arrays from it).  d_scratch keeps the tile counts for step 2.
int cah_fastq_count_lines_device(const uint8_t* d_buf, int64_t len, void* d_scratch, size_t scratch_bytes,
                                 int64_t* d_info, void* stream) {
    if (len < 0 || (len > 0 && !d_buf) || !d_scratch || !d_info) return cah_set_error_(CAH_EINVAL, "cah_fastq_count_lines_device: bad argument");
    if (scratch_bytes < cah_fastq_device_scratch_bytes(len, 0)) return cah_set_error_(CAH_EINVAL, "cah_fastq_count_lines_device: scratch too small");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_tiles = n_tiles_of(len);
    int64_t* tile_counts = (int64_t*)d_scratch;
    GPU_TRY(hipMemsetAsync(d_info, 0, 8 * sizeof(int64_t), s));
    if (n_tiles == 0) return CAH_OK;
    const int grid = (int)(n_tiles < 8 * cus() ? n_tiles : 8 * cus());
    hipLaunchKernelGGL(k_nl_count, dim3(grid), dim3(256), 0, s, d_buf, len, n_tiles, tile_counts);
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, s, tile_counts, n_tiles, d_info);
    GPU_TRY(hipGetLastError());
    return CAH_OK;
}

// Step 2: index n_records = n_lines / 4 records (n_newlines as counted by step 1; a last line without line feed
// counts as a line).  Writes d_rec6 (name_beg, name_end, seq_beg, seq_end, qual_beg, qual_end per record, the
// columns of cah_fastq_scan), d_seq_off / d_seq_len (the `offsets` + `lens` view of the reads inside the raw
// chunk that cah_match_batch takes) and d_info[1] = (first bad record + 1) << 8 | error code, or ~0 if the
// chunk is well formed (codes: 1 no '@', 2 no '+', 3 sequence and quality lengths differ).
int cah_fastq_index_device(const uint8_t* d_buf, int64_t len, int64_t n_newlines, int64_t n_records,
                           void* d_scratch, size_t scratch_bytes, int64_t* d_rec6, int64_t* d_seq_off,
                           int32_t* d_seq_len, int64_t* d_info, void* stream) {
    if (len < 0 || n_records < 0 || n_newlines < 0 || !d_scratch || !d_info) return cah_set_error_(CAH_EINVAL, "cah_fastq_index_device: bad argument");
    if (scratch_bytes < cah_fastq_device_scratch_bytes(len, n_records)) return cah_set_error_(CAH_EINVAL, "cah_fastq_index_device: scratch too small");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_tiles = n_tiles_of(len);
    int64_t* tile_base = (int64_t*)d_scratch;                                  // exclusive scan left by step 1
    int64_t* nl_pos = tile_base + n_tiles + 8;
    GPU_TRY(hipMemsetAsync(d_info + 1, 0xFF, sizeof(int64_t), s));            // "no bad record"
    if (n_records == 0 || n_tiles == 0) return CAH_OK;
    if (!d_rec6 || !d_seq_off || !d_seq_len) return cah_set_error_(CAH_EINVAL, "cah_fastq_index_device: output pointers are NULL");
    const int grid = (int)(n_tiles < 8 * cus() ? n_tiles : 8 * cus());
    hipLaunchKernelGGL(k_nl_write, dim3(grid), dim3(256), 0, s, d_buf, len, n_tiles, tile_base, nl_pos, 4 * n_records + 4);
    const int64_t rb = (n_records + 255) / 256;
    hipLaunchKernelGGL(k_records, dim3((unsigned)(rb < 8 * cus() ? rb : 8 * cus())), dim3(256), 0, s, d_buf, len, nl_pos, n_newlines,
                       n_records, d_rec6, d_seq_off, d_seq_len, (unsigned long long*)d_info);
    GPU_TRY(hipGetLastError());
    return CAH_OK;
}

// ... with --action none / retain / crop (one round of matching; cah_trim_decide_window_device is action trim)
int cah_trim_decide_action_device(const int32_t* d_out6, const uint8_t* d_status, const int32_t* d_best_adapter,
                                  const int32_t* d_win_beg, const int32_t* d_win_len, const int32_t* d_seq_len,
                                  int64_t n_reads, const uint8_t* d_adapter_kind, int32_t action, int32_t min_len,
                                  int32_t max_len, int32_t discard_trimmed, int32_t discard_untrimmed,
                                  int32_t intervals_only, int32_t* d_beg, int32_t* d_end, uint8_t* d_keep,
                                  uint64_t* d_counters, void* stream);

// Step 3b: kept interval and keep flag of every read from the match results (see k_trim_decide).  adapter_kind[a]:
// 0 = 3' adapter, 1 = 5' adapter, 2 = anywhere; min_len / max_len < 0: no limit.  d_counters: uint64[8], accumulated
// (NOT reset) -- reads, with adapters, bp in, bp out, too short, too long, invalid reads.
int cah_trim_decide_device(const int32_t* d_out6, const uint8_t* d_status, const int32_t* d_best_adapter,
                           const int32_t* d_seq_len, int64_t n_reads, const uint8_t* d_adapter_kind, int32_t min_len,
                           int32_t max_len, int32_t discard_trimmed, int32_t discard_untrimmed, int32_t* d_beg,
                           int32_t* d_end, uint8_t* d_keep, uint64_t* d_counters, void* stream) {
    if (n_reads < 0) return cah_set_error_(CAH_EINVAL, "cah_trim_decide_device: bad argument");
    if (n_reads == 0) return CAH_OK;
    if (!d_out6 || !d_status || !d_seq_len || !d_adapter_kind || !d_beg || !d_end || !d_keep || !d_counters)
        return cah_set_error_(CAH_EINVAL, "cah_trim_decide_device: NULL argument");
    const int64_t rb = (n_reads + 255) / 256;
    hipLaunchKernelGGL(k_trim_decide, dim3((unsigned)(rb < 4 * cus() ? rb : 4 * cus())), dim3(256), 0, (hipStream_t)stream, d_out6,
                       d_status, d_best_adapter, d_seq_len, (const int32_t*)nullptr, (const int32_t*)nullptr, n_reads,
                       d_adapter_kind, min_len, max_len, discard_trimmed, discard_untrimmed, d_beg, d_end, d_keep,
                       (unsigned long long*)d_counters);
    GPU_TRY(hipGetLastError());
    return CAH_OK;
}

// ... when modifiers ran in front of the adapter step (-u, --nextseq-trim, -q; reference cli.py:938-954): the matcher saw
// the window [d_win_beg[r], d_win_beg[r] + d_win_len[r]) of read r, whose full length is d_seq_len[r] (d_win_beg NULL:
// the windows start at 0).  d_beg / d_end are relative to the read; "bp in" counts the full reads.
// intervals_only != 0: modifiers follow BEHIND the adapter step too -- no filter is applied, "bp out" is not counted,
// cah_trim_filter_device finishes the job.
int cah_trim_decide_window_device(const int32_t* d_out6, const uint8_t* d_status, const int32_t* d_best_adapter,
                                  const int32_t* d_win_beg, const int32_t* d_win_len, const int32_t* d_seq_len,
                                  int64_t n_reads, const uint8_t* d_adapter_kind, int32_t min_len, int32_t max_len,
                                  int32_t discard_trimmed, int32_t discard_untrimmed, int32_t intervals_only,
                                  int32_t* d_beg, int32_t* d_end, uint8_t* d_keep, uint64_t* d_counters, void* stream) {
    return cah_trim_decide_action_device(d_out6, d_status, d_best_adapter, d_win_beg, d_win_len, d_seq_len, n_reads,
                                         d_adapter_kind, CAH_ACTION_TRIM, min_len, max_len, discard_trimmed, discard_untrimmed,
                                         intervals_only, d_beg, d_end, d_keep, d_counters, stream);
}

int cah_trim_decide_action_device(const int32_t* d_out6, const uint8_t* d_status, const int32_t* d_best_adapter,
                                  const int32_t* d_win_beg, const int32_t* d_win_len, const int32_t* d_seq_len,
                                  int64_t n_reads, const uint8_t* d_adapter_kind, int32_t action, int32_t min_len,
                                  int32_t max_len, int32_t discard_trimmed, int32_t discard_untrimmed,
                                  int32_t intervals_only, int32_t* d_beg, int32_t* d_end, uint8_t* d_keep,
                                  uint64_t* d_counters, void* stream) {
    if (action < CAH_ACTION_TRIM || action > CAH_ACTION_CROP) return cah_set_error_(CAH_EINVAL, "cah_trim_decide_action_device: unknown action");
    if (n_reads < 0) return cah_set_error_(CAH_EINVAL, "cah_trim_decide_window_device: bad argument");
    if (n_reads == 0) return CAH_OK;
    if (!d_out6 || !d_status || !d_win_len || !d_seq_len || !d_adapter_kind || !d_beg || !d_end || !d_keep || !d_counters)
        return cah_set_error_(CAH_EINVAL, "cah_trim_decide_window_device: NULL argument");
    if (intervals_only) { min_len = max_len = -1; discard_trimmed = discard_untrimmed = 0; }
    const int64_t rb = (n_reads + 255) / 256;
    hipLaunchKernelGGL(k_trim_decide, dim3((unsigned)(rb < 4 * cus() ? rb : 4 * cus())), dim3(256), 0, (hipStream_t)stream, d_out6,
                       d_status, d_best_adapter, d_win_len, d_win_beg, d_seq_len, n_reads, d_adapter_kind, min_len, max_len,
                       discard_trimmed, discard_untrimmed, d_beg, d_end, d_keep, (unsigned long long*)d_counters,
                       intervals_only ? 0 : 1, action);
    GPU_TRY(hipGetLastError());
    return CAH_OK;
}

// Step 3c: the filters, when modifiers ran BEHIND the adapter step too (--poly-a, -l) or --max-ee is given: d_beg /
// d_end are then final only after those, so cah_trim_decide*_device is called without limits and this decides what is
// written (see k_trim_filter).  d_ee: expected errors of the kept interval (cah_expected_errors_batch) or NULL;
// max_ee < 0: no limit.  d_counters as in cah_trim_decide_device: [3], [4], [5] and [7] (too many expected errors)
// are accumulated here.
int cah_trim_filter_device(const int32_t* d_beg, const int32_t* d_end, const uint8_t* d_status, const double* d_ee,
                           int64_t n_reads, int32_t min_len, int32_t max_len, double max_ee, int32_t discard_trimmed,
                           int32_t discard_untrimmed, uint8_t* d_keep, uint64_t* d_counters, void* stream) {
    if (n_reads < 0) return cah_set_error_(CAH_EINVAL, "cah_trim_filter_device: bad argument");
    if (n_reads == 0) return CAH_OK;
    if (!d_beg || !d_end || !d_status || !d_keep || !d_counters)
        return cah_set_error_(CAH_EINVAL, "cah_trim_filter_device: NULL argument");
    const int64_t rb = (n_reads + 255) / 256;
    hipLaunchKernelGGL(k_trim_filter, dim3((unsigned)(rb < 4 * cus() ? rb : 4 * cus())), dim3(256), 0, (hipStream_t)stream, d_beg,
                       d_end, d_status, d_ee, n_reads, min_len, max_len, max_ee, discard_trimmed, discard_untrimmed, d_keep,
                       (unsigned long long*)d_counters);
    GPU_TRY(hipGetLastError());
    return CAH_OK;
}

// Step 4: the trimmed records of a chunk, formatted on the device: record r (if d_keep is NULL or d_keep[r] != 0)
// as "@name\nSEQ[beg:end]\n+\nQUAL[beg:end]\n" at the exclusive-scan offset of its length (record order is kept).
// d_info[3] = total bytes written.  out_cap >= chunk length + 4 * n_records always suffices.
static int format_impl(const char* who, const uint8_t* d_buf, const int64_t* d_rec6, int64_t n_records, const int32_t* d_beg,
                       const int32_t* d_end, const uint8_t* d_keep, const uint8_t* d_flags, const char* suffix, int suffix_len,
                       void* d_scratch, size_t scratch_bytes, int64_t chunk_bytes, uint8_t* d_out, int64_t out_cap,
                       int64_t* d_info, void* stream) {
    char msg[96];
    auto fail = [&](const char* what) { snprintf(msg, sizeof msg, "%s: %s", who, what); return cah_set_error_(CAH_EINVAL, msg); };
    if (n_records < 0 || !d_scratch || !d_info) return fail("bad argument");
    if (suffix_len < 0 || suffix_len > CAH_MAX_NAME_SUFFIX || (suffix_len > 0 && !suffix)) return fail("suffix longer than CAH_MAX_NAME_SUFFIX");
    if (scratch_bytes < cah_fastq_device_scratch_bytes(chunk_bytes, n_records)) return fail("scratch too small");
    hipStream_t s = (hipStream_t)stream;
    GPU_TRY(hipMemsetAsync(d_info + 3, 0, sizeof(int64_t), s));
    if (n_records == 0) return CAH_OK;
    if (!d_buf || !d_rec6 || !d_beg || !d_end || !d_out) return fail("NULL argument");
    const bool with_suffix = d_flags && suffix_len > 0;
    int64_t* p = (int64_t*)d_scratch + n_tiles_of(chunk_bytes) + 8 + 4 * n_records + 8;
    int64_t* out_len = p;                        p += n_records + 8;
    int64_t* out_off = p;                        p += n_records + 8;
    int64_t* block_sums = p;
    const int64_t rb = (n_records + 255) / 256;
    const int64_t sb = n_scan_blocks(n_records);
    hipLaunchKernelGGL(k_out_len, dim3((unsigned)(rb < 8 * cus() ? rb : 8 * cus())), dim3(256), 0, s, d_rec6, n_records, d_beg, d_end,
                       d_keep, out_len, with_suffix ? d_flags : (const uint8_t*)nullptr, suffix_len);
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)sb), dim3(256), 0, s, out_len, n_records, block_sums);
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, s, block_sums, sb, d_info + 3);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)sb), dim3(256), 0, s, out_len, n_records, block_sums, out_off);
    const int64_t fb = (n_records + 3) / 4;
    NameSuffix sfx;
    sfx.len = with_suffix ? suffix_len : 0;
    for (int i = 0; i < CAH_MAX_NAME_SUFFIX; ++i) sfx.c[i] = i < sfx.len ? (uint8_t)suffix[i] : 0;
    const dim3 grid((unsigned)(fb < 16 * cus() ? fb : 16 * cus()));
    if (with_suffix)
        hipLaunchKernelGGL(k_format<true>, grid, dim3(256), 0, s, d_buf, d_rec6, n_records, d_beg, d_end, d_keep, out_off, d_out,
                           out_cap, d_flags, sfx);
    else
        hipLaunchKernelGGL(k_format<false>, grid, dim3(256), 0, s, d_buf, d_rec6, n_records, d_beg, d_end, d_keep, out_off, d_out,
                           out_cap, (const uint8_t*)nullptr, sfx);
    GPU_TRY(hipGetLastError());
    return CAH_OK;
}

int cah_fastq_format_device(const uint8_t* d_buf, const int64_t* d_rec6, int64_t n_records, const int32_t* d_beg,
                            const int32_t* d_end, const uint8_t* d_keep, void* d_scratch, size_t scratch_bytes,
                            int64_t chunk_bytes, uint8_t* d_out, int64_t out_cap, int64_t* d_info, void* stream) {
    return format_impl("cah_fastq_format_device", d_buf, d_rec6, n_records, d_beg, d_end, d_keep, nullptr, nullptr, 0, d_scratch,
                       scratch_bytes, chunk_bytes, d_out, out_cap, d_info, stream);
}

// ... with `suffix` (suffix_len <= CAH_MAX_NAME_SUFFIX bytes) behind the name of every record with d_flags[r] != 0: what
// ReverseComplementer does to the reads it turned around (reference modifiers.py:296-297).  out_cap >= chunk length +
// (4 + suffix_len) * n_records always suffices.
int cah_fastq_format_suffix_device(const uint8_t* d_buf, const int64_t* d_rec6, int64_t n_records, const int32_t* d_beg,
                                   const int32_t* d_end, const uint8_t* d_keep, const uint8_t* d_flags, const char* suffix,
                                   int32_t suffix_len, void* d_scratch, size_t scratch_bytes, int64_t chunk_bytes,
                                   uint8_t* d_out, int64_t out_cap, int64_t* d_info, void* stream) {
    return format_impl("cah_fastq_format_suffix_device", d_buf, d_rec6, n_records, d_beg, d_end, d_keep, d_flags, suffix, suffix_len,
                       d_scratch, scratch_bytes, chunk_bytes, d_out, out_cap, d_info, stream);
}

// --info-file rows of a chunk in HBM (k_info_len / scan / k_info_format): the lines of every record in record order, d_total[0] =
// bytes written.  d_out6 / d_status / d_best: the adapter step's results, round after round (rounds >= 1 arrays of n_records
// rows back to back: --times N, or the two parts of a linked adapter -- a round without a match gives no row); d_kinds: per adapter 0 = 3', 1 = 5',
// 2 = anywhere (what a match leaves of the read the next row shows; NULL: all 3'); d_final_beg / d_final_end: what is written
// of every read (the line of a read without a match shows that); d_names / d_name_off (int32[n_names + 1]): the adapters'
// names back to back, in plan order; d_is_rc NULL: no --revcomp, the last column stays empty; else "0" / "1", and the names of
// the reads with d_is_rc[r] != 0 carry `suffix`.
// out_cap >= rounds * (chunk length + n_records * (longest adapter name + suffix_len + 48)) always suffices.
int cah_info_format_device(const uint8_t* d_buf, const int64_t* d_rec6, int64_t n_records, const int32_t* d_out6,
                           const uint8_t* d_status, const int32_t* d_best, int32_t rounds, const uint8_t* d_kinds,
                           const int32_t* d_final_beg, const int32_t* d_final_end,
                           const uint8_t* d_names, const int32_t* d_name_off, int32_t n_names, const uint8_t* d_is_rc,
                           const char* suffix, int32_t suffix_len, void* d_scratch, size_t scratch_bytes, int64_t chunk_bytes,
                           uint8_t* d_out, int64_t out_cap, int64_t* d_total, void* stream) {
    if (n_records < 0 || !d_scratch || !d_total || n_names < 0 || rounds < 1) return cah_set_error_(CAH_EINVAL, "cah_info_format_device: bad argument");
    if (suffix_len < 0 || suffix_len > CAH_MAX_NAME_SUFFIX || (suffix_len > 0 && !suffix))
        return cah_set_error_(CAH_EINVAL, "cah_info_format_device: suffix longer than CAH_MAX_NAME_SUFFIX");
    if (scratch_bytes < cah_fastq_device_scratch_bytes(chunk_bytes, n_records)) return cah_set_error_(CAH_EINVAL, "cah_info_format_device: scratch too small");
    hipStream_t s = (hipStream_t)stream;
    GPU_TRY(hipMemsetAsync(d_total, 0, sizeof(int64_t), s));
    if (n_records == 0) return CAH_OK;
    if (!d_buf || !d_rec6 || !d_out6 || !d_status || !d_best || !d_final_beg || !d_final_end || !d_out || (n_names > 0 && (!d_names || !d_name_off)))
        return cah_set_error_(CAH_EINVAL, "cah_info_format_device: NULL argument");
    InfoArgs a;
    a.buf = d_buf; a.rec6 = d_rec6; a.n_records = n_records; a.out6 = d_out6; a.status = d_status; a.best = d_best;
    a.rounds = rounds; a.kinds = d_kinds;
    a.final_beg = d_final_beg; a.final_end = d_final_end; a.names = d_names; a.name_off = d_name_off; a.n_names = n_names;
    a.is_rc = d_is_rc;
    a.sfx.len = d_is_rc ? suffix_len : 0;
    for (int i = 0; i < CAH_MAX_NAME_SUFFIX; ++i) a.sfx.c[i] = i < a.sfx.len ? (uint8_t)suffix[i] : 0;
    int64_t* p = (int64_t*)d_scratch + n_tiles_of(chunk_bytes) + 8 + 4 * n_records + 8;      // (the formatter's arrays: free again)
    int64_t* out_len = p;                        p += n_records + 8;
    int64_t* out_off = p;                        p += n_records + 8;
    int64_t* block_sums = p;
    const int64_t rb = (n_records + 255) / 256;
    const int64_t sb = n_scan_blocks(n_records);
    hipLaunchKernelGGL(k_info_len, dim3((unsigned)(rb < 8 * cus() ? rb : 8 * cus())), dim3(256), 0, s, a, out_len);
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)sb), dim3(256), 0, s, out_len, n_records, block_sums);
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, s, block_sums, sb, d_total);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)sb), dim3(256), 0, s, out_len, n_records, block_sums, out_off);
    const int64_t fb = (n_records + 3) / 4;
    hipLaunchKernelGGL(k_info_format, dim3((unsigned)(fb < 16 * cus() ? fb : 16 * cus())), dim3(256), 0, s, a, out_off, d_out, out_cap);
    GPU_TRY(hipGetLastError());
    return CAH_OK;
}

// ReverseComplementer's chosen orientation, in place (k_revcomp_in_place): records with d_flags[r] != 0 are turned around
// inside their window (d_win_beg NULL: the windows start at 0), sequence and qualities
int cah_revcomp_in_place_device(uint8_t* d_buf, const int64_t* d_rec6, int64_t n_records, const int32_t* d_win_beg,
                                const int32_t* d_win_len, const uint8_t* d_flags, void* stream) {
    if (n_records < 0) return cah_set_error_(CAH_EINVAL, "cah_revcomp_in_place_device: bad argument");
    if (n_records == 0) return CAH_OK;
    if (!d_buf || !d_rec6 || !d_win_len || !d_flags) return cah_set_error_(CAH_EINVAL, "cah_revcomp_in_place_device: NULL argument");
    const int64_t fb = (n_records + 3) / 4;
    hipLaunchKernelGGL(k_revcomp_in_place, dim3((unsigned)(fb < 16 * cus() ? fb : 16 * cus())), dim3(256), 0, (hipStream_t)stream, d_buf,
                       d_rec6, n_records, d_win_beg, d_win_len, d_flags);
    GPU_TRY(hipGetLastError());
    return CAH_OK;
}

// AdapterCutter's marking actions on a chunk in HBM, in place (k_mark_reads): mode 1 = --action=mask, 2 = --action=lowercase
int cah_mark_reads_device(uint8_t* d_buf, const int64_t* d_rec6, int64_t n_records, const int32_t* d_beg, const int32_t* d_end,
                          const int32_t* d_mark_beg, const int32_t* d_mark_end, int mode, void* stream) {
    if (n_records < 0 || mode < 1 || mode > 2) return cah_set_error_(CAH_EINVAL, "cah_mark_reads_device: bad argument");
    if (n_records == 0) return CAH_OK;
    if (!d_buf || !d_rec6 || !d_beg || !d_end || !d_mark_beg || !d_mark_end) return cah_set_error_(CAH_EINVAL, "cah_mark_reads_device: NULL argument");
    const int64_t fb = (n_records + 3) / 4;
    hipLaunchKernelGGL(k_mark_reads, dim3((unsigned)(fb < 16 * cus() ? fb : 16 * cus())), dim3(256), 0, (hipStream_t)stream, d_buf, d_rec6,
                       n_records, d_beg, d_end, d_mark_beg, d_mark_end, mode);
    GPU_TRY(hipGetLastError());
    return CAH_OK;
}

}  // extern "C"
