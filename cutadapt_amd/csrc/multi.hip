// multi.hip -- the fused multi-adapter prefilter (see CahMultiHeader in cah_device.h) and the decode of the
// per-read best keys.  Replaces, for a plan of many 3' adapters, one k_filter pass PER ADAPTER over the
// whole batch (reference: MultipleAdapters.match_to calls every adapter's kmers_present,
// src/cutadapt/adapters.py:1265-1286, _kmer_finder.pyx:170-213) by ONE pass that yields exactly the
// (read, adapter) pairs whose kmers_present is true.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cah_device.h"
#include "kernels.h"
#include "dev_common.h"

#ifndef MF_TILE
#define MF_TILE 256                // reads per workgroup tile (one 64-read block per wave)
#endif
#ifndef MF_STAGE
#define MF_STAGE 2048              // pairs staged in LDS per tile, 4 bytes each: tile-relative read | adapter | key
                                   // (more go straight to HBM, unsorted)
#endif

// One read per lane, one pass, 16 characters per global load.  Per character: 2-bit base code (LDS byte
// table; anything but ACGT/acgt is invalid and breaks every k-mer, as in KmerFinder without wildcards:
// _match_tables.py:81-98), a rolling 64-bit code of the last 32 bases, the length of the current run of
// valid bases, and one bitmap probe per k-mer class that can end here.  A probe hit is resolved through
// the directory; every entry of the key is verified on all its characters and on its search window
// (whole read, or "starts within the last L characters": _kmer_finder.pyx:188-204) before the pair
// (read, adapter) is emitted -- once per adapter (per-lane bitset in LDS).  key = first-hit position >> 2:
// every k-mer class is probed at every position, so no whole-read k-mer of that adapter ended earlier
// (the property column skipping needs).
__global__ __launch_bounds__(256) void k_multi_filter(MultiFilterArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const CahMultiHeader* hd = a.hdr;
    const uint32_t bm_words = hd->bm_words;
    uint32_t* s_bm = smem;                                   // [bm_words]
    uint32_t* s_stage = s_bm + ((bm_words + 3) & ~3u);       // [MF_STAGE] pairs: read - tile_base << 16 | adapter << 8 | key
    uint32_t* s_seen = s_stage + MF_STAGE;                   // [256 * 4] per-lane "adapter already emitted"
    uint32_t* s_hist = s_seen + 256 * 4;                     // [256]
    uint32_t* s_cursor = s_hist + 256;                       // [256]
    uint32_t* s_b2 = s_cursor + 256;                         // [32] = 128 bytes: base code of every ASCII character
    uint32_t* s_misc = s_b2 + 32;                            // [8]: npairs, tile lo/hi, qbase lo/hi
    for (uint32_t i = threadIdx.x; i < bm_words; i += blockDim.x) s_bm[i] = a.bitmap[i];
    if (threadIdx.x < 128) {
        const int c = threadIdx.x;
        const int u = c & 0xDF;
        const uint8_t code = (c >= 64 && u == 'A') ? 0 : (c >= 64 && u == 'C') ? 1 : (c >= 64 && u == 'G') ? 2 : (c >= 64 && u == 'T') ? 3 : 4;
        reinterpret_cast<uint8_t*>(s_b2)[c] = code;
    }
    // wave-uniform plan constants
    int present[8], everywhere[8], lmax[8];
    uint32_t bm_off[9], dir_off[9];
#pragma unroll
    for (int q = 1; q <= 8; ++q) { bm_off[q] = hd->bm_off[q]; dir_off[q] = hd->dir_off[q]; }
#pragma unroll
    for (int q = 1; q <= 7; ++q) { present[q] = hd->class_present[q]; everywhere[q] = hd->class_everywhere[q]; lmax[q] = hd->class_lmax[q]; }
    const int present8 = hd->class_present[8];
    int lmax_all = 0;
    bool short_everywhere = false;
#pragma unroll
    for (int q = 1; q <= 7; ++q) {
        if (present[q]) { lmax_all = max(lmax_all, lmax[q]); short_everywhere = short_everywhere || everywhere[q]; }
    }
    const int lane = wave_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform, and said so
    const uint8_t* b2 = reinterpret_cast<const uint8_t*>(s_b2);
    const int64_t last_read = a.first_read + a.n_reads;

    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long t = atomicAdd(a.work_counter, (unsigned long long)MF_TILE);
            s_misc[1] = (uint32_t)t; s_misc[2] = (uint32_t)(t >> 32);
            s_misc[0] = 0;
        }
        for (int i = threadIdx.x; i < 256; i += blockDim.x) { s_hist[i] = 0; s_cursor[i] = 0; }
        __syncthreads();
        const int64_t tile_base = a.first_read + (int64_t)(((unsigned long long)s_misc[2] << 32) | s_misc[1]);
        if (tile_base >= last_read) break;

        for (int sub = wave; sub < MF_TILE / WAVE; sub += 4) {
            const int64_t base = tile_base + (int64_t)sub * WAVE;
            if (base >= last_read) break;
            const int64_t r = base + lane;
            const bool valid = r < last_read;
            int64_t off = 0, n64 = 0;
            if (valid) read_extent(a.offsets, a.lens, a.uniform_first, a.uniform_len, r, off, n64);
            bool too_long = false;
            if (n64 > a.max_read_len) { too_long = true; n64 = 0; }
            const int n = (int)n64;
            const uint8_t* q = a.seqs + off;
            int n_max = n, n_min = valid ? n : 0x7fffffff;
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) {
                n_max = max(n_max, __shfl_xor(n_max, d, WAVE));
                n_min = min(n_min, __shfl_xor(n_min, d, WAVE));
            }
            n_max = __builtin_amdgcn_readfirstlane(n_max);
            n_min = __builtin_amdgcn_readfirstlane(n_min);
            // first position at which a tail k-mer of any short class can end in some lane of the wave
            const int tail_from = n_min - lmax_all;
            uint32_t* seen_bits = s_seen + threadIdx.x * 4;
            seen_bits[0] = seen_bits[1] = seen_bits[2] = seen_bits[3] = 0;

            unsigned r_lo = 0, r_hi = 0;        // rolling code of the last 32 bases, newest in the low bits
            int run = 0;                        // consecutive valid bases ending at the current character
            unsigned seen = 0;
            Chunk cur = load_chunk(q, 0, n, valid ? n : 0);
            for (int pos = 0; pos < n_max; pos += 16) {
                const Chunk nxt = load_chunk(q, pos + 16, n, valid ? n : 0);
                seen |= cur.w[0] | cur.w[1] | cur.w[2] | cur.w[3];
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int p = pos + t;                                   // wave-uniform
                    if (p >= n_max) break;
                    const unsigned c = chunk_byte(cur, t);                   // NUL beyond the read end: invalid base
                    const unsigned code = b2[c & 127];
                    r_hi = __builtin_amdgcn_alignbit(r_hi, r_lo, 30);
                    r_lo = (r_lo << 2) | (code & 3u);
                    run = code < 4u ? run + 1 : 0;
                    unsigned hits = 0;                                       // bit q: class q has the key of this position
                    if (present8) {
                        const unsigned k8 = r_lo & 0xFFFFu;
                        const unsigned w = s_bm[bm_off[8] + (k8 >> 5)];
                        hits |= (run >= 8 && ((w >> (k8 & 31)) & 1u)) ? (1u << 8) : 0u;
                    }
                    if (short_everywhere || p >= tail_from)                 // wave-uniform: most positions skip all short classes
#pragma unroll
                    for (int cq = 1; cq <= 7; ++cq) {
                        if (!present[cq]) continue;                          // wave-uniform
                        // a k-mer of this class ending here starts at p - cq + 1; tail k-mers must start within the
                        // last lmax characters (per entry, checked again below), whole-read ones anywhere
                        const bool in_win = everywhere[cq] || (p - cq + 1 >= n - lmax[cq]);
                        if (!__any(in_win && run >= cq)) continue;
                        const unsigned kq = r_lo & ((1u << (2 * cq)) - 1u);
                        const unsigned w = s_bm[bm_off[cq] + (kq >> 5)];
                        hits |= (in_win && run >= cq && ((w >> (kq & 31)) & 1u)) ? (1u << cq) : 0u;
                    }
                    if (__any(hits != 0u)) {
                        while (hits) {
                            const int cq = __ffs((int)hits) - 1;
                            hits &= hits - 1u;
                            const unsigned key_idx = cq == 8 ? (r_lo & 0xFFFFu) : (r_lo & ((1u << (2 * cq)) - 1u));
                            const CahMultiDir d = a.dir[dir_off[cq] + key_idx];
                            for (unsigned e = d.begin; e < d.begin + d.count; ++e) {
                                const CahMultiEntry en = a.entries[e];
                                const int kq = en.q;
                                const unsigned long long rr = ((unsigned long long)r_hi << 32) | r_lo;
                                const unsigned long long msk = kq >= 32 ? ~0ull : ((1ull << (2 * kq)) - 1ull);
                                const bool ok = run >= kq && (rr & msk) == en.code &&
                                                (en.window == 0 || p - kq + 1 >= n - (int)en.window);
                                if (!ok) continue;
                                const unsigned ad = en.adapter;
                                const unsigned wbit = 1u << (ad & 31);
                                const unsigned word = seen_bits[ad >> 5];
                                if (word & wbit) continue;
                                seen_bits[ad >> 5] = word | wbit;
                                const unsigned key = min(p >> CAH_KEY_SHIFT, CAH_QUEUE_BINS - 1);
                                const unsigned slot = atomicAdd(&s_misc[0], 1u);
                                if (slot < MF_STAGE) {
                                    s_stage[slot] = ((unsigned)(r - tile_base) << 16) | (ad << 8) | key;
                                    atomicAdd(&s_hist[key], 1u);
                                } else {
                                    const unsigned long long g = atomicAdd(a.pair_count, 1ull);
                                    if ((int64_t)g < a.pair_cap) a.pairs[g] = ((uint64_t)(unsigned)r << 32) | (ad << 8) | key;
                                }
                            }
                        }
                    }
                }
                cur = nxt;
            }
            if (valid && ((seen & 0x80808080u) != 0 || too_long)) a.status[r] = 2;
        }

        // flush the tile: counting sort by key in LDS, one global atomic for the run
        __syncthreads();
        const unsigned count = min(s_misc[0], (unsigned)MF_STAGE);
        if (threadIdx.x == 0) {
            unsigned run_sum = 0;
            for (int b = 0; b < 256; ++b) { const unsigned c = s_hist[b]; s_hist[b] = run_sum; run_sum += c; }
            const unsigned long long g = count ? atomicAdd(a.pair_count, (unsigned long long)count) : 0ull;
            s_misc[3] = (uint32_t)g; s_misc[4] = (uint32_t)(g >> 32);
        }
        __syncthreads();
        const unsigned long long qbase = ((unsigned long long)s_misc[4] << 32) | s_misc[3];
        for (unsigned e = threadIdx.x; e < count; e += blockDim.x) {
            const unsigned st = s_stage[e];
            const unsigned key = st & 0xFFu;
            const unsigned dst = s_hist[key] + atomicAdd(&s_cursor[key], 1u);
            const uint64_t read = (uint64_t)(tile_base + (int64_t)(st >> 16));
            if ((int64_t)(qbase + dst) < a.pair_cap) a.pairs[qbase + dst] = (read << 32) | (st & 0xFFFFu);
        }
    }
}

// best_key[r] != 0  ->  the winning match of read r (kernels.h: pack_best).  The kernel writes EVERY read's result row,
// best adapter and status (zeros / -1 / 0 where there is no match; status 2 = invalid read stays), so the fused path
// needs no clearing pass over the 24 B rows in front of it.  A block stages the rows of 256 reads in LDS and stores them
// as whole 16-byte words: a lane-per-read store of six ints touches 24 cache lines per instruction.
__global__ __launch_bounds__(256) void k_multi_decode(const unsigned long long* best_key, int64_t n_reads, int32_t* out6,
                                                      uint8_t* status, int32_t* best_adapter, const unsigned long long* err) {
    __shared__ __attribute__((aligned(16))) int32_t s_rows[256 * 6];
    const bool aligned = ((unsigned long long)out6 & 15ull) == 0ull;
    // deferred error check (cah_set_deferred_errors): the error word of the streaming kernels is looked at HERE instead
    // of by the host behind a synchronisation -- a batch whose kernels flagged a broken invariant gets status
    // CAH_STATUS_INTERNAL in every row (the rows are void)
    const bool broken = err != nullptr && *err != 0ull;
    for (int64_t base = (int64_t)blockIdx.x * 256; base < n_reads; base += (int64_t)gridDim.x * 256) {
        const int64_t r = base + threadIdx.x;
        int32_t o[6] = {0, 0, 0, 0, 0, 0};
        if (r < n_reads) {
            const unsigned long long k = broken ? 0ull : best_key[r];
            const bool invalid = status[r] == 2;
            const bool hit = k != 0ull && !invalid;
            if (hit) {
                const int rel = (int)(k & 0xFFu), qstart = (int)((k >> 8) & 0xFFFFFu);
                o[1] = (int)((k >> 28) & 0x7Fu); o[2] = qstart; o[3] = qstart + rel;
                o[4] = (int)((k >> 54) & 0xFFu) - 128; o[5] = 127 - (int)((k >> 47) & 0x7Fu);
            }
            if (!invalid) status[r] = hit ? 1 : 0;
            if (broken) status[r] = 255;                                // CAH_STATUS_INTERNAL
            if (best_adapter) best_adapter[r] = hit ? 4095 - (int)((k >> 35) & 0xFFFu) : -1;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) s_rows[threadIdx.x * 6 + i] = o[i];
        __syncthreads();
        const int64_t left = n_reads - base;
        const int ints = (int)(left < 256 ? left : 256) * 6;
        int32_t* const dst = out6 + base * 6;                           // (base * 24 bytes: a multiple of 16)
        if (aligned) {
            for (int i = threadIdx.x * 4; i < ints; i += 1024) {
                if (i + 4 <= ints) *reinterpret_cast<int4*>(dst + i) = *reinterpret_cast<const int4*>(s_rows + i);
                else for (int q = i; q < ints; ++q) dst[q] = s_rows[q];
            }
        } else {
            for (int i = threadIdx.x; i < ints; i += 256) dst[i] = s_rows[i];
        }
        __syncthreads();
    }
}

hipError_t launch_multi_filter(const MultiFilterArgs& a, const CahMultiHeader& host_hdr, int n_cus, hipStream_t s) {
    const size_t lds = sizeof(uint32_t) * ((size_t)((host_hdr.bm_words + 3) & ~3u) + MF_STAGE + 256 * 4 + 256 + 256 + 32 + 8);
    int64_t need = (a.n_reads + MF_TILE - 1) / MF_TILE;
    if (need < 1) need = 1;
    const int64_t cap = (int64_t)8 * n_cus;
    hipLaunchKernelGGL(k_multi_filter, dim3((unsigned)(need < cap ? need : cap)), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_multi_decode(const unsigned long long* best_key, int64_t n_reads, int32_t* out6, uint8_t* status,
                               int32_t* best_adapter, int n_cus, hipStream_t s, const unsigned long long* err) {
    int64_t need = (n_reads + 255) / 256;
    if (need < 1) need = 1;
    const int64_t cap = (int64_t)8 * n_cus;
    hipLaunchKernelGGL(k_multi_decode, dim3((unsigned)(need < cap ? need : cap)), dim3(256), 0, s, best_key, n_reads, out6,
                       status, best_adapter, err);
    return hipGetLastError();
}
