// stream2.hip -- k_filter_stream2: KmerFinder.kmers_present (reference _kmer_finder.pyx:170-257) for batches of
// equally long short reads (the sequencer's output; every BASELINE config), round 3's form of the headline kernel.
//
// What changed against k_filter_stream (kernels.hip), and why:
//   * occupancy.  A wave's LDS slot held its whole piece (64 reads x 176 B = 11 KB: 12 waves per CU, 3 per SIMD).
//     Here the slot holds HALF of every read -- five 16-byte units, rows of 80 bytes -- and is filled twice per
//     piece: units 0..4 of all 64 reads, then (when the wave has matched those 80 characters) units 5..9.  The
//     copy registers are loaded accordingly: loads 0..4 fetch the first halves, loads 5..9 the second halves, all
//     ten issued together (each cache line is requested by two neighbouring instructions of one wave and crosses
//     the memory system once).  16 waves per CU (one block of 1024 threads, 4 waves per SIMD, <= 128 VGPRs).
//     With five units per half-row the LDS image of a half is simply unit-contiguous (unit u at byte 16 u): the
//     store offsets are immediates, and a row stride of 5 units (odd) keeps every lane group of a ds_read_b128
//     on distinct banks.
//   * instructions.  The tail k-mers are T-words (stream2.h): packed like lead k-mers, four characters per step,
//     found-gated by where a k-mer ends instead of start-gated per character (3 instructions per word and
//     character before, 1 now).  The copy plan costs four instructions per unit (SGPR base + 32-bit lane offset),
//     the found test of a group is one instruction per word.
// Same outputs, queue entries and keys as k_filter_lean / k_filter_stream (tests/test_gpu_stream.py compares them,
// tests/test_stream2_model.py fuzzes the word machinery of stream2.h against the oracle on the CPU).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "cah_device.h"
#include "kernels.h"
#include "dev_common.h"
#include "filter_common.h"
#include "stream2.h"

#define S2_WAVES 16                // waves per block = per CU
#define S2_TILE 8192               // reads per block tile (survivor staging: 3 B each)
#define S2_HALF 5                  // units per half-row
#define S2_ROW (S2_HALF * 16)      // bytes of a slot row
#define S2_MAX_LEN (2 * S2_HALF * 16)
#define S2_SEG (2 * S2_HALF)       // units of a segment: what the slot takes in two fills
#define S2_LONG_MAX_LEN 640        // LONG form: reads of up to four segments
#define S2_RV_BACK 1184            // RV form: the copy resource starts this far in front of a piece (>= S2_MAX_LEN + 63 * 15 + 1)

__host__ __device__ constexpr int s2_pow2(int n) { return n <= 1 ? 1 : (n <= 2 ? 2 : 4); }
__host__ __device__ constexpr int s2_log2(int p) { return p == 1 ? 0 : (p == 2 ? 1 : 2); }

template <int NL, int NT> struct S2Layout {
    static constexpr int NLP = s2_pow2(NL), NTP = s2_pow2(NT);
    static constexpr int LEAD_SHIFT = 2 + s2_log2(NLP), TAIL_SHIFT = 2 + s2_log2(NTP);
    static constexpr int LEAD_TABLE = CAH_TABLE_CHARS * NLP * 4;     // bytes of one of the four lead tables
    static constexpr int TAIL_TABLE = CAH_TABLE_CHARS * NTP * 4;
};

typedef unsigned int s2_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int s2_u32x4 __attribute__((ext_vector_type(4)));

// the first N words of a table entry of NP words (one LDS read)
template <int N, int NP>
__device__ __forceinline__ void s2_read_entry(uint32_t (&out)[N > 0 ? N : 1], const unsigned char* p) {
    if constexpr (N == 0) {
        (void)p; (void)out;
    } else if constexpr (NP == 1 || N == 1) {
        out[0] = *reinterpret_cast<const uint32_t*>(p);
    } else if constexpr (NP == 2 || N == 2) {
        const s2_u32x2 v = *reinterpret_cast<const s2_u32x2*>(p);
        out[0] = v.x; out[1] = v.y;
    } else {
        const s2_u32x4 v = *reinterpret_cast<const s2_u32x4*>(p);
        out[0] = v.x; out[1] = v.y; out[2] = v.z;
        if constexpr (N > 3) out[3] = v.w;
    }
}

template <int NL, int NT>
struct S2Words {
    uint32_t l_init4[NL > 0 ? NL : 1], l_found[NL > 0 ? NL : 1], t_init4[NT > 0 ? NT : 1];
    const unsigned char* lead;           // LDS: T0 | T1 | T2 | T3 of the lead words
    const unsigned char* tail;           // LDS: the same of the T-words
    const unsigned char* found;          // LDS: tw_found, one entry of NTP words per dist
};

// ---- the instructions of a group, written out (inline assembly: the compiler re-associates the AND chains "to shorten
// the critical path" at 5-6 instructions per word where 4 do, picks v_and_or / v_or3 (4 issue cycles) where v_bitop3
// (2) does, and a shift + mask where one SDWA instruction extracts a byte).  Issue cost on gfx950 with two or more
// waves per SIMD (profiles/r03/valu_ubench.txt): v_and / v_or / v_add / v_mov / v_bitop3 2 cycles; shifts, SDWA forms,
// v_lshl_or, v_and_or, v_or3, v_cmp, v_cndmask, v_perm, 24-bit multiplies 4. ----
// entry offsets ("byte << SHIFT") of the four characters of a dword; shv: a register holding SHIFT
__device__ __forceinline__ void s2_addr4(unsigned (&ad)[4], const unsigned w, const unsigned shv) {
    asm("v_lshlrev_b32_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_lshlrev_b32_sdwa %1, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
        "v_lshlrev_b32_sdwa %2, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
        "v_lshlrev_b32_sdwa %3, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3"
        : "=&v"(ad[0]), "=&v"(ad[1]), "=&v"(ad[2]), "=&v"(ad[3]) : "v"(shv), "v"(w));
}
// a word over four characters: R = ((R << 4) | S3) & m0 & m1 & m2 & m3;  f |= R & found  (v_lshl_or + 3 v_bitop3; FIRST:
// f = R & found).  The bitop3 builtin keeps the compiler from re-associating the chain.
template <bool FIRST>
__device__ __forceinline__ void s2_word_step(uint32_t& R, uint32_t& f, const uint32_t init4, const uint32_t found,
                                             const uint32_t m0, const uint32_t m1, const uint32_t m2, const uint32_t m3) {
#ifdef S2_DBG_NOSTEP
    R = R | (m0 & m1 & m2 & m3 & 1u & init4);                                 // developer build: (nearly) no word steps
#else
    R = (R << 4) | init4;
    R = __builtin_amdgcn_bitop3_b32(R, m0, m1, 0x80);
    R = __builtin_amdgcn_bitop3_b32(R, m2, m3, 0x80);
#endif
    if constexpr (FIRST) f = R & found;
    else f = __builtin_amdgcn_bitop3_b32(R, found, f, 0xea);
}
__device__ __forceinline__ uint32_t s2_or3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xfe); }
// what the lanes of a wave know about their reads' first k-mer: `live` -- the lane is still looking (a per-lane
// boolean: the compiler keeps it as a wave-wide mask in SGPRs, "is any lane still looking" is a scalar compare),
// `group` -- the 4-character group of the first hit
struct S2Hits {
    bool live;
    int group;
};
__device__ __forceinline__ bool s2_any(const bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// f[g] != 0 in a lane: a k-mer that counts ended inside group g of the chunk whose first group is g0 (wave-uniform).
// One test per chunk; the group is sorted out only when some lane that is still looking has a hit.
__device__ __forceinline__ void s2_note(S2Hits& h, const uint32_t (&f)[4], const int g0) {
    const bool any = (s2_or3(f[0], f[1], f[2]) | f[3]) != 0;
    const bool nw = any && h.live;
    if (s2_any(nw)) {
        int g = f[2] != 0 ? g0 + 2 : g0 + 3;
        g = f[1] != 0 ? g0 + 1 : g;
        g = f[0] != 0 ? g0 : g;
        h.group = nw ? g : h.group;
        h.live = h.live && !any;
    }
}

// The lead masks of two 4-character groups: while one group is matched the next one's four table entries are on
// their way (across chunk borders too).  (A whole chunk of lookahead -- 16 entries in flight, 32 registers -- was
// measured and did not pay: the waves do not wait for the LDS, and the registers are needed elsewhere.)
template <int NL> struct S2Masks { uint32_t m[2][4][NL > 0 ? NL : 1]; };       // two groups: the one at work, the one requested

template <int NL, int NT>
__device__ __forceinline__ void s2_lead_loads(const S2Words<NL, NT>& K, S2Masks<NL>& M, const int g, const unsigned w,
                                              const unsigned shv) {
    typedef S2Layout<NL, NT> LY;
    if constexpr (NL > 0) {
        unsigned la[4];
        s2_addr4(la, w, shv);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#ifdef S2_DBG_NOLDS
            for (int l = 0; l < NL; ++l) M.m[g][i][l] = la[i] + l;      // developer build: no table reads
#else
            s2_read_entry<NL, LY::NLP>(M.m[g][i], K.lead + (3 - i) * LY::LEAD_TABLE + la[i]);
#endif
        }
    }
}

// One chunk of 16 characters at positions pos .. pos+15 (characters past the read's end are NUL, which no k-mer
// character matches: groups past the end change nothing): the lead words and, when `tails` (wave-uniform), the T-words
// advance over its four groups.  M holds the lead masks of the chunk's first group (requested while the previous chunk
// was matched; `nw`: the next chunk, zeros behind the last).  The T-word masks of a group are requested in front of
// the group's lead steps.  ONE body for every chunk: the mask registers stay where they are from chunk to chunk.
template <int NL, int NT>
__device__ __forceinline__ void s2_chunk(const S2Words<NL, NT>& K, uint32_t (&RL)[NL > 0 ? NL : 1],
                                         uint32_t (&RT)[NT > 0 ? NT : 1], S2Masks<NL>& M, const s2_u32x4 cw,
                                         const s2_u32x4 nw, const int pos, const int n, const bool tails, S2Hits& hits,
                                         const unsigned shv, const unsigned shv_tail) {
    typedef S2Layout<NL, NT> LY;
    const unsigned w[4] = {cw.x, cw.y, cw.z, cw.w};
    const unsigned wn[4] = {nw.x, nw.y, nw.z, nw.w};
    constexpr int NTm = NT > 0 ? NT : 1;
    // the found masks of the chunk's groups sit 4 entries apart, group 3's lowest: one address register per chunk
    const unsigned char* const fm3 = K.found + s2_found_index(n, pos + 15) * (LY::NTP * 4);
    uint32_t f[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint32_t tm[4][NTm], fm[NTm];
        if constexpr (NT > 0) {
            if (tails) {
                unsigned ta[4];
                s2_addr4(ta, w[g], shv_tail);
#pragma unroll
                for (int i = 0; i < 4; ++i) s2_read_entry<NT, LY::NTP>(tm[i], K.tail + (3 - i) * LY::TAIL_TABLE + ta[i]);
                s2_read_entry<NT, LY::NTP>(fm, fm3 + (3 - g) * 4 * (LY::NTP * 4));
            }
        }
        // the next group's lead masks (the next chunk's first group behind this chunk's last) go out first
        s2_lead_loads<NL, NT>(K, M, (g + 1) & 1, g < 3 ? w[g + 1] : wn[0], shv);
        if constexpr (NL > 0) {
            s2_word_step<true>(RL[0], f[g], K.l_init4[0], K.l_found[0], M.m[g & 1][0][0], M.m[g & 1][1][0], M.m[g & 1][2][0], M.m[g & 1][3][0]);
#pragma unroll
            for (int l = 1; l < NL; ++l)
                s2_word_step<false>(RL[l], f[g], K.l_init4[l], K.l_found[l], M.m[g & 1][0][l], M.m[g & 1][1][l], M.m[g & 1][2][l], M.m[g & 1][3][l]);
        }
        if constexpr (NT > 0) {
            if (tails) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (NL == 0 && t == 0) s2_word_step<true>(RT[t], f[g], K.t_init4[t], fm[t], tm[0][t], tm[1][t], tm[2][t], tm[3][t]);
                    else s2_word_step<false>(RT[t], f[g], K.t_init4[t], fm[t], tm[0][t], tm[1][t], tm[2][t], tm[3][t]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    s2_note(hits, f, pos >> 2);
}

// what lean_emit / flush_tile_queue need of the kernel's arguments (fetched from the kernarg segment where they are
// used: as plain kernel arguments they would sit in SGPRs through the matching loops, which have none to spare)
struct S2Out {
    uint8_t* present;
    uint8_t* status;
    int32_t* queue;
    unsigned long long* queue_count;
    uint8_t* queue_keys;
};
// Zero the result rows (24 bytes each) of the `cnt` (<= 64, wave-uniform) consecutive reads from `base` on and set their
// best_adapter entries to -1 (clear_rows of filter_common.h with buffer stores: scalar base, one offset register --
// the 64-bit per-lane addresses of plain stores cost this kernel registers it does not have)
// (a pointer the caller knows to be wave-uniform, said so: buffer resources built from anything the compiler takes
// for per-lane data are looped over value by value -- a "waterfall" of ten instructions around every access)
__device__ __forceinline__ void* s2_uniform_ptr(const void* q) {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ void s2_clear_rows(int32_t* out6, int32_t* best, const int64_t base, int cnt, const int lane) {
    cnt = __builtin_amdgcn_readfirstlane(cnt);
    if (best) {
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(s2_uniform_ptr(best + base), 0, cnt * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(0xFFFFFFFFu, rb, lane * 4, 0, 0);          // lanes >= cnt: out of range, dropped
    }
    int32_t* const o = (int32_t*)s2_uniform_ptr(out6 + base * 6);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)o, 0, cnt * 24, 0x00020000);
    if ((reinterpret_cast<uintptr_t>(o) & 7u) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            __builtin_amdgcn_raw_buffer_store_b64((s2_u32x2)(0u), ro, lane * 8 + k * (WAVE * 8), 0, 0);
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k)
            __builtin_amdgcn_raw_buffer_store_b32(0u, ro, lane * 4 + k * (WAVE * 4), 0, 0);
    }
}

#ifdef S2_TRACE
// developer build only (-DS2_TRACE): s_memtime stamps of wave 0 of block 0 at the stations of its first pieces
#define S2_TRACE_PIECES 64
#define S2_TRACE_STATIONS 12
__device__ unsigned long long g_s2_trace[S2_TRACE_PIECES * S2_TRACE_STATIONS];
#define S2_STAMP(st) do { if (blockIdx.x == 7 && wave == 5 && it < S2_TRACE_PIECES) { \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_s2_trace[it * S2_TRACE_STATIONS + (st)] = t_; } } while (0)
__device__ unsigned long long g_s2_trace_tile[16 * 8];
#define S2_TSTAMP(st) do { if (false) { \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_s2_trace_tile[kt * 8 + (st)] = t_; } } while (0)
extern "C" int cah_debug_s2_trace_tile(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s2_trace_tile), sizeof(g_s2_trace_tile)) == hipSuccess ? 0 : 1;
}
extern "C" int cah_debug_s2_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s2_trace), sizeof(g_s2_trace)) == hipSuccess ? 0 : 1;
}
#else
#define S2_STAMP(st) do { } while (0)
#define S2_TSTAMP(st) do { } while (0)
#endif
// The survivors of a tile, staged in LDS with their keys, leave as one key-ordered run -- flush_tile_queue
// (filter_common.h) done by ONE wave: exclusive scan of the 256-bin histogram (four bins per lane), one atomic for the
// run, counting sort into the global queue; histogram and cursors are left zeroed for the buffer's next tile.
template <class Args>
__device__ __forceinline__ void s2_flush_tile(const Args& a, const int64_t tile_base, const uint16_t* s_idx, const uint8_t* s_key,
                                              unsigned* s_hist, unsigned* s_cursor, const unsigned count, const int lane) {
    unsigned long long qbase = 0;
    if (lane == 0 && count) qbase = atomicAdd(a.queue_count, (unsigned long long)count);   // (its round trip overlaps the scan)
    unsigned h[4], sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { h[i] = s_hist[4 * lane + i]; sum += h[i]; }
    unsigned incl = sum;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const unsigned o = __shfl_up(incl, d, WAVE);
        if (lane >= d) incl += o;
    }
    unsigned run = incl - sum;
#pragma unroll
    for (int i = 0; i < 4; ++i) { s_hist[4 * lane + i] = run; run += h[i]; }
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)qbase);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(qbase >> 32));
    qbase = ((unsigned long long)hi << 32) | lo;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (unsigned e = lane; e < count; e += WAVE) {
        const unsigned key = s_key[e];
        const unsigned p = s_hist[key] + atomicAdd(&s_cursor[key], 1u);
        a.queue[qbase + p] = (int32_t)(tile_base + s_idx[e]);
        a.queue_keys[qbase + p] = (uint8_t)key;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) { s_hist[4 * lane + i] = 0; s_cursor[4 * lane + i] = 0; }
}

typedef const __attribute__((address_space(4))) FilterArgs* s2_kernarg_ptr;
__device__ __forceinline__ S2Out s2_out_args() {
    s2_kernarg_ptr kp = (s2_kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));                                        // keeps the loads where the values are used
    S2Out o;
    o.present = kp->present; o.status = kp->status; o.queue = kp->queue; o.queue_count = kp->queue_count;
    o.queue_keys = kp->queue_keys;
    return o;
}

// SV: the reads are SUFFIX VIEWS of a uniform batch -- view r = [offsets[r], end of read r) of the parent batch
// (uniform_first, uniform_len): the second stage of a linked adapter (reference adapters.py:1222-1224: the 3' adapter
// is searched in read[front_match.rstop:]).  The parent's reads are streamed as they are and the `skip[r]` characters
// in front of a view are made NUL on their way into the words -- no k-mer matches through them.
// FR (with SV): the view's start is not read from an array but DECIDED here -- the anchored 5' adapter of a linked
// adapter that tolerates no error (k_anchored_exact's case: "^NNNNNNNNACGTACGT") is compared with the read's head while
// the head sits in the slot anyway; the kernel writes the front stage's result rows and the views (starts, lengths)
// for the kernels behind it.  One pass over the batch instead of three (front comparison, view arithmetic, prefilter).
// RV (with SV): the views may also END before their reads do -- view r = [offsets[r], offsets[r] + lens[r]) anywhere inside
// read r of the parent batch (reads cut at their 3' end by quality trimming, `-u -N`, `--length`: ragged lengths at a
// uniform stride).  The tail search sets count from the view's end (reference _kmer_finder.pyx:186-204: negative starts
// are relative to the sequence's length), so the views are streamed END-ALIGNED: the copy plan fetches unit u of read r
// from d[r] = (read end - view end) bytes further down, the words see a read of n characters whose last character is the
// view's last and whose first n - lens[r] characters are NUL.  Nothing else changes: the T-words open in the same chunks
// for every lane, the found masks stay wave-uniform.
// LONG: reads of 161 .. S2_LONG_MAX_LEN characters (2 x 250 / 2 x 300 bp runs).  A read is walked in SEGMENTS of ten units
// (160 characters): each segment is what the short form does with a whole read -- two fills of the wave's slot -- and
// the words simply go on from segment to segment (KmerFinder.kmers_present takes reads of any length, reference
// _kmer_finder.pyx:170-213).  A segment's copy registers are loaded while the second half of the segment before is matched.
template <int NL, int NT, bool BUF, bool SV = false, bool FR = false, bool LONG = false, bool RV = false>
__global__ __launch_bounds__(S2_WAVES * WAVE) void k_filter_stream2(FilterArgs a) {
    typedef S2Layout<NL, NT> LY;
    static_assert(!RV || (SV && !FR && !LONG && BUF), "k_filter_stream2: RV is a form of SV");
    constexpr int TILE = S2_TILE, SUBS = TILE / WAVE / S2_WAVES;
    // ONE static object, the tables first: they sit below 64 KB and an entry is read with
    // "ds_read_b64 v, v_entry offset:TABLE" -- the entry offset is the whole address computation of a character.
    // Everything is static: every LDS address is an immediate, none lives in a register.
    struct __attribute__((aligned(16))) S2Lds {
        uint32_t lead[4 * CAH_TABLE_CHARS * LY::NLP];
        uint32_t tail[4 * CAH_TABLE_CHARS * LY::NTP];
        uint32_t found[CAH_TW_DIST_LEN * LY::NTP];
        unsigned char slot[S2_WAVES * WAVE * S2_ROW];
        // survivor staging, TWO tiles' worth: while the last wave to finish a tile writes its survivors out
        // (s2_flush_tile), the others are matching the next tile into the other buffer -- no block-wide barrier
        struct Stage {
            uint16_t idx[TILE];                                         // tile-relative read index
            uint8_t key[TILE];
            unsigned hist[CAH_QUEUE_BINS], cursor[CAH_QUEUE_BINS];
            unsigned count;                                             // survivors staged
            unsigned arrived;                                           // waves that are through with the tile
            unsigned done;                                              // flushes completed on this buffer
            unsigned pad;
        } stage[2];
        unsigned next_piece;                                            // the block's pieces are dealt to its waves on demand
        uint32_t front[FR ? CAH_TABLE_CHARS : 1];                       // FR: bit i = character matches front adapter position i
        uint32_t views[RV ? S2_WAVES * WAVE : 1];                       // RV: d | skip << 16 of the wave's NEXT piece
        uint32_t view_end[RV ? 2 * S2_WAVES * WAVE : 1];                // RV, views anywhere: ... and their ends (byte in seqs, two words)
        uint32_t lowmask[RV ? 16 * 4 : 1];                              // RV: entry c (16 bytes): the first c bytes 0x00, the rest 0xFF
    };
    static_assert(sizeof(S2Lds) <= 160 * 1024, "k_filter_stream2: LDS");
    __shared__ S2Lds s_lds;
    if (a.batch_flag ? *a.batch_flag != 0ull : false) return;           // ragged batch: k_filter_lean<false, ..>
    const CahLeanFilter* lf = a.lean;
    const int64_t first = a.uniform_len > 0 ? a.uniform_first : a.offsets[0];
    const int n = a.uniform_len > 0 ? a.uniform_len : (int)(a.offsets[1] - first);     // every read has this length
    if (n < a.stream_n_lo || n > a.stream_n_hi) return;                 // k_filter_lean<true, ..> takes the batch
    const int n_reads = (int)a.n_reads;                                 // < 2^31 (check_batch)
    const int64_t total = (int64_t)n_reads * n;                         // bytes of the batch
    if (total < 16) return;                                             // (same test there)
    const bool clear0 = a.clear_out6 != nullptr && a.present == nullptr;
    // Ablation values of developer builds (-DCAH_S2_ABLATE lets the launcher put them into max_read_len from CAH_S2_NOMATCH;
    // the product library has no way to: it passes CAH_MAX_READ_LEN): 1 copy, match nothing; 2 ... and no result rows; 3
    // result rows only, no loads; 5 everything but the loads.  The three tests stay in the product kernel on purpose: with
    // them folded to constants (round 6 tried) the register allocator spills four VGPRs of the headline instantiation
    // (20 B of scratch per lane, + 0.1 ms per 100 M reads); as they are they cost three scalar compares per block.
    const bool nomatch = a.max_read_len <= -12345 && a.max_read_len >= -12347;
    const bool noclear = a.max_read_len == -12346 || a.max_read_len == -12348, noload = a.max_read_len == -12347 || a.max_read_len == -12349;   // 4: match, no result rows; 5: match what the slots happen to hold, no loads
    const bool clear = clear0 && !noclear;

    // tables (stream2.h: s2_entry); slots a plan does not use hold zeros
    for (int j = threadIdx.x; j < CAH_TABLE_CHARS * LY::NLP; j += blockDim.x) {
        const int c = j / LY::NLP, w = j % LY::NLP;
        const bool on = w < NL && w < lf->n_lead;
#pragma unroll
        for (int sh = 0; sh < 4; ++sh)
            s_lds.lead[sh * CAH_TABLE_CHARS * LY::NLP + j] = on ? s2_entry(lf->lead_mask[w][c], lf->lead_pass[w], lf->lead_init[w], sh) : 0u;
    }
    for (int j = threadIdx.x; j < CAH_TABLE_CHARS * LY::NTP; j += blockDim.x) {
        const int c = j / LY::NTP, w = j % LY::NTP;
        const bool on = w < NT && w < lf->n_tw;
#pragma unroll
        for (int sh = 0; sh < 4; ++sh)
            s_lds.tail[sh * CAH_TABLE_CHARS * LY::NTP + j] = on ? s2_entry(lf->tw_mask[w][c], lf->tw_pass[w], lf->tw_init[w], sh) : 0u;
    }
    for (int j = threadIdx.x; j < CAH_TW_DIST_LEN * LY::NTP; j += blockDim.x) {
        const int idx = j / LY::NTP, w = j % LY::NTP;
        s_lds.found[j] = (w < NT && w < lf->n_tw) ? lf->tw_found[w][idx] : 0u;
    }
    int front_m = 0, front_span = 0;
    if constexpr (FR) {
        const CahMatcher* fmt = a.front;
        for (int c = threadIdx.x; c < CAH_TABLE_CHARS; c += blockDim.x) s_lds.front[c] = (uint32_t)fmt->rowmask[c];
        front_m = fmt->m;                                               // <= 32 (api.cpp: linked_fusable)
        front_span = min(n, front_m + fmt->k);                          // the columns Aligner.locate reads (_align.pyx:348-352)
    }
    for (int b = 0; b < 2; ++b) {
        for (int i = threadIdx.x; i < CAH_QUEUE_BINS; i += blockDim.x) { s_lds.stage[b].hist[i] = 0; s_lds.stage[b].cursor[i] = 0; }
        if (threadIdx.x == 0) { s_lds.stage[b].count = 0; s_lds.stage[b].arrived = 0; s_lds.stage[b].done = 0; }
    }
    if (threadIdx.x == 0) s_lds.next_piece = S2_WAVES;                 // pieces 0 .. 15 are the waves' first ones
    if constexpr (RV) {
        if (threadIdx.x < 64) {
            const int drop = (int)(threadIdx.x >> 2) - 4 * (int)(threadIdx.x & 3);   // bytes of dword (x & 3) that entry x >> 2 clears
            s_lds.lowmask[threadIdx.x] = drop <= 0 ? 0xFFFFFFFFu : (drop >= 4 ? 0u : (0xFFFFFFFFu << (8 * drop)));
        }
    }
    __syncthreads();                                                    // the kernel's only barrier
    S2Words<NL, NT> K;
#pragma unroll
    for (int w = 0; w < NL; ++w) {
        K.l_init4[w] = w < lf->n_lead ? s2_init4(lf->lead_init[w]) : 0u;
        K.l_found[w] = w < lf->n_lead ? lf->lead_found[w] : 0u;
    }
    int tspan[NT > 0 ? NT : 1];
#pragma unroll
    for (int w = 0; w < NT; ++w) {
        K.t_init4[w] = w < lf->n_tw ? s2_init4(lf->tw_init[w]) : 0u;
        tspan[w] = w < lf->n_tw ? lf->tw_span[w] : -(1 << 30);
    }
    K.lead = reinterpret_cast<const unsigned char*>(s_lds.lead);
    K.tail = reinterpret_cast<const unsigned char*>(s_lds.tail);
    K.found = reinterpret_cast<const unsigned char*>(s_lds.found);

    const int lane = wave_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (the compiler does not know it is wave-uniform)
    unsigned char* const slot = s_lds.slot + wave * (WAVE * S2_ROW);    // this wave's LDS slot
    const unsigned char* const row = slot + lane * S2_ROW;              // this lane's half-read in it
    const unsigned lane16 = (unsigned)lane * 16u;
    unsigned shv = LY::LEAD_SHIFT, shv_tail = LY::TAIL_SHIFT;           // the SDWA shift amounts want registers
    asm volatile("" : "+v"(shv), "+v"(shv_tail));

    // Copy plan.  A read has U = ceil(n / 16) units (the last one runs into the next read -- masked when used):
    // H1 = ceil(U / 2) in the first half-row, H2 = U - H1 in the second.  Load k < H1 takes the first-half unit
    // u = 64 k + lane = unit u % H1 of read u / H1, load 5 + k (k < H2) second-half unit u = unit H1 + u % H2 of read
    // u / H2.  Byte offset within the piece: r * n + 16 (c [+ H1]) = r * (n - 16 H) + 16 u [+ 16 H1]; in the slot:
    // r * 80 + 16 c = r * (80 - 16 H) + 16 u (16 u when H = 5).  u / H as a multiply: exact for u < 320, H <= 5
    // (tests/test_host_logic.py checks it exhaustively).
    const int U_all = (n + 15) >> 4;
    const int n_seg = LONG ? (U_all + S2_SEG - 1) / S2_SEG : 1;         // (short form: one segment = the read)
    const int U_last = U_all - (n_seg - 1) * S2_SEG;                    // units of the last segment (<= 10)
    // the shape of the segment at work: H1 units in the first fill of the slot, H2 in the second (every segment but the
    // last: 5 + 5)
    int H1 = LONG ? S2_HALF : (U_all + 1) >> 1, H2 = LONG ? S2_HALF : U_all - H1;
    unsigned magic1 = (65536u + (unsigned)H1 - 1u) / (unsigned)H1;
    unsigned magic2 = H2 ? (65536u + (unsigned)H2 - 1u) / (unsigned)H2 : 0u;
    int seg_off = 0;                                                    // first character of the segment at work
    auto set_segment = [&](const int seg) {
        if constexpr (LONG) {
            const int u = seg == n_seg - 1 ? U_last : S2_SEG;
            H1 = (u + 1) >> 1; H2 = u - H1;
            magic1 = (65536u + (unsigned)H1 - 1u) / (unsigned)H1;
            magic2 = H2 ? (65536u + (unsigned)H2 - 1u) / (unsigned)H2 : 0u;
            seg_off = seg * (S2_SEG * 16);
        }
    };
    // read index of unit 64 k + lane of a half with H units per read (recomputed where used: a register per unit
    // would cost the kernel its fourth wave per SIMD)
    auto unit_r = [&](int k, unsigned magic) -> int {
        unsigned ln = (unsigned)lane;
        asm volatile("" : "+v"(ln));
        return (int)(__umul24((unsigned)(k * WAVE) + ln, magic) >> 16);
    };
    // Pieces of 64 reads, numbered per block: piece p = sub-tile p % PPT of the block's tile p / PPT, i.e. of tile
    // blockIdx.x + (p / PPT) gridDim.x of the batch.  The waves TAKE pieces (an LDS counter) instead of being dealt a
    // fixed share: the SIMD's arbiter favours its oldest wave, a fixed share left the others a quarter behind and
    // everybody waiting for them at the tile's end.  int64: pieces behind the batch's end may lie beyond 2^31.
    constexpr int PPT = TILE / WAVE;
    auto piece_base = [&](unsigned p) -> int64_t {
        return ((int64_t)blockIdx.x + (int64_t)(p / PPT) * (int64_t)gridDim.x) * TILE + (int64_t)(p % PPT) * WAVE;
    };
    auto take_piece = [&]() -> unsigned {
        unsigned p = 0;
        if (lane == 0) p = __hip_atomic_fetch_add(&s_lds.next_piece, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(p);
    };
    s2_u32x4 pre[2 * S2_HALF];
    const uint8_t* const batch0 = a.seqs + first;
    // the units of the piece starting at read `base`, into VGPRs.  Nothing outside the batch is touched: a unit
    // that would run past the batch's last byte is fetched as the 16 bytes that END there and shifted down.
    // RV: vw = d | skip << 16 of read base + lane (0 in lanes behind the batch's end); a unit of read r starts d[r] bytes
    // earlier -- fetched from lane r by ds_bpermute -- and a unit that lies in front of its read's view altogether is not
    // fetched at all: its offset is put out of the resource's range, the load returns zeros (NUL: what the words must see
    // there).  The resource's base sits S2_RV_BACK bytes in front of the piece (d <= n, and for n < 16 the copy plan's
    // stride n - 16 H is negative): the batch's first pieces go lane by lane.
    // RV, views anywhere in the buffer (suffix_views == 3): the gathering copy
    auto prefetch_general = [&](int64_t base, const uint32_t vw, const int64_t ve_parked, const bool parked) {
        const int64_t left = n_reads - base;                            // wave-uniform
        if (left <= 0) return;
                // Views ANYWHERE in the buffer (cah_match_batch_frames: a packed batch, the reads of a raw FASTQ chunk): unit u
                // of the piece -- frame characters 16 c .. 16 c + 15 of read r -- is gathered from its view's END: byte (view
                // end) - n + 16 c.  The end (relative to lane 0's) and the NULs in front of the view travel from lane r by two
                // lane exchanges per unit.  Nothing outside a view is touched: a unit that reaches in front of its view or
                // behind it is loaded as the 16 bytes at the view's edge and shifted; views under 16 characters byte by byte.
                const int reads = (int)(left < WAVE ? left : (int64_t)WAVE);
                const int64_t r_own = base + (lane < reads ? lane : 0);
                const int sk_own = lane < reads ? (int)(vw >> 16) : n;
                // (the views' ends: parked in LDS a half-piece ahead like the views themselves -- nobody waits for the loads --,
                // or fetched here for a wave's first piece)
                const int64_t ve_own = parked ? ve_parked : a.offsets[r_own] + (int64_t)(n - sk_own);
                const unsigned e0_lo = __builtin_amdgcn_readfirstlane((unsigned)ve_own);
                const unsigned e0_hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)ve_own >> 32));
                const int64_t e0 = (int64_t)(((unsigned long long)e0_hi << 32) | e0_lo);
                // (the extent of the piece's own views: every byte between is inside the buffer.  Views that lie more than 2^30
                // bytes from lane 0's -- a batch of views in no order -- give no such extent: every unit then takes the careful way)
                const int64_t rel64 = ve_own - e0;
                const bool near = lane >= reads || (rel64 > -(1ll << 30) && rel64 < (1ll << 30));
                const bool all_near = __builtin_amdgcn_ballot_w64(!near) == 0ull;
                const int rel_own = near ? (int)rel64 : 0;
                const int len_own = n - sk_own;
                const int64_t safe_lo = all_near ? e0 + wave_min_i32(lane < reads ? rel_own - len_own : 0x7FFFFFFF) : 0;
                const int64_t safe_hi = all_near ? e0 + wave_max_i32(lane < reads ? rel_own : -0x7FFFFFFF) : 0;
                const int ve_lo_own = (int)(unsigned)ve_own, ve_hi_own = (int)(unsigned)((unsigned long long)ve_own >> 32);
                const uint8_t* const sq = a.seqs;
#pragma unroll
                for (int q = 0; q < 2 * S2_HALF; ++q) {
                    const int k = q < S2_HALF ? q : q - S2_HALF;
                    const int H = q < S2_HALF ? H1 : H2;
                    s2_u32x4 got = (s2_u32x4)(0u);
                    if (k < H) {                                        // (wave-uniform: every lane takes part in the exchanges)
                        const int rr = unit_r(k, q < S2_HALF ? magic1 : magic2);
                        const unsigned ve_lo_r = (unsigned)__builtin_amdgcn_ds_bpermute(rr << 2, ve_lo_own);
                        const unsigned ve_hi_r = (unsigned)__builtin_amdgcn_ds_bpermute(rr << 2, ve_hi_own);
                        const int64_t veb_r = (int64_t)(((unsigned long long)ve_hi_r << 32) | ve_lo_r);
                        const int sk_r = __builtin_amdgcn_ds_bpermute(rr << 2, sk_own);
                        const int u = k * WAVE + lane;
                        const int fc = 16 * (u - __mul24(rr, H)) + (q < S2_HALF ? 0 : 16 * H1);
                        const int len = n - sk_r;
                        if (u < reads * H && fc + 15 >= sk_r && fc < n && len > 0) {
                            unsigned x0, x1, x2, x3;
                            gather_frame_unit(sq, veb_r, len, n, fc, safe_lo, safe_hi, x0, x1, x2, x3);
                            got = (s2_u32x4){x0, x1, x2, x3};
                        }
                    }
                    pre[q] = got;
                }
    };
    auto prefetch = [&](int64_t base, const uint32_t vw = 0) {
        const int64_t left = n_reads - base;                            // wave-uniform
        if (left <= 0 || noload) return;
        const int64_t pbyte = base * (int64_t)n + seg_off;              // the first byte of the piece's segment within the batch
        const uint8_t* const src = batch0 + pbyte;
        if (left >= WAVE && pbyte + (int64_t)WAVE * n + 16 <= total && (!RV || pbyte >= S2_RV_BACK)) {
            // a whole piece with 16 bytes of the batch behind it (all but the last pieces): no lane needs a check
            if constexpr (BUF) {
                // buffer loads: the piece's first byte is the (scalar) base of the resource, a lane's offset ONE register
                // (said explicitly to be wave-uniform: otherwise every load is wrapped in a waterfall loop over the
                // values of the four resource words)
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(s2_uniform_ptr(RV ? src - S2_RV_BACK : src), 0, 0x7FFFFFFF, 0x00020000);
                if constexpr (RV) {
                    // What a unit needs of its read r is computed ONCE per read, in lane r, for either half's shape (H units
                    // per read) and handed over in one lane exchange per unit:
                    //   low half  A = r (n - 16 H) + S2_RV_BACK - d[r]: unit 64 k + lane starts at byte A + 16 lane + 1024 k
                    //             of the resource (the copy plan's offset, moved d[r] down);
                    //   high half T = skip[r] + 16 r H: the unit lies in front of the view iff its last character,
                    //             16 (64 k + lane) + 15 [+ 16 H1] in these terms, is < T.
                    const int d = (int)(vw & 0xFFFFu), sk = (int)(vw >> 16);
                    const unsigned w1 = (unsigned)(__mul24(lane, n - 16 * H1) + S2_RV_BACK - d) | ((unsigned)(sk + 16 * __mul24(lane, H1)) << 16);
                    const unsigned w2 = (unsigned)(__mul24(lane, n - 16 * H2) + S2_RV_BACK - d) | ((unsigned)(sk + 16 * __mul24(lane, H2)) << 16);
                    int got[2 * S2_HALF];
#pragma unroll
                    for (int k = 0; k < S2_HALF; ++k) {                // (units a half does not have: fetched, not used)
                        got[k] = __builtin_amdgcn_ds_bpermute(unit_r(k, magic1) << 2, (int)w1);
                        got[S2_HALF + k] = __builtin_amdgcn_ds_bpermute(unit_r(k, magic2) << 2, (int)w2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < S2_HALF; ++k)
                        if (k < H1) {
                            const int w = got[k];
                            const unsigned off = (int)lane16 + (16 * k * WAVE + 15) < (w >> 16) ? 0x80000000u : (unsigned)((w & 0xFFFF) + (int)lane16);
                            pre[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + k * (WAVE * 16), 0, 0);
                        }
#pragma unroll
                    for (int k = 0; k < S2_HALF; ++k)
                        if (k < H2) {
                            const int w = got[S2_HALF + k];
                            const unsigned off = (int)lane16 + (16 * k * WAVE + 15) + 16 * H1 < (w >> 16) ? 0x80000000u : (unsigned)((w & 0xFFFF) + (int)lane16);
                            pre[S2_HALF + k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + k * (WAVE * 16), 16 * H1, 0);
                        }
                    return;
                }
#pragma unroll
                for (int k = 0; k < S2_HALF; ++k)
                    if (k < H1)
                        pre[k] = __builtin_amdgcn_raw_buffer_load_b128(
                            rs, (unsigned)(__mul24(unit_r(k, magic1), n - 16 * H1) + (int)lane16) + k * (WAVE * 16), 0, 0);
#pragma unroll
                for (int k = 0; k < S2_HALF; ++k)
                    if (k < H2)
                        pre[S2_HALF + k] = __builtin_amdgcn_raw_buffer_load_b128(
                            rs, (unsigned)(__mul24(unit_r(k, magic2), n - 16 * H2) + (int)lane16) + k * (WAVE * 16), 16 * H1, 0);
            } else {
#pragma unroll
                for (int k = 0; k < S2_HALF; ++k)
                    if (k < H1) {
                        const unsigned off = (unsigned)(__mul24(unit_r(k, magic1), n - 16 * H1) + (int)lane16);
                        Unaligned16 v;
                        __builtin_memcpy(&v, src + k * (WAVE * 16) + off, 16);
                        pre[k] = (s2_u32x4){v.w[0], v.w[1], v.w[2], v.w[3]};
                    }
#pragma unroll
                for (int k = 0; k < S2_HALF; ++k)
                    if (k < H2) {
                        const unsigned off = (unsigned)(__mul24(unit_r(k, magic2), n - 16 * H2) + (int)lane16);
                        Unaligned16 v;
                        __builtin_memcpy(&v, src + (16 * H1 + k * (WAVE * 16)) + off, 16);
                        pre[S2_HALF + k] = (s2_u32x4){v.w[0], v.w[1], v.w[2], v.w[3]};
                    }
            }
            return;
        }
        // the batch's last piece(s): lane by lane (rolled: this runs once per block at most)
        const int reads = (int)(left < WAVE ? left : (int64_t)WAVE);
#pragma unroll
        for (int q = 0; q < 2 * S2_HALF; ++q) {
            const int k = q < S2_HALF ? q : q - S2_HALF;
            const int H = q < S2_HALF ? H1 : H2;
            s2_u32x4 got = (s2_u32x4)(0u);
            if constexpr (RV) {
                // byte by byte: a shifted unit may begin in front of the batch or end behind it
                if (k < H) {                                            // (wave-uniform: every lane takes part in the bpermute)
                    const int r = unit_r(k, q < S2_HALF ? magic1 : magic2);          // < 64 for k < H
                    const int w = __builtin_amdgcn_ds_bpermute(r << 2, (int)vw);
                    const int sh = w & 0xFFFF;
                    const int last = 16 * (k * WAVE + lane - __mul24(r, H)) + (q < S2_HALF ? 15 : 16 * H1 + 15);
                    if (k * WAVE + lane < reads * H && last >= (w >> 16)) {  // (a unit in front of the view: zeros)
                        const int64_t g0 = pbyte + (int64_t)(__mul24(r, n - 16 * H) + (k * WAVE) * 16 + (int)lane16 +
                                                             (q < S2_HALF ? 0 : 16 * H1)) - sh;
                        unsigned x[4] = {0u, 0u, 0u, 0u};
#pragma unroll 1
                        for (int b = 0; b < 16; ++b) {
                            const int64_t at = g0 + b;
                            if (at >= 0 && at < total) x[b >> 2] |= (unsigned)batch0[at] << (8 * (b & 3));
                        }
                        got = (s2_u32x4){x[0], x[1], x[2], x[3]};
                    }
                }
                pre[q] = got;
                continue;
            }
            if (k < H && k * WAVE + lane < reads * H) {
                const int r = unit_r(k, q < S2_HALF ? magic1 : magic2);
                const unsigned goff = (unsigned)(__mul24(r, n - 16 * H) + (k * WAVE) * 16 + (int)lane16 +
                                                 (q < S2_HALF ? 0 : 16 * H1));
                if (pbyte + goff + 16 <= total) {
                    Unaligned16 v;
                    __builtin_memcpy(&v, src + goff, 16);
                    got = (s2_u32x4){v.w[0], v.w[1], v.w[2], v.w[3]};
                } else if (pbyte + goff < total) {
                    Unaligned16 v;
                    __builtin_memcpy(&v, batch0 + (total - 16), 16);
                    const int sft = (int)(pbyte + goff + 16 - total);   // 1..15 bytes to drop
                    const int dw = sft >> 2, sh = (sft & 3) * 8;
                    unsigned x0 = v.w[0], x1 = v.w[1], x2 = v.w[2], x3 = v.w[3];
                    if (dw >= 2) { x0 = x2; x1 = x3; x2 = 0; x3 = 0; }
                    if (dw & 1) { x0 = x1; x1 = x2; x2 = x3; x3 = 0; }
                    got = (s2_u32x4){(unsigned)((((unsigned long long)x1 << 32) | x0) >> sh),
                                     (unsigned)((((unsigned long long)x2 << 32) | x1) >> sh),
                                     (unsigned)((((unsigned long long)x3 << 32) | x2) >> sh), x3 >> sh};
                }
            }
            pre[q] = got;
        }
    };
    // the registers of one half go to the wave's slot (every lane is done with what the slot held: the LDS
    // operations of a wave execute in order)
    auto to_slot = [&](auto half_c) {
        constexpr int half = decltype(half_c)::value;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int H = half ? H2 : H1;
        const unsigned magic = half ? magic2 : magic1;
#pragma unroll
        for (int k = 0; k < S2_HALF; ++k)
            if (k < H) {
                unsigned off = lane16;
                if (H != S2_HALF) off += (unsigned)__mul24(unit_r(k, magic), S2_ROW - 16 * H);
                *reinterpret_cast<s2_u32x4*>(slot + k * (WAVE * 16) + off) = pre[half * S2_HALF + k];
            }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    int skip = 0;                                                       // SV: this lane's read starts `skip` characters in
    // the chunk at `pos`: characters past the read's end (the next read's, or stale) become NUL
    auto finish = [&](s2_u32x4 v, int pos) -> s2_u32x4 {
        if constexpr (SV && !RV) {                                      // (RV: the slot's rows are cleared in front of the views)
            if (s2_any(skip > pos)) {                                   // (wave-uniform: rare beyond the first chunk or two)
                unsigned x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int drop = skip - pos - 4 * i;                // characters of dword i in front of the view
                    x[i] = drop <= 0 ? x[i] : (drop >= 4 ? 0u : (x[i] & (0xFFFFFFFFu << (8 * drop))));
                }
                v = (s2_u32x4){x[0], x[1], x[2], x[3]};
            }
        }
        if (pos + 16 > n) {
            unsigned x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int keep = n - pos - 4 * i;                       // characters of dword i inside the read
                x[i] &= keep >= 4 ? 0xFFFFFFFFu : (keep <= 0 ? 0u : ((1u << (8 * keep)) - 1u));
            }
            v = (s2_u32x4){x[0], x[1], x[2], x[3]};
        }
        return v;
    };

    // RV: the view of read base + lane inside its read as d | skip << 16 -- d: characters between the view's end and the
    // read's, skip = n - length: where the view starts in the end-aligned read (the wave's first piece; the others: below)
    int skip_next = 0;
    auto view_of = [&](const int64_t base) -> uint32_t {
        const int64_t r = base + lane;
        if (r >= n_reads) return 0u;
        const s2_kernarg_ptr kp = (s2_kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        int ln = kp->lens[r];
        if (kp->suffix_views == 3) {                                    // views anywhere: no read around them (d = 0, the copy gathers)
            ln = ln < 0 ? 0 : (ln > n ? n : ln);
            return (uint32_t)(n - ln) << 16;
        }
        int st = (int)(kp->offsets[r] - (first + r * (int64_t)n));
        st = st < 0 ? 0 : (st > n ? n : st);
        ln = ln < 0 ? 0 : (ln > n - st ? n - st : ln);
        return (uint32_t)(n - (st + ln)) | ((uint32_t)(n - ln) << 16);
    };
    unsigned p = (unsigned)wave;                                        // this wave's piece
    int it = 0;                                                         // (pieces this wave has matched: the trace build's index)
    if constexpr (RV) {
        const uint32_t vw = view_of(piece_base(p));
        skip_next = (int)(vw >> 16);
        if (a.suffix_views == 3) prefetch_general(piece_base(p), vw, 0, false); else prefetch(piece_base(p), vw);
    } else {
        prefetch(piece_base(p));
    }
#pragma unroll 1
    for (;; ++it) {
        const unsigned kt = p / PPT;
        const int64_t tile_base64 = ((int64_t)blockIdx.x + (int64_t)kt * (int64_t)gridDim.x) * TILE;
        if (tile_base64 >= n_reads) break;                              // this piece and every later one: behind the batch
        const int tile_base = (int)tile_base64;
        auto& stage = s_lds.stage[kt & 1];                              // the tile's staging buffer
        {
            const int base = tile_base + (int)(p % PPT) * WAVE;         // < 2^31 + 2^13: compared as unsigned
            const bool more = (unsigned)base < (unsigned)n_reads;       // wave-uniform
            S2Hits hits;                                                // lanes still looking for a first k-mer
            hits.live = more && (unsigned)(base + lane) < (unsigned)n_reads;
            hits.group = -1;
            if constexpr (RV) {
                skip = skip_next;                                       // (fetched with the piece's loads)
            } else if constexpr (SV && !FR) {
                skip = 0;
                if (hits.live) {
                    const s2_kernarg_ptr kp = (s2_kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
                    const int64_t r = (int64_t)base + lane;
                    skip = (int)(kp->offsets[r] - (first + r * (int64_t)n));
                    skip = skip < 0 ? 0 : (skip > n ? n : skip);
                }
            }
            unsigned seen = 0;
            unsigned p_early = 0;                                       // RV: the piece taken for the next round
            int64_t view_at = 0;
            int view_len = 0;
            uint32_t RL[NL > 0 ? NL : 1], RT[NT > 0 ? NT : 1];
#pragma unroll
            for (int w = 0; w < (NL > 0 ? NL : 1); ++w) RL[w] = 0;
#pragma unroll
            for (int w = 0; w < (NT > 0 ? NT : 1); ++w) RT[w] = 0;
            s2_u32x4 cur = (s2_u32x4)(0u);
            S2Masks<NL> M;
            S2_STAMP(0);
#pragma unroll 1
            for (int seg = 0; seg < n_seg; ++seg) {
            set_segment(seg);
#pragma unroll 1
            for (int ph = 0; ph < 2; ++ph) {
                const int H = ph ? H2 : H1;
                const bool alive = H > 0 && s2_any(hits.live) && !nomatch;   // wave-uniform
                if constexpr (RV) {
                    // The views of the wave's NEXT piece (its loads depend on them: a unit's address is shifted by its read's
                    // d): requested once the first half's copy registers are in the slot, parked in LDS when the first half
                    // is matched -- nobody waits for them, and no register holds them through the T-words' chunks.
                    if (ph == 1) {
                        int st = (int)(view_at - (first + (piece_base(p_early) + lane) * (int64_t)n));
                        st = st < 0 ? 0 : (st > n ? n : st);
                        int ln = view_len;
                        if (a.suffix_views == 3) { ln = ln < 0 ? 0 : (ln > n ? n : ln); st = n - ln; }    // (views anywhere: d = 0)
                        ln = ln < 0 ? 0 : (ln > n - st ? n - st : ln);
                        s_lds.views[wave * WAVE + lane] = (uint32_t)(n - (st + ln)) | ((uint32_t)(n - ln) << 16);
                        if (a.suffix_views == 3) {                      // (the gathering copy starts from the views' ENDS)
                            const unsigned long long ve = (unsigned long long)(view_at + ln);
                            s_lds.view_end[2 * (wave * WAVE + lane)] = (uint32_t)ve;
                            s_lds.view_end[2 * (wave * WAVE + lane) + 1] = (uint32_t)(ve >> 32);
                        }
                    }
                }
                if (alive) {
                    if (ph == 0) to_slot(std::integral_constant<int, 0>{}); else to_slot(std::integral_constant<int, 1>{});
                    if constexpr (RV) {
                        // the characters in front of the view in the unit it starts in, made NUL in the lane's own row ONCE
                        // per fill (in the chunks' registers it would be a byte mask per chunk: nearly every chunk holds
                        // some lane's start; the units in front of that one arrived as zeros)
                        const int lead = skip - (ph ? 16 * H1 : 0);     // this half's characters in front of the view
                        if (s2_any(lead > 0)) {
                            unsigned char* const wrow = slot + lane * S2_ROW;
                            const int whole = lead >> 4;
                            const int rem = lead & 15;
                            if (lead > 0 && whole < H && rem) {
                                const s2_u32x4 v = *reinterpret_cast<const s2_u32x4*>(wrow + 16 * whole);
                                const s2_u32x4 m = *reinterpret_cast<const s2_u32x4*>(
                                    reinterpret_cast<const unsigned char*>(s_lds.lowmask) + 16 * rem);
                                *reinterpret_cast<s2_u32x4*>(wrow + 16 * whole) = v & m;
                            }
                        }
                    }
                    cur = *reinterpret_cast<const s2_u32x4*>(row);
                }
                if constexpr (RV) {
                    if (ph == 0) {
                        p_early = take_piece();
                        const int64_t r = piece_base(p_early) + lane;
                        view_at = first + r * (int64_t)n; view_len = n;  // (lanes behind the batch's end: d = skip = 0)
                        if (r < n_reads) {
                            const s2_kernarg_ptr kp = (s2_kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
                            view_at = kp->offsets[r];
                            view_len = kp->lens[r];
                        }
                    }
                }
                if (alive) {
                    if constexpr (LONG) {
                        // the next segment's units: requested now that this segment's copy registers are all in the slot,
                        // on their way while its second half is matched (no T-word is at work before the last segment)
                        if (ph == 1 && seg + 1 < n_seg) {
                            set_segment(seg + 1);
                            prefetch((int64_t)base);
                            set_segment(seg);
                        }
                    }
                }
                if constexpr (FR) {
                    if (ph == 0) {
                        // the front adapter against the read's head (k_anchored_exact, kernels.hip: same rule, same flags)
                        unsigned bad = n < front_m ? 1u : 0u, fseen = 0;
                        if (alive) {
#pragma unroll 1
                            for (int c0 = 0; c0 < front_span; c0 += 16) {                 // wave-uniform: one or two rounds
                                const s2_u32x4 v = *reinterpret_cast<const s2_u32x4*>(row + c0);
                                unsigned x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const int keep = front_span - c0 - 4 * i;             // characters of dword i inside the span
                                    x[i] &= keep >= 4 ? 0xFFFFFFFFu : (keep <= 0 ? 0u : ((1u << (8 * keep)) - 1u));
                                    fseen |= x[i];
                                }
#pragma unroll
                                for (int t = 0; t < 16; ++t) {
                                    const int i = c0 + t;                                 // adapter position (wave-uniform)
                                    if (i < front_m) {
                                        const unsigned ch = (x[t >> 2] >> ((t & 3) * 8)) & (CAH_TABLE_CHARS - 1);
                                        bad |= ~s_lds.front[ch] & (1u << i);
                                    }
                                }
                            }
                        }
                        const bool finvalid = (fseen & 0x80808080u) != 0;
                        const bool ffound = !finvalid && bad == 0;
                        skip = ffound ? front_m : 0;
                        if (hits.live) {
                            const s2_kernarg_ptr kp = (s2_kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
                            const int64_t r = (int64_t)base + lane;
                            int32_t* o = kp->front_out6 + r * 6;
                            const int fm = ffound ? front_m : 0;
                            o[0] = 0; o[1] = fm; o[2] = 0; o[3] = fm; o[4] = fm; o[5] = 0;
                            kp->front_status[r] = finvalid ? (uint8_t)2 : (ffound ? (uint8_t)1 : (uint8_t)0);
                            if (kp->front_best) kp->front_best[r] = ffound ? 0 : -1;
                            kp->view_starts[r] = first + r * (int64_t)n + skip;
                            kp->view_lens[r] = n - skip;
                        }
                    }
                }
                S2_STAMP(1 + 3 * ph);
                S2_STAMP(2 + 3 * ph);
                if (!alive) continue;
                const int pos0 = seg_off + (ph ? 16 * H1 : 0);
                // cur: this chunk's characters; nxt: the next chunk's (its lead masks are requested while this one is
                // matched); the chunk after that is requested from the slot meanwhile
                s2_u32x4 nxt = (s2_u32x4)(0u);
                if (1 < H) nxt = *reinterpret_cast<const s2_u32x4*>(row + 16);
                cur = finish(cur, pos0);
                s2_lead_loads<NL, NT>(K, M, 0, cur.x, shv);
#pragma unroll 1
                for (int c = 0; c < H; ++c) {
                    if (c > 0 && !s2_any(hits.live)) break;
                    const int pos = pos0 + 16 * c;
                    s2_u32x4 nn = (s2_u32x4)(0u);
                    if (c + 2 < H) nn = *reinterpret_cast<const s2_u32x4*>(row + 16 * (c + 2));
                    const s2_u32x4 cw = cur;
                    const s2_u32x4 nw = finish(nxt, pos + 16);
                    seen = s2_or3(seen, cw.x, cw.y);
                    seen = s2_or3(seen, cw.z, cw.w);
                    // every T-word from the chunk on in which the widest window opens (a word whose windows open later
                    // finds nothing before: its found masks are zero there)
                    s2_chunk<NL, NT>(K, RL, RT, M, cw, nw, pos, n, NT > 0 && pos + 16 > n - tspan[0], hits, shv, shv_tail);
                    cur = nw;
                    nxt = nn;
                }
                S2_STAMP(3 + 2 * ph);
            }
            }
            if constexpr (LONG) set_segment(0);                         // (the next piece's first segment is loaded below)
            // The next piece's loads go out now, not earlier: through the second half -- the T-words' chunks -- no copy
            // register is live, through the first only the second half's (a fourth wave per SIMD needs that; the
            // other three cover the wait at the next piece's start).  The stores of the result rows in front of them:
            // loads and stores share the in-order vmcnt counter.
            if (clear && more)
                s2_clear_rows(a.clear_out6, a.clear_best, (int64_t)base, n_reads - base < WAVE ? n_reads - base : WAVE, lane);
            S2_STAMP(8);
            const unsigned p_next = RV ? p_early : take_piece();          // (RV: taken with the views, above)
            S2_STAMP(9);
            if constexpr (RV) {
                const uint32_t vw = s_lds.views[wave * WAVE + lane];
                skip_next = (int)(vw >> 16);
                if (a.suffix_views == 3) {
                    const unsigned long long ve = ((unsigned long long)s_lds.view_end[2 * (wave * WAVE + lane) + 1] << 32) |
                                                  s_lds.view_end[2 * (wave * WAVE + lane)];
                    prefetch_general(piece_base(p_next), vw, (int64_t)ve, true);
                } else prefetch(piece_base(p_next), vw);
            } else {
                prefetch(piece_base(p_next));
            }
            S2_STAMP(6);
            if (!a.present) {
                // the tile's staging buffer is free once the tile before the last one is written out
                while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&stage.done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) !=
                       (kt >> 1))
                    __builtin_amdgcn_s_sleep(2);
            }
            S2_STAMP(10);
            if (more) {
                const bool invalid = (seen & 0x80808080u) != 0;
                const S2Out o = s2_out_args();
                lean_emit(o, (int64_t)base + lane, (int64_t)tile_base, (unsigned)(base + lane) < (unsigned)n_reads, hits.group >= 0,
                          invalid, SV ? max(0, (hits.group << CAH_KEY_SHIFT) - skip) : (hits.group << CAH_KEY_SHIFT),   // (a lower bound of the first hit within the view)
                          stage.idx, stage.key, stage.hist, stage.count);
            }
            S2_STAMP(7);
            if (!a.present) {
                // The wave that stages a tile's last piece writes the tile's survivors out, alone, while the others go
                // on (LDS operations of a wave execute in order: what it staged is in place when its piece is counted).
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                unsigned before = 0;
                if (lane == 0) before = __hip_atomic_fetch_add(&stage.arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                before = __builtin_amdgcn_readfirstlane(before);
                if (before == PPT - 1) {
                    const S2Out o = s2_out_args();
                    s2_flush_tile(o, (int64_t)tile_base, stage.idx, stage.key, stage.hist, stage.cursor, stage.count, lane);
                    if (lane == 0) {
                        stage.count = 0;
                        stage.arrived = 0;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) __hip_atomic_fetch_add(&stage.done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            p = p_next;
        }
    }
}

// classes <lead words, T-words>
hipError_t launch_filter_stream2(const FilterArgs& a_in, int mode, int n_lead, int n_tw, int n_cus, hipStream_t s) {
    FilterArgs a = a_in;
    if (mode != 0) a.present = nullptr;
    a.stream_n_lo = 1; a.stream_n_hi = S2_MAX_LEN;
    // reads of 161 .. 640 characters: the LONG form (CAH_NO_STREAM2_LONG=1: the per-lane kernel, A/B)
    const char* const el = getenv("CAH_NO_STREAM2_LONG");
    const bool long_ok = !(el && *el && *el != '0') && !a.suffix_views;
    const bool want_long = long_ok && (a.uniform_len > S2_MAX_LEN || (a.uniform_len == 0 && a.batch_flag));
    const bool want_short = a.uniform_len == 0 || a.uniform_len <= S2_MAX_LEN;
#ifdef CAH_S2_ABLATE
    // developer builds only (-DCAH_S2_ABLATE, profiles/scripts/r03_nomatch.sh): timing with parts of the kernel switched
    // off -- the results are wrong, so the knob does not exist in the product library
    { const char* e = getenv("CAH_S2_NOMATCH"); if (e && *e && *e != '0') a.max_read_len = -12344 - atoi(e); }
#endif
    const int tiles = (int)((a.n_reads + S2_TILE - 1) / S2_TILE);
    const int grid = std::max(1, std::min(tiles, n_cus));
    const size_t lds = 0;                                 // every LDS object of the kernel is static
    // CAH_S2_GLOBAL=1: plain global loads instead of buffer loads for the copy (A/B)
    const char* const eg = getenv("CAH_S2_GLOBAL");
    const bool use_global = eg && *eg && *eg != '0';
#define S2_LAUNCH(NL, NT)                                                                                               \
    do {                                                                                                                \
        if (use_global) hipLaunchKernelGGL((k_filter_stream2<NL, NT, false>), dim3(grid), dim3(S2_WAVES * WAVE), lds, s, a); \
        else hipLaunchKernelGGL((k_filter_stream2<NL, NT, true>), dim3(grid), dim3(S2_WAVES * WAVE), lds, s, a);        \
    } while (0)
    if (a.suffix_views && a.front) {
        if (n_lead <= 1 && n_tw <= 2) hipLaunchKernelGGL((k_filter_stream2<1, 2, true, true, true>), dim3(grid), dim3(S2_WAVES * WAVE), lds, s, a);
        else if (n_lead <= 2 && n_tw <= 4) hipLaunchKernelGGL((k_filter_stream2<2, 4, true, true, true>), dim3(grid), dim3(S2_WAVES * WAVE), lds, s, a);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (a.suffix_views == 2 || a.suffix_views == 3) {                   // (3: views anywhere in the buffer, frames of uniform_len)
        if (n_lead <= 1 && n_tw <= 2) hipLaunchKernelGGL((k_filter_stream2<1, 2, true, true, false, false, true>), dim3(grid), dim3(S2_WAVES * WAVE), lds, s, a);
        else if (n_lead <= 2 && n_tw <= 4) hipLaunchKernelGGL((k_filter_stream2<2, 4, true, true, false, false, true>), dim3(grid), dim3(S2_WAVES * WAVE), lds, s, a);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (a.suffix_views) {
        if (n_lead <= 1 && n_tw <= 2) hipLaunchKernelGGL((k_filter_stream2<1, 2, true, true>), dim3(grid), dim3(S2_WAVES * WAVE), lds, s, a);
        else if (n_lead <= 2 && n_tw <= 4) hipLaunchKernelGGL((k_filter_stream2<2, 4, true, true>), dim3(grid), dim3(S2_WAVES * WAVE), lds, s, a);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (want_short) {
        if (n_lead <= 1 && n_tw <= 2) S2_LAUNCH(1, 2);
        else if (n_lead <= 2 && n_tw <= 4) S2_LAUNCH(2, 4);
        else return hipErrorInvalidValue;
    }
#undef S2_LAUNCH
    if (want_long) {
        FilterArgs b = a;
        b.stream_n_lo = S2_MAX_LEN + 1; b.stream_n_hi = S2_LONG_MAX_LEN;
        if (n_lead <= 1 && n_tw <= 2) hipLaunchKernelGGL((k_filter_stream2<1, 2, true, false, false, true>), dim3(grid), dim3(S2_WAVES * WAVE), lds, s, b);
        else if (n_lead <= 2 && n_tw <= 4) hipLaunchKernelGGL((k_filter_stream2<2, 4, true, false, false, true>), dim3(grid), dim3(S2_WAVES * WAVE), lds, s, b);
        else return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

int stream2_max_len() { return S2_MAX_LEN; }
int stream2_long_max_len() {
    const char* const el = getenv("CAH_NO_STREAM2_LONG");
    return (el && *el && *el != '0') ? S2_MAX_LEN : S2_LONG_MAX_LEN;
}
bool stream2_class_ok(int n_lead, int n_tw) { return n_lead <= 2 && n_tw <= 4; }
