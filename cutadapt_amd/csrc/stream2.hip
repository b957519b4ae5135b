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
// a lead word over four characters: R = ((R << 4) | S3) & m0 & m1 & m2 & m3;  f |= R & FOUND   (S3, FOUND: SGPRs)
__device__ __forceinline__ void s2_lead_step(uint32_t& R, uint32_t& f, const uint32_t init4, const uint32_t found,
                                             const uint32_t m0, const uint32_t m1, const uint32_t m2, const uint32_t m3) {
    asm("v_lshl_or_b32 %0, %0, 4, %2\n\t"
        "v_bitop3_b32 %0, %0, %4, %5 bitop3:0x80\n\t"
        "v_bitop3_b32 %0, %0, %6, %7 bitop3:0x80\n\t"
        "v_bitop3_b32 %1, %0, %3, %1 bitop3:0xea"
        : "+v"(R), "+v"(f) : "s"(init4), "s"(found), "v"(m0), "v"(m1), "v"(m2), "v"(m3));
}
// a T-word: the found mask is the group's (a register: it comes out of LDS)
__device__ __forceinline__ void s2_tail_step(uint32_t& R, uint32_t& f, const uint32_t init4, const uint32_t fm,
                                             const uint32_t m0, const uint32_t m1, const uint32_t m2, const uint32_t m3) {
    asm("v_lshl_or_b32 %0, %0, 4, %2\n\t"
        "v_bitop3_b32 %0, %0, %4, %5 bitop3:0x80\n\t"
        "v_bitop3_b32 %0, %0, %6, %7 bitop3:0x80\n\t"
        "v_bitop3_b32 %1, %0, %3, %1 bitop3:0xea"
        : "+v"(R), "+v"(f) : "s"(init4), "v"(fm), "v"(m0), "v"(m1), "v"(m2), "v"(m3));
}
__device__ __forceinline__ uint32_t s2_or3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xfe" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// old where the lane's bit of `mask` (a wave-wide SGPR mask) is clear, val where it is set
__device__ __forceinline__ int s2_pick(int old, int val, unsigned long long mask) {
    int d;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(old), "v"(val), "s"(mask));
    return d;
}

// what the lanes of a wave know about their reads' first k-mer: `live` -- the lanes still looking (a wave-wide mask in
// SGPRs: "is any lane still looking" is a scalar compare), `group` -- per lane, the 4-character group of the first hit
struct S2Hits {
    unsigned long long live;
    int group;
};
// f != 0 in a lane: a k-mer that counts ended inside group g (wave-uniform)
__device__ __forceinline__ void s2_note(S2Hits& h, const uint32_t f, const int g) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(f != 0);
    const unsigned long long nw = m & h.live;
    if (nw) {
        h.group = s2_pick(h.group, g, nw);
        h.live &= ~m;
    }
}

// One chunk of 16 characters at positions pos .. pos+15 (characters past the read's end are NUL): the lead words and
// the first NA T-words advance over its four groups.  GUARD: the chunk may reach past the read's end -- groups that start
// there are skipped (wave-uniform).  LDS latency is hidden one step ahead, not more (registers: the kernel lives on its
// fourth wave per SIMD): the T-word masks of a group are requested in front of the group's lead step, the lead masks
// of the NEXT group in front of the group's T-word step.
template <int NL, int NT, int NA, bool GUARD>
__device__ __forceinline__ void s2_chunk(const S2Words<NL, NT>& K, uint32_t (&RL)[NL > 0 ? NL : 1],
                                         uint32_t (&RT)[NT > 0 ? NT : 1], const s2_u32x4 cw, const int pos, const int n,
                                         S2Hits& hits, const unsigned shv) {
    typedef S2Layout<NL, NT> LY;
    const unsigned w[4] = {cw.x, cw.y, cw.z, cw.w};
    constexpr int NLm = NL > 0 ? NL : 1, NAm = NA > 0 ? NA : 1;
    // lead masks: two sets, used alternately, so that a group's masks are requested a whole group ahead
    unsigned la[2][4];
    uint32_t mk[2][4][NLm];
    auto lead_loads = [&](int g) {
        s2_addr4(la[g & 1], w[g], shv);
        if constexpr (NL > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s2_read_entry<NL, LY::NLP>(mk[g & 1][i], K.lead + (3 - i) * LY::LEAD_TABLE + la[g & 1][i]);
        }
    };
    // the found masks of the chunk's groups sit 4 entries apart, group 3's lowest: one address register per chunk
    const unsigned char* const fm3 = K.found + s2_found_index(n, pos + 15) * (LY::NTP * 4);
    lead_loads(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (GUARD && pos + 4 * g >= n) continue;                 // (the masks requested for it are never looked at)
        const bool has_next = g + 1 < 4 && !(GUARD && pos + 4 * (g + 1) >= n);
        uint32_t fg = 0;
        uint32_t tm[4][NAm], fm[NAm];
        if constexpr (NA > 0) {
            // a T-word entry is twice (NTP = 2 NLP) or as wide as a lead entry: its offset is the lead offset, doubled
            // by an add (2 issue cycles; the compiler would make it a shift: 4)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned ta;
                if constexpr (LY::TAIL_SHIFT == LY::LEAD_SHIFT + 1) asm("v_add_u32 %0, %1, %1" : "=v"(ta) : "v"(la[g & 1][i]));
                else if constexpr (LY::TAIL_SHIFT == LY::LEAD_SHIFT) ta = la[g & 1][i];
                else ta = la[g & 1][i] << (LY::TAIL_SHIFT - LY::LEAD_SHIFT);
                s2_read_entry<NA, LY::NTP>(tm[i], K.tail + (3 - i) * LY::TAIL_TABLE + ta);
            }
            s2_read_entry<NA, LY::NTP>(fm, fm3 + (3 - g) * 4 * (LY::NTP * 4));
        } else {
            if (has_next) lead_loads(g + 1);
        }
        if constexpr (NL > 0) {
#pragma unroll
            for (int l = 0; l < NL; ++l)
                s2_lead_step(RL[l], fg, K.l_init4[l], K.l_found[l], mk[g & 1][0][l], mk[g & 1][1][l], mk[g & 1][2][l], mk[g & 1][3][l]);
        }
        if constexpr (NA > 0) {
            if (has_next) lead_loads(g + 1);
#pragma unroll
            for (int t = 0; t < NA; ++t) s2_tail_step(RT[t], fg, K.t_init4[t], fm[t], tm[0][t], tm[1][t], tm[2][t], tm[3][t]);
        }
        s2_note(hits, fg, (pos >> 2) + g);
        __builtin_amdgcn_sched_barrier(0);
    }
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
__device__ __forceinline__ void s2_clear_rows(int32_t* out6, int32_t* best, const int64_t base, const int cnt, const int lane) {
    if (best) {
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(best + base), 0, cnt * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(0xFFFFFFFFu, rb, lane * 4, 0, 0);          // lanes >= cnt: out of range, dropped
    }
    int32_t* const o = out6 + base * 6;
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

typedef const __attribute__((address_space(4))) FilterArgs* s2_kernarg_ptr;
__device__ __forceinline__ S2Out s2_out_args() {
    s2_kernarg_ptr kp = (s2_kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));                                        // keeps the loads where the values are used
    S2Out o;
    o.present = kp->present; o.status = kp->status; o.queue = kp->queue; o.queue_count = kp->queue_count;
    o.queue_keys = kp->queue_keys;
    return o;
}

template <int NL, int NT, bool BUF>
__global__ __launch_bounds__(S2_WAVES * WAVE) void k_filter_stream2(FilterArgs a) {
    typedef S2Layout<NL, NT> LY;
    constexpr int TILE = S2_TILE, SUBS = TILE / WAVE / S2_WAVES;
    // ONE static object, the tables first: they sit below 64 KB and an entry is read with
    // "ds_read_b64 v, v_entry offset:TABLE" -- the entry offset is the whole address computation of a character.
    // Everything is static: every LDS address is an immediate, none lives in a register.
    struct __attribute__((aligned(16))) S2Lds {
        uint32_t lead[4 * CAH_TABLE_CHARS * LY::NLP];
        uint32_t tail[4 * CAH_TABLE_CHARS * LY::NTP];
        uint32_t found[CAH_TW_DIST_LEN * LY::NTP];
        unsigned char slot[S2_WAVES * WAVE * S2_ROW];
        uint16_t idx[TILE];                                             // the tile's survivors: tile-relative read index
        uint8_t key[TILE];                                              // ... and key
        unsigned hist[CAH_QUEUE_BINS], cursor[CAH_QUEUE_BINS];
        unsigned long long qbase;
        unsigned count;
        unsigned scratch[8];
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
    unsigned shv = LY::LEAD_SHIFT;                                      // the SDWA shift amount wants a register
    asm volatile("" : "+v"(shv));

    // Copy plan.  A read has U = ceil(n / 16) units (the last one runs into the next read -- masked when used):
    // H1 = ceil(U / 2) in the first half-row, H2 = U - H1 in the second.  Load k < H1 takes the first-half unit
    // u = 64 k + lane = unit u % H1 of read u / H1, load 5 + k (k < H2) second-half unit u = unit H1 + u % H2 of read
    // u / H2.  Byte offset within the piece: r * n + 16 (c [+ H1]) = r * (n - 16 H) + 16 u [+ 16 H1]; in the slot:
    // r * 80 + 16 c = r * (80 - 16 H) + 16 u (16 u when H = 5).  u / H as a multiply: exact for u < 320, H <= 5
    // (tests/test_host_logic.py checks it exhaustively).
    const int U = (n + 15) >> 4;
    const int H1 = (U + 1) >> 1, H2 = U - H1;                           // <= 5 each
    const unsigned magic1 = (65536u + (unsigned)H1 - 1u) / (unsigned)H1;
    const unsigned magic2 = H2 ? (65536u + (unsigned)H2 - 1u) / (unsigned)H2 : 0u;
    // read index of unit 64 k + lane of a half with H units per read (recomputed where used: a register per unit
    // would cost the kernel its fourth wave per SIMD)
    auto unit_r = [&](int k, unsigned magic) -> int {
        unsigned ln = (unsigned)lane;
        asm volatile("" : "+v"(ln));
        return (int)(__umul24((unsigned)(k * WAVE) + ln, magic) >> 16);
    };
    // first read of this wave's piece `it`: sub-tile wave + 16 (it % SUBS) of tile blockIdx.x + (it / SUBS) gridDim.x;
    // int64: the pieces behind the last one lie beyond the batch, possibly beyond 2^31
    auto piece_base = [&](int it) -> int64_t {
        return ((int64_t)blockIdx.x + (int64_t)(it / SUBS) * (int64_t)gridDim.x) * TILE + (wave + S2_WAVES * (it % SUBS)) * WAVE;
    };
    s2_u32x4 pre[2 * S2_HALF];
    const uint8_t* const batch0 = a.seqs + first;
    // the units of the piece starting at read `base`, into VGPRs.  Nothing outside the batch is touched: a unit
    // that would run past the batch's last byte is fetched as the 16 bytes that END there and shifted down.
    auto prefetch = [&](int64_t base) {
        const int64_t left = n_reads - base;                            // wave-uniform
        if (left <= 0) return;
        const int64_t pbyte = base * (int64_t)n;                        // the piece's first byte within the batch
        const uint8_t* const src = batch0 + pbyte;
        if (left >= WAVE && pbyte + (int64_t)WAVE * n + 16 <= total) {
            // a whole piece with 16 bytes of the batch behind it (all but the last pieces): no lane needs a check
            if constexpr (BUF) {
                // buffer loads: the piece's first byte is the (scalar) base of the resource, a lane's offset ONE register
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7FFFFFFF, 0x00020000);
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
    // the chunk at `pos`: characters past the read's end (the next read's, or stale) become NUL
    auto finish = [&](s2_u32x4 v, int pos) -> s2_u32x4 {
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
    const bool clear = a.clear_out6 != nullptr && a.present == nullptr;

    int it = 0;
    prefetch(piece_base(0));
#pragma unroll 1
    for (int kt = 0;; ++kt) {
        const int64_t tile_base64 = ((int64_t)blockIdx.x + (int64_t)kt * (int64_t)gridDim.x) * TILE;
        if (tile_base64 >= n_reads) break;                              // block-uniform
        const int tile_base = (int)tile_base64;
        __syncthreads();
        if (threadIdx.x == 0) s_lds.count = 0;
        for (int i = threadIdx.x; i < CAH_QUEUE_BINS; i += blockDim.x) { s_lds.hist[i] = 0; s_lds.cursor[i] = 0; }
        __syncthreads();

#pragma unroll 1
        for (int j = 0; j < SUBS; ++j, ++it) {
            const int base = tile_base + (wave + S2_WAVES * j) * WAVE;  // < 2^31 + 2^13: compared as unsigned
            const bool more = (unsigned)base < (unsigned)n_reads;       // wave-uniform
            S2Hits hits;                                                // lanes still looking for a first k-mer
            hits.live = more ? (n_reads - base >= WAVE ? ~0ull : (1ull << (n_reads - base)) - 1ull) : 0ull;
            hits.group = -1;
            unsigned seen = 0;
            uint32_t RL[NL > 0 ? NL : 1], RT[NT > 0 ? NT : 1];
#pragma unroll
            for (int w = 0; w < (NL > 0 ? NL : 1); ++w) RL[w] = 0;
#pragma unroll
            for (int w = 0; w < (NT > 0 ? NT : 1); ++w) RT[w] = 0;
            s2_u32x4 cur = (s2_u32x4)(0u);
#pragma unroll 1
            for (int ph = 0; ph < 2; ++ph) {
                const int H = ph ? H2 : H1;
                const bool alive = H > 0 && hits.live != 0;                 // wave-uniform
                if (alive) {
                    if (ph == 0) to_slot(std::integral_constant<int, 0>{}); else to_slot(std::integral_constant<int, 1>{});
                    cur = *reinterpret_cast<const s2_u32x4*>(row);
                }
                if (!alive) continue;
                const int pos0 = ph ? 16 * H1 : 0;
#pragma unroll 1
                for (int c = 0; c < H; ++c) {
                    if (c > 0 && hits.live == 0) break;
                    const int pos = pos0 + 16 * c;
                    s2_u32x4 nxt = (s2_u32x4)(0u);                      // requested now, looked at a chunk later
                    if (c + 1 < H) nxt = *reinterpret_cast<const s2_u32x4*>(row + 16 * (c + 1));
                    const s2_u32x4 cw = finish(cur, pos);
                    seen = s2_or3(seen, cw.x, cw.y);
                    seen = s2_or3(seen, cw.z, cw.w);
                    if (pos + 16 > n) {
                        // the read's last chunk: every T-word is at work, groups past the end are skipped
                        s2_chunk<NL, NT, NT, true>(K, RL, RT, cw, pos, n, hits, shv);
                    } else {
                        const int na = s2_active_tw(tspan, NT, n, pos);
#define S2_CASE(NA)                                                                                                     \
                        if constexpr (NA <= NT) { if (na == NA) s2_chunk<NL, NT, NA, false>(K, RL, RT, cw, pos, n, hits, shv); }
                        S2_CASE(0) S2_CASE(1) S2_CASE(2) S2_CASE(3) S2_CASE(4)
#undef S2_CASE
                    }
                    cur = nxt;
                }
            }
            // The next piece's loads go out now, not earlier: through the second half -- the T-words' chunks -- no copy
            // register is live, through the first only the second half's (a fourth wave per SIMD needs that; the
            // other three cover the wait at the next piece's start).  The stores of the result rows in front of them:
            // loads and stores share the in-order vmcnt counter.
            if (clear && more)
                s2_clear_rows(a.clear_out6, a.clear_best, (int64_t)base, n_reads - base < WAVE ? n_reads - base : WAVE, lane);
            prefetch(piece_base(it + 1));
            if (!more) continue;                                        // wave-uniform; nothing left in this tile
            const bool invalid = (seen & 0x80808080u) != 0;
            const S2Out o = s2_out_args();
            lean_emit(o, (int64_t)base + lane, (int64_t)tile_base, (unsigned)(base + lane) < (unsigned)n_reads, hits.group >= 0,
                      invalid, hits.group << CAH_KEY_SHIFT, s_lds.idx, s_lds.key, s_lds.hist, s_lds.count);
        }

        const S2Out o = s2_out_args();
        if (!o.present) {
            __syncthreads();
            flush_tile_queue(o, (int64_t)tile_base, s_lds.idx, s_lds.key, s_lds.hist, s_lds.cursor, s_lds.count, s_lds.scratch,
                             s_lds.qbase);
        }
    }
}

// classes <lead words, T-words>
hipError_t launch_filter_stream2(const FilterArgs& a_in, int mode, int n_lead, int n_tw, int n_cus, hipStream_t s) {
    FilterArgs a = a_in;
    if (mode != 0) a.present = nullptr;
    a.stream_n_lo = 1; a.stream_n_hi = S2_MAX_LEN;
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
    if (n_lead <= 1 && n_tw <= 2) S2_LAUNCH(1, 2);
    else if (n_lead <= 2 && n_tw <= 4) S2_LAUNCH(2, 4);
    else return hipErrorInvalidValue;
#undef S2_LAUNCH
    return hipGetLastError();
}

int stream2_max_len() { return S2_MAX_LEN; }
bool stream2_class_ok(int n_lead, int n_tw) { return n_lead <= 2 && n_tw <= 4; }
