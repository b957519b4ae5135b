// kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the adapter-matching hot path.
//
//   k_filter      KmerFinder.kmers_present          reference _kmer_finder.pyx:170-257
//   k_dp<ROWS>    Aligner.locate                    reference _align.pyx:298-587
//   k_comparer    Prefix/SuffixComparer.locate      reference _align.pyx:651-714
//   k_validate    "only ASCII" precondition         reference _align.pyx:44-45
//
// Execution model: ONE READ PER LANE, 64 reads per wavefront.  The DP column of the short
// adapter (m <= 64) lives entirely in VGPRs (two registers per row: cost and a packed
// score|origin payload), the row loop is fully unrolled and predicated per lane by the
// Ukkonen band limit `last`; the compiler turns the nested predicates into exec masks and
// skips row blocks no lane needs (s_cbranch_execz).  Per-character adapter match bitsets
// (64-bit, one bit per adapter row) come from a 1 KiB LDS table, so the inner loop has no
// byte compares at all.  Integer only -- no MFMA (this is a min/compare recurrence, not a
// contraction).
//
// Bit-exactness includes the reference's accidental behaviour: cells outside the band keep
// stale values, the last-column scan compares against a stale `origin`, tie-breaks are
// mismatch >= deletion >= insertion.  See DESIGN.md "Exactness notes".
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cah_device.h"
#include "kernels.h"
#include "back_scan.h"
#include "dev_common.h"
#include "filter_common.h"
#define CAH_M2_NO_HOST
#include "multi2.h"          // (k_dp_packed<.., true>: the corner check of the streaming form's tail pairs)

#ifndef CAH_DEQUEUE
#define CAH_DEQUEUE 8             // sub-batches of 64 work items a wave takes per atomic (k_dp, k_comparer, k_anchored_exact)
#endif

#ifndef CAH_SCHED_ROWS
#define CAH_SCHED_ROWS 1
#endif

__device__ __forceinline__ int pack_cell(int origin, int score) {
    return ((origin + CAH_ORIGIN_BIAS) << CAH_SCORE_BITS) + (score + CAH_SCORE_BIAS);
}
__device__ __forceinline__ int cell_origin(int p) {
    return (int)((unsigned)p >> CAH_SCORE_BITS) - CAH_ORIGIN_BIAS;
}
__device__ __forceinline__ int cell_score(int p) {
    return (p & ((1 << CAH_SCORE_BITS) - 1)) - CAH_SCORE_BIAS;
}

// =============================================================================================
// k_filter: multi-pattern shift-and prefilter (KmerFinder.kmers_present).
//
// One read per lane, ONE pass over the read: characters arrive 16 per global load (next chunk
// requested before the current one is consumed) and every packed word whose window overlaps
// the chunk is advanced on it, its shift-and state R and an OR-accumulator living in VGPRs
// (up to FILTER_SLOTS words at a time).  Because (OR_t R_t) & found == OR_t (R_t & found), the
// hit test is done once per chunk instead of per character.  Character masks (1 KiB per word)
// are in LDS.  Windows follow _kmer_finder.pyx:188-204 per read; a positive stop beyond the
// read end is clamped (the reference reads out of bounds there).
// MODE 0: write present[r].   MODE 1: append survivors to the DP work queue (one ballot +
// popcount + a single atomic per wave) together with a key = chunk index of the first hit,
// and count keys per bin so that the queue can be ordered by approximate adapter position.
// =============================================================================================
#include <type_traits>
#include <algorithm>
#include <cstdlib>


__device__ __forceinline__ void word_window(const int wstart, const int wstop, const int n,
                                            int& ws, int& we) {
    int start = wstart, stop = wstop;
    bool empty = false;
    if (start < 0) { start += n; if (start < 0) start = 0; }
    else if (start > n) empty = true;
    if (stop < 0) { stop += n; if (stop <= 0) empty = true; }
    else if (stop == 0) stop = n;
    if (stop > n) stop = n;
    if (stop <= start) empty = true;
    ws = empty ? 0 : start;
    we = empty ? 0 : stop;
}

template <bool MASKED, typename W, typename T>
__device__ __forceinline__ void filter_word_chunk(const Chunk& ck, const T* tbl, const W init,
                                                  W& R, W& acc, W (&gg)[3], const W found,
                                                  const int lo, const int hi, const bool act) {
    // lo/hi: first / one-past-last chunk character (0..16) inside this word's window.
    // The 16 characters are handled in four groups of four; a group no lane needs is skipped
    // (short suffix windows such as the 3- and 4-character overlap searches touch one group).
    // R << 1 is written R + R: on gfx950 v_add_u32 issues in 2 cycles, v_lshlrev_b32 in 4.
    // gg[g] collects (over all words of the chunk) the found bits seen up to group g: lets the
    // caller name the 4-column group of a first hit without a second pass.
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        if (MASKED) {
            if (!__any(act && lo < 4 * g4 + 4 && hi > 4 * g4)) continue;    // acc unchanged: gg[g4] may stay behind
        }
        if (act) {
#pragma unroll
            for (int t = 4 * g4; t < 4 * g4 + 4; ++t) {
                const unsigned ch = chunk_byte(ck, t) & (CAH_TABLE_CHARS - 1);
                W mk = (W)tbl[ch];
                if (MASKED) mk = (t >= lo && t < hi) ? mk : (W)0;
                if constexpr (sizeof(W) == 4) {
                    // R << 1 as an explicit add: the compiler canonicalises R + R into v_lshlrev_b32
                    // (4 issue cycles on gfx950), v_add_u32 takes 2
                    unsigned dbl;
                    asm("v_add_u32 %0, %1, %1" : "=v"(dbl) : "v"((unsigned)R));
                    R = (W)((dbl | (unsigned)init) & (unsigned)mk);
                } else {
                    R = ((R + R) | init) & mk;
                }
                acc |= R;
            }
            if (g4 < 3) gg[g4] |= acc & found;
        }
    }
}

// Work is handed out in TILES of FILTER_TILE reads per workgroup (one global atomic per tile:
// a single hot counter sustains only ~90 atomics/us on this chip, so per-wave dequeues and
// per-wave queue appends would bound the whole kernel).  Survivors of a tile are collected in
// LDS, ordered by key with an LDS counting sort and appended to the global queue as one run.
// The DP kernel takes 64 consecutive queue entries per wave, so its lanes hold reads whose
// adapters sit at similar columns and their Ukkonen bands widen and narrow together.
#ifndef FILTER_TILE
#define FILTER_TILE 8192
#endif
#define FILTER_WAVES 4

#ifndef CAH_FILTER_WAVES
#define CAH_FILTER_WAVES 4
#endif

// NARROW: every packed word of the plan fits 32 bits (cah_plan_create packs k-mers of <= 32
// characters that way): shift-and state, accumulators and LDS tables are 32-bit, which halves
// the VALU work per character and word.
template <int MODE, bool LDS_TABLES, bool NARROW>
__global__ __launch_bounds__(256, CAH_FILTER_WAVES) void k_filter(FilterArgs a) {
    typedef typename std::conditional<NARROW, uint32_t, uint64_t>::type word_t;
    constexpr int SLOTS = NARROW ? CAH_FILTER_SLOTS_NARROW : CAH_FILTER_SLOTS;
    // all LDS is carved from the dynamic region (16-byte aligned offsets; a static __shared__
    // in front of it could misalign the 8-byte table reads):
    //   [tables: n_words KiB] [s_idx: 16 KiB, 16-bit tile-relative] [s_key: 8 KiB] [s_hist] [s_cursor] [scalars]
    //   [per-word init/found masks and windows: 24 B x n_words]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const CahKmerWord* words = a.words;
    const int n_words = a.n_words;
    word_t* s_mask = reinterpret_cast<word_t*>(smem);
    unsigned char* sp = smem + (LDS_TABLES ? (size_t)n_words * CAH_TABLE_CHARS * sizeof(word_t) : 0);
    uint16_t* s_idx = reinterpret_cast<uint16_t*>(sp);           sp += FILTER_TILE * sizeof(uint16_t);   // tile-relative
    uint8_t* s_key = sp;                                         sp += FILTER_TILE;
    unsigned* s_hist = reinterpret_cast<unsigned*>(sp);          sp += CAH_QUEUE_BINS * sizeof(unsigned);
    unsigned* s_cursor = reinterpret_cast<unsigned*>(sp);        sp += CAH_QUEUE_BINS * sizeof(unsigned);
    unsigned long long& s_qbase = *reinterpret_cast<unsigned long long*>(sp);
    long long& s_tile = *reinterpret_cast<long long*>(sp + 8);
    unsigned& s_count = *reinterpret_cast<unsigned*>(sp + 16);
    sp += 64;
    // word parameters are read from LDS inside the loops (the compiler would otherwise re-load
    // them from global memory per chunk: stores to the queue may alias as far as it knows)
    uint64_t* s_winit = reinterpret_cast<uint64_t*>(sp);         sp += (size_t)n_words * sizeof(uint64_t);
    uint64_t* s_wfound = reinterpret_cast<uint64_t*>(sp);        sp += (size_t)n_words * sizeof(uint64_t);
    int* s_wstart = reinterpret_cast<int*>(sp);                  sp += (size_t)n_words * sizeof(int);
    int* s_wstop = reinterpret_cast<int*>(sp);
    for (int i = threadIdx.x; i < n_words; i += blockDim.x) {
        s_winit[i] = words[i].init_mask;
        s_wfound[i] = words[i].found_mask;
        const int64_t st = words[i].start, sp_ = words[i].stop;      // clamp to int (reads are <= 1e6 long)
        s_wstart[i] = (int)(st < -(1 << 30) ? -(1 << 30) : (st > (1 << 30) ? (1 << 30) : st));
        s_wstop[i] = (int)(sp_ < -(1 << 30) ? -(1 << 30) : (sp_ > (1 << 30) ? (1 << 30) : sp_));
    }
    if (LDS_TABLES) {
        for (int i = threadIdx.x; i < n_words * CAH_TABLE_CHARS; i += blockDim.x)
            s_mask[i] = (word_t)words[i / CAH_TABLE_CHARS].mask[i % CAH_TABLE_CHARS];
    }
    const int lane = wave_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform, and said so

    for (;;) {
        __syncthreads();                                 // previous tile fully flushed
        if (threadIdx.x == 0) {
            s_tile = (long long)atomicAdd(a.work_counter, (unsigned long long)FILTER_TILE);
            s_count = 0;
        }
        for (int i = threadIdx.x; i < CAH_QUEUE_BINS; i += blockDim.x) { s_hist[i] = 0; s_cursor[i] = 0; }
        __syncthreads();
        const int64_t tile_base = s_tile;
        if (tile_base >= a.n_reads) break;

        for (int sub = wave; sub < FILTER_TILE / WAVE; sub += FILTER_WAVES) {
            const int64_t base = tile_base + (int64_t)sub * WAVE;
            if (base >= a.n_reads) break;
            const int64_t r = base + lane;
            const bool valid = r < a.n_reads;
            if (MODE == 1 && a.clear_out6)
                clear_rows(a.clear_out6, a.clear_best, base, (int)(a.n_reads - base < WAVE ? a.n_reads - base : WAVE), lane);
            int64_t off = 0, n64 = 0;
            if (valid) read_extent(a.offsets, a.lens, a.uniform_first, a.uniform_len, r, off, n64);
            const uint8_t* q = a.seqs + off;
            bool hit = false, invalid = false;
            if (n64 > a.max_read_len) { invalid = true; n64 = 0; }
            const int n = (int)n64;
            int hit_pos = 0;
            unsigned seen = 0;

            for (int g = 0; g < n_words; g += SLOTS) {
                if (!__any(valid && !hit)) break;
                // this lane's window of every word of the group, and their union
                int wsv[SLOTS], wev[SLOTS];
                int lo = n, hi = 0;
    #pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    wsv[s] = 0; wev[s] = 0;
                    if (g + s < n_words) {
                        word_window(s_wstart[g + s], s_wstop[g + s], n, wsv[s], wev[s]);
                        if (wev[s] > wsv[s]) { lo = min(lo, wsv[s]); hi = max(hi, wev[s]); }
                    }
                }
                if (!valid) { lo = 0; hi = 0; }
                word_t R[SLOTS], acc[SLOTS];
    #pragma unroll
                for (int s = 0; s < SLOTS; ++s) { R[s] = 0; acc[s] = 0; }

                int pos = lo;
                Chunk cur = load_chunk(q, pos, n, hi);
                for (;;) {
                    const bool live = !hit && pos < hi;
                    if (!__any(live)) break;
                    const Chunk nxt = load_chunk(q, pos + 16, n, live ? hi : 0);      // one chunk ahead
                    seen |= cur.w[0] | cur.w[1] | cur.w[2] | cur.w[3];
                    word_t any_found = 0;
                    word_t gg[3] = {0, 0, 0};             // found bits after 4 / 8 / 12 characters of the chunk
    #pragma unroll
                    for (int s = 0; s < SLOTS; ++s) {
                        if (g + s >= n_words) break;                     // wave-uniform
                        const CahKmerWord* wd = words + (g + s);
                        const int ws = wsv[s], we = wev[s];
                        const bool act = live && pos < we && pos + 16 > ws;
                        if (!__any(act)) continue;
                        const word_t init = (word_t)s_winit[g + s];
                        // masking is needed where a window starts inside the chunk or stops before
                        // the end of the read inside it (beyond the read end the chunk is NUL-padded)
                        const bool partial = act && (ws > pos || (we < pos + 16 && we < n));
                        const word_t fnd = (word_t)s_wfound[g + s];
                        if (LDS_TABLES) {
                            const word_t* tbl = s_mask + (g + s) * CAH_TABLE_CHARS;
                            if (__any(partial)) filter_word_chunk<true, word_t>(cur, tbl, init, R[s], acc[s], gg, fnd, ws - pos, we - pos, act);
                            else filter_word_chunk<false, word_t>(cur, tbl, init, R[s], acc[s], gg, fnd, 0, 16, act);
                        } else {
                            const uint64_t* tbl = wd->mask;      // plans with too many words: tables stay in HBM/L2
                            if (__any(partial)) filter_word_chunk<true, word_t>(cur, tbl, init, R[s], acc[s], gg, fnd, ws - pos, we - pos, act);
                            else filter_word_chunk<false, word_t>(cur, tbl, init, R[s], acc[s], gg, fnd, 0, 16, act);
                        }
                        any_found |= acc[s] & fnd;
                    }
                    if (live && any_found != 0) {
                        // a lane leaves at its first hit, so no found bit was set before this chunk:
                        // the first group accumulator that shows one names the 4-column group
                        hit = true;
                        hit_pos = pos + (gg[0] ? 0 : gg[1] ? 4 : gg[2] ? 8 : 12);
                    }
                    pos += 16;
                    cur = nxt;
                }
            }
            if (seen & 0x80808080u) invalid = true;

            if (MODE == 0) {
                if (valid) a.present[r] = invalid ? (uint8_t)2 : (hit ? (uint8_t)1 : (uint8_t)0);
            } else {
                if (valid && invalid) a.status[r] = 2;
                const bool push = valid && hit && !invalid;
                const unsigned long long bal = __ballot(push);
                if (bal) {
                    unsigned slot = 0;
                    if (lane == 0) slot = atomicAdd(&s_count, (unsigned)__popcll(bal));    // LDS atomic
                    slot = __builtin_amdgcn_readfirstlane(slot);
                    if (push) {
                        const int e = (int)slot + __popcll(bal & ((1ull << lane) - 1ull));
                        const int key = min(hit_pos >> CAH_KEY_SHIFT, CAH_QUEUE_BINS - 1);
                        s_idx[e] = (uint16_t)(r - tile_base);
                        s_key[e] = (uint8_t)key;
                        atomicAdd(&s_hist[key], 1u);
                    }
                }
            }
        }

        if (MODE == 1) {
            // flush the tile: counting sort by key in LDS, one global atomic for the whole run
            __syncthreads();
            const unsigned count = s_count;
            if (threadIdx.x == 0) {
                unsigned run = 0;
                for (int bkt = 0; bkt < CAH_QUEUE_BINS; ++bkt) { const unsigned c = s_hist[bkt]; s_hist[bkt] = run; run += c; }
                s_qbase = count ? atomicAdd(a.queue_count, (unsigned long long)count) : 0ull;
            }
            __syncthreads();
            const unsigned long long qbase = s_qbase;
            for (unsigned e = threadIdx.x; e < count; e += blockDim.x) {
                const unsigned key = s_key[e];
                const unsigned p = s_hist[key] + atomicAdd(&s_cursor[key], 1u);
                a.queue[qbase + p] = (int32_t)(tile_base + s_idx[e]);
                a.queue_keys[qbase + p] = (uint8_t)key;
            }
        }
    }
}

// =============================================================================================
// k_filter_lean: the prefilter for the common shape -- a 3' adapter's search sets (whole read +
// "last L characters", see CahLeanFilter) on a batch whose reads all have the same length n.
// Everything that k_filter keeps per lane (windows, activity masks, per-character masks) is a
// wave-uniform scalar here; tail k-mers of all window lengths share words through gated start bits.
// Same outputs as k_filter (present[] / key-ordered survivor queue), same key semantics.
// =============================================================================================
__global__ void k_uniform_check(const int64_t* offsets, int64_t n_reads, int64_t max_read_len, unsigned long long* flag) {
    // *flag != 0  <=>  the reads do NOT all have the length offsets[1] - offsets[0] (or it is out of range)
    const int64_t n = offsets[1] - offsets[0];
    bool bad = n < 0 || n > max_read_len;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads && !bad; r += (int64_t)gridDim.x * blockDim.x)
        bad = offsets[r + 1] - offsets[r] != n;
    if (__any(bad) && wave_lane() == 0) atomicOr(flag, 1ull);
}

#ifndef LEAN_TILE
#define LEAN_TILE 8192
#endif
#ifndef LEAN_WAVES
#define LEAN_WAVES 4               // measured: 2 -> 2.56 ms, 3 -> 2.03, 4 -> 1.85, 5 -> 2.01, 8 -> 2.61 (20 M reads)
#endif

// Constant-address-space view of plan tables that no kernel writes: a wave-uniform index then becomes a scalar
// load (s_load_dword through the scalar cache) instead of a per-lane LDS/VMEM access.
typedef const __attribute__((address_space(4))) uint32_t* cah_const_u32;

// ---------------------------------------------------------------------------------------------
// The lean prefilter's word machinery, shared by k_filter_lean (per-lane global loads) and k_filter_stream
// (reads staged through LDS).  Class <NL, NG>: NL lead slots, NG gated slots, every loop bound a compile-time
// constant -- straight-line code, no per-word branches (slots a plan does not use have empty masks and closed
// gates: they cost instructions, never results).
//   * character masks: 256-entry tables in a STATIC LDS allocation (its address is a compile-time constant that
//     folds into the ds_read offsets; a table offset is "byte << 2", one SDWA instruction per character; bytes
//     >= 0x80 find zeros), lead tables first
//   * gates of equally long reads are scalar loads (wave-uniform index), of ragged batches LDS gathers
//   * LDS latency is hidden by software pipelining: the masks of the NEXT eight characters' lead words are
//     requested before the current eight are matched (the kernels were waiting, not computing: 55 % of the wave
//     cycles in s_waitcnt with one batch of ds_reads per word and chunk)
// ---------------------------------------------------------------------------------------------
template <int NL, int NG>
struct LeanWords {
    int n_tail, tail_span, head_span;
    uint32_t l_init[NL], l_init4[NL], l_found[NL], g_found[NG];     // l_init4 = S(3): start bits of the last four positions
    int g_open[NG];              // equally long reads: the 8-character group from which gated word g has work (a tail word's
                                 // start bits stay closed until gated_span characters are left; its state is 0 until then)
    const unsigned char* s_lead;                                 // LDS: lead table(s), then gated table (LeanLayout)
    const uint32_t* s_ginit;                                     // LDS: the start-bit gates of the NG gated words
};

// Mask tables in LDS: ONE entry per byte value holds the masks of all lead words (padded to 1, 2 or 4 words), a
// second table the masks of all gated words (2, 4 or 8 words), so that a character costs one wide LDS read per
// table (ds_read_b64 / b128 move twice the bytes per LDS cycle of ds_read_b32: the kernels were LDS-bound with
// one b32 read per word and character).  128 entries: a byte >= 0x80 reads whatever follows the table -- its
// read is flagged invalid and its matches are never used.
__host__ __device__ constexpr int lean_pow2(int n) { return n <= 1 ? 1 : (n <= 2 ? 2 : (n <= 4 ? 4 : 8)); }
__host__ __device__ constexpr int lean_log2(int p) { return p == 1 ? 0 : (p == 2 ? 1 : (p == 4 ? 2 : 3)); }
// DL > 0 (delay bits, see lean_lead8): the lead words advance FOUR characters per step,
//   R4 = ((R << 4) | S3) & T3[c1] & T2[c2] & T1[c3] & T0[c4],   Ts[c] = (M[c] << s) | S(s-1),
//   S(k) = START | START << 1 | .. | START << k
// (four single steps R' = ((R << 1) | START) & M[c] written out: (x & a) | s == (x | s) & (a | s), and an OR
// distributes over the ANDs), so the lead table comes four times, one per position inside a group of four.  The
// bits a shift carries across a k-mer's field border land on the next field's lowest s bits, which S(s-1) sets
// anyway (every field is at least 1 + DL = 4 bits long).
template <int DL, int NL, int NG> struct LeanLayout {
    static constexpr int NLP = lean_pow2(NL), NGP = lean_pow2(NG);
    static constexpr int LEAD_SHIFT = 2 + lean_log2(NLP), GATED_SHIFT = 2 + lean_log2(NGP);
    static constexpr int LEAD_TABLE = CAH_TABLE_CHARS * NLP * 4;                 // bytes of one lead table
    static constexpr int LEAD_BYTES = LEAD_TABLE * (DL > 0 ? 4 : 1);             // T0 = M, T1, T2, T3
    static constexpr int GATED_BYTES = CAH_TABLE_CHARS * NGP * 4;
    static constexpr int WORDS = (LEAD_BYTES + GATED_BYTES) / 4;                 // uint32 words of all tables
};

template <int DL, int NL, int NG>
__device__ __forceinline__ void lean_tables_to_lds(const CahLeanFilter* lf, uint32_t* s_tab) {
    typedef LeanLayout<DL, NL, NG> LY;
    constexpr int LW = CAH_TABLE_CHARS * LY::NLP;                                // words of one lead table
    constexpr int NT = DL > 0 ? 4 : 1;                                           // lead tables T0 .. T3
    // one pass over the base masks writes all shifted copies (the one-read kernel fills the tables per call)
    for (int j = threadIdx.x; j < LW; j += blockDim.x) {
        const int c = j / LY::NLP, w = j % LY::NLP;
        uint32_t v = 0, init = 0;
        if (w < NL && w < lf->n_lead) {
            // the delay bits of a lead word pass EVERY byte (a k-mer end must survive until its group is checked)
            v = lf->lead_mask[w][c] | lf->lead_pass[w];
            init = lf->lead_init[w];
        }
        uint32_t fill = 0;
#pragma unroll
        for (int sh = 0; sh < NT; ++sh) {
            s_tab[sh * LW + j] = (w < NL && w < lf->n_lead) ? ((v << sh) | fill) : 0u;   // Ts = (M << s) | S(s-1)
            fill |= init << sh;
        }
    }
    for (int j = threadIdx.x; j < LY::GATED_BYTES / 4; j += blockDim.x) {
        const int c = j / LY::NGP, w = j % LY::NGP;
        s_tab[LY::LEAD_BYTES / 4 + j] = (w < NG && w < lf->n_gated) ? lf->gated_mask[w][c] : 0u;
    }
}

template <int NL, int NG>
__device__ __forceinline__ void lean_words_init(LeanWords<NL, NG>& L, const CahLeanFilter* lf, const uint32_t* s_tab,
                                                const uint32_t* s_ginit, const int n_uniform = -1) {
    L.n_tail = lf->n_tail; L.tail_span = lf->tail_span; L.head_span = lf->head_span;
#pragma unroll
    for (int w = 0; w < NG; ++w)
        L.g_open[w] = (n_uniform >= 0 && w < lf->n_tail) ? n_uniform - lf->gated_span[w] - 7 : -(1 << 30);
    // per-word constants live in registers: inside the loops the compiler would re-load anything read
    // through a pointer (the queue stores may alias as far as it knows)
#pragma unroll
    for (int w = 0; w < NL; ++w) {
        L.l_init[w] = w < lf->n_lead ? lf->lead_init[w] : 0u;
        L.l_init4[w] = L.l_init[w] | (L.l_init[w] << 1) | (L.l_init[w] << 2) | (L.l_init[w] << 3);
        L.l_found[w] = w < lf->n_lead ? lf->lead_found[w] : 0u;
    }
#pragma unroll
    for (int w = 0; w < NG; ++w) L.g_found[w] = w < lf->n_gated ? lf->gated_found[w] : 0u;
    L.s_lead = reinterpret_cast<const unsigned char*>(s_tab);
    L.s_ginit = s_ginit;
}

// per-lane state of a read
template <int NL, int NG>
struct LeanState {
    uint32_t RL[NL], accL[NL], RG[NG], accG[NG];
    uint32_t mk0[8][NL];                                         // lead masks of the current chunk's characters 0..7
};

// entry offsets ("byte << SHIFT": one SDWA instruction each) of the eight characters of two dwords
template <int SHIFT>
__device__ __forceinline__ void lean_addr8(unsigned (&ad)[8], unsigned w0, unsigned w1) {
#pragma unroll
    for (int t = 0; t < 4; ++t) { ad[t] = ((w0 >> (8 * t)) & 0xFFu) << SHIFT; ad[4 + t] = ((w1 >> (8 * t)) & 0xFFu) << SHIFT; }
}

// the first N words of a table entry with one LDS read (N = 1, 2, 4; 8 = two reads)
template <int N, int NP>
__device__ __forceinline__ void lean_read_entry(uint32_t (&out)[N], const unsigned char* p) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (NP == 1) {
        out[0] = *reinterpret_cast<const uint32_t*>(p);
    } else if constexpr (NP == 2) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(p);
        out[0] = v.x;
        if constexpr (N > 1) out[1] = v.y;
    } else {
        const u32x4 v = *reinterpret_cast<const u32x4*>(p);
        out[0] = v.x;
        if constexpr (N > 1) out[1] = v.y;
        if constexpr (N > 2) out[2] = v.z;
        if constexpr (N > 3) out[3] = v.w;
        if constexpr (NP == 8 && N > 4) {
            const u32x4 u = *reinterpret_cast<const u32x4*>(p + 16);
            out[4] = u.x;
            if constexpr (N > 5) out[5] = u.y;
            if constexpr (N > 6) out[6] = u.z;
            if constexpr (N > 7) out[7] = u.w;
        }
    }
}

template <int DL, int NL, int NG>
__device__ __forceinline__ void lean_issue_lead(const LeanWords<NL, NG>& L, uint32_t (&mk)[8][NL], const unsigned (&ad)[8]) {
    typedef LeanLayout<DL, NL, NG> LY;
#pragma unroll
    for (int t = 0; t < 8; ++t)                                  // DL > 0: the first character of a pair reads M1
        lean_read_entry<NL, LY::NLP>(mk[t], L.s_lead + (DL > 0 ? (3 - (t & 3)) * LY::LEAD_TABLE : 0) + ad[t]);
}

// Eight characters of the lead words; f4 / f8: found bits seen up to the 4th / 8th of them.
// DL == 0: an accumulator ORs every state (one more instruction per two characters and word).  DL == 3: every
// k-mer is followed by three delay bits that pass every byte, so a k-mer end is still visible three characters
// later and the state itself is tested once per 4-character group (l_found then covers end + delay bits).
template <int DL, int NL, int NG>
__device__ __forceinline__ void lean_lead8(const LeanWords<NL, NG>& L, LeanState<NL, NG>& S, const uint32_t (&mk)[8][NL],
                                           uint32_t& f4, uint32_t& f8) {
    if constexpr (DL > 0) {
        // four characters per step (see LeanLayout); the state is looked at after every one
#pragma unroll
        for (int l = 0; l < NL; ++l)
            S.RL[l] = ((S.RL[l] << 4) | L.l_init4[l]) & mk[0][l] & mk[1][l] & mk[2][l] & mk[3][l];
#pragma unroll
        for (int l = 0; l < NL; ++l) f4 |= S.RL[l] & L.l_found[l];
#pragma unroll
        for (int l = 0; l < NL; ++l)
            S.RL[l] = ((S.RL[l] << 4) | L.l_init4[l]) & mk[4][l] & mk[5][l] & mk[6][l] & mk[7][l];
#pragma unroll
        for (int l = 0; l < NL; ++l) f8 |= S.RL[l] & L.l_found[l];
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                unsigned dbl;
                asm("v_add_u32 %0, %1, %1" : "=v"(dbl) : "v"(S.RL[l]));      // R + R: the compiler makes it a 4-cycle shift
                S.RL[l] = (dbl | L.l_init[l]) & mk[t][l];
                S.accL[l] |= S.RL[l];
            }
            if (t == 3) {
#pragma unroll
                for (int l = 0; l < NL; ++l) f4 |= S.accL[l] & L.l_found[l];
            }
        }
#pragma unroll
        for (int l = 0; l < NL; ++l) f8 |= S.accL[l] & L.l_found[l];
    }
}

// Eight characters (positions p0 .. p0+7 of a read of length n) of the gated words, four at a time (the masks of
// four characters x NG words and their start-bit gates are requested together).  Only the START bits are gated: a
// k-mer that started inside its gate ends inside its window (tail sets: before the read's end, past which
// characters are NUL; head sets: the gate closes len - 1 before `stop`), and an end bit can only leak into the next
// k-mer's start after a genuine end has been recorded -- so every recorded end counts.
// The gates sit in LDS.  Equally long reads index them wave-uniformly: four consecutive gates of a word are two
// broadcast ds_read2_b32 (scalar loads would share the lgkmcnt counter with the LDS reads and return out of
// order, so waiting for them drains every prefetched mask; 48 gates per chunk do not fit the SGPRs either).
// Ragged batches gather them per lane.
template <bool UNIFORM, int DL, int NL, int NG>
__device__ __forceinline__ void lean_gated8(const LeanWords<NL, NG>& L, LeanState<NL, NG>& S, const unsigned (&ad)[8],
                                            const int p0, const int n, uint32_t& f4, uint32_t& f8) {
    // gate index: tail words run with the distance from the read end, head words with the position
    const int tail_base = CAH_GATE_ZERO - n + p0;              // scalar if UNIFORM, per lane otherwise
    // equally long reads: clamped once per call -- below CAH_GATE_PAD a tail table is constant (closed), from
    // CAH_LEAN_SPAN on a head table is, so the eight indices stay equivalent
    const int tail_u = max(tail_base, 0), head_u = min(p0, CAH_GATE_LEN - 8);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t mk[4][NG], gt[4][NG];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            lean_read_entry<NG, LeanLayout<DL, NL, NG>::NGP>(mk[t], L.s_lead + LeanLayout<DL, NL, NG>::LEAD_BYTES + ad[4 * h + t]);
        if constexpr (UNIFORM) {
            // a word is stepped only once one of its windows is near (wave-uniform: L.g_open)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (p0 >= L.g_open[g]) {
                    const uint32_t* gp = L.s_ginit + g * CAH_GATE_LEN + (g < L.n_tail ? tail_u : head_u) + 4 * h;
#pragma unroll
                    for (int t = 0; t < 4; ++t) gt[t][g] = gp[t];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        unsigned dbl;
                        asm("v_add_u32 %0, %1, %1" : "=v"(dbl) : "v"(S.RG[g]));
                        S.RG[g] = (dbl | gt[t][g]) & mk[t][g];
                        S.accG[g] |= S.RG[g];
                    }
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int it = min(max(tail_base + 4 * h + t, 0), CAH_GATE_LEN - 1);
                    const int ih = min(p0 + 4 * h + t, CAH_GATE_LEN - 1);
                    gt[t][g] = L.s_ginit[g * CAH_GATE_LEN + (g < L.n_tail ? it : ih)];
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    unsigned dbl;
                    asm("v_add_u32 %0, %1, %1" : "=v"(dbl) : "v"(S.RG[g]));
                    S.RG[g] = (dbl | gt[t][g]) & mk[t][g];
                    S.accG[g] |= S.RG[g];
                }
            }
        }
        if (h == 0) {
#pragma unroll
            for (int g = 0; g < NG; ++g) f4 |= S.accG[g] & L.g_found[g];
        }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) f8 |= S.accG[g] & L.g_found[g];
}

// One chunk (positions pos .. pos+15 of a read of length n; characters past a read's end are NUL).  `nxt` is the
// following chunk (its first eight characters' lead masks are requested here).  Every lane runs the words
// (wave-uniform control flow only: a lane that is done or idle computes values nobody reads).  GATED: the chunk
// may touch a window of the gated words (the callers run the chunks in front of / between the windows with
// GATED = false: straight-line code, nothing but lead words).  Returns the found bits seen so far; gg[i]: the same
// after the chunk's 4-character groups 0..2.
template <bool UNIFORM, bool GATED, int DL, int NL, int NG, class NextFn>
__device__ __forceinline__ uint32_t lean_chunk(const LeanWords<NL, NG>& L, LeanState<NL, NG>& S, const Chunk& cur,
                                               NextFn next_chunk, Chunk& nxt, const int pos, const int n,
                                               uint32_t (&gg)[3]) {
    typedef LeanLayout<DL, NL, NG> LY;
    uint32_t f0 = 0, f1 = 0, f2 = 0, f3 = 0;
    unsigned ad_hi[8];
    uint32_t mk1[8][NL];
    lean_addr8<LY::LEAD_SHIFT>(ad_hi, cur.w[2], cur.w[3]);
    lean_issue_lead<DL, NL, NG>(L, mk1, ad_hi);
    lean_lead8<DL, NL, NG>(L, S, S.mk0, f0, f1);
    if constexpr (GATED) {
        unsigned ad_g[8];
        lean_addr8<LY::GATED_SHIFT>(ad_g, cur.w[0], cur.w[1]);
        lean_gated8<UNIFORM, DL, NL, NG>(L, S, ad_g, pos, n, f0, f1);
    }
    f1 |= f0;
    nxt = next_chunk();                                          // requested earlier by the caller, needed from here on
    {
        unsigned ad_n[8];
        lean_addr8<LY::LEAD_SHIFT>(ad_n, nxt.w[0], nxt.w[1]);
        lean_issue_lead<DL, NL, NG>(L, S.mk0, ad_n);
    }
    f2 = f1; f3 = f1;
    lean_lead8<DL, NL, NG>(L, S, mk1, f2, f3);
    if constexpr (GATED) {
        unsigned ad_g[8];
        lean_addr8<LY::GATED_SHIFT>(ad_g, cur.w[2], cur.w[3]);
        lean_gated8<UNIFORM, DL, NL, NG>(L, S, ad_g, pos + 8, n, f2, f3);
    }
    f3 |= f2;
    gg[0] = f0; gg[1] = f1; gg[2] = f2;
    return f3;
}

// The last chunk of a read when at most eight of its characters are left (150 = 9 x 16 + 6): only the first half,
// whose lead masks are waiting in S.mk0 -- nothing is requested, nothing of a next chunk is looked at.
template <bool UNIFORM, bool GATED, int DL, int NL, int NG>
__device__ __forceinline__ uint32_t lean_half_chunk(const LeanWords<NL, NG>& L, LeanState<NL, NG>& S, const Chunk& cur,
                                                    const int pos, const int n, uint32_t (&gg)[3]) {
    typedef LeanLayout<DL, NL, NG> LY;
    uint32_t f0 = 0, f1 = 0;
    lean_lead8<DL, NL, NG>(L, S, S.mk0, f0, f1);
    if constexpr (GATED) {
        unsigned ad_g[8];
        lean_addr8<LY::GATED_SHIFT>(ad_g, cur.w[0], cur.w[1]);
        lean_gated8<UNIFORM, DL, NL, NG>(L, S, ad_g, pos, n, f0, f1);
    }
    f1 |= f0;
    gg[0] = f0; gg[1] = f1; gg[2] = f1;
    return f1;
}

// Is the chunk at `pos` (16 characters) clear of every window of the gated words?  (tail windows: the last
// tail_span characters of a read of length n; head windows: the first head_span.)  Wave-uniform.  Once a chunk
// touches a tail window every later chunk does, and only the first chunks touch a head window.
template <bool UNIFORM, int NL, int NG>
__device__ __forceinline__ bool lean_chunk_ungated(const LeanWords<NL, NG>& L, int pos, int n) {
    const bool tail = L.tail_span > 0 && (UNIFORM ? pos + 16 > n - L.tail_span : __any(pos + 16 > n - L.tail_span));
    return !tail && pos >= L.head_span;
}

// UNIFORM: every read of the batch has the length offsets[1] - offsets[0] (position, length and gate
// indices are scalars, reads need no offsets); otherwise the length, the read pointer and the tail words'
// gate index are per lane (the gate tables are then gathered from LDS instead of loaded by the scalar unit).
// NL, NG: slot capacities of this instance (the launcher picks the smallest class that holds the plan's words).
template <bool UNIFORM, int DL, int NL, int NG>
__global__ __launch_bounds__(256, LEAN_WAVES) void k_filter_lean(FilterArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_tab[LeanLayout<DL, NL, NG>::WORDS];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the batch check decides which variant works (no check was made for views: they count as ragged)
    // (uniform_len > 0: the host vouches for equally long reads -- cah_match_batch_uniform)
    const bool batch_is_uniform = a.uniform_len > 0 || (a.batch_flag && *a.batch_flag == 0ull);
    if (batch_is_uniform != UNIFORM) return;
    const CahLeanFilter* lf = a.lean;
    unsigned char* sp = smem;
    uint32_t* s_ginit = reinterpret_cast<uint32_t*>(sp);         sp += (size_t)CAH_LEAN_MAX_GATED * CAH_GATE_LEN * sizeof(uint32_t);
    uint16_t* s_idx = reinterpret_cast<uint16_t*>(sp);           sp += LEAN_TILE * sizeof(uint16_t);
    uint8_t* s_key = sp;                                         sp += LEAN_TILE;
    unsigned* s_hist = reinterpret_cast<unsigned*>(sp);          sp += CAH_QUEUE_BINS * sizeof(unsigned);
    unsigned* s_cursor = reinterpret_cast<unsigned*>(sp);        sp += CAH_QUEUE_BINS * sizeof(unsigned);
    unsigned long long& s_qbase = *reinterpret_cast<unsigned long long*>(sp);
    long long& s_tile = *reinterpret_cast<long long*>(sp + 8);
    unsigned& s_count = *reinterpret_cast<unsigned*>(sp + 16);
    unsigned* s_scratch = reinterpret_cast<unsigned*>(sp + 32);
    const int64_t first = !UNIFORM ? 0 : (a.uniform_len > 0 ? a.uniform_first : a.offsets[0]);
    const int n_uniform = !UNIFORM ? 0 : (a.uniform_len > 0 ? a.uniform_len : (int)(a.offsets[1] - first));   // every read has this length
    // equally long short reads are the streaming kernel's (k_filter_stream)
    if (UNIFORM && n_uniform >= a.stream_n_lo && n_uniform <= a.stream_n_hi && a.n_reads * (int64_t)n_uniform >= 16) return;
    lean_tables_to_lds<DL, NL, NG>(lf, s_tab);
    for (int i = threadIdx.x; i < NG * CAH_GATE_LEN; i += blockDim.x)
        s_ginit[i] = lf->gate_init[i / CAH_GATE_LEN][i % CAH_GATE_LEN];
    LeanWords<NL, NG> L;
    lean_words_init<NL, NG>(L, lf, s_tab, s_ginit, UNIFORM ? n_uniform : -1);
    const int lane = wave_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform, and said so

    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            s_tile = (long long)atomicAdd(a.work_counter, (unsigned long long)LEAN_TILE);
            s_count = 0;
        }
        for (int i = threadIdx.x; i < CAH_QUEUE_BINS; i += blockDim.x) { s_hist[i] = 0; s_cursor[i] = 0; }
        __syncthreads();
        const int64_t tile_base = s_tile;
        if (tile_base >= a.n_reads) break;

        for (int sub = wave; sub < LEAN_TILE / WAVE; sub += FILTER_WAVES) {
            const int64_t base = tile_base + (int64_t)sub * WAVE;
            if (base >= a.n_reads) break;
            const int64_t r = base + lane;
            const bool valid = r < a.n_reads;
            // n, n_max: this lane's read length and the longest of the wave (both n_uniform if UNIFORM)
            int n = n_uniform, n_max = n_uniform;
            const uint8_t* q;
            bool too_long = false;
            if constexpr (UNIFORM) {
                q = a.seqs + first + (valid ? r : base) * (int64_t)n;
            } else {
                int64_t off = 0, n64 = 0;
                if (valid) read_extent(a.offsets, a.lens, a.uniform_first, a.uniform_len, r, off, n64);
                if (n64 > a.max_read_len) { too_long = true; n64 = 0; }
                n = (int)n64;
                q = a.seqs + off;
                n_max = n;
#pragma unroll
                for (int d = 1; d < WAVE; d <<= 1) n_max = max(n_max, __shfl_xor(n_max, d, WAVE));
                n_max = __builtin_amdgcn_readfirstlane(n_max);
            }
            bool hit = false;
            int hit_pos = 0;
            unsigned seen = 0;
            LeanState<NL, NG> S;
#pragma unroll
            for (int w = 0; w < NL; ++w) { S.RL[w] = 0; S.accL[w] = 0; }
#pragma unroll
            for (int w = 0; w < NG; ++w) { S.RG[w] = 0; S.accG[w] = 0; }

            // two chunks are in flight ahead of the one being matched (the next one's first masks are requested
            // half a chunk early)
            Chunk cur = load_chunk(q, 0, n, valid ? n : 0);
            Chunk nxt = load_chunk(q, 16, n, valid ? n : 0);
            // (behind the first loads: waiting for those does not wait for these stores)
            if (a.clear_out6 && !a.present)
                clear_rows(a.clear_out6, a.clear_best, base, (int)(a.n_reads - base < WAVE ? a.n_reads - base : WAVE), lane);
            {
                unsigned ad[8];
                lean_addr8<LeanLayout<DL, NL, NG>::LEAD_SHIFT>(ad, cur.w[0], cur.w[1]);
                lean_issue_lead<DL, NL, NG>(L, S.mk0, ad);
            }
            // one chunk; false: every lane is done
            auto step = [&](auto gated, int pos) -> bool {
                const bool live = valid && !hit && (UNIFORM || pos < n);
                if (!__any(live)) return false;
                const Chunk nx2 = load_chunk(q, pos + 32, n, live ? n : 0);
                seen |= cur.w[0] | cur.w[1] | cur.w[2] | cur.w[3];
                uint32_t gg[3];
                Chunk got;
                const uint32_t found = lean_chunk<UNIFORM, decltype(gated)::value, DL, NL, NG>(
                    L, S, cur, [&]() { return nxt; }, got, pos, n, gg);
                if (live && found != 0) {
                    // key semantics as in k_filter: the 4-column group of the first hit of any word (no
                    // whole-read k-mer ended before it: those words were scanned through this chunk)
                    hit = true;
                    hit_pos = pos + (gg[0] ? 0 : gg[1] ? 4 : gg[2] ? 8 : 12);
                }
                cur = got;
                nxt = nx2;
                return true;
            };
            // chunks at the head windows, the stretch without windows (lead words only), chunks at the tail windows
            int pos = 0;                                             // wave-uniform
            bool more = true;
            for (; more && pos < n_max && pos < L.head_span; pos += 16) more = step(std::true_type{}, pos);
            for (; more && pos < n_max && lean_chunk_ungated<UNIFORM, NL, NG>(L, pos, n); pos += 16)
                more = step(std::false_type{}, pos);
            for (; more && pos < n_max; pos += 16) more = step(std::true_type{}, pos);
            const bool invalid = (seen & 0x80808080u) != 0 || too_long;
            lean_emit(a, r, tile_base, valid, hit, invalid, hit_pos, s_idx, s_key, s_hist, s_count);
        }

        if (!a.present) {
            __syncthreads();
            flush_tile_queue(a, tile_base, s_idx, s_key, s_hist, s_cursor, s_count, s_scratch, s_qbase);
        }
    }
}

// =============================================================================================
// k_filter_stream: the lean prefilter for batches of equally long SHORT reads (the sequencer's output; BASELINE
// C2-C5).  A wave's 64 reads are one contiguous piece of HBM (64 * n bytes).  k_filter_lean fetches them with one
// 16-byte load per lane and chunk, so every cache line is touched by ~9 separate loads and the kernel lives on
// all waves' lines staying in L2 (4.9 MB per XCD at 4 waves/SIMD against its 4 MB: measured, removing all
// matching work makes that kernel SLOWER).  Here the wave copies its piece with coalesced loads -- lane i of
// load k takes 16-byte unit u = 64 k + i of the piece, i.e. unit u % U of read u / U (U = units per read), so
// every cache line crosses the memory system once -- into its own LDS slot, one 16-byte aligned row per read
// (row stride: an odd number of units, which spreads the lanes' ds_read_b128 over all banks), and every lane
// then reads its characters with one aligned ds_read_b128 per chunk.  The copy of the NEXT piece is in flight in
// NU x 4 VGPRs while the current one is matched.  Work is dealt statically (tiles of STREAM_TILE reads
// round-robin over the blocks: equal reads, equal work), which is what lets a wave know its next piece.  Same
// words, same outputs and queue keys as k_filter_lean.
// NU: row stride in 16-byte units (odd); UM <= NU: units of a read the instance copies, i.e. it takes reads of up
// to 16 * UM characters (the copy registers are sized by UM).
// =============================================================================================
// One block of 12 waves per CU (3 per SIMD: the LDS slots allow no more): a single block shares ONE survivor
// staging area, which leaves room for tiles of 6144 reads -- the DP behind the queue wants long key-ordered runs
// (with 1024-read tiles of three 4-wave blocks, scan + DP lost 1.1 ms per 100 M reads of lock-step efficiency).
#ifndef STREAM_BLOCK_WAVES
#define STREAM_BLOCK_WAVES 12
#endif
#define STREAM_WAVES 3             // waves per SIMD
// reads per block tile (survivor staging: 3 B each; a multiple of 64 * 12); the classes with six gated words have
// larger mask and gate tables and take a smaller tile to stay inside the CU's 160 KiB
// reads per tile (a multiple of 12 waves x 64): what the 160 KB of LDS leave for the survivor staging behind the
// read slots and the tables of the class (four lead tables since the lead words take four characters per step)
__host__ __device__ constexpr int stream_tile(int nl, int ng) { return ng <= 3 ? 6144 : (nl <= 2 ? 4608 : 3840); }
__host__ __device__ constexpr int stream_piece_bytes(int nu) { return WAVE * nu * 16; }
__host__ __device__ constexpr int stream_max_len(int um) { return um * 16; }

template <int DL, int NL, int NG, int NU, int UM>
__global__ __launch_bounds__(STREAM_BLOCK_WAVES * WAVE, STREAM_WAVES) void k_filter_stream(FilterArgs a) {
    // ONE static object, the mask tables first: they then sit below 64 KB and a table entry is read with
    // "ds_read_b64 v, v_entry offset:TABLE" -- the entry offset (byte << 3, one SDWA instruction) is the whole
    // address computation of a character.  (As separate variables the tables were placed behind the 135 KB of read
    // slots and every address needed the base added in a register.)
    struct __attribute__((aligned(16))) StreamLds {
        uint32_t tab[LeanLayout<DL, NL, NG>::WORDS];
        uint32_t gate[NG * CAH_GATE_LEN];
        unsigned char piece[STREAM_BLOCK_WAVES * stream_piece_bytes(NU)];
    };
    static_assert((LeanLayout<DL, NL, NG>::WORDS * 4) % 16 == 0 && (NG * CAH_GATE_LEN * 4) % 16 == 0, "slots must stay 16-byte aligned");
    __shared__ StreamLds s_lds;
    uint32_t* const s_tab = s_lds.tab;
    uint32_t* const s_gate = s_lds.gate;
    unsigned char* const s_piece = s_lds.piece;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TILE = stream_tile(NL, NG), SUBS = TILE / WAVE / STREAM_BLOCK_WAVES;
    static_assert(sizeof(StreamLds) + (size_t)TILE * 3 + CAH_QUEUE_BINS * 8 + 64 <= 160 * 1024, "k_filter_stream: LDS");
    if (a.batch_flag ? *a.batch_flag != 0ull : false) return;           // ragged batch: k_filter_lean<false, ..>
    const CahLeanFilter* lf = a.lean;
    const int64_t first = a.uniform_len > 0 ? a.uniform_first : a.offsets[0];
    const int n = a.uniform_len > 0 ? a.uniform_len : (int)(a.offsets[1] - first);     // every read has this length
    if (n < a.stream_n_lo || n > a.stream_n_hi) return;                 // another instance's (or k_filter_lean's) batch
    const int64_t total = a.n_reads * (int64_t)n;                       // bytes of the batch
    if (total < 16) return;                                             // k_filter_lean<true, ..> takes it (same test there)
    unsigned char* sp = smem;
    uint16_t* s_idx = reinterpret_cast<uint16_t*>(sp);           sp += TILE * sizeof(uint16_t);
    uint8_t* s_key = sp;                                         sp += TILE;
    unsigned* s_hist = reinterpret_cast<unsigned*>(sp);          sp += CAH_QUEUE_BINS * sizeof(unsigned);
    unsigned* s_cursor = reinterpret_cast<unsigned*>(sp);        sp += CAH_QUEUE_BINS * sizeof(unsigned);
    unsigned long long& s_qbase = *reinterpret_cast<unsigned long long*>(sp);
    unsigned& s_count = *reinterpret_cast<unsigned*>(sp + 16);
    unsigned* s_scratch = reinterpret_cast<unsigned*>(sp + 32);
    lean_tables_to_lds<DL, NL, NG>(lf, s_tab);
    for (int i = threadIdx.x; i < NG * CAH_GATE_LEN; i += blockDim.x)
        s_gate[i] = lf->gate_init[i / CAH_GATE_LEN][i % CAH_GATE_LEN];
    LeanWords<NL, NG> L;
    lean_words_init<NL, NG>(L, lf, s_tab, s_gate, n);
    const int lane = wave_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform, and said so
    unsigned char* const piece = s_piece + wave * stream_piece_bytes(NU);   // this wave's LDS slot
    const unsigned char* const row = piece + lane * (NU * 16);             // this lane's read in it

    // Copy plan of a lane: load k takes unit u = 64 k + lane of the piece = unit c of read r, r = u / U, c = u % U
    // (U = ceil(n / 16) units per read; the last unit of a read runs into the next read -- masked when used).
    // goff: its byte offset from the piece's first byte in HBM, (r * NU + c) * 16: in the LDS slot.  u / U as a multiply
    // (exact for u < 64 * 16, U <= 16: checked exhaustively by tests/test_host_logic.py).
    const int U = (n + 15) >> 4;                                        // <= UM
    const unsigned magic = U ? (65536u + (unsigned)U - 1u) / (unsigned)U : 0u;
    // Both are recomputed where they are used (a handful of instructions per unit and piece; keeping them would
    // cost 2 x UM registers that the kernel does not have).
    auto unit_rc = [&](int k, unsigned& r, unsigned& c) {
        unsigned ln = (unsigned)lane;
        asm volatile("" : "+v"(ln));                                    // keeps the compiler from hoisting (and keeping) the results
        const unsigned u = (unsigned)(k * WAVE) + ln;
        r = __umul24(u, magic) >> 16;                                   // 24-bit multiplies: u < 2^10, magic <= 2^16
        c = u - __umul24(r, (unsigned)U);
    };
    // piece `it` of this wave: sub-tile wave + 12 * (it % SUBS) of tile blockIdx.x + (it / SUBS) * gridDim.x
    auto piece_base = [&](int64_t it) -> int64_t {
        return ((int64_t)blockIdx.x + (it / SUBS) * (int64_t)gridDim.x) * TILE +
               ((int64_t)wave + STREAM_BLOCK_WAVES * (it % SUBS)) * WAVE;
    };
    // the units of the piece starting at read `base`, into VGPRs.  Nothing outside the batch is touched: a unit
    // that would run past the batch's last byte is fetched as the 16 bytes that END there and shifted down.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 pre[UM];
    const uint8_t* const batch0 = a.seqs + first;
    auto prefetch = [&](int64_t base) {
        const int64_t left = a.n_reads - base;                          // wave-uniform
        const int64_t pbyte = base * (int64_t)n;                        // the piece's first byte within the batch
        const uint8_t* const src = batch0 + pbyte;
        if (left >= WAVE && pbyte + (int64_t)WAVE * n + 16 <= total) {
            // a whole piece with 16 bytes of the batch behind it (all but the last pieces): unit u = 64 k + lane
            // exists iff k < U, no lane needs a check
#pragma unroll
            for (int k = 0; k < UM; ++k) {
                pre[k] = (u32x4)(0u);
                if (k < U) {
                    unsigned ur, uc;
                    unit_rc(k, ur, uc);
                    Unaligned16 v;
                    __builtin_memcpy(&v, src + (__umul24(ur, (unsigned)n) + 16u * uc), 16);
                    pre[k] = (u32x4){v.w[0], v.w[1], v.w[2], v.w[3]};
                }
            }
            return;
        }
        const int units = left <= 0 ? 0 : (int)(left < WAVE ? left : (int64_t)WAVE) * U;
#pragma unroll
        for (int k = 0; k < UM; ++k) {
            pre[k] = (u32x4)(0u);
            if (k * WAVE + lane < units) {
                unsigned ur, uc;
                unit_rc(k, ur, uc);
                const unsigned goff = __umul24(ur, (unsigned)n) + 16u * uc;
                if (pbyte + goff + 16 <= total) {
                    Unaligned16 v;
                    __builtin_memcpy(&v, src + goff, 16);
                    pre[k] = (u32x4){v.w[0], v.w[1], v.w[2], v.w[3]};
                } else {
                    Unaligned16 v;
                    __builtin_memcpy(&v, batch0 + (total - 16), 16);
                    const int sft = (int)(pbyte + goff + 16 - total);   // 1..15 bytes to drop
                    const int dw = sft >> 2, sh = (sft & 3) * 8;
                    unsigned x0 = v.w[0], x1 = v.w[1], x2 = v.w[2], x3 = v.w[3];
                    if (dw >= 2) { x0 = x2; x1 = x3; x2 = 0; x3 = 0; }
                    if (dw & 1) { x0 = x1; x1 = x2; x2 = x3; x3 = 0; }
                    pre[k] = (u32x4){(unsigned)((((unsigned long long)x1 << 32) | x0) >> sh),
                                     (unsigned)((((unsigned long long)x2 << 32) | x1) >> sh),
                                     (unsigned)((((unsigned long long)x3 << 32) | x2) >> sh), x3 >> sh};
                }
            }
        }
    };
    // the 16 characters at positions pos .. pos+15 of this lane's read; characters past its end come back as NUL
    auto finish = [&](u32x4 v, int pos) -> Chunk {
        Chunk c;
        c.w[0] = v.x; c.w[1] = v.y; c.w[2] = v.z; c.w[3] = v.w;
        if (pos + 16 > n) {
            // the last chunk (or none at all): what follows in the row is the next read's or stale
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int keep = n - pos - 4 * i;                       // characters of dword i inside the read
                c.w[i] &= keep >= 4 ? 0xFFFFFFFFu : (keep <= 0 ? 0u : ((1u << (8 * keep)) - 1u));
            }
        }
        return c;
    };

    int64_t it = 0;
    prefetch(piece_base(0));
    for (int64_t k = 0;; ++k) {
        const int64_t tile_base = ((int64_t)blockIdx.x + k * (int64_t)gridDim.x) * TILE;
        if (tile_base >= a.n_reads) break;                              // block-uniform
        __syncthreads();
        if (threadIdx.x == 0) s_count = 0;
        for (int i = threadIdx.x; i < CAH_QUEUE_BINS; i += blockDim.x) { s_hist[i] = 0; s_cursor[i] = 0; }
        __syncthreads();

        for (int j = 0; j < SUBS; ++j, ++it) {
            const int64_t base = tile_base + ((int64_t)wave + STREAM_BLOCK_WAVES * j) * WAVE;
            // the piece in the registers goes to the LDS slot (every lane is done with the previous piece: LDS
            // operations of a wave execute in order); then the next piece's loads are issued
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < UM; ++q)
                if (q < U) {                                            // unit 64 q + lane exists iff q < U
                    unsigned ur, uc;
                    unit_rc(q, ur, uc);
                    *reinterpret_cast<u32x4*>(piece + (__umul24(ur, (unsigned)NU) + uc) * 16u) = pre[q];
                }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (in front of the next piece's loads: stores and loads share the in-order vmcnt counter, and the next
            // wait for loads is a whole piece of matching work away -- the stores are long finished by then)
            if (a.clear_out6 && !a.present && base < a.n_reads)
                clear_rows(a.clear_out6, a.clear_best, base, (int)(a.n_reads - base < WAVE ? a.n_reads - base : WAVE), lane);
            prefetch(piece_base(it + 1));
            if (base >= a.n_reads) continue;                            // wave-uniform; nothing left in this tile
            const int64_t r = base + lane;
            const bool valid = r < a.n_reads;
            int hit_pos = -1;                                           // >= 0: a k-mer was found, in this 4-column group
            unsigned seen = 0;
            LeanState<NL, NG> S;
#pragma unroll
            for (int w = 0; w < NL; ++w) { S.RL[w] = 0; S.accL[w] = 0; }
#pragma unroll
            for (int w = 0; w < NG; ++w) { S.RG[w] = 0; S.accG[w] = 0; }
            Chunk cur = finish(*reinterpret_cast<const u32x4*>(row), 0);
            {
                unsigned ad[8];
                lean_addr8<LeanLayout<DL, NL, NG>::LEAD_SHIFT>(ad, cur.w[0], cur.w[1]);
                lean_issue_lead<DL, NL, NG>(L, S.mk0, ad);
            }
            // one chunk; false: every lane is done
            auto step = [&](auto gated, int pos) -> bool {
                const bool live = valid && hit_pos < 0;
                if (!__any(live)) return false;
                // the next chunk is requested now and looked at half a chunk later (a row has NU units: the
                // request stays inside it while pos + 16 < n)
                u32x4 raw = (u32x4)(0u);
                if (pos + 16 < n) raw = *reinterpret_cast<const u32x4*>(row + pos + 16);
                seen |= cur.w[0] | cur.w[1] | cur.w[2] | cur.w[3];
                uint32_t gg[3];
                Chunk nxt;
                const uint32_t found = lean_chunk<true, decltype(gated)::value, DL, NL, NG>(
                    L, S, cur, [&]() { return finish(raw, pos + 16); }, nxt, pos, n, gg);
                if (live && found != 0) hit_pos = pos + (gg[0] ? 0 : gg[1] ? 4 : gg[2] ? 8 : 12);
                cur = nxt;
                return true;
            };
            // chunks at the head windows, the stretch without windows (lead words only), chunks at the tail windows
            int pos = 0;                                                // wave-uniform
            bool more = true;
            const int n_full = n - 8;                                   // chunks starting below it have > 8 characters
            for (; more && pos < n_full && pos < L.head_span; pos += 16) more = step(std::true_type{}, pos);
            for (; more && pos < n_full && lean_chunk_ungated<true, NL, NG>(L, pos, n); pos += 16) more = step(std::false_type{}, pos);
            for (; more && pos < n_full; pos += 16) more = step(std::true_type{}, pos);
            if (more && pos < n && __any(valid && hit_pos < 0)) {
                // at most eight characters left: half a chunk
                const bool live = valid && hit_pos < 0;
                seen |= cur.w[0] | cur.w[1] | cur.w[2] | cur.w[3];
                uint32_t gg[3];
                uint32_t found;
                if (lean_chunk_ungated<true, NL, NG>(L, pos, n)) found = lean_half_chunk<true, false, DL, NL, NG>(L, S, cur, pos, n, gg);
                else found = lean_half_chunk<true, true, DL, NL, NG>(L, S, cur, pos, n, gg);
                if (live && found != 0) hit_pos = pos + (gg[0] ? 0 : 4);
            }
            const bool invalid = (seen & 0x80808080u) != 0;
            lean_emit(a, r, tile_base, valid, hit_pos >= 0, invalid, hit_pos, s_idx, s_key, s_hist, s_count);
        }

        if (!a.present) {
            __syncthreads();
            flush_tile_queue(a, tile_base, s_idx, s_key, s_hist, s_cursor, s_count, s_scratch, s_qbase);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// One DP column, rows I..ROWS, as a compile-time recursion: guarantees full unrolling (so the
// column arrays stay in VGPRs) and yields *nested* per-lane predicates -- a lane whose band
// ends at row `last` drops out of exec for the rest of the column, and once exec is empty the
// remaining rows are skipped with one scalar branch.
// ---------------------------------------------------------------------------------------------
template <int I, int ROWS, bool UNIT>
__device__ __forceinline__ void dp_rows(int (&c)[ROWS + 1], int (&p)[ROWS + 1], const uint64_t mk,
                                        int dc, int dp, int& nl, int& cm_c, int& cm_p,
                                        const int last, const int m, const int k, const int Dm1) {
    if constexpr (I <= ROWS) {
        // leave the column as soon as no lane of the wave has band left
        if (!__any(last >= I)) return;                    // in front of every row (+3 % over every other row)
        // Straight-line, select-only cell update (no exec-mask regions: lanes whose band ended
        // compute a value that is discarded by the final select, so stale cells stay stale).
        // (:446-476) match: take the diagonal unconditionally; otherwise min of {diag+1, del,
        // ins} with ties resolved mismatch >= deletion >= insertion.
        const int oc = c[I], op = p[I];
        const int cprev = c[I - 1], pprev = p[I - 1];
        const unsigned mword = (I - 1) < 32 ? (unsigned)mk : (unsigned)(mk >> 32);
        // --- independent of the cell above (can be issued while its result is in flight) -------
        const bool eq = (mword & (1u << ((I - 1) & 31))) != 0;
        const bool in_band = I <= last;                   // per lane: Ukkonen band (last <= m)
        const int p_diag = dp + (eq ? 1 : -1);            // diagonal payload: +1 match / -1 mismatch
        // --- the serial part: cost through the cell above ------------------------------------
        // x = min(c_del, c_ins) - 1 with c_del = cprev + D, c_ins = oc + D  (Dm1 = D - 1, wave-uniform;
        // UNIT = unit indel cost, the default: one add less per cell)
        const int x = UNIT ? min(cprev, oc) : min(cprev, oc) + Dm1;
        const int c_ne = min(dc, x) + 1;                  // min(c_diag, c_del, c_ins)
        const bool mis = dc <= x;                         // c_diag <= c_del && c_diag <= c_ins
        const bool del = cprev <= oc;                     // c_del <= c_ins
        const int cost = eq ? dc : c_ne;                  // match takes the diagonal unconditionally
        // payload: diag (match/mismatch) | deletion (cell above, -2) | insertion (old cell, -2)
        const int p_indel = (del ? pprev : op) - 2;
        const int pay = (eq || mis) ? p_diag : p_indel;
        c[I] = in_band ? cost : oc;                       // out of band: the stale cell stays
        p[I] = in_band ? pay : op;
        // A cell outside the band has cost > k (it left the band that way, or is an initial
        // cell below row k+1), so testing the stored value needs no "in_band &&".
        nl = c[I] <= k ? I : nl;
        if constexpr (I > ROWS - 8) {                     // m is in (ROWS-8, ROWS]: wave-uniform capture
            // (row capacities come in steps of 4; 8 keeps the smallest kernel, m <= 8, correct)
            if (I == m) { cm_c = cost; cm_p = pay; }
        }
        if constexpr ((I % CAH_SCHED_ROWS) == 0) __builtin_amdgcn_sched_barrier(0);
        dp_rows<I + 1, ROWS, UNIT>(c, p, mk, oc, op, nl, cm_c, cm_p, last, m, k, Dm1);   // diag := old cell (:479)
    }
}

// =============================================================================================
// k_dp<ROWS>: the banded semi-global aligner, one read per lane, column in VGPRs.
// =============================================================================================

// Row i (wave-uniform) of the register-resident column, via a scalar switch.
#define CAH_ROW_CASE(K) case K: if constexpr (K <= ROWS) { ci = c[K]; pi = p[K]; } break;
template <int ROWS>
__device__ __forceinline__ void get_row(const int (&c)[ROWS + 1], const int (&p)[ROWS + 1], int i, int& ci, int& pi) {
    ci = 0; pi = 0;
    switch (__builtin_amdgcn_readfirstlane(i)) {
        CAH_ROW_CASE(0) CAH_ROW_CASE(1) CAH_ROW_CASE(2) CAH_ROW_CASE(3) CAH_ROW_CASE(4) CAH_ROW_CASE(5) CAH_ROW_CASE(6) CAH_ROW_CASE(7)
        CAH_ROW_CASE(8) CAH_ROW_CASE(9) CAH_ROW_CASE(10) CAH_ROW_CASE(11) CAH_ROW_CASE(12) CAH_ROW_CASE(13) CAH_ROW_CASE(14) CAH_ROW_CASE(15)
        CAH_ROW_CASE(16) CAH_ROW_CASE(17) CAH_ROW_CASE(18) CAH_ROW_CASE(19) CAH_ROW_CASE(20) CAH_ROW_CASE(21) CAH_ROW_CASE(22) CAH_ROW_CASE(23)
        CAH_ROW_CASE(24) CAH_ROW_CASE(25) CAH_ROW_CASE(26) CAH_ROW_CASE(27) CAH_ROW_CASE(28) CAH_ROW_CASE(29) CAH_ROW_CASE(30) CAH_ROW_CASE(31)
        CAH_ROW_CASE(32) CAH_ROW_CASE(33) CAH_ROW_CASE(34) CAH_ROW_CASE(35) CAH_ROW_CASE(36) CAH_ROW_CASE(37) CAH_ROW_CASE(38) CAH_ROW_CASE(39)
        CAH_ROW_CASE(40) CAH_ROW_CASE(41) CAH_ROW_CASE(42) CAH_ROW_CASE(43) CAH_ROW_CASE(44) CAH_ROW_CASE(45) CAH_ROW_CASE(46) CAH_ROW_CASE(47)
        CAH_ROW_CASE(48) CAH_ROW_CASE(49) CAH_ROW_CASE(50) CAH_ROW_CASE(51) CAH_ROW_CASE(52) CAH_ROW_CASE(53) CAH_ROW_CASE(54) CAH_ROW_CASE(55)
        CAH_ROW_CASE(56) CAH_ROW_CASE(57) CAH_ROW_CASE(58) CAH_ROW_CASE(59) CAH_ROW_CASE(60) CAH_ROW_CASE(61) CAH_ROW_CASE(62) CAH_ROW_CASE(63)
        CAH_ROW_CASE(64)
        default: break;
    }
}

// Register budget: 2*(ROWS+1) VGPRs hold the column; ask the register allocator for an occupancy
// that leaves room for that plus ~40 temporaries (without the bound the scheduler hoists all the
// chain-independent work of a column up front and lands at 256 VGPRs = 1 wave per SIMD).
#ifndef CAH_DP_WAVES
#define CAH_DP_WAVES(ROWS) ((ROWS) <= 16 ? 5 : ((ROWS) <= 40 ? 4 : ((ROWS) <= 56 ? 3 : 2)))
#endif

template <int ROWS, bool UNIT>
__global__ __launch_bounds__(256, CAH_DP_WAVES(ROWS)) void k_dp(DpArgs a) {
    __shared__ uint64_t s_rowmask[CAH_TABLE_CHARS];
    __shared__ int s_ncnt[CAH_MAX_M + 1];
    __shared__ int s_thr[CAH_MAX_M + 1];
    const CahMatcher* mt = a.matcher;
    for (int i = threadIdx.x; i < CAH_TABLE_CHARS; i += blockDim.x) s_rowmask[i] = mt->rowmask[i];
    for (int i = threadIdx.x; i <= CAH_MAX_M; i += blockDim.x) {
        s_ncnt[i] = mt->n_counts[i];
        s_thr[i] = mt->thr[i];
    }
    __syncthreads();

    // wave-uniform adapter constants (SGPRs)
    const int m = mt->m, k = mt->k, D = mt->indel_cost;
    const int flags = mt->flags;
    const bool start_in_ref = flags & 1, start_in_query = flags & 2;
    const bool stop_in_ref = flags & 4, stop_in_query = flags & 8;
    const int min_overlap = mt->min_overlap;
    const bool wildcard_ref = mt->wildcard_ref != 0;
    const int eff_full = mt->effective_length;
    const int half_m = m / 2;
    // row 0 update per column (_align.pyx:413-415, :438-440), in packed form
    const int row0_cost_inc = start_in_query ? 0 : D;
    const int row0_pay_inc = start_in_query ? (1 << CAH_SCORE_BITS) : -2;
    // first-column weights (see below)
    const int init_org_lo = start_in_ref ? -(1 << 24) : 0, init_org_hi = start_in_query ? (1 << 24) : 0;
    const int w_max = (!start_in_ref && !start_in_query) ? 1 : 0, w_min = (start_in_ref && start_in_query) ? 1 : 0;
    const int w_y = (start_in_ref && !start_in_query) ? 1 : 0, w_x = (!start_in_ref && start_in_query) ? 1 : 0;
    const int init_score_mul = start_in_ref ? 0 : -2;

    const int lane = wave_lane();
    int64_t total = a.n_reads;
    if (a.queue_count) total = (int64_t)(*a.queue_count);
    const bool skip_cols = a.queue && a.queue_keys && mt->skip_ok != 0;

    // (CAH_DEQUEUE sub-batches of 64 per atomic: one counter serves 100 M reads when there is no work list, and
    // same-address atomics complete at ~100 M/s -- one per wave made an anchored adapter's short DP atomic-bound)
    for (;;) {
      const int64_t base0 = wave_dequeue(a.work_counter, CAH_DEQUEUE * WAVE);
      if (base0 >= total) break;
      for (int sub = 0; sub < CAH_DEQUEUE; ++sub) {
        const int64_t base = base0 + (int64_t)sub * WAVE;
        if (base >= total) break;
        const int64_t idx = base + lane;
        const bool valid = idx < total;
        int64_t r = 0;
        if (valid) r = a.queue ? (int64_t)a.queue[idx] : idx;
        int64_t off = 0, n64 = 0;
        if (valid) read_extent(a.offsets, a.lens, a.uniform_first, a.uniform_len, r, off, n64);
        bool invalid = false;
        if (n64 > a.max_read_len) { invalid = true; n64 = 0; }
        const int n = (int)n64;
        const uint8_t* q = a.seqs + off;

        // columns to compute (_align.pyx:346-352)
        int max_n = n, min_n = 0;
        if (!start_in_query) max_n = min(n, m + k);
        if (!stop_in_query) min_n = max(0, n - m - k);
        // Column skipping (3' adapters with a verified pigeonhole prefilter only, skip_ok): no
        // k-mer of the prefilter occurs before character `first_hit`, hence no full-length match
        // can end before it, and every cell that can influence the result (cost <= k+1, at or
        // after column first_hit) has all its optimal paths inside columns >= first_hit-m-k-1.
        // Starting the banded DP there with the plain first column gives bit-identical results
        // ("Column skipping").
        int skip_to = 0;
        if (skip_cols && valid) skip_to = max(0, ((int)a.queue_keys[idx] << CAH_KEY_SHIFT) - m - k - 1);
        if (skip_to > 0) min_n = skip_to;

        // first column (_align.pyx:364-383).  The four (start_in_reference, start_in_query)
        // cases are folded into wave-uniform weights/clamps so that no branch is needed:
        //   origin = clamp(min_n - i, start_in_ref ? -inf : 0, start_in_query ? +inf : 0)
        //   cost   = D * {max(i,min_n) | min_n | i | min(i,min_n)},  score = start_in_ref ? 0 : -2i
        int c[ROWS + 1], p[ROWS + 1];
#pragma unroll
        for (int i = 0; i <= ROWS; ++i) {
            const int d = min_n - i;
            const int org = min(max(d, init_org_lo), init_org_hi);
            const int co = (w_max * max(i, min_n) + w_min * min(i, min_n) + w_y * min_n + w_x * i) * D;
            c[i] = i > m ? (1 << 28) : (skip_to > 0 ? i * D : co);   // rows beyond the adapter: never <= k
            p[i] = pack_cell(skip_to > 0 ? skip_to : org, init_score_mul * i);
        }

        const int SENT = m + n + 1;                       // :394
        int b_cost = SENT, b_pay = pack_cell(0, 0), b_refstop = m, b_qstop = n;

        int last = start_in_ref ? m : min(m, k + 1);      // :399-401
        int last_filled = 0;   // `last_filled_i` of the most recent column
        int lf_ran = 0;        // ... of the most recent column whose row loop wrote a cell
        int j = min_n;
        bool done = !valid;

        // Columns are walked in lock step by all lanes (lane-local column j = min_n + 1 + step).
        // The read characters arrive 16 per global load (next chunk requested one chunk ahead);
        // the chunk is consumed by shifting it down one byte per column, and the LDS lookup of
        // the NEXT column's row bitset is issued before the current column is computed, so
        // neither the global nor the LDS latency sits on the column's critical path.
        int pos = min_n;
        Chunk cur = load_chunk(q, pos, n, valid ? max_n : 0);
        Chunk nxt = load_chunk(q, pos + 16, n, valid ? max_n : 0);
        int left = 16;                                    // characters left in `cur` (wave-uniform)
        unsigned bad_chars = cur.w[0] | cur.w[1] | cur.w[2] | cur.w[3];
        uint64_t mk_next = s_rowmask[cur.w[0] & (CAH_TABLE_CHARS - 1)];
        for (;;) {
            {
                const bool act = !done && j < max_n;
                if (!__any(act)) break;
                const uint64_t mk = mk_next;
                // advance the character stream by one byte and start the next lookup
                cur.w[0] = (cur.w[0] >> 8) | (cur.w[1] << 24);
                cur.w[1] = (cur.w[1] >> 8) | (cur.w[2] << 24);
                cur.w[2] = (cur.w[2] >> 8) | (cur.w[3] << 24);
                cur.w[3] >>= 8;
                if (--left == 0) {
                    cur = nxt;
                    bad_chars |= cur.w[0] | cur.w[1] | cur.w[2] | cur.w[3];
                    pos += 16;
                    nxt = load_chunk(q, pos + 16, n, (!done) ? max_n : 0);
                    left = 16;
                }
                mk_next = s_rowmask[cur.w[0] & (CAH_TABLE_CHARS - 1)];
                if (act) {
                    ++j;

                    int dc = c[0], dp = p[0];                 // diagonal for row 1
                    c[0] += row0_cost_inc;
                    p[0] += row0_pay_inc;
                    int nl = c[0] <= k ? 0 : -1;              // largest computed row with cost <= k
                    int cm_c = c[0], cm_p = p[0];             // cell (m, j), captured for the candidate test
                                                              // (row 0 itself when m == 0)
                    dp_rows<1, ROWS, UNIT>(c, p, mk, dc, dp, nl, cm_c, cm_p, last, m, k, D - 1);
                    last_filled = last;                       // :484
                    if (last >= 1) lf_ran = last;
                    // band update (:490-495)
                    if (nl < m) {
                        last = nl + 1;
                    } else {
                        last = m;
                        if (stop_in_query) {                  // candidate in the last row (:496-533)
                            const int cost = cm_c, origin = cell_origin(cm_p), score = cell_score(cm_p);
                            const int length = m + min(origin, 0);
                            int eff = length;
                            if (wildcard_ref)
                                eff = length < m ? length - (s_ncnt[m] - s_ncnt[m - length]) : eff_full;
                            const bool ok = length >= min_overlap && cost <= s_thr[eff];
                            const int b_origin = cell_origin(b_pay), b_score = cell_score(b_pay);
                            const int best_len = m + min(b_origin, 0);
                            if (ok && (b_cost == SENT || (origin <= b_origin + half_m && score > b_score) ||
                                       (length > best_len && score > b_score))) {
                                b_cost = cost; b_pay = cm_p; b_refstop = m; b_qstop = j;
                                if (cost == 0 && origin >= 0) done = true;   // exact match: stop early
                            }
                        }
                    }
                }
            }
        }
        if (bad_chars & 0x80808080u) invalid = true;

        // last column (:536-572).  The update test uses the *stale* scalar `origin`: the origin
        // of the last cell the row loop wrote, i.e. cell lf_ran of the column it last ran in.
        int stale_origin = 0, scan_lo = 0;
        {
            int stale_p = pack_cell(0, 0);
#pragma unroll
            for (int i = 1; i <= ROWS; ++i) stale_p = (i == lf_ran) ? p[i] : stale_p;
            stale_origin = cell_origin(stale_p);
            const int first_i = stop_in_ref ? 0 : m;
            scan_lo = first_i;
        }
        // The scan itself is a ROLLED loop over a wave-uniform row index; the row is fetched
        // from the register column with a scalar switch.  (Unrolling it 65 times makes the
        // compiler keep hundreds of temporaries alive and spill inside the DP loop.)
        const bool scan_lane = valid && max_n == n;
#pragma unroll 1
        for (int i = m; i >= 0; --i) {
            const bool want = scan_lane && i <= last_filled && i >= scan_lo;
            if (!__any(want)) continue;
            int ci, pi;
            get_row<ROWS>(c, p, i, ci, pi);
            if (want) {
                const int o_i = cell_origin(pi), score = cell_score(pi), cost = ci;
                const int ref_start = -min(o_i, 0);
                const int length = i - ref_start;
                int eff = length;
                if (wildcard_ref)
                    eff = length < m ? length - (s_ncnt[i] - s_ncnt[ref_start]) : eff_full;
                const bool ok = length >= min_overlap && cost <= s_thr[eff];
                const int b_origin = cell_origin(b_pay), b_score = cell_score(b_pay);
                const int best_len = b_refstop + min(b_origin, 0);
                if (ok && (b_cost == SENT || (stale_origin <= b_origin + half_m && score > b_score) ||
                           (length > best_len && score > b_score))) {
                    b_cost = cost; b_pay = pi; b_refstop = i; b_qstop = n;
                }
            }
        }

        if (valid) {
            const bool found = b_cost != SENT && !invalid;
            const int origin = cell_origin(b_pay), score = cell_score(b_pay);
            int32_t* o = a.out6 + r * 6;
            if (a.merge_best) {
                // MultipleAdapters.match_to (adapters.py:1278-1285): adapters are launched in
                // order on one stream; a later adapter replaces the current best only if strictly
                // better (higher score, or equal score and fewer errors).
                if (invalid) {
                    a.status[r] = 2;
                } else if (found) {
                    const bool had = a.status[r] == 1;
                    if (a.status[r] != 2 && (!had || score > o[4] || (score == o[4] && b_cost < o[5]))) {
                        o[0] = origin >= 0 ? 0 : -origin; o[1] = b_refstop;
                        o[2] = origin >= 0 ? origin : 0;  o[3] = b_qstop;
                        o[4] = score; o[5] = b_cost;
                        a.status[r] = 1;
                        if (a.best_adapter) a.best_adapter[r] = a.adapter_index;
                    }
                }
            } else {
                a.status[r] = invalid ? (uint8_t)2 : (found ? (uint8_t)1 : (uint8_t)0);
                if (found) {
                    o[0] = origin >= 0 ? 0 : -origin; o[1] = b_refstop;
                    o[2] = origin >= 0 ? origin : 0;  o[3] = b_qstop;
                    o[4] = score; o[5] = b_cost;
                } else {
                    o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0;
                }
            }
        }
      }
    }
}

// =============================================================================================
// k_dp_packed<ROWS>: Aligner.locate for the common case -- a 3' adapter (flags = QUERY_START |
// QUERY_STOP | REFERENCE_END, reference Where.BACK) with unit indel cost -- with ONE 32-bit word
// per DP cell:
//
//      bits 20..27  cost          (<= 64: a cell of row i costs at most i, by i deletions)
//      bits 18..19  priority      (only inside the 3-way minimum, cleared afterwards)
//      bits  9..17  rel = j - origin, the number of read columns the alignment spans so far
//                   (<= i + cost <= 128; row 0 is the constant (0, 0, 0))
//      bits  0..8   score + 256   (in [-128, 64])
//
// On gfx950 v_min/v_cmp/v_cndmask issue at 4 cycles per wave64, v_add/v_and at 2 (measured,
// DESIGN.md), so the cell is organised around adds and ONE v_min3_u32: the three candidates
// are predecessor + constant (cost step, score step, rel step and a priority code 0 / 1 / 2 for
// mismatch / deletion / insertion), and the unsigned minimum picks the cheapest candidate and,
// among equal costs, the reference's order mismatch >= deletion >= insertion (_align.pyx:462-476)
// together with its payload.  A matching character takes the diagonal unconditionally
// (:446-453).  One VGPR per adapter row instead of two.
// Everything around the cell (band, candidate rules, last-column scan, early exit, column
// skipping) is the same as in k_dp; origin = j - rel is decoded where the reference reads it.
// =============================================================================================
// index of a read character in an adapter's 32-entry match table (fused multi-adapter mode): letters only
// (bit 6 set), by their low five bits; everything else takes entry 0, which matches nothing
__device__ __forceinline__ unsigned multi_tab_index(unsigned c) { return (c & 0x40u) ? (c & 31u) : 0u; }

#define PK_COST_SHIFT 20
#define PK_PRIO_SHIFT 18
#define PK_REL_SHIFT 9
#define PK_SCORE_BIAS 256
#define PK_PRIO_MASK (3u << PK_PRIO_SHIFT)
#define PK_D_MATCH ((1u << PK_REL_SHIFT) + 1u)
#define PK_D_MIS ((1u << PK_COST_SHIFT) + (1u << PK_REL_SHIFT) - 1u)
#define PK_D_DEL ((1u << PK_COST_SHIFT) + (1u << PK_PRIO_SHIFT) - 2u)
#define PK_D_INS ((1u << PK_COST_SHIFT) + (2u << PK_PRIO_SHIFT) + (1u << PK_REL_SHIFT) - 2u)

__device__ __forceinline__ unsigned pk_cell(int cost, int rel, int score) {
    return ((unsigned)cost << PK_COST_SHIFT) | ((unsigned)rel << PK_REL_SHIFT) | (unsigned)(score + PK_SCORE_BIAS);
}
__device__ __forceinline__ int pk_cost(unsigned w) { return (int)(w >> PK_COST_SHIFT); }
__device__ __forceinline__ int pk_rel(unsigned w) { return (int)((w >> PK_REL_SHIFT) & 0x1FFu); }
__device__ __forceinline__ int pk_score(unsigned w) { return (int)(w & 0x1FFu) - PK_SCORE_BIAS; }

#ifdef CAH_DP_COUNT
__device__ unsigned long long g_dp_dbg[8];      // debug build only: lock-step accounting of k_dp_packed
#endif
// (store_result: dev_common.h)


// Round 5: the match predicate and the "cost <= k" predicate as MASKS (0 / ~0: v_bfe_i32 of the match word, the sign of
// cell - klim) and their selects as v_bitop3 -- 2-cycle forms where a compare + v_cndmask pair is 4 + 4 (DESIGN 3.2's table);
// the band predicate stays a compare: its result also feeds the wave-wide "any band left" test.  -DCAH_DPP_PLAIN: round 4's form.
#ifdef CAH_DPP_PLAIN
typedef bool dpp_pred;
__device__ __forceinline__ dpp_pred dpp_match_bit(const unsigned mword, const int bit) { return (mword & (1u << bit)) != 0u; }
__device__ __forceinline__ unsigned dpp_select(const dpp_pred p, const unsigned a, const unsigned b) { return p ? a : b; }
#else
typedef unsigned dpp_pred;
__device__ __forceinline__ dpp_pred dpp_match_bit(const unsigned mword, const int bit) {
    return (unsigned)__builtin_amdgcn_sbfe((int)mword, (unsigned)bit, 1u);
}
__device__ __forceinline__ unsigned dpp_select(const dpp_pred p, const unsigned a, const unsigned b) {
    return __builtin_amdgcn_bitop3_b32(p, a, b, 0xCA);
}
#endif

template <int I, int ROWS>
__device__ __forceinline__ void dp_rows_packed(unsigned (&w)[ROWS + 1], const unsigned mk_lo, const unsigned mk_hi,
                                               unsigned wd, int& nl, unsigned& cm_w, const int last, const int m,
                                               const unsigned klim, const dpp_pred eq, const bool in_band) {
    // eq / in_band: this row's predicates, computed at the end of the previous row.  On gfx950 a VALU
    // compare result (VCC / SGPR pair) cannot feed the very next VALU instruction (the compiler pads
    // with s_nop), so the compares of row I+1 are issued between row I's `cost <= k` compare and the
    // select that consumes it; the match bits are tested on 32-bit halves (a 64-bit mask makes the
    // compiler emit v_cmp_eq_u64).
    if constexpr (I <= ROWS) {
        if (!__any(in_band)) return;                      // no lane has band left (the predicate exists anyway:
                                                          // asking in front of every row beats every other row, -4 %)
        const unsigned wold = w[I];
        const unsigned wprev = w[I - 1];
        const unsigned a = wd + PK_D_MIS;                 // mismatch: from the diagonal
        const unsigned b = wprev + PK_D_DEL;              // deletion: from the cell above (this column)
        const unsigned c3 = wold + PK_D_INS;              // insertion: from this row, previous column
        const unsigned mn = min(min(a, b), c3) & ~PK_PRIO_MASK;
        const unsigned e = wd + PK_D_MATCH;
        const unsigned wn = dpp_select(eq, e, mn);
        w[I] = in_band ? wn : wold;                       // out of band: the stale cell stays
        if constexpr (I > ROWS - 8) {
            if (I == m) cm_w = wn;                        // wave-uniform capture of cell (m, j)
        }
        const unsigned mword = I < 32 ? mk_lo : mk_hi;    // predicates of row I + 1
        const dpp_pred eq_next = dpp_match_bit(mword, I & 31);
        const bool band_next = I + 1 <= last;
#ifdef CAH_DPP_PLAIN
        const bool ok = w[I] < klim;                      // stale cells cost > k (see k_dp)
        __builtin_amdgcn_sched_barrier(0);
        nl = ok ? I : nl;
#else
        // (cells and klim are below 2^29: the sign of the difference is the comparison)
        const unsigned okm = (unsigned)((int)(w[I] - klim) >> 31);
        __builtin_amdgcn_sched_barrier(0);
        nl = (int)__builtin_amdgcn_bitop3_b32(okm, (unsigned)I, (unsigned)nl, 0xCA);
#endif
        if constexpr ((I % CAH_SCHED_ROWS) == 0) __builtin_amdgcn_sched_barrier(0);
        dp_rows_packed<I + 1, ROWS>(w, mk_lo, mk_hi, wold, nl, cm_w, last, m, klim, eq_next, band_next);
    }
}

#define CAH_ROWP_CASE(K) case K: if constexpr (K <= ROWS) { wi = w[K]; } break;
template <int ROWS>
__device__ __forceinline__ unsigned get_row_packed(const unsigned (&w)[ROWS + 1], int i) {
    unsigned wi = 0;
    switch (__builtin_amdgcn_readfirstlane(i)) {
        CAH_ROWP_CASE(0) CAH_ROWP_CASE(1) CAH_ROWP_CASE(2) CAH_ROWP_CASE(3) CAH_ROWP_CASE(4) CAH_ROWP_CASE(5) CAH_ROWP_CASE(6) CAH_ROWP_CASE(7)
        CAH_ROWP_CASE(8) CAH_ROWP_CASE(9) CAH_ROWP_CASE(10) CAH_ROWP_CASE(11) CAH_ROWP_CASE(12) CAH_ROWP_CASE(13) CAH_ROWP_CASE(14) CAH_ROWP_CASE(15)
        CAH_ROWP_CASE(16) CAH_ROWP_CASE(17) CAH_ROWP_CASE(18) CAH_ROWP_CASE(19) CAH_ROWP_CASE(20) CAH_ROWP_CASE(21) CAH_ROWP_CASE(22) CAH_ROWP_CASE(23)
        CAH_ROWP_CASE(24) CAH_ROWP_CASE(25) CAH_ROWP_CASE(26) CAH_ROWP_CASE(27) CAH_ROWP_CASE(28) CAH_ROWP_CASE(29) CAH_ROWP_CASE(30) CAH_ROWP_CASE(31)
        CAH_ROWP_CASE(32) CAH_ROWP_CASE(33) CAH_ROWP_CASE(34) CAH_ROWP_CASE(35) CAH_ROWP_CASE(36) CAH_ROWP_CASE(37) CAH_ROWP_CASE(38) CAH_ROWP_CASE(39)
        CAH_ROWP_CASE(40) CAH_ROWP_CASE(41) CAH_ROWP_CASE(42) CAH_ROWP_CASE(43) CAH_ROWP_CASE(44) CAH_ROWP_CASE(45) CAH_ROWP_CASE(46) CAH_ROWP_CASE(47)
        CAH_ROWP_CASE(48) CAH_ROWP_CASE(49) CAH_ROWP_CASE(50) CAH_ROWP_CASE(51) CAH_ROWP_CASE(52) CAH_ROWP_CASE(53) CAH_ROWP_CASE(54) CAH_ROWP_CASE(55)
        CAH_ROWP_CASE(56) CAH_ROWP_CASE(57) CAH_ROWP_CASE(58) CAH_ROWP_CASE(59) CAH_ROWP_CASE(60) CAH_ROWP_CASE(61) CAH_ROWP_CASE(62) CAH_ROWP_CASE(63)
        CAH_ROWP_CASE(64)
        default: break;
    }
    return wi;
}

#ifndef CAH_DPP_WAVES
#define CAH_DPP_WAVES(ROWS) ((ROWS) <= 24 ? 6 : 4)
#endif
#ifndef PK_DEQUEUE
#define PK_DEQUEUE 4
#endif

template <int ROWS, bool MULTI>
__global__ __launch_bounds__(256, CAH_DPP_WAVES(ROWS)) void k_dp_packed(DpArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_rowmask[];     // [128], or [n_adapters * CAH_MULTI_TAB_STRIDE]
    if (MULTI && a.queue_count && *a.queue_count == 0ull && (!a.queue_count_back || *a.queue_count_back == 0ull)) return;
    int deq = PK_DEQUEUE;
    if (MULTI && a.queue_count) {
        // (work items are drawn from one counter, PK_DEQUEUE x 64 per wave and draw: blocks beyond what the list holds
        // leave before the tables are copied)
        const unsigned long long items = *a.queue_count + (a.queue_count_back ? *a.queue_count_back : 0ull);
        // (a short list -- a small batch -- is spread over as many waves as it has sub-batches: the kernel's duration is
        // that of its slowest wave, and a wave works its sub-batches off one after the other)
        deq = items < 1024ull * 1024ull ? 1 : PK_DEQUEUE;
        if ((unsigned long long)blockIdx.x * (unsigned)(deq * WAVE) >= items) return;
    }
    __shared__ int s_ncnt[CAH_MAX_M + 1];
    __shared__ int s_thr[CAH_MAX_M + 1];
    const CahMatcher* mt = a.matcher;
    if (MULTI) {
        for (int i = threadIdx.x; i < a.n_adapters * CAH_MULTI_TAB_STRIDE; i += blockDim.x) s_rowmask[i] = a.tab[i];
    } else {
        for (int i = threadIdx.x; i < CAH_TABLE_CHARS; i += blockDim.x) s_rowmask[i] = mt->rowmask[i];
    }
    for (int i = threadIdx.x; i <= CAH_MAX_M; i += blockDim.x) {
        s_ncnt[i] = mt->n_counts[i];
        s_thr[i] = mt->thr[i];
    }
    __syncthreads();

    const int m = mt->m, k = mt->k;
    const int min_overlap = mt->min_overlap;
    const bool wildcard_ref = mt->wildcard_ref != 0;
    const int eff_full = mt->effective_length;
    const int half_m = m / 2;
    // cost <= k  <=>  word < (k+1) << 20 (priority bits are clear in stored cells); costs never
    // exceed 64 here, so any k >= 200 behaves the same
#ifdef CAH_DPP_PLAIN
    const unsigned klim = (unsigned)(min(k, 200) + 1) << PK_COST_SHIFT;
#else
    unsigned klim = (unsigned)(min(k, 200) + 1) << PK_COST_SHIFT;
    asm volatile("" : "+v"(klim));                       // (a VGPR: as an SGPR operand it doubles the subtraction's issue time)
#endif

    const int lane = wave_lane();
    int64_t total = a.n_reads, front = a.n_reads;
    if (a.queue_count) {
        front = (int64_t)(*a.queue_count);
        total = front + (a.queue_count_back ? (int64_t)(*a.queue_count_back) : 0);
    }
    const bool skip_cols = a.queue && a.queue_keys && mt->skip_ok != 0;

    // a wave takes PK_DEQUEUE x 64 consecutive work items per atomic (a single hot counter sustains
    // ~90 atomics/us on this chip)
    for (;;) {
      const int64_t base0 = wave_dequeue(a.work_counter, deq * WAVE);
      if (base0 >= total) break;
#pragma unroll 1
      for (int sub = 0; sub < deq; ++sub) {
        const int64_t base = base0 + (int64_t)sub * WAVE;
        if (base >= total) break;
        const int64_t idx = base + lane;
        bool valid = idx < total;
        // the windowed work list is filled from both ends (see DpArgs)
        const int64_t slot = idx < front ? idx : a.queue_cap - 1 - (idx - front);
        int64_t r = 0;
        unsigned tab_base = 0, adapter = 0;
        bool pair_tail = false;
        if (MULTI) {
            int32_t qi = 0;
            if (valid) qi = a.queue[slot];
            if (valid) {
                const uint64_t pr = a.pairs[qi];
                r = (int64_t)(pr >> 32);
                adapter = (unsigned)(pr >> 8) & 0xFFFFu;
                tab_base = adapter * CAH_MULTI_TAB_STRIDE;
                pair_tail = ((unsigned)(pr >> 24) & CAH_M2_PAIR_TAIL) != 0;
            }
        } else if (valid) {
            r = a.queue ? (int64_t)a.queue[slot] : idx;
        }
        auto table_index = [&](unsigned c) -> unsigned {
            return MULTI ? tab_base + multi_tab_index(c & 0xFFu) : (c & (CAH_TABLE_CHARS - 1));
        };
        int64_t off = 0, n64 = 0;
        if (valid) read_extent(a.offsets, a.lens, a.uniform_first, a.uniform_len, r, off, n64);
        bool invalid = false;
        if (n64 > a.max_read_len) { invalid = true; n64 = 0; }
        const int n = (int)n64;
        const uint8_t* q = a.seqs + off;

        // flags = BACK: every column 1..n (_align.pyx:346-352), unless the prefilter proves the
        // first columns irrelevant (column skipping, see k_dp / DESIGN.md) or the cost scan gave the
        // exact window (back_scan.h)
        int max_n = n;
        int min_n = 0;
        bool do_scan = true;
        if (a.win) {
            if (valid) {
                min_n = a.win[2 * slot];
                const int e2 = a.win[2 * slot + 1];
                max_n = min(n, e2 >> 1);
                do_scan = (e2 & 1) != 0;
            }
        } else if (skip_cols && valid) {
            min_n = max(0, ((int)a.queue_keys[idx] << CAH_KEY_SHIFT) - m - k - 1);
        }

        // first column (:374-378): cost i, score -2i, origin = this column (rel 0)
        unsigned w[ROWS + 1];
#pragma unroll
        for (int i = 0; i <= ROWS; ++i) w[i] = i > m ? pk_cell(255, 0, 0) : pk_cell(i, 0, -2 * i);
        const unsigned w_row0 = pk_cell(0, 0, 0);         // row 0 of every column: cost 0, origin j

        const int SENT = m + n + 1;                       // :394
        int b_cost = SENT, b_origin = 0, b_score = 0, b_refstop = m, b_qstop = n;

        int last = min(m, k + 1);                         // :399
        int last_filled = 0, lf_ran = 0;
        int j = min_n;
        bool done = !valid;

        int pos = min_n;
        Chunk cur = load_chunk(q, pos, n, valid ? max_n : 0);
        Chunk nxt = load_chunk(q, pos + 16, n, valid ? max_n : 0);
        int left = 16;
#ifdef CAH_DP_COUNT
        int dbg_cols = 0, dbg_rows = 0, dbg_ideal = 0;
#endif
        unsigned bad_chars = cur.w[0] | cur.w[1] | cur.w[2] | cur.w[3];
        uint64_t mk_next = s_rowmask[table_index(cur.w[0])];
        for (;;) {
            const bool act = !done && j < max_n;
            if (!__any(act)) break;
            const uint64_t mk = mk_next;
            cur.w[0] = (cur.w[0] >> 8) | (cur.w[1] << 24);
            cur.w[1] = (cur.w[1] >> 8) | (cur.w[2] << 24);
            cur.w[2] = (cur.w[2] >> 8) | (cur.w[3] << 24);
            cur.w[3] >>= 8;
            if (--left == 0) {
                cur = nxt;
                bad_chars |= cur.w[0] | cur.w[1] | cur.w[2] | cur.w[3];
                pos += 16;
                nxt = load_chunk(q, pos + 16, n, (!done) ? max_n : 0);
                left = 16;
            }
            mk_next = s_rowmask[table_index(cur.w[0])];
            if (act) {
                ++j;
                // row 0 is the same in every column (start_in_query: cost 0, score 0, origin j);
                // it is also the diagonal of row 1
#ifdef CAH_DP_COUNT
                {
                    int wmax = 0;
                    for (int t = ROWS; t >= 1; --t) if (__any(last >= t)) { wmax = t; break; }
                    dbg_cols += 1; dbg_rows += min(ROWS, wmax + (wmax & 1)); dbg_ideal += last;
                }
#endif
                int nl = 0;                               // row 0 always has cost 0 <= k
                unsigned cm_w = w_row0;                   // (m == 0: the candidate cell is row 0)
                dp_rows_packed<1, ROWS>(w, (unsigned)mk, (unsigned)(mk >> 32), w_row0, nl, cm_w, last, m, klim,
                                        dpp_match_bit((unsigned)mk, 0), 1 <= last);
                last_filled = last;                       // :484
                if (last >= 1) lf_ran = last;
                if (nl < m) {                             // band update (:490-495)
                    last = nl + 1;
                } else {
                    last = m;
                    // candidate in the last row (:496-533); stop_in_query is set for BACK
                    const int cost = pk_cost(cm_w), origin = j - pk_rel(cm_w), score = pk_score(cm_w);
                    const int length = m + min(origin, 0);
                    int eff = length;
                    if (wildcard_ref)
                        eff = length < m ? length - (s_ncnt[m] - s_ncnt[m - length]) : eff_full;
                    const bool ok = length >= min_overlap && cost <= s_thr[eff];
                    const int best_len = m + min(b_origin, 0);
                    if (ok && (b_cost == SENT || (origin <= b_origin + half_m && score > b_score) ||
                               (length > best_len && score > b_score))) {
                        b_cost = cost; b_origin = origin; b_score = score; b_refstop = m; b_qstop = j;
                        if (cost == 0 && origin >= 0) done = true;       // exact match: stop early
                    }
                }
            }
        }
        if (bad_chars & 0x80808080u) invalid = true;
#ifdef CAH_DP_COUNT
        if (valid) { atomicAdd(&g_dp_dbg[2], (unsigned long long)dbg_ideal); atomicAdd(&g_dp_dbg[3], 1ull);
                     atomicAdd(&g_dp_dbg[5], (unsigned long long)(j - min_n)); }
        if (lane == 0) { atomicAdd(&g_dp_dbg[0], (unsigned long long)__builtin_amdgcn_readfirstlane(dbg_cols));
                         atomicAdd(&g_dp_dbg[1], (unsigned long long)__builtin_amdgcn_readfirstlane(dbg_rows));
                         atomicAdd(&g_dp_dbg[4], 1ull); }
#endif

        // last column (:536-572); max_n == n always for BACK.  Cells were written in column j (the
        // last one processed), so origin = j - rel.  The stale `origin` of :565 is that of the
        // last cell the row loop wrote (row lf_ran).
        int stale_origin = 0;
        {
            unsigned stale_w = w_row0;
#pragma unroll
            for (int i = 1; i <= ROWS; ++i) stale_w = (i == lf_ran) ? w[i] : stale_w;
            stale_origin = lf_ran >= 1 ? j - pk_rel(stale_w) : 0;
        }
#pragma unroll 1
        for (int i = m; i >= 0; --i) {
            const bool want = valid && do_scan && i <= last_filled;
            if (!__any(want)) continue;
            const unsigned wi = i == 0 ? w_row0 : get_row_packed<ROWS>(w, i);
            if (want) {
                const int o_i = j - pk_rel(wi), score = pk_score(wi), cost = pk_cost(wi);
                const int ref_start = -min(o_i, 0);
                const int length = i - ref_start;
                int eff = length;
                if (wildcard_ref)
                    eff = length < m ? length - (s_ncnt[i] - s_ncnt[ref_start]) : eff_full;
                const bool ok = length >= min_overlap && cost <= s_thr[eff];
                const int best_len = b_refstop + min(b_origin, 0);
                if (ok && (b_cost == SENT || (stale_origin <= b_origin + half_m && score > b_score) ||
                           (length > best_len && score > b_score))) {
                    b_cost = cost; b_origin = o_i; b_score = score; b_refstop = i; b_qstop = n;
                }
            }
        }

        if (MULTI) {
            bool merge = valid && b_cost != SENT && !invalid;
            // Streaming form (multi2.h, head of the file): a TAIL pair was made by OUR k-mer family; the reference's
            // kmers_present follows for every match whose alignment begins within its error class's last row's reach --
            // a match that begins further back (insertions in a row near the class's end; rare) is checked against the
            // reference's own tail k-mers of that adapter
            if (a.m2_hdr && merge && pair_tail) {
                const int qs = max(b_origin, 0);
                if (n - qs > (int)a.m2_hdr->lmax_row[b_refstop])
                    merge = m2_ref_present(a.m2_ref_list, a.m2_ref_begin[adapter], a.m2_ref_begin[adapter + 1], q, n, a.m2_hdr->ref_span);
            }
            if (merge)
                atomicMax(a.best_key + r, pack_best(b_score, b_cost, (int)adapter, b_refstop, max(b_origin, 0), b_qstop));
        } else if (valid) {
            const bool found = b_cost != SENT && !invalid;
            store_result(a.out6, a.status, a.best_adapter, a.adapter_index, a.merge_best, r, invalid, found,
                         b_origin >= 0 ? 0 : -b_origin, b_refstop, b_origin >= 0 ? b_origin : 0, b_qstop, b_score, b_cost);
        }
      }
    }
}

// =============================================================================================
// k_back_scan: bit-parallel cost scan + classification of 3' adapter reads (back_scan.h) in front of
// k_dp_packed.  One read per lane; the whole column of vertical deltas is one 64-bit word, so a
// column costs ~40 VALU instructions whatever the band -- about a tenth of the cell loop -- and all
// lanes of a wave do the same work (no band divergence).  Reads that end as NONE / EXACT_FULL /
// EXACT_TAIL are finished here; the rest go to the cell DP with an exact column window.
// Work is taken in tiles of SCAN_TILE entries per workgroup (one atomic per tile); the DP work list of
// a tile is collected in LDS and appended with two atomics: entries with a bounded window from the
// front, entries that run to the read end from the back (lanes of one DP wave then do alike work).
// =============================================================================================
#ifndef SCAN_TILE
#define SCAN_TILE 1024
#endif
#ifndef SCAN_AHEAD
#define SCAN_AHEAD 2
#endif

#ifdef SCAN_TRACE
// developer build only: s_memtime stamps of one wave of the cost scan, per sub-batch of 64 work items
#define SCAN_TRACE_N 256
__device__ unsigned long long g_scan_trace[SCAN_TRACE_N * 8];
__device__ unsigned g_scan_trace_n;
extern "C" int cah_debug_scan_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan_trace), sizeof(g_scan_trace)) == hipSuccess ? 0 : 1;
}
#define SCAN_STAMP(st, extra) do { if (blockIdx.x == 11 && wave == 1 && trace_i < SCAN_TRACE_N) { \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        if (lane == 0) { g_scan_trace[trace_i * 8 + (st)] = t_; if ((st) == 2) g_scan_trace[trace_i * 8 + 6] = (unsigned long long)(extra); } } } while (0)
#else
#define SCAN_STAMP(st, extra) do { } while (0)
#endif

// The kernel's arguments re-read from the kernarg segment where they are used (the result stores, the straggler list, the
// DP work list -- once per sub-batch or per tile): as plain arguments they sit in SGPRs through the column loops, which
// have none to spare (186 spilled, v_readlane / v_writelane around every sub-batch).  k_filter_stream2's trick.
typedef const __attribute__((address_space(4))) ScanArgs* scan_kernarg_ptr;
__device__ __forceinline__ scan_kernarg_ptr scan_kernargs() {
    scan_kernarg_ptr kp = (scan_kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));                                        // keeps the loads where the values are used
    return kp;
}

// MULTI: the work list holds (read, adapter, key) pairs of the fused multi-adapter prefilter; all adapters
// have one shape (m, k, thresholds: matcher 0), only the match table differs per lane.
// KIND: the form of the column (back_scan.h, bs_kind_of): 0 = one 64-bit word, 1 = one 32-bit word (adapters up to 32
// characters), 2 / 3 = a 32-bit word + 1 / 2 explicit rows (33 / 34 characters).  The 32-bit forms cost about half the
// instructions per column.
template <bool MULTI, int KIND>
#ifndef CAH_SCAN_WAVES
#define CAH_SCAN_WAVES 5           // (developer builds: -DCAH_SCAN_WAVES=4 -- 118 VGPRs and no scratch instead of 96 and 12 spilled)
#endif
__global__ __launch_bounds__(256, (KIND == 3 || KIND == 0) ? 4 : CAH_SCAN_WAVES) void k_back_scan(ScanArgs a) {
    constexpr int XR = KIND >= 2 ? KIND - 1 : 0;
    extern __shared__ __attribute__((aligned(16))) uint64_t s_scanmask[];   // [128], or [n_adapters * CAH_MULTI_TAB_STRIDE]
    __shared__ int s_thr_last[CAH_MAX_M + 1];
    __shared__ int s_list[SCAN_TILE * 3];          // (read, first column, last column * 2 + scan): front entries
                                                   // from slot 0 up, back entries from the last slot down
    __shared__ unsigned s_nf, s_nb;
    __shared__ long long s_tile;
    __shared__ unsigned long long s_gf, s_gb;
    // one adapter: a STATIC 256-entry table (its address folds into the ds_read offset, the entry offset is
    // "byte << 3": one SDWA instruction per column; bytes >= 0x80 find zeros and are flagged anyway)
    __shared__ __attribute__((aligned(16))) uint64_t s_sm256[MULTI ? 1 : 256];
    const CahMatcher* mt = a.matcher;
    if (MULTI) {
        for (int i = threadIdx.x; i < a.n_adapters * CAH_MULTI_TAB_STRIDE; i += blockDim.x)
            s_scanmask[i] = KIND == 0 ? a.tab[i] : bs32_table_entry(a.tab[i], mt->m);   // (all adapters have one shape)
    } else {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) {
            const uint64_t sm = i < CAH_TABLE_CHARS ? mt->scanmask[i] : 0ull;
            s_sm256[i] = KIND == 0 ? sm : bs32_table_entry(sm, mt->m);
        }
    }
    for (int i = threadIdx.x; i <= CAH_MAX_M; i += blockDim.x) s_thr_last[i] = mt->thr_last[i];
    BackScanParams p;
    p.m = mt->m; p.k = mt->k; p.kacc = mt->kacc; p.min_overlap = mt->min_overlap; p.half_m = mt->m / 2;
    const int lane = wave_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform, and said so
    int64_t total = a.n_reads;
    if (a.queue_count) total = (int64_t)(*a.queue_count);
    if (a.queue_limit > 0 && total > a.queue_limit) total = a.queue_limit;
    const bool skip_cols = MULTI ? a.multi_skip_ok != 0 : (a.queue && a.queue_keys && mt->skip_ok != 0);
    const int stop_gap = bs_stop_gap(p);
    const int tile = a.tile;                               // entries per workgroup and atomic: 256 .. SCAN_TILE
#ifdef SCAN_TRACE
    int trace_i = 0;
#endif

    for (;;) {
        __syncthreads();                                   // previous tile flushed
        if (threadIdx.x == 0) {
            s_tile = (long long)atomicAdd(a.work_counter, (unsigned long long)tile);
            s_nf = 0; s_nb = 0;
        }
        __syncthreads();
        const int64_t tile_base = s_tile;
        if (tile_base >= total) break;

        // (single-adapter mode) the queue entry and the read's extent of the wave's NEXT sub-batch are fetched while
        // this one is scanned: three dependent round trips (queue -> offsets -> characters) in front of every
        // sub-batch were a ninth of the kernel's time
        bool pf_have = false, pf_valid = false;
        int pf_r = 0, pf_n = 0;
        unsigned pf_key = 0;
        int64_t pf_off = 0;
        for (int sub = wave; sub < tile / WAVE; sub += 4) {
            const int64_t base = tile_base + (int64_t)sub * WAVE;
            if (base >= total) break;
            SCAN_STAMP(0, 0);
            const int64_t idx = base + lane;
            bool valid = idx < total;
            int64_t r = 0;
            unsigned tab_base = 0, adapter = 0, key = 0;
            if (MULTI) {
                if (valid) {
                    const uint64_t pr = a.pairs[idx];
                    r = (int64_t)(pr >> 32);
                    adapter = (unsigned)(pr >> 8) & 0xFFFFu;
                    key = (unsigned)pr & 0xFFu;
                    tab_base = adapter * CAH_MULTI_TAB_STRIDE;
                }
            } else if (!pf_have) {
                const scan_kernarg_ptr kq = scan_kernargs();
                if (valid) r = kq->queue ? (int64_t)kq->queue[idx] : idx;
                // (a slot of the straggler list that its wave reserved but could not use: the list was full)
                if (r < 0) { valid = false; r = 0; }
                if (skip_cols && valid) key = kq->queue_keys[idx];
            }
            int64_t off = 0, n64 = 0;
            const scan_kernarg_ptr kr = scan_kernargs();
            if (!MULTI && pf_have) { valid = pf_valid; r = pf_r; key = pf_key; off = pf_off; n64 = pf_n; }
            else if (valid) read_extent(kr->offsets, kr->lens, kr->uniform_first, kr->uniform_len, r, off, n64);
            bool invalid = false;
            if (n64 > kr->max_read_len) { invalid = true; n64 = 0; }
            const int n = (int)n64;
            const uint8_t* q = kr->seqs + off;
            // first column of the window (column skipping): nothing of the whole-read k-mer set
            // ends before 4 * key, so no row-m cost <= k occurs before it
            int j0 = 0;
            if (skip_cols && valid) j0 = max(0, ((int)key << CAH_KEY_SHIFT) - p.m - p.k - 1);
            // ... moved back so that the window is a whole number of 16-column chunks (an earlier start is as
            // exact, and the wave's chunk count -- the longest window's -- stays what it was): the last chunk then
            // ends at the read end and takes the unguarded path below
            if constexpr (!MULTI) j0 = bs_align_window(j0, n);
            // the match word of character t of a chunk
            auto eq_of = [&](const Chunk& ck, int t) -> uint64_t {
                if constexpr (MULTI) {
                    return s_scanmask[tab_base + multi_tab_index(chunk_byte(ck, t) & 0xFFu)];
                } else {
                    const unsigned byte_off = ((ck.w[t >> 2] >> (8 * (t & 3))) & 0xFFu) << 3;
                    return *reinterpret_cast<const uint64_t*>(reinterpret_cast<const unsigned char*>(s_sm256) + byte_off);
                }
            };

            typename std::conditional<KIND == 0, BackScanState, BackScanState32<XR>>::type st;
            if constexpr (KIND == 0) bs_init(st, p); else bs32_init(st, p);
            // one column: the character's table entry is the 64-bit match word, or {rows 1..32, rows 33..}
            auto step = [&](const uint64_t eq, const int jj, const int joff = 0) -> bool {
                if constexpr (KIND == 0) return bs_step<!MULTI>(st, eq, jj + joff, p);
                else return bs32_step<!MULTI, XR>(st, (uint32_t)eq, (uint32_t)(eq >> 32), jj, p, 0x7FFFFFFF, joff);
            };
            int j = j0, exact_j = 0;
            bool done = !valid, exact = false, stopped = false, retry = false, valid_out = valid;
            int retry_at = MULTI ? 0 : a.retry_threshold;
            // one 16-character chunk per iteration (per lane: its own window start), the next chunk requested
            // before this one is consumed, the LDS lookup of the next column's match word issued one column
            // ahead; bytes are taken with constant shifts (v_bfe), the "any lane left?" test costs one scalar
            // branch per chunk
            int pos = j0;
            Chunk cur = load_chunk(q, pos, n, valid ? n : 0);
            unsigned bad_chars = 0;
            int n_chunks = 0;
            // the next sub-batch's queue entries (looked at when the scan of this one is over)
            const int64_t idx_n = base + 4 * WAVE + lane;
            const bool next_sub = !MULTI && sub + 4 < tile / WAVE && base + 4 * WAVE < total;      // wave-uniform
            bool valid_n = false;
            int r_n = 0;
            unsigned key_n = 0;
            if (next_sub) {
                valid_n = idx_n < total;
                if (valid_n) r_n = kr->queue ? kr->queue[idx_n] : (int)idx_n;
                if (skip_cols && valid_n) key_n = kr->queue_keys[idx_n];
            }
            SCAN_STAMP(1, 0);
            for (;;) {
                ++n_chunks;
                const unsigned long long act = __ballot(!done && j < n);
                if (!act) break;
                if constexpr (!MULTI) {
                    // stragglers: a wave costs the same with 3 lanes at work as with 64.  When few lanes are left
                    // and they have a long way to go (false-positive k-mer hits far from the read end, mostly),
                    // they are set aside and scanned again from their window start by a second launch, packed
                    // 64 to a wave, instead of holding this wave to the read end.
                    if (retry_at > 0 && __popcll(act) <= retry_at) {
                        const bool far = !done && n - j >= 32;
                        const unsigned long long bfar = __ballot(far);
                        if (bfar) {
                            const int cnt = __popcll(bfar);
                            unsigned long long slot = 0;
                            const scan_kernarg_ptr ka = scan_kernargs();
                            if (lane == 0) slot = atomicAdd(ka->retry_count, (unsigned long long)cnt);
                            slot = __shfl(slot, 0, WAVE);
                            if ((int64_t)(slot + cnt) <= ka->retry_cap) {
                                if (far) {
                                    const int64_t e = (int64_t)slot + __popcll(bfar & ((1ull << lane) - 1ull));
                                    ka->retry_queue[e] = (int32_t)r;
                                    ka->retry_keys[e] = (uint8_t)key;
                                    done = true; retry = true;
                                }
                                if (bfar == act) break;
                            } else {
                                // The list is full: this wave runs to the end.  What it reserved inside the list is
                                // marked unused -- the second launch walks min(count, capacity) slots and must not
                                // meet whatever the scratch held before.
                                if (far) {
                                    const int64_t e = (int64_t)slot + __popcll(bfar & ((1ull << lane) - 1ull));
                                    if (e < ka->retry_cap) ka->retry_queue[e] = -1;
                                }
                                retry_at = 0;
                            }
                        }
                    }
                }
                const Chunk nxt = load_chunk(q, pos + 16, n, (!done) ? n : 0);
                bad_chars |= cur.w[0] | cur.w[1] | cur.w[2] | cur.w[3];
                // the match words of the chunk's characters come from LDS; SCAN_AHEAD lookups are in flight ahead of the
                // column being computed (one was not enough once a column took ~100 cycles: the scan waited on LDS)
                uint64_t eqq[SCAN_AHEAD];
#pragma unroll
                for (int t = 0; t < SCAN_AHEAD; ++t) eqq[t] = eq_of(cur, t);
                if (__all(done || j + 16 <= n)) {
                    // every lane still at work has the whole chunk ahead of it (the queue is ordered by window
                    // start, so this is the rule): no per-column guards.  Lanes that are done -- as EXACT_FULL, or
                    // idle from the start -- step along on NUL chunks; their state is not looked at again.
                    // (the column number is only needed where a column is booked: j is counted per chunk, not per column)
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const uint64_t eq = eqq[t % SCAN_AHEAD];
                        if (t + SCAN_AHEAD < 16) eqq[t % SCAN_AHEAD] = eq_of(cur, t + SCAN_AHEAD);
                        if (step(eq, j, t + 1) && !exact) { exact = true; exact_j = j + t + 1; }
                    }
                    j += 16;
                    if (exact) done = true;
                } else {
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const uint64_t eq = eqq[t % SCAN_AHEAD];
                        if (t + SCAN_AHEAD < 16) eqq[t % SCAN_AHEAD] = eq_of(cur, t + SCAN_AHEAD);
                        if (!done && j < n) {
                            ++j;
                            if (step(eq, j)) { exact = true; exact_j = j; done = true; }
                        }
                    }
                }
                pos += 16;
                cur = nxt;
                if constexpr (!MULTI) {
                    // nothing beyond the last acceptable candidate can matter any more (back_scan.h, "early stop")
                    if (a.early_stop && !done && bs_may_stop(st, j, n, stop_gap)) { done = true; stopped = true; }
                }
            }
            if (bad_chars & 0x80808080u) invalid = true;
            pf_have = next_sub;
            if (next_sub) {
                if (r_n < 0) { valid_n = false; r_n = 0; }
                int64_t o_n = 0, n_n = 0;
                const scan_kernarg_ptr kn = scan_kernargs();
                if (valid_n) read_extent(kn->offsets, kn->lens, kn->uniform_first, kn->uniform_len, (int64_t)r_n, o_n, n_n);
                // (a read beyond max_read_len keeps its length here: the next sub-batch flags it as this code would)
                pf_valid = valid_n; pf_r = r_n; pf_key = key_n; pf_off = o_n;
                pf_n = n_n > (int64_t)0x7FFFFFFF ? 0x7FFFFFFF : (int)n_n;
            }
            SCAN_STAMP(2, n_chunks);

            int o0 = 0, o1 = 0;
            int cls = BS_NONE;
            // the classification -- above all the walk down the last column's rows -- only for lanes whose result it is:
            // a lane that ended as EXACT_FULL, was set aside or never had a read skips it, and a wave made of such lanes
            // (the common wave of reads with the adapter inside) skips the row loop altogether
            if (valid && !exact && !retry) {
                if constexpr (KIND == 0) cls = bs_finish<!MULTI>(st, n, j0, p, [&](int i) { return s_thr_last[i]; }, o0, o1, stopped);
                else cls = bs32_finish<XR, !MULTI>(st, n, j0, p, [&](int i) { return s_thr_last[i]; }, o0, o1, stopped);
            }
            if (exact) { cls = BS_EXACT_FULL; o0 = exact_j; }
            if (retry) { cls = BS_NONE; valid_out = false; }
            SCAN_STAMP(3, 0);
            if (MULTI) {
                // invalid reads were flagged by the prefilter (it sees every character); matches are merged
                // with one atomic max on the read's best key
                if (valid && !invalid) {
                    if (cls == BS_EXACT_FULL) atomicMax(a.best_key + r, pack_best(p.m, 0, (int)adapter, p.m, o0 - p.m, o0));
                    else if (cls == BS_EXACT_TAIL) atomicMax(a.best_key + r, pack_best(o0 - 2 * o1, o1, (int)adapter, o0, n - o0, n));
                    else if (cls == BS_SUBS_FULL) atomicMax(a.best_key + r, pack_best(p.m - 2 * o1, o1, (int)adapter, p.m, o0 - p.m, o0));
                }
            } else if (valid_out) {
                // the tuple of every class the scan finishes, put together with selects: ONE store sequence for the
                // wave instead of one per class (as divergent branches: up to five times the store instructions)
                const bool full = cls == BS_EXACT_FULL || cls == BS_SUBS_FULL || cls == BS_INDEL1_FULL;
                const bool found = !invalid && (full || cls == BS_EXACT_TAIL);
                int t1 = p.m, t2 = o0 - p.m, t3 = o0, sc = p.m, cost = 0;                 // EXACT_FULL
                if (cls == BS_EXACT_TAIL) { t1 = o0; t2 = n - o0; t3 = n; sc = o0 - 2 * o1; cost = o1; }   // row o0, o1 substitutions
                if (cls == BS_SUBS_FULL) { sc = p.m - 2 * o1; cost = o1; }
                if (cls == BS_INDEL1_FULL) {              // o1 = cost * 2 + (1: one deletion, 0: one insertion)
                    t2 = o0 - p.m + ((o1 & 1) ? 1 : -1); sc = p.m - 2 * (o1 >> 1) - (o1 & 1); cost = o1 >> 1;
                }
                if (invalid || cls != BS_DP) {
                    const scan_kernarg_ptr ka = scan_kernargs();
                    store_result(ka->out6, ka->status, ka->best_adapter, ka->adapter_index, ka->merge_best, r, invalid, found,
                                 0, t1, t2, t3, sc, cost);
                }
            }
            const bool to_dp = valid_out && !invalid && cls == BS_DP;
            const bool to_back = to_dp && (o1 & 1);
            const bool to_front = to_dp && !(o1 & 1);
            const unsigned long long bf = __ballot(to_front), bb = __ballot(to_back);
            if (bf | bb) {
                unsigned sf = 0, sb = 0;
                if (lane == 0) {
                    if (bf) sf = atomicAdd(&s_nf, (unsigned)__popcll(bf));
                    if (bb) sb = atomicAdd(&s_nb, (unsigned)__popcll(bb));
                }
                sf = __builtin_amdgcn_readfirstlane(sf);
                sb = __builtin_amdgcn_readfirstlane(sb);
                const unsigned long long below = (1ull << lane) - 1ull;
                const int item = MULTI ? (int)idx : (int)r;          // multi: the cell DP looks the pair up again
                if (to_front) {
                    const int e = (int)sf + __popcll(bf & below);
                    s_list[3 * e] = item; s_list[3 * e + 1] = o0; s_list[3 * e + 2] = o1;
                } else if (to_back) {
                    const int e = SCAN_TILE - 1 - ((int)sb + __popcll(bb & below));
                    s_list[3 * e] = item; s_list[3 * e + 1] = o0; s_list[3 * e + 2] = o1;
                }
            }
            SCAN_STAMP(4, 0);
#ifdef SCAN_TRACE
            if (blockIdx.x == 11 && wave == 1) ++trace_i;
#endif
        }

        // flush the tile's DP work list
        __syncthreads();
        const unsigned nf = s_nf, nb = s_nb;
        const scan_kernarg_ptr kf = scan_kernargs();
        if (threadIdx.x == 0) {
            s_gf = nf ? atomicAdd(kf->dp_count_front, (unsigned long long)nf) : 0ull;
            s_gb = nb ? atomicAdd(kf->dp_count_back, (unsigned long long)nb) : 0ull;
        }
        __syncthreads();
        const unsigned long long gf = s_gf, gb = s_gb;
        for (unsigned e = threadIdx.x; e < nf; e += blockDim.x) {
            const int64_t slot = (int64_t)(gf + e);
            kf->dp_queue[slot] = s_list[3 * e];
            kf->dp_win[2 * slot] = s_list[3 * e + 1];
            kf->dp_win[2 * slot + 1] = s_list[3 * e + 2];
        }
        for (unsigned e = threadIdx.x; e < nb; e += blockDim.x) {
            const int64_t slot = kf->dp_cap - 1 - (int64_t)(gb + e);
            const int le = SCAN_TILE - 1 - (int)e;
            kf->dp_queue[slot] = s_list[3 * le];
            kf->dp_win[2 * slot] = s_list[3 * le + 1];
            kf->dp_win[2 * slot + 1] = s_list[3 * le + 2];
        }
    }
}

// =============================================================================================
// k_anchored_exact: Aligner.locate of an ANCHORED adapter that tolerates no error -- flags = QUERY_STOP (5',
// "^ADAPTER", Where.PREFIX) or QUERY_START (3', "ADAPTER$", Where.SUFFIX) and thr[effective_length] == 0 (short
// adapters, adapters made mostly of N wildcards such as "^NNNNNNNNACGTACGT": the threshold counts the non-N
// characters, reference _align.pyx:505-513).  A candidate then needs cost 0, and an alignment of cost 0 that starts
// at (0, 0) (prefix; the only last-row cell a cost-0 path reaches is (m, m), :496-533, where the loop breaks) or ends
// in (m, n) (suffix: the last-column scan looks at row m only, :536-572) is the adapter itself, character by
// character under the matcher's compare mode: the result is (0, m, 0, m, m, 0) / (0, m, n - m, n, m, 0) or None,
// whatever the indel cost -- no cell is needed.  Bytes >= 0x80 are flagged over the columns the DP would have read
// (the first / last min(n, m + k) characters).
// =============================================================================================
__global__ __launch_bounds__(256) void k_anchored_exact(DpArgs a) {
    __shared__ uint64_t s_rowmask[CAH_TABLE_CHARS];
    const CahMatcher* mt = a.matcher;
    for (int i = threadIdx.x; i < CAH_TABLE_CHARS; i += blockDim.x) s_rowmask[i] = mt->rowmask[i];
    __syncthreads();
    const int m = mt->m, k = mt->k;
    const bool suffix = mt->flags == 2;
    const int lane = wave_lane();
    int64_t total = a.n_reads;
    if (a.queue_count) total = (int64_t)(*a.queue_count);
    for (;;) {
      const int64_t base0 = wave_dequeue(a.work_counter, CAH_DEQUEUE * WAVE);
      if (base0 >= total) break;
      for (int sub = 0; sub < CAH_DEQUEUE; ++sub) {
        const int64_t base = base0 + (int64_t)sub * WAVE;
        if (base >= total) break;
        const int64_t idx = base + lane;
        if (idx >= total) continue;
        const int64_t r = a.queue ? (int64_t)a.queue[idx] : idx;
        int64_t off, n64;
        read_extent(a.offsets, a.lens, a.uniform_first, a.uniform_len, r, off, n64);
        bool invalid = false;
        if (n64 > a.max_read_len) { invalid = true; n64 = 0; }
        const int n = (int)n64;
        const uint8_t* q = a.seqs + off;
        const int span = min(n, m + k);                    // the columns Aligner.locate reads (:348-352)
        const int qbeg = suffix ? n - span : 0;
        const int shift = suffix ? span - m : 0;           // region offset rel <-> adapter position rel - shift
        bool differ = n < m;
        unsigned seen = 0;
#pragma unroll 1
        for (int c0 = 0; c0 < CAH_MAX_M + 16; c0 += 16) {
            if (!__any(c0 < span)) break;
            if (c0 < span) {
                const Chunk ck = load_chunk(q, qbeg + c0, n, qbeg + span);
                seen |= ck.w[0] | ck.w[1] | ck.w[2] | ck.w[3];
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int rel = c0 + t;
                    const int i = rel - shift;
                    const bool inside = rel < span && i >= 0 && i < m;
                    const uint64_t mk = s_rowmask[chunk_byte(ck, t) & (CAH_TABLE_CHARS - 1)];
                    differ |= inside && !((mk >> (inside ? i : 0)) & 1ull);
                }
            }
        }
        if (seen & 0x80808080u) invalid = true;
        const bool found = !invalid && !differ;
        store_result(a.out6, a.status, a.best_adapter, a.adapter_index, a.merge_best, r, invalid, found,
                     0, m, suffix ? n - m : 0, suffix ? n : m, m, 0);
      }
    }
}

// =============================================================================================
// k_tiny: prefilter + cost scan of up to 64 reads in ONE launch of one wave -- the per-read API of the reference
// (Adapter.match_to(read), Aligner.locate(read)) is a batch of one, and three launches (prefilter, scan, cell DP)
// plus their queues cost more than the work.  Lane = read.  The prefilter is the ragged lean machinery, the scan
// the same bs_* code as k_back_scan; reads the scan cannot finish go to the DP work list exactly as k_back_scan
// leaves them, and *need_dp (mapped host memory) tells the host whether the cell-DP kernel has anything to do.
// lean == NULL: no prefilter (Aligner.locate).  Eligible plans: one 3' aligner with the cost scan (scan_ok).
// =============================================================================================
// completion ticket behind other kernels of a stream (the host polls mapped memory instead of waiting on the stream)
__global__ void k_ticket(int32_t* done, int32_t ticket) {
    __threadfence_system();
    __hip_atomic_store(done, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

#define TINY_STAGE_BYTES 4096
template <int DL, int NL, int NG>
__global__ __launch_bounds__(64) void k_tiny(TinyArgs a) {
    // every table the kernel reads, as ONE block: a call copies it from the image the first call of the plan left in
    // HBM (a.image; 16 bytes per lane and load) instead of deriving it from the plan's structures again
    struct __attribute__((aligned(16))) TinyTables {
        uint32_t tab[LeanLayout<DL, NL, NG>::WORDS];
        uint32_t gate[NG * CAH_GATE_LEN];
        uint64_t scanmask[CAH_TABLE_CHARS];
        int thr_last[CAH_MAX_M + 4];
    };
    static_assert(sizeof(TinyTables) % 16 == 0 && sizeof(TinyTables) <= CAH_TINY_IMAGE_BYTES, "k_tiny: table image");
    __shared__ TinyTables s_t;
    uint32_t* const s_tab = s_t.tab;
    uint32_t* const s_gate = s_t.gate;
    uint64_t* const s_scanmask = s_t.scanmask;
    int* const s_thr_last = s_t.thr_last;
    const CahMatcher* mt = a.matcher;
    const CahLeanFilter* lf = a.lean;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    if (a.image) {
        const u32x4* src = reinterpret_cast<const u32x4*>(a.image);
        u32x4* dst = reinterpret_cast<u32x4*>(&s_t);
        for (int i = threadIdx.x; i < (int)(sizeof(TinyTables) / 16); i += blockDim.x) dst[i] = src[i];
    } else {
        if (lf) {
            lean_tables_to_lds<DL, NL, NG>(lf, s_tab);
            for (int i = threadIdx.x; i < NG * CAH_GATE_LEN; i += blockDim.x)
                s_gate[i] = lf->gate_init[i / CAH_GATE_LEN][i % CAH_GATE_LEN];
        } else {
            for (int i = threadIdx.x; i < (int)(LeanLayout<DL, NL, NG>::WORDS + NG * CAH_GATE_LEN); i += blockDim.x) s_tab[i] = 0;
        }
        for (int i = threadIdx.x; i < CAH_TABLE_CHARS; i += blockDim.x) s_scanmask[i] = mt->scanmask[i];
        for (int i = threadIdx.x; i < CAH_MAX_M + 4; i += blockDim.x) s_thr_last[i] = i <= CAH_MAX_M ? mt->thr_last[i] : 0;
    }
    __syncthreads();
    if (a.image_out) {                                           // the plan's first call: leave the image, nothing else
        const u32x4* src = reinterpret_cast<const u32x4*>(&s_t);
        u32x4* dst = reinterpret_cast<u32x4*>(a.image_out);
        for (int i = threadIdx.x; i < (int)(sizeof(TinyTables) / 16); i += blockDim.x) dst[i] = src[i];
        return;
    }
    const int lane = wave_lane();
    const int64_t r = lane;
    const bool valid = r < a.n_reads;
    int64_t off = 0, n64 = 0;
    if (valid) read_extent(a.offsets, nullptr, r, off, n64);
    bool invalid = false;
    if (n64 > a.max_read_len) { invalid = true; n64 = 0; }
    const int n = (int)n64;
    // The reads may sit in mapped HOST memory (the one-read path queues no copies): every load from there is a
    // trip over PCIe, so a small batch is brought into LDS once, with the whole wave, and matched from there
    // (behind 32 NUL bytes: see the one-read prefilter below).
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[32 + TINY_STAGE_BYTES + 32];
    const int64_t batch_bytes = a.offsets[a.n_reads] - a.offsets[0];
    const bool staged = batch_bytes <= TINY_STAGE_BYTES;
    if (staged) {
        const uint8_t* src = a.seqs + a.offsets[0];
        if (threadIdx.x < 32) s_stage[threadIdx.x] = 0;
        for (int i = threadIdx.x; i < (int)batch_bytes; i += blockDim.x) s_stage[32 + i] = src[i];
        __syncthreads();
    }
    const uint8_t* q = staged ? (const uint8_t*)s_stage + 32 + (off - a.offsets[0]) : a.seqs + off;

    // ---- prefilter (the loop of k_filter_lean<false, ..>) ------------------------------------------------
    // ONE read (the per-read API): all 64 lanes work on it -- lane l owns characters [16 l, 16 l + 16) and warms its
    // words up on the 32 characters in front of them (a k-mer is at most 32 characters long, so after them every
    // state bit is exact; in front of the read there are NULs).  To the word machinery that is a batch of 64
    // "reads": the suffix of (32 NULs + read) from 16 l on, of which only three chunks are run and only the third
    // one's ends count.  The tail gates look at the distance from the true end, which the suffix keeps.  An end
    // counted again by a later lane (accumulated words) names a later position; the earliest one is taken.
    // Needs no head windows (3' adapters have none).
    const int n0 = (int)(a.offsets[1] - a.offsets[0]);                     // wave-uniform
    const bool par = lf && a.n_reads == 1 && staged && n0 > 16 && n0 <= 1008 && lf->head_span == 0 &&
                     n0 <= a.max_read_len;
    bool hit = valid && !invalid;
    int hit_pos = 0;
    if (lf) {
        LeanWords<NL, NG> L;
        lean_words_init<NL, NG>(L, lf, s_tab, s_gate);
        // the filter's view of "its" read
        const uint8_t* qf = q;
        int nf = n;
        bool validf = valid;
        if (par) {
            qf = (const uint8_t*)s_stage + 16 * lane;
            nf = n0 + 32 - 16 * lane;
            validf = 16 * lane < n0;
            if (!validf) nf = 0;
        }
        int n_max = nf;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) n_max = max(n_max, __shfl_xor(n_max, d, WAVE));
        n_max = __builtin_amdgcn_readfirstlane(n_max);
        if (par) n_max = min(n_max, 48);
        hit = false;
        unsigned seen = 0;
        LeanState<NL, NG> S;
#pragma unroll
        for (int w = 0; w < NL; ++w) { S.RL[w] = 0; S.accL[w] = 0; }
#pragma unroll
        for (int w = 0; w < NG; ++w) { S.RG[w] = 0; S.accG[w] = 0; }
        Chunk cur = load_chunk(qf, 0, nf, validf ? nf : 0);
        Chunk nxt = load_chunk(qf, 16, nf, validf ? nf : 0);
        {
            unsigned ad[8];
            lean_addr8<LeanLayout<DL, NL, NG>::LEAD_SHIFT>(ad, cur.w[0], cur.w[1]);
            lean_issue_lead<DL, NL, NG>(L, S.mk0, ad);
        }
        auto step = [&](auto gated, int pos) -> bool {
            const bool live = validf && !hit && pos < nf;
            if (!__any(live)) return false;
            const Chunk nx2 = load_chunk(qf, pos + 32, nf, live ? nf : 0);
            if (!par || pos == 32) seen |= cur.w[0] | cur.w[1] | cur.w[2] | cur.w[3];
            uint32_t gg[3];
            Chunk got;
            const uint32_t found = lean_chunk<false, decltype(gated)::value, DL, NL, NG>(
                L, S, cur, [&]() { return nxt; }, got, pos, nf, gg);
            if (live && found != 0 && (!par || pos == 32)) {
                hit = true;
                hit_pos = pos + (gg[0] ? 0 : gg[1] ? 4 : gg[2] ? 8 : 12);
            }
            cur = got;
            nxt = nx2;
            return true;
        };
        int pos = 0;
        bool more = true;
        for (; more && pos < n_max && pos < L.head_span; pos += 16) more = step(std::true_type{}, pos);
        for (; more && pos < n_max && lean_chunk_ungated<false, NL, NG>(L, pos, nf); pos += 16) more = step(std::false_type{}, pos);
        for (; more && pos < n_max; pos += 16) more = step(std::true_type{}, pos);
        if (par) {
            // back to one read on lane 0: any end, the earliest position, every byte seen
            int first = hit ? 16 * lane + (hit_pos - 32) : 0x7fffffff;
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) {
                first = min(first, __shfl_xor(first, d, WAVE));
                seen |= __shfl_xor(seen, d, WAVE);
            }
            hit = lane == 0 && first != 0x7fffffff;
            hit_pos = hit ? first : 0;
        }
        if (seen & 0x80808080u) invalid = true;
        hit = hit && !invalid;
    }

    // ---- cost scan (the body of k_back_scan<false>) -------------------------------------------------------
    BackScanParams p;
    p.m = mt->m; p.k = mt->k; p.kacc = mt->kacc; p.min_overlap = mt->min_overlap; p.half_m = mt->m / 2;
    const bool scanning = hit;                                   // lanes whose read goes on
    int j0 = 0;
    if (lf && mt->skip_ok != 0 && scanning)
        j0 = max(0, (min(hit_pos >> CAH_KEY_SHIFT, CAH_QUEUE_BINS - 1) << CAH_KEY_SHIFT) - p.m - p.k - 1);
    BackScanState st;
    bs_init(st, p);
    int j = j0, exact_j = 0;
    bool done = !scanning, exact = false, stopped = false;
    const int stop_gap = bs_stop_gap(p);
    {
        int pos = j0;
        Chunk cur = load_chunk(q, pos, n, scanning ? n : 0);
        unsigned bad_chars = 0;
        for (;;) {
            if (!__any(!done && j < n)) break;
            const Chunk nxt = load_chunk(q, pos + 16, n, (!done) ? n : 0);
            bad_chars |= cur.w[0] | cur.w[1] | cur.w[2] | cur.w[3];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const uint64_t eq = s_scanmask[chunk_byte(cur, t) & (CAH_TABLE_CHARS - 1)];
                if (!done && j < n) {
                    ++j;
                    if (bs_step<true>(st, eq, j, p)) { exact = true; exact_j = j; done = true; }
                }
            }
            pos += 16;
            cur = nxt;
            // (with a prefilter in front, which has seen every character of the read: back_scan.h, "early stop")
            if (lf && !done && bs_may_stop(st, j, n, stop_gap)) { done = true; stopped = true; }
        }
        if (bad_chars & 0x80808080u) invalid = true;
    }
    int o0 = 0, o1 = 0;
    int cls = bs_finish(st, n, j0, p, [&](int i) { return s_thr_last[i]; }, o0, o1, stopped);
    if (exact) { cls = BS_EXACT_FULL; o0 = exact_j; }
    if (!scanning) cls = BS_NONE;

    // ---- results ----------------------------------------------------------------------------------------------
    if (valid) {
        if (invalid || cls == BS_NONE)
            store_result(a.out6, a.status, nullptr, 0, 0, r, invalid, false, 0, 0, 0, 0, 0, 0);
        else if (cls == BS_EXACT_FULL)
            store_result(a.out6, a.status, nullptr, 0, 0, r, false, true, 0, p.m, o0 - p.m, o0, p.m, 0);
        else if (cls == BS_EXACT_TAIL)
            store_result(a.out6, a.status, nullptr, 0, 0, r, false, true, 0, o0, n - o0, n, o0 - 2 * o1, o1);
        else if (cls == BS_SUBS_FULL)
            store_result(a.out6, a.status, nullptr, 0, 0, r, false, true, 0, p.m, o0 - p.m, o0, p.m - 2 * o1, o1);
        else
            store_result(a.out6, a.status, nullptr, 0, 0, r, false, false, 0, 0, 0, 0, 0, 0);   // the cell DP decides
    }
    const bool to_dp = valid && !invalid && cls == BS_DP;
    const bool to_back = to_dp && (o1 & 1), to_front = to_dp && !(o1 & 1);
    const unsigned long long bf = __ballot(to_front), bb = __ballot(to_back);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (to_front) {
        const int64_t slot = __popcll(bf & below);
        a.dp_queue[slot] = (int32_t)r; a.dp_win[2 * slot] = o0; a.dp_win[2 * slot + 1] = o1;
    } else if (to_back) {
        const int64_t slot = a.dp_cap - 1 - __popcll(bb & below);
        a.dp_queue[slot] = (int32_t)r; a.dp_win[2 * slot] = o0; a.dp_win[2 * slot + 1] = o1;
    }
    if (lane == 0) {
        // the three counters the cell-DP kernel looks at: written here, so that the path needs no memset at all
        *a.dp_count_front = (unsigned long long)__popcll(bf);
        *a.dp_count_back = (unsigned long long)__popcll(bb);
        *a.dp_work = 0ull;
        *a.need_dp = (int32_t)(__popcll(bf) + __popcll(bb));
    }
    // the host polls `done` (mapped host memory) instead of paying for a stream wait: everything above is visible
    // system-wide before the ticket is
    __threadfence_system();
    if (lane == 0) __hip_atomic_store(a.done, a.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// =============================================================================================
// k_comparer: PrefixComparer / SuffixComparer.locate -- Hamming distance over min(m, n)
// characters, again via the per-character bitset table (bit i = i-th compared position).
// =============================================================================================
__global__ __launch_bounds__(256) void k_comparer(DpArgs a) {
    __shared__ uint64_t s_rowmask[CAH_TABLE_CHARS];
    const CahMatcher* mt = a.matcher;
    for (int i = threadIdx.x; i < CAH_TABLE_CHARS; i += blockDim.x) s_rowmask[i] = mt->rowmask[i];
    __syncthreads();
    const int m = mt->m, max_k = mt->cmp_max_k, min_overlap = mt->min_overlap;
    const bool suffix = mt->kind == 2;
    const int lane = wave_lane();
    int64_t total = a.n_reads;
    if (a.queue_count) total = (int64_t)(*a.queue_count);
    for (;;) {
      const int64_t base0 = wave_dequeue(a.work_counter, CAH_DEQUEUE * WAVE);
      if (base0 >= total) break;
      for (int sub = 0; sub < CAH_DEQUEUE; ++sub) {
        const int64_t base = base0 + (int64_t)sub * WAVE;
        if (base >= total) break;
        const int64_t idx = base + lane;
        if (idx >= total) continue;
        const int64_t r = a.queue ? (int64_t)a.queue[idx] : idx;
        int64_t off, n64;
        read_extent(a.offsets, a.lens, a.uniform_first, a.uniform_len, r, off, n64);
        bool invalid = false;
        if (n64 > a.max_read_len) { invalid = true; n64 = 0; }
        const int n = (int)n64;
        const uint8_t* q = a.seqs + off;
        const int length = min(m, n);
        // compared region of the read: its first (prefix) or last (suffix) `length` characters,
        // fetched 16 per global load; compared position i is bit i of the row bitset
        const int qbeg = suffix ? n - length : 0;
        int errors = 0;
        unsigned seen = 0;
#pragma unroll 1
        for (int c0 = 0; c0 < CAH_MAX_M; c0 += 16) {
            if (!__any(c0 < length)) break;
            if (c0 < length) {
                const Chunk ck = load_chunk(q, qbeg + c0, n, qbeg + length);
                seen |= ck.w[0] | ck.w[1] | ck.w[2] | ck.w[3];
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int rel = c0 + t;                               // offset inside the region
                    const int i = suffix ? length - 1 - rel : rel;        // compared position
                    const unsigned ch = chunk_byte(ck, t) & (CAH_TABLE_CHARS - 1);
                    const uint64_t mk = s_rowmask[ch];
                    const bool inside = rel < length;
                    errors += (inside && !((mk >> (inside ? i : 0)) & 1ull)) ? 1 : 0;
                }
            }
        }
        if (seen & 0x80808080u) invalid = true;
        const bool found = !invalid && !(errors > max_k || length < min_overlap);   // :690-691
        const int score = length - 2 * errors;                                     // :692
        int32_t* o = a.out6 + r * 6;
        int t0, t1, t2, t3;
        if (!suffix) { t0 = 0; t1 = length; t2 = 0; t3 = length; }
        else { t0 = m - length; t1 = m; t2 = n - length; t3 = n; }                  // :714
        if (a.merge_best) {
            if (invalid) a.status[r] = 2;
            else if (found) {
                const bool had = a.status[r] == 1;
                if (a.status[r] != 2 && (!had || score > o[4] || (score == o[4] && errors < o[5]))) {
                    o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = score; o[5] = errors;
                    a.status[r] = 1;
                    if (a.best_adapter) a.best_adapter[r] = a.adapter_index;
                }
            }
        } else {
            a.status[r] = invalid ? (uint8_t)2 : (found ? (uint8_t)1 : (uint8_t)0);
            if (found) { o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = score; o[5] = errors; }
            else { o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0; }
        }
      }
    }
}

// =============================================================================================
// k_validate: count reads that hold a byte >= 0x80.
// =============================================================================================
__global__ __launch_bounds__(256) void k_validate(const uint8_t* seqs, const int64_t* offsets,
                                                  const int32_t* lens, int64_t n_reads, int32_t* bad) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += stride) {
        int64_t off, n;
        read_extent(offsets, lens, r, off, n);
        unsigned acc = 0;
        for (int64_t i = 0; i < n; ++i) acc |= seqs[off + i];
        if (acc & 0x80u) atomicAdd(bad, 1);
    }
}

__global__ void k_init_best(int32_t* best_adapter, int64_t n_reads) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += stride)
        best_adapter[r] = -1;
}

// =============================================================================================
// launchers (called from api.cpp)
// =============================================================================================
static int grid_for(int64_t n_items, int blocks_per_cu, int n_cus) {
    int64_t need = (n_items + 255) / 256;
    int64_t cap = (int64_t)blocks_per_cu * n_cus;
    if (need < 1) need = 1;
    return (int)(need < cap ? need : cap);
}

#define FILTER_MAX_LDS_TABLE_BYTES (32 * 1024)

hipError_t launch_filter(const FilterArgs& a, int mode, bool narrow, int n_cus, hipStream_t s) {
    const int grid = grid_for(a.n_reads, 4, n_cus);
    const size_t entry = narrow ? sizeof(uint32_t) : sizeof(uint64_t);
    const bool in_lds = (size_t)a.n_words * CAH_TABLE_CHARS * entry <= FILTER_MAX_LDS_TABLE_BYTES;
    const size_t lds = (in_lds ? (size_t)a.n_words * CAH_TABLE_CHARS * entry : 0) +
                       FILTER_TILE * 3 + CAH_QUEUE_BINS * 8 + 64 + (size_t)a.n_words * 24;
#define CAH_FILTER_LAUNCH(M, L, N) hipLaunchKernelGGL((k_filter<M, L, N>), dim3(grid), dim3(256), lds, s, a)
    if (mode == 0) {
        if (in_lds) { if (narrow) CAH_FILTER_LAUNCH(0, true, true); else CAH_FILTER_LAUNCH(0, true, false); }
        else { if (narrow) CAH_FILTER_LAUNCH(0, false, true); else CAH_FILTER_LAUNCH(0, false, false); }
    } else {
        if (in_lds) { if (narrow) CAH_FILTER_LAUNCH(1, true, true); else CAH_FILTER_LAUNCH(1, true, false); }
        else { if (narrow) CAH_FILTER_LAUNCH(1, false, true); else CAH_FILTER_LAUNCH(1, false, false); }
    }
#undef CAH_FILTER_LAUNCH
    return hipGetLastError();
}

static int getenv_flag(const char* name) {
    const char* e = getenv(name);
    return e && *e && *e != '0';
}

// streaming instances <row stride in 16-byte units (odd), units copied>: equally long reads of up to 112 / 160 characters
#define STREAM_NU_A 7
#define STREAM_UM_A 7
#define STREAM_NU_B 11
#define STREAM_UM_B 10

hipError_t launch_filter_lean(const FilterArgs& a_in, int mode, int n_lead, int n_gated, int delay, int tw_ok, int n_tw,
                              int n_cus, hipStream_t s) {
    FilterArgs a = a_in;
    if (mode != 0) a.present = nullptr;                   // the kernels tell the modes apart by `present`
    // equally long short reads of a plan with T-words: k_filter_stream2 (CAH_NO_STREAM2=1: the round-2 kernels)
    const bool stream2 = (a.batch_flag != nullptr || a.uniform_len > 0) && tw_ok && stream2_class_ok(n_lead, n_tw) &&
                         getenv_flag("CAH_NO_STREAM") == 0 && getenv_flag("CAH_NO_STREAM2") == 0;
    if (stream2) {
        hipError_t e = launch_filter_stream2(a, mode, n_lead, n_tw, n_cus, s);
        if (e != hipSuccess) return e;
        if (a.uniform_len > 0 && a.uniform_len <= (a.suffix_views ? stream2_max_len() : stream2_long_max_len()) &&
            a.n_reads * (int64_t)a.uniform_len >= 16)
            return hipSuccess;                            // the host knows the length: nothing else to launch
    }
    const int grid = grid_for(a.n_reads, LEAN_WAVES, n_cus);
    // dynamic part only (start gates of the ragged variant, tile staging, histogram); the character masks
    // are a static allocation of the kernel
    const size_t lds = (size_t)CAH_LEAN_MAX_GATED * CAH_GATE_LEN * sizeof(uint32_t) + (size_t)LEAN_TILE * 3 +
                       CAH_QUEUE_BINS * 8 + 64;
    // A packed batch gets every variant (all but one return at once, see batch_flag and the length ranges: no
    // host synchronisation): the streaming kernel for equally long short reads, the per-lane uniform kernel for
    // equally long longer reads, the ragged kernel for everything else.  Views (explicit lengths) have no batch
    // check and go to the ragged variant directly.
    const bool stream = !stream2 && (a.batch_flag != nullptr || a.uniform_len > 0) && getenv_flag("CAH_NO_STREAM") == 0;
#define CAH_STREAM_LAUNCH(DL, NL, NG, NU, UM, LO)                                                                   \
    do {                                                                                                            \
        const int tiles = (int)((a.n_reads + stream_tile(NL, NG) - 1) / stream_tile(NL, NG));                       \
        const int sgrid = std::max(1, std::min(tiles, n_cus));                                                      \
        const size_t slds = (size_t)stream_tile(NL, NG) * 3 + CAH_QUEUE_BINS * 8 + 64;                              \
        FilterArgs b = a;                                                                                           \
        b.stream_n_lo = (LO); b.stream_n_hi = stream_max_len(UM);                                                   \
        hipLaunchKernelGGL((k_filter_stream<DL, NL, NG, NU, UM>), dim3(sgrid), dim3(STREAM_BLOCK_WAVES * WAVE), slds, s, b); \
    } while (0)
#define CAH_LEAN_ALL(DL, NL, NG)                                                                                    \
    do {                                                                                                            \
        if (stream) {                                                                                               \
            CAH_STREAM_LAUNCH(DL, NL, NG, STREAM_NU_A, STREAM_UM_A, 0);                                             \
            CAH_STREAM_LAUNCH(DL, NL, NG, STREAM_NU_B, STREAM_UM_B, stream_max_len(STREAM_UM_A) + 1);               \
        }                                                                                                           \
        if (a.batch_flag || a.uniform_len > 0) hipLaunchKernelGGL((k_filter_lean<true, DL, NL, NG>), dim3(grid), dim3(256), lds, s, a);  \
        if (a.uniform_len <= 0) hipLaunchKernelGGL((k_filter_lean<false, DL, NL, NG>), dim3(grid), dim3(256), lds, s, a);    \
    } while (0)
#define CAH_LEAN_CLASS(NL, NG)                                                                                      \
    do {                                                                                                            \
        if (delay) CAH_LEAN_ALL(CAH_LEAN_DELAY, NL, NG); else CAH_LEAN_ALL(0, NL, NG);                              \
    } while (0)
    // the per-lane uniform kernel leaves the streaming kernels' lengths alone
    a.stream_n_lo = stream2 ? 1 : 0;
    a.stream_n_hi = stream2 ? (a.suffix_views ? stream2_max_len() : stream2_long_max_len()) : (stream ? stream_max_len(STREAM_UM_B) : -1);
    // slot classes <lead, gated>: TruSeq / e = 0.1 is <2, 3> as a 3' or 5' adapter, <2, 6> as an anywhere adapter
    if (n_lead <= 1 && n_gated <= 2) CAH_LEAN_CLASS(1, 2);
    else if (n_lead <= 2 && n_gated <= 3) CAH_LEAN_CLASS(2, 3);
    else if (n_lead <= 2) CAH_LEAN_CLASS(2, 6);
    else CAH_LEAN_CLASS(3, 6);
#undef CAH_LEAN_CLASS
#undef CAH_LEAN_ALL
#undef CAH_STREAM_LAUNCH
    return hipGetLastError();
}

hipError_t launch_uniform_check(const int64_t* offsets, int64_t n_reads, int64_t max_read_len, unsigned long long* flag,
                                int n_cus, hipStream_t s) {
    const int grid = grid_for(n_reads, 8, n_cus);
    hipLaunchKernelGGL(k_uniform_check, dim3(grid), dim3(256), 0, s, offsets, n_reads, max_read_len, flag);
    return hipGetLastError();
}

hipError_t launch_dp(const DpArgs& a, int m, bool unit, bool back_adapter, int64_t max_items, int n_cus,
                     hipStream_t s) {
    const int grid = grid_for(max_items, 8, n_cus);
    if (unit && back_adapter) {
        // the common case: 3' adapter, unit costs -> one-word-per-cell kernel
#define CAH_DPP_CASE(R)                                                                                            \
        if (m <= R) {                                                                                              \
            if (a.pairs) hipLaunchKernelGGL((k_dp_packed<R, true>), dim3(grid), dim3(256),                         \
                                            sizeof(uint64_t) * (size_t)a.n_adapters * CAH_MULTI_TAB_STRIDE, s, a); \
            else hipLaunchKernelGGL((k_dp_packed<R, false>), dim3(grid), dim3(256), sizeof(uint64_t) * CAH_TABLE_CHARS, s, a); \
            return hipGetLastError();                                                                              \
        }
        CAH_DPP_CASE(8) CAH_DPP_CASE(12) CAH_DPP_CASE(16) CAH_DPP_CASE(20) CAH_DPP_CASE(24) CAH_DPP_CASE(28)
        CAH_DPP_CASE(32) CAH_DPP_CASE(36) CAH_DPP_CASE(40) CAH_DPP_CASE(44) CAH_DPP_CASE(48) CAH_DPP_CASE(52)
        CAH_DPP_CASE(56) CAH_DPP_CASE(60) CAH_DPP_CASE(64)
#undef CAH_DPP_CASE
    }
    // UNIT: unit indel cost (the default); the general kernel carries D in an SGPR
#define CAH_DP_CASE(R)                                                                          \
    if (m <= R) {                                                                               \
        if (unit) hipLaunchKernelGGL((k_dp<R, true>), dim3(grid), dim3(256), 0, s, a);          \
        else hipLaunchKernelGGL((k_dp<R, false>), dim3(grid), dim3(256), 0, s, a);              \
        return hipGetLastError();                                                               \
    }
    CAH_DP_CASE(8)
    CAH_DP_CASE(12)
    CAH_DP_CASE(16)
    CAH_DP_CASE(20)
    CAH_DP_CASE(24)
    CAH_DP_CASE(28)
    CAH_DP_CASE(32)
    CAH_DP_CASE(36)
    CAH_DP_CASE(40)
    CAH_DP_CASE(44)
    CAH_DP_CASE(48)
    CAH_DP_CASE(52)
    CAH_DP_CASE(56)
    CAH_DP_CASE(60)
    CAH_DP_CASE(64)
#undef CAH_DP_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_back_scan(const ScanArgs& a0, int64_t max_items, int n_cus, hipStream_t s) {
    ScanArgs a = a0;
    if (a.tile < 256 || a.tile > SCAN_TILE || a.tile % 256) a.tile = SCAN_TILE;
    int64_t need = (max_items + a.tile - 1) / a.tile;
    if (need < 1) need = 1;
    const int64_t cap = (int64_t)8 * n_cus;
    const dim3 grid((unsigned)(need < cap ? need : cap));
    if (a.pairs) {
        const size_t lds = sizeof(uint64_t) * (size_t)a.n_adapters * CAH_MULTI_TAB_STRIDE;
        switch (a.kind) {
            case 1: hipLaunchKernelGGL((k_back_scan<true, 1>), grid, dim3(256), lds, s, a); break;
            case 2: hipLaunchKernelGGL((k_back_scan<true, 2>), grid, dim3(256), lds, s, a); break;
            case 3: hipLaunchKernelGGL((k_back_scan<true, 3>), grid, dim3(256), lds, s, a); break;
            default: hipLaunchKernelGGL((k_back_scan<true, 0>), grid, dim3(256), lds, s, a); break;
        }
    } else {
        const size_t lds = sizeof(uint64_t) * CAH_TABLE_CHARS;
        switch (a.kind) {
            case 1: hipLaunchKernelGGL((k_back_scan<false, 1>), grid, dim3(256), lds, s, a); break;
            case 2: hipLaunchKernelGGL((k_back_scan<false, 2>), grid, dim3(256), lds, s, a); break;
            case 3: hipLaunchKernelGGL((k_back_scan<false, 3>), grid, dim3(256), lds, s, a); break;
            default: hipLaunchKernelGGL((k_back_scan<false, 0>), grid, dim3(256), lds, s, a); break;
        }
    }
    return hipGetLastError();
}

hipError_t launch_tiny(const TinyArgs& a, int n_lead, int n_gated, int delay, hipStream_t s) {
#define CAH_TINY_CLASS(NL, NG)                                                                                      \
    do {                                                                                                            \
        if (delay) hipLaunchKernelGGL((k_tiny<CAH_LEAN_DELAY, NL, NG>), dim3(1), dim3(64), 0, s, a);                \
        else hipLaunchKernelGGL((k_tiny<0, NL, NG>), dim3(1), dim3(64), 0, s, a);                                   \
    } while (0)
    if (n_lead <= 1 && n_gated <= 2) CAH_TINY_CLASS(1, 2);
    else if (n_lead <= 2 && n_gated <= 3) CAH_TINY_CLASS(2, 3);
    else if (n_lead <= 2) CAH_TINY_CLASS(2, 6);
    else CAH_TINY_CLASS(3, 6);
#undef CAH_TINY_CLASS
    return hipGetLastError();
}

hipError_t launch_ticket(int32_t* done, int32_t ticket, hipStream_t s) {
    hipLaunchKernelGGL(k_ticket, dim3(1), dim3(1), 0, s, done, ticket);
    return hipGetLastError();
}

hipError_t launch_comparer(const DpArgs& a, int64_t max_items, int n_cus, hipStream_t s) {
    const int grid = grid_for(max_items, 8, n_cus);
    hipLaunchKernelGGL(k_comparer, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_anchored_exact(const DpArgs& a, int64_t max_items, int n_cus, hipStream_t s) {
    const int grid = grid_for(max_items, 8, n_cus);
    hipLaunchKernelGGL(k_anchored_exact, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_validate(const uint8_t* seqs, const int64_t* offsets, const int32_t* lens,
                           int64_t n_reads, int32_t* bad, int n_cus, hipStream_t s) {
    const int grid = grid_for(n_reads, 8, n_cus);
    hipLaunchKernelGGL(k_validate, dim3(grid), dim3(256), 0, s, seqs, offsets, lens, n_reads, bad);
    return hipGetLastError();
}

hipError_t launch_init_best(int32_t* best_adapter, int64_t n_reads, int n_cus, hipStream_t s) {
    const int grid = grid_for(n_reads, 8, n_cus);
    hipLaunchKernelGGL(k_init_best, dim3(grid), dim3(256), 0, s, best_adapter, n_reads);
    return hipGetLastError();
}

#ifdef CAH_DP_COUNT
extern "C" int cah_debug_dp_counters(unsigned long long* out8, int reset) {
    unsigned long long z[8] = {0};
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_dp_dbg), sizeof(z)) != hipSuccess) return 1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_dp_dbg), z, sizeof(z)) != hipSuccess) return 1;
    return 0;
}
#endif
