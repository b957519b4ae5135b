// multi2.hip -- the streaming form of the fused multi-adapter path for batches of equally long short reads (the
// sequencer's output; BASELINE config C4: 96 adapters): k_multi_stream (prefilter of ALL adapters in one pass) and
// k_multi_scan (cost scan of the (read, adapter) pairs it emits, page by page).  Rules and tables: multi2.h; the same
// rules are replayed on the CPU by tests/host_model/multi2_model.cpp against the oracle.
//
// Replaces k_multi_filter + k_back_scan<true> (multi.hip, kernels.hip) where the batch and the plan allow it
// (multi2_ok below); those stay for ragged batches, longer reads and plans with k-mers over ten characters.
// Reference: MultipleAdapters.match_to (src/cutadapt/adapters.py:1265-1286) = per adapter KmerFinder.kmers_present
// (_kmer_finder.pyx:170-257) then Aligner.locate (_align.pyx:298-587).
//
// k_multi_stream, per workgroup of 16 waves (one per CU):
//   * the reads are copied HBM -> LDS with coalesced 16-byte loads, half a read at a time (k_filter_stream2's copy
//     plan: every cache line crosses the memory system once), one read per lane;
//   * per character: 2-bit code (LDS byte table), a rolling word of the last sixteen characters, ONE probe of an exact
//     bitmap for the whole-read k-mers (one AND of that word); a probe hit becomes an EVENT (lane, position, the word) in
//     the wave's LDS ring;
//   * events are resolved 64 at a time by ALL lanes (directory -> entries in LDS, verified on every character and on
//     the k-mer's windows) -- no lane waits for another lane's hit;
//   * behind the main pass: the tail classes' hits (probed in the read's last chunks, one 32-bit mask per (class, index
//     class) pass, exact class-specific bitmaps) become events pass by pass in class order (hi, lo, E0), each class resolved
//     before the next; error-free overlaps of up to four characters come from a table of first adapters, without events;
//   * a pair is emitted once (an LDS bitset `seen` per read and adapter; which pairs saw a FURTHER hit is two words per read)
//     into a PAGE of its class: pages of
//     1024 pairs from a device-wide pool, owned by one wave, one class of pairs per page -- the scan's waves then hold 64
//     pairs of one window shape.  Pairs that only the error-free rows can match are decided here (suffix compare).
//   * a whole-read pair whose first hit is ONE chunk of the adapter's k + 1 carries that occurrence (position, chunk):
//     if it stays the pair's only hit -- a byte per read tells -- the scan needs the ~40 columns around it only
//     (CAH_M2_PAIR_PRECISE, multi2.h);
//   * tiles of 1024 reads are drawn from one counter of the batch, and only while the page pool holds what the tiles in
//     flight could ask for in the worst case: a launch that stops early is followed by another ROUND over the rest
//     (api.cpp: match_batch_multi) -- a typical batch takes one round whatever the worst case would need.
// k_multi_scan orders a whole-read page's pairs by their windows before it scans them, runs full-window waves with the
// substitution / one-indel bookkeeping (most adapters with sequencing errors finish there, not in the cell DP) and tail
// pages as the bare recurrence (lo pages: the adapter's first 32 rows in one 32-bit word).
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "cah_device.h"
#include "kernels.h"
#include "dev_common.h"
#include "back_scan.h"
#include "multi2.h"

#ifndef M2_WAVES
#define M2_WAVES 16                // waves per block = per CU (developer builds: -DM2_WAVES=12 gives the kernel 168 VGPRs)
#endif
#define M2_TILE 1024               // reads per block tile (one piece per wave: the blocks end within one piece of each other)
#define M2_HALF 5                  // 16-byte units per half-row
#define M2_ROW (M2_HALF * 16)
#define M2_MAX_LEN (2 * M2_HALF * 16)
#define M2_RING 128                // events per wave ring (two rounds)
#define M2_TILE_RING 16           // the block's last tiles: {its own count, the batch's tile} (see take_tile)
#define M2_TILEMAP_SPINS (1u << 24)     // polls (with s_sleep) a wave waits for its tile entry before it gives up: seconds, not forever
#define M2_NO_TILE 0xFFFFFFFFu

typedef unsigned int m2_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int m2_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool m2_any(const bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ void* m2_uniform_ptr(const void* q) {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned m2_rank(const unsigned long long mask) {      // set bits below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// dynamic LDS layout (bytes), computed alike on host and device
struct M2Layout {
    unsigned bitmap, dir, entries, prefix, xlat, slot, seen, wide, ring, rlast, pages, misc, total;
};
// (what has a fixed size comes first: those offsets are compile-time constants and fold into the LDS instructions)
__host__ __device__ inline M2Layout m2_layout(const int n_entries, const int n_adapters) {
    M2Layout L;
    unsigned o = 0;
    const unsigned words = (unsigned)(n_adapters + 31) / 32;
    L.xlat = o; o += 256;                                               // (3-bit codes, then the same at two bits: 'A' for anything else)
    L.prefix = o; o += 128 * 4;
    L.bitmap = o; o += CAH_M2_BM_WORDS * 4;
    L.dir = o; o += CAH_M2_SLOTS * 2;
    L.slot = o; o += M2_WAVES * WAVE * M2_ROW;
    L.ring = o; o += M2_WAVES * M2_RING * 8;
    L.rlast = o; o += M2_WAVES * WAVE * 4;
    L.pages = o; o += M2_WAVES * CAH_M2_PAIR_CLASSES * 8;
    L.misc = o; o += 64 + M2_TILE_RING * 8;
    L.entries = o; o += (unsigned)((n_entries + 2) & ~1) * 8;            // (+ one slot: the walk reads entries in twos)
    L.seen = o; o += M2_WAVES * WAVE * words * 4;
    L.wide = o; o += M2_WAVES * WAVE * 8;                               // {smallest, largest} adapter + 1 whose pair saw a further hit
    L.total = o;
    return L;
}

__device__ __forceinline__ unsigned multi_tab_index2(unsigned c) { return (c & 0x40u) ? (c & 31u) : 0u; }   // (kernels.hip: multi_tab_index)

// class of a pair's page: 0 lo, 1 hi, 2.. whole-read pairs by the length of their window (16-column chunks)
__host__ __device__ inline int m2_pair_class_w(const int chunks) { return chunks <= 4 ? 2 : (chunks <= 6 ? 3 : (chunks <= 8 ? 4 : 5)); }

// W8: every whole-read k-mer has eight or more characters (one index class: the plans kmer_heuristic builds for
// adapters of 24+ characters at rate 0.1) -- the main pass then makes ONE probe per character, straight-line
#ifdef M2_TRACE
// developer build only (-DM2_TRACE): s_memtime stamps of one wave at the stations of its first pieces
#define M2_TRACE_PIECES 64
#define M2_TRACE_STATIONS 16
__device__ unsigned long long g_m2_trace[M2_TRACE_PIECES * M2_TRACE_STATIONS];
extern "C" int cah_debug_m2_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_m2_trace), sizeof(g_m2_trace)) == hipSuccess ? 0 : 1;
}
#define M2_STAMP(st) do { if (blockIdx.x == 7 && wave == 5 && trace_it < M2_TRACE_PIECES) { \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_m2_trace[trace_it * M2_TRACE_STATIONS + (st)] = t_; } } while (0)
#define M2_COUNT(st, v) do { if (blockIdx.x == 7 && wave == 5 && trace_it < M2_TRACE_PIECES && lane == 0) g_m2_trace[trace_it * M2_TRACE_STATIONS + (st)] += (v); } while (0)
#define M2_TIC() const unsigned long long tic_ = __builtin_amdgcn_s_memtime()
#define M2_TOC(st) M2_COUNT(st, __builtin_amdgcn_s_memtime() - tic_)
#else
#define M2_TIC() do { } while (0)
#define M2_TOC(st) do { } while (0)
#define M2_STAMP(st) do { } while (0)
#define M2_COUNT(st, v) do { } while (0)
#endif

// RV (round 6): the reads are VIEWS inside the reads of a uniform batch (a.view_starts / a.view_lens: what a pipeline holds
// behind the quality trimmers) -- streamed END-ALIGNED like k_filter_stream2's RV form: unit u of read r is fetched d[r] =
// (read end - view end) bytes further down, the characters in front of a view are NUL (they break every k-mer, and every
// window of the tail classes counts from the END: multi2.h), every position the kernel reports is one of this frame of n
// characters.  k_multi_scan works on the same frame and reports in the view's coordinates.
#define M2_RV_BACK 1184            // the copy resource starts this far in front of a piece (>= M2_MAX_LEN + 63 * 15 + 1)
template <bool W8, bool RV = false>
__global__ __launch_bounds__(M2_WAVES * WAVE) void k_multi_stream(Multi2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const CahMulti2Header* const hd = a.hdr;
    const int n_adapters = hd->n_adapters;
    const int n_entries = (int)hd->n_entries;
    const M2Layout LY = m2_layout(n_entries, n_adapters);
    uint32_t* const s_bm = reinterpret_cast<uint32_t*>(s_raw + LY.bitmap);
    uint16_t* const s_dir = reinterpret_cast<uint16_t*>(s_raw + LY.dir);
    CahM2Slot* const s_ent = reinterpret_cast<CahM2Slot*>(s_raw + LY.entries);
    uint32_t* const s_prefix = reinterpret_cast<uint32_t*>(s_raw + LY.prefix);
    uint8_t* const s_xlat = s_raw + LY.xlat;
    const int words = (n_adapters + 31) / 32;                           // bitset words per read

    const int n = a.uniform_len;
    const int n_reads = (int)a.n_reads;                                 // reads of this launch (< 2^31)
    const int64_t first_byte = a.uniform_first + a.first_read * (int64_t)n;
    const int64_t total = (int64_t)n_reads * n;
    if (*a.tile_counter >= (unsigned long long)a.n_tiles) return;       // (a round that finds no tile left)
    for (int i = threadIdx.x; i < CAH_M2_BM_WORDS; i += blockDim.x) s_bm[i] = a.bitmap[i];
    for (int i = threadIdx.x; i < CAH_M2_SLOTS / 2; i += blockDim.x)
        reinterpret_cast<uint32_t*>(s_dir)[i] = reinterpret_cast<const uint32_t*>(a.dir)[i];
    for (int i = threadIdx.x; i < n_entries; i += blockDim.x) s_ent[i] = a.entries[i];
    for (int i = threadIdx.x; i < 128; i += blockDim.x) {
        s_prefix[i] = i < n_adapters ? a.prefix[i] : 0u;
        s_xlat[i] = (uint8_t)m2_code((unsigned)i);
        s_xlat[128 + i] = (uint8_t)(m2_code((unsigned)i) & 3u);
    }
    const int lane = wave_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned char* const slot = s_raw + LY.slot + wave * (WAVE * M2_ROW);
    const unsigned char* const row = slot + lane * M2_ROW;
    uint32_t* const s_seen = reinterpret_cast<uint32_t*>(s_raw + LY.seen) + wave * WAVE * words;
    uint32_t* const s_wide = reinterpret_cast<uint32_t*>(s_raw + LY.wide) + wave * WAVE * 2;   // per read: {min, max} of (adapter + 1) over the pairs with a further hit
    m2_u32x2* const s_ring = reinterpret_cast<m2_u32x2*>(s_raw + LY.ring) + wave * M2_RING;
    uint32_t* const s_rlast = reinterpret_cast<uint32_t*>(s_raw + LY.rlast) + wave * WAVE;
    uint32_t* const s_wmin = s_rlast;                                   // (see resolve_round: until class W is resolved)
    uint32_t* const s_pg = reinterpret_cast<uint32_t*>(s_raw + LY.pages) + wave * CAH_M2_PAIR_CLASSES * 2;   // {page, fill} per class
    unsigned* const s_next_piece = reinterpret_cast<unsigned*>(s_raw + LY.misc);
    volatile unsigned long long* const s_tilemap = reinterpret_cast<volatile unsigned long long*>(s_raw + LY.misc + 64);
    if (lane < CAH_M2_PAIR_CLASSES) { s_pg[2 * lane] = 0xFFFFFFFFu; s_pg[2 * lane + 1] = 0u; }
    // ---- tiles are handed out by ONE counter of the batch (blocks that drew cheap tiles take more of them), and only
    // while the page pool still holds what every tile in flight could ask for in the worst case (a.gate_pages, api.cpp):
    // the launch then ends early and the next round of the same batch goes on from the counter.  A block knows its
    // tiles by its own count kt; entry kt & 15 of s_tilemap is {kt, batch tile} -- the tile of kt + 1 is drawn by the wave
    // that takes the first piece of kt, a whole tile ahead of its first use.
    auto draw_tile = [&]() -> unsigned {
        const unsigned long long used = __hip_atomic_load(a.page_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((long long)used > a.gate_pages) return M2_NO_TILE;
        const unsigned long long t = atomicAdd(a.tile_counter, 1ull);
        return t < (unsigned long long)a.n_tiles ? (unsigned)t : M2_NO_TILE;
    };
    // (LDS keeps what the last block on this CU left: an entry of ITS map must not pass for one of ours)
    if (threadIdx.x < M2_TILE_RING) s_tilemap[threadIdx.x] = ~0ull;
    // the passes of the tail classes (multi2.h: tq_*), one packed word each: class | index class << 2 | first position << 8 |
    // last position << 16 (first > last: no position of the pass lies in a read of n characters) | mask register << 24 | its
    // upper half << 27
    uint32_t* const s_pass = reinterpret_cast<uint32_t*>(s_raw + LY.misc + 16);
    if (threadIdx.x < CAH_M2_MAX_PASSES) {
        const int j = (int)threadIdx.x;
        uint32_t pw = (1u << 8);                                        // (first 1, last 0: an empty pass)
        if (j < hd->tq_n) {
            const int qc = hd->tq_qc[j], qx = qc < 8 ? qc : CAH_M2_MAXQ;
            const int plo = max(0, n + qc - 1 - hd->tq_open[j]), phi = min(n - 1, n + qx - 1 - hd->tq_close[j]);
            if (phi >= plo) pw = (uint32_t)hd->tq_cls[j] | ((uint32_t)qc << 2) | ((uint32_t)plo << 8) | ((uint32_t)phi << 16) |
                                 ((uint32_t)hd->tq_slot[j] << 24) | (hd->tq_shift[j] ? 1u << 27 : 0u);
        }
        s_pass[j] = pw;
    }
    if (threadIdx.x == 0) {
        *s_next_piece = M2_WAVES;
        const unsigned t0 = draw_tile();
        // (a batch of no more tiles than blocks: one tile each, none drawn ahead)
        const unsigned t1 = (t0 == M2_NO_TILE || a.n_tiles <= (int64_t)gridDim.x) ? M2_NO_TILE : draw_tile();
        s_tilemap[0] = (unsigned long long)t0;
        s_tilemap[1] = (1ull << 32) | t1;
    }
    __syncthreads();

    // plan constants (wave-uniform)
    const int m = hd->m, k = hd->k, min_overlap = hd->min_overlap, lmax0 = hd->lmax0;
    const int qmask_w = hd->q_mask[M2_W];
    // the tail classes (multi2.h): one hit mask per index class, probed in the read's last chunks; behind the main pass one
    // event pass per (class, index class) in class order.  Their constants are read from the header where they are used
    // (scalar loads; the kernel has no SGPR to keep 44 of them in)
    const int tq_n = hd->tq_n;
    const int qm_fixed = hd->qm_fixed;
    constexpr bool w_only8 = W8;
    const unsigned lane16 = (unsigned)lane * 16u;

    // ---- copy plan (k_filter_stream2's): H1 units of every read in the first half-row, H2 in the second
    // (a read of at most M2_HALF units is ONE half-row: its tail chunks, whose words wait in the row, then never straddle
    // the two fills of the slot -- reads of 16 .. 80 characters)
    const int U = (n + 15) >> 4;
    const int H1 = U <= M2_HALF ? U : (U + 1) >> 1, H2 = U - H1;
    const unsigned magic1 = (65536u + (unsigned)H1 - 1u) / (unsigned)H1;
    const unsigned magic2 = H2 ? (65536u + (unsigned)H2 - 1u) / (unsigned)H2 : 0u;
    auto unit_r = [&](int kk, unsigned magic) -> int {
        unsigned ln = (unsigned)lane;
        asm volatile("" : "+v"(ln));
        return (int)(__umul24((unsigned)(kk * WAVE) + ln, magic) >> 16);
    };
    constexpr int PPT = M2_TILE / WAVE;
    constexpr int64_t NO_PIECE = (int64_t)1 << 40;                      // "the batch has no more tiles (for this round)"
    // first read of piece p (the block's count): the tile of p / PPT from the map; the first piece of a tile draws the next
    auto piece_base = [&](unsigned p) -> int64_t {
        const unsigned kt = p / PPT;
        unsigned long long e;
        // (the entry is written a whole tile ahead of its first use: the wait is short -- and bounded: a wave that waits
        // in vain says so (err bit 1, match_batch_multi returns CAH_EINTERNAL) instead of hanging the device)
        unsigned spins = 0;
        unsigned long long t_first = 0ull;
        for (;;) {
            e = s_tilemap[kt & (M2_TILE_RING - 1)];
            if ((unsigned)(e >> 32) == kt) break;
            // (bounded in polls AND in time -- 2^31 ticks of s_memtime, about a second at the shader clock, twenty at the
            // 100 MHz reference: a device shared with other work, a debugger, a preempted producer wave must not turn a
            // slow batch into a failed one -- advisor, round 5; the host tries a batch that gave up here once more)
            if (spins == 0) t_first = __builtin_amdgcn_s_memtime();
            if (++spins > M2_TILEMAP_SPINS && __builtin_amdgcn_s_memtime() - t_first > (1ull << 31)) {
                if (lane == 0) atomicOr(a.err, 2ull);
                return NO_PIECE;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        const unsigned t = __builtin_amdgcn_readfirstlane((unsigned)e);
        if (p % PPT == 0 && kt >= 1) {
            unsigned nt = M2_NO_TILE;
            if (t != M2_NO_TILE) { if (lane == 0) nt = draw_tile(); nt = __builtin_amdgcn_readfirstlane(nt); }
            if (lane == 0) s_tilemap[(kt + 1) & (M2_TILE_RING - 1)] = ((unsigned long long)(kt + 1) << 32) | nt;
        }
        if (t == M2_NO_TILE) return NO_PIECE;
        return (int64_t)t * M2_TILE + (int64_t)(p % PPT) * WAVE;
    };
    auto take_piece = [&]() -> unsigned {
        unsigned p = 0;
        if (lane == 0) p = __hip_atomic_fetch_add(s_next_piece, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(p);
    };
    m2_u32x4 pre[2 * M2_HALF];
    const uint8_t* const batch0 = a.seqs + first_byte;
    // RV: the view of read base + lane inside its read as d | skip << 16 -- d: characters between the view's end and the
    // read's, skip = n - length: where the view starts in the end-aligned frame (clamped to the read: batch.py checks)
    auto view_of = [&](const int64_t base) -> uint32_t {
        if (base + lane >= n_reads) return 0u;
        const int64_t r = a.first_read + base + lane;
        if (a.view_general) {                                           // (views anywhere: d is not used, the copy gathers)
            int ln = a.view_lens[r];
            ln = ln < 0 ? 0 : (ln > n ? n : ln);
            return (uint32_t)(n - ln) << 16;
        }
        int st = (int)(a.view_starts[r] - (a.uniform_first + r * (int64_t)n));
        st = st < 0 ? 0 : (st > n ? n : st);
        int ln = a.view_lens[r];
        ln = ln < 0 ? 0 : (ln > n - st ? n - st : ln);
        return (uint32_t)(n - (st + ln)) | ((uint32_t)(n - ln) << 16);
    };
    auto prefetch = [&](int64_t base, const uint32_t vw = 0u) {
        const int64_t left = n_reads - base;
        if (left <= 0) return;
        if constexpr (RV) {
            if (a.view_general) {
                // Views ANYWHERE in the buffer (a packed batch, the reads of a raw FASTQ chunk): unit u of the piece --
                // frame characters 16 c .. 16 c + 15 of read r -- is gathered from its view's END: byte (view end) - n + 16 c.
                // The end (relative to lane 0's, 32 bits) and the NULs in front of the view travel from lane r by two lane
                // exchanges per unit.  Nothing outside a view is touched: a unit that reaches in front of its view or behind
                // it is loaded as the 16 bytes at the view's edge and shifted (what lies outside is masked by finish()
                // anyway); views shorter than 16 characters byte by byte.
                const int reads = (int)(left < WAVE ? left : (int64_t)WAVE);
                const int64_t r_own = a.first_read + base + (lane < reads ? lane : 0);
                const int sk_own = (int)(vw >> 16);
                const int64_t ve_own = a.view_starts[r_own] + (int64_t)(n - sk_own);
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
#pragma unroll
                for (int q = 0; q < 2 * M2_HALF; ++q) {
                    const int kk = q < M2_HALF ? q : q - M2_HALF;
                    const int H = q < M2_HALF ? H1 : H2;
                    m2_u32x4 got = (m2_u32x4)(0u);
                    if (kk < H) {                                       // (wave-uniform: every lane takes part in the exchanges)
                        const int rr = unit_r(kk, q < M2_HALF ? magic1 : magic2);
                        const unsigned ve_lo_r = (unsigned)__builtin_amdgcn_ds_bpermute(rr << 2, ve_lo_own);
                        const unsigned ve_hi_r = (unsigned)__builtin_amdgcn_ds_bpermute(rr << 2, ve_hi_own);
                        const int64_t veb_r = (int64_t)(((unsigned long long)ve_hi_r << 32) | ve_lo_r);
                        const int sk_r = __builtin_amdgcn_ds_bpermute(rr << 2, sk_own);
                        const int u = kk * WAVE + lane;
                        const int fc = 16 * (u - __mul24(rr, H)) + (q < M2_HALF ? 0 : 16 * H1);       // the unit's first frame character
                        const int len = n - sk_r;
                        if (u < reads * H && fc + 15 >= sk_r && fc < n && len > 0) {
                            unsigned x0, x1, x2, x3;
                            gather_frame_unit(a.seqs, veb_r, len, n, fc, safe_lo, safe_hi, x0, x1, x2, x3);
                            got = (m2_u32x4){x0, x1, x2, x3};
                        }
                    }
                    pre[q] = got;
                }
                return;
            }
        }
        const int64_t pbyte = base * (int64_t)n;
        const uint8_t* const src = batch0 + pbyte;
        if (left >= WAVE && pbyte + (int64_t)WAVE * n + 16 <= total && (!RV || pbyte >= M2_RV_BACK)) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(m2_uniform_ptr(RV ? src - M2_RV_BACK : src), 0, 0x7FFFFFFF, 0x00020000);
            if constexpr (RV) {
                // (k_filter_stream2's RV copy plan: what a unit needs of its read r is computed once per read, in lane r, for
                // either half's shape and handed over by one lane exchange per unit: the low half A = r (n - 16 H) + BACK -
                // d[r] -- unit 64 k + lane starts at byte A + 16 lane + 1024 k of the resource --, the high half T = skip[r]
                // + 16 r H: the unit lies in front of its view iff its last character is < T, and is then not fetched: an
                // offset out of the resource's range returns zeros)
                const int d = (int)(vw & 0xFFFFu), sk = (int)(vw >> 16);
                const unsigned w1 = (unsigned)(__mul24(lane, n - 16 * H1) + M2_RV_BACK - d) | ((unsigned)(sk + 16 * __mul24(lane, H1)) << 16);
                const unsigned w2 = (unsigned)(__mul24(lane, n - 16 * H2) + M2_RV_BACK - d) | ((unsigned)(sk + 16 * __mul24(lane, H2)) << 16);
                int got[2 * M2_HALF];
#pragma unroll
                for (int kk = 0; kk < M2_HALF; ++kk) {
                    got[kk] = __builtin_amdgcn_ds_bpermute(unit_r(kk, magic1) << 2, (int)w1);
                    got[M2_HALF + kk] = __builtin_amdgcn_ds_bpermute(unit_r(kk, magic2) << 2, (int)w2);
                }
#pragma unroll
                for (int kk = 0; kk < M2_HALF; ++kk)
                    if (kk < H1) {
                        const int w = got[kk];
                        const unsigned off = (int)lane16 + (16 * kk * WAVE + 15) < (w >> 16) ? 0x80000000u : (unsigned)((w & 0xFFFF) + (int)lane16);
                        pre[kk] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + kk * (WAVE * 16), 0, 0);
                    }
#pragma unroll
                for (int kk = 0; kk < M2_HALF; ++kk)
                    if (kk < H2) {
                        const int w = got[M2_HALF + kk];
                        const unsigned off = (int)lane16 + (16 * kk * WAVE + 15) + 16 * H1 < (w >> 16) ? 0x80000000u : (unsigned)((w & 0xFFFF) + (int)lane16);
                        pre[M2_HALF + kk] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + kk * (WAVE * 16), 16 * H1, 0);
                    }
                return;
            }
#pragma unroll
            for (int kk = 0; kk < M2_HALF; ++kk)
                if (kk < H1)
                    pre[kk] = __builtin_amdgcn_raw_buffer_load_b128(
                        rs, (unsigned)(__mul24(unit_r(kk, magic1), n - 16 * H1) + (int)lane16) + kk * (WAVE * 16), 0, 0);
#pragma unroll
            for (int kk = 0; kk < M2_HALF; ++kk)
                if (kk < H2)
                    pre[M2_HALF + kk] = __builtin_amdgcn_raw_buffer_load_b128(
                        rs, (unsigned)(__mul24(unit_r(kk, magic2), n - 16 * H2) + (int)lane16) + kk * (WAVE * 16), 16 * H1, 0);
            return;
        }
        const int reads = (int)(left < WAVE ? left : (int64_t)WAVE);
#pragma unroll
        for (int q = 0; q < 2 * M2_HALF; ++q) {
            const int kk = q < M2_HALF ? q : q - M2_HALF;
            const int H = q < M2_HALF ? H1 : H2;
            m2_u32x4 got = (m2_u32x4)(0u);
            if constexpr (RV) {
                // byte by byte: a shifted unit may begin in front of the batch or end behind it
                if (kk < H) {                                           // (wave-uniform: every lane takes part in the exchange)
                    const int r = unit_r(kk, q < M2_HALF ? magic1 : magic2);
                    const int w = __builtin_amdgcn_ds_bpermute(r << 2, (int)vw);
                    const int sh = w & 0xFFFF;
                    const int last = 16 * (kk * WAVE + lane - __mul24(r, H)) + (q < M2_HALF ? 15 : 16 * H1 + 15);
                    if (kk * WAVE + lane < reads * H && last >= (w >> 16)) {       // (a unit in front of the view: zeros)
                        const int64_t g0 = pbyte + (int64_t)(__mul24(r, n - 16 * H) + (kk * WAVE) * 16 + (int)lane16 +
                                                             (q < M2_HALF ? 0 : 16 * H1)) - sh;
                        unsigned x[4] = {0u, 0u, 0u, 0u};
#pragma unroll 1
                        for (int b = 0; b < 16; ++b) {
                            const int64_t at = g0 + b;
                            if (at >= 0 && at < total) x[b >> 2] |= (unsigned)batch0[at] << (8 * (b & 3));
                        }
                        got = (m2_u32x4){x[0], x[1], x[2], x[3]};
                    }
                }
                pre[q] = got;
                continue;
            }
            if (kk < H && kk * WAVE + lane < reads * H) {
                const int r = unit_r(kk, q < M2_HALF ? magic1 : magic2);
                const unsigned goff = (unsigned)(__mul24(r, n - 16 * H) + (kk * WAVE) * 16 + (int)lane16 + (q < M2_HALF ? 0 : 16 * H1));
                if (pbyte + goff + 16 <= total) {
                    Unaligned16 v;
                    __builtin_memcpy(&v, src + goff, 16);
                    got = (m2_u32x4){v.w[0], v.w[1], v.w[2], v.w[3]};
                } else if (pbyte + goff < total) {
                    Unaligned16 v;
                    __builtin_memcpy(&v, batch0 + (total - 16), 16);
                    const int sft = (int)(pbyte + goff + 16 - total);
                    const int dw = sft >> 2, sh = (sft & 3) * 8;
                    unsigned x0 = v.w[0], x1 = v.w[1], x2 = v.w[2], x3 = v.w[3];
                    if (dw >= 2) { x0 = x2; x1 = x3; x2 = 0; x3 = 0; }
                    if (dw & 1) { x0 = x1; x1 = x2; x2 = x3; x3 = 0; }
                    got = (m2_u32x4){(unsigned)((((unsigned long long)x1 << 32) | x0) >> sh),
                                     (unsigned)((((unsigned long long)x2 << 32) | x1) >> sh),
                                     (unsigned)((((unsigned long long)x3 << 32) | x2) >> sh), x3 >> sh};
                }
            }
            pre[q] = got;
        }
    };
    auto to_slot = [&](auto half_c) {
        constexpr int half = decltype(half_c)::value;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int H = half ? H2 : H1;
        const unsigned magic = half ? magic2 : magic1;
#pragma unroll
        for (int kk = 0; kk < M2_HALF; ++kk)
            if (kk < H) {
                unsigned off = lane16;
                if (H != M2_HALF) off += (unsigned)__mul24(unit_r(kk, magic), M2_ROW - 16 * H);
                *reinterpret_cast<m2_u32x4*>(slot + kk * (WAVE * 16) + off) = pre[half * M2_HALF + kk];
            }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // the chunk at `pos`: characters past the read's end become NUL
    int skip = 0;                                                       // RV: this lane's view starts `skip` characters into the frame
    auto finish = [&](m2_u32x4 v, int pos) -> m2_u32x4 {
        if constexpr (RV) {
            if (m2_any(skip > pos)) {                                   // (the unit a view starts in holds its read's characters in front of it)
                unsigned x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int drop = skip - pos - 4 * i;                // characters of dword i in front of the view
                    x[i] = drop <= 0 ? x[i] : (drop >= 4 ? 0u : (x[i] & (0xFFFFFFFFu << (8 * drop))));
                }
                v = (m2_u32x4){x[0], x[1], x[2], x[3]};
            }
        }
        if (pos + 16 > n) {
            unsigned x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int keep = n - pos - 4 * i;
                x[i] &= keep >= 4 ? 0xFFFFFFFFu : (keep <= 0 ? 0u : ((1u << (8 * keep)) - 1u));
            }
            v = (m2_u32x4){x[0], x[1], x[2], x[3]};
        }
        return v;
    };

#ifdef M2_TRACE
    int trace_it = 0;
#endif
    // ---- events: ring[tail % RING] = {rolling word, lane | position << 8 | qc << 16}
    unsigned ring_head = 0, ring_count = 0;                             // wave-uniform
    int64_t piece_first = 0;                                            // read index (within the launch) of lane 0's read
    int cur_cls = M2_W;                                                 // the class being probed (events in the ring are of it)

    // pairs of class pc from the lanes of `mask` (wave-uniform, non-empty) go to the wave's open page of that class
    auto append_pairs = [&](const unsigned long long mask, const int pc, const bool mine, const uint64_t pair) {
        const unsigned cnt = (unsigned)__popcll(mask);
        unsigned page = __builtin_amdgcn_readfirstlane(s_pg[2 * pc]);
        unsigned fill = __builtin_amdgcn_readfirstlane(s_pg[2 * pc + 1]);
        if (page == 0xFFFFFFFFu || fill + cnt > CAH_M2_PAGE) {
            unsigned np = 0;
            if (lane == 0) {
                if (page != 0xFFFFFFFFu && (int64_t)page < a.max_pages) a.page_hdr[page] = ((unsigned)pc << 24) | fill;
                np = (unsigned)atomicAdd(a.page_counter, 1ull);
                // The gate (api.cpp) keeps the pool from running out; should its arithmetic ever be wrong the pairs of this
                // page are lost -- never silently: the flag makes match_batch_multi fail with CAH_EINTERNAL.
                if ((int64_t)np >= a.max_pages) atomicOr(a.err, 1ull);
            }
            page = __builtin_amdgcn_readfirstlane(np);
            fill = 0;
        }
        if (mine && (int64_t)page < a.max_pages) a.pairs[(int64_t)page * CAH_M2_PAGE + fill + m2_rank(mask)] = pair;
        if (lane == 0) { s_pg[2 * pc] = page; s_pg[2 * pc + 1] = fill + cnt; }
    };

    // resolve up to 64 events of class cur_cls from the head of the ring.  The pairs a round emits are staged in the ring
    // slots it has just consumed (one per lane at a time) and leave for their pages once per round, class by class.
    auto resolve_round = [&]() {
        const unsigned cnt = ring_count < 64u ? ring_count : 64u;
        M2_COUNT(10, 1ull + ((unsigned long long)cnt << 32));
        M2_TIC();
        const bool have = (unsigned)lane < cnt;
        const unsigned stage0 = ring_head;                              // the consumed slots: [stage0, stage0 + 64) mod RING
        m2_u32x2 ev = (m2_u32x2)(0u);
        if (have) ev = s_ring[(ring_head + (unsigned)lane) & (M2_RING - 1)];
        ring_head = (ring_head + cnt) & (M2_RING - 1);
        ring_count -= cnt;
        const uint32_t r = ev.x;
        const int lr = (int)(ev.y & 63u), p = (int)((ev.y >> 8) & 255u), qc = (int)((ev.y >> 16) & 15u);
        const int p_head = __builtin_amdgcn_readfirstlane(p);
        const uint32_t d = have ? (uint32_t)s_dir[m2_index(r, qc) & (CAH_M2_SLOTS - 1)] : 0u;
        int u = m2_dir_begin(d);
        int left = m2_dir_count(d);
        const int cls = cur_cls;
        uint32_t* const sw = s_wide + lr * 2;
        uint32_t* const ss = s_seen + lr * words;
        const uint32_t rd = (uint32_t)(a.first_read + piece_first + lr);
        // class and key of the pairs of this round that are not "whole read" pairs
        int pc_cls, key_cls;
        unsigned flags_cls = 0;
        if (cls == M2_W) {
            // every event of this round ends in or behind the chunk of the round's first event (events are pushed
            // chunk by chunk), and an earlier event of the pair would have emitted it in an earlier round
            key_cls = (p_head & ~15) >> CAH_KEY_SHIFT;
            key_cls = key_cls < CAH_QUEUE_BINS - 1 ? key_cls : CAH_QUEUE_BINS - 1;
            pc_cls = m2_pair_class_w((n - max(0, (key_cls << CAH_KEY_SHIFT) - m - k - 1) + 15) >> 4);
        } else if (cls == M2_HI) { pc_cls = 1; key_cls = max(0, n - a.win_hi) >> 2; flags_cls = CAH_M2_PAIR_TAIL; }
        else if (cls == M2_LO) { pc_cls = 0; key_cls = max(0, n - a.win_lo) >> 2; flags_cls = CAH_M2_PAIR_TAIL; }
        else { pc_cls = 7; key_cls = 0; }                              // class E0: the suffix compare decides
        const unsigned lo_cls = ((unsigned)pc_cls << 28) | (flags_cls << 24) | (unsigned)key_cls;
        unsigned staged = 0;                                            // wave-uniform
        // the staged pairs (at most 64, one per lane) leave: pairs for their class's page, suffix compares to best_key
        auto flush = [&]() {
            const bool mine = (unsigned)lane < staged;
            m2_u32x2 st = (m2_u32x2)(0u);
            if (mine) st = s_ring[(stage0 + (unsigned)lane) & (M2_RING - 1)];
            const int pc = (int)((st.x >> 28) & 7u);
            const int lr2 = (int)(st.y - (uint32_t)(a.first_read + piece_first));
            // (RV: a result is reported in the VIEW's coordinates: the frame's minus the NULs in front of the view -- the
            // staged pair's read is another lane's)
            int nv2 = n;
            if constexpr (RV) nv2 = n - __builtin_amdgcn_ds_bpermute(lr2 << 2, skip);
            if (mine && pc == 7) {
                const int adapter = (int)((st.x >> 8) & 255u);
                const int i = m2_exact_tail(s_rlast[lr2], s_prefix[adapter], min_overlap, lmax0, n);
                if (i > 0) atomicMax(a.best_key + st.y, pack_best(i, 0, adapter, i, nv2 - i, nv2));
            }
            unsigned long long em = __ballot(mine && pc != 7);
            const uint64_t pair = ((uint64_t)st.y << 32) | (uint64_t)(st.x & 0x0FFFFFFFu);
            while (em) {
                const int src = __ffsll((long long)em) - 1;
                const int pc0 = __builtin_amdgcn_readlane(pc, src);
                const unsigned long long mk = __ballot(mine && pc == pc0);
                append_pairs(mk, pc0, mine && pc == pc0, pair);
                em &= ~mk;
            }
            staged = 0;
        };
        M2_TOC(12);
        // An event's home holds the entries of every k-mer that shares its low index bits: the lane walks them once with
        // the comparisons alone (class, index class, all characters, window -- multi2.h) for its FIRST entry that counts
        // and the number of further ones; the bitsets, the staging and the ballots below run once per entry that counts
        // (one for most events, none for a false positive), not once per entry of the home
        auto entry_ok = [&](const CahM2Slot& e) -> bool {
            const int q = m2_q(e.meta);
            return m2_cls(e.meta) == cls && (q < 8 ? q : 8) == qc && (r & m2_mask2(q)) == e.key && p - q + 1 >= 0 &&
                   m2_in_window(e.meta, n - (p - q + 1));
        };
        // (two entries per step -- one ds_read2_b64: the walk is a chain of dependent LDS reads, its length the longest home
        // among the round's 64 events; the slot behind the last entry exists: m2_layout)
        int uf = -1, more = 0;
        while (m2_any(left > 0)) {
            M2_COUNT(11, 1);
            const bool on0 = left > 0, on1 = left > 1;
            const int ua = on0 ? u : 0;
            const CahM2Slot e0 = s_ent[ua], e1 = s_ent[ua + 1];
            const bool ok0 = on0 && entry_ok(e0), ok1 = on1 && entry_ok(e1);
            more += (ok0 && uf >= 0) ? 1 : 0;
            uf = (ok0 && uf < 0) ? u : uf;
            more += (ok1 && uf >= 0) ? 1 : 0;
            uf = (ok1 && uf < 0) ? u + 1 : uf;
            u += 2; left -= 2;
        }
        u = m2_dir_begin(d) + m2_dir_count(d);                          // (an odd home: the walk looked one slot too far)
        {
            // (a home's count saturates at CAH_M2_MAX_GROUP -- adapters that share their k-mers: such a home is walked while
            // the entries are its own)
            bool sat = have && m2_dir_count(d) == CAH_M2_MAX_GROUP;
            if (m2_any(sat)) {
                const uint32_t home = m2_index(r, qc) & (CAH_M2_SLOTS - 1);
                while (m2_any(sat)) {
                    const bool in = sat && u < n_entries;
                    const CahM2Slot e = s_ent[in ? u : 0];
                    sat = in && m2_home_of(e.key, e.meta) == home;
                    const bool ok = sat && entry_ok(e);
                    more += (ok && uf >= 0) ? 1 : 0;
                    uf = (ok && uf < 0) ? u : uf;
                    ++u;
                }
            }
        }
        M2_TOC(15);
        while (m2_any(uf >= 0)) {
            const bool is_ref = uf >= 0;
            const CahM2Slot e = s_ent[is_ref ? uf : 0];
            const int adapter = m2_adapter(e.meta);
            const unsigned bit = 1u << (adapter & 31);
            const int word = adapter >> 5;
            // (the bitset is touched by every lane: a zero changes nothing, and no lane branches)
            const unsigned old = atomicOr(ss + word, is_ref ? bit : 0u);
            const bool emit = is_ref && (old & bit) == 0;
            const bool again = is_ref && (old & bit) != 0;              // a further hit of a pair that exists
            const unsigned long long em = __ballot(emit);
            if (em) {
                const unsigned c = (unsigned)__popcll(em);
                if (staged + c > 64u) flush();
                if (emit) {
                    // (a whole-read pair whose occurrence is ONE chunk of the adapter's k + 1 -- CAH_M2_PAIR_PRECISE, multi2.h --
                    // carries the occurrence's position and chunk index instead of the round's key and goes to the pages of
                    // the short windows: k_multi_scan orders a page's pairs by their windows)
#if defined(M2_ABL) && (M2_ABL & 8)
                    const unsigned we = 0u;                             // developer build: no pair takes the window of its one occurrence
#else
                    const unsigned we = cls == M2_W ? m2_precise_chunk(e.meta) : 0u;
#endif
                    unsigned lo = lo_cls;
                    if (we) lo = (2u << 28) | ((CAH_M2_PAIR_PRECISE | ((we - 1u) << CAH_M2_PAIR_CHUNK_SHIFT)) << 24) | (unsigned)p;
                    lo |= (unsigned)adapter << 8;
                    s_ring[(stage0 + staged + m2_rank(em)) & (M2_RING - 1)] = (m2_u32x2){lo, rd};
                }
                staged += c;
            }
            // (behind the emission: a pair emitted by this very instruction reads its "wide" bit before a further hit of
            // the same round sets it -- seen & wide at the end of the piece = the pair saw more than its first hit)
#if !(defined(M2_ABL) && (M2_ABL & 16))
            if (m2_any(again)) {
                // (which pairs of the read saw a further hit: the smallest and the largest adapter say "none", "this one" or
                // "several" -- all k_multi_scan asks)
                if (again) { atomicMin(sw, (unsigned)adapter + 1u); atomicMax(sw + 1, (unsigned)adapter + 1u); }
                // ... and where: the lane that emitted the pair need not hold its EARLIEST occurrence of the round (the lanes
                // race for the `seen` bit), so the full window such a pair falls back to starts at the earliest of its own
                // position and the read's further whole-read hits (s_wmin: the read's slot of s_rlast, free until class W is
                // resolved)
                if (cls == M2_W && again) atomicMin(s_wmin + lr, (unsigned)p);
            }
#endif
            // the lane's next entry that counts (adapters that share the k-mer: rare)
            bool need = is_ref && more > 0;
            more -= need ? 1 : 0;
            int un = uf + 1;
            uf = need ? uf : -1;
            while (m2_any(need)) {
                const CahM2Slot e2 = s_ent[need ? un : 0];
                const bool ok2 = need && entry_ok(e2);
                uf = ok2 ? un : uf;
                need = need && !ok2;
                ++un;
            }
        }
        M2_TOC(13);
        if (staged) flush();
        M2_TOC(14);
    };
    // the lanes of `mask` push one event each; the ring always has room for 64
    auto push_events = [&](const unsigned long long mask, const bool mine, const uint32_t r, const int p, const int qc) {
#if defined(M2_ABL) && (M2_ABL & 2)
        if (ring_count > M2_RING - 64) ring_count = 0;
#endif
        if (ring_count > M2_RING - 64) resolve_round();
        const unsigned at = (ring_head + ring_count + m2_rank(mask)) & (M2_RING - 1);
        if (mine) s_ring[at] = (m2_u32x2){r, (unsigned)lane | ((unsigned)p << 8) | ((unsigned)qc << 16)};
        ring_count += (unsigned)__popcll(mask);
    };
    auto drain = [&]() {
#if defined(M2_ABL) && (M2_ABL & 2)
        ring_count = 0;                                                 // developer build: events are dropped (timing only)
#endif
        while (ring_count) resolve_round();
    };
    // one bitmap probe: does a k-mer of index class qc end with the word r?
    // (class W and index class 8: the hashed bitmap; a tail class's k-mer of fewer characters: its exact one -- multi2.h)
    // one bitmap probe of class W: does a k-mer of index class qc end with the word r2 (two bits per character, multi2.h)?
    auto probe = [&](const uint32_t r2, const int qc) -> bool {
        const uint32_t idx = qc >= 8 ? (r2 & 0xFFFFu) : m2_index(r2, qc);
        return ((s_bm[idx >> 5] >> (idx & 31)) & 1u) != 0;
    };

    // the first position a pass is probed at (a k-mer of index class qc that ends at p starts n - p + q - 1 characters before
    // the end, q = qc -- or up to CAH_M2_MAXQ for index class 8), the first chunk that holds one (tail_base), and where that
    // chunk sits in the slot's row (the launcher checked: behind the first half-row, at most four chunks to the read's end)
    int tail_p0 = n;
    for (int j = 0; j < tq_n; ++j) tail_p0 = min(tail_p0, max(0, n + hd->tq_qc[j] - 1 - hd->tq_open[j]));
    // (the passes' words: lane j holds pass j's for the whole kernel -- a v_readlane where an LDS round trip per pass and
    // tail chunk was)
    const uint32_t pass_words = (unsigned)lane < (unsigned)CAH_M2_MAX_PASSES ? s_pass[lane] : 0u;
    const int tail_base = tail_p0 & ~15;
    const int tail_off = H2 > 0 ? 16 * H1 : 0;                          // first position of the last half-row
    const int tail_unit0 = (tail_base - tail_off) >> 4;                 // row unit of the first tail chunk

    int64_t base_cur = piece_base((unsigned)wave);
    uint32_t vw_cur = 0u;
    if constexpr (RV) vw_cur = base_cur < NO_PIECE ? view_of(base_cur) : 0u;
    prefetch(base_cur, vw_cur);
#pragma unroll 1
    for (;;) {
        if (base_cur >= NO_PIECE) break;
        const int base = (int)base_cur;
        const bool more = (unsigned)base < (unsigned)n_reads;
        const bool valid = more && (unsigned)(base + lane) < (unsigned)n_reads;
        piece_first = base;
        if constexpr (RV) skip = (int)(vw_cur >> 16);
        M2_STAMP(0);
        if (more) {
            // ---- per-read state
            for (int w = 0; w < words; ++w) s_seen[lane * words + w] = 0;
            s_wide[2 * lane] = 0xFFFFFFFFu; s_wide[2 * lane + 1] = 0u;
            s_wmin[lane] = 255u;
            // the rolling word the tables are looked up with: two bits per character, sixteen characters (multi2.h: m2_roll2 --
            // anything but A / C / G / T reads as 'A' there)
            uint32_t r2 = 0u;
            uint32_t r2_prev = 0u;                                      // ... at the end of the chunk before
            // the read's last ten characters at THREE bits each (what the error-free overlaps are compared with) and its last
            // sixteen at two (the tables of first adapters): rolled in the last two chunks only
            uint32_t r3 = 0x24924924u, rlast = 0x24924924u, rlast2 = 0u;
            unsigned seen_chars = 0;
            uint32_t tm[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};          // per pass: bit b = a hit at the pass's first position + b
            uint32_t tw0[2] = {0u, 0u};                                 // the two words of the first tail chunk (before it, at its end)
            cur_cls = M2_W;
            m2_u32x4 cur = (m2_u32x4)(0u);
#pragma unroll 1
            for (int ph = 0; ph < 2; ++ph) {
                const int H = ph ? H2 : H1;
                if (H == 0) continue;
                if (ph == 0) to_slot(std::integral_constant<int, 0>{}); else to_slot(std::integral_constant<int, 1>{});
                M2_STAMP(1 + 2 * ph);
                const int pos0 = ph ? 16 * H1 : 0;
                cur = finish(*reinterpret_cast<const m2_u32x4*>(row), pos0);
#pragma unroll 1
                for (int c = 0; c < H; ++c) {
                    const int pos = pos0 + 16 * c;
                    m2_u32x4 nxt = (m2_u32x4)(0u);
                    if (c + 1 < H) nxt = finish(*reinterpret_cast<const m2_u32x4*>(row + 16 * (c + 1)), pos + 16);
                    const unsigned w4[4] = {cur.x, cur.y, cur.z, cur.w};
                    seen_chars |= cur.x | cur.y | cur.z | cur.w;
                    // translate the chunk's characters (16 independent LDS reads), roll, probe
                    uint32_t e[16];
#pragma unroll
                    for (int t = 0; t < 16; ++t) e[t] = s_xlat[128u + ((w4[t >> 2] >> (8 * (t & 3))) & 127u)];
                    uint32_t rr2[16];
#pragma unroll
                    for (int t = 0; t < 16; ++t) { r2 = (r2 << 2) | e[t]; rr2[t] = r2; }
                    // the read's last characters at three bits each (n is the batch's: the chunk and the last position in it are
                    // wave-uniform): translated once more with the 3-bit table, in the read's last two chunks only
                    if (pos + 32 > n && pos < n) {
                        const int t_last = n - 1 - pos;                 // >= 16 in the chunk before the last
#pragma unroll
                        for (int t = 0; t < 16; ++t)
                            r3 = t <= t_last ? ((r3 << 3) | (uint32_t)s_xlat[(w4[t >> 2] >> (8 * (t & 3))) & 127u]) : r3;
                    }
                    unsigned hits = 0;
                    if constexpr (w_only8) {
                        uint32_t wd[16];
#pragma unroll
                        for (int t = 0; t < 16; ++t) wd[t] = s_bm[(rr2[t] & 0xFFFFu) >> 5];   // m2_bit(r, 8, .): the last eight characters
#pragma unroll
                        for (int t = 0; t < 16; ++t) hits |= ((wd[t] >> (rr2[t] & 31u)) & 1u) << t;
                    } else {
#pragma unroll
                        for (int t = 0; t < 16; ++t) {
                            bool h = false;
                            for (int qc = 1; qc <= 8; ++qc)
                                if ((qmask_w >> qc) & 1) h = h || probe(rr2[t], qc);
                            hits |= (h ? 1u : 0u) << t;
                        }
                    }
                    if (pos + 16 > n) hits &= (1u << (n - pos)) - 1u;   // positions past the read's end
                    if (!valid) hits = 0;
                    if constexpr (RV && w_only8) {
                        // (the NULs in front of a view read as 'A' in the 2-bit word: a k-mer that would begin among them is none)
                        const int lim = skip + 7 - pos;
                        if (lim > 0) hits &= lim >= 16 ? 0u : ~((1u << lim) - 1u);
                    }
                    // the word at position pos + t: the sixteen characters up to it, from the words at the chunk's end and the
                    // chunk before's (one v_alignbit)
                    const uint32_t r2_end = r2;
                    auto word_at = [&](const int t) -> uint32_t {
                        return __builtin_amdgcn_alignbit(r2_prev, r2_end, (unsigned)(2 * (15 - t)));
                    };
                    if (pos < n && pos + 16 >= n) { rlast = r3; rlast2 = word_at(n - 1 - pos); }   // (the read's last chunk)
                    // ---- the tail slots: probed in the read's last chunks with the words the main pass has anyway; the hits
                    // wait (a bit mask per slot) until class W is resolved.  The chunk's three words wait with them: the
                    // first tail chunk's in registers, a later one's in the row unit in front of it (its characters are spent)
#if !(defined(M2_ABL) && (M2_ABL & 1))
                    if (pos + 16 > tail_p0 && pos < n) {
                        const int kch = (pos - tail_base) >> 4;
                        if (kch == 0) { tw0[0] = r2_prev; tw0[1] = r2_end; }
                        else {
                            uint32_t* const keep = reinterpret_cast<uint32_t*>(const_cast<unsigned char*>(row) + 16 * (c - 1));
                            keep[0] = r2_prev; keep[1] = r2_end;
                        }
#pragma unroll 1
                        for (int j = 0; j < tq_n; ++j) {
                            // (the loop stays rolled: ONE copy of the probes; the pass's mask by a chain of scalar branches)
                            const uint32_t pw = __builtin_amdgcn_readlane(pass_words, j);
                            const int pcls = (int)(pw & 3u), qc = (int)((pw >> 2) & 15u);
                            const int plo = (int)((pw >> 8) & 255u), phi = (int)((pw >> 16) & 255u);
                            if (phi < plo || pos + 16 <= plo || pos > phi) continue;       // wave-uniform
                            unsigned h16 = 0;
                            if (qc >= 8) {
                                if constexpr (w_only8) {
                                    h16 = hits;                                            // the main pass's own probe
                                } else {
                                    uint32_t wd2[16];
#pragma unroll
                                    for (int t = 0; t < 16; ++t) wd2[t] = s_bm[(rr2[t] & 0xFFFFu) >> 5];
#pragma unroll
                                    for (int t = 0; t < 16; ++t) h16 |= ((wd2[t] >> (rr2[t] & 31u)) & 1u) << t;
                                }
                            } else {
                                // the exact bitmap of the class and index class: the k-mer at two bits per character
                                const uint32_t msk2 = (1u << (2 * qc)) - 1u;
                                const uint32_t region = CAH_M2_BM8_WORDS * 32u + m2_tail_region(qc, pcls);   // (a multiple of 32)
                                // (four positions at a time, and only those the pass's window [plo, phi] reaches: a window
                                // of 8 to 20 positions lies in two or three chunks and would cost 32 to 48 probes otherwise)
#pragma unroll
                                for (int t0 = 0; t0 < 16; t0 += 4) {
                                    if (pos + t0 + 4 <= plo || pos + t0 > phi) continue;   // wave-uniform
                                    uint32_t wd2[4];
#pragma unroll
                                    for (int t = 0; t < 4; ++t) wd2[t] = s_bm[(region + (rr2[t0 + t] & msk2)) >> 5];
#pragma unroll
                                    for (int t = 0; t < 4; ++t) h16 |= ((wd2[t] >> (rr2[t0 + t] & msk2 & 31u)) & 1u) << (t0 + t);
                                }
                            }
                            if (phi - pos < 15) h16 &= (2u << (phi - pos)) - 1u;          // positions behind the pass's last
                            const int lo_t = plo - pos;
                            if (lo_t > 0) h16 &= ~((1u << lo_t) - 1u);
                            if (pos + 16 > n) h16 &= (1u << (n - pos)) - 1u;
                            if (!valid) h16 = 0;
                            if constexpr (RV) {
                                const int lim = skip + qc - 1 - pos;      // (a k-mer that would begin in front of the view)
                                if (lim > 0) h16 &= lim >= 16 ? 0u : ~((1u << lim) - 1u);
                            }
                            uint32_t add = lo_t > 0 ? h16 >> lo_t : h16 << (pos - plo);   // (bit b = position plo + b)
                            if ((pw >> 27) & 1u) add <<= 16;                               // (a narrow pass in its register's upper half)
                            const int sl = (int)((pw >> 24) & 7u);
                            if (sl == 0) tm[0] |= add; else if (sl == 1) tm[1] |= add; else if (sl == 2) tm[2] |= add;
                            else if (sl == 3) tm[3] |= add; else if (sl == 4) tm[4] |= add; else if (sl == 5) tm[5] |= add;
                            else if (sl == 6) tm[6] |= add; else tm[7] |= add;
                        }
                    }
#endif
#if defined(M2_ABL) && (M2_ABL & 4)
                    hits = 0;                                           // developer build: no events from the main pass
#endif
                    if (m2_any(hits != 0)) {
                        while (m2_any(hits != 0)) {
                            const bool mine = hits != 0;
                            const int t = mine ? (int)__builtin_ctz(hits) : 0;
                            hits &= hits - 1u;
                            const uint32_t rt = word_at(t);
                            const unsigned long long mk = __ballot(mine);
                            if constexpr (w_only8) {
                                push_events(mk, mine, rt, pos + t, 8);
                            } else {
                                // (several index classes: one event per class that hits)
                                for (int qc = 1; qc <= 8; ++qc) {
                                    if (!((qmask_w >> qc) & 1)) continue;
                                    const bool hq = mine && probe(rt, qc);
                                    const unsigned long long mq = __ballot(hq);
                                    if (mq) push_events(mq, hq, rt, pos + t, qc);
                                }
                            }
                        }
                    }
                    r2_prev = r2_end;
                    cur = nxt;
                }
                M2_STAMP(2 + 2 * ph);
            }
            drain();                                                    // class W is resolved
            const unsigned w_again_chunk = s_wmin[lane] >> 4;           // chunk of the read's earliest further whole-read hit (15: none)
            s_rlast[lane] = rlast;
            M2_STAMP(5);
            // ---- the tail masks' hits become events, pass by pass in class order (hi, lo, E0), each class resolved before
            // the next: a pass takes the hits of its index class's mask whose positions fit ITS window
            {
                int prev_cls = M2_W;
#pragma unroll 1
                for (int j = 0; j < tq_n; ++j) {
                    const uint32_t pw = __builtin_amdgcn_readlane(pass_words, j);
                    const int pcls = (int)(pw & 3u), qc = (int)((pw >> 2) & 15u);
                    const int plo = (int)((pw >> 8) & 255u), phi = (int)((pw >> 16) & 255u);
                    if (phi < plo) continue;                                               // wave-uniform: no position of the pass is in the read
                    if (pcls != prev_cls) {
                        drain();
                        M2_STAMP(4 + pcls);
                        prev_cls = pcls;
                        cur_cls = pcls;
                    }
                    const int sl = (int)((pw >> 24) & 7u);
                    uint32_t mk = sl == 0 ? tm[0] : (sl == 1 ? tm[1] : (sl == 2 ? tm[2] : (sl == 3 ? tm[3] : (sl == 4 ? tm[4] : (sl == 5 ? tm[5] :
                                  (sl == 6 ? tm[6] : tm[7]))))));
                    // (a pass of at most 16 positions has half a register: the other half is another pass's)
                    if (phi - plo < 16) mk = ((pw >> 27) & 1u) ? mk >> 16 : mk & 0xFFFFu;
                    while (m2_any(mk != 0u)) {
                        const bool mine = mk != 0u;
                        const int bit = (mine ? (int)__builtin_ctz(mk) : 0) + plo - tail_base;   // bit b of the tail = position tail_base + b
                        mk &= mk - 1u;
                        const int kch = bit >> 4, t = bit & 15;
                        // the chunk's two words: in front of it and at its end
                        const int unit = max(tail_unit0 + kch - 1, 0);
                        const uint32_t* const keep = reinterpret_cast<const uint32_t*>(row + 16 * unit);
                        uint32_t a0 = keep[0], a1 = keep[1];
                        if (kch == 0) { a0 = tw0[0]; a1 = tw0[1]; }
                        const uint32_t rt = __builtin_amdgcn_alignbit(a0, a1, (unsigned)(2 * (15 - t)));
                        push_events(__ballot(mine), mine, rt, tail_base + bit, qc);
                    }
                }
                drain();
                M2_STAMP(8);
            }
            if (qm_fixed) {
                // error-free overlaps of q <= 4 characters: "the read's last q characters are adapter[0:q]" -- every lane looks
                // its own read's end up in the table of first adapters (multi2.h: CAH_M2_FIXED_WORD): no events, no walk
                const uint8_t* const s_fixed = reinterpret_cast<const uint8_t*>(s_bm + CAH_M2_FIXED_WORD);
                const uint32_t* const ss = s_seen + lane * words;
                unsigned long long bestk = 0;
#pragma unroll
                for (int q = 1; q <= CAH_M2_FIXED_MAXQ; ++q) {
                    if (!((qm_fixed >> q) & 1)) continue;                              // wave-uniform
                    const bool plain = (rlast & m2_mask(q) & 0x24924924u) == 0u;      // the last q characters are A, C, G, T
                    unsigned adapter = (valid && plain && q <= n - skip) ? s_fixed[m2_fixed_off(q) + (rlast2 & m2_mask2(q))] : 0xFFu;
                    // (an adapter whose pair exists is the scan's business: the next one that begins with these characters)
                    while (m2_any(adapter != 0xFFu && ((ss[(adapter & 127u) >> 5] >> (adapter & 31u)) & 1u) != 0u)) {
                        const bool taken = adapter != 0xFFu && ((ss[(adapter & 127u) >> 5] >> (adapter & 31u)) & 1u) != 0u;
                        if (taken) adapter = s_fixed[m2_fixed_next(q) + adapter];
                    }
                    if (adapter != 0xFFu) {
                        const unsigned long long kk = pack_best(q, 0, (int)adapter, q, n - skip - q, n - skip);
                        bestk = kk > bestk ? kk : bestk;
                    }
                }
                if (bestk) atomicMax(a.best_key + (a.first_read + base + lane), bestk);
            }
            // the read's flagged adapter (every class is resolved): what k_multi_scan asks before it trusts the window of
            // a CAH_M2_PAIR_PRECISE pair
#if !(defined(M2_ABL) && (M2_ABL & 32))
            if (valid) {
                const unsigned alo = s_wide[2 * lane], ahi = s_wide[2 * lane + 1];
                a.wmeta[a.first_read + base + lane] =
                    (uint16_t)((ahi == 0u ? CAH_M2_NO_FLAG : (alo == ahi ? ahi - 1u : CAH_M2_MANY_FLAGS)) | (w_again_chunk << 8));
            }
#endif
            if (valid && (seen_chars & 0x80808080u) != 0) a.status[a.first_read + base + lane] = 2;
        }
        base_cur = piece_base(take_piece());
        if constexpr (RV) vw_cur = base_cur < NO_PIECE ? view_of(base_cur) : 0u;
        prefetch(base_cur, vw_cur);
        M2_STAMP(9);
#ifdef M2_TRACE
        if (blockIdx.x == 7 && wave == 5) ++trace_it;
#endif
    }
    // close the wave's open pages
    if (lane < CAH_M2_PAIR_CLASSES) {
        const unsigned page = s_pg[2 * lane], fill = s_pg[2 * lane + 1];
        if (page != 0xFFFFFFFFu && (int64_t)page < a.max_pages) a.page_hdr[page] = ((unsigned)lane << 24) | fill;
    }
}

// =============================================================================================
// k_multi_scan: the cost scan (back_scan.h) of the pairs k_multi_stream left, page by page.  A page holds pairs of ONE
// class.  "lo" pages: ~24 columns and the rows an overlap of the first error class can reach; "hi" pages: the reach of
// the widest tail class -- every pair of such a page has the same window.  Whole-read pages: a pair's window runs from
// its first k-mer hit - m - k - 1 to the read's end, or -- a CAH_M2_PAIR_PRECISE pair whose occurrence stayed its only
// hit (multi2.h) -- around that one occurrence; the block first orders the page's pairs by (first chunk, last chunk) of
// their windows (a counting sort in LDS), so that the 64 lanes of a wave walk nearly the same columns: the wave starts at
// its earliest window's first chunk and ends behind its latest window's last (more columns than a lane's window needs are
// as exact).  Matches are merged with one atomicMax on the read's best key (kernels.h: pack_best); pairs that need cells
// go to k_dp_packed<ROWS, true>'s work list with a window that is safe for the cell DP (tail pairs: the full reach
// m + k + 1, see multi2_model.cpp).
// =============================================================================================
template <int KIND>
__global__ __launch_bounds__(256, (KIND == 3 || KIND == 0) ? 4 : 5) void k_multi_scan(Multi2ScanArgs a) {
    constexpr int XR = KIND >= 2 ? KIND - 1 : 0;
    extern __shared__ __attribute__((aligned(16))) uint64_t s_scanmask[];   // [n_adapters * CAH_MULTI_TAB_STRIDE]
    __shared__ int s_thr_last[CAH_MAX_M + 1];
    __shared__ uint32_t s_dp[CAH_M2_PAGE];          // the page's DP work list: pair in page | first column << 10 | (last * 2 + scan) << 18
    __shared__ uint32_t s_ord[CAH_M2_PAGE];         // whole-read pages: the pairs in window order (see phase A)
    __shared__ unsigned s_bins[256];
    __shared__ uint32_t s_prefix[128];
    __shared__ uint8_t s_xlat[128];
    __shared__ unsigned s_nf, s_nb;
    __shared__ long long s_page;
    __shared__ unsigned long long s_gf, s_gb;
    const CahMatcher* mt = a.matcher;
    if (*a.page_counter == 0ull) return;                                // (a round that found no tile left)
    // (pages are drawn from one counter: blocks beyond their number have nothing to do -- they leave before the tables are
    // copied, which is what a launch over a small batch would otherwise spend its time on)
    if ((unsigned long long)blockIdx.x >= *a.page_counter) return;
    for (int i = threadIdx.x; i < a.n_adapters * CAH_MULTI_TAB_STRIDE; i += blockDim.x)
        s_scanmask[i] = KIND == 0 ? a.tab[i] : bs32_table_entry(a.tab[i], mt->m);
    for (int i = threadIdx.x; i <= CAH_MAX_M; i += blockDim.x) s_thr_last[i] = mt->thr_last[i];
    for (int i = threadIdx.x; i < 128; i += blockDim.x) {
        s_prefix[i] = i < a.n_adapters ? a.prefix[i] : 0u;
        s_xlat[i] = (uint8_t)m2_code((unsigned)i);
    }
    BackScanParams p;
    p.m = mt->m; p.k = mt->k; p.kacc = mt->kacc; p.min_overlap = mt->min_overlap; p.half_m = mt->m / 2;
    const int reach = p.m + p.k + 1;
    const int chunk_base = p.m / (p.k + 1), chunk_extra = p.m % (p.k + 1);
    const int lane = wave_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = a.uniform_len;
    // the read of a pair: its first character and, for a view inside its read (k_multi_stream's RV form), the NULs in front
    // of it in the end-aligned frame of n characters the scan works on (clamped to the read exactly as view_of does there)
    auto frame_of = [&](const int64_t r, int& pad) -> const uint8_t* {
        const int64_t at = a.uniform_first + r * (int64_t)n;
        pad = 0;
        if (!a.view_starts) return a.seqs + at;
        if (a.view_general) {                                           // views anywhere in the buffer: no read around them
            int ln = a.view_lens[r];
            ln = ln < 0 ? 0 : (ln > n ? n : ln);
            pad = n - ln;
            return a.seqs + a.view_starts[r];
        }
        int st = (int)(a.view_starts[r] - at);
        st = st < 0 ? 0 : (st > n ? n : st);
        int ln = a.view_lens[r];
        ln = ln < 0 ? 0 : (ln > n - st ? n - st : ln);
        pad = n - ln;
        return a.seqs + at + st;
    };
    int64_t n_pages = (int64_t)(*a.page_counter);
    if (n_pages > a.max_pages) {                                        // (k_multi_stream has flagged it; said again here)
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.err, 1ull);
        n_pages = a.max_pages;
    }

    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            s_page = (long long)atomicAdd(a.work_counter, 1ull);
            s_nf = 0; s_nb = 0;
        }
        s_bins[threadIdx.x] = 0u;                                       // (256 threads)
        __syncthreads();
        const int64_t page = s_page;
        if (page >= n_pages) break;
        const uint32_t hdr = a.page_hdr[page];
        const int count = (int)(hdr & 0xFFFFFFu);
        const bool tail_page = (hdr >> 24) < 2u;                          // classes lo, hi: every pair has the same window
        if (!tail_page) {
            // ---- phase A: every pair's window in 16-column chunks counted from the read's END (cs: chunks from the
            // window's first column to the end, ce: whole chunks behind its last), the suffix compare of the pairs that
            // stop early, and the pairs in the order of (cs, ce)
            uint32_t desc[CAH_M2_PAGE / 256];
            unsigned slot[CAH_M2_PAGE / 256];
#pragma unroll
            for (int t = 0; t < CAH_M2_PAGE / 256; ++t) {
                const int e0 = (int)threadIdx.x + 256 * t;
                desc[t] = 0; slot[t] = 0;
                if (e0 < count) {
                    const uint64_t pr = a.pairs[page * CAH_M2_PAGE + e0];
                    const unsigned flags = (unsigned)(pr >> 24) & 0xFFu, key = (unsigned)pr & 0xFFu;
                    const unsigned adapter = (unsigned)(pr >> 8) & 0xFFFFu;
                    const int64_t r = (int64_t)(pr >> 32);
                    int j0w = max(0, ((int)key << CAH_KEY_SHIFT) - p.m - p.k - 1), jb = n;
                    unsigned precise = 0, tail0 = 0;
                    if (flags & CAH_M2_PAIR_PRECISE) {
                        const unsigned wm16 = a.wmeta[r], wm = wm16 & 0xFFu;
                        if (wm != adapter && wm != CAH_M2_MANY_FLAGS) {
                            precise = 1;
                            m2_precise_window((int)key, (int)(flags >> CAH_M2_PAIR_CHUNK_SHIFT) & 3, p.m, p.k, chunk_base, chunk_extra, n, j0w, jb);
                            // the read's last ten characters against the adapter's first: the error-free overlaps
                            int pad_a = 0;
                            const uint8_t* q = frame_of(r, pad_a);
                            const Chunk tl = load_chunk_frame(q, n - 16, n, pad_a, n);
                            uint32_t rlast = 0;
#pragma unroll
                            for (int c = 6; c < 16; ++c) rlast = (rlast << 3) | (uint32_t)s_xlat[chunk_byte(tl, c) & 127u];
                            tail0 = (unsigned)m2_exact_tail(rlast, s_prefix[adapter], p.min_overlap, a.lmax0, n);
                        } else {
                            // (flagged: the window of a whole-read pair, from the earliest of the pair's hits)
                            j0w = max(0, (int)(min(key >> 4, wm16 >> 8) << 4) - p.m - p.k - 1);
                        }
                    }
                    j0w = min(j0w, n);
                    const unsigned cs = (unsigned)((n - bs_align_window(j0w, n) + 15) >> 4);
                    const unsigned ce = (unsigned)((n - jb) >> 4);
                    desc[t] = (unsigned)e0 | (cs << 10) | (ce << 14) | (precise << 18) | (tail0 << 19);
                    // (the pairs with full windows first: their waves scan with the substitution bookkeeping, see below)
                    const unsigned bin = precise * 110u + cs * 10u + ce;
                    slot[t] = bin | (atomicAdd(&s_bins[bin], 1u) << 8);
                }
            }
            __syncthreads();
            if (wave == 0) {
                // exclusive prefix over the 256 bins: four per lane
                const unsigned b0 = s_bins[4 * lane], b1 = s_bins[4 * lane + 1], b2 = s_bins[4 * lane + 2], b3 = s_bins[4 * lane + 3];
                unsigned sum = b0 + b1 + b2 + b3;
#pragma unroll
                for (int sft = 1; sft < WAVE; sft <<= 1) {
                    const unsigned o = __shfl_up(sum, sft, WAVE);
                    if (lane >= sft) sum += o;
                }
                const unsigned first = sum - (b0 + b1 + b2 + b3);
                s_bins[4 * lane] = first;
                s_bins[4 * lane + 1] = first + b0;
                s_bins[4 * lane + 2] = first + b0 + b1;
                s_bins[4 * lane + 3] = first + b0 + b1 + b2;
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < CAH_M2_PAGE / 256; ++t)
                if ((int)threadIdx.x + 256 * t < count) s_ord[s_bins[slot[t] & 255u] + (slot[t] >> 8)] = desc[t];
            __syncthreads();
        }
        for (int sub = wave; sub * WAVE < count; sub += 4) {
            const int e_sorted = sub * WAVE + lane;
            const bool valid = e_sorted < count;
            // the pair: tail pages in page order, whole-read pages in window order
            int e0 = e_sorted;
            unsigned cs = 0, ce = 15, precise = 0, tail0 = 0;
            if (!tail_page && valid) {
                const uint32_t d = s_ord[e_sorted];
                e0 = (int)(d & 1023u); cs = (d >> 10) & 15u; ce = (d >> 14) & 15u; precise = (d >> 18) & 1u; tail0 = (d >> 19) & 15u;
            }
            const int64_t idx = page * CAH_M2_PAGE + e0;
            int64_t r = 0;
            unsigned tab_base = 0, adapter = 0, key = 0;
            if (valid) {
                const uint64_t pr = a.pairs[idx];
                r = (int64_t)(pr >> 32);
                adapter = (unsigned)(pr >> 8) & 0xFFFFu;
                key = (unsigned)pr & 0xFFu;
                tab_base = adapter * CAH_MULTI_TAB_STRIDE;
            }
            int pad = 0;
            const uint8_t* q = frame_of(r, pad);
            // A tail page's pairs all have the same window (one class, one read length), from column 4 * key: the scan
            // starts a whole number of 16-column chunks in front of the read end and skips the first chunk's columns in
            // front of the window.  A whole-read page's wave starts at its earliest window's chunk and ends behind its
            // latest window's (any earlier start, any later end is as exact): the lanes walk the same chunks, no lane
            // steps past its read.
            int j0, t0 = 0, jend = n, j0w_tail = 0;
            if (tail_page) {
                j0w_tail = min((int)key << 2, n);
                j0 = valid ? bs_align_window(j0w_tail, n) : n;
                j0 = __builtin_amdgcn_readfirstlane(j0);                // (lane 0 of a sub-batch is always valid)
                j0w_tail = __builtin_amdgcn_readfirstlane(j0w_tail);
                t0 = (j0w_tail - j0) & 15;
            } else {
                unsigned csm = cs, cem = ce;
#pragma unroll
                for (int sft = 1; sft < WAVE; sft <<= 1) {
                    csm = max(csm, (unsigned)__shfl_xor((int)csm, sft, WAVE));
                    cem = min(cem, (unsigned)__shfl_xor((int)cem, sft, WAVE));
                }
                csm = __builtin_amdgcn_readfirstlane(csm);
                cem = __builtin_amdgcn_readfirstlane(cem);
                j0 = max(0, n - 16 * (int)csm);
                // (a window from column 0 of a read that is no whole number of chunks: the chunks count from column 0)
                jend = min(n, j0 + ((n - 16 * (int)cem - j0 + 15) & ~15));
            }
            const int jstart = j0 + t0;                                 // first column the scan really looks at
            auto eq_of = [&](const Chunk& ck, int t) -> uint64_t {
                return s_scanmask[tab_base + multi_tab_index2(chunk_byte(ck, t) & 0xFFu)];
            };
            typename std::conditional<KIND == 0, BackScanState, BackScanState32<XR>>::type st;
            if constexpr (KIND == 0) bs_init(st, p); else bs32_init(st, p);
            auto step = [&](const uint64_t eq, const int jj) -> bool {
                if constexpr (KIND == 0) return bs_step<false>(st, eq, jj, p);
                else return bs32_step<false, XR>(st, (uint32_t)eq, (uint32_t)(eq >> 32), jj, p);
            };
            // A wave that holds whole-read pairs with their full windows -- adapters inside the read, most of them with
            // sequencing errors -- keeps the substitution / one-indel bookkeeping of back_scan.h (a third more per column):
            // four of five such pairs then finish here instead of in the cell DP.  The other waves (windows of one
            // occurrence, tail pairs) hardly ever see a candidate and scan without it.
            const bool track = !tail_page && __ballot(valid && precise == 0u) != 0ull;
            // A tail pair's window holds no last-row candidate (that takes a whole-adapter chunk: the pair would be a
            // whole-read pair): only the rows of its last column are asked -- the recurrence alone, nothing booked
            auto step_plain = [&](const uint64_t eq, const int jj) -> bool {
                if constexpr (KIND == 0) return bs_step<false, false>(st, eq, jj, p);
                else return bs32_step<false, XR, false>(st, (uint32_t)eq, (uint32_t)(eq >> 32), jj, p);
            };
            auto step_tracked = [&](const uint64_t eq, const int jj) -> bool {
                if constexpr (KIND == 0) return bs_step<true>(st, eq, jj, p);
                else return bs32_step<true, XR>(st, (uint32_t)eq, (uint32_t)(eq >> 32), jj, p);
            };
            int j = jstart, exact_j = 0;                                // (wave-uniform: every lane walks the same columns)
            bool done = !valid, exact = false;
            int pos = j0;
            Chunk cur = load_chunk_frame(q, pos, n, pad, valid ? n : 0);
            int first_t = t0;
            auto walk = [&](auto stepf) {
                while (j < jend) {
                    const Chunk nxt = load_chunk_frame(q, pos + 16, n, pad, (!done && pos + 16 < jend) ? n : 0);
                    // a window that starts at column 0 need not be a whole number of chunks: the last chunk then ends early
                    const int last_t = min(16, n - pos);
                    if (first_t == 0 && last_t == 16) {
                        // a whole chunk (the rule): no per-column tests
                        uint64_t eqq[2];
                        eqq[0] = eq_of(cur, 0);
                        eqq[1] = eq_of(cur, 1);
#pragma unroll
                        for (int t = 0; t < 16; ++t) {
                            const uint64_t eq = eqq[t & 1];
                            if (t + 2 < 16) eqq[t & 1] = eq_of(cur, t + 2);
                            ++j;
                            if (stepf(eq, j) && !exact) { exact = true; exact_j = j; }
                        }
                    } else {
#pragma unroll 1
                        for (int t = first_t; t < last_t; ++t) {       // wave-uniform bounds
                            ++j;
                            if (stepf(eq_of(cur, t), j) && !exact) { exact = true; exact_j = j; }
                        }
                    }
                    if (exact) done = true;
                    first_t = 0;
                    pos += 16;
                    cur = nxt;
                }
            };
            int o0 = 0, o1 = 0;
            // rows that cannot be acceptable are not looked at: an acceptable row's alignment lies inside the window
            int max_row;
            if (tail_page) {
                // (lo pages: the rows of the first error class -- a row with a higher threshold needs a chunk of ITS class)
                max_row = min(p.m, n - j0w_tail + p.kacc);
                if ((hdr >> 24) == 0u) max_row = min(max_row, a.rows_lo);
            } else {
                max_row = min(p.m, n - j0 + p.kacc);
            }
            // a pair scanned on the window of its one occurrence: nothing behind the window matters -- the state is that
            // of an inner column ("stopped", back_scan.h), whether or not the wave went on to the read's end
            const bool stopped = precise != 0;
            int cls, jfa = -1;
            auto thr_of = [&](int i) { return s_thr_last[i]; };
            if (KIND != 1 && tail_page && (hdr >> 24) == 0u && a.rows_lo <= 32) {
                // a "lo" page asks for rows 1 .. rows_lo of the last column alone: the adapter's first 32 rows in ONE plain
                // 32-bit word do (a row's cost depends on the rows above it only) -- no explicit rows, no second word
                BackScanParams p32 = p;
                p32.m = 32;
                BackScanState32<0> s0;
                bs32_init(s0, p32);
                walk([&](const uint64_t eq, const int jj) -> bool {
                    uint32_t rows;
                    if constexpr (KIND == 0) rows = (uint32_t)(eq >> (64 - p.m));
                    else rows = ((uint32_t)eq << XR) | ((uint32_t)(eq >> 32) & ((1u << XR) - 1u));
                    return bs32_step<false, 0, false>(s0, rows, 0u, jj, p32);
                });
                cls = bs32_finish<0, false>(s0, n, jstart, p32, thr_of, o0, o1, false, max_row);
            } else {
                if (tail_page) walk(step_plain); else if (track) walk(step_tracked); else walk(step);
                if (track) {
                    if constexpr (KIND == 0) cls = bs_finish<true>(st, n, jstart, p, thr_of, o0, o1, stopped, max_row);
                    else cls = bs32_finish<XR, true>(st, n, jstart, p, thr_of, o0, o1, stopped, max_row);
                } else {
                    if constexpr (KIND == 0) cls = bs_finish<false>(st, n, jstart, p, thr_of, o0, o1, stopped, max_row);
                    else cls = bs32_finish<XR, false>(st, n, jstart, p, thr_of, o0, o1, stopped, max_row);
                }
                jfa = st.jfa;
            }
            if (stopped) {
                if (jfa < 0) cls = BS_NONE;                             // no candidate in the window: there is none at all
                else if (tail0 > 0) { cls = BS_DP; o0 = max(jstart, jfa - reach); o1 = 2 * n + 1; }   // ... unless an
                // error-free overlap is acceptable too: the cell DP sorts that out, to the read's end
            }
            if (exact) { cls = BS_EXACT_FULL; o0 = exact_j; }
            // the cell DP of a tail pair runs over the full reach: band, last_filled and the stale origin of its final
            // scan are only proven equal to the reference's from column start + m + k + 1 on (column skipping)
            if (cls == BS_DP && tail_page) o0 = max(0, (jfa >= 0 ? jfa : n) - reach);
            if (valid) {
                // (a view's frame -> the view: every coordinate moves by the pad; a shortcut whose alignment would BEGIN in the
                // pad is no shortcut -- there the reference deletes adapter characters where the frame substitutes NULs, other
                // scores, other origins: the pair takes the cell DP on the view, from its first column, every row of its last)
                bool in_pad = false;
                auto shortcut = [&](const int score, const int errors, const int ref_stop, const int qs, const int qe) {
                    if (qs < pad) in_pad = true;
                    else atomicMax(a.best_key + r, pack_best(score, errors, (int)adapter, ref_stop, qs - pad, qe - pad));
                };
                if (cls == BS_EXACT_FULL) shortcut(p.m, 0, p.m, o0 - p.m, o0);
                else if (cls == BS_EXACT_TAIL) shortcut(o0 - 2 * o1, o1, o0, n - o0, n);
                else if (cls == BS_SUBS_FULL) shortcut(p.m - 2 * o1, o1, p.m, o0 - p.m, o0);
                else if (cls == BS_INDEL1_FULL)
                    shortcut(p.m - 2 * (o1 >> 1) - (o1 & 1), o1 >> 1, p.m, o0 - p.m + ((o1 & 1) ? 1 : -1), o0);
                else if (cls == BS_NONE && stopped && tail0 > 0)
                    atomicMax(a.best_key + r, pack_best((int)tail0, 0, (int)adapter, (int)tail0, n - pad - (int)tail0, n - pad));
                if (in_pad) { cls = BS_DP; o0 = 0; o1 = 2 * n + 1; }
                if (cls == BS_DP && pad > 0) {
                    // (the cell DP sees the view: the window's columns moved by the pad)
                    const int last = max(0, min(n, o1 >> 1) - pad);
                    o0 = max(0, o0 - pad);
                    o1 = 2 * last + (o1 & 1);
                }
            }
            const bool to_dp = valid && cls == BS_DP;
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
                const uint32_t ent = (uint32_t)e0 | ((uint32_t)o0 << 10) | ((uint32_t)o1 << 18);
                if (to_front) s_dp[(int)sf + __popcll(bf & below)] = ent;
                else if (to_back) s_dp[CAH_M2_PAGE - 1 - ((int)sb + __popcll(bb & below))] = ent;
            }
        }
        // flush the page's DP work list
        __syncthreads();
        const unsigned nf = s_nf, nb = s_nb;
        if (threadIdx.x == 0) {
            s_gf = nf ? atomicAdd(a.dp_count_front, (unsigned long long)nf) : 0ull;
            s_gb = nb ? atomicAdd(a.dp_count_back, (unsigned long long)nb) : 0ull;
        }
        __syncthreads();
        const unsigned long long gf = s_gf, gb = s_gb;
        for (unsigned e = threadIdx.x; e < nf; e += blockDim.x) {
            const int64_t slot = (int64_t)(gf + e);
            const uint32_t ent = s_dp[e];
            a.dp_queue[slot] = (int32_t)(page * CAH_M2_PAGE + (ent & 1023u));
            a.dp_win[2 * slot] = (int32_t)((ent >> 10) & 255u);
            a.dp_win[2 * slot + 1] = (int32_t)(ent >> 18);
        }
        for (unsigned e = threadIdx.x; e < nb; e += blockDim.x) {
            const int64_t slot = a.dp_cap - 1 - (int64_t)(gb + e);
            const uint32_t ent = s_dp[CAH_M2_PAGE - 1 - (int)e];
            a.dp_queue[slot] = (int32_t)(page * CAH_M2_PAGE + (ent & 1023u));
            a.dp_win[2 * slot] = (int32_t)((ent >> 10) & 255u);
            a.dp_win[2 * slot + 1] = (int32_t)(ent >> 18);
        }
    }
}

// ---- launchers ----------------------------------------------------------------------------------------------------
size_t multi2_lds_bytes(const CahMulti2Header& h) { return m2_layout((int)h.n_entries, h.n_adapters).total; }

// the batch's read length must leave every tail sweep inside the half-row the slot still holds after the main pass
bool multi2_read_len_ok(const CahMulti2Header& h, int n) {
    if (!h.ok || n < 16 || n > M2_MAX_LEN) return false;
    if (multi2_lds_bytes(h) > 160 * 1024) return false;
    const int U = (n + 15) >> 4, H1 = U <= M2_HALF ? U : (U + 1) >> 1, H2 = U - H1;
    const int tail_off = H2 > 0 ? 16 * H1 : 0;
    // the tail slots are probed in the last chunks of the last half-row: every slot's first position must lie in it,
    // and at most four chunks (the words of a tail chunk wait in the row unit in front of it) reach from there to the read's end
    int p0 = n;
    for (int j = 0; j < h.tq_n; j++) p0 = std::min(p0, std::max(0, n + h.tq_qc[j] - 1 - h.tq_open[j]));
    const int tail_base = p0 & ~15;
    if (tail_base < tail_off || n - tail_base > 64) return false;
    return true;
}

int multi2_tile_reads() { return M2_TILE; }

hipError_t launch_multi_stream(const Multi2Args& a, const CahMulti2Header& h, int grid, hipStream_t s) {
    const size_t lds = multi2_lds_bytes(h);
    // (the attribute belongs to the kernel ON A DEVICE: one flag per device, set under a lock -- feeder threads of several
    // GPUs come through here at once)
    static std::mutex attr_mu;
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(attr_mu);
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            e = hipFuncSetAttribute((const void*)k_multi_stream<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute((const void*)k_multi_stream<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute((const void*)k_multi_stream<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute((const void*)k_multi_stream<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    if (grid < 1) grid = 1;
    const bool w8 = h.q_mask[M2_W] == (1 << 8);
    if (a.view_starts) {
        if (w8) hipLaunchKernelGGL((k_multi_stream<true, true>), dim3(grid), dim3(M2_WAVES * WAVE), lds, s, a);
        else hipLaunchKernelGGL((k_multi_stream<false, true>), dim3(grid), dim3(M2_WAVES * WAVE), lds, s, a);
    } else {
        if (w8) hipLaunchKernelGGL((k_multi_stream<true, false>), dim3(grid), dim3(M2_WAVES * WAVE), lds, s, a);
        else hipLaunchKernelGGL((k_multi_stream<false, false>), dim3(grid), dim3(M2_WAVES * WAVE), lds, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_multi_scan(const Multi2ScanArgs& a, int64_t max_pages, int n_cus, hipStream_t s) {
    int64_t need = max_pages < 1 ? 1 : max_pages;
    const int64_t cap = (int64_t)8 * n_cus;
    const dim3 grid((unsigned)(need < cap ? need : cap));
    const size_t lds = sizeof(uint64_t) * (size_t)a.n_adapters * CAH_MULTI_TAB_STRIDE;
    switch (a.kind) {
        case 1: hipLaunchKernelGGL((k_multi_scan<1>), grid, dim3(256), lds, s, a); break;
        case 2: hipLaunchKernelGGL((k_multi_scan<2>), grid, dim3(256), lds, s, a); break;
        case 3: hipLaunchKernelGGL((k_multi_scan<3>), grid, dim3(256), lds, s, a); break;
        default: hipLaunchKernelGGL((k_multi_scan<0>), grid, dim3(256), lds, s, a); break;
    }
    return hipGetLastError();
}
