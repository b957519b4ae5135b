// scan3.hip -- k_back_scan3: the cost scan of a 3' adapter (back_scan.h) on windows chosen from the adapter's own chunk
// occurrences ("bs3" in back_scan.h; round 5).  Same inputs and outputs as k_back_scan<false, KIND> (kernels.hip): the
// prefilter's survivor queue with its first-hit keys in, result rows / the cell DP's work list / the straggler list out.
// OPT-IN (CAH_SCAN3=1): exact, but not the faster of the two -- DESIGN.md 9, profiles/r05/scan3_vs_scan_pmc.txt.
//
// k_back_scan walks every survivor from (first k-mer hit) - m - k - 1 to 23 columns behind its last acceptable column (or
// the read's end): six to seven 16-column chunks per wave on the headline workload, of which a lane needs 4.6.  Here a
// lane first walks a SHIFT-AND word over the columns behind the prefilter's position (8 instructions per column against
// ~28): where do the adapter's k + 1 chunks occur, and on which diagonals (G = (G << 1) | hits: every occurrence of one
// diagonal lands on one bit)?  No occurrence: only the read's tail can match (class T, the last m + k + 1 columns).
// Diagonals Smin .. Smax: every candidate that can matter lies within [Smin - kacc - 1, Smax + m + kacc] -- 40 columns + the
// spread, three chunks -- and the scan STOPS there when the read goes on for more than gap_last columns (class F), or the
// window is joined with the tail (class E).  Far-apart copies (class C) and class-F reads whose window holds no acceptable
// column (2 % of the headline's survivors) go to the straggler list, which the second launch scans the old, conservative
// way; should that list be full the lane goes round again in place with the window to the read's end.  Exactness:
// back_scan.h ("bs3"), replayed against the oracle by tests/host_model/back_model.cpp: bm_locate_batch3 (the same header
// under g++); tests/test_gpu_scan.py::test_scan3_windows_from_chunk_occurrences.
//
// One read per lane, sub-batches of 64 queue entries per wave, tiles of SCAN3_TILE entries per workgroup and atomic.
// Windows are per lane (every lane loads its own 16-character chunks), so the lanes of a wave need not walk the same
// columns -- only the same NUMBER of chunks: three for most.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <type_traits>

#include "cah_device.h"
#include "kernels.h"
#include "dev_common.h"
#include "back_scan.h"

#define SCAN3_TILE 1024

#ifdef SCAN_TRACE
// developer build only (-DSCAN_TRACE): s_memtime stamps of one wave, per sub-batch of 64 queue entries:
// [0] start, [1] pre-pass done, [2] scan done, [3] classified, [4] stored; [5] pre-pass chunks, [6] scan chunks, [7] lanes retried
#define S3_TRACE_N 256
__device__ unsigned long long g_scan3_trace[S3_TRACE_N * 8];
extern "C" int cah_debug_scan3_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan3_trace), sizeof(g_scan3_trace)) == hipSuccess ? 0 : 1;
}
#define S3_STAMP(st) do { if (blockIdx.x == 11 && wave == 1 && trace_i < S3_TRACE_N) { \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_scan3_trace[trace_i * 8 + (st)] = t_; } } while (0)
#define S3_NOTE(st, v) do { if (blockIdx.x == 11 && wave == 1 && trace_i < S3_TRACE_N && lane == 0) g_scan3_trace[trace_i * 8 + (st)] = (unsigned long long)(v); } while (0)
#else
#define S3_STAMP(st) do { } while (0)
#define S3_NOTE(st, v) do { } while (0)
#endif

template <int KIND>
#ifndef CAH_SCAN3_WAVES
#define CAH_SCAN3_WAVES 4
#endif
__global__ __launch_bounds__(256, CAH_SCAN3_WAVES) void k_back_scan3(ScanArgs a) {
    static_assert(KIND >= 1 && KIND <= 3, "k_back_scan3: the 32-bit forms");
    constexpr int XR = KIND >= 2 ? KIND - 1 : 0;
    __shared__ int s_thr_last[CAH_MAX_M + 1];
    __shared__ int s_list[SCAN3_TILE * 3];         // the tile's DP work list: (read, first column, last column * 2 + scan);
                                                   // bounded windows from slot 0 up, windows to the read's end from the top down
    __shared__ unsigned s_nf, s_nb;
    __shared__ long long s_tile;
    __shared__ unsigned long long s_gf, s_gb;
    // {rows X+1..m, rows 1..X} of every byte value: a static 256-entry table, the entry offset is "byte << 3"
    __shared__ __attribute__((aligned(16))) uint64_t s_sm256[256];
    const CahMatcher* mt = a.matcher;
    for (int i = threadIdx.x; i < 256; i += blockDim.x)
        s_sm256[i] = i < CAH_TABLE_CHARS ? bs32_table_entry(mt->scanmask[i], mt->m) : 0ull;
    for (int i = threadIdx.x; i <= CAH_MAX_M; i += blockDim.x) s_thr_last[i] = mt->thr_last[i];
    BackScanParams p;
    p.m = mt->m; p.k = mt->k; p.kacc = mt->kacc; p.min_overlap = mt->min_overlap; p.half_m = mt->m / 2;
    Bs3Geom g;
    g.start = mt->bs3_start; g.end = mt->bs3_end; g.roff = mt->bs3_roff; g.maxlen = mt->bs3_maxlen; g.ok = 1;
    const int reach = p.m + p.k + 1, range = bs3_range(p);
    const int lane = wave_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int64_t total = (int64_t)(*a.queue_count);
    if (a.queue_limit > 0 && total > a.queue_limit) total = a.queue_limit;
    const int tile = SCAN3_TILE;
    unsigned low16 = 0xFFFFu;                                          // (a VGPR: v_bitop3 takes no literal)
    unsigned v_start = g.start, v_end = g.end;                         // (VGPRs: an SGPR operand doubles a simple instruction's issue time)
    asm volatile("" : "+v"(low16), "+v"(v_start), "+v"(v_end));
#ifdef SCAN_TRACE
    int trace_i = 0;
#endif
    auto eq_lo = [&](const unsigned w, const int b) -> uint32_t {      // rows X+1..m of byte b of a chunk dword
        const unsigned byte_off = ((w >> (8 * b)) & 0xFFu) << 3;
        return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const unsigned char*>(s_sm256) + byte_off);
    };
    auto eq_of = [&](const Chunk& ck, const int t) -> uint64_t {
        const unsigned byte_off = ((ck.w[t >> 2] >> (8 * (t & 3))) & 0xFFu) << 3;
        return *reinterpret_cast<const uint64_t*>(reinterpret_cast<const unsigned char*>(s_sm256) + byte_off);
    };
    // the chunk moved down by one character (the rolled loops of the rare paths take their character from byte 0: no
    // register is indexed by a loop counter, and the code stays small -- an unrolled copy of every path made the kernel
    // five times the size of k_back_scan, and the instruction cache its bound)
    auto shift_chunk = [](Chunk& c) {
        c.w[0] = __builtin_amdgcn_alignbit(c.w[1], c.w[0], 8);
        c.w[1] = __builtin_amdgcn_alignbit(c.w[2], c.w[1], 8);
        c.w[2] = __builtin_amdgcn_alignbit(c.w[3], c.w[2], 8);
        c.w[3] >>= 8;
    };

    for (;;) {
        __syncthreads();                                   // previous tile flushed
        if (threadIdx.x == 0) {
            s_tile = (long long)atomicAdd(a.work_counter, (unsigned long long)tile);
            s_nf = 0; s_nb = 0;
        }
        __syncthreads();
        const int64_t tile_base = s_tile;
        if (tile_base >= total) break;

        // the wave's NEXT sub-batch: its queue entries are fetched while this one's pre-pass runs, and the cache lines its own
        // pre-pass will ask for are touched while this one is scanned (the pass is short against a trip to HBM)
        bool pf_have = false, pf_valid = false;
        int pf_r = 0;
        unsigned pf_key = 0;
        for (int sub = wave; sub < tile / WAVE; sub += 4) {
            const int64_t base = tile_base + (int64_t)sub * WAVE;
            if (base >= total) break;
            S3_STAMP(0);
            const int64_t idx = base + lane;
            bool valid = idx < total;
            int64_t r = 0;
            unsigned key = 0;
            if (pf_have) { valid = pf_valid; r = pf_r; key = pf_key; }
            else if (valid) { r = (int64_t)a.queue[idx]; key = a.queue_keys[idx]; }
            if (r < 0) { valid = false; r = 0; }           // (an unused slot of a straggler list)
            const bool next_sub = sub + 4 < tile / WAVE && base + 4 * WAVE < total;      // wave-uniform
            bool valid_n = false;
            int r_n = 0;
            unsigned key_n = 0;
            if (next_sub) {
                const int64_t idx_n = base + 4 * WAVE + lane;
                valid_n = idx_n < total;
                if (valid_n) { r_n = a.queue[idx_n]; key_n = a.queue_keys[idx_n]; }
            }
            int64_t off = 0, n64 = 0;
            if (valid) read_extent(a.offsets, a.lens, a.uniform_first, a.uniform_len, r, off, n64);
            bool invalid = false;
            if (n64 > a.max_read_len) { invalid = true; n64 = 0; }
            const int n = (int)n64;
            const uint8_t* q = a.seqs + off;
            const int key4 = (int)key << CAH_KEY_SHIFT;
            const int j0_old = max(0, key4 - reach);
            unsigned bad_chars = 0;

            // ---- pre-pass: the diagonals of the chunk occurrences from the prefilter's position on (back_scan.h: bs3_pre_step
            // column by column, bs3_pre_harvest at every chunk's end -- written out here in the cheap instruction forms)
            Bs3Pre pre;
            bs3_pre_init(pre);
            const int p0 = bs3_pre_start(key4, g, n);
            {
                // The characters come in BLOCKS of four chunks, the loads in flight together (a chunk of the pre-pass is ~250
                // issue cycles, far less than a trip to memory: chunk by chunk the pass waited for every load).  The first
                // block lies inside the read (bs3_pre_start): four plain loads; a block that holds a read's tail takes
                // load_chunk with its three cases -- and most lanes are through after one block.
                int pos = p0;
                int last = valid ? n : 0;                           // last column this lane looks at (bs3_pre_last)
#pragma unroll 1
                for (;;) {
                    if (!__any(pos < last)) break;
                    Chunk b0, b1, b2, b3;
                    if (__all(pos >= last || pos + 64 <= n)) {
                        const bool on = pos < last;
                        b0 = load_chunk_interior(q, pos, on); b1 = load_chunk_interior(q, pos + 16, on);
                        b2 = load_chunk_interior(q, pos + 32, on); b3 = load_chunk_interior(q, pos + 48, on);
                    } else {
                        const int lim = pos < last ? n : 0;
                        b0 = load_chunk(q, pos, n, lim); b1 = load_chunk(q, pos + 16, n, lim);
                        b2 = load_chunk(q, pos + 32, n, lim); b3 = load_chunk(q, pos + 48, n, lim);
                    }
#pragma unroll 1
                    for (int u = 0; u < 4; ++u, pos += 16) {
                        const bool active = pos < last;
                        if (!__any(active)) break;
                        const Chunk cur = b0;
                        b0 = b1; b1 = b2; b2 = b3;
                        bad_chars |= cur.w[0] | cur.w[1] | cur.w[2] | cur.w[3];
                        uint32_t M = pre.M, glo = 0, ghi = 0;
#pragma unroll
                        for (int t = 0; t < 16; ++t) {
                            const uint32_t e = eq_lo(cur.w[t >> 2], t & 3);
                            M = BS_BITOP3(bs_dbl(M), v_start, e, 0xA8u);               // ((M << 1) | START) & eq
                            const uint32_t h = M & v_end;
                            glo = BS_BITOP3(bs_dbl(glo), h, low16, 0xF8u);             // (glo << 1) | (h & 0xFFFF)
                            ghi = bs_dbl(ghi) | (h >> 16);
                        }
                        pre.M = M;
                        // (a lane behind its range walks along: its occurrences do not count)
                        if (__any(active && (glo | ghi) != 0)) {
                            pre.glo = active ? glo : 0u; pre.ghi = active ? ghi : 0u;
                            bs3_pre_harvest(pre, pos + 16, g);
                            if (active) last = bs3_pre_last(n, pre.found != 0, pre.smax, range);
                        }
                        S3_NOTE(5, (pos - p0) / 16 + 1);
                    }
                }
            }
            S3_STAMP(1);
            // (the next sub-batch's lines: two words, 60 characters apart, from where its pre-pass will start)
            unsigned touch = 0;
            pf_have = next_sub;
            if (next_sub) {
                pf_valid = valid_n && r_n >= 0; pf_r = r_n; pf_key = key_n;
                if (pf_valid) {
                    int64_t o_n = 0, n_n = 0;
                    read_extent(a.offsets, a.lens, a.uniform_first, a.uniform_len, (int64_t)r_n, o_n, n_n);
                    if (n_n >= 4 && n_n <= a.max_read_len) {
                        const int nn = (int)n_n;
                        const int t0 = min(bs3_pre_start((int)key_n << CAH_KEY_SHIFT, g, nn), nn - 4), t1 = min(t0 + 60, nn - 4);
                        unsigned x0, x1;
                        __builtin_memcpy(&x0, a.seqs + o_n + t0, 4);
                        __builtin_memcpy(&x1, a.seqs + o_n + t1, 4);
                        touch = x0 ^ x1;
                    }
                }
            }

            // ---- the window; class C leaves for the straggler list when that has room
            Bs3Win w = bs3_window(pre, n, j0_old, p);
            bool retry = false;
            bool want_retry = valid && w.cls == BS3_C;
            auto to_retry_list = [&](const bool want) -> bool {       // true: the lane's read is in the list
                const unsigned long long bw = __ballot(want);
                if (!bw) return false;
                if (!a.retry_queue || a.retry_cap <= 0) return false;
                const int cnt = __popcll(bw);
                unsigned long long slot = 0;
                if (lane == 0) slot = atomicAdd(a.retry_count, (unsigned long long)cnt);
                slot = __shfl(slot, 0, WAVE);
                const int64_t e = (int64_t)slot + __popcll(bw & ((1ull << lane) - 1ull));
                if ((int64_t)(slot + cnt) <= a.retry_cap) {
                    if (want) { a.retry_queue[e] = (int32_t)r; a.retry_keys[e] = (uint8_t)key; }
                    return want;
                }
                // the list is full: what the wave reserved inside it is marked unused (the second launch walks
                // min(count, capacity) slots), the lanes go on in place
                if (want && e < a.retry_cap) a.retry_queue[e] = -1;
                return false;
            };
            if (to_retry_list(want_retry)) retry = true;

            // ---- the scan, on every lane's own window
            // (FAST: whole chunks take the unguarded, unrolled path -- the rule; the second round below does without it)
            auto scan_window = [&](BackScanState32<XR>& S, bool& ex, int& exj, const bool on, const Bs3Win& ww, auto fast_c) {
                constexpr bool FAST = decltype(fast_c)::value;
                bs32_init(S, p);
                ex = false;
                bool done = !on;
                int j = ww.start, pos = ww.start;
                const int jend = on ? ww.jend : ww.start, jlim = ww.jlim;
#pragma unroll 1
                for (;;) {
                    if (!__any(!done && j < jend)) break;
                    // (blocks of three chunks, the loads in flight together: a class-F window is one block.  Windows are whole
                    // chunks inside the read -- plain loads; only a read shorter than its window takes load_chunk)
                    Chunk b0, b1, b2;
                    {
                        const bool w0 = !done && j < jend, w1 = !done && j + 16 < jend, w2 = !done && j + 32 < jend;
                        if (__all((!w0 || pos + 16 <= n) && (!w1 || pos + 32 <= n) && (!w2 || pos + 48 <= n))) {
                            b0 = load_chunk_interior(q, pos, w0); b1 = load_chunk_interior(q, pos + 16, w1);
                            b2 = load_chunk_interior(q, pos + 32, w2);
                        } else {
                            b0 = load_chunk(q, pos, n, w0 ? n : 0); b1 = load_chunk(q, pos + 16, n, w1 ? n : 0);
                            b2 = load_chunk(q, pos + 32, n, w2 ? n : 0);
                        }
                    }
#pragma unroll 1
                    for (int u = 0; u < 3; ++u, pos += 16) {
                        if (!__any(!done && j < jend)) break;
                        Chunk cur = b0;
                        b0 = b1; b1 = b2;
                        bad_chars |= cur.w[0] | cur.w[1] | cur.w[2] | cur.w[3];
                        if (FAST && __all(done || j + 16 <= jend)) {
                            // Whole chunks, no per-column tests.  Lanes that are through sit the chunk out (ONE branch around
                            // the sixteen columns): their windows end at other columns than their neighbours', and the state
                            // they ended with -- the last column's rows, the booked candidates -- is what the classification reads.
                            if (!done) {
                                uint64_t eqq[2];
                                eqq[0] = eq_of(cur, 0); eqq[1] = eq_of(cur, 1);
#pragma unroll
                                for (int t = 0; t < 16; ++t) {
                                    const uint64_t eq = eqq[t & 1];
                                    if (t + 2 < 16) eqq[t & 1] = eq_of(cur, t + 2);
                                    ++j;
                                    if (bs32_step<true, XR>(S, (uint32_t)eq, (uint32_t)(eq >> 32), j, p, jlim) && !ex) { ex = true; exj = j; }   // (an exact lane's state is not read again)
                                }
                                if (ex) done = true;
                            }
                        } else {
                            // (a read shorter than its window: column by column, rolled)
#pragma unroll 1
                            for (int t = 0; t < 16; ++t) {
                                const uint64_t eq = eq_of(cur, 0);
                                shift_chunk(cur);
                                if (!done && j < jend) {
                                    ++j;
                                    if (bs32_step<true, XR>(S, (uint32_t)eq, (uint32_t)(eq >> 32), j, p, jlim)) { ex = true; exj = j; done = true; }
                                }
                            }
                        }
                        if (j >= jend) done = true;
                    }
                }
            };
            BackScanState32<XR> st;
            bool exact = false;
            int exact_j = 0;
            scan_window(st, exact, exact_j, valid && !retry, w, std::true_type{});
            // A class-F window without an acceptable column -- a chance occurrence, or a copy with more errors than the adapter
            // takes: what lies behind the pre-pass range is unknown, so the read goes the conservative way: the straggler
            // list, or (no room there) once more in place, from the same start to the read's end, every column booked.
            const bool again = valid && !retry && !exact && w.cls == BS3_F && st.jla < 0;
            if (__any(again)) {
                if (to_retry_list(again)) retry = true;
                const bool redo = again && !retry;
                if (__any(redo)) {
                    const Bs3Win w2 = bs3_window_to_end(w.start, n, p);
                    BackScanState32<XR> st2;
                    bool ex2 = false;
                    int exj2 = 0;
                    scan_window(st2, ex2, exj2, redo, w2, std::false_type{});
                    if (redo) { st = st2; exact = ex2; exact_j = exj2; w = w2; }
                }
            }
            S3_STAMP(2);
            S3_NOTE(6, __popcll(__ballot(valid && w.cls == BS3_F)) + 100 * __popcll(__ballot(valid && w.cls == BS3_E)) + 10000 * __popcll(__ballot(valid && w.cls == BS3_T)));
            S3_NOTE(7, __popcll(__ballot(retry)));
            if (bad_chars & 0x80808080u) invalid = true;
            const bool stopped = valid && !retry && !exact && w.cls == BS3_F && st.jla >= 0;

            // ---- classification and results (k_back_scan's)
            int o0 = 0, o1 = 0;
            int cls = BS_NONE;
            bool valid_out = valid;
            if (valid && !exact && !retry)
                cls = bs32_finish<XR, true>(st, n, w.start, p, [&](int i) { return s_thr_last[i]; }, o0, o1, stopped, CAH_BS_ALL_ROWS, j0_old);
            if (exact) { cls = BS_EXACT_FULL; o0 = exact_j; }
            if (retry) { cls = BS_NONE; valid_out = false; }
            S3_STAMP(3);
            if (valid_out) {
                const bool full = cls == BS_EXACT_FULL || cls == BS_SUBS_FULL || cls == BS_INDEL1_FULL;
                const bool found = !invalid && (full || cls == BS_EXACT_TAIL);
                int t1 = p.m, t2 = o0 - p.m, t3 = o0, sc = p.m, cost = 0;                 // EXACT_FULL
                if (cls == BS_EXACT_TAIL) { t1 = o0; t2 = n - o0; t3 = n; sc = o0 - 2 * o1; cost = o1; }
                if (cls == BS_SUBS_FULL) { sc = p.m - 2 * o1; cost = o1; }
                if (cls == BS_INDEL1_FULL) { t2 = o0 - p.m + ((o1 & 1) ? 1 : -1); sc = p.m - 2 * (o1 >> 1) - (o1 & 1); cost = o1 >> 1; }
                if (invalid || cls != BS_DP)
                    store_result(a.out6, a.status, a.best_adapter, a.adapter_index, a.merge_best, r, invalid, found,
                                 0, t1, t2, t3, sc, cost);
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
                if (to_front) {
                    const int e = (int)sf + __popcll(bf & below);
                    s_list[3 * e] = (int)r; s_list[3 * e + 1] = o0; s_list[3 * e + 2] = o1;
                } else if (to_back) {
                    const int e = SCAN3_TILE - 1 - ((int)sb + __popcll(bb & below));
                    s_list[3 * e] = (int)r; s_list[3 * e + 1] = o0; s_list[3 * e + 2] = o1;
                }
            }
            asm volatile("" :: "v"(touch));                 // (the touched words are needed by nobody: only their cache lines)
            S3_STAMP(4);
#ifdef SCAN_TRACE
            if (blockIdx.x == 11 && wave == 1) ++trace_i;
#endif
        }

        // flush the tile's DP work list
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
            a.dp_queue[slot] = s_list[3 * e];
            a.dp_win[2 * slot] = s_list[3 * e + 1];
            a.dp_win[2 * slot + 1] = s_list[3 * e + 2];
        }
        for (unsigned e = threadIdx.x; e < nb; e += blockDim.x) {
            const int64_t slot = a.dp_cap - 1 - (int64_t)(gb + e);
            const int le = SCAN3_TILE - 1 - (int)e;
            a.dp_queue[slot] = s_list[3 * le];
            a.dp_win[2 * slot] = s_list[3 * le + 1];
            a.dp_win[2 * slot + 1] = s_list[3 * le + 2];
        }
    }
}

// the launcher: kind = bs_kind_of(m) in 1..3 (api.cpp asks bs3_ok first)
hipError_t launch_back_scan3(const ScanArgs& a, int64_t max_items, int n_cus, hipStream_t s) {
    int64_t need = (max_items + SCAN3_TILE - 1) / SCAN3_TILE;
    if (need < 1) need = 1;
    const int64_t cap = (int64_t)8 * n_cus;
    const dim3 grid((unsigned)(need < cap ? need : cap));
    switch (a.kind) {
        case 1: hipLaunchKernelGGL((k_back_scan3<1>), grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_back_scan3<2>), grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((k_back_scan3<3>), grid, dim3(256), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
