// scan3.hip -- k_back_scan3: the cost scan of a 3' adapter (back_scan.h) on windows chosen from the adapter's own chunk
// occurrences ("bs3" in back_scan.h; round 5).  Same inputs and outputs as k_back_scan<false, KIND> (kernels.hip): the
// prefilter's survivor queue with its first-hit keys in, result rows / the cell DP's work list / the straggler list out.
//
// k_back_scan walks every survivor from (first k-mer hit) - m - k - 1 to 23 columns behind its last acceptable column (or
// the read's end): six to seven 16-column chunks per wave on the headline workload, of which a lane needs 4.6.  Here a
// lane first walks a SHIFT-AND word over the columns behind the prefilter's position (4 instructions per column against
// ~28): where do the adapter's k + 1 chunks occur?  No occurrence: only the read's tail can match (class T, the last
// m + k + 1 columns).  Occurrences on one diagonal band: every candidate that can matter lies within
// [S1 - 2 kacc - 1, S1 + 2 kacc + m] -- 47 columns, three chunks -- and the scan STOPS there when the read goes on for
// more than gap_last columns (class F), or the window is joined with the tail (class E).  Anything else (a second copy,
// a chance occurrence next to a real one: 0.2 % of the headline's survivors) and class-F reads whose window holds no
// acceptable column (2 %) go to the straggler list, which the second launch scans the old, conservative way; should
// that list be full the lane goes round again in place with the window to the read's end.  Exactness: back_scan.h
// ("bs3"), replayed against the oracle by tests/host_model/back_model.cpp: bm_locate_batch3 (the same header under g++).
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

template <int KIND>
__global__ __launch_bounds__(256, 5) void k_back_scan3(ScanArgs a) {
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
    auto eq_lo = [&](const unsigned w, const int b) -> uint32_t {      // rows X+1..m of byte b of a chunk dword
        const unsigned byte_off = ((w >> (8 * b)) & 0xFFu) << 3;
        return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const unsigned char*>(s_sm256) + byte_off);
    };
    auto eq_of = [&](const Chunk& ck, const int t) -> uint64_t {
        const unsigned byte_off = ((ck.w[t >> 2] >> (8 * (t & 3))) & 0xFFu) << 3;
        return *reinterpret_cast<const uint64_t*>(reinterpret_cast<const unsigned char*>(s_sm256) + byte_off);
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

        for (int sub = wave; sub < tile / WAVE; sub += 4) {
            const int64_t base = tile_base + (int64_t)sub * WAVE;
            if (base >= total) break;
            const int64_t idx = base + lane;
            bool valid = idx < total;
            int64_t r = 0;
            unsigned key = 0;
            if (valid) { r = (int64_t)a.queue[idx]; key = a.queue_keys[idx]; }
            if (r < 0) { valid = false; r = 0; }           // (an unused slot of a straggler list)
            int64_t off = 0, n64 = 0;
            if (valid) read_extent(a.offsets, a.lens, a.uniform_first, a.uniform_len, r, off, n64);
            bool invalid = false;
            if (n64 > a.max_read_len) { invalid = true; n64 = 0; }
            const int n = (int)n64;
            const uint8_t* q = a.seqs + off;
            const int key4 = (int)key << CAH_KEY_SHIFT;
            const int j0_old = max(0, key4 - reach);
            unsigned bad_chars = 0;

            // ---- pre-pass: the chunk occurrences from the prefilter's position on (bs3_pre_step, column by column)
            Bs3Pre pre;
            bs3_pre_init(pre);
            const int p0 = bs3_pre_start(key4, g);
            {
                int pos = p0;
                Chunk cur = load_chunk(q, pos, n, valid ? n : 0);
                for (int c = 0;; ++c) {
                    const bool active = valid && c < bs3_pre_chunks(p0, n, pre.found != 0, pre.s1, range);
                    if (!__any(active)) break;
                    const Chunk nxt = load_chunk(q, pos + 16, n, active ? n : 0);
                    bad_chars |= cur.w[0] | cur.w[1] | cur.w[2] | cur.w[3];
                    const uint32_t bad_before = pre.bad;
                    if (__all(!active || pre.found != 0)) {
                        // every lane at work knows its first occurrence: shift-and, band and the out-of-band bits only
                        uint32_t M = pre.M, band = pre.band, bad = pre.bad;
#pragma unroll
                        for (int t = 0; t < 16; ++t) {
                            const uint32_t e = eq_lo(cur.w[t >> 2], t & 3);
                            M = BS_BITOP3(bs_dbl(M), g.start, e, 0xA8u);           // ((M << 1) | START) & eq
                            band = bs_dbl(band);
                            const uint32_t h = M & g.end;
                            bad = BS_BITOP3(h, band, bad, 0xBAu);                  // (h & ~band) | bad
                        }
                        pre.M = M; pre.band = band; pre.bad = bad;
                    } else {
#pragma unroll
                        for (int t = 0; t < 16; ++t)
                            bs3_pre_step(pre, eq_lo(cur.w[t >> 2], t & 3), pos + t + 1, g, p.kacc, true);
                    }
                    // (whether the chunk's occurrences count is decided with what the lane knows at the chunk's end)
                    if (!(active && c < bs3_pre_chunks(p0, n, pre.found != 0, pre.s1, range))) pre.bad = bad_before;
                    pos += 16;
                    cur = nxt;
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
                Chunk cur = load_chunk(q, pos, n, done ? 0 : n);
                for (;;) {
                    if (!__any(!done && j < jend)) break;
                    const Chunk nxt = load_chunk(q, pos + 16, n, (!done && j + 16 < jend) ? n : 0);
                    bad_chars |= cur.w[0] | cur.w[1] | cur.w[2] | cur.w[3];
                    uint64_t eqq[2];
                    eqq[0] = eq_of(cur, 0); eqq[1] = eq_of(cur, 1);
                    if (FAST && __all(done || j + 16 <= jend)) {
                        // lanes that are through step along on NUL chunks; their state is not looked at again
#pragma unroll
                        for (int t = 0; t < 16; ++t) {
                            const uint64_t eq = eqq[t & 1];
                            if (t + 2 < 16) eqq[t & 1] = eq_of(cur, t + 2);
                            ++j;
                            if (bs32_step<true, XR>(S, (uint32_t)eq, (uint32_t)(eq >> 32), j, p, jlim) && !ex) { ex = true; exj = j; }
                        }
                        if (ex) done = true;
                    } else {
#pragma unroll
                        for (int t = 0; t < 16; ++t) {
                            const uint64_t eq = eqq[t & 1];
                            if (t + 2 < 16) eqq[t & 1] = eq_of(cur, t + 2);
                            if (!done && j < jend) {
                                ++j;
                                if (bs32_step<true, XR>(S, (uint32_t)eq, (uint32_t)(eq >> 32), j, p, jlim)) { ex = true; exj = j; done = true; }
                            }
                        }
                    }
                    if (j >= jend) done = true;
                    pos += 16;
                    cur = nxt;
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
