// multi2_model.cpp -- HOST MODEL of the streaming multi-adapter path (k_multi_stream + k_multi_scan, multi2.hip).
// TEST INFRASTRUCTURE ONLY: the product never loads this file.
//
// Compiles the product's own table builder and rules (cutadapt_amd/csrc/multi2.h) and classification
// (back_scan.h) with g++ and replays, read by read and sequentially, what the kernels do: the main pass over the
// read (class W k-mers), the tail classes' event passes (hi, lo, E0) in this order, `seen` / "again" / `first`, the
// error-free suffix compare, and for every pair the cost scan on the pair's window followed by the shortcut or the
// windowed cell DP.  tests/test_multi2_model.py compares the merged result (MultipleAdapters' order) with the
// oracle's kmers_present + locate on the whole read.
#include "back_model.cpp"

#include "../../cutadapt_amd/csrc/multi2.h"

// (kernels.h needs the HIP runtime; the three things of it the model uses, restated)
#define CAH_KEY_SHIFT 2
#define CAH_QUEUE_BINS 256
static inline unsigned long long pack_best(int score, int errors, int adapter, int ref_stop, int query_start, int query_stop) {
    return ((unsigned long long)(unsigned)(score + 128) << 54) | ((unsigned long long)(unsigned)(127 - errors) << 47) |
           ((unsigned long long)(unsigned)(4095 - adapter) << 35) | ((unsigned long long)(unsigned)ref_stop << 28) |
           ((unsigned long long)(unsigned)query_start << 8) | (unsigned long long)(unsigned)(query_stop - query_start);
}

namespace {

// developer statistics (M2M_PASS_STATS=1): per event pass (8 = class W, 9 = the fixed E0 look-up) events / events whose
// k-mer is an entry's / ... inside that entry's window
long long g_pass[10][3];
int g_cur_pass = 8;

struct Pair { int adapter; unsigned flags; int key; int cls = M2_W; };   // cls: the class the pair was emitted in (the kernel: its page's)

struct ReadState {
    bool seen[128], wideonly[128];
    int first = -1;
    std::vector<Pair> pairs;
    std::vector<std::pair<int, int>> exact;      // (adapter, overlap) decided by the suffix compare
    int conservative = 0;
    int events[4] = {0, 0, 0, 0};
};

void resolve(const M2Tables& t, ReadState& st, uint32_t r, int qc, int cls, int p, int n, uint32_t rlast) {
    const CahMulti2Header& h = t.hdr;
    const uint32_t home = m2_index(r, qc) & (CAH_M2_SLOTS - 1);
    const uint32_t d = t.dir[home];
    for (int u = m2_dir_begin(d);; u++) {
        // (a home's count saturates at CAH_M2_MAX_GROUP: such a home is walked while the entries are its own)
        if (u >= m2_dir_begin(d) + m2_dir_count(d) &&
            (m2_dir_count(d) < CAH_M2_MAX_GROUP || u >= (int)t.entries.size() || m2_home_of(t.entries[(size_t)u].key, t.entries[(size_t)u].meta) != home)) break;
        const CahM2Slot& e = t.entries[(size_t)u];
        const uint32_t meta = e.meta;
        const int q = m2_q(meta);
        if (m2_cls(meta) != cls || std::min(q, 8) != qc || (r & m2_mask2(q)) != e.key) continue;
        g_pass[g_cur_pass][1]++;
        const int dist = n - (p - q + 1);
        const int a = m2_adapter(meta);
        if (p - q + 1 < 0 || !m2_in_window(meta, dist)) continue;
        g_pass[g_cur_pass][2]++;
        // (a further hit of a pair that exists: the kernel sets the pair's "again" bit -- seen & again = flagged)
        if (st.seen[a]) { st.wideonly[a] = true; continue; }
        st.seen[a] = true;
        if (cls == M2_W && m2_precise_chunk(meta))
            st.pairs.push_back({a, CAH_M2_PAIR_PRECISE | ((m2_precise_chunk(meta) - 1u) << CAH_M2_PAIR_CHUNK_SHIFT), p});
        else if (cls == M2_W) st.pairs.push_back({a, 0u, std::min((p & ~15) >> CAH_KEY_SHIFT, CAH_QUEUE_BINS - 1)});
        else if (cls == M2_HI) st.pairs.push_back({a, CAH_M2_PAIR_TAIL, std::max(0, n - h.win_dist[M2_HI]) >> 2, M2_HI});
        else if (cls == M2_LO) st.pairs.push_back({a, CAH_M2_PAIR_TAIL, std::max(0, n - h.win_dist[M2_LO]) >> 2, M2_LO});
        else {
            const int i = m2_exact_tail(rlast, t.prefix[(size_t)a], h.min_overlap, h.lmax0, n);
            if (i > 0) st.exact.push_back({a, i});
        }
    }
}

void filter_read(const M2Tables& t, const uint8_t* q, int n, ReadState& st, int pad = 0) {
    const CahMulti2Header& h = t.hdr;
    memset(st.seen, 0, sizeof(st.seen));
    memset(st.wideonly, 0, sizeof(st.wideonly));
    // the words the tables are looked up with: two bits per character (multi2.h: m2_roll2), rolled exactly as the kernel rolls
    // them; the 3-bit word of the read's last ten characters for the error-free overlaps
    std::vector<uint32_t> rr((size_t)n + 1);
    uint32_t r = 0u, r3 = 0x24924924u;                                // (r3: ten invalid characters)
    for (int p = 0; p < n; p++) { r = m2_roll2(r, q[p]); rr[(size_t)p] = r; r3 = (r3 << 3) | m2_code(q[p]); }
    const uint32_t rlast = r3;
    const uint32_t rlast2 = n > 0 ? rr[(size_t)n - 1] : 0u;
    auto probe = [&](int p, int qc, int cls) {
        const uint32_t bit = m2_bit(rr[(size_t)p], qc, cls);
        return ((t.bitmap[bit >> 5] >> (bit & 31)) & 1u) != 0;
    };
    // class W: every position, every index class of the class
    for (int p = 0; p < n; p++)
        for (int qc = 1; qc <= 8; qc++) {
            if (!((h.q_mask[M2_W] >> qc) & 1) || !probe(p, qc, M2_W)) continue;
            if (st.first < 0) st.first = p & ~15;
            st.events[M2_W]++;
            g_cur_pass = 8; g_pass[8][0]++;
            resolve(t, st, rr[(size_t)p], qc, M2_W, p, n, rlast);
        }
    // the tail classes: the event passes in class order, each over the positions its window opens (the kernel probes a
    // mask per index class over the union of its passes' windows and hands every pass the hits inside its own)
    for (int j = 0; j < h.tq_n; j++) {
        const int cls = h.tq_cls[j], qc = h.tq_qc[j];
        const int qx = qc < 8 ? qc : CAH_M2_MAXQ;                      // the longest k-mer of the index class
        const int plo = std::max(0, n + qc - 1 - h.tq_open[j]), phi = std::min(n - 1, n + qx - 1 - h.tq_close[j]);
        for (int p = plo; p <= phi; p++) {
            if (!probe(p, qc, cls)) continue;
            st.events[cls]++;
            g_cur_pass = j; g_pass[j][0]++;
            resolve(t, st, rr[(size_t)p], qc, cls, p, n, rlast);
        }
    }
    // error-free overlaps of q <= 4 characters: every read looks its own end up in the table of first adapters
    // (multi2.h: CAH_M2_FIXED_WORD; the kernel's skip = pad: the NULs in front of a view are not characters of the read)
    const uint8_t* const fixed_tab = reinterpret_cast<const uint8_t*>(t.bitmap.data() + CAH_M2_FIXED_WORD);
    for (int qq = 1; qq <= CAH_M2_FIXED_MAXQ; qq++) {
        if (!((h.qm_fixed >> qq) & 1) || qq > n - pad || (rlast & m2_mask(qq) & 0x24924924u)) continue;
        unsigned a = fixed_tab[m2_fixed_off(qq) + (rlast2 & m2_mask2(qq))];
        while (a != 0xFFu && st.seen[a]) a = fixed_tab[m2_fixed_next(qq) + a];      // (a pair that exists: the scan's business)
        if (a == 0xFFu) continue;
        g_cur_pass = 9; g_pass[9][0]++;
        st.events[M2_SHORT]++;
        st.exact.push_back({(int)a, qq});
    }
}

}  // namespace

extern "C" {

// adapters: A strings of m characters back to back.  blobs: A CahMatcher structs (cah_plan_debug_matcher).
// ref_*: the reference search sets, flattened: adapter index, window (255 = whole read, else L of (-L, None)), k-mer
// (NUL-terminated strings back to back).  subs: 1 = the scan keeps the SUBS_FULL / INDEL1_FULL bookkeeping.
// stats (may be NULL): [0] pairs W, [1] pairs hi, [2] pairs lo, [3] suffix compares that matched, [4] (unused since round 6),
// [5] scan columns, [6] pairs to the cell DP, [7] whole-read pairs scanned on the window of their one occurrence.
// Returns 0; 1 when the tables cannot be built (the plan would take the older path).
int m2m_match_batch(const char* adapters, int A, int m, const void* blobs, int32_t n_ref, const int32_t* ref_adapter,
                    const int32_t* ref_window, const char* ref_kmers, const uint8_t* seqs, const int64_t* offsets,
                    int64_t n_reads, int32_t* out6, uint8_t* status, int32_t* best, int subs, int64_t* stats,
                    const int32_t* pads) {
    std::vector<std::string> ads;
    for (int a = 0; a < A; a++) ads.push_back(std::string(adapters + (size_t)a * m, (size_t)m));
    std::vector<CahMatcher> mts((size_t)A);
    memcpy(mts.data(), blobs, sizeof(CahMatcher) * (size_t)A);
    std::vector<std::vector<M2RefKmer>> ref((size_t)A);
    const char* kp = ref_kmers;
    for (int i = 0; i < n_ref; i++) {
        ref[(size_t)ref_adapter[i]].push_back({std::string(kp), ref_window[i]});
        kp += strlen(kp) + 1;
    }
    M2Tables t;
    const CahMatcher& m0 = mts[0];
    if (!m2_build(ads, m0.thr_last, m0.kacc, m0.k, m0.min_overlap, ref, t)) return 1;
    const bool pass_stats = getenv("M2M_PASS_STATS") != nullptr;
    if (pass_stats) {
        memset(g_pass, 0, sizeof(g_pass));
        const CahMulti2Header& h = t.hdr;
        fprintf(stderr, "entries %d, passes %d\n", (int)t.entries.size(), h.tq_n);
        int cnt[4][16] = {};
        for (const CahM2Slot& e : t.entries) cnt[m2_cls(e.meta)][m2_q(e.meta)]++;
        for (int c = 0; c < 4; c++) for (int q = 0; q < 16; q++) if (cnt[c][q]) fprintf(stderr, "  class %d q %d: %d entries\n", c, q, cnt[c][q]);
    }
    BackScanParams p;
    p.m = m0.m; p.k = m0.k; p.kacc = m0.kacc; p.min_overlap = m0.min_overlap; p.half_m = m0.m / 2;
    const int kind = bs_kind_of(p.m);
    const int reach = p.m + p.k + 1;
    for (int64_t r = 0; r < n_reads; r++) {
        // pads (may be NULL): read r is a VIEW streamed END-ALIGNED in a frame of pads[r] + its length characters (what the
        // kernels do with the views of a uniform batch: pads[r] NULs in front, multi2.hip): the prefilter and the scan work on
        // the padded frame (q, n), the cell DP and every reported coordinate on the view itself (qv, nv); a shortcut of the
        // scan whose alignment would begin inside the pad is no shortcut -- the pair takes the cell DP on the view
        const uint8_t* const qv = seqs + offsets[r];
        const int nv = (int)(offsets[r + 1] - offsets[r]);
        const int pad = pads ? pads[r] : 0;
        std::vector<uint8_t> padded;
        const uint8_t* q = qv;
        int n = nv;
        if (pad > 0) {
            padded.assign((size_t)(pad + nv), 0);
            memcpy(padded.data() + pad, qv, (size_t)nv);
            q = padded.data(); n = pad + nv;
        }
        int32_t* o = out6 + r * 6;
        for (int i = 0; i < 6; i++) o[i] = 0;
        status[r] = 0; best[r] = -1;
        bool invalid = false;
        for (int i = 0; i < n; i++) invalid = invalid || q[i] >= 0x80;
        if (invalid) { status[r] = 2; continue; }
        ReadState st;
        filter_read(t, q, n, st, pad);
        unsigned long long bestkey = 0;
        for (auto& ex : st.exact) {
            bestkey = std::max(bestkey, pack_best(ex.second, 0, ex.first, ex.second, nv - ex.second, nv));
            if (stats) stats[3]++;
        }
        if (stats) { stats[4] += st.conservative; for (int c = 0; c < 4; c++) stats[8 + c] += st.events[c]; }
        if (pass_stats && r == n_reads - 1) {
            const CahMulti2Header& h = t.hdr;
            for (int j = 0; j < 10; j++) {
                if (!g_pass[j][0]) continue;
                if (j < 8) fprintf(stderr, "  pass %d (class %d, qc %d, dist %d..%d):", j, h.tq_cls[j], h.tq_qc[j], h.tq_close[j], h.tq_open[j]);
                else fprintf(stderr, j == 8 ? "  class W:" : "  fixed E0:");
                fprintf(stderr, " %.3f events per read, %.3f with an entry's k-mer, %.3f inside its window\n", g_pass[j][0] / (double)n_reads,
                        g_pass[j][1] / (double)n_reads, g_pass[j][2] / (double)n_reads);
            }
        }
        // the byte k_multi_stream leaves per read: the adapter whose pair saw a further hit
        unsigned wm = CAH_M2_NO_FLAG;
        for (int a = 0; a < A; a++)
            if (st.seen[a] && st.wideonly[a]) wm = wm == CAH_M2_NO_FLAG ? (unsigned)a : CAH_M2_MANY_FLAGS;
        uint32_t rlast = 0x24924924u;
        for (int i = 0; i < n; i++) rlast = (rlast << 3) | m2_code(q[i]);
        for (const Pair& pr : st.pairs) {
            const CahMatcher& mt = mts[(size_t)pr.adapter];
            const bool tail = (pr.flags & CAH_M2_PAIR_TAIL) != 0;
            int j0 = tail ? (pr.key << 2) : std::max(0, (pr.key << CAH_KEY_SHIFT) - p.m - p.k - 1);
            // a PRECISE pair whose occurrence stayed its only hit: the window around that occurrence (multi2.h)
            static const bool no_precise = getenv("M2M_NO_PRECISE") != nullptr;     // (for the column statistics: the windows before)
            const bool precise = !no_precise && (pr.flags & CAH_M2_PAIR_PRECISE) != 0 && wm != (unsigned)pr.adapter && wm != CAH_M2_MANY_FLAGS;
            int jend = n, tail0 = 0;
            if (pr.flags & CAH_M2_PAIR_PRECISE) {
                int jw, jb;
                m2_precise_window(pr.key, (int)(pr.flags >> CAH_M2_PAIR_CHUNK_SHIFT) & 3, p.m, p.k, p.m / (p.k + 1), p.m % (p.k + 1), n, jw, jb);
                if (precise) {
                    j0 = jw; jend = jb;
                    tail0 = m2_exact_tail(rlast, t.prefix[(size_t)pr.adapter], p.min_overlap, t.hdr.lmax0, n);
                } else {
                    j0 = std::max(0, (pr.key & ~15) - p.m - p.k - 1);   // (flagged: the window of a whole-read pair)
                }
            }
            // rows above this cannot be acceptable (k_multi_scan does not look at them)
            int max_row = std::min(p.m, n - std::min(j0, n) + p.kacc);
            const bool is_lo = tail && pr.cls == M2_LO;
            if (is_lo) max_row = std::min(max_row, t.hdr.rows_lo);
            j0 = bs_align_window(std::min(j0, n), n);
            if (precise) {
                // the kernel stops at the first chunk boundary behind the window (later if another lane of the wave needs
                // more: M2M_EXTEND chunks more here, to show that it does not matter)
                static const int extend = getenv("M2M_EXTEND") ? atoi(getenv("M2M_EXTEND")) : 0;
                jend = std::min(n, j0 + ((jend - j0 + 15) & ~15) + 16 * extend);
                if (stats) stats[7]++;
            }
            if (stats) { stats[tail ? (is_lo ? 2 : 1) : 0]++; stats[5] += jend - j0; }
            uint64_t tab32[128];
            for (int c = 0; c < 128; c++) tab32[c] = bs32_table_entry(mt.scanmask[c], p.m);
            bool exact = false;
            int j = j0, o0 = 0, o1 = 0, cls = BS_NONE, jfa = -1;
            auto thr = [&](int i) { return mt.thr_last[i]; };
            auto run = [&](auto& s, auto step, auto finish, auto finish_stopped) {
                while (j < jend) {
                    ++j;
                    if (step(s, q[j - 1] & 127)) { exact = true; break; }
                }
                jfa = s.jfa;
                if (exact) { cls = BS_EXACT_FULL; o0 = j; }
                else if (!precise) cls = finish(s);
                else if (jfa < 0) cls = BS_NONE;                       // no candidate in the window: there is none at all
                else {
                    // nothing behind the window matters: the state is that of an inner column ("stopped")
                    cls = finish_stopped(s);
                    // ... unless an error-free overlap is acceptable too: the cell DP sorts that out, to the read's end
                    if (tail0 > 0) { cls = BS_DP; o0 = std::max(j0, jfa - reach); o1 = 2 * n + 1; }
                }
            };
#define M2M_RUN32(X)                                                                                                      \
            {                                                                                                             \
                BackScanState32<X> s;                                                                                     \
                bs32_init(s, p);                                                                                          \
                if (subs) run(s, [&](BackScanState32<X>& z, int c) { return bs32_step<true, X>(z, (uint32_t)tab32[c], (uint32_t)(tab32[c] >> 32), j, p); }, \
                              [&](BackScanState32<X>& z) { return bs32_finish<X, true>(z, n, j0, p, thr, o0, o1, false, max_row); }, \
                              [&](BackScanState32<X>& z) { return bs32_finish<X, true>(z, n, j0, p, thr, o0, o1, true, max_row); }); \
                else if (tail) run(s, [&](BackScanState32<X>& z, int c) { return bs32_step<false, X, false>(z, (uint32_t)tab32[c], (uint32_t)(tab32[c] >> 32), j, p); }, \
                         [&](BackScanState32<X>& z) { return bs32_finish<X, false>(z, n, j0, p, thr, o0, o1, false, max_row); },  \
                         [&](BackScanState32<X>& z) { return bs32_finish<X, false>(z, n, j0, p, thr, o0, o1, true, max_row); });  \
                else run(s, [&](BackScanState32<X>& z, int c) { return bs32_step<false, X>(z, (uint32_t)tab32[c], (uint32_t)(tab32[c] >> 32), j, p); }, \
                         [&](BackScanState32<X>& z) { return bs32_finish<X, false>(z, n, j0, p, thr, o0, o1, false, max_row); },  \
                         [&](BackScanState32<X>& z) { return bs32_finish<X, false>(z, n, j0, p, thr, o0, o1, true, max_row); });  \
            }
            if (!subs && is_lo && kind != 1 && t.hdr.rows_lo <= 32) {
                // k_multi_scan's "lo" pages: the adapter's first 32 rows in one plain 32-bit word
                BackScanParams p32 = p;
                p32.m = 32;
                BackScanState32<0> s;
                bs32_init(s, p32);
                const int xr = kind >= 2 ? kind - 1 : 0;
                run(s, [&](BackScanState32<0>& z, int c) {
                        const uint32_t rows = kind == 0 ? (uint32_t)(mt.scanmask[c] >> (64 - p.m))
                                                        : (((uint32_t)tab32[c] << xr) | ((uint32_t)(tab32[c] >> 32) & ((1u << xr) - 1u)));
                        return bs32_step<false, 0, false>(z, rows, 0u, j, p32);
                    },
                    [&](BackScanState32<0>& z) { return bs32_finish<0, false>(z, n, j0, p32, thr, o0, o1, false, max_row); },
                    [&](BackScanState32<0>& z) { return bs32_finish<0, false>(z, n, j0, p32, thr, o0, o1, true, max_row); });
            } else if (kind == 0) {
                BackScanState s;
                bs_init(s, p);
                if (subs) run(s, [&](BackScanState& z, int c) { return bs_step<true>(z, mt.scanmask[c], j, p); },
                              [&](BackScanState& z) { return bs_finish<true>(z, n, j0, p, thr, o0, o1, false, max_row); },
                              [&](BackScanState& z) { return bs_finish<true>(z, n, j0, p, thr, o0, o1, true, max_row); });
                else if (tail) run(s, [&](BackScanState& z, int c) { return bs_step<false, false>(z, mt.scanmask[c], j, p); },
                         [&](BackScanState& z) { return bs_finish<false>(z, n, j0, p, thr, o0, o1, false, max_row); },
                         [&](BackScanState& z) { return bs_finish<false>(z, n, j0, p, thr, o0, o1, true, max_row); });
                else run(s, [&](BackScanState& z, int c) { return bs_step<false>(z, mt.scanmask[c], j, p); },
                         [&](BackScanState& z) { return bs_finish<false>(z, n, j0, p, thr, o0, o1, false, max_row); },
                         [&](BackScanState& z) { return bs_finish<false>(z, n, j0, p, thr, o0, o1, true, max_row); });
            } else if (kind == 1) M2M_RUN32(0)
            else if (kind == 2) M2M_RUN32(1)
            else M2M_RUN32(2)
#undef M2M_RUN32
            unsigned long long key = 0;
            bool in_pad = false;
            // a shortcut's tuple, in the coordinates of the view
            auto shortcut = [&](int score, int errors, int ref_stop, int qs, int qe) {
                if (qs < pad) { in_pad = true; return; }
                key = pack_best(score, errors, pr.adapter, ref_stop, qs - pad, qe - pad);
            };
            if (cls == BS_EXACT_FULL) shortcut(p.m, 0, p.m, o0 - p.m, o0);
            else if (cls == BS_EXACT_TAIL) shortcut(o0 - 2 * o1, o1, o0, n - o0, n);
            else if (cls == BS_SUBS_FULL) shortcut(p.m - 2 * o1, o1, p.m, o0 - p.m, o0);
            else if (cls == BS_INDEL1_FULL)
                shortcut(p.m - 2 * (o1 >> 1) - (o1 & 1), o1 >> 1, p.m, o0 - p.m + ((o1 & 1) ? 1 : -1), o0);
            if (in_pad) { cls = BS_DP; o0 = 0; o1 = 2 * n + 1; }      // (the view from its first column, every row of its last)
            if (cls == BS_DP) {
                // a tail pair's cell DP runs over the FULL reach (the band, last_filled and the stale origin of the
                // final scan are only proven equal to the reference's from column start + m + k + 1 on)
                if (tail && !in_pad) o0 = std::max(0, (jfa >= 0 ? jfa : n) - reach);
                int t6[6];
                if (stats) { stats[6]++; stats[12 + (tail ? (is_lo ? 2 : 1) : (precise ? 3 : 0))]++; }
                // (the cell DP sees the VIEW: the window's columns moved by the pad)
                const int first = std::max(0, o0 - pad), last = std::max(0, std::min(n, o1 >> 1) - pad);
                if (dp_window(mt, qv, nv, std::min(first, nv), std::min(last, nv), (o1 & 1) != 0, t6)) {
                    // the corner of multi2.h's head: a tail pair's match that reaches further back than its error class's
                    // last row -- is the reference's kmers_present true for the pair?
                    bool ref_ok = true;
                    static const bool no_refcheck = getenv("M2M_NO_REFCHECK") != nullptr;     // (to show that the tests need the check)
                    if (tail && !no_refcheck && nv - t6[2] > (int)t.hdr.lmax_row[t6[1]]) {
                        ref_ok = m2_ref_present(t.ref_list.data(), t.ref_begin[(size_t)pr.adapter], t.ref_begin[(size_t)pr.adapter + 1], qv, nv, t.hdr.ref_span);
                        if (stats) stats[15]++;
                    }
                    if (ref_ok) key = pack_best(t6[4], t6[5], pr.adapter, t6[1], t6[2], t6[3]);
                }
            }
            // the error-free overlap of a pair without a candidate in its window
            if (precise && cls == BS_NONE && tail0 > 0) key = pack_best(tail0, 0, pr.adapter, tail0, nv - tail0, nv);
            bestkey = std::max(bestkey, key);
        }
        if (bestkey) {
            const unsigned long long k = bestkey;
            const int rel = (int)(k & 0xFFu), qstart = (int)((k >> 8) & 0xFFFFFu), ref_stop = (int)((k >> 28) & 0x7Fu);
            o[0] = 0; o[1] = ref_stop; o[2] = qstart; o[3] = qstart + rel;
            o[4] = (int)((k >> 54) & 0xFFu) - 128; o[5] = 127 - (int)((k >> 47) & 0x7Fu);
            best[r] = 4095 - (int)((k >> 35) & 0xFFFu);
            status[r] = 1;
        }
    }
    return 0;
}

}  // extern "C"
