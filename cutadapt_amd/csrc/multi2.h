// multi2.h -- tables and rules of the streaming multi-adapter prefilter (k_multi_stream, multi2.hip), in a form the
// plan builder (api.cpp), the kernel AND a plain C++ host model (tests/host_model/multi2_model.cpp, g++) share.
//
// What the path computes: MultipleAdapters.match_to over many 3' adapters of one shape (reference
// src/cutadapt/adapters.py:1265-1286) = for every adapter `kmers_present(read)` (reference _kmer_finder.pyx:170-257,
// search sets of kmer_heuristic.py:87-164) and, where true, `Aligner.locate(read)` (_align.pyx:298-587).
//
// Two families of k-mers per adapter live in ONE table:
//   REF   the k-mers of the reference's own search sets with their windows: a (read, adapter) pair exists iff one of
//         them occurs inside its window -- exactly kmers_present, nothing more, nothing less;
//   WIDE  our own lossless family, built from the aligner's thresholds alone (pigeonhole): rows of the last column
//         whose threshold is e (overlap lengths Lmin_e .. Lmax_e) can only be acceptable (_align.pyx:536-572) if one
//         of the e + 1 consecutive chunks of adapter[0:Lmin_e] occurs unedited at a start position >= n - Lmax_e - e
//         (the reference's window carries no slack for insertions: n - Lmax_e); a last-row candidate (:496-533) only
//         if one of the k + 1 chunks of the whole adapter occurs anywhere.  An alignment that contains chunk c
//         (adapter offset o_c) unedited at read position s starts at column >= s - o_c - e.
//         With the reference's own chunking (kmer_heuristic.kmer_chunks) the two families share their k-mers; only
//         the tail windows differ by the e "margin" positions.
// The WIDE family decides how much of `locate` a pair needs:
//   class W   some whole-adapter chunk occurs: the cost scan runs from (first hit) - m - k - 1 to the read end;
//   class hi  no whole-adapter chunk, but a chunk of a tail class with e >= 2: only rows of the last column can be
//             acceptable and every optimal path to one lies in the last max(Lmax_e + e) columns: the scan starts there;
//   class lo  only chunks of the tail class e = 1: the same with that class's reach;
//   none      no WIDE k-mer at all: only rows without error tolerance (overlaps min_overlap .. Lmax_0) can match and
//             they match exactly or not at all: the prefilter itself compares the read's suffix with the adapter's
//             prefix and writes the result -- the pair never reaches the scan.
// The classes are probed in this order (W over the whole read, then hi, lo and the REF-only k-mers over the read's
// tail), and a pair is emitted at its FIRST REF hit, with the class being probed then: a WIDE hit of a wider class
// that was a REF hit too would have emitted the pair earlier.  A WIDE k-mer that is not a REF k-mer where it occurs (the
// margins, or a plan whose search sets were not built by kmer_heuristic) says nothing about kmers_present; it sets the
// pair's "wide only" bit, and a pair whose bit is set when its first REF hit arrives takes the whole read (class W
// from column 0): exact whatever the order, just slower -- and rare.
//
// Characters: 3 bits each (A 0, C 1, G 2, T 3, either case; anything else 4 = breaks every k-mer, as KmerFinder
// without wildcards does: _match_tables.py:81-98), the newest character in the lowest bits; ten characters per
// 32-bit word -- k-mers of up to CAH_M2_MAXQ characters.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define M2_HD __host__ __device__ __forceinline__
#else
#define M2_HD inline
#endif

#define CAH_M2_SLOTS 4096            // directory slots: home = low 12 bits of the bitmap index -> (first entry, entries)
#define CAH_M2_MAX_ENTRIES 2048
#define CAH_M2_MAX_GROUP 15          // entries that may share a home
#define CAH_M2_BM_WORDS 3072         // presence bitmaps: 64 Kbit for the index class 8 (probed at every character), then
#define CAH_M2_BM8_WORDS 2048        // 32 Kbit shared by the shorter classes (probed in the tail sweeps only)
#define CAH_M2_MAXQ 10
#define CAH_M2_EMPTY 0xFFFFFFFFu
#define CAH_M2_WHOLE 255             // window value: the whole read
#define CAH_M2_NEVER 1000

// classes of the k-mer table (the class a k-mer is PROBED in)
enum { M2_W = 0, M2_HI = 1, M2_LO = 2, M2_SHORT = 3 };
// pair flags (bits 24.. of a pair record: read << 32 | flags << 24 | adapter << 8 | key)
#define CAH_M2_PAIR_TAIL 1u          // the scan window starts at column 4 * key; only rows of the last column can match
// A whole-read pair whose first hit is chunk c of the adapter's k + 1 chunks, ending at read position key (0-based; the
// chunk index sits in flag bits 2..3).  If that occurrence stays the pair's ONLY k-mer hit (no second whole-read hit, no
// hit of a tail class: the prefilter leaves the read's flagged adapter in a byte per read, CAH_M2_NO_FLAG = none,
// CAH_M2_MANY_FLAGS = more than one), every last-row candidate (_align.pyx:496-533, cost <= k over all m rows) holds THIS
// occurrence as its chunk c, so its alignment lies in columns [f - E_c - k, f + (m - E_c) + k] (f = key + 1, E_c = end
// offset of chunk c in the adapter): the scan runs from f - E_c - k - 1 to that end, no further candidate follows, and
// of the last column's rows only the error-free overlaps can be acceptable (a row with tolerance e >= 1 needs a hit of a
// tail class) -- the suffix compare m2_exact_tail decides those.
// The per-read word: bits 0..7 the flagged adapter (or CAH_M2_NO_FLAG / CAH_M2_MANY_FLAGS), bits 8..11 the 16-column chunk of
// the read's earliest FURTHER whole-read hit (15: none) -- a pair that does not stay alone falls back to the full window from
// the earliest of its own position and that chunk (the lane that emitted it need not hold its earliest occurrence).
#define CAH_M2_PAIR_PRECISE 2u
#define CAH_M2_PAIR_CHUNK_SHIFT 2
#define CAH_M2_NO_FLAG 255u
#define CAH_M2_MANY_FLAGS 254u
// Pairs leave the prefilter in PAGES of one class each (0 lo, 1 hi, 2..5 whole-read pairs by window length), taken from
// a device-wide pool by the wave that fills them; page_hdr[page] = class << 24 | pairs in it.
#define CAH_M2_PAGE 1024
#define CAH_M2_PAIR_CLASSES 6

struct CahM2Slot { uint32_t key, meta; };      // one entry = one (k-mer, adapter); entries of a home are consecutive
M2_HD uint32_t m2_dir(int begin, int count) { return (uint32_t)begin | ((uint32_t)count << 12); }
M2_HD int m2_dir_begin(uint32_t d) { return (int)(d & 0xFFFu); }
M2_HD int m2_dir_count(uint32_t d) { return (int)(d >> 12); }
// meta: adapter : 8 | q : 4 @8 | cls : 2 @12 | ref_L : 8 @14 | wide_L : 8 @22
M2_HD uint32_t m2_meta(int adapter, int q, int cls, int ref_L, int wide_L) {
    return (uint32_t)adapter | ((uint32_t)q << 8) | ((uint32_t)cls << 12) | ((uint32_t)ref_L << 14) | ((uint32_t)wide_L << 22);
}
M2_HD int m2_adapter(uint32_t meta) { return (int)(meta & 255u); }
M2_HD int m2_q(uint32_t meta) { return (int)((meta >> 8) & 15u); }
M2_HD int m2_cls(uint32_t meta) { return (int)((meta >> 12) & 3u); }
M2_HD int m2_ref_L(uint32_t meta) { return (int)((meta >> 14) & 255u); }
M2_HD int m2_wide_L(uint32_t meta) { return (int)((meta >> 22) & 255u); }
// A class-W entry whose k-mer is chunk c -- and only that -- of the adapter's k + 1 <= 4 whole-adapter chunks carries
// CAH_M2_WHOLE - 4 + c as its WIDE window (as good as "the whole read": no read is that long); 1 + c, or 0
#define CAH_M2_WHOLE_CHUNK0 251
M2_HD unsigned m2_precise_chunk(uint32_t meta) {
    const unsigned w = (meta >> 22) & 255u;
    return (((meta >> 12) & 3u) == 0u && w >= CAH_M2_WHOLE_CHUNK0 && w < 255u) ? w - (CAH_M2_WHOLE_CHUNK0 - 1) : 0u;
}

// end offset (exclusive) of chunk c when a string of m characters is cut into `chunks` pieces the way m2_chunks does
M2_HD int m2_chunk_end(int base, int extra, int c) { return (c + 1) * base + ((c + 1) < extra ? (c + 1) : extra); }
// the window of a PRECISE pair: first column in front of it (the scan starts behind j0w) and the last column a last-row
// candidate can end in.  chunk_base / chunk_extra: m / (k + 1), m % (k + 1) (a division the kernels do once)
M2_HD void m2_precise_window(int key, int chunk, int m, int k, int chunk_base, int chunk_extra, int n, int& j0w, int& jb) {
    const int f = key + 1, E = m2_chunk_end(chunk_base, chunk_extra, chunk);
    j0w = f - E - k - 1;
    if (j0w < 0) j0w = 0;
    jb = f + (m - E) + k;
    if (jb > n) jb = n;
}

M2_HD uint32_t m2_code(unsigned c) {
    const unsigned u = c & 0xDFu;
    return (c >= 64 && c < 128) ? (u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : u == 'T' ? 3u : 4u) : 4u;
}
M2_HD uint32_t m2_mask(int q) { return q >= 10 ? 0x3FFFFFFFu : ((1u << (3 * q)) - 1u); }
// index (16 bits) of the last qc = min(q, 8) characters: the home slot of the directory is its low 12 bits, the bitmap
// bit m2_bit(index, qc)
M2_HD uint32_t m2_salt(int qc) { return (uint32_t)(8 - qc) * 0x1D3Bu; }
M2_HD uint32_t m2_index(uint32_t r, int qc) {
    const uint32_t key = r & m2_mask(qc);
    return ((key ^ (key >> 8)) ^ m2_salt(qc)) & 0xFFFFu;
}
M2_HD uint32_t m2_bit(uint32_t idx, int qc) { return qc >= 8 ? idx : CAH_M2_BM8_WORDS * 32u + (idx & 0x7FFFu); }
// does a k-mer of window L (0: none, CAH_M2_WHOLE: whole read) count when it starts `dist` characters before the end?
M2_HD bool m2_in_window(int L, int dist) { return L >= CAH_M2_WHOLE_CHUNK0 || (L != 0 && dist <= L); }

struct CahMulti2Header {
    int32_t ok;
    int32_t n_adapters, m, k, min_overlap;
    int32_t lmax0;                 // the largest overlap length without error tolerance (rows min_overlap .. lmax0; 0: none)
    int32_t q_mask[4];             // per class: bit qc set = k-mers of index class qc = min(q, 8) exist
    int32_t span[4];               // classes 1..3: a k-mer of the class starts at most this many characters before the end
    int32_t open_L[4][9];          // [cls][qc]: the class is probed while dist_min <= open_L (dist_min: as if q == qc)
    int32_t win_dist[4];           // classes hi, lo: a pair's scan window starts at column n - win_dist
    // The tail classes are probed inside the main pass, in the read's last chunks: up to four "slots" = (class, index
    // class) pairs in class order (hi, lo, then the REF-only k-mers unless short_fixed); a plan that needs more does
    // not take the streaming form.
    int32_t tq_n;
    int32_t tq_cls[4], tq_qc[4];
    int32_t tq_open[4];            // a slot is probed at position p while n - p + qc - 1 <= tq_open (open_L of the pair)
    int32_t short_fixed;           // 1: every REF-only tail k-mer must be the read's last q characters (window = its length)
    int32_t rows_lo;               // the longest overlap whose threshold is <= 1 (rows a pair of class lo can match)
    int32_t tail_warm;             // characters in front of a sweep's first probe that must be in the word (max q - 1)
    uint32_t n_entries;
};

#if !defined(CAH_M2_NO_HOST)
#include <algorithm>
#include <map>
#include <string>
#include <vector>

struct M2Tables {
    CahMulti2Header hdr;
    std::vector<uint16_t> dir;             // CAH_M2_SLOTS
    std::vector<CahM2Slot> entries;        // hdr.n_entries, ordered by home
    std::vector<uint32_t> bitmap;          // CAH_M2_BM_WORDS
    std::vector<uint32_t> prefix;          // per adapter: its first 10 characters, 3 bits each, adapter[0] in bits 27..29
};

struct M2RefKmer { std::string kmer; int window; };   // window: CAH_M2_WHOLE, or L of the tail set (-L, None)

inline uint32_t m2_encode(const std::string& s) {
    uint32_t r = 0;
    for (char ch : s) r = (r << 3) | m2_code((unsigned char)ch);
    return r;
}

// the reference's chunking (kmer_heuristic.py:6-21): `chunks` nearly equal consecutive pieces, longer pieces first
inline std::vector<std::string> m2_chunks(const std::string& s, int chunks) {
    std::vector<std::string> out;
    const int base = (int)s.size() / chunks, extra = (int)s.size() % chunks;
    size_t pos = 0;
    for (int i = 0; i < chunks; i++) {
        const size_t len = (size_t)(base + (i < extra ? 1 : 0));
        out.push_back(s.substr(pos, len));
        pos += len;
    }
    return out;
}

// Builds the tables for adapters of ONE shape (length m, thresholds thr_last[0..m] = threshold of row i in the last
// column, kacc = thr of a last-row candidate, min_overlap).  ref[a]: the reference search sets of adapter a.
// Returns false (t.hdr.ok = 0) when the plan does not fit: a k-mer longer than CAH_M2_MAXQ, too many entries,
// non-monotone thresholds, an error-free class longer than a word.
inline bool m2_build(const std::vector<std::string>& adapters, const int32_t* thr_last, int kacc, int k, int min_overlap,
                     const std::vector<std::vector<M2RefKmer>>& ref, M2Tables& t) {
    CahMulti2Header& h = t.hdr;
    h = CahMulti2Header();
    const int A = (int)adapters.size();
    if (A < 1 || A > 128) return false;
    const int m = (int)adapters[0].size();
    if (m < 1 || m > 64 || kacc < 0 || kacc != k) return false;
    for (int i = std::max(min_overlap, 1) + 1; i <= m; i++)
        if (thr_last[i] < thr_last[i - 1] || thr_last[i] > thr_last[i - 1] + 1) return false;
    if (min_overlap >= 1 && min_overlap <= m && thr_last[m] != kacc) return false;
    // error classes of the last column's rows
    struct Tail { int e, lmin, lmax; };
    std::vector<Tail> tails;
    int lmax0 = 0;
    for (int i = std::max(min_overlap, 1); i <= m; i++) {
        const int e = thr_last[i];
        if (e < 0) return false;
        if (e == 0) { lmax0 = i; continue; }
        if (tails.empty() || tails.back().e != e) tails.push_back({e, i, i});
        else tails.back().lmax = i;
    }
    if (lmax0 > CAH_M2_MAXQ) return false;
    h.n_adapters = A; h.m = m; h.k = k; h.min_overlap = min_overlap; h.lmax0 = lmax0;
    h.rows_lo = lmax0;
    for (const Tail& tl : tails) if (tl.e == 1) h.rows_lo = tl.lmax;
    struct Ent { std::string kmer; int adapter; int cls; int ref_L; int wide_L; bool tail_role; };   // tail_role: the k-mer is a tail-class chunk or a REF tail k-mer too
    std::vector<Ent> ents;
    auto find = [&](int a, const std::string& s) -> Ent* {
        for (Ent& e : ents) if (e.adapter == a && e.kmer == s) return &e;
        return nullptr;
    };
    for (int a = 0; a < A; a++) {
        const std::string& ad = adapters[(size_t)a];
        if ((int)ad.size() != m) return false;
        const size_t first = ents.size();
        auto add_wide = [&](const std::string& s, int cls, int wide_L) -> bool {
            if (s.empty() || (int)s.size() > CAH_M2_MAXQ) return false;
            for (size_t i = first; i < ents.size(); i++)
                if (ents[i].kmer == s) {
                    ents[i].cls = std::min(ents[i].cls, cls);
                    ents[i].wide_L = std::max(ents[i].wide_L, wide_L);
                    ents[i].tail_role = ents[i].tail_role || cls != M2_W;
                    return true;
                }
            ents.push_back({s, a, cls, 0, wide_L, cls != M2_W});
            return true;
        };
        // a last-row candidate (cost <= kacc) holds one of the k + 1 chunks of the whole adapter
        if (min_overlap <= m)
            for (const std::string& s : m2_chunks(ad, kacc + 1))
                if (!add_wide(s, M2_W, CAH_M2_WHOLE)) return false;
        for (const Tail& tl : tails) {
            if (tl.lmax + tl.e >= CAH_M2_WHOLE - 16) return false;
            for (const std::string& s : m2_chunks(ad.substr(0, (size_t)tl.lmin), tl.e + 1))
                if (!add_wide(s, tl.e >= 2 ? M2_HI : M2_LO, tl.lmax + tl.e)) return false;
        }
        for (const M2RefKmer& rk : ref[(size_t)a]) {
            if (rk.kmer.empty() || (int)rk.kmer.size() > CAH_M2_MAXQ) return false;
            if (rk.window != CAH_M2_WHOLE && (rk.window < 1 || rk.window >= CAH_M2_WHOLE - 16)) return false;
            Ent* e = nullptr;
            for (size_t i = first; i < ents.size(); i++) if (ents[i].kmer == rk.kmer) e = &ents[i];
            if (!e) { ents.push_back({rk.kmer, a, M2_SHORT, 0, 0, true}); e = &ents.back(); }
            if (rk.window != CAH_M2_WHOLE) e->tail_role = true;
            // (several windows of one k-mer: the widest counts, as the reference's own dedup does, kmer_heuristic.py:29-64)
            if (rk.window == CAH_M2_WHOLE || e->ref_L == CAH_M2_WHOLE) e->ref_L = CAH_M2_WHOLE;
            else e->ref_L = std::max(e->ref_L, rk.window);
            if (e->ref_L == CAH_M2_WHOLE) e->cls = M2_W;        // probed everywhere
        }
    }
    (void)find;
    if (ents.size() > CAH_M2_MAX_ENTRIES) return false;
    h.n_entries = (uint32_t)ents.size();
    t.dir.assign(CAH_M2_SLOTS, 0);
    t.bitmap.assign(CAH_M2_BM_WORDS, 0u);
    for (int c = 0; c < 4; c++)
        for (int q = 0; q < 9; q++) h.open_L[c][q] = -1;
    int max_q = 1;
    h.short_fixed = 1;
    struct Placed { uint32_t home, key, meta; };
    std::vector<Placed> placed;
    for (const Ent& e : ents) {
        const int q = (int)e.kmer.size(), qc = std::min(q, 8);
        const uint32_t code = m2_encode(e.kmer);
        if (code & 0x24924924u) return false;                        // not plain ACGT
        const uint32_t idx = m2_index(code, qc);
        t.bitmap[m2_bit(idx, qc) >> 5] |= 1u << (idx & 31);
        // which whole-adapter chunk is it?  (no answer when the string is two of them, or occurs elsewhere in the adapter
        // too: an alignment could then hold the occurrence at another offset)
        uint8_t wend = 0;
        // ... and no answer when the string is a chunk of a tail class as well: its occurrence near the read's end then
        // stands for rows of the last column too, which the window of one occurrence does not look at
        if (e.cls == M2_W && e.wide_L == CAH_M2_WHOLE && !e.tail_role && kacc + 1 <= 4) {
            const std::string& ad = adapters[(size_t)e.adapter];
            const std::vector<std::string> ch = m2_chunks(ad, kacc + 1);
            int which = -1, times = 0;
            for (int c = 0; c < (int)ch.size(); c++) if (ch[(size_t)c] == e.kmer) { which = c; times++; }
            int occurrences = 0;
            for (size_t at = ad.find(e.kmer); at != std::string::npos; at = ad.find(e.kmer, at + 1)) occurrences++;
            if (times == 1 && occurrences == 1) wend = (uint8_t)(1 + which);
        }
        placed.push_back({idx & (CAH_M2_SLOTS - 1), code,
                          m2_meta(e.adapter, q, e.cls, e.ref_L, wend ? CAH_M2_WHOLE_CHUNK0 + wend - 1 : e.wide_L)});
        h.q_mask[e.cls] |= 1 << qc;
        const int reach = std::max(e.ref_L, e.wide_L);
        h.open_L[e.cls][qc] = std::max(h.open_L[e.cls][qc], reach);
        if (e.cls != M2_W) h.span[e.cls] = std::max(h.span[e.cls], reach);
        max_q = std::max(max_q, q);
        if (e.cls == M2_SHORT && e.ref_L != q) h.short_fixed = 0;
        if (e.wide_L && (e.cls == M2_HI || e.cls == M2_LO)) h.win_dist[e.cls] = std::max(h.win_dist[e.cls], e.wide_L + 1);
    }
    std::stable_sort(placed.begin(), placed.end(), [](const Placed& x, const Placed& y) { return x.home < y.home; });
    t.entries.clear();
    for (size_t i = 0; i < placed.size();) {
        size_t j = i;
        while (j < placed.size() && placed[j].home == placed[i].home) j++;
        if (j - i > CAH_M2_MAX_GROUP) return false;
        t.dir[placed[i].home] = (uint16_t)m2_dir((int)i, (int)(j - i));
        for (size_t u = i; u < j; u++) t.entries.push_back(CahM2Slot{placed[u].key, placed[u].meta});
        i = j;
    }
    // (hi windows must cover lo's: nothing depends on it, but a plan without a hi class still sweeps with one distance)
    h.win_dist[M2_HI] = std::max(h.win_dist[M2_HI], h.win_dist[M2_LO]);
    if (h.win_dist[M2_LO] == 0) h.win_dist[M2_LO] = h.win_dist[M2_HI];
    h.tail_warm = max_q - 1;
    h.tq_n = 0;
    for (int c = M2_HI; c <= M2_SHORT; c++) {
        if (c == M2_SHORT && h.short_fixed) continue;
        for (int q = 1; q <= 8; q++) {
            if (!((h.q_mask[c] >> q) & 1)) continue;
            if (h.tq_n == 4) return false;
            h.tq_cls[h.tq_n] = c; h.tq_qc[h.tq_n] = q; h.tq_open[h.tq_n] = h.open_L[c][q];
            h.tq_n++;
        }
    }
    t.prefix.assign((size_t)A, 0u);
    for (int a = 0; a < A; a++) {
        uint32_t p = 0;
        for (int j = 0; j < 10; j++) p = (p << 3) | (j < m ? m2_code((unsigned char)adapters[(size_t)a][(size_t)j]) : 4u);
        t.prefix[(size_t)a] = p;
    }
    h.ok = 1;
    return true;
}
#endif

// the longest error-free overlap i in [min_overlap, lmax0] with read[n - i:] == adapter[0:i] (0: none); rlast: the
// read's last ten characters (newest lowest), prefix: the adapter's first ten (adapter[0] highest)
M2_HD int m2_exact_tail(uint32_t rlast, uint32_t prefix, int min_overlap, int lmax0, int n) {
    int best = 0;
    for (int i = min_overlap < 1 ? 1 : min_overlap; i <= lmax0 && i <= n && i <= 10; ++i)
        if ((rlast & m2_mask(i)) == (prefix >> (3 * (10 - i)))) best = i;
    return best;
}
