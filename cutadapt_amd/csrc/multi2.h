// multi2.h -- tables and rules of the streaming multi-adapter prefilter (k_multi_stream, multi2.hip), in a form the
// plan builder (api.cpp), the kernel AND a plain C++ host model (tests/host_model/multi2_model.cpp, g++) share.
//
// What the path computes: MultipleAdapters.match_to over many 3' adapters of one shape (reference
// src/cutadapt/adapters.py:1265-1286) = for every adapter `kmers_present(read)` (reference _kmer_finder.pyx:170-257,
// search sets of kmer_heuristic.py:87-164) and, where true, `Aligner.locate(read)` (_align.pyx:298-587).
//
// Round 6: the table holds OUR lossless family only (WIDE), built from the aligner's thresholds alone (pigeonhole);
// what the reference's own search sets (REF) add -- a pair exists iff kmers_present -- is PROVEN at plan time to follow
// from a WIDE hit for every match the aligner can report, except in one corner that the cell DP's epilogue checks by
// evaluating the reference's sets directly (m2_ref_present).  A plan whose search sets do not have the shape the proof
// needs (anything kmer_heuristic builds has it) does not take the streaming form (m2_build returns false: the older
// kernels serve it).
//   class W   the k + 1 chunks of the whole adapter, anywhere in the read: a last-row candidate (_align.pyx:496-533,
//             cost <= k) holds one of them unedited.  Condition (i): every one of them is a whole-read k-mer of the
//             reference's sets too, so a W hit IS a REF hit.
//   tails     the rows of the last column (_align.pyx:536-572) with threshold e >= 1 form error classes (rows
//             lmin_e .. lmax_e).  A class is cut into SUB-CLASSES of rows La .. Lb; a row L of a sub-class can only be
//             acceptable if one of the e + 1 chunks of adapter[0:La] (offset o, q characters) occurs unedited, and it
//             then STARTS dist = n - s characters before the read's end with L - o - e <= dist <= L - o + e (at most e
//             insertions / deletions behind it): the entry's window [La - o - e, Lb - o + e].  Longer prefixes give longer
//             k-mers, narrow windows give few hits: a read of BASELINE's C4 meets ~1.5 tail k-mers instead of the ~6 the
//             reference's 5- to 7-mers with their windows of 19 .. 33 characters give.  class hi: e >= 2, class lo:
//             e == 1 (such a pair scans ~22 columns of a 32-row word).
//             Condition (iv): for every error class the e + 1 chunks of adapter[0:lmin_e] are k-mers of the reference's
//             sets with windows >= lmax_e.  An acceptable row L (cost c <= e, query_start qs) holds one of THOSE unedited
//             inside [qs, n), i.e. starting at most n - qs characters before the end: whenever n - qs <= lmax_e the
//             reference's kmers_present is true for the pair.  A match with n - qs > lmax_e (insertions in a row near
//             lmax_e) is the corner: the cell DP -- only it reports such a match; the scan's shortcuts are without
//             insertions -- evaluates the reference's sets of that adapter on that read (m2_ref_present) before it merges.
//   class E0  rows without tolerance (min_overlap .. lmax0) match exactly or not at all: adapter[0:i] is the read's
//             suffix.  Groups of rows ia .. ib are found through the k-mer adapter[0:ia] starting ia .. ib characters
//             before the end, and decided by the suffix compare (m2_exact_tail) in the prefilter itself.  Condition
//             (iii): for every such row some PREFIX adapter[0:q], q <= i, is a k-mer of the reference's sets with a window
//             >= i -- the exact match holds it, so kmers_present is true.
// The classes are probed in this order (W over the whole read, then hi, lo, E0 over the read's tail); a pair is emitted
// at its FIRST hit with the class being probed then.  A further hit of a pair that exists sets its "again" bit (the
// `wide` bitset): a whole-read pair that took the window of its one occurrence (CAH_M2_PAIR_PRECISE) falls back to the
// full window then.
//
// Characters: A 0, C 1, G 2, T 3, either case; anything else 4 (no k-mer holds it, as in KmerFinder without wildcards:
// _match_tables.py:81-98).  Two kinds of words, the newest character in the lowest bits:
//   * what is LOOKED UP (bitmaps, directory, entries' keys, events) takes two bits per character, sixteen characters per
//     32-bit word, and reads code 4 as 'A' (m2_roll2): a hit through a k-mer that holds such a character makes a pair whose
//     scan finds nothing -- the filter stays lossless, and what a scan accepts kmers_present accepts by (i), (iii), (iv);
//   * what is COMPARED for a result (the error-free overlaps at the read's end: m2_exact_tail, the adapters' prefixes) takes
//     three bits per character, ten characters per word -- k-mers of up to CAH_M2_MAXQ characters.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define M2_HD __host__ __device__ __forceinline__
#else
#define M2_HD inline
#endif

#define CAH_M2_SLOTS 4096            // directory slots: home = low 12 bits of the bitmap index -> (first entry, entries)
#define CAH_M2_MAX_ENTRIES 2304
#define CAH_M2_MAX_GROUP 15          // the directory's count field saturates here: a home with more entries is walked to its end (m2_home_of)
#define CAH_M2_BM_WORDS 3112         // presence bitmaps: 64 Kbit hashed (class W, index class 8: probed at every character), then
#define CAH_M2_BM8_WORDS 2048        // the tail classes' exact bitmaps and the tables of first adapters (m2_bit, CAH_M2_FIXED_WORD)
#define CAH_M2_MAXQ 10
#define CAH_M2_MAX_PASSES 12       // passes of the tail classes (their hit masks share eight registers)
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
// meta: adapter : 7 | q : 4 @7 | cls : 2 @11 | dlo : 8 @13 | dhi : 8 @21 | precise chunk : 3 @29
// [dlo, dhi]: the k-mer counts when it starts dlo .. dhi characters before the read's end (class W: 0 .. 255 = anywhere)
M2_HD uint32_t m2_meta(int adapter, int q, int cls, int dlo, int dhi, int pchunk) {
    return (uint32_t)adapter | ((uint32_t)q << 7) | ((uint32_t)cls << 11) | ((uint32_t)dlo << 13) | ((uint32_t)dhi << 21) |
           ((uint32_t)pchunk << 29);
}
M2_HD int m2_adapter(uint32_t meta) { return (int)(meta & 127u); }
M2_HD int m2_q(uint32_t meta) { return (int)((meta >> 7) & 15u); }
M2_HD int m2_cls(uint32_t meta) { return (int)((meta >> 11) & 3u); }
M2_HD int m2_dlo(uint32_t meta) { return (int)((meta >> 13) & 255u); }
M2_HD int m2_dhi(uint32_t meta) { return (int)((meta >> 21) & 255u); }
// A class-W entry whose k-mer is chunk c -- and only that -- of the adapter's k + 1 <= 4 whole-adapter chunks: 1 + c, or 0
M2_HD unsigned m2_precise_chunk(uint32_t meta) { return meta >> 29; }
// does the k-mer count when it starts `dist` characters before the end?  (reads of the streaming form have <= 160)
M2_HD bool m2_in_window(uint32_t meta, int dist) {
    const unsigned lo = (meta >> 13) & 255u, hi = (meta >> 21) & 255u;
    return (unsigned)dist - lo <= hi - lo;
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
// ---- the words the tables are looked up with: TWO bits per character, sixteen characters per 32-bit word, the newest lowest.
// A character that is not A / C / G / T reads as 'A' there: a hit through a k-mer that holds such a character makes a pair
// that the scan finds nothing for (no alignment holds the character unedited) -- the filter stays lossless, and what a pair's
// scan accepts the reference's kmers_present accepts (conditions (i), (iii), (iv) above).  The 3-bit words (m2_mask,
// m2_exact_tail, the adapters' prefixes) remain where characters are COMPARED for a result: the error-free overlaps at the
// read's end.
M2_HD uint32_t m2_roll2(uint32_t r2, unsigned c) { return (r2 << 2) | (m2_code(c) & 3u); }
M2_HD uint32_t m2_mask2(int q) { return q >= 16 ? 0xFFFFFFFFu : ((1u << (2 * q)) - 1u); }
// index (16 bits) of the last qc = min(q, 8) characters: the home slot of the directory is its low 12 bits
M2_HD uint32_t m2_salt(int qc) { return (uint32_t)(8 - qc) * 0x1D3Bu; }
M2_HD uint32_t m2_index(uint32_t r2, int qc) {
    const uint32_t key = r2 & m2_mask2(qc);
    return ((key ^ (key >> 7)) ^ m2_salt(qc)) & 0xFFFFu;
}
// the home of an entry (from its own k-mer): the entries of a home are consecutive, and a home with CAH_M2_MAX_GROUP or more
// of them -- adapters that share their k-mers -- is walked while this stays the event's home
M2_HD uint32_t m2_home_of(uint32_t key, uint32_t meta) {
    const int q = (int)((meta >> 7) & 15u);
    return m2_index(key, q < 8 ? q : 8) & (CAH_M2_SLOTS - 1);
}
// The presence bitmaps.  Bits [0, 64 K): every k-mer of eight or more characters by its last eight -- EXACT (4^8 bits), one AND
// of the rolling word per probe, at every character -- and, hashed (m2_index), class W's k-mers of fewer characters.  Behind
// them the tail classes' k-mers of fewer than eight characters, EXACT too: bit = m2_tail_region(q, class) + the k-mer.  Round 6:
// these k-mers used to share 32 Kbit through a hash of 3-bit words, which folded 18- and 21-bit keys into 15 bits -- every
// second event of the six- and seven-character passes was a k-mer that no entry holds (C4: 6.0 tail events per read, 3.6 now).
// (regions are multiples of 32 bits.  q 7: 16 Kbit shared by the tail classes; q 6: 4 Kbit for class hi and 4 for lo / E0; q 5:
// 1 Kbit for hi / lo and 1 for E0 -- a pass of one class does not see the other's k-mers; q <= 4: shared)
M2_HD uint32_t m2_tail_region(int q, int cls) {
    return q >= 7 ? 0u : q == 6 ? (cls == 1 ? 16384u : 20480u) : q == 5 ? (cls == 3 ? 25600u : 24576u)
         : q == 4 ? 26624u : q == 3 ? 26880u : q == 2 ? 26944u : 26976u;
}
#define CAH_M2_TAIL_BITS 27008u
// Behind the exact regions (bit 27 008 of the second bitmap): one BYTE per string of q <= 4 characters -- the first
// adapter that begins with it (0xFF: none).  An error-free overlap of q characters IS "the read's last q characters are
// adapter[0:q]": every read looks its own end up there and merges pack_best(q, 0, adapter, ...) -- no event, no directory walk
// (round 6; 1.9 entries per read of C4 went through the directory for this before).  The candidate is the first adapter
// WITHOUT a pair (the merge keeps the higher score, then fewer errors, then the first adapter -- kernels.h: pack_best; an
// adapter whose pair exists gets its answer from the scan, and that answer may be worth less than this overlap: at rate 0.2
// five characters with one mismatch have more matches and the same score): behind the tables, per q, a byte per adapter =
// the next adapter with the same first q characters (0xFF: none).
#define CAH_M2_FIXED_WORD (CAH_M2_BM8_WORDS + 844)          // word of the bitmap array where the tables begin (214 words)
#define CAH_M2_FIXED_MAXQ 4
M2_HD uint32_t m2_fixed_off(int q) { return q >= 4 ? 0u : q == 3 ? 256u : q == 2 ? 320u : 336u; }   // bytes; 340 in all
M2_HD uint32_t m2_fixed_next(int q) { return 344u + 128u * (uint32_t)(q - 1); }                       // bytes; up to 856
M2_HD uint32_t m2_bit(uint32_t r2, int qc, int cls) {
    if (qc >= 8) return r2 & 0xFFFFu;                    // the last eight characters, exact: 4^8 = the first bitmap's 64 Kbit
    if (cls == 0) return m2_index(r2, qc);               // class W's shorter k-mers: hashed into the same 64 Kbit
    return CAH_M2_BM8_WORDS * 32u + m2_tail_region(qc, cls) + (r2 & m2_mask2(qc));
}

struct CahMulti2Header {
    int32_t ok;
    int32_t n_adapters, m, k, min_overlap;
    int32_t lmax0;                 // the largest overlap length without error tolerance (rows min_overlap .. lmax0; 0: none)
    int32_t q_mask[4];             // per class: bit qc set = k-mers of index class qc = min(q, 8) exist
    int32_t span[4];               // classes 1..3: a k-mer of the class starts at most this many characters before the end
    int32_t open_L[4][9];          // [cls][qc]: the largest dhi of the class's entries of that index class (-1: none)
    int32_t close_L[4][9];         // ... and the smallest dlo
    int32_t win_dist[4];           // classes hi, lo: a pair's scan window starts at column n - win_dist
    // The tail classes are probed inside the main pass, in the read's last chunks: one PASS per (class, index class) in class
    // order (hi, lo, E0; <= 8) with a hit mask of its own -- 32 positions from the pass's first (a k-mer of the pass starts
    // tq_close .. tq_open characters before the end; a wider window is two passes) --, whose hits become events of class
    // tq_cls[j] behind the main pass, class by class.  A pass of at most 16 positions shares its mask register with another
    // such pass (tq_slot: which of the eight registers, tq_shift: 0 or 16)
    int32_t tq_n;
    int32_t tq_cls[CAH_M2_MAX_PASSES], tq_qc[CAH_M2_MAX_PASSES];
    int32_t tq_open[CAH_M2_MAX_PASSES], tq_close[CAH_M2_MAX_PASSES];
    int32_t tq_slot[CAH_M2_MAX_PASSES], tq_shift[CAH_M2_MAX_PASSES];
    // E0 entries that must be the read's last q characters (dlo == dhi == q: the rows below 5) never become events: every
    // lane looks its own read's end up.  (Measured, round 6: the other E0 entries the same way -- the word shifted by one
    // and two characters -- cost 560 VALU instructions per 64 reads more than their 0.5 events per read.)
    int32_t qm_fixed;              // index classes with such entries
    int32_t rows_lo;               // the longest overlap whose threshold is <= 1 (rows a pair of class lo can match)
    int32_t tail_warm;             // characters in front of a sweep's first probe that must be in the word (max q - 1)
    uint32_t n_entries;
    // the corner the cell DP checks (see the head of this file): row i belongs to an error class that ends at lmax_row[i]
    int32_t ref_span;              // the widest tail window among the reference's k-mers
    uint8_t lmax_row[72];
};

// Is one of the reference's tail k-mers (list entries [begin, end): {code, q | window << 8}; whole-read k-mers are not in
// the list) inside its window in this read?  KmerFinder.kmers_present restricted to the sets (-L, None)
// (_kmer_finder.pyx:186-213).  Rare path: plain loops.
M2_HD bool m2_ref_present(const uint32_t* list, int begin, int end, const uint8_t* read, int n, int span) {
    uint32_t r = 0x24924924u;
    int p0 = n - span;
    if (p0 < 0) p0 = 0;
    for (int p = p0; p < n; ++p) {
        r = (r << 3) | m2_code(read[p]);
        for (int u = begin; u < end; ++u) {
            const uint32_t code = list[2 * u], qw = list[2 * u + 1];
            const int q = (int)(qw & 255u), w = (int)(qw >> 8);
            if (p - q + 1 >= 0 && n - (p - q + 1) <= w && (r & m2_mask(q)) == code) return true;
        }
    }
    return false;
}

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
    std::vector<int32_t> ref_begin;        // [n_adapters + 1]: adapter a's tail k-mers of the reference's sets are
    std::vector<uint32_t> ref_list;        // ref_list[2 * u], u in [ref_begin[a], ref_begin[a + 1]): {code, q | window << 8}
};

struct M2RefKmer { std::string kmer; int window; };   // window: CAH_M2_WHOLE, or L of the tail set (-L, None)

inline uint32_t m2_encode(const std::string& s) {
    uint32_t r = 0;
    for (char ch : s) r = (r << 3) | m2_code((unsigned char)ch);
    return r;
}
inline uint32_t m2_encode2(const std::string& s) {          // (plain A / C / G / T strings of up to 16 characters)
    uint32_t r2 = 0;
    for (char ch : s) r2 = m2_roll2(r2, (unsigned char)ch);
    return r2;
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
// non-monotone thresholds, an error-free class longer than a word, or search sets without the properties (i), (iii),
// (iv) of this file's head.
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
    const int row0 = std::max(min_overlap, 1);
    for (int i = row0; i <= m; i++) {
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
    for (int i = 0; i < 72; i++) h.lmax_row[i] = 0;
    for (int i = row0; i <= lmax0; i++) h.lmax_row[i] = (uint8_t)lmax0;
    for (const Tail& tl : tails) for (int i = tl.lmin; i <= tl.lmax; i++) h.lmax_row[i] = (uint8_t)tl.lmax;
    // sub-classes of the tail classes (rows La .. Lb) and groups of the error-free rows
    struct Sub { int e, la, lb; };
    std::vector<Sub> subs;
    for (const Tail& tl : tails) {
        const int rows = tl.lmax - tl.lmin + 1;
        if (tl.lmax + tl.e >= 240) return false;
        // (the shortest k-mers -- the first rows of the one-error class -- get the narrowest windows)
        if (tl.e == 1 && rows >= 8) {
            const int half = (rows - 2 + 1) / 2;
            subs.push_back({tl.e, tl.lmin, tl.lmin + 1});
            subs.push_back({tl.e, tl.lmin + 2, tl.lmin + 2 + half - 1});
            subs.push_back({tl.e, tl.lmin + 2 + half, tl.lmax});
        } else if (rows >= 6) {
            const int half = (rows + 1) / 2;
            subs.push_back({tl.e, tl.lmin, tl.lmin + half - 1});
            subs.push_back({tl.e, tl.lmin + half, tl.lmax});
        } else subs.push_back({tl.e, tl.lmin, tl.lmax});
    }
    std::vector<Sub> groups0;
    for (int i = row0; i <= lmax0;) {
        const int last = i <= 4 ? i : (i <= 6 ? std::min(6, lmax0) : lmax0);
        groups0.push_back({0, i, last});
        i = last + 1;
    }
    struct Ent { std::string kmer; int adapter; int cls; int dlo, dhi; };
    std::vector<Ent> ents;
    t.ref_begin.assign((size_t)A + 1, 0);
    t.ref_list.clear();
    h.ref_span = 0;
    for (int a = 0; a < A; a++) {
        const std::string& ad = adapters[(size_t)a];
        if ((int)ad.size() != m) return false;
        const std::vector<M2RefKmer>& rf = ref[(size_t)a];
        const size_t first = ents.size();
        auto add = [&](const std::string& s, int cls, int dlo, int dhi) -> bool {
            if (s.empty() || (int)s.size() > CAH_M2_MAXQ) return false;
            dlo = std::max(dlo, (int)s.size());                         // (a k-mer of q characters starts q or more before the end)
            if (dhi < dlo) return true;                                 // (never inside the read: no entry)
            for (size_t i = first; i < ents.size(); i++)
                if (ents[i].kmer == s && ents[i].cls == cls) {
                    ents[i].dlo = std::min(ents[i].dlo, dlo);
                    ents[i].dhi = std::max(ents[i].dhi, dhi);
                    return true;
                }
            ents.push_back({s, a, cls, dlo, dhi});
            return true;
        };
        auto ref_has = [&](const std::string& s, int window) -> bool {      // a k-mer of the reference's sets with a window >= `window`
            for (const M2RefKmer& rk : rf)
                if (rk.kmer == s && (rk.window == CAH_M2_WHOLE || (window != CAH_M2_WHOLE && rk.window >= window))) return true;
            return false;
        };
        // a last-row candidate (cost <= kacc) holds one of the k + 1 chunks of the whole adapter -- (i): a REF k-mer too
        if (min_overlap <= m)
            for (const std::string& s : m2_chunks(ad, kacc + 1)) {
                if (!ref_has(s, CAH_M2_WHOLE)) return false;
                if (!add(s, M2_W, 0, 255)) return false;
            }
        // (iv): the reference's own chunks of every error class, with windows that reach the class's last row
        for (const Tail& tl : tails)
            for (const std::string& s : m2_chunks(ad.substr(0, (size_t)tl.lmin), tl.e + 1))
                if (!ref_has(s, tl.lmax)) return false;
        for (const Sub& sb : subs) {
            int o = 0;
            for (const std::string& s : m2_chunks(ad.substr(0, (size_t)sb.la), sb.e + 1)) {
                if (!add(s, sb.e >= 2 ? M2_HI : M2_LO, sb.la - o - sb.e, sb.lb - o + sb.e)) return false;
                o += (int)s.size();
            }
        }
        // (iii): an exact overlap of i characters holds a prefix of the adapter that the reference looks for there
        for (int i = row0; i <= lmax0; i++) {
            bool ok = false;
            for (const M2RefKmer& rk : rf)
                if ((int)rk.kmer.size() <= i && ad.compare(0, rk.kmer.size(), rk.kmer) == 0 &&
                    (rk.window == CAH_M2_WHOLE || rk.window >= i)) ok = true;
            if (!ok) return false;
        }
        for (const Sub& g : groups0)
            if (!add(ad.substr(0, (size_t)g.la), M2_SHORT, g.la, g.lb)) return false;
        // the reference's tail k-mers, for m2_ref_present
        for (const M2RefKmer& rk : rf) {
            if (rk.kmer.empty() || (int)rk.kmer.size() > CAH_M2_MAXQ) return false;
            if (rk.window == CAH_M2_WHOLE) continue;
            if (rk.window < 1 || rk.window >= 240) return false;
            const uint32_t code = m2_encode(rk.kmer);
            if (code & 0x24924924u) return false;                       // not plain ACGT
            t.ref_list.push_back(code);
            t.ref_list.push_back((uint32_t)rk.kmer.size() | ((uint32_t)rk.window << 8));
            h.ref_span = std::max(h.ref_span, rk.window);
        }
        t.ref_begin[(size_t)a + 1] = (int32_t)(t.ref_list.size() / 2);
    }
    t.dir.assign(CAH_M2_SLOTS, 0);
    t.bitmap.assign(CAH_M2_BM_WORDS, 0u);
    uint8_t* const fixed_tab = reinterpret_cast<uint8_t*>(t.bitmap.data() + CAH_M2_FIXED_WORD);
    for (int i = 0; i < 856; i++) fixed_tab[i] = 0xFFu;
    for (int c = 0; c < 4; c++)
        for (int q = 0; q < 9; q++) { h.open_L[c][q] = -1; h.close_L[c][q] = 1000; }
    int max_q = 1;
    struct Placed { uint32_t home, key, meta; };
    std::vector<Placed> placed;
    int nonfixed_mask[4] = {0, 0, 0, 0};
    for (const Ent& e : ents) {
        const int q = (int)e.kmer.size(), qc = std::min(q, 8);
        if (m2_encode(e.kmer) & 0x24924924u) return false;            // not plain ACGT
        const uint32_t code = m2_encode2(e.kmer);                     // the entry's key: two bits per character
        const uint32_t idx = m2_index(code, qc);
        // an error-free overlap of q <= 4 characters: the table of first adapters (CAH_M2_FIXED_WORD), no entry
        const bool fixed = e.cls == M2_SHORT && e.dlo == q && e.dhi == q && q <= CAH_M2_FIXED_MAXQ;
        if (fixed) {
            // (the adapters of a string in ascending order: head in the table, then the chain)
            if (e.adapter >= 128) return false;
            uint8_t* at = &fixed_tab[m2_fixed_off(q) + (code & m2_mask2(q))];
            while (*at != 0xFFu && *at < e.adapter) at = &fixed_tab[m2_fixed_next(q) + *at];
            if (*at != (uint8_t)e.adapter) {
                fixed_tab[m2_fixed_next(q) + (uint32_t)e.adapter] = *at;
                *at = (uint8_t)e.adapter;
            }
        } else {
            const uint32_t bit = m2_bit(code, qc, e.cls);
            t.bitmap[bit >> 5] |= 1u << (bit & 31);
        }
        // which whole-adapter chunk is it?  (no answer when the string is two of them or occurs elsewhere in the adapter too:
        // an alignment could then hold the occurrence at another offset.  A string that is a k-mer of a tail class or an
        // error-free group as well has an entry of its own there: where its occurrence stands for rows of the last column
        // too, that entry's event finds the pair made and sets its "again" bit -- the pair then takes its full window)
        int pchunk = 0;
        if (e.cls == M2_W && kacc + 1 <= 4) {
            const std::string& ad = adapters[(size_t)e.adapter];
            const std::vector<std::string> ch = m2_chunks(ad, kacc + 1);
            int which = -1, times = 0;
            for (int c = 0; c < (int)ch.size(); c++) if (ch[(size_t)c] == e.kmer) { which = c; times++; }
            int occurrences = 0;
            for (size_t at = ad.find(e.kmer); at != std::string::npos; at = ad.find(e.kmer, at + 1)) occurrences++;
            if (times == 1 && occurrences == 1) pchunk = 1 + which;
        }
        if (!fixed) placed.push_back({idx & (CAH_M2_SLOTS - 1), code, m2_meta(e.adapter, q, e.cls, e.dlo, e.dhi, pchunk)});
        h.q_mask[e.cls] |= 1 << qc;
        h.open_L[e.cls][qc] = std::max(h.open_L[e.cls][qc], e.dhi);
        h.close_L[e.cls][qc] = std::min(h.close_L[e.cls][qc], e.dlo);
        if (e.cls != M2_W) h.span[e.cls] = std::max(h.span[e.cls], e.dhi);
        max_q = std::max(max_q, q);
        if (e.cls == M2_SHORT) {
            if (fixed) h.qm_fixed |= 1 << qc; else nonfixed_mask[M2_SHORT] |= 1 << qc;
        } else if (e.cls != M2_W) nonfixed_mask[e.cls] |= 1 << qc;
        if (e.cls == M2_HI || e.cls == M2_LO) h.win_dist[e.cls] = std::max(h.win_dist[e.cls], e.dhi + 1);
    }
    if (placed.size() > CAH_M2_MAX_ENTRIES) return false;
    h.n_entries = (uint32_t)placed.size();
    std::stable_sort(placed.begin(), placed.end(), [](const Placed& x, const Placed& y) { return x.home < y.home; });
    t.entries.clear();
    for (size_t i = 0; i < placed.size();) {
        size_t j = i;
        while (j < placed.size() && placed[j].home == placed[i].home) j++;
        t.dir[placed[i].home] = (uint16_t)m2_dir((int)i, (int)std::min<size_t>(j - i, CAH_M2_MAX_GROUP));
        for (size_t u = i; u < j; u++) t.entries.push_back(CahM2Slot{placed[u].key, placed[u].meta});
        i = j;
    }
    // (hi windows must cover lo's: nothing depends on it, but a plan without a hi class still sweeps with one distance)
    h.win_dist[M2_HI] = std::max(h.win_dist[M2_HI], h.win_dist[M2_LO]);
    if (h.win_dist[M2_LO] == 0) h.win_dist[M2_LO] = h.win_dist[M2_HI];
    h.tail_warm = max_q - 1;
    // the passes of the tail classes (class order: hi, lo, E0); a pass's positions must fit its 32-bit hit mask
    h.tq_n = 0;
    for (int c = M2_HI; c <= M2_SHORT; c++)
        for (int q = 1; q <= 8; q++) {
            if (!((nonfixed_mask[c] >> q) & 1)) continue;
            const int qx = q < 8 ? q : CAH_M2_MAXQ;
            int close = h.close_L[c][q];
            const int open = h.open_L[c][q];
            // positions n + q - 1 - open .. n + qx - 1 - close: (open - close) + (qx - q) + 1 of them
            while (close <= open) {
                const int top = std::min(open, close + 31 - (qx - q));
                if (h.tq_n == CAH_M2_MAX_PASSES) return false;
                h.tq_cls[h.tq_n] = c; h.tq_qc[h.tq_n] = q; h.tq_open[h.tq_n] = top; h.tq_close[h.tq_n] = close;
                h.tq_n++;
                close = top + 1;
            }
        }
    {
        // mask registers: a pass of more than 16 positions takes one, two narrower ones share one
        int slots = 0, half_open = -1;
        for (int j = 0; j < h.tq_n; j++) {
            const int q = h.tq_qc[j], qx = q < 8 ? q : CAH_M2_MAXQ;
            const int width = (h.tq_open[j] - h.tq_close[j]) + (qx - q) + 1;
            if (width > 16) { h.tq_slot[j] = slots++; h.tq_shift[j] = 0; }
            else if (half_open >= 0) { h.tq_slot[j] = half_open; h.tq_shift[j] = 16; half_open = -1; }
            else { h.tq_slot[j] = half_open = slots++; h.tq_shift[j] = 0; }
        }
        if (slots > 8) return false;
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
