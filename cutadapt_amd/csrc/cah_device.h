// cah_device.h -- structures shared by the host plan builder and the gfx950 kernels.
//
// HBM layout of a plan (all immutable after cah_plan_create):
//   CahMatcher  matchers[n_adapters]   one per adapter: DP constants + char->row-bitset table
//   CahKmerWord kmer_words[total]      packed shift-and words of all adapters' prefilters
// A kernel launch works on ONE matcher (wave-uniform constants live in SGPRs); it copies the
// tables it needs into LDS once per workgroup.
#pragma once
#include <stdint.h>

#define CAH_MAX_M 64           // adapter length limit: one 64-bit row bitset per read char
#define CAH_TABLE_CHARS 128    // ASCII; bytes >= 0x80 are invalid input
#ifndef CAH_FILTER_SLOTS
#define CAH_FILTER_SLOTS 6      // packed 64-bit k-mer words the prefilter advances together (state in VGPRs)
#endif
#ifndef CAH_FILTER_SLOTS_NARROW
#define CAH_FILTER_SLOTS_NARROW 8   // ... when every word of the plan fits 32 bits
#endif

// Packed DP cell payload (one VGPR): ((origin + CAH_ORIGIN_BIAS) << 12) + (score + CAH_SCORE_BIAS)
// origin in [-64, 1e6], score in [-2048, 2047] (bounds derived in DESIGN.md); match/mismatch/
// indel score updates are the inline constants +1/-1/-2 on the low field.
#define CAH_SCORE_BITS 12
#define CAH_SCORE_BIAS 2048
#define CAH_ORIGIN_BIAS 128

struct CahMatcher {
    int32_t kind;              // CAH_KIND_*
    int32_t m;                 // adapter length
    int32_t k;                 // (int)(max_error_rate * m)              _align.pyx:343
    int32_t flags;             // EndSkip bits
    int32_t indel_cost;        // insertion == deletion cost            _align.pyx:219-220
    int32_t min_overlap;
    int32_t wildcard_ref;
    int32_t effective_length;  // m - #N when wildcard_ref              _align.pyx:268-271
    int32_t cmp_max_k;         // comparers: (int)(rate * effective_length)  _align.pyx:633
    int32_t has_filter;        // 0 = MockKmerFinder (always present)
    int32_t first_word;        // index into kmer_words
    int32_t n_words;
    int32_t n_counts[CAH_MAX_M + 1];   // #N/n in adapter[:i]            _align.pyx:261-266
    int32_t thr[CAH_MAX_M + 1];        // thr[L] = floor(L * max_error_rate): integer form of
                                       // `cost <= cur_effective_length * max_error_rate` (:513, :559)
    int32_t skip_ok;           // 1: DP may start at (first k-mer hit) - m - k - 1 (see api.cpp)
    int32_t narrow_words;      // 1: every packed k-mer word of this matcher fits 32 bits
    uint64_t rowmask[CAH_TABLE_CHARS]; // bit i set <=> adapter[i] matches this read character
                                       // (folds translate() + the three compare modes, :322-328, :442-445;
                                       //  for comparers bit i refers to the i-th compared position)
    // ---- bit-parallel cost scan (back_scan.h; 3' adapters with unit costs) ------------------------
    int32_t long_dp;           // 1: m > CAH_MAX_M -- the aligner / comparer runs in k_dp_long (column in HBM scratch);
                               // rowmask, n_counts, thr of this struct are unused then
    int32_t scan_ok;           // 1: k_back_scan may classify reads before the cell DP
    int32_t kacc;              // thr[effective_length] (-1 if m < min_overlap): acceptable last-row cost
    int32_t thr_last[CAH_MAX_M + 1];   // thr[effective length of adapter[0:i]]: threshold of row i in the last column
    uint64_t scanmask[CAH_TABLE_CHARS]; // rowmask << (64 - m) | ones below: the adapter in the top m bits
    // 1: a read that holds the adapter unedited at its anchored place (prefix aligner: position 0, suffix aligner:
    // the last m characters) passes this plan's prefilter -- some k-mer of a search set is a piece of the adapter
    // inside its window there, and every character the aligner accepts at that place the k-mer table accepts too.
    // For anchored adapters that tolerate no error (k_anchored_exact) the prefilter is then redundant and skipped.
    int32_t filter_implied;
    int32_t pad_;
};

struct CahKmerWord {
    int64_t start;             // search window, KmerFinder semantics (stop == 0: to the end)
    int64_t stop;
    uint64_t init_mask;        // bit at every k-mer start           _kmer_finder.pyx:143
    uint64_t found_mask;       // bit at every k-mer end             _kmer_finder.pyx:147
    uint64_t mask[CAH_TABLE_CHARS];
};

// ---------------------------------------------------------------------------------------------
// Lean prefilter for plans whose search sets are all "whole read" (start 0, stop None), "last L
// characters" (start -L, stop None) or "characters start..stop" near the 5' end -- everything
// kmer_heuristic builds for 3', 5' and anywhere adapters.
//   lead words   whole-read k-mers: constant start bits, every k-mer end counts
//   gated words  tail k-mers (sets (-L, None)) of ALL window lengths share words: a k-mer of (-L, None) may only
//                START at positions >= n - L, so its start bit is injected only there; head k-mers (sets
//                (start, stop)) may start at p in [start, stop - len] and count when they END at p in
//                [start + len - 1, stop - 1].  Both are one mechanism: per word a start-bit gate and a found-bit
//                gate, tables indexed by idx = p + base(word) -- base = CAH_GATE_ZERO - n for a tail word (the
//                index runs with the distance from the read end), 0 for a head word (the position itself).
// No per-lane window bookkeeping and no per-character window masks are left.  The kernels come in classes
// <NL lead slots, NG gated slots> with every loop bound a compile-time constant; a plan takes the smallest class
// that holds its words (unused slots have empty masks and closed gates).
// ---------------------------------------------------------------------------------------------
#define CAH_LEAN_MAX_LEAD 3
#define CAH_LEAN_MAX_GATED 6
#define CAH_LEAN_SPAN 64                          // longest tail window / largest head stop the lean kernels take
#define CAH_LEAN_DELAY 3                          // delay bits behind a lead k-mer (see lead_delay)
#define CAH_GATE_PAD 16                           // a 16-character chunk may begin this far before a window / end after it
#define CAH_GATE_ZERO (CAH_LEAN_SPAN + CAH_GATE_PAD)     // tail words: index of distance 0 (one past the last character)
#define CAH_GATE_LEN (CAH_LEAN_SPAN + 2 * CAH_GATE_PAD)  // entries per gate table
#define CAH_LEAN_MAX_TW 4                         // T-words (tail k-mers with delay bits, found-gated; see tw_* below)
#define CAH_TW_DIST0 15                           // tw_found index of dist = 0 (the group's last character is the read's last);
                                                  // below it: groups that end up to 15 characters past the read (all zero from -4 on)
#define CAH_TW_DIST_LEN (CAH_LEAN_SPAN + 36)      // dist = -15 .. CAH_LEAN_SPAN + 20: every group of a chunk that touches a window
struct CahLeanFilter {
    int32_t ok;                                   // 1: this matcher can use the lean kernels
    int32_t n_lead, n_gated;                      // words in use
    int32_t n_tail;                               // gated words [0, n_tail) are tail words, [n_tail, n_gated) head words
    int32_t tail_span;                            // longest tail window (0: none)
    int32_t head_span;                            // largest stop of a head window (0: none)
    // CAH_LEAN_DELAY if every lead k-mer is followed by that many delay bits (they pass every byte, so a k-mer end
    // stays visible for three more characters and the kernel tests the state once per 4-character group instead of
    // accumulating it per character; used whenever it does not cost an extra word), else 0
    int32_t lead_delay;
    uint32_t lead_init[CAH_LEAN_MAX_LEAD];        // start bits of a lead word
    uint32_t lead_found[CAH_LEAN_MAX_LEAD];       // bit at every k-mer end (and its delay bits)
    uint32_t lead_pass[CAH_LEAN_MAX_LEAD];        // the delay bits: set in the mask of every byte value
    uint32_t gated_found[CAH_LEAN_MAX_GATED];     // bit at every k-mer end
    int32_t gated_span[CAH_LEAN_MAX_GATED];       // tail words: the widest tail window of the word's k-mers (the word is
                                                  // idle until that many characters are left); head words: 0
    uint32_t lead_mask[CAH_LEAN_MAX_LEAD][CAH_TABLE_CHARS];
    uint32_t gated_mask[CAH_LEAN_MAX_GATED][CAH_TABLE_CHARS];
    // START-bit gates.  Tail word, idx = CAH_GATE_ZERO - d (d = n - p: 1 is the last character): the start bits of
    // every set (-L, None) with L >= d.  Head word, idx = p: the start bits of every set (start, stop) with
    // start <= p <= stop - len.  Everything else is 0 (closed).
    uint32_t gate_init[CAH_LEAN_MAX_GATED][CAH_GATE_LEN];
    // ---- T-words (k_filter_stream2, stream2.h): the tail k-mers once more, packed like lead k-mers -- every
    // k-mer followed by CAH_LEAN_DELAY delay bits, start bits injected at EVERY position (no start gates) -- so
    // that they take the same four-characters-per-step update as the lead words.  What a window restricts is
    // then the END of a k-mer: a k-mer of length q of the set (-L, None) that ends at position e started at
    // e - q + 1 >= n - L  <=>  (n - 1 - e) <= L - q.  The state is looked at after every 4-character group (last
    // character t): delay bit d of a k-mer says "ended at t - d", so it counts iff dist + d <= L - q with
    // dist = n - 1 - t.  tw_found[w][dist + CAH_TW_DIST0] holds exactly those bits (dist = -3 .. CAH_LEAN_SPAN).
    int32_t tw_ok;                                // 1: lead words (with delay bits) + T-words describe this prefilter
    int32_t n_tw;                                 // T-words in use (ordered by widest window, widest first)
    int32_t tw_span[CAH_LEAN_MAX_TW];             // widest tail window of the word's k-mers
    uint32_t tw_init[CAH_LEAN_MAX_TW];            // start bits
    uint32_t tw_pass[CAH_LEAN_MAX_TW];            // delay bits: set in the mask of every byte value
    uint32_t tw_mask[CAH_LEAN_MAX_TW][CAH_TABLE_CHARS];
    uint32_t tw_found[CAH_LEAN_MAX_TW][CAH_TW_DIST_LEN];
};

// ---------------------------------------------------------------------------------------------
// Fused multi-adapter path (MultipleAdapters.match_to over many 3' adapters of one shape, e.g.
// `-a file:` with 96 adapters; reference adapters.py:1265-1286).  One pass over the reads finds, for
// every read, exactly the adapters whose KmerFinder.kmers_present() is true: the k-mers of ALL adapters
// are indexed by their last min(q, 8) characters (2 bits per base) in direct-address tables -- classes
// 1..7 hold the k-mers of exactly that length, class 8 every k-mer of 8..32 characters under its last
// 8 -- with one presence bitmap per class in LDS.  A bitmap hit is resolved through the directory
// (HBM/L2) to the entries of that key; an entry is verified against the read's rolling 2-bit code
// (all q characters, search window) before the (read, adapter) pair is emitted.
// ---------------------------------------------------------------------------------------------
#define CAH_MULTI_MAX_ADAPTERS 128      // per fused class pass (per-lane "already emitted" bitset: 16 B of LDS)
#define CAH_MULTI_CLASSES 9             // index 1..8 used
#define CAH_MULTI_TAB 32                // per-adapter match table: index = read character & 31 (letters only)

struct CahMultiEntry {
    uint64_t code;        // the k-mer, 2 bits per base, last character in the low bits
    uint32_t adapter;
    uint8_t q;            // k-mer length 1..32
    uint8_t window;       // 0: whole read; L: the k-mer must start within the last L characters
    uint16_t pad;
};

struct CahMultiDir { uint32_t begin, count; };

struct CahMultiHeader {
    int32_t ok;                               // 1: the plan has a fused multi-adapter path
    int32_t n_adapters;
    int32_t skip_ok;                          // every adapter's whole-read set has the pigeonhole property
    int32_t class_present[CAH_MULTI_CLASSES];
    int32_t class_everywhere[CAH_MULTI_CLASSES];   // some whole-read k-mer falls in this class
    int32_t class_lmax[CAH_MULTI_CLASSES];         // widest tail window of the class's tail k-mers (0: none)
    uint32_t dir_off[CAH_MULTI_CLASSES];           // first directory slot of the class
    uint32_t bm_off[CAH_MULTI_CLASSES];            // first bitmap word of the class
    uint32_t bm_words;                             // bitmap words in total
    uint32_t n_entries;
};

// ---------------------------------------------------------------------------------------------
// Adapters longer than CAH_MAX_M characters (long.hip): the scalar constants; the encoded adapter and its
// n_counts live in separate HBM arrays of the plan.
// ---------------------------------------------------------------------------------------------
struct CahLongMatcher {
    int32_t kind, m, k, flags, indel_cost, min_overlap, wildcard_ref, effective_length, cmp_max_k;
    int32_t cmp_equal;          // 1: characters are compared for equality, 0: encoded sets are ANDed (:442-445)
    double rate;                // max_error_rate: `cost <= effective_length * rate` is evaluated in double (:513)
    uint8_t qtab[CAH_TABLE_CHARS];   // read character -> encoding (IUPAC / ACGT / upper case, :322-328)
};
