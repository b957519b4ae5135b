// index.hip -- AdapterIndex on the GPU.
//
// The reference speeds up many anchored adapters of one kind (5' "^ADAPTER" or 3' "ADAPTER$",
// e.g. demultiplexing barcodes) with a dictionary that maps every string within k <= 3 errors of
// any adapter to (adapter, errors, matches) and looks up the read's prefix / suffix in it
// (reference src/cutadapt/adapters.py:1289-1551 AdapterIndex; the string sets come from
// _align.pyx:717-781 hamming_sphere and :784-882 edit_environment).
//
// Here the dictionary is built on the host by cah_index_create (same strings, same
// errors/matches, same collision and ambiguity rules), stored as an open-addressing hash table
// and probed by one GPU lane per read (k_index_lookup).  The rare reads whose affix contains 'N'
// take the reference's re-alignment path (_lookup_with_n, :1532-1551) inside the same kernel with
// a scalar per-lane restatement of Aligner.locate / the comparers.
//
// Two table layouts:
//   packed  every adapter is plain ACGT and at most 60 characters long (barcodes, primers): the
//           key IS the string, two bits per character in 2 x 64 bits -- no verification needed;
//   wide    anything else the reference accepts (adapters.py:1373-1386 has no length or alphabet
//           rule): adapters of up to 1000 characters and of any ASCII characters.  Keys are byte
//           strings (walked from the anchored end) in a pool; an entry holds the 64-bit FNV-1a
//           hash, which is a PREFIX hash -- one walk over the read yields the hash of every
//           indexed length -- and a hit is verified against the pool.  Non-ACGT characters only
//           survive in keys of --no-indels adapters (hamming_sphere keeps them, _align.pyx:717-781;
//           edit_environment emits ACGT only and counts them as mismatches, :794, :835) and match
//           a read holding the very same character there.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/cutadapt_hip.h"

extern int cah_set_error_(int code, const char* msg);   // api.cpp

#define IDX_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess) {                                                       \
            char b__[256];                                                             \
            snprintf(b__, sizeof(b__), "%s failed: %s", #expr, hipGetErrorString(e__)); \
            return cah_set_error_(CAH_EHIP, b__);                                      \
        }                                                                              \
    } while (0)

namespace {

constexpr int IDX_MAX_ADAPTER = 60;        // strings up to 60 + 3 characters fit 2 x 64 bits
constexpr int IDX_MAX_STRING = 64;
constexpr int IDX_WIDE_MAX_ADAPTER = 1000; // wide layout: the re-alignment columns of one wave live in LDS
constexpr uint64_t IDX_FNV_OFFSET = 0xCBF29CE484222325ull, IDX_FNV_PRIME = 0x100000001B3ull;
constexpr int IDX_MAX_LENGTHS = 64;
constexpr int IDX_MAX_DEVICES = 16;
constexpr uint32_t IDX_EMPTY = 0xFFFFFFFFu;

struct IdxEntry {             // 24 bytes
    uint64_t lo, hi;          // characters 0..31 / 32..63, two bits each (A0 C1 G2 T3), unused bits 0
    uint32_t len;             // IDX_EMPTY marks a free slot
    uint32_t val;             // adapter << 12 | errors << 8 | matches
};

struct IdxWideEntry {         // 24 bytes
    uint64_t hash;            // idx_wide_hash(FNV-1a state after len characters, len)
    uint32_t off;             // key characters in the pool, walked from the anchored end
    uint32_t len;             // IDX_EMPTY marks a free slot
    uint32_t adapter;
    uint32_t em;              // errors << 16 | matches
};

struct IdxAdapter {
    int32_t off, m, indels, max_k;      // max_k: the comparer's int(rate * m) (_align.pyx:633)
    int32_t set_beg, set_end;           // this adapter's k-mer search sets (set_end < 0: no prefilter)
    double rate;
};

struct IdxKmerSet { int32_t start, stop, kmer_beg, kmer_end; };   // window as KmerFinder takes it (stop 0 = end)
struct IdxKmer { int32_t off, len; };                             // into the k-mer character blob

struct IdxDeviceCopy {
    bool ready = false;
    IdxEntry* d_table = nullptr;
    IdxWideEntry* d_wtable = nullptr;
    uint8_t* d_pool = nullptr;
    int32_t* d_lengths = nullptr;
    IdxAdapter* d_adapters = nullptr;
    uint8_t* d_seqs = nullptr;
    IdxKmerSet* d_sets = nullptr;
    IdxKmer* d_kmers = nullptr;
    uint8_t* d_kmer_chars = nullptr;
    int n_cus = 256;
};

__host__ __device__ inline uint64_t idx_hash(uint64_t lo, uint64_t hi, uint32_t len) {
    uint64_t h = lo * 0x9E3779B97F4A7C15ull ^ hi * 0xC2B2AE3D27D4EB4Full ^ (uint64_t)len * 0x165667B19E3779F9ull;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return h;
}

__host__ __device__ inline uint64_t idx_wide_hash(uint64_t state, uint32_t len) {
    uint64_t h = state ^ ((uint64_t)len * 0x9E3779B97F4A7C15ull);
    h ^= h >> 31;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 29;
    return h;
}

inline int code_of(char c) {
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return -1;
    }
}

}  // namespace

struct cah_index {
    bool prefix = true;
    bool wide = false;                      // see the header: packed 2-bit keys or hashed byte strings
    std::vector<IdxEntry> table;
    std::vector<IdxWideEntry> wtable;
    std::string pool;
    uint64_t mask = 0;
    std::vector<int32_t> lengths;           // descending (adapters.py:1482)
    std::vector<IdxAdapter> adapters;
    std::string seqs;
    std::vector<IdxKmerSet> sets;           // k-mer prefilter of each adapter (only the 'N' path needs it)
    std::vector<IdxKmer> kmers;
    std::string kmer_chars;
    int64_t n_strings = 0;
    int32_t n_ambiguous = 0;
    mutable std::mutex mu;
    mutable IdxDeviceCopy dev[IDX_MAX_DEVICES];
};

namespace {

// packs s (forward for prefix indexes, reversed for suffix indexes: the kernel walks a read from
// the anchored end) -- false if s holds anything but ACGT
bool pack_key(const std::string& s, bool prefix, uint64_t& lo, uint64_t& hi) {
    lo = hi = 0;
    const int L = (int)s.size();
    for (int t = 0; t < L; t++) {
        const int c = code_of(prefix ? s[t] : s[L - 1 - t]);
        if (c < 0) return false;
        if (t < 32) lo |= (uint64_t)c << (2 * t);
        else hi |= (uint64_t)c << (2 * (t - 32));
    }
    return true;
}

// All strings within edit distance k of t over ACGT, with the edit distance and the number of
// matches of the optimal alignment under the reference's tie rule diagonal >= left >= up
// (_align.pyx:784-882: depth-first over the strings, one DP row per character, rows pruned when
// every cell exceeds k).  emit(s, errors, matches).
template <typename F>
void edit_environment(const std::string& t, int k, F&& emit) {
    const int n = (int)t.size();
    const int W = n + 1, rows = n + k + 1;
    const int INF = (k + 1) * 0x01010101;                 // the reference's memset(costs, k+1) pattern (:812)
    std::vector<int> cost((size_t)rows * W, INF), mat((size_t)rows * W, 0);
    for (int i = 0; i < rows; i++) cost[(size_t)i * W] = i;
    for (int j = 0; j <= n; j++) cost[j] = j;
    std::vector<int> tc(n);
    for (int j = 0; j < n; j++) {
        const char c = t[j];
        tc[j] = (c == 'A' || c == 'a') ? 0 : (c == 'C' || c == 'c') ? 1 : (c == 'G' || c == 'g') ? 2
              : (c == 'T' || c == 't') ? 3 : 255;
    }
    std::string s((size_t)(n + k), 'A');
    static const char ALPHA[4] = {'A', 'C', 'G', 'T'};
    std::function<void(int, int)> rec = [&](int i, int min_cost) {
        if (cost[(size_t)i * W + n] <= k) emit(s.substr(0, (size_t)i), cost[(size_t)i * W + n], mat[(size_t)i * W + n]);
        if (!(min_cost <= k && i < n + k)) return;        // no extension can come back below k (:866-870)
        for (int ch = 0; ch < 4; ch++) {
            s[(size_t)i] = ALPHA[ch];
            const int r = i + 1;
            int mc = 999999999;
            for (int j = std::max(1, r - k); j <= std::min(n, r + k); j++) {
                const int mism = tc[j - 1] == ch ? 0 : 1;
                const int diag = cost[(size_t)(r - 1) * W + j - 1] + mism;
                const int left = cost[(size_t)r * W + j - 1] + 1;
                const int up = cost[(size_t)(r - 1) * W + j] + 1;
                int c, m;
                if (diag <= left && diag <= up) { c = diag; m = mat[(size_t)(r - 1) * W + j - 1] + (1 - mism); }
                else if (left <= up) { c = left; m = mat[(size_t)r * W + j - 1]; }
                else { c = up; m = mat[(size_t)(r - 1) * W + j]; }
                cost[(size_t)r * W + j] = c;
                mat[(size_t)r * W + j] = m;
                mc = std::min(mc, c);
            }
            rec(r, mc);
        }
    };
    rec(0, 0);
}

// all strings at Hamming distance exactly k from s over ACGT (_align.pyx:717-781)
template <typename F>
void hamming_sphere(const std::string& s, int k, F&& emit) {
    static const char ALPHA[4] = {'A', 'C', 'G', 'T'};
    std::string cur = s;
    std::function<void(int, int)> rec = [&](int start, int left) {
        if (left == 0) { emit(cur); return; }
        for (int i = start; i + left <= (int)s.size(); i++) {
            const char orig = cur[(size_t)i];
            for (char ch : ALPHA) {
                if (ch == s[(size_t)i]) continue;
                cur[(size_t)i] = ch;
                rec(i + 1, left - 1);
            }
            cur[(size_t)i] = orig;
        }
    };
    rec(0, k);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------
struct IndexArgs {
    const IdxEntry* table;
    const IdxWideEntry* wtable;
    const uint8_t* pool;
    uint64_t mask;
    const int32_t* lengths;
    int n_lengths;
    int prefix;
    const IdxAdapter* adapters;
    const uint8_t* adapter_seqs;
    const IdxKmerSet* sets;
    const IdxKmer* kmers;
    const uint8_t* kmer_chars;
    const uint8_t* seqs;
    const int64_t* offsets;
    const int32_t* lens;
    int64_t n_reads;
    int32_t* out6;
    int32_t* best_adapter;
    uint8_t* status;
};

__device__ __forceinline__ uint8_t dev_upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

// Scalar restatement of Aligner.locate (_align.pyx:298-587) for one lane: no wildcards (bytes
// compared for equality after upper-casing the query, :322-328), unit indel cost, min_overlap = m
// (anchored adapters), flags = QUERY_STOP (PrefixAdapter) or QUERY_START (SuffixAdapter).
// Returns true and (score, errors) for a match.  C, S, O: the column (m + 1 entries each) -- scratch
// memory in the packed kernel, the wave's LDS in the wide one: this is the rare 'N in the affix' path.
template <class IntPtr>
__device__ bool dev_locate_anchored(const uint8_t* ref, const int m, const double rate, const bool is_prefix,
                                    const uint8_t* query, const int n, int& out_score, int& out_errors,
                                    IntPtr C, IntPtr S, IntPtr O) {
    const bool start_in_query = !is_prefix, stop_in_query = is_prefix;      // Where.SUFFIX = 2, Where.PREFIX = 8
    const int k = (int)(rate * m);                                          // :343
    int max_n = n, min_n = 0;
    if (!start_in_query) max_n = min(n, m + k);                             // :348-350
    if (!stop_in_query) min_n = max(0, n - m - k);                          // :351-352
    for (int i = 0; i <= m; i++) {                                          // :364-383, start_in_reference = false
        if (!start_in_query) { S[i] = -2 * i; C[i] = max(i, min_n); O[i] = 0; }
        else { S[i] = -2 * i; C[i] = i; O[i] = max(0, min_n - i); }
    }
    const int SENTINEL = m + n + 1;                                         // :394
    int b_refstop = m, b_cost = SENTINEL, b_origin = 0, b_score = 0;
    int last = min(m, k + 1);                                               // :399
    int last_filled = 0;
    int cost = 0, score = 0, origin = 0;
    for (int j = min_n + 1; j <= max_n; j++) {                              // :433
        int dc = C[0], ds = S[0], dor = O[0];
        if (start_in_query) O[0] += 1; else { C[0] += 1; S[0] -= 2; }       // :413-415
        const uint8_t q = dev_upper(query[j - 1]);
        for (int i = 1; i <= last; i++) {
            if (ref[i - 1] == q) { cost = dc; origin = dor; score = ds + 1; }                 // :446-453
            else {
                const int c_diag = dc + 1, c_ins = C[i] + 1, c_del = C[i - 1] + 1;            // :455-476
                if (c_diag <= c_del && c_diag <= c_ins) { cost = c_diag; origin = dor; score = ds - 1; }
                else if (c_del <= c_ins) { cost = c_del; origin = O[i - 1]; score = S[i - 1] - 2; }
                else { cost = c_ins; origin = O[i]; score = S[i] - 2; }
            }
            dc = C[i]; ds = S[i]; dor = O[i];
            C[i] = cost; O[i] = origin; S[i] = score;
        }
        last_filled = last;                                                 // :484
        while (last >= 0 && C[last] > k) last--;                            // :490-491
        if (last < m) last++;
        else if (stop_in_query) {                                           // :496-533
            cost = C[m]; score = S[m]; origin = O[m];
            const int length = m + min(origin, 0);
            const bool ok = length >= m && (double)cost <= length * rate;
            const int best_len = m + min(b_origin, 0);
            if (ok && (b_cost == SENTINEL || (origin <= b_origin + m / 2 && score > b_score)
                       || (length > best_len && score > b_score))) {
                b_score = score; b_cost = cost; b_origin = origin; b_refstop = m;
                if (cost == 0 && origin >= 0) break;
            }
        }
    }
    if (max_n == n) {                                                       // :536-572, stop_in_reference = false
        for (int i = last_filled; i >= m; i--) {
            const int length = i + min(O[i], 0);
            cost = C[i]; score = S[i];
            const bool ok = length >= m && (double)cost <= length * rate;
            const int best_len = b_refstop + min(b_origin, 0);
            if (ok && (b_cost == SENTINEL || (origin <= b_origin + m / 2 && score > b_score)
                       || (length > best_len && score > b_score))) {
                b_score = score; b_cost = cost; b_origin = O[i]; b_refstop = i;
            }
        }
    }
    if (b_cost == SENTINEL) return false;
    out_score = b_score;
    out_errors = b_cost;
    return true;
}

// PrefixComparer / SuffixComparer (_align.pyx:651-714), no wildcards, min_overlap = m
__device__ bool dev_compare_anchored(const uint8_t* ref, const int m, const int max_k, const bool is_prefix,
                                     const uint8_t* query, const int n, int& out_score, int& out_errors) {
    const int length = min(m, n);
    int errors = 0;
    for (int i = 0; i < length; i++) {
        const uint8_t q = dev_upper(is_prefix ? query[i] : query[n - 1 - i]);
        const uint8_t r = is_prefix ? ref[i] : ref[m - 1 - i];
        errors += r != q;
    }
    if (errors > max_k || length < m) return false;                          // :690-691
    out_score = length - 2 * errors;                                         // :692
    out_errors = errors;
    return true;
}

// KmerFinder.kmers_present (_kmer_finder.pyx:170-213) for one adapter on one short string, without
// wildcards (characters equal after upper-casing): is any k-mer of any search set inside its window?
// The adapter's match_to() asks this before aligning (adapters.py:707-724), and for anchored
// adapters with indels the heuristic is NOT lossless, so it has to be reproduced.
__device__ bool dev_kmers_present(const IndexArgs& a, const IdxAdapter& A, const uint8_t* query, const int n) {
    for (int si = A.set_beg; si < A.set_end; si++) {
        const IdxKmerSet ks = a.sets[si];
        int start = ks.start, stop = ks.stop;                               // :188-204
        if (start < 0) { start += n; if (start < 0) start = 0; }
        else if (start > n) continue;
        if (stop < 0) { stop += n; if (stop <= 0) continue; }
        else if (stop == 0) stop = n;
        if (stop > n) stop = n;                                             // (the reference reads past the end here)
        if (stop - start <= 0) continue;
        for (int ki = ks.kmer_beg; ki < ks.kmer_end; ki++) {
            const IdxKmer km = a.kmers[ki];
            const uint8_t* kc = a.kmer_chars + km.off;
            for (int p = start; p + km.len <= stop; p++) {
                int t = 0;
                while (t < km.len && kc[t] == dev_upper(query[p + t])) t++;
                if (t == km.len) return true;
            }
        }
    }
    return false;
}

__device__ __forceinline__ bool idx_probe(const IdxEntry* table, const uint64_t mask, const uint64_t lo,
                                          const uint64_t hi, const uint32_t len, uint32_t& val) {
    uint64_t h = idx_hash(lo, hi, len) & mask;
    for (;;) {
        const IdxEntry e = table[h];
        if (e.len == IDX_EMPTY) return false;
        if (e.len == len && e.lo == lo && e.hi == hi) { val = e.val; return true; }
        h = (h + 1) & mask;
    }
}

// One read per lane.  The affix (prefix, or suffix walked backwards) of the longest indexed length
// is packed once; shorter affixes are bit masks of it (AdapterIndex._match_to_multiple_lengths,
// adapters.py:1487-1530, and _match_to_one_length :1468-1485).
__global__ __launch_bounds__(256) void k_index_lookup(IndexArgs a) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_reads) return;
    const int64_t off = a.offsets[r];
    const int64_t n64 = a.lens ? (int64_t)a.lens[r] : a.offsets[r + 1] - off;
    const int n = (int)(n64 > CAH_MAX_READ_LEN ? 0 : n64);
    const uint8_t* q = a.seqs + off;
    const bool prefix = a.prefix != 0;
    const int lmax = a.lengths[0];
    const int la = min(lmax, n);
    uint64_t lo = 0, hi = 0;
    int first_n = IDX_MAX_STRING + 1, first_bad = IDX_MAX_STRING + 1;
    bool non_ascii = n64 > CAH_MAX_READ_LEN;
    // the first 16 characters from the anchored end arrive in one (unaligned) 16-byte load
    struct __attribute__((packed, aligned(1))) U16 { uint32_t w[4]; };
    U16 c16 = {{0, 0, 0, 0}};
    const bool have16 = n >= 16;
    if (have16) c16 = *reinterpret_cast<const U16*>(prefix ? q : q + (n - 16));
    for (int t = 0; t < la; t++) {
        uint8_t raw;
        if (have16 && t < 16) {
            const int bi = prefix ? t : 15 - t;
            raw = (uint8_t)(c16.w[bi >> 2] >> ((bi & 3) * 8));
        } else {
            raw = prefix ? q[t] : q[n - 1 - t];
        }
        non_ascii |= raw >= 0x80;
        const uint8_t c = dev_upper(raw);
        int code = 0;
        if (c == 'C') code = 1;
        else if (c == 'G') code = 2;
        else if (c == 'T') code = 3;
        else if (c == 'N') first_n = min(first_n, t);                        // looked up as 'A' (:1535)
        else if (c != 'A') first_bad = min(first_bad, t);
        if (t < 32) lo |= (uint64_t)code << (2 * t);
        else hi |= (uint64_t)code << (2 * (t - 32));
    }
    const bool single = a.n_lengths == 1;
    int best_m = -1, best_e = 1000, best_len = 0, best_ad = -1;
    for (int li = 0; li < a.n_lengths; li++) {
        const int L = a.lengths[li];
        if (!single && L < best_m) break;                                    // :1503-1505
        const int lq = min(L, n);                                            // s[:L] / s[-L:] of a shorter read
        if (first_bad < lq) continue;                                        // not in the dictionary
        const uint64_t klo = lq >= 32 ? lo : (lo & ((1ull << (2 * lq)) - 1ull));
        const uint64_t khi = lq <= 32 ? 0ull : (lq >= 64 ? hi : (hi & ((1ull << (2 * (lq - 32))) - 1ull)));
        uint32_t val;
        if (!idx_probe(a.table, a.mask, klo, khi, (uint32_t)lq, val)) continue;
        int ad = (int)(val >> 12), e = (int)((val >> 8) & 0xF), m = (int)(val & 0xFF);
        if (first_n < lq) {
            // the looked-up counts assume 'A' where the read has 'N': redo the alignment (:1543-1551)
            const IdxAdapter A = a.adapters[ad];
            const uint8_t* affix = prefix ? q : q + (n - lq);
            int sc = 0, er = 0;
            if (A.set_end >= 0 && !dev_kmers_present(a, A, affix, lq)) continue;     // match_to() -> None
            int C[IDX_MAX_ADAPTER + 1], S[IDX_MAX_ADAPTER + 1], O[IDX_MAX_ADAPTER + 1];
            const bool ok = A.indels ? dev_locate_anchored(a.adapter_seqs + A.off, A.m, A.rate, prefix, affix, lq, sc, er,
                                                           &C[0], &S[0], &O[0])
                                     : dev_compare_anchored(a.adapter_seqs + A.off, A.m, A.max_k, prefix, affix, lq, sc, er);
            if (!ok) continue;
            e = er; m = sc;
        }
        if (single) { best_ad = ad; best_m = m; best_e = e; best_len = L; break; }
        if (m > best_m || (m == best_m && e < best_e)) { best_ad = ad; best_e = e; best_m = m; best_len = L; }
    }
    int32_t* o = a.out6 + r * 6;
    if (non_ascii) { a.status[r] = CAH_INVALID; if (a.best_adapter) a.best_adapter[r] = -1; return; }
    if (best_ad < 0) { a.status[r] = CAH_NONE; if (a.best_adapter) a.best_adapter[r] = -1; return; }
    // _make_prefix_match / _make_suffix_match (:1345-1371): the INDEXED length is reported, even for
    // a shorter read
    o[0] = 0;
    o[1] = a.adapters[best_ad].m;
    o[2] = prefix ? 0 : n - best_len;
    o[3] = prefix ? best_len : n;
    o[4] = best_m;
    o[5] = best_e;
    a.status[r] = CAH_MATCH;
    if (a.best_adapter) a.best_adapter[r] = best_ad;
}

// The wide layout (see the header).  One read per lane; the read is walked once from its anchored end, the
// FNV-1a state after t characters is the key hash of the affix of length t, so every indexed length is probed
// on the way -- shortest first.  The reference tries the longest first and keeps a later candidate only if it
// is strictly better (more matches, or as many with fewer errors; :1516-1524), its early exit (:1503-1505) skips
// lengths that cannot have as many matches as the best: the winner is the LONGEST length among the best
// (matches, errors), which going upwards is "replace unless strictly worse".
// Reads shorter than an indexed length L are looked up whole (s[:L] / s[-L:] of a shorter string) and still
// reported with the length L (_make_*_match, :1345-1371).
// 'N' re-alignment: the column arrays of Aligner.locate need m + 1 entries; they live in LDS, one set per wave,
// and the lanes that need them take turns (reads with 'N' in the affix are rare).
constexpr int IDX_WIDE_COLS = IDX_WIDE_MAX_ADAPTER + 1;

__global__ __launch_bounds__(256) void k_index_lookup_wide(IndexArgs a) {
    __shared__ int s_cols[4][3][IDX_WIDE_COLS];
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_reads) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t off = a.offsets[r];
    const int64_t n64 = a.lens ? (int64_t)a.lens[r] : a.offsets[r + 1] - off;
    const int n = (int)(n64 > CAH_MAX_READ_LEN ? 0 : n64);
    const uint8_t* q = a.seqs + off;
    const bool prefix = a.prefix != 0;
    const int lmax = a.lengths[0];
    const int la = min(lmax, n);
    bool non_ascii = n64 > CAH_MAX_READ_LEN;
    int first_n = 1 << 30;
    int best_m = -1, best_e = 1000, best_len = 0, best_ad = -1;
    uint64_t state = IDX_FNV_OFFSET;
    int li = a.n_lengths - 1;                                                // shortest indexed length
    // the character at distance t from the anchored end as the dictionary sees it: upper case (:1471, :1492),
    // 'N' looked up as 'A' (:1535)
    auto key_char = [&](const int t) -> uint8_t {
        const uint8_t c = dev_upper(prefix ? q[t] : q[n - 1 - t]);
        return c == 'N' ? (uint8_t)'A' : c;
    };
    auto probe = [&](const int lq, const int L) {
        const uint64_t h = idx_wide_hash(state, (uint32_t)lq);
        uint64_t slot = h & a.mask;
        for (;;) {
            const IdxWideEntry e = a.wtable[slot];
            if (e.len == IDX_EMPTY) return;
            if (e.hash == h && e.len == (uint32_t)lq) {
                const uint8_t* key = a.pool + e.off;
                int t = 0;
                while (t < lq && key[t] == key_char(t)) t++;
                if (t == lq) {
                    int ad = (int)e.adapter, er = (int)(e.em >> 16), m = (int)(e.em & 0xFFFF);
                    bool ok = true;
                    bool redo = first_n < lq;        // the counts assume 'A' where the read has 'N' (:1543-1551)
                    while (redo) {
                        if (lane == __builtin_amdgcn_readfirstlane(lane)) {
                            const IdxAdapter A = a.adapters[ad];
                            const uint8_t* affix = prefix ? q : q + (n - lq);
                            int sc = 0, e2 = 0;
                            if (A.set_end >= 0 && !dev_kmers_present(a, A, affix, lq)) ok = false;   // match_to() -> None
                            else
                                ok = A.indels ? dev_locate_anchored(a.adapter_seqs + A.off, A.m, A.rate, prefix, affix, lq, sc, e2,
                                                                    &s_cols[wave][0][0], &s_cols[wave][1][0], &s_cols[wave][2][0])
                                              : dev_compare_anchored(a.adapter_seqs + A.off, A.m, A.max_k, prefix, affix, lq, sc, e2);
                            er = e2; m = sc;
                            redo = false;
                        }
                    }
                    if (ok && (m > best_m || (m == best_m && er <= best_e))) { best_ad = ad; best_e = er; best_m = m; best_len = L; }
                    return;
                }
            }
            slot = (slot + 1) & a.mask;
        }
    };
    for (int t = 0;; t++) {
        while (li >= 0 && a.lengths[li] == t) { probe(t, t); li--; }
        if (t >= la) break;
        const uint8_t raw = prefix ? q[t] : q[n - 1 - t];
        non_ascii |= raw >= 0x80;
        const uint8_t c = dev_upper(raw);
        if (c == 'N') first_n = min(first_n, t);
        state = (state ^ (uint64_t)(c == 'N' ? (uint8_t)'A' : c)) * IDX_FNV_PRIME;
    }
    // the indexed lengths beyond the read's own: all look the whole read up, the longest reports
    if (li >= 0) probe(n, a.lengths[0]);
    int32_t* o = a.out6 + r * 6;
    if (non_ascii) { a.status[r] = CAH_INVALID; if (a.best_adapter) a.best_adapter[r] = -1; return; }
    if (best_ad < 0) { a.status[r] = CAH_NONE; if (a.best_adapter) a.best_adapter[r] = -1; return; }
    o[0] = 0;
    o[1] = a.adapters[best_ad].m;
    o[2] = prefix ? 0 : n - best_len;
    o[3] = prefix ? best_len : n;
    o[4] = best_m;
    o[5] = best_e;
    a.status[r] = CAH_MATCH;
    if (a.best_adapter) a.best_adapter[r] = best_ad;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
namespace {
struct Val { int32_t adapter, errors, matches; };

int index_on_device(const cah_index* ix, const IdxDeviceCopy** out) {
    int device = 0;
    IDX_TRY(hipGetDevice(&device));
    if (device < 0 || device >= IDX_MAX_DEVICES) return cah_set_error_(CAH_EUNSUPPORTED, "device index too large");
    std::lock_guard<std::mutex> lk(ix->mu);
    IdxDeviceCopy& dc = ix->dev[device];
    if (!dc.ready) {
        hipDeviceProp_t prop;
        IDX_TRY(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return cah_set_error_(CAH_EUNSUPPORTED, "this library is built for gfx950 (MI355X) only");
        dc.n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (ix->wide) {
            IDX_TRY(hipMalloc((void**)&dc.d_wtable, sizeof(IdxWideEntry) * ix->wtable.size()));
            IDX_TRY(hipMemcpy(dc.d_wtable, ix->wtable.data(), sizeof(IdxWideEntry) * ix->wtable.size(), hipMemcpyHostToDevice));
            IDX_TRY(hipMalloc((void**)&dc.d_pool, ix->pool.size() + 1));
            IDX_TRY(hipMemcpy(dc.d_pool, ix->pool.data(), ix->pool.size(), hipMemcpyHostToDevice));
        } else {
            IDX_TRY(hipMalloc((void**)&dc.d_table, sizeof(IdxEntry) * ix->table.size()));
            IDX_TRY(hipMemcpy(dc.d_table, ix->table.data(), sizeof(IdxEntry) * ix->table.size(), hipMemcpyHostToDevice));
        }
        IDX_TRY(hipMalloc((void**)&dc.d_lengths, sizeof(int32_t) * ix->lengths.size()));
        IDX_TRY(hipMemcpy(dc.d_lengths, ix->lengths.data(), sizeof(int32_t) * ix->lengths.size(), hipMemcpyHostToDevice));
        IDX_TRY(hipMalloc((void**)&dc.d_adapters, sizeof(IdxAdapter) * ix->adapters.size()));
        IDX_TRY(hipMemcpy(dc.d_adapters, ix->adapters.data(), sizeof(IdxAdapter) * ix->adapters.size(), hipMemcpyHostToDevice));
        IDX_TRY(hipMalloc((void**)&dc.d_seqs, ix->seqs.size() + 1));
        IDX_TRY(hipMemcpy(dc.d_seqs, ix->seqs.data(), ix->seqs.size(), hipMemcpyHostToDevice));
        IDX_TRY(hipMalloc((void**)&dc.d_sets, sizeof(IdxKmerSet) * (ix->sets.size() + 1)));
        IDX_TRY(hipMemcpy(dc.d_sets, ix->sets.data(), sizeof(IdxKmerSet) * ix->sets.size(), hipMemcpyHostToDevice));
        IDX_TRY(hipMalloc((void**)&dc.d_kmers, sizeof(IdxKmer) * (ix->kmers.size() + 1)));
        IDX_TRY(hipMemcpy(dc.d_kmers, ix->kmers.data(), sizeof(IdxKmer) * ix->kmers.size(), hipMemcpyHostToDevice));
        IDX_TRY(hipMalloc((void**)&dc.d_kmer_chars, ix->kmer_chars.size() + 1));
        IDX_TRY(hipMemcpy(dc.d_kmer_chars, ix->kmer_chars.data(), ix->kmer_chars.size(), hipMemcpyHostToDevice));
        dc.ready = true;
    }
    *out = &dc;
    return CAH_OK;
}
}  // namespace

extern "C" {

static int index_create_impl(const cah_index_adapter* adapters, int32_t n_adapters, int32_t prefix, cah_index** out);

int cah_index_create(const cah_index_adapter* adapters, int32_t n_adapters, int32_t prefix, cah_index** out) {
    try {                                                    // nothing may be thrown across the C ABI
        return index_create_impl(adapters, n_adapters, prefix, out);
    } catch (const std::bad_alloc&) {
        return cah_set_error_(CAH_ENOMEM, "cah_index_create: out of memory (the index of long adapters with 3 errors is large)");
    } catch (...) {
        return cah_set_error_(CAH_EINVAL, "cah_index_create: internal error");
    }
}

static int index_create_impl(const cah_index_adapter* adapters, int32_t n_adapters, int32_t prefix, cah_index** out) {
    if (!out) return cah_set_error_(CAH_EINVAL, "cah_index_create: out is NULL");
    *out = nullptr;
    if (!adapters || n_adapters <= 0) return cah_set_error_(CAH_EINVAL, "Adapter list is empty");
    if (n_adapters >= (1 << 20)) return cah_set_error_(CAH_EUNSUPPORTED, "more than 2^20 adapters in one index");
    std::unique_ptr<cah_index> ix(new cah_index());
    ix->prefix = prefix != 0;
    std::unordered_map<std::string, Val> index;
    std::unordered_set<std::string> ambiguous;
    std::set<int> lengths;
    for (int32_t a = 0; a < n_adapters; a++) {
        const cah_index_adapter& d = adapters[a];
        if (d.length <= 0 || !d.sequence) return cah_set_error_(CAH_EINVAL, "cah_index_create: empty adapter sequence");
        if (d.length > IDX_WIDE_MAX_ADAPTER) {
            char msg[128];
            snprintf(msg, sizeof(msg), "adapter %d: length %d exceeds the %d-character limit of the index", a, d.length, IDX_WIDE_MAX_ADAPTER);
            return cah_set_error_(CAH_EUNSUPPORTED, msg);
        }
        const std::string seq(d.sequence, (size_t)d.length);
        if (d.length > IDX_MAX_ADAPTER) ix->wide = true;
        for (char c : seq) {
            if ((unsigned char)c >= 0x80 || c == 0) return cah_set_error_(CAH_EINVAL, "cah_index_create: adapter sequences must be ASCII");
            if (code_of(c) < 0) ix->wide = true;
        }
        const int k = (int)(d.length * d.max_error_rate);                    // adapters.py:1383
        if (k > 3 || k < 0) return cah_set_error_(CAH_EINVAL, "Error rate too high");
        IdxAdapter ia;
        ia.off = (int32_t)ix->seqs.size(); ia.m = d.length; ia.indels = d.indels != 0; ia.rate = d.max_error_rate;
        ia.max_k = (int)(d.max_error_rate * d.length);
        ia.set_beg = (int32_t)ix->sets.size();
        ia.set_end = -1;
        if (d.n_kmer_sets >= 0) {
            if (d.n_kmer_sets > 0 && !d.kmer_sets) return cah_set_error_(CAH_EINVAL, "cah_index_create: kmer_sets is NULL");
            for (int32_t si = 0; si < d.n_kmer_sets; si++) {
                const cah_kmer_set& ks = d.kmer_sets[si];
                if (ks.start < -(1 << 20) || ks.start > (1 << 20) || ks.stop < -(1 << 20) || ks.stop > (1 << 20))
                    return cah_set_error_(CAH_EUNSUPPORTED, "cah_index_create: k-mer window out of range");
                IdxKmerSet out_set{(int32_t)ks.start, (int32_t)ks.stop, (int32_t)ix->kmers.size(), 0};
                for (int32_t ki = 0; ki < ks.n_kmers; ki++) {
                    const char* km = ks.kmers ? ks.kmers[ki] : nullptr;
                    if (!km) return cah_set_error_(CAH_EINVAL, "cah_index_create: NULL k-mer");
                    const size_t len = strlen(km);
                    if (len == 0 || len > 64) return cah_set_error_(CAH_EINVAL, "cah_index_create: k-mer length out of range");
                    ix->kmers.push_back(IdxKmer{(int32_t)ix->kmer_chars.size(), (int32_t)len});
                    for (size_t t = 0; t < len; t++) {
                        const char c = km[t];
                        ix->kmer_chars.push_back((c >= 'a' && c <= 'z') ? (char)(c - 32) : c);
                    }
                }
                out_set.kmer_end = (int32_t)ix->kmers.size();
                ix->sets.push_back(out_set);
            }
            ia.set_end = (int32_t)ix->sets.size();
        }
        ix->seqs += seq;
        ix->adapters.push_back(ia);
        auto store = [&](const std::string& s, int errors, int matches) -> bool {      // adapters.py:1425-1432
            auto it = index.find(s);
            if (it != index.end()) {
                if (matches < it->second.matches) return false;
                if (it->second.matches == matches && !ambiguous.count(s)) ambiguous.insert(s);
                it->second = Val{a, errors, matches};
            } else {
                index.emplace(s, Val{a, errors, matches});
            }
            return true;
        };
        const int kk = (int)(d.max_error_rate * (double)d.length);           // :1422 (same product, float order as written there)
        if (d.indels) {
            edit_environment(seq, kk, [&](const std::string& s, int e, int m) {
                if (store(s, e, m)) lengths.insert((int)s.size());
            });
        } else {
            for (int e = 0; e <= kk; e++)
                hamming_sphere(seq, e, [&](const std::string& s) { store(s, e, d.length - e); });
            lengths.insert(d.length);
        }
    }
    ix->n_ambiguous = (int32_t)ambiguous.size();
    for (const std::string& s : ambiguous) index.erase(s);                   // :1463-1464
    // (an adapter with a non-ACGT character, indels and k = 0 contributes no string at all: lengths may be empty)
    if ((int)lengths.size() > IDX_MAX_LENGTHS)
        return cah_set_error_(CAH_EUNSUPPORTED, "more than 64 different string lengths in one index");
    if (!ix->wide && !lengths.empty() && *lengths.rbegin() > IDX_MAX_STRING - 1)
        return cah_set_error_(CAH_EUNSUPPORTED, "index string lengths out of range");
    ix->lengths.assign(lengths.rbegin(), lengths.rend());                    // sorted, longest first (:1482)
    ix->n_strings = (int64_t)index.size();
    uint64_t cap = 16;
    while (cap < 2 * (uint64_t)index.size() + 2) cap <<= 1;
    ix->mask = cap - 1;
    if (ix->wide) {
        ix->wtable.assign(cap, IdxWideEntry{0, 0, IDX_EMPTY, 0, 0});
        size_t total = 0;
        for (const auto& kv : index) total += kv.first.size();
        if (total >= 0xFFFFFFFFull) return cah_set_error_(CAH_EUNSUPPORTED, "the index strings exceed 4 GiB");
        ix->pool.reserve(total + 1);
        for (const auto& kv : index) {
            const std::string& str = kv.first;
            const uint32_t len = (uint32_t)str.size(), off = (uint32_t)ix->pool.size();
            uint64_t st = IDX_FNV_OFFSET;
            for (uint32_t t = 0; t < len; t++) {
                const uint8_t c = (uint8_t)(ix->prefix ? str[t] : str[len - 1 - t]);
                ix->pool.push_back((char)c);
                st = (st ^ (uint64_t)c) * IDX_FNV_PRIME;
            }
            const uint64_t hsh = idx_wide_hash(st, len);
            uint64_t slot = hsh & ix->mask;
            while (ix->wtable[slot].len != IDX_EMPTY) slot = (slot + 1) & ix->mask;
            ix->wtable[slot] = IdxWideEntry{hsh, off, len, (uint32_t)kv.second.adapter,
                                            ((uint32_t)kv.second.errors << 16) | (uint32_t)kv.second.matches};
        }
        *out = ix.release();
        return CAH_OK;
    }
    ix->table.assign(cap, IdxEntry{0, 0, IDX_EMPTY, 0});
    for (const auto& kv : index) {
        uint64_t lo, hi;
        if (!pack_key(kv.first, ix->prefix, lo, hi)) return cah_set_error_(CAH_EINVAL, "internal: non-ACGT index string");
        const uint32_t len = (uint32_t)kv.first.size();
        uint64_t h = idx_hash(lo, hi, len) & ix->mask;
        while (ix->table[h].len != IDX_EMPTY) h = (h + 1) & ix->mask;
        ix->table[h] = IdxEntry{lo, hi, len,
                                ((uint32_t)kv.second.adapter << 12) | ((uint32_t)kv.second.errors << 8) | (uint32_t)kv.second.matches};
    }
    *out = ix.release();
    return CAH_OK;
}

void cah_index_destroy(cah_index* ix) {
    if (!ix) return;
    for (int d = 0; d < IDX_MAX_DEVICES; d++) {
        IdxDeviceCopy& dc = ix->dev[d];
        if (!dc.ready) continue;
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        if (hipSetDevice(d) == hipSuccess) {
            (void)hipFree(dc.d_table); (void)hipFree(dc.d_wtable); (void)hipFree(dc.d_pool); (void)hipFree(dc.d_lengths); (void)hipFree(dc.d_adapters); (void)hipFree(dc.d_seqs);
            (void)hipFree(dc.d_sets); (void)hipFree(dc.d_kmers); (void)hipFree(dc.d_kmer_chars);
        }
        if (cur >= 0) (void)hipSetDevice(cur);
    }
    delete ix;
}

int cah_index_info(const cah_index* ix, int64_t* n_strings, int32_t* n_ambiguous, int32_t* lengths, int32_t* n_lengths) {
    if (!ix) return cah_set_error_(CAH_EINVAL, "cah_index_info: index is NULL");
    if (n_strings) *n_strings = ix->n_strings;
    if (n_ambiguous) *n_ambiguous = ix->n_ambiguous;
    if (n_lengths) *n_lengths = (int32_t)ix->lengths.size();
    if (lengths) for (size_t i = 0; i < ix->lengths.size(); i++) lengths[i] = ix->lengths[i];
    return CAH_OK;
}

// host-side dictionary lookup of one string (tests, introspection): *found = 0/1
int cah_index_get(const cah_index* ix, const char* s, int32_t len, int32_t* found, int32_t* adapter,
                  int32_t* errors, int32_t* matches) {
    if (!ix || !found || (len > 0 && !s)) return cah_set_error_(CAH_EINVAL, "cah_index_get: NULL argument");
    *found = 0;
    if (len < 0) return CAH_OK;
    if (ix->wide) {
        uint64_t st = IDX_FNV_OFFSET;
        std::string key((size_t)len, 'A');
        for (int32_t t = 0; t < len; t++) {
            key[(size_t)t] = ix->prefix ? s[t] : s[len - 1 - t];
            st = (st ^ (uint64_t)(uint8_t)key[(size_t)t]) * IDX_FNV_PRIME;
        }
        const uint64_t hsh = idx_wide_hash(st, (uint32_t)len);
        for (uint64_t slot = hsh & ix->mask;; slot = (slot + 1) & ix->mask) {
            const IdxWideEntry& e = ix->wtable[slot];
            if (e.len == IDX_EMPTY) return CAH_OK;
            if (e.hash == hsh && e.len == (uint32_t)len && memcmp(ix->pool.data() + e.off, key.data(), (size_t)len) == 0) {
                *found = 1;
                if (adapter) *adapter = (int32_t)e.adapter;
                if (errors) *errors = (int32_t)(e.em >> 16);
                if (matches) *matches = (int32_t)(e.em & 0xFFFF);
                return CAH_OK;
            }
        }
    }
    if (len >= IDX_MAX_STRING) return CAH_OK;
    uint64_t lo, hi;
    if (!pack_key(std::string(s, (size_t)len), ix->prefix, lo, hi)) return CAH_OK;
    uint64_t h = idx_hash(lo, hi, (uint32_t)len) & ix->mask;
    for (;;) {
        const IdxEntry& e = ix->table[h];
        if (e.len == IDX_EMPTY) return CAH_OK;
        if (e.len == (uint32_t)len && e.lo == lo && e.hi == hi) {
            *found = 1;
            if (adapter) *adapter = (int32_t)(e.val >> 12);
            if (errors) *errors = (int32_t)((e.val >> 8) & 0xF);
            if (matches) *matches = (int32_t)(e.val & 0xFF);
            return CAH_OK;
        }
        h = (h + 1) & ix->mask;
    }
}

int cah_index_lookup_batch(const cah_index* ix, const uint8_t* d_seqs, const int64_t* d_offsets,
                           const int32_t* d_lens, int64_t n_reads, int32_t* d_out6, int32_t* d_best_adapter,
                           uint8_t* d_status, void* stream) {
    if (!ix) return cah_set_error_(CAH_EINVAL, "cah_index_lookup_batch: index is NULL");
    if (n_reads < 0) return cah_set_error_(CAH_EINVAL, "n_reads < 0");
    if (n_reads == 0) return CAH_OK;
    if (!d_offsets || !d_out6 || !d_status) return cah_set_error_(CAH_EINVAL, "cah_index_lookup_batch: NULL argument");
    if (ix->lengths.empty()) {                         // nothing indexed: the reference's loop over lengths finds nothing (:1502)
        IDX_TRY(hipMemsetAsync(d_status, CAH_NONE, (size_t)n_reads, (hipStream_t)stream));
        if (d_best_adapter) IDX_TRY(hipMemsetAsync(d_best_adapter, 0xFF, sizeof(int32_t) * (size_t)n_reads, (hipStream_t)stream));
        return CAH_OK;
    }
    const IdxDeviceCopy* dc = nullptr;
    int rc = index_on_device(ix, &dc);
    if (rc) return rc;
    IndexArgs a;
    a.table = dc->d_table; a.wtable = dc->d_wtable; a.pool = dc->d_pool; a.mask = ix->mask; a.lengths = dc->d_lengths; a.n_lengths = (int)ix->lengths.size();
    a.prefix = ix->prefix ? 1 : 0; a.adapters = dc->d_adapters; a.adapter_seqs = dc->d_seqs;
    a.sets = dc->d_sets; a.kmers = dc->d_kmers; a.kmer_chars = dc->d_kmer_chars;
    a.seqs = d_seqs; a.offsets = d_offsets; a.lens = d_lens; a.n_reads = n_reads;
    a.out6 = d_out6; a.best_adapter = d_best_adapter; a.status = d_status;
    const int64_t blocks = (n_reads + 255) / 256;
    if (ix->wide) hipLaunchKernelGGL(k_index_lookup_wide, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_index_lookup, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    IDX_TRY(hipGetLastError());
    return CAH_OK;
}

int cah_index_lookup_batch_host(const cah_index* ix, const uint8_t* seqs, const int64_t* offsets, int64_t n_reads,
                                int32_t* out6, int32_t* best_adapter, uint8_t* status) {
    if (!ix) return cah_set_error_(CAH_EINVAL, "cah_index_lookup_batch_host: index is NULL");
    if (n_reads < 0) return cah_set_error_(CAH_EINVAL, "n_reads < 0");
    if (n_reads == 0) return CAH_OK;
    if (!offsets || !out6 || !status) return cah_set_error_(CAH_EINVAL, "cah_index_lookup_batch_host: NULL argument");
    const int64_t total = offsets[n_reads];
    void *d_seqs = nullptr, *d_off = nullptr, *d_out = nullptr, *d_best = nullptr, *d_st = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_seqs); (void)hipFree(d_off); (void)hipFree(d_out); (void)hipFree(d_best); (void)hipFree(d_st); };
    hipError_t e = hipMalloc(&d_seqs, (size_t)(total > 0 ? total : 1));
    if (e == hipSuccess) e = hipMalloc(&d_off, sizeof(int64_t) * (size_t)(n_reads + 1));
    if (e == hipSuccess) e = hipMalloc(&d_out, sizeof(int32_t) * 6 * (size_t)n_reads);
    if (e == hipSuccess) e = hipMalloc(&d_best, sizeof(int32_t) * (size_t)n_reads);
    if (e == hipSuccess) e = hipMalloc(&d_st, (size_t)n_reads);
    if (e == hipSuccess && total > 0) e = hipMemcpy(d_seqs, seqs, (size_t)total, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_off, offsets, sizeof(int64_t) * (size_t)(n_reads + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_out, 0, sizeof(int32_t) * 6 * (size_t)n_reads);
    if (e != hipSuccess) { cleanup(); return cah_set_error_(CAH_EHIP, hipGetErrorString(e)); }
    int rc = cah_index_lookup_batch(ix, (const uint8_t*)d_seqs, (const int64_t*)d_off, nullptr, n_reads,
                                    (int32_t*)d_out, (int32_t*)d_best, (uint8_t*)d_st, nullptr);
    if (rc == CAH_OK) {
        e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(out6, d_out, sizeof(int32_t) * 6 * (size_t)n_reads, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(status, d_st, (size_t)n_reads, hipMemcpyDeviceToHost);
        if (e == hipSuccess && best_adapter) e = hipMemcpy(best_adapter, d_best, sizeof(int32_t) * (size_t)n_reads, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = cah_set_error_(CAH_EHIP, hipGetErrorString(e));
    }
    cleanup();
    return rc;
}

}  // extern "C"
