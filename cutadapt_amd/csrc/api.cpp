// api.cpp -- host side of libcutadapt_hip.so: plan construction (character tables, k-mer
// packing) and the C-ABI entry points declared in include/cutadapt_hip.h.
//
// Plan construction restates, in table form, what the reference does at object creation:
//   Aligner.__cinit__/_set_reference      reference src/cutadapt/_align.pyx:195-277
//   PrefixComparer.__init__               reference src/cutadapt/_align.pyx:615-642
//   KmerFinder.__cinit__                  reference src/cutadapt/_kmer_finder.pyx:106-165
//   character tables / match lists        reference src/cutadapt/_match_tables.py:4-98
// The per-read work (translate() + compare per DP cell) is folded into one 64-bit "row
// bitset" per read character, so the kernels never compare characters.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <chrono>
#include <atomic>
#include <string>
#include <memory>
#include <new>
#include <vector>

#include "../../include/cutadapt_hip.h"
#include "cah_device.h"
#include "kernels.h"
#include "multi2.h"
#include "back_scan.h"

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

// for the other translation units of the library
int cah_set_error_(int code, const char* msg) { return fail(code, "%s", msg); }

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess)                                                           \
            return fail(CAH_EHIP, "%s failed: %s", #expr, hipGetErrorString(e__));       \
    } while (0)

// ---------------------------------------------------------------------------------------------
// character tables (_match_tables.py)
// ---------------------------------------------------------------------------------------------
namespace {

struct CharTables {
    uint8_t acgt[256], iupac[256], upper[256];
    CharTables() {
        // A=1 C=2 G=4 T=U=8, everything else 0x80 (_match_tables.py:4-17)
        memset(acgt, 0x80, sizeof(acgt));
        const char* bases = "ACGTU";
        const uint8_t codes[5] = {1, 2, 4, 8, 8};
        for (int i = 0; i < 5; i++) both(acgt, bases[i], codes[i]);
        // IUPAC nibble sets; N additionally has 0x80 so that it matches non-ACGT characters
        // encoded with the ACGT table; X and unknown characters are 0 (:20-61)
        memset(iupac, 0, sizeof(iupac));
        struct { char c; uint8_t v; } codes_iupac[] = {
            {'X', 0}, {'A', 1}, {'C', 2}, {'G', 4}, {'T', 8}, {'U', 8}, {'R', 1 | 4}, {'Y', 2 | 8},
            {'S', 4 | 2}, {'W', 1 | 8}, {'K', 4 | 8}, {'M', 1 | 2}, {'B', 2 | 4 | 8},
            {'D', 1 | 4 | 8}, {'H', 1 | 2 | 8}, {'V', 1 | 2 | 4}, {'N', 0x8F}};
        for (auto& e : codes_iupac) both(iupac, e.c, e.v);
        for (int i = 0; i < 256; i++) upper[i] = (uint8_t)((i >= 'a' && i <= 'z') ? i - 32 : i);
    }
    static void both(uint8_t* t, char c, uint8_t v) {
        t[(uint8_t)c] = v;
        t[(uint8_t)(c + 32)] = v;   // lower case
    }
};

const CharTables& tables() {
    static CharTables t;
    return t;
}

bool is_ascii(const char* s, size_t n) {
    for (size_t i = 0; i < n; i++)
        if ((uint8_t)s[i] & 0x80) return false;
    return true;
}

// Does read character `qc` match adapter character `rc` inside Aligner.locate?
// reference encoding (_align.pyx:272-276) x query encoding (:322-328) x compare (:442-445).
// NB: without wildcards the adapter is compared as given (NOT upper-cased), the read is.
bool aligner_chars_match(uint8_t rc, uint8_t qc, bool wildcard_ref, bool wildcard_query) {
    const CharTables& t = tables();
    if (wildcard_query) {
        const uint8_t r = wildcard_ref ? t.iupac[rc] : t.acgt[rc];
        return (r & t.iupac[qc]) != 0;
    }
    if (wildcard_ref) return (t.iupac[rc] & t.acgt[qc]) != 0;
    return rc == t.upper[qc];
}

// PrefixComparer encodes its reference with the upper table when no wildcards are used
// (_align.pyx:637-642), unlike Aligner.
bool comparer_chars_match(uint8_t rc, uint8_t qc, bool wildcard_ref, bool wildcard_query) {
    const CharTables& t = tables();
    if (wildcard_query) {
        const uint8_t r = wildcard_ref ? t.iupac[rc] : t.acgt[rc];
        return (r & t.iupac[qc]) != 0;
    }
    if (wildcard_ref) return (t.iupac[rc] & t.acgt[qc]) != 0;
    return t.upper[rc] == t.upper[qc];
}

// KmerFinder's match lists (_match_tables.py:81-98; NUL never matches, :73-78)
bool kmer_chars_match(uint8_t rc, uint8_t qc, bool ref_wc, bool query_wc) {
    const CharTables& t = tables();
    if (qc == 0 || qc >= 128) return false;
    if (!ref_wc && !query_wc) return t.upper[rc] == t.upper[qc];
    if (ref_wc && !query_wc) return (t.iupac[rc] & t.acgt[qc]) != 0;
    if (!ref_wc && query_wc) return (t.acgt[rc] & t.iupac[qc]) != 0;
    return (t.iupac[rc] & t.iupac[qc]) != 0;
}

bool env_flag_early(const char* name) { const char* e = getenv(name); return e && *e && *e != '0'; }

// CAH_NO_SCAN=1 (read at plan creation): the plan is built without the cost scan (A/B measurements,
// parity tests of the scan against the plain cell kernel)
bool scan_disabled() {
    const char* e = getenv("CAH_NO_SCAN");
    return e && *e && *e != '0';
}

int clamp_floor(double x) {
    if (!(x == x)) return -1;            // NaN: `cost <= NaN` is false in the reference
    double f = std::floor(x);
    if (f > 2147483000.0) return 2147483000;
    if (f < -2147483000.0) return -2147483000;
    return (int)f;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------
#define CAH_MAX_DEVICES 64

// adapters longer than CAH_MAX_M characters (long.hip)
struct LongTables {
    CahLongMatcher lm;
    std::vector<uint8_t> ref;
    std::vector<int32_t> ncnt;
};
struct LongDeviceCopy {
    CahLongMatcher* d_lm = nullptr;
    uint8_t* d_ref = nullptr;
    int32_t* d_ncnt = nullptr;
};

struct PlanDeviceCopy {
    bool ready = false;
    int n_cus = 256;
    CahMatcher* d_matchers = nullptr;     // HBM
    CahKmerWord* d_words = nullptr;
    CahLeanFilter* d_lean = nullptr;      // one per matcher (ok = 0 where the lean prefilter does not apply)
    // fused multi-adapter path (hdr.ok only)
    CahMultiHeader* d_mhdr = nullptr;
    CahMultiDir* d_mdir = nullptr;
    CahMultiEntry* d_mentries = nullptr;
    uint32_t* d_mbitmap = nullptr;
    uint64_t* d_mscan = nullptr;          // [n_adapters][CAH_MULTI_TAB_STRIDE] padded match words (cost scan)
    uint64_t* d_mrow = nullptr;           // ... row bitsets (cell DP)
    // ... its streaming form (multi2.h; m2.hdr.ok only)
    CahMulti2Header* d_m2hdr = nullptr;
    uint16_t* d_m2dir = nullptr;
    CahM2Slot* d_m2entries = nullptr;
    uint32_t* d_m2bitmap = nullptr;
    uint32_t* d_m2prefix = nullptr;
    int32_t* d_m2refbegin = nullptr;
    uint32_t* d_m2reflist = nullptr;
    std::vector<LongDeviceCopy> d_long;   // one per matcher (null pointers for bare k-mer finders)
    // the one-read kernel's table images ([0]: Aligner.locate, no prefilter; [1]: match_to), built by its first call
    void* d_tiny_image[2] = {nullptr, nullptr};
};

// host tables of the fused multi-adapter path (see CahMultiHeader)
struct MultiPlan {
    CahMultiHeader hdr;
    std::vector<CahMultiDir> dir;
    std::vector<CahMultiEntry> entries;
    std::vector<uint32_t> bitmap;
    std::vector<uint64_t> scan_tab, row_tab;
    M2Tables m2;                          // the streaming form's tables (m2.hdr.ok: equally long short reads take it)
    MultiPlan() { memset(&hdr, 0, sizeof(hdr)); memset(&m2.hdr, 0, sizeof(m2.hdr)); }
};

// Host tables are built (and validated) at creation; the HBM copy for a device is made the
// first time the plan is used on that device, so one plan serves every GPU of the node.
struct cah_plan {
    std::vector<CahMatcher> matchers;     // host copies
    std::vector<CahKmerWord> words;
    std::vector<CahLeanFilter> lean;
    MultiPlan multi;
    std::vector<LongTables> long_tabs;    // one per matcher (empty tables unless long_dp)
    int max_long_m = 0;                   // longest long_dp adapter (0: none): sizes the column scratch
    mutable std::mutex mu;
    mutable PlanDeviceCopy dev[CAH_MAX_DEVICES];
};

// returns the copy of the plan's tables on the current device (uploading on first use)
static int plan_on_device(const cah_plan* plan, const PlanDeviceCopy** out) {
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    if (device < 0 || device >= CAH_MAX_DEVICES) return fail(CAH_EUNSUPPORTED, "device index %d too large", device);
    std::lock_guard<std::mutex> lk(plan->mu);
    PlanDeviceCopy& dc = plan->dev[device];
    if (!dc.ready) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(CAH_EUNSUPPORTED, "device %d is %s; this library is built for gfx950 (MI355X) only",
                        device, prop.gcnArchName);
        dc.n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        HIP_TRY(hipMalloc((void**)&dc.d_matchers, sizeof(CahMatcher) * plan->matchers.size()));
        HIP_TRY(hipMemcpy(dc.d_matchers, plan->matchers.data(), sizeof(CahMatcher) * plan->matchers.size(),
                          hipMemcpyHostToDevice));
        if (!plan->words.empty()) {
            HIP_TRY(hipMalloc((void**)&dc.d_words, sizeof(CahKmerWord) * plan->words.size()));
            HIP_TRY(hipMemcpy(dc.d_words, plan->words.data(), sizeof(CahKmerWord) * plan->words.size(),
                              hipMemcpyHostToDevice));
        }
        HIP_TRY(hipMalloc((void**)&dc.d_lean, sizeof(CahLeanFilter) * plan->lean.size()));
        HIP_TRY(hipMemcpy(dc.d_lean, plan->lean.data(), sizeof(CahLeanFilter) * plan->lean.size(), hipMemcpyHostToDevice));
        dc.d_long.resize(plan->matchers.size());
        for (size_t i = 0; i < plan->matchers.size(); i++) {
            if (!plan->matchers[i].long_dp) continue;
            const LongTables& lt = plan->long_tabs[i];
            LongDeviceCopy& ld = dc.d_long[i];
            HIP_TRY(hipMalloc((void**)&ld.d_lm, sizeof(CahLongMatcher)));
            HIP_TRY(hipMemcpy(ld.d_lm, &lt.lm, sizeof(CahLongMatcher), hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc((void**)&ld.d_ref, std::max<size_t>(lt.ref.size(), 1)));
            if (!lt.ref.empty()) HIP_TRY(hipMemcpy(ld.d_ref, lt.ref.data(), lt.ref.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc((void**)&ld.d_ncnt, sizeof(int32_t) * lt.ncnt.size()));
            HIP_TRY(hipMemcpy(ld.d_ncnt, lt.ncnt.data(), sizeof(int32_t) * lt.ncnt.size(), hipMemcpyHostToDevice));
        }
        const MultiPlan& mp = plan->multi;
        if (mp.hdr.ok) {
#define CAH_UPLOAD(dst, vec)                                                                              \
            HIP_TRY(hipMalloc((void**)&(dst), sizeof((vec)[0]) * std::max<size_t>((vec).size(), 1)));      \
            if (!(vec).empty()) HIP_TRY(hipMemcpy((dst), (vec).data(), sizeof((vec)[0]) * (vec).size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc((void**)&dc.d_mhdr, sizeof(CahMultiHeader)));
            HIP_TRY(hipMemcpy(dc.d_mhdr, &mp.hdr, sizeof(CahMultiHeader), hipMemcpyHostToDevice));
            CAH_UPLOAD(dc.d_mdir, mp.dir)
            CAH_UPLOAD(dc.d_mentries, mp.entries)
            CAH_UPLOAD(dc.d_mbitmap, mp.bitmap)
            CAH_UPLOAD(dc.d_mscan, mp.scan_tab)
            CAH_UPLOAD(dc.d_mrow, mp.row_tab)
            if (mp.m2.hdr.ok) {
                HIP_TRY(hipMalloc((void**)&dc.d_m2hdr, sizeof(CahMulti2Header)));
                HIP_TRY(hipMemcpy(dc.d_m2hdr, &mp.m2.hdr, sizeof(CahMulti2Header), hipMemcpyHostToDevice));
                CAH_UPLOAD(dc.d_m2dir, mp.m2.dir)
                CAH_UPLOAD(dc.d_m2entries, mp.m2.entries)
                CAH_UPLOAD(dc.d_m2bitmap, mp.m2.bitmap)
                CAH_UPLOAD(dc.d_m2prefix, mp.m2.prefix)
                CAH_UPLOAD(dc.d_m2refbegin, mp.m2.ref_begin)
                CAH_UPLOAD(dc.d_m2reflist, mp.m2.ref_list)
            }
#undef CAH_UPLOAD
        }
        dc.ready = true;
    }
    *out = &dc;
    return CAH_OK;
}

// The lean prefilter of a matcher (see CahLeanFilter): possible when every search set is a whole-read
// set (0, None), a tail set (-L, None) or a head set (start, stop) within the first CAH_LEAN_SPAN
// characters, every k-mer fits 32 bits and the packing fits the largest kernel class.  Same matches as the
// generic packing: kmers_present is an OR over k-mers, a k-mer of (-L, None) is found iff it occurs with
// start >= n - L, which is what a gated start bit says.
static void build_lean_filter(const cah_adapter_desc& d, CahLeanFilter& lf) {
    memset(&lf, 0, sizeof(lf));
    if (d.n_kmer_sets <= 0 || !d.kmer_sets) return;
    const bool rwc = d.kmer_ref_wildcards != 0, qwc = d.kmer_query_wildcards != 0;
    struct Item { const char* kmer; int len; int L; int start, stop; };   // L: tail window; start/stop: head window
    std::vector<Item> lead, tail, head;
    for (int s = 0; s < d.n_kmer_sets; s++) {
        const cah_kmer_set& ks = d.kmer_sets[s];
        int kind;                                                 // 0 lead, 1 tail, 2 head
        if (ks.start == 0 && ks.stop == 0) kind = 0;
        else if (ks.start < 0 && ks.stop == 0 && ks.start >= -(int64_t)CAH_LEAN_SPAN) kind = 1;
        else if (ks.start >= 0 && ks.stop > ks.start && ks.stop <= CAH_LEAN_SPAN) kind = 2;
        else return;
        for (int t = 0; t < ks.n_kmers; t++) {
            const char* k = ks.kmers[t];
            if (!k) return;
            const size_t len = strlen(k);
            if (len == 0 || len > 32 || !is_ascii(k, len)) return;
            const Item it{k, (int)len, (int)-ks.start, (int)ks.start, (int)ks.stop};
            (kind == 0 ? lead : kind == 1 ? tail : head).push_back(it);
        }
    }
    std::stable_sort(tail.begin(), tail.end(), [](const Item& a, const Item& b) { return a.L > b.L; });
    // words a greedy packing of these k-mers needs when every k-mer takes `extra` more bits
    auto words_needed = [](const std::vector<Item>& items, int extra) {
        int w = 0, used = 32;
        for (const Item& it : items) {
            if (it.len + extra > 32) return 1 << 20;
            if (used + it.len + extra > 32) { ++w; used = 0; }
            used += it.len + extra;
        }
        return w;
    };
    const int delay = words_needed(lead, CAH_LEAN_DELAY) == words_needed(lead, 0) ? CAH_LEAN_DELAY : 0;
    int w = -1, used = 32, cap = 0;
    uint32_t (*mask)[CAH_TABLE_CHARS] = nullptr;
    auto place = [&](const Item& it, int kind) -> bool {
        const int extra = kind == 0 ? delay : 0;
        if (used + it.len + extra > 32) {
            if (++w >= cap) return false;
            used = 0;
        }
        const uint32_t start_bit = 1u << used, end_bit = 1u << (used + it.len - 1);
        for (int p = 0; p < it.len; p++)
            for (int qc = 0; qc < CAH_TABLE_CHARS; qc++)
                if (kmer_chars_match((uint8_t)it.kmer[p], (uint8_t)qc, rwc, qwc)) mask[w][qc] |= 1u << (used + p);
        if (kind == 0) {
            lf.lead_init[w] |= start_bit;
            lf.lead_found[w] |= end_bit;
            for (int j = 1; j <= delay; j++) {
                lf.lead_found[w] |= end_bit << j;
                lf.lead_pass[w] |= end_bit << j;
            }
        } else if (kind == 1) {
            // distance d = n - p from the read end: 1 is the last character; idx = CAH_GATE_ZERO - d
            for (int dist = 1; dist <= it.L; dist++) lf.gate_init[w][CAH_GATE_ZERO - dist] |= start_bit;
            lf.gated_found[w] |= end_bit;
            lf.gated_span[w] = std::max(lf.gated_span[w], (int32_t)it.L);
            lf.tail_span = std::max(lf.tail_span, it.L);
        } else {
            for (int p = it.start; p + it.len <= it.stop; p++) lf.gate_init[w][p] |= start_bit;
            lf.gated_found[w] |= end_bit;
            lf.head_span = std::max(lf.head_span, it.stop);
        }
        used += it.len + extra;
        return true;
    };
    cap = CAH_LEAN_MAX_LEAD; mask = lf.lead_mask;
    for (const Item& it : lead) if (!place(it, 0)) return;
    lf.n_lead = w + 1;
    lf.lead_delay = delay;
    w = -1; used = 32; cap = CAH_LEAN_MAX_GATED; mask = lf.gated_mask;
    for (const Item& it : tail) if (!place(it, 1)) return;
    lf.n_tail = w + 1;
    used = 32;                                                   // head words start a new word
    for (const Item& it : head) if (!place(it, 2)) return;
    lf.n_gated = w + 1;
    lf.ok = (lf.n_lead + lf.n_gated) >= 1 ? 1 : 0;
    // T-words (k_filter_stream2): the tail k-mers packed like lead k-mers (delay bits, ungated start bits); the
    // window becomes a condition on where a k-mer ENDS (tw_found, see CahLeanFilter).  Widest windows first, so
    // that a word's first k-mer names the word's span and the spans fall from word to word.
    if (lf.ok && head.empty() && (lf.n_lead == 0 || delay == CAH_LEAN_DELAY)) {
        int tw = -1, tused = 32;
        bool fits = true;
        for (const Item& it : tail) {
            const int bits = it.len + CAH_LEAN_DELAY;
            if (bits > 32) { fits = false; break; }
            if (tused + bits > 32) {
                if (++tw >= CAH_LEAN_MAX_TW) { fits = false; break; }
                tused = 0;
                lf.tw_span[tw] = it.L;
            }
            const uint32_t start_bit = 1u << tused, end_bit = 1u << (tused + it.len - 1);
            for (int p = 0; p < it.len; p++)
                for (int qc = 0; qc < CAH_TABLE_CHARS; qc++)
                    if (kmer_chars_match((uint8_t)it.kmer[p], (uint8_t)qc, rwc, qwc)) lf.tw_mask[tw][qc] |= 1u << (tused + p);
            lf.tw_init[tw] |= start_bit;
            for (int j = 1; j <= CAH_LEAN_DELAY; j++) lf.tw_pass[tw] |= end_bit << j;
            // the k-mer ended at t - d (t: last character of the group, d: delay bit) = n - 1 - dist - d; it lies in
            // its window iff 0 <= dist + d <= L - q
            for (int dist = -CAH_TW_DIST0; dist + CAH_TW_DIST0 < CAH_TW_DIST_LEN; dist++)
                for (int dd = 0; dd <= CAH_LEAN_DELAY; dd++)
                    if (dist + dd >= 0 && dist + dd <= it.L - it.len) lf.tw_found[tw][dist + CAH_TW_DIST0] |= end_bit << dd;
            tused += bits;
        }
        if (fits) { lf.n_tw = tw + 1; lf.tw_ok = 1; }
        else {
            lf.n_tw = 0;
            memset(lf.tw_span, 0, sizeof(lf.tw_span)); memset(lf.tw_init, 0, sizeof(lf.tw_init));
            memset(lf.tw_pass, 0, sizeof(lf.tw_pass)); memset(lf.tw_mask, 0, sizeof(lf.tw_mask));
            memset(lf.tw_found, 0, sizeof(lf.tw_found));
        }
    }
}

#define CAH_LONG_ADAPTER_LIMIT 100000     // sanity bound for the HBM column of k_dp_long (3 x 4 B per row and lane)

// m > CAH_MAX_M: constants + encoded adapter for k_dp_long (the reference's own representation:
// _align.pyx:250-277 for Aligner, :615-642 for the comparers)
static int build_long(const cah_adapter_desc& d, int index, CahMatcher& mt, LongTables& lt) {
    const CharTables& t = tables();
    const int m = d.length;
    const char* seq = d.sequence;
    const bool wr = d.wildcard_ref != 0, wq = d.wildcard_query != 0;
    const double rate = d.max_error_rate;
    if (m > CAH_LONG_ADAPTER_LIMIT)
        return fail(CAH_EUNSUPPORTED, "adapter %d: length %d exceeds %d characters", index, m, CAH_LONG_ADAPTER_LIMIT);
    CahLongMatcher& lm = lt.lm;
    memset(&lm, 0, sizeof(lm));
    lm.kind = d.kind; lm.m = m; lm.flags = d.flags & 15; lm.indel_cost = d.indel_cost; lm.min_overlap = d.min_overlap;
    lm.wildcard_ref = wr; lm.rate = rate;
    lt.ref.resize((size_t)m);
    lt.ncnt.assign((size_t)m + 1, 0);
    // read character -> encoding (:322-328); adapter encoding (:272-276, comparers :637-642)
    const uint8_t* qsrc = wq ? t.iupac : (wr ? t.acgt : t.upper);
    for (int c = 0; c < CAH_TABLE_CHARS; c++) lm.qtab[c] = qsrc[c];
    lm.cmp_equal = (!wq && !wr) ? 1 : 0;
    if (d.kind == CAH_KIND_ALIGNER) {
        int nn = 0;
        for (int i = 0; i < m; i++) { lt.ncnt[(size_t)i] = nn; if (seq[i] == 'N' || seq[i] == 'n') nn++; }
        lt.ncnt[(size_t)m] = nn;
        lm.effective_length = m;
        if (wr) {
            lm.effective_length = m - nn;
            if (lm.effective_length == 0) return fail(CAH_EINVAL, "Cannot have only N wildcards in the sequence");
        }
        if (d.indel_cost < 1) return fail(CAH_EINVAL, "indel_cost must be at least 1");
        if (d.indel_cost > CAH_MAX_INDEL_COST)
            return fail(CAH_EUNSUPPORTED, "indel_cost above %d is not supported", CAH_MAX_INDEL_COST);
        if (!(rate == rate) || std::fabs(rate) > 1e6) return fail(CAH_EINVAL, "max_error_rate is not a usable number");
        lm.k = (int)(rate * m);
        for (int i = 0; i < m; i++)
            lt.ref[(size_t)i] = wr ? t.iupac[(uint8_t)seq[i]] : (wq ? t.acgt[(uint8_t)seq[i]] : (uint8_t)seq[i]);   // raw, NOT upper-cased
    } else {
        int eff = m;
        if (wr) {
            int nN = 0, nn = 0;
            for (int i = 0; i < m; i++) { nN += seq[i] == 'N'; nn += seq[i] == 'n'; }
            eff -= nN - nn;                                                 // quirk kept (:628)
            if (eff == 0) return fail(CAH_EINVAL, "Cannot have only N wildcards in the sequence");
        }
        if (!(rate >= 0.0 && rate <= 1.0)) return fail(CAH_EINVAL, "max_error_rate must be between 0 and 1");
        if (d.min_overlap < 1) return fail(CAH_EINVAL, "min_overlap must be at least 1");
        lm.effective_length = eff;
        lm.cmp_max_k = (int)(rate * eff);
        for (int i = 0; i < m; i++)
            lt.ref[(size_t)i] = wr ? t.iupac[(uint8_t)seq[i]] : (wq ? t.acgt[(uint8_t)seq[i]] : t.upper[(uint8_t)seq[i]]);
    }
    if (m > CAH_MAX_M) {
        mt.long_dp = 1;
        mt.m = m; mt.k = lm.k; mt.flags = lm.flags; mt.min_overlap = d.min_overlap; mt.wildcard_ref = wr;
        mt.indel_cost = d.indel_cost; mt.effective_length = lm.effective_length; mt.cmp_max_k = lm.cmp_max_k;
    }
    return CAH_OK;
}

static int build_matcher(const cah_adapter_desc& d, int index, CahMatcher& mt,
                         std::vector<CahKmerWord>& words, LongTables& lt) {
    memset(&mt, 0, sizeof(mt));
    if (d.kind < CAH_KIND_ALIGNER || d.kind > CAH_KIND_KMER_ONLY)
        return fail(CAH_EINVAL, "adapter %d: unknown kind %d", index, d.kind);
    const int m = d.length;
    const bool has_aligner = d.kind != CAH_KIND_KMER_ONLY;
    mt.kind = d.kind;
    if (has_aligner) {
        if (m < 0 || (m > 0 && !d.sequence))
            return fail(CAH_EINVAL, "adapter %d: bad sequence", index);
        if (!is_ascii(d.sequence, (size_t)m))
            return fail(CAH_EINVAL, "String must contain only ASCII characters");
        const bool wr = d.wildcard_ref != 0, wq = d.wildcard_query != 0;
        const double rate = d.max_error_rate;
        mt.m = m;
        mt.flags = d.flags & 15;
        mt.min_overlap = d.min_overlap;
        mt.wildcard_ref = wr;
        mt.indel_cost = d.indel_cost;
        const char* seq = d.sequence;
        if (m > CAH_MAX_M) {
            const int rc = build_long(d, index, mt, lt);
            if (rc != CAH_OK) return rc;
        } else if (d.kind == CAH_KIND_ALIGNER) {
            // _align.pyx:250-277
            int nn = 0;
            for (int i = 0; i < m; i++) {
                mt.n_counts[i] = nn;
                if (seq[i] == 'N' || seq[i] == 'n') nn++;
            }
            mt.n_counts[m] = nn;
            for (int i = m + 1; i <= CAH_MAX_M; i++) mt.n_counts[i] = nn;
            mt.effective_length = m;
            if (wr) {
                mt.effective_length = m - nn;
                if (mt.effective_length == 0)
                    return fail(CAH_EINVAL, "Cannot have only N wildcards in the sequence");
            }
            if (d.indel_cost < 1) return fail(CAH_EINVAL, "indel_cost must be at least 1");
            if (d.indel_cost > CAH_MAX_INDEL_COST)
                return fail(CAH_EUNSUPPORTED, "indel_cost above %d is not supported", CAH_MAX_INDEL_COST);
            if (!(rate == rate) || std::fabs(rate) > 1e6)
                return fail(CAH_EINVAL, "max_error_rate is not a usable number");
            mt.k = (int)(rate * m);                                     // :343
            for (int L = 0; L <= CAH_MAX_M; L++) mt.thr[L] = clamp_floor(L * rate);   // :513, :559
            for (int ch = 0; ch < CAH_TABLE_CHARS; ch++) {
                uint64_t bits = 0;
                for (int i = 0; i < m; i++)
                    if (aligner_chars_match((uint8_t)seq[i], (uint8_t)ch, wr, wq)) bits |= 1ull << i;
                mt.rowmask[ch] = bits;
            }
            // ---- bit-parallel cost scan in front of the cell DP (back_scan.h) ---------------------
            // 3' adapters (Where.BACK) with unit costs and an ordinary error budget: 0 <= k < m,
            // rate in [0, 1], min_overlap >= 1.  Everything else keeps the plain cell kernels.
            mt.scan_ok = 0;
            if (mt.flags == 14 && d.indel_cost == 1 && m >= 1 && m <= CAH_MAX_M && mt.k >= 0 && mt.k < m &&
                rate >= 0.0 && rate <= 1.0 && d.min_overlap >= 1 && !scan_disabled()) {
                mt.scan_ok = 1;
                // a last-row candidate spans the whole adapter: effective length (:504-510)
                mt.kacc = m >= d.min_overlap ? mt.thr[mt.effective_length] : -1;
                if (mt.kacc > mt.k) mt.kacc = mt.k;
                // row i of the last column: origin >= 0 for a 3' adapter, so length = i and the
                // effective length is i minus the N's of adapter[0:i] (:543-553)
                for (int i = 0; i <= CAH_MAX_M; i++) {
                    int eff = i;
                    if (wr) eff = i < m ? i - (mt.n_counts[std::min(i, m)] - mt.n_counts[0]) : mt.effective_length;
                    if (i > m) eff = 0;
                    mt.thr_last[i] = i <= m ? mt.thr[eff] : -1;
                }
                const int pad = 64 - m;
                const uint64_t low = pad == 0 ? 0ull : ((1ull << pad) - 1ull);
                for (int ch = 0; ch < CAH_TABLE_CHARS; ch++)
                    mt.scanmask[ch] = (pad == 64 ? 0ull : (mt.rowmask[ch] << pad)) | low;
            }
        } else {
            // PrefixComparer / SuffixComparer (_align.pyx:615-642, :698-706)
            int eff = m;
            if (wr) {
                int nN = 0, nn = 0;
                for (int i = 0; i < m; i++) { nN += seq[i] == 'N'; nn += seq[i] == 'n'; }
                eff -= nN - nn;                                         // quirk kept (:628)
                if (eff == 0) return fail(CAH_EINVAL, "Cannot have only N wildcards in the sequence");
            }
            if (!(rate >= 0.0 && rate <= 1.0))
                return fail(CAH_EINVAL, "max_error_rate must be between 0 and 1");
            if (d.min_overlap < 1) return fail(CAH_EINVAL, "min_overlap must be at least 1");
            mt.effective_length = eff;
            mt.cmp_max_k = (int)(rate * eff);                           // :633
            const bool suffix = d.kind == CAH_KIND_SUFFIX;
            for (int ch = 0; ch < CAH_TABLE_CHARS; ch++) {
                uint64_t bits = 0;
                for (int i = 0; i < m; i++) {
                    // the suffix variant compares reversed reference against reversed query
                    const uint8_t rc = (uint8_t)seq[suffix ? m - 1 - i : i];
                    if (comparer_chars_match(rc, (uint8_t)ch, wr, wq)) bits |= 1ull << i;
                }
                mt.rowmask[ch] = bits;
            }
        }
    }

    // ---- prefilter: pack k-mers into 64-bit shift-and words (_kmer_finder.pyx:121-164) -------
    mt.first_word = (int32_t)words.size();
    mt.has_filter = d.n_kmer_sets >= 0 ? 1 : 0;
    if (d.n_kmer_sets > 0 && !d.kmer_sets) return fail(CAH_EINVAL, "adapter %d: kmer_sets is NULL", index);
    const bool rwc = d.kmer_ref_wildcards != 0, qwc = d.kmer_query_wildcards != 0;
    // Word width: the reference packs k-mers greedily into 64-bit words (_kmer_finder.pyx:131-149).
    // kmers_present is an OR over all k-mers, so the packing is free; 32-bit words halve the
    // kernel's work per character, so they are used whenever every k-mer is <= 32 characters.
    size_t longest_kmer = 0;
    for (int s = 0; s < d.n_kmer_sets; s++)
        for (int t = 0; t < d.kmer_sets[s].n_kmers; t++)
            if (d.kmer_sets[s].kmers[t]) longest_kmer = std::max(longest_kmer, strlen(d.kmer_sets[s].kmers[t]));
    const size_t word_bits = longest_kmer <= 32 ? 32 : 64;
    mt.narrow_words = word_bits == 32 ? 1 : 0;
    for (int s = 0; s < d.n_kmer_sets; s++) {
        const cah_kmer_set& ks = d.kmer_sets[s];
        int idx = 0;
        while (idx < ks.n_kmers) {
            uint8_t word[64];
            memset(word, 0, sizeof(word));
            size_t off = 0;
            CahKmerWord kw;
            memset(&kw, 0, sizeof(kw));
            while (idx < ks.n_kmers) {
                const char* kmer = ks.kmers[idx];
                if (!kmer) return fail(CAH_ETYPE, "Kmer should be a string");
                const size_t len = strlen(kmer);
                if (!is_ascii(kmer, len)) return fail(CAH_EINVAL, "Only ASCII strings are supported");
                if (len > 64)
                    return fail(CAH_EINVAL, "%s of length %zu is longer than the maximum of 64.", kmer, len);
                if (len == 0)        // the reference shifts by -1 here (undefined behaviour, _kmer_finder.pyx:147)
                    return fail(CAH_EINVAL, "adapter %d: empty k-mer in a search set", index);
                if (off + len > word_bits) break;
                kw.init_mask |= 1ull << off;
                memcpy(word + off, kmer, len);
                kw.found_mask |= 1ull << (off + len - 1);
                off += len;
                idx++;
            }
            kw.start = ks.start;
            kw.stop = ks.stop;
            for (size_t p = 0; p < off; p++) {
                if (word[p] == 0) continue;
                for (int qc = 0; qc < CAH_TABLE_CHARS; qc++)
                    if (kmer_chars_match(word[p], (uint8_t)qc, rwc, qwc)) kw.mask[qc] |= 1ull << p;
            }
            words.push_back(kw);
        }
    }
    mt.n_words = (int32_t)words.size() - mt.first_word;
    // Whole-read words (window 0..end) go first: the prefilter scans words in groups of
    // CAH_FILTER_SLOTS and a read leaves at its first hit, so only words of the FIRST group are
    // guaranteed to have been scanned up to the hit position (kmers_present is an OR over all
    // words; their order is free).
    std::stable_partition(words.begin() + mt.first_word, words.end(),
                          [](const CahKmerWord& w) { return w.start == 0 && w.stop == 0; });
    int n_whole_read_words = 0;
    for (size_t i = (size_t)mt.first_word; i < words.size(); i++)
        n_whole_read_words += words[i].start == 0 && words[i].stop == 0;

    // ---- may the DP skip the columns before the first k-mer hit? -------------------------------
    // Only for 3' adapters (flags == QUERY_START|QUERY_STOP|REFERENCE_END) and only if this very
    // plan's prefilter provably has the pigeonhole property the argument needs: a search set over
    // the whole read (start 0, stop None) that contains every one of the k+1 consecutive chunks
    // of the adapter (reference kmer_heuristic.py:161-163 builds exactly that), matched with the
    // same wildcard relation as the aligner.  Hand-made k-mer sets that lack it keep skip_ok = 0.
    mt.skip_ok = 0;
    if (d.kind == CAH_KIND_ALIGNER && mt.flags == 14 && d.n_kmer_sets > 0 && m >= 1 && m <= CAH_MAX_M &&
        n_whole_read_words <= (mt.narrow_words ? CAH_FILTER_SLOTS_NARROW : CAH_FILTER_SLOTS) &&
        (d.kmer_ref_wildcards != 0) == (d.wildcard_ref != 0) &&
        (d.kmer_query_wildcards != 0) == (d.wildcard_query != 0)) {
        const int chunks = mt.k + 1;
        if (chunks <= m) {
            const int base = m / chunks, extra = m % chunks;
            for (int s = 0; s < d.n_kmer_sets && !mt.skip_ok; s++) {
                const cah_kmer_set& ks = d.kmer_sets[s];
                if (ks.start != 0 || ks.stop != 0) continue;
                bool all = true;
                int pos = 0;
                for (int cidx = 0; cidx < chunks && all; cidx++) {
                    const int len = base + (cidx < extra ? 1 : 0);
                    bool present = false;
                    for (int t = 0; t < ks.n_kmers && !present; t++)
                        present = (int)strlen(ks.kmers[t]) == len && strncmp(ks.kmers[t], d.sequence + pos, (size_t)len) == 0;
                    all = present;
                    pos += len;
                }
                if (all) mt.skip_ok = 1;
            }
        }
    }

    // ---- is the prefilter implied by an unedited, anchored occurrence of the adapter? (CahMatcher::filter_implied)
    mt.filter_implied = 0;
    if (d.kind == CAH_KIND_ALIGNER && (mt.flags == 8 || mt.flags == 2) && d.n_kmer_sets > 0 && m >= 1 && m <= CAH_MAX_M) {
        const bool prefix = mt.flags == 8;
        for (int s = 0; s < d.n_kmer_sets && !mt.filter_implied; s++) {
            const cah_kmer_set& ks = d.kmer_sets[s];
            for (int t = 0; t < ks.n_kmers && !mt.filter_implied; t++) {
                const char* kmer = ks.kmers[t];
                const int len = (int)strlen(kmer);
                for (int p = 0; p + len <= m && !mt.filter_implied; p++) {
                    // the window, with the adapter at read positions 0.. (prefix) or n-m.. (suffix, any n >= m)
                    bool inside;
                    if (ks.start == 0 && ks.stop == 0) inside = true;
                    else if (prefix) inside = ks.start >= 0 && ks.start <= p && (ks.stop == 0 || (ks.stop > 0 && p + len <= ks.stop));
                    else inside = ks.start < 0 && ks.stop == 0 && p >= m + ks.start;
                    if (!inside) continue;
                    // whatever the aligner accepts at adapter position p + i, the k-mer's character i accepts
                    bool implied = true;
                    for (int i = 0; i < len && implied; i++)
                        for (int qc = 0; qc < CAH_TABLE_CHARS && implied; qc++)
                            if (((mt.rowmask[qc] >> (p + i)) & 1ull) && !kmer_chars_match((uint8_t)kmer[i], (uint8_t)qc, rwc, qwc))
                                implied = false;
                    if (implied) mt.filter_implied = 1;
                }
            }
        }
    }
    return CAH_OK;
}


// The fused multi-adapter path (CahMultiHeader): possible when the plan holds 2..CAH_MULTI_MAX_ADAPTERS
// matchers that are all plain-ACGT 3' adapters of ONE shape (same length, error thresholds and
// min_overlap -- then only the match tables differ between lanes), every one scan-eligible, and every
// k-mer set is a whole-read set (0, None) or a tail set (-L, None) of ACGT k-mers of 1..32 characters
// matched without wildcards: exactly what `-a file:` with equally long adapters gives (BASELINE C4).
// Anything else keeps the one-adapter-at-a-time loop.  CAH_NO_MULTI=1 disables it (A/B, parity tests).
static void build_multi(const cah_adapter_desc* descs, int n, cah_plan* plan) {
    MultiPlan& mp = plan->multi;
    const char* off = getenv("CAH_NO_MULTI");
    if (off && *off && *off != '0') return;
    // below ~8 adapters one lean prefilter pass per adapter is faster than the fused pass (measured on C5:
    // 2 adapters, 2 x 9.9 ms against 75 ms per 125 M reads); CAH_MULTI_MIN overrides the threshold
    int min_adapters = 8;
    if (const char* e = getenv("CAH_MULTI_MIN")) { const int v = atoi(e); if (v >= 2) min_adapters = v; }
    if (n < min_adapters || n > CAH_MULTI_MAX_ADAPTERS) return;
    const CahMatcher& m0 = plan->matchers[0];
    auto base2 = [](char c) -> int { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; };
    struct Item { uint64_t code; uint32_t adapter; int q; int window; };
    std::vector<Item> items;
    bool skip_all = true;
    for (int a = 0; a < n; a++) {
        const cah_adapter_desc& d = descs[a];
        const CahMatcher& mt = plan->matchers[(size_t)a];
        if (d.kind != CAH_KIND_ALIGNER || !mt.scan_ok || d.wildcard_ref || d.wildcard_query) return;
        if (mt.m != m0.m || mt.k != m0.k || mt.kacc != m0.kacc || mt.min_overlap != m0.min_overlap) return;
        if (memcmp(mt.thr, m0.thr, sizeof(mt.thr)) != 0 || memcmp(mt.thr_last, m0.thr_last, sizeof(mt.thr_last)) != 0) return;
        for (int i = 0; i < mt.m; i++) if (base2(d.sequence[i]) < 0) return;
        if (d.n_kmer_sets <= 0 || !d.kmer_sets || d.kmer_ref_wildcards || d.kmer_query_wildcards) return;
        skip_all = skip_all && mt.skip_ok != 0;
        for (int si = 0; si < d.n_kmer_sets; si++) {
            const cah_kmer_set& ks = d.kmer_sets[si];
            int window;
            if (ks.start == 0 && ks.stop == 0) window = 0;
            else if (ks.start < 0 && ks.stop == 0 && ks.start >= -255) window = (int)-ks.start;
            else return;
            for (int t = 0; t < ks.n_kmers; t++) {
                const char* kmer = ks.kmers[t];
                if (!kmer) return;
                const size_t q = strlen(kmer);
                if (q < 1 || q > 32) return;
                uint64_t code = 0;
                for (size_t i = 0; i < q; i++) {
                    const int b = base2(kmer[i]);
                    if (b < 0) return;
                    code = (code << 2) | (uint64_t)b;
                }
                items.push_back({code, (uint32_t)a, (int)q, window});
            }
        }
    }
    if (items.empty()) return;
    CahMultiHeader& h = mp.hdr;
    h.n_adapters = n;
    h.skip_ok = skip_all ? 1 : 0;
    auto cls_of = [](int q) { return q < 8 ? q : 8; };
    auto key_of = [](const Item& it) -> uint32_t { return it.q < 8 ? (uint32_t)it.code : (uint32_t)(it.code & 0xFFFFu); };
    for (const Item& it : items) {
        const int c = cls_of(it.q);
        h.class_present[c] = 1;
        if (it.window == 0) h.class_everywhere[c] = 1;
        else h.class_lmax[c] = std::max(h.class_lmax[c], it.window);
    }
    uint32_t dir_total = 0, bm_total = 0;
    for (int c = 1; c <= 8; c++) {
        h.dir_off[c] = dir_total; h.bm_off[c] = bm_total;
        if (!h.class_present[c]) continue;
        const uint32_t keys = c < 8 ? (1u << (2 * c)) : 65536u;
        dir_total += keys;
        bm_total += (keys + 31) / 32;
    }
    h.bm_words = bm_total;
    mp.dir.assign(dir_total, CahMultiDir{0, 0});
    mp.bitmap.assign(bm_total, 0u);
    std::stable_sort(items.begin(), items.end(), [&](const Item& x, const Item& y) {
        const int cx = cls_of(x.q), cy = cls_of(y.q);
        if (cx != cy) return cx < cy;
        return key_of(x) < key_of(y);
    });
    mp.entries.reserve(items.size());
    for (const Item& it : items) {
        const int c = cls_of(it.q);
        const uint32_t key = key_of(it);
        CahMultiDir& d = mp.dir[h.dir_off[c] + key];
        if (d.count == 0) d.begin = (uint32_t)mp.entries.size();
        d.count++;
        mp.bitmap[h.bm_off[c] + (key >> 5)] |= 1u << (key & 31);
        CahMultiEntry en;
        memset(&en, 0, sizeof(en));
        en.code = it.code; en.adapter = it.adapter; en.q = (uint8_t)it.q; en.window = (uint8_t)it.window;
        mp.entries.push_back(en);
    }
    h.n_entries = (uint32_t)mp.entries.size();
    // per-adapter match tables indexed by (read character & 31): only A/C/G/T (either case) match anything
    mp.scan_tab.assign((size_t)n * CAH_MULTI_TAB_STRIDE, 0);
    mp.row_tab.assign((size_t)n * CAH_MULTI_TAB_STRIDE, 0);
    for (int a = 0; a < n; a++) {
        const CahMatcher& mt = plan->matchers[(size_t)a];
        for (int i = 0; i < CAH_MULTI_TAB; i++) {
            const bool letter = i == 1 || i == 3 || i == 7 || i == 20;                 // A C G T
            const int ch = 64 + i;
            mp.scan_tab[(size_t)a * CAH_MULTI_TAB_STRIDE + i] = letter ? mt.scanmask[ch] : mt.scanmask[0];
            mp.row_tab[(size_t)a * CAH_MULTI_TAB_STRIDE + i] = letter ? mt.rowmask[ch] : 0ull;
        }
    }
    h.ok = 1;
    // the streaming form (multi2.h): the same plan as tables of REF + WIDE k-mers of up to ten characters.
    // CAH_NO_MULTI2=1 keeps the older kernels (A/B, parity tests)
    const char* off2 = getenv("CAH_NO_MULTI2");
    if (!(off2 && *off2 && *off2 != '0')) {
        std::vector<std::string> ads;
        std::vector<std::vector<M2RefKmer>> ref((size_t)n);
        for (int a = 0; a < n; a++) {
            ads.push_back(std::string(descs[a].sequence, (size_t)descs[a].length));
            for (int si = 0; si < descs[a].n_kmer_sets; si++) {
                const cah_kmer_set& ks = descs[a].kmer_sets[si];
                for (int t = 0; t < ks.n_kmers; t++)
                    ref[(size_t)a].push_back({std::string(ks.kmers[t]), ks.start == 0 ? CAH_M2_WHOLE : (int)-ks.start});
            }
        }
        if (!m2_build(ads, m0.thr_last, m0.kacc, m0.k, m0.min_overlap, ref, mp.m2)) memset(&mp.m2.hdr, 0, sizeof(mp.m2.hdr));
    }
}

extern "C" {

int cah_abi_version(void) { return CAH_ABI_VERSION; }

#ifndef CAH_BUILD_ID
#define CAH_BUILD_ID "unknown"
#endif
// (the marker in front lets cutadapt_amd/build.py read the id from the file's bytes without loading the library)
static const char g_build_id_marker[] = "CAH_BUILD_ID=" CAH_BUILD_ID;
const char* cah_build_id(void) { return g_build_id_marker + 13; }

void cah_last_error(char* buf, size_t buflen) {
    if (!buf || !buflen) return;
    snprintf(buf, buflen, "%s", g_last_error.c_str());
}

int cah_device_count(int* count) {
    if (!count) return fail(CAH_EINVAL, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(CAH_EHIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return CAH_OK;
}

int cah_set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return CAH_OK;
}

int cah_device_info(int device, char* name, size_t name_len, char* arch, size_t arch_len,
                    int* compute_units, int64_t* hbm_bytes) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (name && name_len) snprintf(name, name_len, "%s", prop.name);
    if (arch && arch_len) snprintf(arch, arch_len, "%s", prop.gcnArchName);
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return CAH_OK;
}

static int plan_create_impl(const cah_adapter_desc* adapters, int32_t n_adapters, cah_plan** out);

int cah_plan_create(const cah_adapter_desc* adapters, int32_t n_adapters, cah_plan** out) {
    try {                                                    // nothing may be thrown across the C ABI
        return plan_create_impl(adapters, n_adapters, out);
    } catch (const std::bad_alloc&) {
        return fail(CAH_ENOMEM, "cah_plan_create: out of memory");
    } catch (...) {
        return fail(CAH_EINVAL, "cah_plan_create: internal error");
    }
}

static int plan_create_impl(const cah_adapter_desc* adapters, int32_t n_adapters, cah_plan** out) {
    if (!out) return fail(CAH_EINVAL, "out is NULL");
    *out = nullptr;
    if (n_adapters < 1 || !adapters) return fail(CAH_EINVAL, "need at least one adapter");
    std::unique_ptr<cah_plan> holder(new cah_plan());
    cah_plan* plan = holder.get();
    plan->matchers.resize((size_t)n_adapters);
    plan->lean.resize((size_t)n_adapters);
    plan->long_tabs.resize((size_t)n_adapters);
    for (int i = 0; i < n_adapters; i++) {
        int rc = build_matcher(adapters[i], i, plan->matchers[(size_t)i], plan->words, plan->long_tabs[(size_t)i]);
        if (rc != CAH_OK) return rc;
        if (plan->matchers[(size_t)i].long_dp) plan->max_long_m = std::max(plan->max_long_m, plan->matchers[(size_t)i].m);
        build_lean_filter(adapters[i], plan->lean[(size_t)i]);
#ifdef CAH_NO_LEAN
        plan->lean[(size_t)i].ok = 0;                 // A/B builds
#endif
    }
    build_multi(adapters, n_adapters, plan);
    *out = holder.release();
    return CAH_OK;
}

void cah_plan_destroy(cah_plan* plan) {
    if (!plan) return;
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < CAH_MAX_DEVICES; d++) {
        PlanDeviceCopy& dc = plan->dev[d];
        if (!dc.ready) continue;
        (void)hipSetDevice(d);
        if (dc.d_matchers) (void)hipFree(dc.d_matchers);
        if (dc.d_words) (void)hipFree(dc.d_words);
        if (dc.d_lean) (void)hipFree(dc.d_lean);
        if (dc.d_mhdr) (void)hipFree(dc.d_mhdr);
        if (dc.d_mdir) (void)hipFree(dc.d_mdir);
        if (dc.d_mentries) (void)hipFree(dc.d_mentries);
        if (dc.d_mbitmap) (void)hipFree(dc.d_mbitmap);
        if (dc.d_mscan) (void)hipFree(dc.d_mscan);
        if (dc.d_mrow) (void)hipFree(dc.d_mrow);
        if (dc.d_m2hdr) (void)hipFree(dc.d_m2hdr);
        if (dc.d_m2dir) (void)hipFree(dc.d_m2dir);
        if (dc.d_m2entries) (void)hipFree(dc.d_m2entries);
        if (dc.d_m2bitmap) (void)hipFree(dc.d_m2bitmap);
        if (dc.d_m2prefix) (void)hipFree(dc.d_m2prefix);
        if (dc.d_m2refbegin) (void)hipFree(dc.d_m2refbegin);
        if (dc.d_m2reflist) (void)hipFree(dc.d_m2reflist);
        if (dc.d_tiny_image[0]) (void)hipFree(dc.d_tiny_image[0]);
        if (dc.d_tiny_image[1]) (void)hipFree(dc.d_tiny_image[1]);
        for (LongDeviceCopy& ld : dc.d_long) {
            if (ld.d_lm) (void)hipFree(ld.d_lm);
            if (ld.d_ref) (void)hipFree(ld.d_ref);
            if (ld.d_ncnt) (void)hipFree(ld.d_ncnt);
        }
    }
    if (cur >= 0) (void)hipSetDevice(cur);
    delete plan;
}

int cah_plan_n_adapters(const cah_plan* plan) { return plan ? (int)plan->matchers.size() : 0; }

static int check_adapter(const cah_plan* plan, int32_t adapter) {
    if (!plan) return fail(CAH_EINVAL, "plan is NULL");
    if (adapter < 0 || (size_t)adapter >= plan->matchers.size())
        return fail(CAH_EINVAL, "adapter index %d out of range", adapter);
    return CAH_OK;
}

int cah_plan_effective_length(const cah_plan* plan, int32_t adapter, int32_t* out) {
    int rc = check_adapter(plan, adapter);
    if (rc) return rc;
    *out = plan->matchers[(size_t)adapter].effective_length;
    return CAH_OK;
}

int cah_plan_prefilter_kind(const cah_plan* plan, int32_t adapter, int32_t* out) {
    int rc = check_adapter(plan, adapter);
    if (rc) return rc;
    if (!out) return fail(CAH_EINVAL, "out is NULL");
    const CahMatcher& mt = plan->matchers[(size_t)adapter];
    *out = !mt.has_filter ? CAH_PREFILTER_NONE : (plan->lean[(size_t)adapter].ok ? CAH_PREFILTER_LEAN : CAH_PREFILTER_GENERAL);
    return CAH_OK;
}

int cah_plan_multi_kind(const cah_plan* plan, int32_t read_len, int32_t* out) {
    if (!plan || !out) return fail(CAH_EINVAL, "plan or out is NULL");
    const MultiPlan& mp = plan->multi;
    *out = !mp.hdr.ok ? CAH_MULTI_SEQUENTIAL
                      : ((mp.m2.hdr.ok && multi2_read_len_ok(mp.m2.hdr, read_len) && !env_flag_early("CAH_NO_MULTI2")) ? CAH_MULTI_STREAM : CAH_MULTI_FUSED);
    return CAH_OK;
}

int cah_plan_debug_matcher(const cah_plan* plan, int32_t adapter, void* buf, size_t buflen, size_t* need) {
    int rc = check_adapter(plan, adapter);
    if (rc) return rc;
    if (need) *need = sizeof(CahMatcher);
    if (buf) {
        if (buflen < sizeof(CahMatcher)) return fail(CAH_EINVAL, "buffer too small: need %zu bytes", sizeof(CahMatcher));
        memcpy(buf, &plan->matchers[(size_t)adapter], sizeof(CahMatcher));
    }
    return CAH_OK;
}

int cah_plan_debug_lean(const cah_plan* plan, int32_t adapter, void* buf, size_t buflen, size_t* need) {
    int rc = check_adapter(plan, adapter);
    if (rc) return rc;
    if (need) *need = sizeof(CahLeanFilter);
    if (buf) {
        if (buflen < sizeof(CahLeanFilter)) return fail(CAH_EINVAL, "buffer too small: need %zu bytes", sizeof(CahLeanFilter));
        memcpy(buf, &plan->lean[(size_t)adapter], sizeof(CahLeanFilter));
    }
    return CAH_OK;
}

int cah_plan_n_kmer_entries(const cah_plan* plan, int32_t adapter, int32_t* out) {
    int rc = check_adapter(plan, adapter);
    if (rc) return rc;
    *out = plan->matchers[(size_t)adapter].n_words;
    return CAH_OK;
}

// ---------------------------------------------------------------------------------------------
// profiling (HIP events on the launch stream)
// ---------------------------------------------------------------------------------------------
namespace {
struct ProfRec { hipEvent_t a, b; int family; int64_t units; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::mutex g_prof_mu;

struct ProfScope {
    hipStream_t s; int family; int64_t units; hipEvent_t a = nullptr, b = nullptr; bool on;
    ProfScope(hipStream_t s_, int family_, int64_t units_) : s(s_), family(family_), units(units_), on(g_prof_on) {
        if (on) {
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            (void)hipEventRecord(a, s);
        }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(b, s);
            std::lock_guard<std::mutex> lk(g_prof_mu);
            g_prof.push_back({a, b, family, units});
        }
    }
};
}  // namespace

int cah_profile_enable(int enable) { g_prof_on = enable != 0; return CAH_OK; }

int cah_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    return CAH_OK;
}

int cah_profile_read(double ms[CAH_PROF_N], int64_t launches[CAH_PROF_N], int64_t units[CAH_PROF_N]) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < CAH_PROF_N; i++) { ms[i] = 0; launches[i] = 0; units[i] = 0; }
    for (auto& r : g_prof) {
        HIP_TRY(hipEventSynchronize(r.b));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
        ms[r.family] += t;
        if (r.units < 0) continue;                   // a further round of the same pass over the same units
        launches[r.family] += 1;
        units[r.family] += r.units;
    }
    return CAH_OK;
}

// ---------------------------------------------------------------------------------------------
// batch entry points
// ---------------------------------------------------------------------------------------------
// workspace layout: counters, each on its own line (they are hammered by different kernels), then the
// survivor queue with its per-entry keys and the cell-DP work list the cost scan writes:
//   [0,8) filter tile counter | [128,136) uniform-batch flag | [256,264) queue count |
//   [512,520) DP work counter | [640,648) scan tile counter | [768,776) DP list front count |
//   [896,904) DP list back count | [576,584) straggler count | [704,712) tile counter of the stragglers' scan |
//   [1024, +4n) queue | [.., +n) queue keys | [.., +4n) DP work list | [.., +8n) its column windows |
//   [.., +4c) the cost scan's straggler list | [.., +c) its keys      (c = n / 8 + 1024)
static const size_t WS_HEADER = 1024;
static const size_t WS_UFLAG = 128 / sizeof(unsigned long long);    // 0 after the check: all reads of the batch have one length
static const size_t WS_QCOUNT = 256 / sizeof(unsigned long long), WS_DPWORK = 512 / sizeof(unsigned long long);
static const size_t WS_SCANWORK = 640 / sizeof(unsigned long long), WS_DPFRONT = 768 / sizeof(unsigned long long);
static const size_t WS_DPBACK = 896 / sizeof(unsigned long long);
// the streaming multi-adapter form's tile counter, which lives through the rounds of a batch (the rounds clear the header in front of it)
static const size_t WS_M2_TILE = 960 / sizeof(unsigned long long);
static const size_t WS_M2_ERR = 968 / sizeof(unsigned long long);     // multi2's error bits (kernels.h: Multi2Args::err); survives the rounds' re-zeroing
static const size_t WS_RETRYCOUNT = 576 / sizeof(unsigned long long), WS_RETRYWORK = 704 / sizeof(unsigned long long);
static int64_t ws_retry_cap(int64_t n_reads) { return n_reads / 8 + 1024; }

static size_t ws_queue_bytes(int64_t n_reads) { return (sizeof(int32_t) * (size_t)n_reads + 255) & ~(size_t)255; }
static size_t ws_keys_bytes(int64_t n_reads) { return ((size_t)n_reads + 255) & ~(size_t)255; }

size_t cah_workspace_bytes(int64_t n_reads) {
    if (n_reads < 0) n_reads = 0;
    return WS_HEADER + 2 * ws_queue_bytes(n_reads) + ws_keys_bytes(n_reads) + 2 * ws_queue_bytes(n_reads) +
           ws_queue_bytes(ws_retry_cap(n_reads)) + ws_keys_bytes(ws_retry_cap(n_reads)) + 256;
}

// ---- extra scratch of the fused multi-adapter path: [best key 8n] [pairs 8*cap] [DP list 4*cap] [windows 8*cap]
// cap = pairs one chunk of reads can produce in the worst case (every adapter on every read), bounded by
// CAH_MULTI_PAIR_CAP (default 1 G pairs = 20 GB of scratch on a 288 GB device: every chunk costs the cell DP a launch
// whose duration is that of its slowest wave, ~0.45 ms, however few pairs it has -- at 256 M pairs a 100 M-read batch
// of 96 adapters paid that 36 times, 16 of its 114 ms); larger batches are processed in chunks of
// cap / n_adapters reads.
// The streaming form's pool (multi2.h: pages of CAH_M2_PAGE pairs, one class each).  A closed page may lack up to 63
// pairs; every wave of a block holds one open page per class; a block has at most three tiles in flight (its waves may
// straddle two, the third is drawn ahead).
static int64_t m2_pages_per_tile(int64_t A) { return (1024 * A + (CAH_M2_PAGE - 64)) / (CAH_M2_PAGE - 63); }   // (multi2.hip: M2_TILE)
static int64_t m2_block_reserve(int64_t A) { return 3 * m2_pages_per_tile(A) + 16 * CAH_M2_PAIR_CLASSES; }
static int64_t m2_pages_worst(int64_t A, int64_t n_reads) {
    const int64_t tiles = (n_reads + 1023) / 1024;
    return tiles * m2_pages_per_tile(A) + std::min<int64_t>(std::max<int64_t>(tiles, 1), 512) * 16 * CAH_M2_PAIR_CLASSES + 2;
}
// pairs the scratch has room for: what the batch can produce in the worst case (every adapter on every read), bounded
// by CAH_MULTI_PAIR_CAP (default: default_pair_limit() below -- 2 G pairs = 40 GB on an empty 288 GB device).  The fused form handles larger
// batches in chunks of cap / n_adapters reads; the streaming form in ROUNDS that end when the pool runs low (a typical
// batch of 100 M reads x 96 adapters has 0.45 G pairs and takes one).
// The default bound follows the device: a quarter of the HBM that is FREE when a device's first multi-adapter workspace is
// sized (20 bytes of scratch per pair), at most 2 G pairs, at least 64 M; asked once per device and kept, because the
// sizing call (cah_plan_workspace_bytes) and every later match call must agree on the layout.  Without a usable device
// (the CPU-side tests ask for sizes too) it is 2 G pairs.
static int64_t default_pair_limit() {
    static std::mutex mu;
    static int64_t per_device[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 2048ll << 20; }
    std::lock_guard<std::mutex> g(mu);
    if (!per_device[dev]) {
        size_t free_b = 0, total_b = 0;
        int64_t limit = 2048ll << 20;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            limit = std::min<int64_t>(limit, std::max<int64_t>(64ll << 20, (int64_t)(free_b / 4 / 20)));
        else (void)hipGetLastError();
        per_device[dev] = limit;
    }
    return per_device[dev];
}
static int64_t multi_pair_cap(const cah_plan* plan, int64_t n_reads) {
    int64_t limit = 0;
    if (const char* e = getenv("CAH_MULTI_PAIR_CAP")) { const long long v = atoll(e); if (v > 0) limit = v; }
    if (limit <= 0) limit = default_pair_limit();
    // (the cell DP's work list holds pair indices as int32: the pool never has 2^31 pairs or more, whatever the knob says)
    limit = std::min<int64_t>(limit, ((int64_t)1 << 31) - 4 * CAH_M2_PAGE);
    const int64_t A = (int64_t)plan->matchers.size();
    if (limit < A) limit = A;
    const int64_t fused = std::min(n_reads * A, limit);
    if (!plan->multi.m2.hdr.ok) return fused;               // (no streaming form for this plan: no page pool, no floor)
    // (one block of the streaming prefilter must always be able to run)
    const int64_t floor_pages = 2 * m2_block_reserve(A) + 2 * m2_pages_per_tile(A);
    const int64_t pages = std::min(m2_pages_worst(A, n_reads), std::max(limit / CAH_M2_PAGE, floor_pages));
    // (a page header word per page lives behind the pages, inside the pair area)
    return std::max(fused, pages * CAH_M2_PAGE + pages / 2 + 2);
}
static size_t ws_key_bytes(int64_t n_reads) { return (sizeof(unsigned long long) * (size_t)n_reads + 255) & ~(size_t)255; }

// column scratch of k_dp_long: 3 int32 per row and lane, for as many lanes as the kernel is launched with
// (sized for a 256-CU device; fewer lanes are launched if the caller's workspace is smaller)
static size_t long_scratch_bytes(const cah_plan* plan, int64_t lanes) {
    return (size_t)lanes * 3 * sizeof(int32_t) * ((size_t)plan->max_long_m + 1);
}

size_t cah_plan_workspace_bytes(const cah_plan* plan, int64_t n_reads) {
    if (n_reads < 0) n_reads = 0;
    size_t need = cah_workspace_bytes(n_reads);
    if (plan && plan->multi.hdr.ok)
        need += ws_key_bytes(n_reads) + (size_t)multi_pair_cap(plan, n_reads) * 20 + 2 * ws_keys_bytes(n_reads) + 256;
    if (plan && plan->max_long_m > 0) need += long_scratch_bytes(plan, long_scratch_lanes(n_reads, 256)) + 256;
    return need;
}

namespace {
struct Workspace {
    unsigned long long* counters;
    int32_t* queue;
    uint8_t* keys;
    int32_t* dp_queue;
    int32_t* dp_win;
    int32_t* retry_queue;
    uint8_t* retry_keys;
    int64_t retry_cap;
    char* extra = nullptr;            // scratch behind the base layout (cah_plan_workspace_bytes), if any
    size_t extra_bytes = 0;
    Workspace(void* base, int64_t n_reads, size_t total_bytes = 0) {
        const size_t base_bytes = cah_workspace_bytes(n_reads);
        if (total_bytes > base_bytes) { extra = (char*)base + base_bytes; extra_bytes = total_bytes - base_bytes; }
        char* p = (char*)base;
        counters = (unsigned long long*)p;              p += WS_HEADER;
        queue = (int32_t*)p;                            p += ws_queue_bytes(n_reads);
        keys = (uint8_t*)p;                             p += ws_keys_bytes(n_reads);
        dp_queue = (int32_t*)p;                         p += ws_queue_bytes(n_reads);
        dp_win = (int32_t*)p;                           p += 2 * ws_queue_bytes(n_reads);
        retry_cap = ws_retry_cap(n_reads);
        retry_queue = (int32_t*)p;                      p += ws_queue_bytes(retry_cap);
        retry_keys = (uint8_t*)p;
    }
};
}  // namespace

static int check_batch(const cah_plan* plan, const void* d_seqs, const void* d_offsets, int64_t n_reads) {
    if (!plan) return fail(CAH_EINVAL, "plan is NULL");
    if (n_reads < 0 || n_reads > 2147483647LL) return fail(CAH_EINVAL, "n_reads out of range (0..2^31-1)");
    if (n_reads > 0 && (!d_offsets)) return fail(CAH_EINVAL, "offsets is NULL");
    (void)d_seqs;
    return CAH_OK;
}

// Tiny batches (the per-read calls of the Python mirror classes send batches of one) are launch-bound: what a caller
// has already done for the callee travels as explicit arguments (no per-thread state between entry points):
//   header_fresh    the whole counter header of the workspace was zeroed with ONE memset: run_filter / run_aligner skip theirs;
//   outputs_ready   the host conveniences cleared their contiguous output block (status / out6 / best_adapter) with one memset;
//   precleaned_ws   the one-read path cleared the counters of THIS workspace for the next call as soon as the kernels of the
//                   current one were queued (the memset runs while the host is busy elsewhere); kept in the thread's
//                   HostScratch, which owns that workspace.
#define CAH_TINY_BATCH 64
struct CallHints {
    bool outputs_ready = false;
    const void* precleaned_ws = nullptr;
};

// Aligner of an anchored adapter (Where.PREFIX = QUERY_STOP, Where.SUFFIX = QUERY_START) whose error threshold at
// full length is 0: see k_anchored_exact.  CAH_NO_ANCHORED_EXACT=1 keeps the cell DP (A/B, parity tests).
static bool anchored_exact_ok(const CahMatcher& mt) {
    static const bool off = [] { const char* e = getenv("CAH_NO_ANCHORED_EXACT"); return e && *e && *e != '0'; }();
    return !off && !mt.long_dp && (mt.flags == 8 || mt.flags == 2) && mt.m >= 1 && mt.m <= CAH_MAX_M &&
           mt.effective_length >= 0 && mt.effective_length <= CAH_MAX_M && mt.thr[mt.effective_length] == 0 &&
           mt.min_overlap <= mt.m;
}

// does cah_match_batch run this matcher's prefilter?  (not when it cannot reject anything the aligner would accept)
static bool runs_filter(const CahMatcher& mt) {
    return mt.has_filter && !(mt.filter_implied && mt.kind == CAH_KIND_ALIGNER && anchored_exact_ok(mt));
}

// CAH_SCAN_WORD64=1: the cost scan always uses the 64-bit form of the column (A/B measurements, parity tests)
static int scan_word_kind(int m) {
    static const bool force64 = [] { const char* e = getenv("CAH_SCAN_WORD64"); return e && *e && *e != '0'; }();
    return force64 ? 0 : bs_kind_of(m);
}

// CAH_SCAN_RETRY=<lanes> (default 12; 0 = off): see ScanArgs::retry_threshold
static int scan_retry_threshold() {
    const char* e = getenv("CAH_SCAN_RETRY");                 // (read per call: tests switch it inside one process)
    if (!e || !*e) return 12;
    const int x = atoi(e);
    return x < 0 ? 0 : (x > 63 ? 63 : x);
}
// CAH_SCAN_RETRY_CAP=<entries>: test knob, a smaller straggler list than the workspace has room for (overflow path)
static int64_t scan_retry_cap(int64_t cap) {
    const char* e = getenv("CAH_SCAN_RETRY_CAP");
    if (!e || !*e) return cap;
    const long long x = atoll(e);
    return x < 1 ? 1 : (x < cap ? x : cap);
}

// Aligner / comparer over a work list (d_queue == NULL: all reads).  3' adapters with unit costs go
// through the cost scan first (k_back_scan finishes most reads, the rest reach k_dp_packed with an exact
// column window); everything else runs the cell kernel directly.
// equally long reads the host vouches for (cah_match_batch_uniform): read r = d_seqs[first + r * len ..); len == 0: the
// packed layout (offsets / lens)
static bool env_flag(const char* name) { const char* e = getenv(name); return e && *e && *e != '0'; }

// suffix: (first, len) describe a PARENT batch of equally long reads and d_offsets / d_lens views that end where
// its reads end (cah_match_batch_suffix_views): only the streaming prefilter makes use of that, every other kernel
// sees plain views.
// inner (with suffix): the views may end before the parent's reads do (cah_match_batch_views)
struct UniformLayout { int64_t first = 0; int32_t len = 0; bool suffix = false; bool inner = false; bool general = false; };
// cah_linked_match_batch_uniform, fused form: the streaming prefilter of the back adapter decides the views itself
// (kernels.h: FilterArgs::front) and writes the front stage's outputs and the views
struct FrontFuse {
    const CahMatcher* d_matcher = nullptr;
    int32_t* out6 = nullptr;
    uint8_t* status = nullptr;
    int32_t* best = nullptr;
    int64_t* starts = nullptr;
    int32_t* lens = nullptr;
};

static int run_aligner(const cah_plan* plan, const PlanDeviceCopy* pd, int32_t adapter, const uint8_t* d_seqs,
                       const int64_t* d_offsets, const int32_t* d_lens, int64_t n_reads,
                       const int32_t* d_queue, const unsigned long long* d_queue_count,
                       const uint8_t* d_queue_keys, const Workspace& ws, int32_t* d_out6,
                       uint8_t* d_status, int32_t* d_best, int merge_best, hipStream_t s,
                       const UniformLayout ul = UniformLayout(), const bool header_fresh = false) {
    const CahMatcher& mt = plan->matchers[(size_t)adapter];
    DpArgs a;
    a.uniform_first = ul.first; a.uniform_len = ul.len;
    a.matcher = pd->d_matchers + adapter;
    a.seqs = d_seqs; a.offsets = d_offsets; a.lens = d_lens; a.n_reads = n_reads;
    a.max_read_len = CAH_MAX_READ_LEN;
    a.queue = d_queue; a.queue_count = d_queue_count; a.queue_keys = d_queue_keys;
    a.work_counter = ws.counters + WS_DPWORK;
    a.out6 = d_out6; a.status = d_status; a.best_adapter = d_best;
    a.adapter_index = adapter; a.merge_best = merge_best;
    a.win = nullptr; a.queue_count_back = nullptr; a.queue_cap = 0;
    a.pairs = nullptr; a.tab = nullptr; a.n_adapters = 0; a.best_key = nullptr;
    // DP work counter, scan tile counter, DP list counts: one memset over their lines
    if (!header_fresh)
        HIP_TRY(hipMemsetAsync(ws.counters + WS_DPWORK, 0, WS_HEADER - WS_DPWORK * sizeof(unsigned long long), s));
    if (mt.long_dp) {
        // adapter longer than 64 characters: column in HBM scratch (long.hip)
        int64_t lanes = long_scratch_lanes(n_reads, pd->n_cus);
        const size_t per_lane = long_scratch_bytes(plan, 1);
        if (ws.extra_bytes / per_lane < (size_t)lanes) lanes = (int64_t)(ws.extra_bytes / per_lane) / 256 * 256;
        if (lanes < 256)
            return fail(CAH_EINVAL, "workspace too small for an adapter of %d characters: need cah_plan_workspace_bytes() = %zu bytes",
                        mt.m, cah_plan_workspace_bytes(plan, n_reads));
        const LongDeviceCopy& ld = pd->d_long[(size_t)adapter];
        LongArgs la;
        la.uniform_first = ul.first; la.uniform_len = ul.len;
        la.lm = ld.d_lm; la.ref = ld.d_ref; la.ncnt = ld.d_ncnt;
        la.seqs = d_seqs; la.offsets = d_offsets; la.lens = d_lens; la.n_reads = n_reads; la.max_read_len = CAH_MAX_READ_LEN;
        la.queue = d_queue; la.queue_count = d_queue_count; la.work_counter = ws.counters + WS_DPWORK;
        la.scratch = (int32_t*)ws.extra;
        la.out6 = d_out6; la.status = d_status; la.best_adapter = d_best; la.adapter_index = adapter; la.merge_best = merge_best;
        la.dbg_cost = nullptr; la.dbg_score = nullptr;
        ProfScope ps(s, mt.kind == CAH_KIND_ALIGNER ? CAH_PROF_DP : CAH_PROF_COMPARER, n_reads);
        HIP_TRY(launch_dp_long(la, lanes, s));
        return CAH_OK;
    }
    if (mt.kind == CAH_KIND_ALIGNER && anchored_exact_ok(mt)) {
        // an anchored adapter that tolerates no error: a character-by-character comparison (k_anchored_exact)
        ProfScope ps(s, CAH_PROF_DP, n_reads);
        HIP_TRY(launch_anchored_exact(a, n_reads, pd->n_cus, s));
        return CAH_OK;
    }
    if (mt.kind == CAH_KIND_ALIGNER) {
        if (mt.scan_ok) {
            ScanArgs sa;
            sa.uniform_first = ul.first; sa.uniform_len = ul.len;
            sa.matcher = a.matcher;
            sa.seqs = d_seqs; sa.offsets = d_offsets; sa.lens = d_lens; sa.n_reads = n_reads;
            sa.max_read_len = CAH_MAX_READ_LEN;
            sa.queue = d_queue; sa.queue_count = d_queue_count; sa.queue_keys = d_queue_keys;
            sa.work_counter = ws.counters + WS_SCANWORK;
            sa.out6 = d_out6; sa.status = d_status; sa.best_adapter = d_best;
            sa.adapter_index = adapter; sa.merge_best = merge_best;
            sa.pairs = nullptr; sa.tab = nullptr; sa.n_adapters = 0; sa.multi_skip_ok = 0; sa.best_key = nullptr;
            sa.dp_queue = ws.dp_queue; sa.dp_win = ws.dp_win;
            sa.dp_count_front = ws.counters + WS_DPFRONT; sa.dp_count_back = ws.counters + WS_DPBACK;
            sa.dp_cap = n_reads;
            // stragglers of a wave are set aside and scanned again, packed, by a second launch (kernels.h); tiny
            // batches are launch-bound and keep the single launch
            sa.retry_threshold = n_reads > 4096 ? scan_retry_threshold() : 0;
            sa.retry_queue = ws.retry_queue; sa.retry_keys = ws.retry_keys; sa.retry_cap = scan_retry_cap(ws.retry_cap);
            sa.retry_count = ws.counters + WS_RETRYCOUNT; sa.queue_limit = 0;
            sa.early_stop = d_queue != nullptr && d_queue_keys != nullptr;
            sa.tile = 0;
            sa.kind = scan_word_kind(mt.m);
            {
                ProfScope ps(s, CAH_PROF_SCAN, n_reads);
                HIP_TRY(launch_back_scan(sa, n_reads, pd->n_cus, s));
                if (sa.retry_threshold > 0) {
                    ScanArgs sb = sa;
                    sb.queue = ws.retry_queue; sb.queue_keys = ws.retry_keys; sb.queue_count = ws.counters + WS_RETRYCOUNT;
                    sb.queue_limit = sa.retry_cap; sb.work_counter = ws.counters + WS_RETRYWORK;
                    sb.retry_threshold = 0;
                    sb.tile = 256;                   // few reads, long scans: one wave-load per wave keeps the chip busy
                    HIP_TRY(launch_back_scan(sb, sa.retry_cap < n_reads ? sa.retry_cap : n_reads, pd->n_cus, s));
                }
            }
            a.queue = ws.dp_queue; a.queue_keys = nullptr; a.win = ws.dp_win;
            a.queue_count = ws.counters + WS_DPFRONT; a.queue_count_back = ws.counters + WS_DPBACK;
            a.queue_cap = n_reads;
        }
        ProfScope ps(s, CAH_PROF_DP, n_reads);
        HIP_TRY(launch_dp(a, mt.m, mt.indel_cost == 1, mt.flags == 14, n_reads, pd->n_cus, s));
    } else {
        ProfScope ps(s, CAH_PROF_COMPARER, n_reads);
        HIP_TRY(launch_comparer(a, n_reads, pd->n_cus, s));
    }
    return CAH_OK;
}

int cah_locate_batch(const cah_plan* plan, int32_t adapter, const uint8_t* d_seqs,
                     const int64_t* d_offsets, const int32_t* d_lens, int64_t n_reads,
                     int32_t* d_out6, uint8_t* d_status, void* d_workspace, size_t workspace_bytes,
                     void* stream) {
    int rc = check_batch(plan, d_seqs, d_offsets, n_reads);
    if (rc) return rc;
    rc = check_adapter(plan, adapter);
    if (rc) return rc;
    if (plan->matchers[(size_t)adapter].kind == CAH_KIND_KMER_ONLY)
        return fail(CAH_EINVAL, "adapter %d has no aligner", adapter);
    if (n_reads == 0) return CAH_OK;
    if (!d_out6 || !d_status) return fail(CAH_EINVAL, "output pointers are NULL");
    if (!d_workspace || workspace_bytes < cah_workspace_bytes(n_reads))
        return fail(CAH_EINVAL, "workspace too small: need %zu bytes", cah_workspace_bytes(n_reads));
    const Workspace ws(d_workspace, n_reads, workspace_bytes);
    const PlanDeviceCopy* pd = nullptr;
    rc = plan_on_device(plan, &pd);
    if (rc) return rc;
    return run_aligner(plan, pd, adapter, d_seqs, d_offsets, d_lens, n_reads, nullptr, nullptr, nullptr,
                       ws, d_out6, d_status, nullptr, 0, (hipStream_t)stream);      // (its one memset is the header's)
}

// can the streaming prefilter take views that are suffixes of a uniform batch's reads (k_filter_stream2, SV form)?
static bool stream2_suffix_ok(const cah_plan* plan, int32_t adapter, int32_t len, int64_t n_reads) {
    const CahLeanFilter& lf = plan->lean[(size_t)adapter];
    return lf.ok && lf.tw_ok && stream2_class_ok(lf.n_lead, lf.n_tw) && len >= 1 && len <= stream2_max_len() &&
           n_reads * (int64_t)len >= 16 && !env_flag("CAH_NO_STREAM") && !env_flag("CAH_NO_STREAM2");
}

static int run_filter(const cah_plan* plan, const PlanDeviceCopy* pd, int32_t adapter, const uint8_t* d_seqs,
                      const int64_t* d_offsets, const int32_t* d_lens, int64_t n_reads, int mode,
                      uint8_t* d_present, uint8_t* d_status, int32_t* d_queue,
                      unsigned long long* d_queue_count, uint8_t* d_queue_keys,
                      unsigned long long* d_work_counter, const unsigned long long* d_batch_flag, hipStream_t s,
                      int32_t* d_clear_out6 = nullptr, int32_t* d_clear_best = nullptr,
                      const UniformLayout ul = UniformLayout(), const FrontFuse* fuse = nullptr,
                      const bool header_fresh = false) {
    const CahMatcher& mt = plan->matchers[(size_t)adapter];
    if (mt.n_words > 1024)
        return fail(CAH_EUNSUPPORTED, "adapter %d: %d packed k-mer words exceed the 1024-word limit of the prefilter kernel",
                    adapter, mt.n_words);
    FilterArgs f;
    f.words = pd->d_words + mt.first_word;
    f.n_words = mt.n_words;
    f.seqs = d_seqs; f.offsets = d_offsets; f.lens = d_lens; f.n_reads = n_reads;
    f.max_read_len = CAH_MAX_READ_LEN;
    f.work_counter = d_work_counter;
    f.present = d_present; f.status = d_status; f.queue = d_queue; f.queue_count = d_queue_count;
    f.queue_keys = d_queue_keys;
    f.batch_flag = nullptr;
    f.lean = nullptr;
    f.stream_n_lo = 0; f.stream_n_hi = -1;
    f.uniform_first = ul.first; f.uniform_len = ul.len;
    if (ul.suffix) {
        // views into a uniform parent: k_filter_stream2's SV form if the plan and the length are its, else plain views
        const bool s2 = stream2_suffix_ok(plan, adapter, ul.len, n_reads);
        f.suffix_views = s2 ? (ul.general ? 3 : (ul.inner ? 2 : 1)) : 0;     // (3: views anywhere, frames of ul.len characters)
        if (!s2) { f.uniform_first = 0; f.uniform_len = 0; }
        if (fuse) {
            if (!s2) return fail(CAH_EINVAL, "internal: fused linked path on a plan the streaming prefilter does not take");
            f.front = fuse->d_matcher; f.front_out6 = fuse->out6; f.front_status = fuse->status; f.front_best = fuse->best;
            f.view_starts = fuse->starts; f.view_lens = fuse->lens;
        }
    }
    f.clear_out6 = mode == 1 ? d_clear_out6 : nullptr;
    f.clear_best = f.clear_out6 ? d_clear_best : nullptr;
    if (!header_fresh) {
        HIP_TRY(hipMemsetAsync(d_work_counter, 0, sizeof(unsigned long long), s));
        if (d_queue_count) HIP_TRY(hipMemsetAsync(d_queue_count, 0, sizeof(unsigned long long), s));
    }
    ProfScope ps(s, CAH_PROF_FILTER, n_reads);
    if (plan->lean[(size_t)adapter].ok) {
        // 3' adapter plans: k_filter_lean.  For a packed batch the device-side batch check (*d_batch_flag)
        // picks its equal-length or its ragged variant -- both are launched, one leaves at once, no host
        // sync; views (explicit lengths) and calls without a check take the ragged variant.
        f.batch_flag = (d_lens || f.uniform_len > 0) ? nullptr : d_batch_flag;
        f.lean = pd->d_lean + adapter;
        const CahLeanFilter& lf = plan->lean[(size_t)adapter];
        HIP_TRY(launch_filter_lean(f, mode, lf.n_lead, lf.n_gated, lf.lead_delay, lf.tw_ok, lf.n_tw, pd->n_cus, s));
        return CAH_OK;
    }
    HIP_TRY(launch_filter(f, mode, mt.narrow_words != 0, pd->n_cus, s));
    return CAH_OK;
}

int cah_kmers_present_batch(const cah_plan* plan, int32_t adapter, const uint8_t* d_seqs,
                            const int64_t* d_offsets, const int32_t* d_lens, int64_t n_reads,
                            uint8_t* d_present, void* stream) {
    int rc = check_batch(plan, d_seqs, d_offsets, n_reads);
    if (rc) return rc;
    rc = check_adapter(plan, adapter);
    if (rc) return rc;
    if (n_reads == 0) return CAH_OK;
    if (!d_present) return fail(CAH_EINVAL, "present is NULL");
    hipStream_t s = (hipStream_t)stream;
    const CahMatcher& mt = plan->matchers[(size_t)adapter];
    if (!mt.has_filter) {   // MockKmerFinder: always present
        HIP_TRY(hipMemsetAsync(d_present, 1, (size_t)n_reads, s));
        return CAH_OK;
    }
    const PlanDeviceCopy* pd = nullptr;
    rc = plan_on_device(plan, &pd);
    if (rc) return rc;
    // private counter line: a per-thread, per-device allocation made once (the call synchronises the stream
    // before it returns, so consecutive calls of a thread never overlap)
    static thread_local unsigned long long* t_counter = nullptr;
    static thread_local int t_counter_device = -1;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (!t_counter || t_counter_device != dev) {
        t_counter = nullptr;
        HIP_TRY(hipMalloc((void**)&t_counter, 256));
        t_counter_device = dev;
    }
    rc = run_filter(plan, pd, adapter, d_seqs, d_offsets, d_lens, n_reads, 0, d_present, nullptr, nullptr,
                    nullptr, nullptr, t_counter, nullptr, s);
    hipError_t e = hipStreamSynchronize(s);
    if (rc) return rc;
    if (e != hipSuccess) return fail(CAH_EHIP, "k_filter failed: %s", hipGetErrorString(e));
    return CAH_OK;
}

// The fused multi-adapter path: ONE prefilter pass emits the (read, adapter) pairs whose kmers_present is
// true, the cost scan and the cell DP run over pairs (the adapter's match table is looked up per lane) and
// every match is merged into the read's best key with an atomic max (MultipleAdapters' order,
// kernels.h: pack_best); a last kernel decodes the keys.  status/out6/best_adapter were initialised by
// the caller; `extra` is the scratch behind the base workspace (cah_plan_workspace_bytes).
// the form the calling thread's last cah_match_batch* call took for its plan's adapters (cah_last_multi_path)
static thread_local int t_last_multi_path = -1;
int cah_last_multi_path(void) { return t_last_multi_path; }
// Deferred error check (per calling thread; off by default).  The streaming multi-adapter path reads the device's error
// word back after its kernels -- one synchronisation per call, so that a broken invariant is CAH_EINTERNAL.  A caller that
// issues many small batches on several streams switches the
// deferred form on: a call whose pool holds the batch's worst case returns without waiting, and k_multi_decode writes
// CAH_STATUS_INTERNAL into every status byte of a batch whose kernels flagged something -- the caller looks for it.
static thread_local int t_deferred_errors = 0;
int cah_set_deferred_errors(int on) { const int old = t_deferred_errors; t_deferred_errors = on ? 1 : 0; return old; }

static thread_local bool t_m2_waited_in_vain = false;     // the last failure of match_batch_multi_once was err bit 1 alone
static int match_batch_multi_once(const cah_plan* plan, const PlanDeviceCopy* pd, const uint8_t* d_seqs,
                                  const int64_t* d_offsets, const int32_t* d_lens, int64_t n_reads, int32_t* d_out6,
                                  int32_t* d_best_adapter, uint8_t* d_status, const Workspace& ws, char* extra,
                                  hipStream_t s, const UniformLayout ul);
// (a wave that gave up waiting for its tile -- err bit 1: nothing is wrong with the batch, the device was busy with
// something else for seconds -- is no reason to fail the call: the batch is matched once more from the start)
static int match_batch_multi(const cah_plan* plan, const PlanDeviceCopy* pd, const uint8_t* d_seqs,
                             const int64_t* d_offsets, const int32_t* d_lens, int64_t n_reads, int32_t* d_out6,
                             int32_t* d_best_adapter, uint8_t* d_status, const Workspace& ws, char* extra,
                             hipStream_t s, const UniformLayout ul = UniformLayout()) {
    t_m2_waited_in_vain = false;
    int rc = match_batch_multi_once(plan, pd, d_seqs, d_offsets, d_lens, n_reads, d_out6, d_best_adapter, d_status, ws, extra, s, ul);
    if (rc == CAH_EINTERNAL && t_m2_waited_in_vain) {
        t_m2_waited_in_vain = false;
        rc = match_batch_multi_once(plan, pd, d_seqs, d_offsets, d_lens, n_reads, d_out6, d_best_adapter, d_status, ws, extra, s, ul);
    }
    return rc;
}
static int match_batch_multi_once(const cah_plan* plan, const PlanDeviceCopy* pd, const uint8_t* d_seqs,
                                  const int64_t* d_offsets, const int32_t* d_lens, int64_t n_reads, int32_t* d_out6,
                                  int32_t* d_best_adapter, uint8_t* d_status, const Workspace& ws, char* extra,
                                  hipStream_t s, const UniformLayout ul) {
    const MultiPlan& mp = plan->multi;
    const int64_t A = (int64_t)plan->matchers.size();
    const int64_t cap = multi_pair_cap(plan, n_reads);
    const int64_t chunk = std::max<int64_t>(1, cap / A);
    unsigned long long* d_best_key = (unsigned long long*)extra;     extra += ws_key_bytes(n_reads);
    uint64_t* d_pairs = (uint64_t*)extra;                            extra += (size_t)cap * 8;
    int32_t* d_dpq = (int32_t*)extra;                                extra += (size_t)cap * 4;
    int32_t* d_win = (int32_t*)extra;                                extra += (size_t)cap * 8;
    uint16_t* d_wmeta = (uint16_t*)extra;                            // (streaming form: a 16-bit word per read, multi2.h)
    unsigned long long* counters = ws.counters;
    const CahMatcher& m0 = plan->matchers[0];
    HIP_TRY(hipMemsetAsync(d_best_key, 0, sizeof(unsigned long long) * (size_t)n_reads, s));
    // Equally long short reads take the streaming form (multi2.hip): k_multi_stream emits the pairs in pages of one
    // class each (the suffix compare of the error-free rows happens there), k_multi_scan scans them page by page.
    t_last_multi_path = CAH_MULTI_FUSED;
    const bool views = ul.inner && d_lens != nullptr && d_offsets != nullptr;    // views inside the reads of a uniform batch
    if (mp.m2.hdr.ok && ul.len > 0 && (!d_lens || (views && !env_flag("CAH_NO_MULTI2_VIEWS"))) &&
        multi2_read_len_ok(mp.m2.hdr, ul.len) && !env_flag("CAH_NO_MULTI2")) {
        t_last_multi_path = CAH_MULTI_STREAM;
        // the pool: pages of CAH_M2_PAGE pairs + one header word each, inside the pair area
        const int64_t max_pages = ((int64_t)cap * 8) / (CAH_M2_PAGE * 8 + 4);
        uint32_t* d_page_hdr = (uint32_t*)(d_pairs + max_pages * CAH_M2_PAGE);
        const int64_t TILE = multi2_tile_reads(), per_tile = m2_pages_per_tile(A), open_pages = 16 * CAH_M2_PAIR_CLASSES;
        // (pairs carry the read's index in 32 bits, the prefilter counts reads in an int)
        const int64_t BLOCK = (int64_t)1 << 30;
        bool deferred = false;
        for (int64_t lo = 0; lo < n_reads; lo += BLOCK) {
            const int64_t cnt = std::min(BLOCK, n_reads - lo);
            const int64_t n_tiles = (cnt + TILE - 1) / TILE;
            int64_t grid = std::min<int64_t>(n_tiles, pd->n_cus);
            int64_t gate = max_pages, rounds = 1;
            bool ungated = false;                                       // (developer builds: the test switch below)
            if (n_tiles * per_tile + grid * open_pages > max_pages) {
                // the batch might not fit the pool: blocks stop drawing tiles once what is in flight could fill it, and
                // the rounds go on until the tiles are done -- `rounds` is the count for the worst case (every adapter on
                // every read); a round that finds no tile left costs four empty launches
                grid = std::max<int64_t>(1, std::min(grid, (max_pages / 2) / m2_block_reserve(A)));
                gate = max_pages - grid * m2_block_reserve(A);
                const int64_t sure = std::max<int64_t>(1, (gate - grid * open_pages) / per_tile);
                rounds = (n_tiles + sure - 1) / sure;
            }
#ifdef CAH_DEV_KNOBS
            // libcutadapt_hip_dev.so only (cutadapt_amd/build.py: build_dev_library; tests/test_gpu_multi2.py): takes the
            // gate away to provoke the pool's overflow.  The product library does not hold this switch.
            if (env_flag("CAH_TEST_M2_UNGATED")) { gate = (int64_t)1 << 60; rounds = 1; ungated = true; }
#endif
            HIP_TRY(hipMemsetAsync(counters, 0, WS_HEADER, s));
            // (the pool holds the worst case of the whole batch, one launch draws every tile: nothing for the host to decide)
            deferred = t_deferred_errors && rounds == 1 && (gate == max_pages || ungated) && n_reads <= BLOCK;
          for (int64_t round = 0;; round++) {
            if (round >= rounds && deferred) break;
            if (round >= rounds) {
                // The planned rounds are through.  Never trust the arithmetic above silently: the device says how many
                // tiles were drawn and whether a kernel ran out of pages or waited in vain (kernels.h: Multi2Args::err).
                unsigned long long st[2] = {0ull, 0ull};                // {tiles drawn, error bits}
                HIP_TRY(hipMemcpyAsync(st, counters + WS_M2_TILE, sizeof(st), hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                t_m2_waited_in_vain = st[1] == 2ull;
                if (st[1])
                    return fail(CAH_EINTERNAL, "multi-adapter path: %s (pool of %lld pages, %lld adapters, %lld reads): results discarded",
                                (st[1] & 1ull) ? "the page pool ran out" : "a wave waited in vain for its tile",
                                (long long)max_pages, (long long)A, (long long)cnt);
                if ((int64_t)st[0] >= n_tiles) break;
                // tiles are left (the worst-case round count was too small): go on, a round at a time
                if (round >= rounds + 4 * n_tiles)
                    return fail(CAH_EINTERNAL, "multi-adapter path: %lld of %lld tiles left after %lld rounds",
                                (long long)(n_tiles - (int64_t)st[0]), (long long)n_tiles, (long long)round);
            }
            if (round) HIP_TRY(hipMemsetAsync(counters, 0, WS_M2_TILE * sizeof(unsigned long long), s));
            {
                Multi2Args f;
                f.uniform_first = ul.first; f.uniform_len = ul.len;
                f.win_hi = mp.m2.hdr.win_dist[M2_HI]; f.win_lo = mp.m2.hdr.win_dist[M2_LO];
                f.hdr = pd->d_m2hdr; f.dir = pd->d_m2dir; f.entries = pd->d_m2entries; f.bitmap = pd->d_m2bitmap;
                f.prefix = pd->d_m2prefix;
                f.seqs = d_seqs; f.first_read = lo; f.n_reads = cnt; f.status = d_status; f.best_key = d_best_key;
                f.pairs = d_pairs; f.page_hdr = d_page_hdr; f.page_counter = counters + WS_QCOUNT; f.max_pages = max_pages;
                f.tile_counter = counters + WS_M2_TILE; f.n_tiles = n_tiles; f.gate_pages = gate;
                f.err = counters + WS_M2_ERR;
                f.wmeta = d_wmeta;
                if (views) { f.view_starts = d_offsets; f.view_lens = d_lens; f.view_general = ul.general ? 1 : 0; }
                ProfScope ps(s, CAH_PROF_FILTER, round ? -1 : cnt);
                HIP_TRY(launch_multi_stream(f, mp.m2.hdr, (int)grid, s));
            }
            {
                Multi2ScanArgs sa;
                sa.uniform_first = ul.first; sa.uniform_len = ul.len;
                sa.kind = scan_word_kind(m0.m);
                sa.rows_lo = mp.m2.hdr.rows_lo;
                sa.matcher = pd->d_matchers; sa.tab = pd->d_mscan; sa.n_adapters = (int32_t)A;
                sa.seqs = d_seqs; sa.pairs = d_pairs; sa.page_hdr = d_page_hdr;
                sa.page_counter = counters + WS_QCOUNT; sa.max_pages = max_pages; sa.err = counters + WS_M2_ERR;
                sa.work_counter = counters + WS_SCANWORK; sa.best_key = d_best_key;
                sa.dp_queue = d_dpq; sa.dp_win = d_win;
                sa.dp_count_front = counters + WS_DPFRONT; sa.dp_count_back = counters + WS_DPBACK; sa.dp_cap = cap;
                sa.wmeta = d_wmeta; sa.prefix = pd->d_m2prefix; sa.lmax0 = mp.m2.hdr.lmax0;
                if (views) { sa.view_starts = d_offsets; sa.view_lens = d_lens; sa.view_general = ul.general ? 1 : 0; }
                // (how many pages there are is known on the device only: the blocks draw pages until none is left)
                ProfScope ps(s, CAH_PROF_SCAN, round ? -1 : cnt);
                HIP_TRY(launch_multi_scan(sa, std::min(max_pages, n_tiles * per_tile + grid * open_pages), pd->n_cus, s));
            }
            {
                // the cell DP over the pairs the scan left (the launch returns at once when there are none)
                DpArgs a;
                a.uniform_first = ul.first; a.uniform_len = views ? 0 : ul.len;     // (views: the cell DP reads starts + lengths)
                a.matcher = pd->d_matchers;
                a.seqs = d_seqs; a.offsets = d_offsets; a.lens = d_lens; a.n_reads = cnt * A;
                a.max_read_len = CAH_MAX_READ_LEN;
                a.queue_keys = nullptr;
                a.out6 = d_out6; a.status = d_status; a.best_adapter = d_best_adapter;
                a.adapter_index = 0; a.merge_best = 1;
                a.pairs = d_pairs; a.tab = pd->d_mrow; a.n_adapters = (int32_t)A; a.best_key = d_best_key;
                a.queue = d_dpq; a.queue_count = counters + WS_DPFRONT; a.queue_count_back = counters + WS_DPBACK;
                a.win = d_win; a.queue_cap = cap; a.work_counter = counters + WS_DPWORK;
                a.m2_hdr = pd->d_m2hdr; a.m2_ref_begin = pd->d_m2refbegin; a.m2_ref_list = pd->d_m2reflist;
                ProfScope ps(s, CAH_PROF_DP, round ? -1 : cnt);
                HIP_TRY(launch_dp(a, m0.m, true, true, std::min(cap, cnt * A), pd->n_cus, s));
            }
          }
        }
        ProfScope ps(s, CAH_PROF_MERGE, n_reads);
        HIP_TRY(launch_multi_decode(d_best_key, n_reads, d_out6, d_status, d_best_adapter, pd->n_cus, s,
                                    deferred ? counters + WS_M2_ERR : nullptr));
        return CAH_OK;
    }
    // (the older kernels take views through their starts and lengths alone)
    const UniformLayout ulo = ul.inner ? UniformLayout() : ul;
    for (int64_t lo = 0; lo < n_reads; lo += chunk) {
        const int64_t cnt = std::min(chunk, n_reads - lo);
        HIP_TRY(hipMemsetAsync(counters, 0, WS_HEADER, s));
        {
            MultiFilterArgs f;
            f.uniform_first = ulo.first; f.uniform_len = ulo.len;
            f.hdr = pd->d_mhdr; f.dir = pd->d_mdir; f.entries = pd->d_mentries; f.bitmap = pd->d_mbitmap;
            f.seqs = d_seqs; f.offsets = d_offsets; f.lens = d_lens;
            f.first_read = lo; f.n_reads = cnt; f.max_read_len = CAH_MAX_READ_LEN;
            f.work_counter = counters + 0; f.status = d_status;
            f.pairs = d_pairs; f.pair_count = counters + WS_QCOUNT; f.pair_cap = cap;
            ProfScope ps(s, CAH_PROF_FILTER, cnt);
            HIP_TRY(launch_multi_filter(f, mp.hdr, pd->n_cus, s));
        }
        {
            ScanArgs sa;
            sa.uniform_first = ulo.first; sa.uniform_len = ulo.len;
            sa.matcher = pd->d_matchers;
            sa.seqs = d_seqs; sa.offsets = d_offsets; sa.lens = d_lens; sa.n_reads = cnt * A;
            sa.max_read_len = CAH_MAX_READ_LEN;
            sa.queue = nullptr; sa.queue_count = counters + WS_QCOUNT; sa.queue_keys = nullptr;
            sa.work_counter = counters + WS_SCANWORK;
            sa.out6 = d_out6; sa.status = d_status; sa.best_adapter = d_best_adapter;
            sa.adapter_index = 0; sa.merge_best = 1;
            sa.pairs = d_pairs; sa.tab = pd->d_mscan; sa.n_adapters = (int32_t)A; sa.multi_skip_ok = mp.hdr.skip_ok;
            sa.best_key = d_best_key;
            sa.dp_queue = d_dpq; sa.dp_win = d_win;
            sa.dp_count_front = counters + WS_DPFRONT; sa.dp_count_back = counters + WS_DPBACK;
            sa.dp_cap = cap;
            sa.retry_threshold = 0; sa.retry_queue = nullptr; sa.retry_keys = nullptr; sa.retry_count = nullptr;
            sa.retry_cap = 0; sa.queue_limit = 0; sa.early_stop = 0; sa.tile = 0;
            sa.kind = scan_word_kind(plan->matchers[0].m);         // all adapters of the fused path have one shape
            ProfScope ps(s, CAH_PROF_SCAN, cnt);
            HIP_TRY(launch_back_scan(sa, cnt * A, pd->n_cus, s));
        }
        {
            DpArgs a;
            a.uniform_first = ulo.first; a.uniform_len = ulo.len;
            a.matcher = pd->d_matchers;
            a.seqs = d_seqs; a.offsets = d_offsets; a.lens = d_lens; a.n_reads = cnt * A;
            a.max_read_len = CAH_MAX_READ_LEN;
            a.queue = d_dpq; a.queue_count = counters + WS_DPFRONT; a.queue_keys = nullptr;
            a.work_counter = counters + WS_DPWORK;
            a.out6 = d_out6; a.status = d_status; a.best_adapter = d_best_adapter;
            a.adapter_index = 0; a.merge_best = 1;
            a.win = d_win; a.queue_count_back = counters + WS_DPBACK; a.queue_cap = cap;
            a.pairs = d_pairs; a.tab = pd->d_mrow; a.n_adapters = (int32_t)A; a.best_key = d_best_key;
            ProfScope ps(s, CAH_PROF_DP, cnt);
            HIP_TRY(launch_dp(a, m0.m, true, true, cnt * A, pd->n_cus, s));
        }
    }
    ProfScope ps(s, CAH_PROF_MERGE, n_reads);
    HIP_TRY(launch_multi_decode(d_best_key, n_reads, d_out6, d_status, d_best_adapter, pd->n_cus, s));
    return CAH_OK;
}

static int match_batch_impl(const cah_plan* plan, const uint8_t* d_seqs, const int64_t* d_offsets,
                            const int32_t* d_lens, const UniformLayout ul_in, int64_t n_reads, int32_t* d_out6,
                            int32_t* d_best_adapter, uint8_t* d_status, void* d_workspace, size_t workspace_bytes,
                            void* stream, const FrontFuse* fuse = nullptr, const CallHints* hints = nullptr) {
    const bool outputs_ready = hints && hints->outputs_ready;
    int rc = check_batch(plan, d_seqs, (ul_in.len > 0 && !ul_in.suffix) ? (const void*)d_seqs : (const void*)d_offsets, n_reads);
    if (rc) return rc;
    if (n_reads == 0) return CAH_OK;
    if (!d_out6 || !d_status) return fail(CAH_EINVAL, "output pointers are NULL");
    if (!d_workspace || workspace_bytes < cah_workspace_bytes(n_reads))
        return fail(CAH_EINVAL, "workspace too small: need %zu bytes", cah_workspace_bytes(n_reads));
    hipStream_t s = (hipStream_t)stream;
    const PlanDeviceCopy* pd = nullptr;
    rc = plan_on_device(plan, &pd);
    if (rc) return rc;
    // (cah_match_batch_frames: views anywhere in the buffer with a frame length -- only the streaming multi-adapter form
    // has a use for the frame; every other path takes them as the plain views they are)
    // (... a single adapter's streaming prefilter likewise, adapter by adapter: run_filter)
    const bool multi_form = plan->multi.hdr.ok && workspace_bytes >= cah_plan_workspace_bytes(plan, n_reads);
    const UniformLayout ul = (ul_in.general && multi_form &&
                              !(plan->multi.m2.hdr.ok && multi2_read_len_ok(plan->multi.m2.hdr, ul_in.len) && !env_flag("CAH_NO_MULTI2") &&
                                !env_flag("CAH_NO_MULTI2_VIEWS")))
                                 ? UniformLayout() : ul_in;
    const Workspace ws(d_workspace, n_reads, workspace_bytes);
    unsigned long long* counters = ws.counters;
    const UniformLayout ul_rest = ul.suffix ? UniformLayout() : ul;     // what every kernel but the prefilter sees
    // The result rows are zeroed by the first adapter's prefilter on its way through the batch when there is one
    // (FilterArgs::clear_out6: 24 B per read that would otherwise be a memset pass of its own), by a memset if not.
    const bool multi_path = plan->multi.hdr.ok && workspace_bytes >= cah_plan_workspace_bytes(plan, n_reads);
    int32_t first_aligner = -1;
    for (int32_t ad = 0; ad < (int32_t)plan->matchers.size() && first_aligner < 0; ad++)
        if (plan->matchers[(size_t)ad].kind != CAH_KIND_KMER_ONLY) first_aligner = ad;
    // (CAH_NO_FILTER_CLEAR=1: memsets instead, A/B)
    const bool filter_clears = !outputs_ready && !multi_path && first_aligner >= 0 &&
                               runs_filter(plan->matchers[(size_t)first_aligner]) && n_reads > CAH_TINY_BATCH &&
                               !(env_flag("CAH_NO_FILTER_CLEAR") && !fuse);
    if (!outputs_ready) {
        HIP_TRY(hipMemsetAsync(d_status, 0, (size_t)n_reads, s));
        // (the fused multi-adapter path's last kernel, k_multi_decode, writes every row, status and best adapter)
        if (!filter_clears && !multi_path) HIP_TRY(hipMemsetAsync(d_out6, 0, sizeof(int32_t) * 6 * (size_t)n_reads, s));
        if (d_best_adapter && !filter_clears && !multi_path) HIP_TRY(launch_init_best(d_best_adapter, n_reads, pd->n_cus, s));
    }
    // tiny single-adapter batches: one memset for all counters, no batch check (the ragged prefilter serves them)
    const bool tiny = n_reads <= CAH_TINY_BATCH && plan->matchers.size() == 1;
    if (tiny && !(hints && hints->precleaned_ws == d_workspace)) HIP_TRY(hipMemsetAsync(counters, 0, WS_HEADER, s));
    const bool header_fresh = tiny;
    // one pass over the offsets decides, on the device, which prefilter kernel works on this batch
    const unsigned long long* d_batch_flag = nullptr;
    if (!d_lens && !tiny && ul.len == 0 && !ul.suffix) {
        bool any_lean = false;
        for (size_t ad = 0; ad < plan->matchers.size(); ad++)
            any_lean |= runs_filter(plan->matchers[ad]) && plan->matchers[ad].kind != CAH_KIND_KMER_ONLY && plan->lean[ad].ok;
        if (any_lean) {
            HIP_TRY(hipMemsetAsync(counters + WS_UFLAG, 0, sizeof(unsigned long long), s));
            HIP_TRY(launch_uniform_check(d_offsets, n_reads, CAH_MAX_READ_LEN, counters + WS_UFLAG, pd->n_cus, s));
            d_batch_flag = counters + WS_UFLAG;
        }
    }
    t_last_multi_path = CAH_MULTI_SEQUENTIAL;
    if (plan->multi.hdr.ok && workspace_bytes >= cah_plan_workspace_bytes(plan, n_reads))
        // (views inside the reads of a uniform batch -- ul.inner: d_offsets / d_lens are the views' starts and lengths -- go
        // through with their layout: the streaming form takes them end-aligned, multi2.hip's RV form)
        return match_batch_multi(plan, pd, d_seqs, d_offsets, d_lens, n_reads, d_out6, d_best_adapter, d_status, ws,
                                 (char*)d_workspace + cah_workspace_bytes(n_reads), s, (ul.inner && d_lens) ? ul : ul_rest);
    for (int32_t ad = 0; ad < (int32_t)plan->matchers.size(); ad++) {
        const CahMatcher& mt = plan->matchers[(size_t)ad];
        if (mt.kind == CAH_KIND_KMER_ONLY) continue;
        if (runs_filter(mt)) {
            // prefilter -> queue of surviving reads -> (cost scan ->) DP on dense waves
            rc = run_filter(plan, pd, ad, d_seqs, d_offsets, d_lens, n_reads, 1, nullptr, d_status, ws.queue,
                            counters + WS_QCOUNT, ws.keys, counters + 0, d_batch_flag, s,
                            (filter_clears && ad == first_aligner) ? d_out6 : nullptr,
                            (filter_clears && ad == first_aligner) ? d_best_adapter : nullptr, ul,
                            ad == first_aligner ? fuse : nullptr, header_fresh);
            if (rc) return rc;
            // merge mode 2: the plan's first adapter writes into zeroed rows -- nothing to compare with (kernels.hip,
            // store_result)
            rc = run_aligner(plan, pd, ad, d_seqs, d_offsets, d_lens, n_reads, ws.queue, counters + WS_QCOUNT,
                             ws.keys, ws, d_out6, d_status, d_best_adapter, ad == first_aligner ? 2 : 1, s, ul_rest,
                             header_fresh);
        } else {
            rc = run_aligner(plan, pd, ad, d_seqs, d_offsets, d_lens, n_reads, nullptr, nullptr, nullptr,
                             ws, d_out6, d_status, d_best_adapter, 1, s, ul_rest, header_fresh);
        }
        if (rc) return rc;
    }
    return CAH_OK;
}

int cah_match_batch(const cah_plan* plan, const uint8_t* d_seqs, const int64_t* d_offsets,
                    const int32_t* d_lens, int64_t n_reads, int32_t* d_out6, int32_t* d_best_adapter,
                    uint8_t* d_status, void* d_workspace, size_t workspace_bytes, void* stream) {
    return match_batch_impl(plan, d_seqs, d_offsets, d_lens, UniformLayout(), n_reads, d_out6, d_best_adapter, d_status,
                            d_workspace, workspace_bytes, stream);
}

// Equally long reads, back to back, no offsets array: read r = d_seqs[r * read_len, (r + 1) * read_len) -- what a
// sequencer emits and what every BASELINE config is.  Nothing has to look at 8 bytes of offset per read to find
// that out (k_uniform_check), the prefilter is ONE launch instead of one per candidate kernel, and the scan / DP
// kernels compute a survivor's address instead of fetching it.
int cah_match_batch_uniform(const cah_plan* plan, const uint8_t* d_seqs, int32_t read_len, int64_t n_reads,
                            int32_t* d_out6, int32_t* d_best_adapter, uint8_t* d_status, void* d_workspace,
                            size_t workspace_bytes, void* stream) {
    if (read_len < 1 || read_len > CAH_MAX_READ_LEN)
        return fail(CAH_EINVAL, "read_len out of range (1..%d)", CAH_MAX_READ_LEN);
    if (n_reads > 0 && !d_seqs) return fail(CAH_EINVAL, "seqs is NULL");
    UniformLayout ul;
    ul.first = 0; ul.len = read_len;
    return match_batch_impl(plan, d_seqs, nullptr, nullptr, ul, n_reads, d_out6, d_best_adapter, d_status, d_workspace,
                            workspace_bytes, stream);
}

// Views that are SUFFIXES of the reads of a uniform batch: view r = d_seqs[d_starts[r], d_starts[r] + d_lens[r]) with
// d_starts[r] + d_lens[r] == (r + 1) * parent_read_len -- the second stage of a linked adapter (reference
// adapters.py:1222-1224 searches the 3' adapter in read[front_match.rstop:]) on what a sequencer emits.  Results are
// relative to the views, exactly as cah_match_batch(d_seqs, d_starts, d_lens, ...) returns them; the difference is
// that the prefilter can stream the parent's reads through LDS (k_filter_stream2) instead of fetching ragged views
// lane by lane.
int cah_match_batch_suffix_views(const cah_plan* plan, const uint8_t* d_seqs, const int64_t* d_starts,
                                 const int32_t* d_lens, int32_t parent_read_len, int64_t n_reads, int32_t* d_out6,
                                 int32_t* d_best_adapter, uint8_t* d_status, void* d_workspace, size_t workspace_bytes,
                                 void* stream) {
    if (parent_read_len < 1 || parent_read_len > CAH_MAX_READ_LEN)
        return fail(CAH_EINVAL, "parent_read_len out of range (1..%d)", CAH_MAX_READ_LEN);
    if (n_reads > 0 && (!d_starts || !d_lens)) return fail(CAH_EINVAL, "starts / lens are NULL");
    UniformLayout ul;
    ul.first = 0; ul.len = parent_read_len; ul.suffix = true;
    return match_batch_impl(plan, d_seqs, d_starts, d_lens, ul, n_reads, d_out6, d_best_adapter, d_status, d_workspace,
                            workspace_bytes, stream);
}

// Views ANYWHERE inside the reads of a uniform batch: view r = d_seqs[d_starts[r], d_starts[r] + d_lens[r]) with
// r * parent_read_len <= d_starts[r] and d_starts[r] + d_lens[r] <= (r + 1) * parent_read_len -- what a pipeline holds
// once a modifier in front of the adapter search has cut the reads of a sequencer's batch (quality trimming, -u, --length:
// reference modifiers.py QualityTrimmer / UnconditionalCutter / Shortener run before AdapterCutter).  Results relative
// to the views, as cah_match_batch(d_seqs, d_starts, d_lens, ...) returns them; the prefilter streams the parent's
// reads end-aligned (k_filter_stream2, RV form) instead of fetching ragged views lane by lane.
int cah_match_batch_views(const cah_plan* plan, const uint8_t* d_seqs, const int64_t* d_starts, const int32_t* d_lens,
                          int32_t parent_read_len, int64_t n_reads, int32_t* d_out6, int32_t* d_best_adapter,
                          uint8_t* d_status, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (parent_read_len < 1 || parent_read_len > CAH_MAX_READ_LEN)
        return fail(CAH_EINVAL, "parent_read_len out of range (1..%d)", CAH_MAX_READ_LEN);
    if (n_reads > 0 && (!d_starts || !d_lens)) return fail(CAH_EINVAL, "starts / lens are NULL");
    UniformLayout ul;
    ul.first = 0; ul.len = parent_read_len; ul.suffix = true; ul.inner = true;
    return match_batch_impl(plan, d_seqs, d_starts, d_lens, ul, n_reads, d_out6, d_best_adapter, d_status, d_workspace,
                            workspace_bytes, stream);
}

// Views ANYWHERE in d_seqs, none longer than frame_len characters: a packed batch with its offsets (d_starts[r] =
// offsets[r], d_lens[r] its length), the reads of a raw FASTQ chunk in place.  Same results as cah_match_batch(plan,
// d_seqs, d_starts, d_lens, ...); what the frame length buys: a plan of several adapters that has the streaming form
// takes the views end-aligned in frames of frame_len characters (multi2.hip, RV form with the gathering copy) instead
// of the per-lane kernels.  The caller guarantees d_lens[r] <= frame_len.
int cah_match_batch_frames(const cah_plan* plan, const uint8_t* d_seqs, const int64_t* d_starts, const int32_t* d_lens,
                           int32_t frame_len, int64_t n_reads, int32_t* d_out6, int32_t* d_best_adapter,
                           uint8_t* d_status, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (frame_len < 1 || frame_len > CAH_MAX_READ_LEN)
        return fail(CAH_EINVAL, "frame_len out of range (1..%d)", CAH_MAX_READ_LEN);
    if (n_reads > 0 && (!d_starts || !d_lens)) return fail(CAH_EINVAL, "starts / lens are NULL");
    UniformLayout ul;
    ul.first = 0; ul.len = frame_len; ul.suffix = true; ul.inner = true; ul.general = true;
    return match_batch_impl(plan, d_seqs, d_starts, d_lens, ul, n_reads, d_out6, d_best_adapter, d_status, d_workspace,
                            workspace_bytes, stream);
}

extern "C" int cah_linked_views(const int32_t*, const uint8_t*, const int64_t*, const int32_t*, int32_t, int64_t, int64_t*,
                                int32_t*, void*);

// can the front stage be folded into the back adapter's streaming prefilter?  One anchored 5' adapter that tolerates
// no error (k_anchored_exact's case) of at most 32 characters whose columns fit the first half-row of the slot, and
// one back adapter the SV form takes, on a batch large enough to be worth a filter that clears the result rows.
static bool linked_fusable(const cah_plan* front, const cah_plan* back, int32_t read_len, int64_t n_reads) {
    if (front->matchers.size() != 1 || back->matchers.size() != 1 || env_flag("CAH_NO_LINKED_FUSE")) return false;
    const CahMatcher& fm = front->matchers[0];
    const CahMatcher& bm = back->matchers[0];
    if (fm.kind != CAH_KIND_ALIGNER || fm.flags != 8 || !anchored_exact_ok(fm) || runs_filter(fm)) return false;
    const int units = (read_len + 15) / 16, first_half = 16 * ((units + 1) / 2);
    if (fm.m > 32 || fm.m + fm.k > first_half) return false;
    if (bm.kind == CAH_KIND_KMER_ONLY || !runs_filter(bm) || n_reads <= CAH_TINY_BATCH) return false;
    return stream2_suffix_ok(back, 0, read_len, n_reads);
}

// LinkedAdapter.match_to over a batch of equally long reads (reference adapters.py:1215-1227): the 5' plan on the
// reads, the 3' plan on read[front_match.rstop:] (the whole read where the 5' adapter was not found).  Outputs: the
// two stages' result arrays as cah_match_batch writes them (the back stage's coordinates are relative to its view) and
// the views themselves (d_starts int64[n], d_view_lens int32[n]); required / optional is the caller's verdict.
// When the 5' adapter is anchored and tolerates no error and the 3' adapter is the streaming prefilter's, ONE pass over
// the batch does the 5' comparison, the view arithmetic and the 3' prefilter (k_filter_stream2, FR form); otherwise
// the stages run one after the other.  One workspace serves both stages.
int cah_linked_match_batch_uniform(const cah_plan* front_plan, const cah_plan* back_plan, const uint8_t* d_seqs,
                                   int32_t read_len, int64_t n_reads, int32_t* d_out6_front, int32_t* d_best_front,
                                   uint8_t* d_status_front, int32_t* d_out6_back, int32_t* d_best_back,
                                   uint8_t* d_status_back, int64_t* d_starts, int32_t* d_view_lens, void* d_workspace,
                                   size_t workspace_bytes, void* stream) {
    if (!front_plan || !back_plan) return fail(CAH_EINVAL, "plan is NULL");
    if (read_len < 1 || read_len > CAH_MAX_READ_LEN)
        return fail(CAH_EINVAL, "read_len out of range (1..%d)", CAH_MAX_READ_LEN);
    if (n_reads < 0 || n_reads > 2147483647LL) return fail(CAH_EINVAL, "n_reads out of range (0..2^31-1)");
    if (n_reads == 0) return CAH_OK;
    if (!d_seqs || !d_out6_front || !d_status_front || !d_out6_back || !d_status_back || !d_starts || !d_view_lens)
        return fail(CAH_EINVAL, "cah_linked_match_batch_uniform: NULL argument");
    UniformLayout views;
    views.first = 0; views.len = read_len; views.suffix = true;
    if (linked_fusable(front_plan, back_plan, read_len, n_reads)) {
        const PlanDeviceCopy* fpd = nullptr;
        int rc = plan_on_device(front_plan, &fpd);
        if (rc) return rc;
        FrontFuse fuse;
        fuse.d_matcher = fpd->d_matchers;
        fuse.out6 = d_out6_front; fuse.status = d_status_front; fuse.best = d_best_front;
        fuse.starts = d_starts; fuse.lens = d_view_lens;
        return match_batch_impl(back_plan, d_seqs, d_starts, d_view_lens, views, n_reads, d_out6_back, d_best_back,
                                d_status_back, d_workspace, workspace_bytes, stream, &fuse);
    }
    UniformLayout ul;
    ul.first = 0; ul.len = read_len;
    int rc = match_batch_impl(front_plan, d_seqs, nullptr, nullptr, ul, n_reads, d_out6_front, d_best_front, d_status_front,
                              d_workspace, workspace_bytes, stream);
    if (rc) return rc;
    rc = cah_linked_views(d_out6_front, d_status_front, nullptr, nullptr, read_len, n_reads, d_starts, d_view_lens, stream);
    if (rc) return rc;
    return match_batch_impl(back_plan, d_seqs, d_starts, d_view_lens, views, n_reads, d_out6_back, d_best_back,
                            d_status_back, d_workspace, workspace_bytes, stream);
}

int cah_validate_ascii_batch(const uint8_t* d_seqs, const int64_t* d_offsets, const int32_t* d_lens,
                             int64_t n_reads, int32_t* d_bad, void* stream) {
    if (!d_bad) return fail(CAH_EINVAL, "d_bad is NULL");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(d_bad, 0, sizeof(int32_t), s));
    if (n_reads <= 0) return CAH_OK;
    int dev = 0, n_cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n_cus = v;
    }
    HIP_TRY(launch_validate(d_seqs, d_offsets, d_lens, n_reads, d_bad, n_cus, s));
    return CAH_OK;
}

// ---------------------------------------------------------------------------------------------
// host-pointer conveniences
// ---------------------------------------------------------------------------------------------
namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
};

// Per-thread persistent staging for the host-pointer conveniences: one grow-only device buffer, one pinned
// host buffer and a private stream, so that a small batch -- the per-read match_to()/locate() calls of the
// Python mirror classes send batches of one -- costs one H2D copy, the kernels and one D2H copy: no
// hipMalloc / hipFree / hipDeviceSynchronize per call.
struct HostScratch {
    const void* precleaned_ws = nullptr;    // the workspace whose counter header the last one-read call left zeroed (CallHints)
    int device = -1;
    hipStream_t stream = nullptr;
    char* dev = nullptr;  size_t dev_cap = 0;
    char* pin = nullptr;  size_t pin_cap = 0;
    int ensure(size_t dev_bytes, size_t pin_bytes) {
        int cur = 0;
        HIP_TRY(hipGetDevice(&cur));
        if (cur != device) { release(); device = cur; }
        if (!stream) HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        if (dev_bytes > dev_cap) {
            if (dev) (void)hipFree(dev);
            dev = nullptr; dev_cap = 0;
            const size_t want = dev_bytes + dev_bytes / 2 + 4096;
            HIP_TRY(hipMalloc((void**)&dev, want));
            dev_cap = want;
        }
        if (pin_bytes > pin_cap) {
            if (pin) (void)hipHostFree(pin);
            pin = nullptr; pin_cap = 0;
            const size_t want = pin_bytes + pin_bytes / 2 + 4096;
            HIP_TRY(hipHostMalloc((void**)&pin, want, hipHostMallocMapped));       // the one-read path lets the kernels read / write it
            pin_cap = want;
        }
        return CAH_OK;
    }
    void release() {
        if (dev) (void)hipFree(dev);
        if (pin) (void)hipHostFree(pin);
        if (stream) (void)hipStreamDestroy(stream);
        dev = nullptr; pin = nullptr; stream = nullptr; dev_cap = pin_cap = 0;
    }
    // no destructor work: at thread / process exit the HIP runtime may already be gone
};
thread_local HostScratch g_host_scratch;

static size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

enum { HOST_LOCATE = 0, HOST_PRESENT = 1, HOST_MATCH = 2 };

// ONE read through a one-adapter plan -- what Adapter.match_to(str) / Aligner.locate(str) of the reference's per-read
// API amount to.  No copies are queued at all: the kernels read the read from, and write the tuple to, mapped pinned
// host memory; the workspace lives at a fixed place of the thread's device scratch and its counters are cleared for
// the NEXT call while the host is away.  What is left on the critical path: the kernel launches and one stream wait.
static int host_call_one(int mode, const cah_plan* plan, const uint8_t* seq, int64_t n, int32_t* out6, uint8_t* status) {
    const size_t in_bytes = align16(2 * sizeof(int64_t) + (size_t)n);
    const size_t out_bytes = 64;                                 // out6 (24) + best (4) + status (1)
    const size_t ws_bytes = plan->max_long_m > 0 ? cah_plan_workspace_bytes(plan, 1) : cah_workspace_bytes(1);
    HostScratch& hs = g_host_scratch;
    const char* dev_before = hs.dev;
    int rc = hs.ensure(ws_bytes + 256, in_bytes + out_bytes);
    if (rc) return rc;
    if (hs.dev != dev_before) hs.precleaned_ws = nullptr;
    int64_t* h_off = (int64_t*)hs.pin;
    h_off[0] = 0; h_off[1] = n;
    if (n > 0) memcpy(hs.pin + 2 * sizeof(int64_t), seq, (size_t)n);
    char* h_out = hs.pin + in_bytes;
    memset(h_out, 0, out_bytes);
    char* d_pin = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void**)&d_pin, hs.pin, 0));
    const int64_t* d_offsets = (const int64_t*)d_pin;
    const uint8_t* d_seqs = (const uint8_t*)(d_pin + 2 * sizeof(int64_t));
    int32_t* d_out6 = (int32_t*)(d_pin + in_bytes);
    int32_t* d_best = (int32_t*)(d_pin + in_bytes + 24);
    uint8_t* d_status = (uint8_t*)(d_pin + in_bytes + 28);
    char* d_ws = hs.dev;                                         // always the same place: the marker below stays valid
    // One 3' aligner with the cost scan (every plain -a adapter): prefilter + scan in ONE single-wave launch
    // (k_tiny); only a read the scan cannot finish costs a second launch (the cell DP over the list k_tiny left).
    const CahMatcher& mt0 = plan->matchers[0];
    if (mt0.kind == CAH_KIND_ALIGNER && mt0.scan_ok && !mt0.long_dp &&
        (mode == HOST_LOCATE || !mt0.has_filter || plan->lean[0].ok) && !getenv("CAH_NO_TINY")) {
        const PlanDeviceCopy* pd = nullptr;
        rc = plan_on_device(plan, &pd);
        if (rc) return rc;
        const Workspace ws(d_ws, 1, ws_bytes);
        hs.precleaned_ws = nullptr;                              // k_tiny sets the counters it and the cell DP use itself
        int32_t* h_need = (int32_t*)(h_out + 32);
        TinyArgs ta;
        const bool filter = mode == HOST_MATCH && mt0.has_filter;
        ta.lean = filter ? pd->d_lean : nullptr;
        ta.matcher = pd->d_matchers;
        ta.seqs = d_seqs; ta.offsets = d_offsets; ta.n_reads = 1; ta.max_read_len = CAH_MAX_READ_LEN;
        ta.out6 = d_out6; ta.status = d_status;
        ta.dp_queue = ws.dp_queue; ta.dp_win = ws.dp_win;
        ta.dp_count_front = ws.counters + WS_DPFRONT; ta.dp_count_back = ws.counters + WS_DPBACK;
        ta.dp_work = ws.counters + WS_DPWORK; ta.dp_cap = 1;
        ta.need_dp = (int32_t*)(d_pin + in_bytes + 32);
        const CahLeanFilter& lf = plan->lean[0];
        // completion: a kernel's last store is a ticket in the mapped block; polling it costs a fraction of a
        // stream wait (which stays as the fallback)
        static thread_local int32_t t_ticket = 0;
        volatile int32_t* h_done = (volatile int32_t*)(h_out + 36);
        int32_t* d_done = (int32_t*)(d_pin + in_bytes + 36);
        auto wait_ticket = [&](int32_t ticket) -> int {
            const auto t0 = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (*h_done != ticket) {
                if ((++spins & 0xFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                    HIP_TRY(hipStreamSynchronize(hs.stream));
                    break;
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            return CAH_OK;
        };
        t_ticket = t_ticket >= 0x7ffffff0 ? 1 : t_ticket + 1;
        ta.done = d_done; ta.ticket = t_ticket;
        // the kernel's LDS tables come as one image from HBM; the plan's first call (per device and mode) leaves it there
        {
            int device = 0;
            HIP_TRY(hipGetDevice(&device));
            std::lock_guard<std::mutex> lk(plan->mu);
            PlanDeviceCopy& dc = plan->dev[device];
            void*& img = dc.d_tiny_image[filter ? 1 : 0];
            if (!img) {
                void* fresh = nullptr;
                HIP_TRY(hipMalloc(&fresh, CAH_TINY_IMAGE_BYTES));
                TinyArgs tb = ta;
                tb.image = nullptr; tb.image_out = fresh;
                HIP_TRY(launch_tiny(tb, filter ? lf.n_lead : 1, filter ? lf.n_gated : 1, filter ? lf.lead_delay : 0, hs.stream));
                HIP_TRY(hipStreamSynchronize(hs.stream));
                img = fresh;
            }
            ta.image = img; ta.image_out = nullptr;
        }
        HIP_TRY(launch_tiny(ta, filter ? lf.n_lead : 1, filter ? lf.n_gated : 1, filter ? lf.lead_delay : 0, hs.stream));
        if ((rc = wait_ticket(t_ticket))) return rc;
        if (*h_need > 0) {
            DpArgs a;
            a.matcher = pd->d_matchers;
            a.seqs = d_seqs; a.offsets = d_offsets; a.lens = nullptr; a.n_reads = 1; a.max_read_len = CAH_MAX_READ_LEN;
            a.queue = ws.dp_queue; a.queue_keys = nullptr; a.win = ws.dp_win;
            a.queue_count = ws.counters + WS_DPFRONT; a.queue_count_back = ws.counters + WS_DPBACK; a.queue_cap = 1;
            a.work_counter = ws.counters + WS_DPWORK;
            a.out6 = d_out6; a.status = d_status; a.best_adapter = nullptr; a.adapter_index = 0; a.merge_best = 0;
            a.pairs = nullptr; a.tab = nullptr; a.n_adapters = 0; a.best_key = nullptr;
            HIP_TRY(launch_dp(a, mt0.m, mt0.indel_cost == 1, mt0.flags == 14, 1, pd->n_cus, hs.stream));
            t_ticket += 1;
            HIP_TRY(launch_ticket(d_done, t_ticket, hs.stream));
            if ((rc = wait_ticket(t_ticket))) return rc;
        }
        memcpy(out6, h_out, 24);
        *status = *(const uint8_t*)(h_out + 28);
        return CAH_OK;
    }
    if (mode == HOST_LOCATE) {
        hs.precleaned_ws = nullptr;                              // run_aligner clears what it needs itself
        rc = cah_locate_batch(plan, 0, d_seqs, d_offsets, nullptr, 1, d_out6, d_status, d_ws, ws_bytes, hs.stream);
    } else {
        CallHints hints;
        hints.outputs_ready = true;                              // (h_out was zeroed above: the outputs live in mapped host memory)
        hints.precleaned_ws = hs.precleaned_ws;
        rc = match_batch_impl(plan, d_seqs, d_offsets, nullptr, UniformLayout(), 1, d_out6, d_best, d_status, d_ws, ws_bytes,
                              hs.stream, nullptr, &hints);
        // the next call's counters (tiny path of the batch entry), off its critical path
        if (rc == CAH_OK && hipMemsetAsync(d_ws, 0, WS_HEADER, hs.stream) == hipSuccess) hs.precleaned_ws = d_ws;
        else hs.precleaned_ws = nullptr;
    }
    if (rc) { hs.precleaned_ws = nullptr; return rc; }
    // the memset queued above is not waited for: an event after the kernels would do, but the stream wait below is
    // cheaper than creating one -- it covers the memset as well (a few hundred nanoseconds of GPU time)
    HIP_TRY(hipStreamSynchronize(hs.stream));
    memcpy(out6, h_out, 24);
    *status = *(const uint8_t*)(h_out + 28);
    return CAH_OK;
}

// seqs/offsets in, (out6, best, status | present) out; everything staged through g_host_scratch
static int host_call(int mode, const cah_plan* plan, int32_t adapter, const uint8_t* seqs, const int64_t* offsets,
                     int64_t n, int32_t* out6, int32_t* best_adapter, uint8_t* status) {
    if (n == 1 && plan->matchers.size() == 1 && !best_adapter && offsets[0] == 0 && offsets[1] >= 0 &&
        (mode == HOST_MATCH || (mode == HOST_LOCATE && adapter == 0)) && (offsets[1] == 0 || seqs))
        return host_call_one(mode, plan, seqs, offsets[1], out6, status);
    g_host_scratch.precleaned_ws = nullptr;                      // this path lays the device scratch out differently
    const int64_t total = offsets[n];
    if (offsets[0] != 0) return fail(CAH_EINVAL, "offsets[0] must be 0");
    if (total > 0 && !seqs) return fail(CAH_EINVAL, "seqs is NULL");
    for (int64_t i = 0; i < n; i++)
        if (offsets[i + 1] < offsets[i]) return fail(CAH_EINVAL, "offsets must be non-decreasing");
    const size_t off_bytes = sizeof(int64_t) * (size_t)(n + 1);
    const size_t in_bytes = align16(off_bytes + (size_t)total);
    const size_t o6_bytes = sizeof(int32_t) * 6 * (size_t)n, best_bytes = sizeof(int32_t) * (size_t)n;
    const size_t out_bytes = align16(o6_bytes + best_bytes + (size_t)n);
    // the host conveniences keep to the base scratch unless the plan cannot do without more (long adapters)
    const size_t ws_bytes = plan->max_long_m > 0 ? cah_plan_workspace_bytes(plan, n) : cah_workspace_bytes(n);
    HostScratch& hs = g_host_scratch;
    int rc = hs.ensure(in_bytes + out_bytes + 256 + ws_bytes, in_bytes + out_bytes);
    if (rc) return rc;
    memcpy(hs.pin, offsets, off_bytes);
    if (total > 0) memcpy(hs.pin + off_bytes, seqs, (size_t)total);
    char* d_in = hs.dev;
    char* d_out = hs.dev + in_bytes;
    char* d_ws = hs.dev + ((in_bytes + out_bytes + 255) & ~(size_t)255);
    HIP_TRY(hipMemcpyAsync(d_in, hs.pin, off_bytes + (size_t)total, hipMemcpyHostToDevice, hs.stream));
    const int64_t* d_offsets = (const int64_t*)d_in;
    const uint8_t* d_seqs = (const uint8_t*)(d_in + off_bytes);
    int32_t* d_out6 = (int32_t*)d_out;
    int32_t* d_best = (int32_t*)(d_out + o6_bytes);
    uint8_t* d_status = (uint8_t*)(d_out + o6_bytes + best_bytes);
    if (mode == HOST_LOCATE)
        rc = cah_locate_batch(plan, adapter, d_seqs, d_offsets, nullptr, n, d_out6, d_status, d_ws, ws_bytes, hs.stream);
    else if (mode == HOST_PRESENT)
        rc = cah_kmers_present_batch(plan, adapter, d_seqs, d_offsets, nullptr, n, d_status, hs.stream);
    else {
        // one memset clears out6 / best / status (they are contiguous here); "no adapter" (-1) is filled in below
        HIP_TRY(hipMemsetAsync(d_out, 0, o6_bytes + best_bytes + (size_t)n, hs.stream));
        CallHints hints;
        hints.outputs_ready = true;
        rc = match_batch_impl(plan, d_seqs, d_offsets, nullptr, UniformLayout(), n, d_out6, d_best, d_status, d_ws, ws_bytes,
                              hs.stream, nullptr, &hints);
    }
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(hs.pin + in_bytes, d_out, o6_bytes + best_bytes + (size_t)n, hipMemcpyDeviceToHost, hs.stream));
    HIP_TRY(hipStreamSynchronize(hs.stream));
    const char* h_out = hs.pin + in_bytes;
    if (out6) memcpy(out6, h_out, o6_bytes);
    if (status) memcpy(status, h_out + o6_bytes + best_bytes, (size_t)n);
    if (best_adapter) {
        memcpy(best_adapter, h_out + o6_bytes, best_bytes);
        if (mode == HOST_MATCH) {
            const uint8_t* st = (const uint8_t*)(h_out + o6_bytes + best_bytes);
            for (int64_t i = 0; i < n; i++) if (st[i] != CAH_MATCH) best_adapter[i] = -1;
        }
    }
    return CAH_OK;
}
}  // namespace

int cah_locate_batch_host(const cah_plan* plan, int32_t adapter, const uint8_t* seqs,
                          const int64_t* offsets, int64_t n_reads, int32_t* out6, uint8_t* status) {
    int rc = check_batch(plan, seqs, offsets, n_reads);
    if (rc) return rc;
    rc = check_adapter(plan, adapter);
    if (rc) return rc;
    if (n_reads == 0) return CAH_OK;
    if (!out6 || !status) return fail(CAH_EINVAL, "output pointers are NULL");
    return host_call(HOST_LOCATE, plan, adapter, seqs, offsets, n_reads, out6, nullptr, status);
}

int cah_kmers_present_batch_host(const cah_plan* plan, int32_t adapter, const uint8_t* seqs,
                                 const int64_t* offsets, int64_t n_reads, uint8_t* present) {
    int rc = check_batch(plan, seqs, offsets, n_reads);
    if (rc) return rc;
    rc = check_adapter(plan, adapter);
    if (rc) return rc;
    if (n_reads == 0) return CAH_OK;
    if (!present) return fail(CAH_EINVAL, "present is NULL");
    return host_call(HOST_PRESENT, plan, adapter, seqs, offsets, n_reads, nullptr, nullptr, present);
}

int cah_match_batch_host(const cah_plan* plan, const uint8_t* seqs, const int64_t* offsets,
                         int64_t n_reads, int32_t* out6, int32_t* best_adapter, uint8_t* status) {
    int rc = check_batch(plan, seqs, offsets, n_reads);
    if (rc) return rc;
    if (n_reads == 0) return CAH_OK;
    if (!out6 || !status) return fail(CAH_EINVAL, "output pointers are NULL");
    return host_call(HOST_MATCH, plan, -1, seqs, offsets, n_reads, out6, best_adapter, status);
}

// Adapter.match_to(str) / Aligner.locate(str): one read, no offsets array (see host_call_one)
int cah_match_one_host(const cah_plan* plan, const uint8_t* seq, int64_t n, int32_t* out6, uint8_t* status) {
    if (!plan) return fail(CAH_EINVAL, "plan is NULL");
    if (n < 0 || (n > 0 && !seq) || !out6 || !status) return fail(CAH_EINVAL, "cah_match_one_host: bad argument");
    if (plan->matchers.size() != 1) return fail(CAH_EINVAL, "cah_match_one_host: the plan must hold one adapter");
    return host_call_one(HOST_MATCH, plan, seq, n, out6, status);
}

int cah_locate_one_host(const cah_plan* plan, const uint8_t* seq, int64_t n, int32_t* out6, uint8_t* status) {
    if (!plan) return fail(CAH_EINVAL, "plan is NULL");
    if (n < 0 || (n > 0 && !seq) || !out6 || !status) return fail(CAH_EINVAL, "cah_locate_one_host: bad argument");
    if (plan->matchers.size() != 1) return fail(CAH_EINVAL, "cah_locate_one_host: the plan must hold one adapter");
    if (plan->matchers[0].kind == CAH_KIND_KMER_ONLY) return fail(CAH_EINVAL, "adapter 0 has no aligner");
    return host_call_one(HOST_LOCATE, plan, seq, n, out6, status);
}

// ---------------------------------------------------------------------------------------------
// Aligner.enable_debug(): one read through the statement-by-statement kernel (long.hip, any adapter length) with
// the DP matrices written out -- the reference fills DPMatrix objects while locate() runs (_align.pyx:58-92,
// :279-296, :385-390, :485-489).  Synchronous, allocates what it needs: a debugging aid, not a hot path.
// ---------------------------------------------------------------------------------------------
int cah_locate_debug_host(const cah_adapter_desc* adapter, const uint8_t* seq, int64_t n, int32_t* out6,
                          uint8_t* status, int32_t* cost_matrix, int32_t* score_matrix) {
    try {
        if (!adapter || !out6 || !status || !cost_matrix || !score_matrix || (n > 0 && !seq))
            return fail(CAH_EINVAL, "cah_locate_debug_host: NULL argument");
        if (adapter->kind != CAH_KIND_ALIGNER) return fail(CAH_EINVAL, "only an Aligner has DP matrices");
        if (n < 0 || n > CAH_MAX_READ_LEN) return fail(CAH_EINVAL, "read length out of range");
        if (adapter->length < 0 || (adapter->length > 0 && !adapter->sequence)) return fail(CAH_EINVAL, "bad sequence");
        if (!is_ascii(adapter->sequence, (size_t)adapter->length)) return fail(CAH_EINVAL, "String must contain only ASCII characters");
        CahMatcher mt;
        memset(&mt, 0, sizeof(mt));
        LongTables lt;
        const int rc = build_long(*adapter, 0, mt, lt);
        if (rc != CAH_OK) return rc;
        int device = 0;
        HIP_TRY(hipGetDevice(&device));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(CAH_EUNSUPPORTED, "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        const int m = adapter->length;
        const size_t cells = (size_t)(m + 1) * (size_t)(n + 1);
        const int64_t lanes = 256;
        struct Bufs {
            void* p[10] = {nullptr};
            ~Bufs() { for (void* q : p) if (q) (void)hipFree(q); }
        } b;
        auto dev = [&](int slot, size_t bytes) -> int { HIP_TRY(hipMalloc(&b.p[slot], std::max<size_t>(bytes, 16))); return CAH_OK; };
        int e;
        if ((e = dev(0, sizeof(CahLongMatcher))) || (e = dev(1, lt.ref.size())) || (e = dev(2, sizeof(int32_t) * lt.ncnt.size())) ||
            (e = dev(3, (size_t)n)) || (e = dev(4, 2 * sizeof(int64_t))) || (e = dev(5, sizeof(unsigned long long))) ||
            (e = dev(6, sizeof(int32_t) * 3 * (size_t)(m + 1) * (size_t)lanes)) || (e = dev(7, 6 * sizeof(int32_t) + 16)) ||
            (e = dev(8, sizeof(int32_t) * cells)) || (e = dev(9, sizeof(int32_t) * cells)))
            return e;
        HIP_TRY(hipMemcpy(b.p[0], &lt.lm, sizeof(CahLongMatcher), hipMemcpyHostToDevice));
        if (!lt.ref.empty()) HIP_TRY(hipMemcpy(b.p[1], lt.ref.data(), lt.ref.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b.p[2], lt.ncnt.data(), sizeof(int32_t) * lt.ncnt.size(), hipMemcpyHostToDevice));
        if (n) HIP_TRY(hipMemcpy(b.p[3], seq, (size_t)n, hipMemcpyHostToDevice));
        const int64_t offs[2] = {0, n};
        HIP_TRY(hipMemcpy(b.p[4], offs, sizeof(offs), hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(b.p[5], 0, sizeof(unsigned long long)));
        // the caller's matrices go down as they are: cells the algorithm does not compute keep the caller's marker
        HIP_TRY(hipMemcpy(b.p[8], cost_matrix, sizeof(int32_t) * cells, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b.p[9], score_matrix, sizeof(int32_t) * cells, hipMemcpyHostToDevice));
        LongArgs la;
        la.lm = (const CahLongMatcher*)b.p[0]; la.ref = (const uint8_t*)b.p[1]; la.ncnt = (const int32_t*)b.p[2];
        la.seqs = (const uint8_t*)b.p[3]; la.offsets = (const int64_t*)b.p[4]; la.lens = nullptr;
        la.n_reads = 1; la.max_read_len = CAH_MAX_READ_LEN;
        la.queue = nullptr; la.queue_count = nullptr; la.work_counter = (unsigned long long*)b.p[5];
        la.scratch = (int32_t*)b.p[6];
        la.out6 = (int32_t*)b.p[7]; la.status = (uint8_t*)b.p[7] + 6 * sizeof(int32_t);
        la.best_adapter = nullptr; la.adapter_index = 0; la.merge_best = 0;
        la.dbg_cost = (int32_t*)b.p[8]; la.dbg_score = (int32_t*)b.p[9];
        HIP_TRY(launch_dp_long(la, lanes, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out6, b.p[7], 6 * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(status, (uint8_t*)b.p[7] + 6 * sizeof(int32_t), 1, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(cost_matrix, b.p[8], sizeof(int32_t) * cells, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(score_matrix, b.p[9], sizeof(int32_t) * cells, hipMemcpyDeviceToHost));
        return CAH_OK;
    } catch (const std::exception& ex) {
        return fail(CAH_EINVAL, "cah_locate_debug_host: %s", ex.what());
    } catch (...) {
        return fail(CAH_EINVAL, "cah_locate_debug_host: unknown error");
    }
}

// ---------------------------------------------------------------------------------------------
// synthetic reads
// ---------------------------------------------------------------------------------------------
int cah_synth_reads(uint64_t seed, int64_t first_index, int64_t n_reads, int32_t read_len,
                    uint32_t p_adapter_u32, uint32_t p_edit_u32, uint32_t p_n_u16, const char* adapters,
                    const int32_t* adapter_off, int32_t n_adapters, uint8_t* d_seqs, int64_t* d_offsets,
                    void* stream) {
    if (n_reads < 0 || read_len < 0) return fail(CAH_EINVAL, "negative size");
    if (n_adapters < 0 || n_adapters > 4096) return fail(CAH_EINVAL, "n_adapters out of range");
    if (n_adapters > 0 && (!adapters || !adapter_off)) return fail(CAH_EINVAL, "adapters is NULL");
    if (!d_seqs || !d_offsets) return fail(CAH_EINVAL, "output pointers are NULL");
    hipStream_t s = (hipStream_t)stream;
    DevBuf d_ad, d_off;
    const int total = n_adapters > 0 ? adapter_off[n_adapters] : 0;
    HIP_TRY(d_ad.alloc((size_t)total));
    HIP_TRY(d_off.alloc(sizeof(int32_t) * (size_t)(n_adapters + 1)));
    if (total > 0) HIP_TRY(hipMemcpyAsync(d_ad.p, adapters, (size_t)total, hipMemcpyHostToDevice, s));
    int32_t zero = 0;
    HIP_TRY(hipMemcpyAsync(d_off.p, n_adapters > 0 ? adapter_off : &zero, sizeof(int32_t) * (size_t)(n_adapters + 1),
                           hipMemcpyHostToDevice, s));
    int dev = 0, n_cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n_cus = v;
    }
    HIP_TRY(launch_synth(seed, first_index, n_reads, read_len, p_adapter_u32, p_edit_u32, p_n_u16,
                         (const char*)d_ad.p, (const int32_t*)d_off.p, n_adapters, d_seqs, d_offsets, n_cus, s));
    HIP_TRY(hipStreamSynchronize(s));   // d_ad/d_off are freed on return
    return CAH_OK;
}

}  // extern "C"
