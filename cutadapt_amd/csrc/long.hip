// long.hip -- Aligner.locate / PrefixComparer / SuffixComparer for adapters LONGER than 64 characters.
//
// The register-resident kernels (kernels.hip) hold the DP column in VGPRs and the match relation in one
// 64-bit bitset per read character, which bounds the adapter at 64 characters.  The reference has no such
// bound (it allocates a column of any length, src/cutadapt/_align.pyx:250-257, and only the k-mer
// prefilter gives up beyond 64, adapters.py:633-639), so longer adapters -- long primers, linked
// constructs -- take this kernel: one read per lane, the column (cost, score, origin per row) in HBM
// scratch laid out [row][field][lane] so that the lanes of a wave touch consecutive addresses, the
// algorithm written as in the reference, statement by statement (all flag combinations, indel costs,
// wildcard modes, stale cells and the stale `origin` of the last-column scan included: the column
// simply persists, like the reference's).  Throughput is secondary here; exactness is not.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cutadapt_hip.h"
#include "cah_device.h"
#include "kernels.h"
#include "dev_common.h"

namespace {

struct Column {
    int32_t* base;       // element (row i, field f) at base[(i * 3 + f) * stride]
    int64_t stride;
    __device__ __forceinline__ int& cost(int i) const { return base[(int64_t)(i * 3 + 0) * stride]; }
    __device__ __forceinline__ int& score(int i) const { return base[(int64_t)(i * 3 + 1) * stride]; }
    __device__ __forceinline__ int& origin(int i) const { return base[(int64_t)(i * 3 + 2) * stride]; }
};


// Aligner.locate / PrefixComparer.locate / SuffixComparer.locate of ONE read, written as in the reference; the DP
// column lives wherever `col` points (HBM scratch).  `seen` ORs the bytes looked at (bit 7 set = non-ASCII input).
struct LongResult { bool found; int t0, t1, t2, t3, score, cost; unsigned seen; };

// dbg_cost / dbg_score (NULL in production): the matrices of Aligner.enable_debug() (_align.pyx:385-390, :485-489),
// (m + 1) x (n + 1) row-major, entries the algorithm never computes keep what the caller put there.
__device__ __forceinline__ LongResult long_locate(const CahLongMatcher* lm, const uint8_t* ref, const int32_t* ncnt,
                                                  const uint8_t* s_qtab, const Column& col, const uint8_t* q, const int n,
                                                  int32_t* dbg_cost = nullptr, int32_t* dbg_score = nullptr) {
    const int m = lm->m, k = lm->k, D = lm->indel_cost, kind = lm->kind;
    const bool start_in_ref = lm->flags & 1, start_in_query = lm->flags & 2;
    const bool stop_in_ref = lm->flags & 4, stop_in_query = lm->flags & 8;
    const bool cmp_equal = lm->cmp_equal != 0, wildcard_ref = lm->wildcard_ref != 0;
    const int min_overlap = lm->min_overlap, eff_full = lm->effective_length;
    const double rate = lm->rate;
        bool found = false;
        int t0 = 0, t1 = 0, t2 = 0, t3 = 0, r_score = 0, r_cost = 0;
        unsigned seen = 0;

        if (kind != CAH_KIND_ALIGNER) {
            // PrefixComparer / SuffixComparer (_align.pyx:651-714): Hamming distance over min(m, n) characters
            const bool suffix = kind == CAH_KIND_SUFFIX;
            const int length = min(m, n);
            int errors = 0;
            for (int x = 0; x < length; ++x) {
                const int ri = suffix ? m - length + x : x;
                const unsigned c = q[suffix ? n - length + x : x];
                seen |= c;
                const uint8_t qc = s_qtab[c & 127], rc = ref[ri];
                const bool eq = cmp_equal ? rc == qc : (rc & qc) != 0;
                errors += eq ? 0 : 1;
            }
            found = !(errors > lm->cmp_max_k || length < min_overlap);
            r_score = length - 2 * errors; r_cost = errors;
            if (!suffix) { t0 = 0; t1 = length; t2 = 0; t3 = length; }
            else { t0 = m - length; t1 = m; t2 = n - length; t3 = n; }
        } else {
            int max_n = n, min_n = 0;                                      // :346-352
            if (!start_in_query) max_n = min(n, m + k);
            if (!stop_in_query) min_n = max(0, n - m - k);
            for (int i = 0; i <= m; ++i) {                                 // first column (:364-383)
                int sc, co, og;
                if (!start_in_ref && !start_in_query) { sc = -2 * i; co = max(i, min_n) * D; og = 0; }
                else if (start_in_ref && !start_in_query) { sc = 0; co = min_n * D; og = min(0, min_n - i); }
                else if (!start_in_ref && start_in_query) { sc = -2 * i; co = i * D; og = max(0, min_n - i); }
                else { sc = 0; co = min(i, min_n) * D; og = min_n - i; }
                col.cost(i) = co; col.score(i) = sc; col.origin(i) = og;
                if (dbg_cost) { dbg_cost[(int64_t)i * (n + 1) + min_n] = co; dbg_score[(int64_t)i * (n + 1) + min_n] = sc; }
            }
            const int SENT = m + n + 1;                                    // :394
            int b_refstop = m, b_qstop = n, b_cost = SENT, b_origin = 0, b_score = 0;
            int last = min(m, k + 1);                                      // :399-401
            if (start_in_ref) last = m;
            int last_filled = 0;
            int cost = 0, score = 0, origin = 0;                           // the reference's scalar locals
            const int o_inc = start_in_query ? 1 : 0, c_inc = start_in_query ? 0 : D, s_inc = start_in_query ? 0 : -2;
            for (int j = min_n + 1; j <= max_n; ++j) {                     // :433
                int dc = col.cost(0), ds = col.score(0), dor = col.origin(0);
                col.origin(0) = dor + o_inc; col.cost(0) = dc + c_inc; col.score(0) = ds + s_inc;
                const unsigned c = q[j - 1];
                seen |= c;
                const uint8_t qc = s_qtab[c & 127];
                int pc = col.cost(0), ps = col.score(0), po = col.origin(0);      // cell above (this column)
                for (int i = 1; i <= last; ++i) {                          // :441
                    const int oc = col.cost(i), os = col.score(i), oo = col.origin(i);
                    const uint8_t rc = ref[i - 1];
                    const bool eq = cmp_equal ? rc == qc : (rc & qc) != 0;
                    if (eq) {                                              // :446-453
                        cost = dc; origin = dor; score = ds + 1;
                    } else {                                               // :455-476
                        const int c_diag = dc + 1, c_ins = oc + D, c_del = pc + D;
                        if (c_diag <= c_del && c_diag <= c_ins) { cost = c_diag; origin = dor; score = ds - 1; }
                        else if (c_del <= c_ins) { cost = c_del; origin = po; score = ps - 2; }
                        else { cost = c_ins; origin = oo; score = os - 2; }
                    }
                    dc = oc; ds = os; dor = oo;                            // :479
                    col.cost(i) = cost; col.score(i) = score; col.origin(i) = origin;
                    pc = cost; ps = score; po = origin;
                }
                last_filled = last;                                        // :484
                if (dbg_cost)                                              // :485-489
                    for (int i = 0; i <= last; ++i) {
                        dbg_cost[(int64_t)i * (n + 1) + j] = col.cost(i);
                        dbg_score[(int64_t)i * (n + 1) + j] = col.score(i);
                    }
                while (last >= 0 && col.cost(last) > k) --last;            // :490-491
                if (last < m) {
                    ++last;
                } else if (stop_in_query) {                                // :496-533
                    cost = col.cost(m); score = col.score(m); origin = col.origin(m);
                    const int length = m + min(origin, 0);
                    int eff = length;
                    if (wildcard_ref) eff = length < m ? length - (ncnt[m] - ncnt[m - length]) : eff_full;
                    const bool ok = length >= min_overlap && (double)cost <= eff * rate;
                    const int best_len = m + min(b_origin, 0);
                    if (ok && (b_cost == SENT || (origin <= b_origin + m / 2 && score > b_score) ||
                               (length > best_len && score > b_score))) {
                        b_score = score; b_cost = cost; b_origin = origin; b_refstop = m; b_qstop = j;
                        if (cost == 0 && origin >= 0) break;               // :531-533
                    }
                }
            }
            if (max_n == n) {                                              // :536-572
                const int first_i = stop_in_ref ? 0 : m;
                for (int i = last_filled; i >= first_i; --i) {
                    const int oi = col.origin(i), ci = col.cost(i), si = col.score(i);
                    const int length = i + min(oi, 0);
                    const int lo = -min(oi, 0);
                    int eff = length;
                    if (wildcard_ref) eff = length < m ? length - (ncnt[i] - ncnt[lo]) : eff_full;
                    const bool ok = length >= min_overlap && (double)ci <= eff * rate;
                    const int best_len = b_refstop + min(b_origin, 0);
                    // NB: `origin` is the stale scalar, not the cell's (:565)
                    if (ok && (b_cost == SENT || (origin <= b_origin + m / 2 && si > b_score) ||
                               (length > best_len && si > b_score))) {
                        b_score = si; b_cost = ci; b_origin = oi; b_refstop = i; b_qstop = n;
                    }
                }
            }
            found = b_cost != SENT;
            t0 = b_origin >= 0 ? 0 : -b_origin; t1 = b_refstop;
            t2 = b_origin >= 0 ? b_origin : 0;  t3 = b_qstop;
            r_score = b_score; r_cost = b_cost;
        }
        LongResult res;
        res.found = found; res.t0 = t0; res.t1 = t1; res.t2 = t2; res.t3 = t3; res.score = r_score; res.cost = r_cost; res.seen = seen;
        return res;
}

}  // namespace

__global__ __launch_bounds__(256) void k_dp_long(LongArgs a) {
    __shared__ uint8_t s_qtab[CAH_TABLE_CHARS];
    const CahLongMatcher* lm = a.lm;
    for (int i = threadIdx.x; i < CAH_TABLE_CHARS; i += blockDim.x) s_qtab[i] = lm->qtab[i];
    __syncthreads();
    const uint8_t* ref = a.ref;
    const int32_t* ncnt = a.ncnt;
    Column col;
    col.stride = (int64_t)gridDim.x * blockDim.x;
    col.base = a.scratch + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
    const int lane = wave_lane();
    int64_t total = a.n_reads;
    if (a.queue_count) total = (int64_t)(*a.queue_count);

    for (;;) {
        const int64_t base = wave_dequeue(a.work_counter);
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
        const LongResult lr = long_locate(lm, ref, ncnt, s_qtab, col, q, n, a.dbg_cost, a.dbg_score);
        const bool found = lr.found;
        const int t0 = lr.t0, t1 = lr.t1, t2 = lr.t2, t3 = lr.t3, r_score = lr.score, r_cost = lr.cost;
        const unsigned seen = lr.seen;
        if (seen & 0x80u) invalid = true;

        int32_t* o = a.out6 + r * 6;
        if (a.merge_best) {
            if (invalid) {
                a.status[r] = 2;
            } else if (found) {
                const bool had = a.status[r] == 1;
                if (a.status[r] != 2 && (!had || r_score > o[4] || (r_score == o[4] && r_cost < o[5]))) {
                    o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = r_score; o[5] = r_cost;
                    a.status[r] = 1;
                    if (a.best_adapter) a.best_adapter[r] = a.adapter_index;
                }
            }
        } else {
            a.status[r] = invalid ? (uint8_t)2 : (found ? (uint8_t)1 : (uint8_t)0);
            if (found && !invalid) { o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = r_score; o[5] = r_cost; }
            else { o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0; }
        }
    }
}

int64_t long_scratch_lanes(int64_t max_items, int n_cus) {
    int64_t need = (max_items + 255) / 256;
    if (need < 1) need = 1;
    const int64_t cap = (int64_t)2 * n_cus;
    return (need < cap ? need : cap) * 256;
}

hipError_t launch_dp_long(const LongArgs& a, int64_t lanes, hipStream_t s) {
    hipLaunchKernelGGL(k_dp_long, dim3((unsigned)(lanes / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
