// back_scan.h -- bit-parallel COST scan of a 3' adapter (flags = QUERY_START | QUERY_STOP |
// REFERENCE_END, unit costs) over one read, and the classification built on it.
//
// Why: Aligner.locate (reference src/cutadapt/_align.pyx:298-587) spends its time filling DP cells
// with (cost, score, origin).  The COSTS alone follow Myers'/Hyyro's bit-vector recurrence -- one
// 64-bit word holds a whole column of vertical deltas -- at ~1/10 of the instructions of the cell
// loop, independent of the band.  From the costs alone one can decide, exactly:
//
//   * which columns j have C(m, j) <= k          -> where the reference evaluates a candidate in the
//                                                   last row (:496-533)
//   * which rows i of the last column are acceptable (cost <= thr, i >= min_overlap; :536-572)
//
// and that is enough to finish most reads without any cell DP:
//
//   NONE        no acceptable candidate anywhere                        -> locate() returns None
//   EXACT_FULL  the first column whose row-m cost is 0 (the adapter occurs unedited) wins and the
//               reference `break`s there (:531-533); a cost-0 alignment has no indels, so
//               score = m, origin = j - m                               -> (0, m, j-m, j, m, 0)
//   EXACT_TAIL  no acceptable last-row candidate, and the read ends with adapter[0:i] unedited: row i of the
//               last column is acceptable and costs 0, i the largest such row.  The downward scan of the last
//               column (:536-572) takes the largest acceptable row first; a later (smaller) row replaces the
//               best iff its score is higher and `origin <= best.origin + m // 2` (:563-567; the `length >`
//               clause cannot hold for a smaller row).  Smaller rows than i have score <= their row < i: they
//               never replace row i.  LARGER acceptable rows i' (cost c' >= 1, e.g. row i + 1 reached by one
//               deletion -- acceptable as soon as thr(i + 1) >= 1, which is the rule for tails of 9+
//               characters at rate 0.1) score at most i' - 2c' (see SUBS_FULL); if that is < i for all of them,
//               the best is below i when the scan reaches row i, and row i replaces it PROVIDED the origin
//               clause holds.  `origin` there is a stale variable: the origin of the last cell the main loop
//               computed, cell (r_s, n) with r_s = last_filled >= every acceptable row (:484).  That cell costs
//               <= k + 1 (its diagonal neighbour costs <= k by the definition of `last`) and is a real path
//               (stale neighbours cost > k and never win), so its origin is <= n - r_s + k + 1; the best's
//               origin is >= n - i' - c' >= n - r_s - c'.  Hence origin - best.origin <= k + 1 + c', and the
//               clause holds when k + 1 + c' <= m // 2 for every larger acceptable row (checked; 7 <= 16 for
//               a 33-character adapter at rate 0.1).                     -> (0, i, n-i, n, i, 0)
//               The same replay works with SUBSTITUTIONS in the tail: a row i whose diagonal into (i, n) is clean
//               (bit i of A in the last column; cost-0 rows are clean by themselves) carries exactly
//               (c_i, i - 2 c_i, n - i), like (m, e) of SUBS_FULL.  Let W be the clean acceptable row of the highest
//               score (the largest such row on ties).  Rows with an unclean diagonal score at most i - 2 c_i; if that
//               is < score(W) for all of them, then going down the column the best is below score(W) when row W
//               is reached (clean rows above it score less by the choice of W, unclean ones by their bound), W
//               replaces it (origin clause as above, for every acceptable row with errors), and nothing below W
//               scores more.                                           -> (0, W, n-W, n, W-2c, c)
//   SUBS_FULL   the adapter occurs with substitutions only -- what sequencing errors are: c = the smallest row-m
//               cost of any column, 1 <= c <= kacc, first reached at column e, and the diagonal that ends in
//               (m, e) is "clean": every cell on it whose characters differ costs one more than its diagonal
//               predecessor (bit-vector D0 = 0 there), so the reference's cells on it take the diagonal
//               (match: unconditionally, :446-453; mismatch: the diagonal candidate is minimal and wins ties,
//               :462-476) and (m, e) carries origin e - m and score m - 2c.  Any alignment of the whole adapter
//               with cost x scores at most m - 2x (score = m - 2 mismatches - 3 deletions - 2 insertions), so
//               earlier acceptable columns (cost >= c + 1) score less and column e replaces them when
//               e - jfa <= m/2 - kacc (the origin clause, as for EXACT_FULL); later columns (cost >= c) cannot
//               score more; a row i of the last column could only win with i - 2 C(i, n) > m - 2c, which is
//               checked.  No `break` (cost > 0).                          -> (0, m, e-m, e, m-2c, c)
//   INDEL1_FULL the adapter occurs with ONE insertion or ONE deletion and otherwise substitutions only (32-bit forms):
//               c = the smallest row-m cost of any column, first reached at column e, and the diagonal d = e - m that
//               ends in (m, e) is NOT clean.  Walk the reference's cell (m, e) back up that diagonal: cells whose
//               characters match take the diagonal (:446-453), differing cells with diagonal delta +1 take it too
//               (the diagonal candidate is minimal and wins ties, :462-476) -- until the LOWEST unclean cell (i*, j*)
//               (characters differ, diagonal delta 0: the diagonal candidate costs one too many).  There the
//               reference takes the cell above if that attains the minimum (vertical delta +1: bit i* of the new
//               VP) -- a deletion, the path goes on from (i*-1, j*) on diagonal d + 1 -- else the cell to the left
//               -- an insertion, on from (i*, j*-1) on diagonal d - 1.  If the diagonal of THAT cell is clean up to
//               it (its accumulator bit: A of this column one row up / A of the last column), the rest of the path
//               is substitutions only, so the alignment has c - 1 substitutions and one indel:
//               deletion: origin d + 1, score m - 2c - 1; insertion: origin d - 1, score m - 2c.  Both facts ride
//               down every diagonal like A does, overwritten at each unclean cell (the lowest one counts):
//               U' = ((U << 1) & ~X) | (VP' & X),  Z' = ((Z << 1) & ~X) | (pa & X),  pa = VP' ? A' << 1 : A.
//               Earlier acceptable columns cost >= c + 1 and score <= m - 2c - 2: column e replaces them when the
//               origin clause holds with the deletion's origin, e + 1 - jfa <= m/2 - kacc.  Later columns cost
//               >= c: an insertion's m - 2c is the most they can score, so nothing replaces it; a deletion could
//               lose to a later column of cost c, so there must be none (checked).  Rows of the last column are
//               checked as for SUBS_FULL.       -> (0, m, e-m+1, e, m-2c-1, c) / (0, m, e-m-1, e, m-2c, c)
//   (early stop) Last-row candidates are replaced only by candidates that OVERLAP them: (:521-524) the new origin
//               must be <= best.origin + m // 2 (all last-row candidates have length m, so the `length >` clause
//               is dead once a best exists).  A candidate of column j with cost c <= kacc has its origin in
//               [j - m - c, j - m + c].  So after an acceptable column jla, a later acceptable column j2 is
//               irrelevant when j2 - jla > 2 kacc + m // 2 -- it cannot replace a best that stems from a column
//               <= jla, and neither can anything after it.  The last-column scan needs
//               `origin(stale) <= best.origin + m // 2` too (the `length >` clause: i > m never holds), and the
//               stale origin is >= n - m - k - 1 (real path of cost <= k + 1 over <= m rows): it cannot update
//               when n - jla > k + 1 + kacc + m // 2 =: gap.  Therefore, once `gap` columns after jla have
//               shown no acceptable candidate and the read goes on, the result is decided by the columns
//               <= jla alone: the scan stops there (bs_may_stop, checked once per 16-column chunk), SUBS_FULL
//               needs no look at the last column, and the DP window ends at jla without the last-column scan.
//   DP          everything else: the cell kernel runs, but only over the columns that can matter:
//               from (first acceptable candidate column, or n) - m - k - 1 -- the windowing argument
//               of DESIGN.md "Column skipping" with the exact position instead of the k-mer hit --
//               and, when no row of the last column is acceptable, only up to the last acceptable
//               candidate column (the last-column scan would change nothing and is skipped).
//
// The recurrence (Hyyro 2003, "search" variant: row 0 costs 0 in every column = QUERY_START):
//     Xv = Eq | VN;  Xh = (((Eq & VP) + VP) ^ VP) | Eq
//     HP = VN | ~(Xh | VP);  HN = VP & Xh
//     VP' = (HN << 1) | ~(Xv | (HP << 1));  VN' = (HP << 1) & Xv
// The adapter sits in the TOP m bits of the word; the 64 - m bits below it are "rows" that match
// every character (Eq bit 1, vertical delta 0): they cost 0 in every column, exactly like row 0, so
// row m is always bit 63 and its horizontal delta is the sign of the high half.
//
// The same code is compiled by hipcc into k_back_scan (kernels.hip) and by g++ into the host model
// that tests/test_back_scan_model.py fuzzes against the oracle (test infrastructure; the product has
// no CPU path).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CAH_HD __host__ __device__ __forceinline__
#else
#define CAH_HD inline
#endif

#define CAH_BS_ALL_ROWS 64
enum { BS_NONE = 0, BS_EXACT_FULL = 1, BS_EXACT_TAIL = 2, BS_DP = 3, BS_SUBS_FULL = 4, BS_INDEL1_FULL = 5 };

struct BackScanParams {
    int m;            // adapter length, 1..64
    int k;            // (int)(rate * m), 0 <= k < m
    int kacc;         // thr[effective_length]: a last-row candidate is acceptable iff cost <= kacc (<= k);
                      // -1 when m < min_overlap (never acceptable)
    int min_overlap;  // >= 1
    int half_m;       // m / 2
};

// what the scan remembers about row m, whatever the representation of the column
struct BackScanBook {
    int cm;           // C(m, j) of the column just processed
    int jfa, jla;     // first / last column with cm <= kacc (-1: none)
    // SUBS_FULL: cmin / je: smallest acceptable row-m cost so far and the first column that reached it;
    // eclean: that column's diagonal was clean (see A below).
    int cmin, je;
    bool eclean;
    bool edel;        // ONE_INDEL: the lowest unclean cell of that diagonal was reached from the cell above (a deletion)
    bool epred_unclean;   // ... and the diagonal of the cell it was reached from is not clean either (or not tracked)
    bool emore;       // a later column reached cmin again
};

struct BackScanState : BackScanBook {
    uint64_t VP, VN;
    // SUBS_FULL: A accumulates, along every diagonal, the cells whose diagonal delta is 0 although their characters
    // differ (an insertion / deletion path is as cheap as the diagonal): bit 63 = the diagonal ending in row m of the
    // current column.
    uint64_t A;
};

// The 32-bit form of the same scan (half the instructions of the 64-bit form on a 32-bit ALU):
//   X == 0  adapters of at most 32 characters: the adapter in the top m bits of a 32-bit word, pad rows below;
//   X >= 1  adapters of 32 + X characters (X = 1, 2 -- e.g. the 33-character TruSeq adapter): the FIRST X rows as
//           plain integers -- row 1 of a "search" column costs 0 where the characters match and 1 where they do not,
//           row r <= X follows the textbook recurrence from the row above -- and rows X+1..m as a 32-bit word without
//           pad rows whose top boundary is row X instead of the constant row 0: Myers' / Hyyro's block step with a
//           horizontal input delta hin = C(X, j) - C(X, j-1) (hin < 0 sets bit 0 of Eq inside Xh and shifts a one
//           into HN, hin > 0 into HP).  Row m is the word's top bit, as in the X == 0 form.  The diagonal
//           accumulators (A, U, Z) take row X's bits of the last column where the 64-bit form shifts in zeros.
template <int X>
struct BackScanState32 : BackScanBook {
    uint32_t VP, VN, A;
    uint32_t U, Z;                 // ONE_INDEL, of each diagonal's lowest unclean cell: U = it is left upwards (a
                                   // deletion), Z = the diagonal of the cell it is left to is unclean too
    int cx[X > 0 ? X : 1];         // X > 0: C(1 + t, j), the explicit rows
    unsigned ax;                   // X > 0: their diagonal bits: bit t = A, bit 8 + t = U, bit 16 + t = Z of row 1 + t
};

CAH_HD uint64_t bs_shl1(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    // 64-bit shifts are slow on gfx950; by one bit: low half doubles, high half is a funnel shift
    const unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
    const unsigned nhi = __builtin_amdgcn_alignbit(hi, lo, 31);
    return ((uint64_t)nhi << 32) | (uint64_t)(lo + lo);
#else
    return x << 1;
#endif
}


// ---- instruction forms (round 5).  Issue cost on gfx950 with two or more waves per SIMD (profiles/r03/valu_ubench.txt,
// DESIGN 3.2b): v_and / v_or / v_xor / v_add / v_sub / v_mov / v_ashrrev / v_bitop3 take 2 cycles per wave64, shifts,
// v_lshl_or, v_and_or, v_or3, v_bfi, v_add3, every v_cmp and v_cndmask 4.  The compiler writes "x << 1" as v_lshlrev,
// (m & a) | (~m & b) as v_bfi and (a & b) | c as v_and_or: a 32-bit column of the scan then spends half of its ~95 issue
// cycles on 4-cycle forms.  Said explicitly (device code only; the host model keeps plain C): ~66 cycles.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CAH_BS_PLAIN_OPS)
CAH_HD uint32_t bs_dbl(const uint32_t x) {                    // x << 1 as x + x
    uint32_t r;
    asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(x));
    return r;
}
#define BS_BITOP3(a, b, c, tt) __builtin_amdgcn_bitop3_b32((a), (b), (c), (tt))
#else
CAH_HD uint32_t bs_dbl(const uint32_t x) { return x << 1; }
CAH_HD uint32_t bs_bitop3_host(const uint32_t a, const uint32_t b, const uint32_t c, const unsigned tt) {
    uint32_t r = 0;                                            // truth table: bit (a << 2 | b << 1 | c) of tt
    for (unsigned idx = 0; idx < 8; ++idx)
        if ((tt >> idx) & 1u)
            r |= ((idx & 4u) ? a : ~a) & ((idx & 2u) ? b : ~b) & ((idx & 1u) ? c : ~c);
    return r;
}
#define BS_BITOP3(a, b, c, tt) bs_bitop3_host((a), (b), (c), (tt))
#endif
CAH_HD uint32_t bs_sel(const uint32_t m, const uint32_t a, const uint32_t b) { return BS_BITOP3(m, a, b, 0xCAu); }   // m ? a : b, bitwise

// first column of the window: costs 0 (pad rows), 1, 2, ..., m
CAH_HD void bs_init(BackScanState& s, const BackScanParams& p) {
    const int pad = 64 - p.m;
    s.VP = pad == 0 ? ~0ull : ~((1ull << pad) - 1ull);
    s.VN = 0;
    s.cm = p.m;
    s.jfa = -1; s.jla = -1;
    s.A = 0; s.cmin = 1 << 20; s.je = -1; s.eclean = false; s.edel = false; s.epred_unclean = true; s.emore = false;
}

// Row m of the column just processed: the bookkeeping every representation shares.  clean: the diagonal that ends
// in (m, j) has met no cell whose diagonal delta is 0 although its characters differ.  Returns true when the read
// is finished as EXACT_FULL at this column.
// jlim: columns behind it are not booked (for callers whose window runs a few columns past the last one they answer for;
// round 5's k_back_scan3 was one -- removed in round 6, DESIGN_HISTORY.md).
template <bool SUBS>
CAH_HD bool bs_book(BackScanBook& s, const bool clean, const int j, const BackScanParams& p, const bool del = false,
                    const bool pred_unclean = true, const int jlim = 0x7FFFFFFF) {
    if (s.cm <= p.kacc && j <= jlim) {
        if (s.jfa < 0) s.jfa = j;
        s.jla = j;
        if (SUBS && s.cm == s.cmin) s.emore = true;
        if (SUBS && s.cm < s.cmin) {
            s.cmin = s.cm; s.je = j; s.eclean = clean; s.edel = del; s.epred_unclean = pred_unclean; s.emore = false;
        }
        // (:521-533) a cost-0 candidate has score m, more than any earlier best (those cost >= 1: the
        // first cost-0 column ends the loop), so it replaces the best iff it is the first or
        // origin = j - m <= best.origin + m/2.  Every earlier acceptable candidate sits at a column
        // >= jfa and spans rel <= m + kacc columns, i.e. best.origin >= jfa - m - kacc.
        if (s.cm == 0 && j - s.jfa <= p.half_m - p.kacc) return true;
    }
    return false;
}

// One column.  eq: the padded match word of this read character (CahMatcher::scanmask[c]).
// Returns true when the read is finished as EXACT_FULL at this column.
// SUBS = false leaves the SUBS_FULL bookkeeping out (the class then never applies): the fused multi-adapter scan
// runs ~6 (read, adapter) pairs per read, most of them ending as NONE, and is cheaper without it.
// BOOK = false: the recurrence alone -- for windows in which no column but the last can hold an acceptable candidate
// (the tail pairs of the streaming multi-adapter path, multi2.h): row m's cost is not followed, nothing is booked, the
// rows of the last column are read off the state by bs_finish as ever (jfa stays -1).
template <bool SUBS = true, bool BOOK = true>
CAH_HD bool bs_step(BackScanState& s, const uint64_t eq, const int j, const BackScanParams& p) {
    const uint64_t VP = s.VP, VN = s.VN;
    const uint64_t Xv = eq | VN;
    const uint64_t Xh = (((eq & VP) + VP) ^ VP) | eq;
    const uint64_t HP = VN | ~(Xh | VP);
    const uint64_t HN = VP & Xh;
    // row m is bit 63: horizontal delta +1 / -1
    if (BOOK) s.cm += (int)(HP >> 63) - (int)(HN >> 63);
    const uint64_t HPs = bs_shl1(HP), HNs = bs_shl1(HN);
    s.VP = HNs | ~(Xv | HPs);
    s.VN = HPs & Xv;
    if (!BOOK) return false;
    // D0 = Xh | VN: C(i, j) == C(i-1, j-1); set where the characters differ = an indel path is as cheap
    if (SUBS) s.A = bs_shl1(s.A) | ((Xh | VN) & ~eq);
    if (__builtin_expect(s.cm <= p.kacc, 0)) return bs_book<SUBS>(s, (s.A >> 63) == 0, j, p);     // rare: off the straight path
    return false;
}

// ---- the 32-bit form ----------------------------------------------------------------------------------------
// which form serves an adapter of m characters: 0 = 64-bit, 1 = 32-bit, 2 / 3 = 32-bit + 1 / 2 explicit rows
CAH_HD int bs_kind_of(const int m) { return m <= 32 ? 1 : (m <= 34 ? m - 31 : 0); }

// the two table words of a character from its 64-bit match word (adapter in the top m bits, pad rows below):
// eq32 = rows X+1..m (X > 0) or the top 32 bits (X == 0), eqx = rows 1..X in bits 0..
CAH_HD uint64_t bs32_table_entry(const uint64_t sm, const int m) {
    if (m <= 32) return sm >> 32;
    const int x = m - 32;                                          // explicit rows 1..x sit in bits 64-m .. 64-m+x-1
    const uint32_t lo = (uint32_t)(sm >> (64 - m + x)), hi = (uint32_t)(sm >> (64 - m)) & ((1u << x) - 1u);
    return ((uint64_t)hi << 32) | lo;
}

template <int X>
CAH_HD void bs32_init(BackScanState32<X>& s, const BackScanParams& p) {
    const int pad = X > 0 ? 0 : 32 - p.m;
    s.VP = pad == 0 ? ~0u : ~((1u << pad) - 1u);
    s.VN = 0; s.A = 0; s.ax = 0;
    for (int t = 0; t < (X > 0 ? X : 1); ++t) s.cx[t] = 1 + t;
    s.cm = p.m;
    s.jfa = -1; s.jla = -1;
    s.cmin = 1 << 20; s.je = -1; s.eclean = false; s.edel = false; s.epred_unclean = true; s.emore = false;
    s.Z = 0; s.U = 0;
}

template <bool SUBS, int X, bool BOOK = true>
// (joff: added to j only where the column number is needed -- the booking of an acceptable column, off the straight path;
// a kernel that unrolls sixteen columns passes its chunk's first column and the constants 1..16 instead of counting)
CAH_HD bool bs32_step(BackScanState32<X>& s, const uint32_t eq, const uint32_t eqx, const int j, const BackScanParams& p,
                      const int jlim = 0x7FFFFFFF, const int joff = 0) {
    // ---- the explicit rows 1..X and what they hand to the word: hin, and the bits that enter its diagonals
    uint32_t hpos = 0, hneg = 0, a_in = 0, u_in = 0, z_in = 0, a_top_new = 0;
    if (X == 1) {
        // row 1: C(1, j) = [characters differ] (row 0 costs 0 everywhere); a differing cell there has diagonal
        // delta +1 -- never unclean -- so nothing but the horizontal delta enters the word
        const uint32_t neq = ~eqx & 1u, prev = (uint32_t)s.cx[0];
        hpos = neq & ~prev; hneg = prev & ~neq;
        s.cx[0] = (int)neq;
    } else if (X > 0) {
        int up_prev = 0, up = 0;                                   // C(r-1, j-1), C(r-1, j): row 0 costs 0
        unsigned a_above = 0, a_above_new = 0, u_above = 0, z_above = 0;   // row r-1: column j-1 / column j
        unsigned ax_new = 0, ux_new = 0, zx_new = 0;
        const int c_last_old = s.cx[X - 1];
        for (int t = 0; t < X; ++t) {
            const int neq = (int)(~(eqx >> t) & 1u);
            const int cprev = s.cx[t];                             // C(r, j-1)
            int c = up_prev + neq;
            if (up + 1 < c) c = up + 1;
            if (cprev + 1 < c) c = cprev + 1;
            const unsigned x = (c == up_prev && neq) ? 1u : 0u;    // diagonal delta 0 although the characters differ
            const unsigned a_row_old = (s.ax >> t) & 1u, u_row_old = (s.ax >> (8 + t)) & 1u, z_row_old = (s.ax >> (16 + t)) & 1u;
            const unsigned a_new = a_above | x;
            const unsigned isdel = c == up + 1 ? 1u : 0u;
            const unsigned u_new = x ? isdel : u_above;
            const unsigned z_new = x ? (isdel ? a_above_new : a_row_old) : z_above;
            ax_new |= a_new << t; ux_new |= u_new << t; zx_new |= z_new << t;
            a_above = a_row_old; a_above_new = a_new; u_above = u_row_old; z_above = z_row_old;
            up_prev = cprev; up = c;
            s.cx[t] = c;
        }
        // row X of the LAST column enters the word's diagonals; its horizontal delta is the word's input
        a_in = a_above; u_in = u_above; z_in = z_above; a_top_new = a_above_new;
        const int hin = s.cx[X - 1] - c_last_old;
        hpos = hin > 0 ? 1u : 0u; hneg = hin < 0 ? 1u : 0u;
        if (SUBS) s.ax = ax_new | (ux_new << 8) | (zx_new << 16);
    }
    // ---- the word (every three-input form as ONE v_bitop3, every shift by one as an addition: see bs_dbl)
    const uint32_t VP = s.VP, VN = s.VN;
    const uint32_t Xv = eq | VN;
    const uint32_t eqm = X > 0 ? (eq | hneg) : eq;
    const uint32_t t = (eqm & VP) + VP;
    const uint32_t Xh = BS_BITOP3(t, VP, eqm, 0xBEu);              // (t ^ VP) | eqm
    const uint32_t HP = BS_BITOP3(VN, Xh, VP, 0xF1u);              // VN | ~(Xh | VP)
    const uint32_t HN = VP & Xh;
    // row m is the top bit: +1 where HP has it, -1 where HN has it (arithmetic shifts: 0 / -1)
    if (BOOK) s.cm = s.cm - ((int)HP >> 31) + ((int)HN >> 31);
    const uint32_t HPs = X > 0 ? ((HP << 1) | hpos) : bs_dbl(HP), HNs = X > 0 ? ((HN << 1) | hneg) : bs_dbl(HN);
    s.VP = BS_BITOP3(HNs, Xv, HPs, 0xF1u);                         // HNs | ~(Xv | HPs)
    s.VN = HPs & Xv;
    if (!BOOK) return false;
    if (SUBS) {
        const uint32_t a_old = s.A;
        const uint32_t Xc = BS_BITOP3(Xh, VN, eq, 0x54u);          // (Xh | VN) & ~eq: unclean cells (diagonal delta 0, characters differ)
        s.A = X > 1 ? ((bs_dbl(a_old) | a_in) | Xc) : (bs_dbl(a_old) | Xc);
        // ONE_INDEL: at an unclean cell the reference leaves the diagonal -- upwards (deletion) iff the vertical
        // delta is +1 (new VP); the accumulator bit of the cell it comes from: above in this column / left in the last
        s.U = bs_sel(Xc, s.VP, X > 1 ? (bs_dbl(s.U) | u_in) : bs_dbl(s.U));
        const uint32_t pa = bs_sel(s.VP, X > 1 ? (bs_dbl(s.A) | a_top_new) : bs_dbl(s.A), a_old);
        s.Z = bs_sel(Xc, pa, X > 1 ? (bs_dbl(s.Z) | z_in) : bs_dbl(s.Z));
    }
    if (__builtin_expect(s.cm <= p.kacc, 0))
        return bs_book<SUBS>(s, (s.A >> 31) == 0, j + joff, p, (s.U >> 31) != 0, (s.Z >> 31) != 0, jlim);
    return false;
}

// The window start moved back to a whole number of 16-column chunks in front of the read end (0 if the read is
// shorter than that): any earlier start is as exact as j0 (the costs of a window are >= the true ones and equal
// wherever they are <= k + 1 from column start + m + k + 1 on).
CAH_HD int bs_align_window(const int j0, const int n) {
    const int a = n - ((n - j0 + 15) & ~15);
    return a > 0 ? a : 0;
}

// May the scan stop after column j < n?  (see "early stop" in the header)
CAH_HD int bs_stop_gap(const BackScanParams& p) { return p.k + 1 + p.kacc + p.half_m; }
CAH_HD bool bs_may_stop(const BackScanBook& s, const int j, const int n, const int gap) {
    return s.jla >= 0 && j - s.jla >= gap && j < n;
}

// "does any lane of the wave still ..." (the host model has one lane)
CAH_HD bool bs_any(const bool pred) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(pred) != 0ull;
#else
    return pred;
#endif
}
CAH_HD int bs_popc(const uint32_t x) { return __builtin_popcount(x); }
CAH_HD int bs_popc(const uint64_t x) { return __builtin_popcountll(x); }
CAH_HD int bs_top_bit(const uint32_t x) { return 31 - __builtin_clz(x); }          // x != 0
CAH_HD int bs_top_bit(const uint64_t x) { return 63 - __builtin_clzll(x); }

// The last column as the classification sees it: rows X+1..m as a word (bit t = row X + 1 + t: vertical deltas vp / vn,
// unclean-diagonal bits a), rows 1..X as explicit costs cx[] with their unclean bits in ax.
template <class W, int X>
struct BsLastColumn {
    W vp, vn, a;
    int cx[X > 0 ? X : 1];
    unsigned ax;
};

// The walk down the last column (reference _align.pyx:536-572), in two passes (round 5; one pass over all rows with the
// whole bookkeeping per row was a third of the cost scan's time -- 33 rows x ~25 instructions, most of them compares and
// selects, for a column in which two or three rows are acceptable):
//   pass 1  every row: the running cost and ONE bit "acceptable" (cost <= thr_last(i), i >= min_overlap);
//   pass 2  every ACCEPTABLE row, lane by lane from the top (a wave makes as many rounds as its lane with the most
//           acceptable rows has): cost from two population counts, clean diagonal, and the maxima the rules ask for.
// Everything the rules ask of the acceptable rows is a maximum over them -- the order does not matter:
//   best_i          the largest acceptable row
//   (w_score, w_row) the clean acceptable row of the highest score i - 2 c, the largest row on ties -- one maximum
//                    over the key (score + 256) * 128 + row; its cost is (w_row - w_score) / 2
//   unclean_bound   the most an acceptable row with an unclean diagonal scores
//   c_max           the largest cost of an acceptable row: the origin clause holds for every acceptable row with
//                    errors iff k + 1 + c_max <= m / 2 (or c_max == 0)
struct BsRowStats {
    int best_i, w_key, unclean_bound, c_max;
};
CAH_HD void bs_row_update(BsRowStats& r, const bool on, const int i, const int c, const bool clean) {
    const int sc = i - 2 * c;
    r.best_i = (on && i > r.best_i) ? i : r.best_i;
    const int key_c = (on && clean) ? (sc + 256) * 128 + i : -1;
    r.w_key = key_c > r.w_key ? key_c : r.w_key;
    const int sc_u = (on && !clean) ? sc : -(1 << 20);
    r.unclean_bound = sc_u > r.unclean_bound ? sc_u : r.unclean_bound;
    const int c_a = on ? c : 0;
    r.c_max = c_a > r.c_max ? c_a : r.c_max;
}

template <bool TRACKED, class W, int X, class ThrLast>
CAH_HD BsRowStats bs_last_column_stats(const BsLastColumn<W, X>& col, const int n, const int j0, const BackScanParams& p,
                                       ThrLast thr_last, const int max_row) {
    BsRowStats r;
    r.best_i = 0; r.w_key = -1; r.unclean_bound = -(1 << 20); r.c_max = 0;
    const int rows = p.m < max_row ? p.m : max_row;
    // the explicit rows (at most two)
    for (int i = 1; i <= X && i <= rows; ++i) {
        const int c = col.cx[i - 1];
        const bool acc = i >= p.min_overlap && c <= thr_last(i);
        const bool clean = c == 0 || (n - i >= j0 && TRACKED && ((col.ax >> (i - 1)) & 1u) == 0);
        bs_row_update(r, acc, i, c, clean);
    }
    // pass 1 over the word's rows
    const int nw = rows - X;
    const int cbase = X > 0 ? col.cx[X - 1] : 0;
    int c = cbase;
    W acc = 0;
    for (int t = 0; t < nw; ++t) {
        const int i = X + 1 + t;
        c += (int)((col.vp >> t) & 1) - (int)((col.vn >> t) & 1);
        const bool ok = i >= p.min_overlap && c <= thr_last(i);
        acc |= ok ? ((W)1 << t) : (W)0;
    }
    // pass 2
    while (bs_any(acc != 0)) {
        const bool on = acc != 0;
        const int t = on ? bs_top_bit(acc) : 0;
        const W bit = (W)1 << t;
        acc &= ~bit;
        const W low = bit | (bit - 1);
        const int ci = cbase + bs_popc((W)(col.vp & low)) - bs_popc((W)(col.vn & low));
        const int i = X + 1 + t;
        const bool clean = ci == 0 || (n - i >= j0 && TRACKED && (col.a & bit) == 0);
        bs_row_update(r, on, i, ci, clean);
    }
    return r;
}

// After the last column (j == n) without EXACT_FULL, or after an early stop (stopped: the state is that of an
// inner column and nothing beyond jla can matter).  thr_last(i): error threshold of row i in the last
// column = thr[effective length of adapter[0:i]] (CahMatcher::thr_last).  j0 = first column of the window
// (the cost scan itself started there).  Outputs: o0/o1 = (row i, -) for EXACT_TAIL, (first DP column,
// last DP column * 2 + scan flag) for DP.
// max_row: rows above it are known not to be acceptable (a window that holds every acceptable candidate's alignment
// from column j0 on leaves rows > n - j0 + kacc no room) and are not looked at.
template <bool INDEL1, bool TRACKED, class W, int X, class ThrLast>
CAH_HD int bs_finish_rows(const BackScanBook& s, const BsLastColumn<W, X>& col, const int n, const int j0,
                          const BackScanParams& p, ThrLast thr_last, int& o0, int& o1, const bool stopped,
                          const int max_row = CAH_BS_ALL_ROWS, const int dp_lo_in = -1) {
    o0 = 0; o1 = 0;
    const int reach = p.m + p.k + 1;
    // the cell DP's first column is never before dp_lo: the scan's own window start, unless the caller knows an earlier
    // column that is as safe
    const int dp_lo = dp_lo_in >= 0 ? dp_lo_in : j0;
    // one insertion / one deletion (see the header); INDEL1 = false: the form does not keep the bits
    const bool indel1 = INDEL1 && s.cmin >= 1 && !s.eclean && !s.epred_unclean && s.je - p.m - 1 >= j0 &&
                        s.je + 1 - s.jfa <= p.half_m - p.kacc && !(s.edel && s.emore);
    const int indel1_score = p.m - 2 * s.cmin - (s.edel ? 1 : 0);
    if (stopped) {                   // jfa >= 0
        if (s.cmin >= 1 && s.eclean && s.je - p.m >= j0 && s.je - s.jfa <= p.half_m - p.kacc) {
            o0 = s.je; o1 = s.cmin;
            return BS_SUBS_FULL;
        }
        if (indel1) { o0 = s.je; o1 = s.cmin * 2 + (s.edel ? 1 : 0); return BS_INDEL1_FULL; }
        const int s0 = s.jfa - reach;
        o0 = s0 > dp_lo ? s0 : dp_lo;
        o1 = s.jla * 2;
        return BS_DP;
    }
    const BsRowStats r = bs_last_column_stats<TRACKED>(col, n, j0, p, thr_last, max_row);
    const int best_i = r.best_i, unclean_bound = r.unclean_bound, c_max = r.c_max;
    const int w_row = r.w_key >= 0 ? (r.w_key & 127) : 0;
    const int w_score = r.w_key >= 0 ? (r.w_key >> 7) - 256 : -(1 << 20);
    const int w_cost = (w_row - w_score) / 2;                   // (unused when w_row == 0)
    const bool clause_ok = c_max == 0 || p.k + 1 + c_max <= p.half_m;
    const int sc_max = w_score > unclean_bound ? w_score : unclean_bound;     // -(1 << 20) without an acceptable row
    const bool tail_may_win = sc_max > p.m - 2 * s.cmin;      // an acceptable row of the last column that could outscore m - 2 * cmin
    const bool tail_may_win1 = sc_max > indel1_score;         // ... the score of the one-indel alignment
    if (s.jfa < 0) {
        if (best_i == 0) return BS_NONE;
        if (w_row > 0 && unclean_bound < w_score && clause_ok) { o0 = w_row; o1 = w_cost; return BS_EXACT_TAIL; }
        const int s0 = n - reach;
        o0 = s0 > dp_lo ? s0 : dp_lo;
        o1 = n * 2 + 1;
        return BS_DP;
    }
    // substitutions only (see the header): the diagonal (0, je - m) .. (m, je) lies inside the window
    if (s.cmin >= 1 && s.eclean && s.je - p.m >= j0 && s.je - s.jfa <= p.half_m - p.kacc && !tail_may_win) {
        o0 = s.je; o1 = s.cmin;
        return BS_SUBS_FULL;
    }
    if (indel1 && !tail_may_win1) { o0 = s.je; o1 = s.cmin * 2 + (s.edel ? 1 : 0); return BS_INDEL1_FULL; }
    const int s0 = s.jfa - reach;
    o0 = s0 > dp_lo ? s0 : dp_lo;
    o1 = best_i == 0 ? s.jla * 2 : n * 2 + 1;
    return BS_DP;
}

// TRACKED: the scan ran with SUBS (the accumulator A means something)
template <bool TRACKED = true, class ThrLast>
CAH_HD int bs_finish(const BackScanState& s, const int n, const int j0, const BackScanParams& p,
                     ThrLast thr_last, int& o0, int& o1, const bool stopped = false, const int max_row = CAH_BS_ALL_ROWS,
                     const int dp_lo = -1) {
    const int pad = 64 - p.m;
    BsLastColumn<uint64_t, 0> col;
    col.vp = pad == 0 ? s.VP : (s.VP >> pad); col.vn = pad == 0 ? s.VN : (s.VN >> pad);
    col.a = pad == 0 ? s.A : (s.A >> pad);
    col.cx[0] = 0; col.ax = 0;
    return bs_finish_rows<false, TRACKED>(s, col, n, j0, p, thr_last, o0, o1, stopped, max_row, dp_lo);
}

template <int X, bool TRACKED = true, class ThrLast>
CAH_HD int bs32_finish(const BackScanState32<X>& s, const int n, const int j0, const BackScanParams& p,
                       ThrLast thr_last, int& o0, int& o1, const bool stopped = false, const int max_row = CAH_BS_ALL_ROWS,
                       const int dp_lo = -1) {
    const int pad = X > 0 ? 0 : 32 - p.m;
    BsLastColumn<uint32_t, X> col;
    col.vp = s.VP >> pad; col.vn = s.VN >> pad; col.a = s.A >> pad;
    for (int t = 0; t < (X > 0 ? X : 1); ++t) col.cx[t] = s.cx[t];
    col.ax = s.ax;
    return bs_finish_rows<TRACKED, TRACKED>(s, col, n, j0, p, thr_last, o0, o1, stopped, max_row, dp_lo);
}

