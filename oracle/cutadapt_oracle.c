/*
 * cutadapt_oracle.c -- CPU restatement of cutadapt's adapter-matching hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may link, load or call this file.  The product
 * (cutadapt_amd + libcutadapt_hip.so) never does.
 *
 * Parity status: PINNED.  This restatement is checked (tests/test_oracle_pinning.py,
 * tests/golden/) against
 *   - the reference's own Cython code compiled unmodified into oracle/_ref/
 *     (oracle/build_ref.py) on randomized fuzz covering all 16 flag combinations,
 *     wildcard modes, indel costs and error rates, and
 *   - committed golden vectors generated from that build (tests/golden/make_golden.py),
 *     which include the reference's known-answer tests and info-file coordinates.
 *
 * What is restated (reference paths relative to /root/reference/src/cutadapt/):
 *   orc_tables_*            _match_tables.py:4-66   (ACGT / IUPAC / upper-case LUTs)
 *   orc_aligner_new         _align.pyx:195-225, 250-277 (flags, n_counts, eff. length, encoding)
 *   orc_aligner_locate      _align.pyx:298-587     (banded semi-global DP, candidate rules)
 *   orc_comparer_*          _align.pyx:594-714     (Prefix/SuffixComparer, Hamming only)
 *   orc_kmer_finder_*       _kmer_finder.pyx:106-257 + _match_tables.py:69-98
 *
 * All arithmetic is 32-bit integer, except the two double products the reference
 * computes (rate*m and effective_length*rate), which are reproduced as doubles.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------
 * Character tables (_match_tables.py:4-66)
 * ---------------------------------------------------------------------------------- */
static unsigned char T_ACGT[256], T_IUPAC[256], T_UPPER[256];
static int tables_ready = 0;

static void set_both_cases(unsigned char *t, char c, unsigned char v) {
    t[(unsigned char)c] = v;
    if (c >= 'A' && c <= 'Z') t[(unsigned char)(c + 32)] = v;
}

static void orc_tables_init(void) {
    if (tables_ready) return;
    /* ACGT table: A=1 C=2 G=4 T=U=8, anything else 0x80 (_match_tables.py:12-16) */
    memset(T_ACGT, 0x80, 256);
    set_both_cases(T_ACGT, 'A', 1); set_both_cases(T_ACGT, 'C', 2);
    set_both_cases(T_ACGT, 'G', 4); set_both_cases(T_ACGT, 'T', 8);
    set_both_cases(T_ACGT, 'U', 8);
    /* IUPAC table: nibble sets; N carries 0x80 too; X and unknown are 0 (:38-60).
     * Note `A | C | G | T + 0x80` parses as A|C|G|(T+0x80) = 0x8F. */
    memset(T_IUPAC, 0, 256);
    const unsigned char A = 1, C = 2, G = 4, T = 8;
    set_both_cases(T_IUPAC, 'X', 0);
    set_both_cases(T_IUPAC, 'A', A); set_both_cases(T_IUPAC, 'C', C);
    set_both_cases(T_IUPAC, 'G', G); set_both_cases(T_IUPAC, 'T', T);
    set_both_cases(T_IUPAC, 'U', T);
    set_both_cases(T_IUPAC, 'R', A | G); set_both_cases(T_IUPAC, 'Y', C | T);
    set_both_cases(T_IUPAC, 'S', G | C); set_both_cases(T_IUPAC, 'W', A | T);
    set_both_cases(T_IUPAC, 'K', G | T); set_both_cases(T_IUPAC, 'M', A | C);
    set_both_cases(T_IUPAC, 'B', C | G | T); set_both_cases(T_IUPAC, 'D', A | G | T);
    set_both_cases(T_IUPAC, 'H', A | C | T); set_both_cases(T_IUPAC, 'V', A | C | G);
    set_both_cases(T_IUPAC, 'N', (unsigned char)(A | C | G | (T + 0x80)));
    /* upper table: bytes(range(256)).upper() only touches ASCII a-z (:64-66) */
    for (int i = 0; i < 256; i++) T_UPPER[i] = (unsigned char)i;
    for (int i = 'a'; i <= 'z'; i++) T_UPPER[i] = (unsigned char)(i - 32);
    tables_ready = 1;
}

static int all_ascii(const unsigned char *s, int64_t n) {
    for (int64_t i = 0; i < n; i++) if (s[i] & 0x80) return 0;
    return 1;
}

/* ------------------------------------------------------------------------------------
 * Aligner (_align.pyx:93-591)
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int m;
    unsigned char *ref;      /* encoded reference (_align.pyx:272-276) */
    int *n_counts;           /* n_counts[i] = #N/n in reference[:i] (:261-266) */
    double rate;
    int start_in_ref, start_in_query, stop_in_ref, stop_in_query;   /* :206-209 */
    int wildcard_ref, wildcard_query;
    int indel_cost, min_overlap, effective_length;
    /* scratch column (:172); one locate at a time per aligner, like the reference */
    int *cost, *score, *origin;
} orc_aligner;

/* returns NULL with *err set: 1 = only-N reference (:270-271), 2 = indel_cost < 1 (:217),
 * 3 = non-ASCII reference */
orc_aligner *orc_aligner_new(const char *reference, int m, double rate, int flags,
                             int wildcard_ref, int wildcard_query, int indel_cost,
                             int min_overlap, int *err) {
    orc_tables_init();
    *err = 0;
    if (!all_ascii((const unsigned char *)reference, m)) { *err = 3; return NULL; }
    orc_aligner *a = (orc_aligner *)calloc(1, sizeof(orc_aligner));
    a->m = m;
    a->rate = rate;
    a->start_in_ref = (flags & 1) != 0;
    a->start_in_query = (flags & 2) != 0;
    a->stop_in_ref = (flags & 4) != 0;
    a->stop_in_query = (flags & 8) != 0;
    a->wildcard_ref = wildcard_ref != 0;
    a->wildcard_query = wildcard_query != 0;
    a->min_overlap = min_overlap;
    a->indel_cost = indel_cost;
    a->ref = (unsigned char *)malloc((size_t)m + 1);
    a->n_counts = (int *)malloc(sizeof(int) * ((size_t)m + 1));
    a->cost = (int *)malloc(sizeof(int) * ((size_t)m + 1));
    a->score = (int *)malloc(sizeof(int) * ((size_t)m + 1));
    a->origin = (int *)malloc(sizeof(int) * ((size_t)m + 1));
    int nn = 0;
    for (int i = 0; i < m; i++) {
        a->n_counts[i] = nn;
        if (reference[i] == 'N' || reference[i] == 'n') nn++;
    }
    a->n_counts[m] = nn;
    a->effective_length = m;
    int bad = 0;
    if (a->wildcard_ref) {
        a->effective_length = m - nn;
        if (a->effective_length == 0) bad = 1;
        for (int i = 0; i < m; i++) a->ref[i] = T_IUPAC[(unsigned char)reference[i]];
    } else if (a->wildcard_query) {
        for (int i = 0; i < m; i++) a->ref[i] = T_ACGT[(unsigned char)reference[i]];
    } else {
        /* raw ASCII, NOT upper-cased (:276) */
        for (int i = 0; i < m; i++) a->ref[i] = (unsigned char)reference[i];
    }
    if (!bad && indel_cost < 1) bad = 2;
    if (bad) {
        *err = bad;
        free(a->ref); free(a->n_counts); free(a->cost); free(a->score); free(a->origin);
        free(a);
        return NULL;
    }
    return a;
}

void orc_aligner_free(orc_aligner *a) {
    if (!a) return;
    free(a->ref); free(a->n_counts); free(a->cost); free(a->score); free(a->origin);
    free(a);
}

int orc_aligner_effective_length(const orc_aligner *a) { return a->effective_length; }

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* Number of adapter characters that count towards the error budget for an alignment
 * covering reference[lo:hi) (hi - lo == length).  (_align.pyx:503-510, 543-553) */
static int budget_length(const orc_aligner *a, int length, int lo, int hi) {
    if (!a->wildcard_ref) return length;
    if (length < a->m) return length - (a->n_counts[hi] - a->n_counts[lo]);
    return a->effective_length;
}

/*
 * locate(): returns 1 and fills out6 = (ref_start, ref_stop, query_start, query_stop,
 * score, errors); 0 for "None"; -1 if the query holds a non-ASCII byte (ValueError in the
 * reference, _align.pyx:44-45).
 */
int orc_aligner_locate(orc_aligner *a, const unsigned char *query, int n, int out6[6]) {
    if (!all_ascii(query, n)) return -1;
    const int m = a->m;
    const unsigned char *s1 = a->ref;
    /* query encoding + comparison mode (:322-328, :442-445) */
    const unsigned char *qtab;
    int cmp_equal;
    if (a->wildcard_query) { qtab = T_IUPAC; cmp_equal = 0; }
    else if (a->wildcard_ref) { qtab = T_ACGT; cmp_equal = 0; }
    else { qtab = T_UPPER; cmp_equal = 1; }

    int *C = a->cost, *S = a->score, *O = a->origin;
    const int D = a->indel_cost;
    const int k = (int)(a->rate * m);                                   /* :343 */
    int max_n = n, min_n = 0;
    if (!a->start_in_query) max_n = imin(n, m + k);                     /* :348-350 */
    if (!a->stop_in_query) min_n = imax(0, n - m - k);                  /* :351-352 */

    /* first column (:364-383) */
    for (int i = 0; i <= m; i++) {
        if (!a->start_in_ref && !a->start_in_query) {
            S[i] = -2 * i; C[i] = imax(i, min_n) * D; O[i] = 0;
        } else if (a->start_in_ref && !a->start_in_query) {
            S[i] = 0; C[i] = min_n * D; O[i] = imin(0, min_n - i);
        } else if (!a->start_in_ref && a->start_in_query) {
            S[i] = -2 * i; C[i] = i * D; O[i] = imax(0, min_n - i);
        } else {
            S[i] = 0; C[i] = imin(i, min_n) * D; O[i] = min_n - i;
        }
    }
    const int SENTINEL = m + n + 1;                                     /* :394 */
    int b_refstop = m, b_qstop = n, b_cost = SENTINEL, b_origin = 0, b_score = 0;

    int last = imin(m, k + 1);                                          /* :399-401 */
    if (a->start_in_ref) last = m;
    int last_filled = 0;
    /* the reference's scalar locals; `origin` is deliberately kept across the loop and
     * re-used (stale) by the last-column scan (:565).  Initialised to 0 here where the
     * reference leaves them uninitialised; never read before being written in any path
     * that can reach a second candidate. */
    int cost = 0, score = 0, origin = 0;
    const int row0_origin_inc = a->start_in_query ? 1 : 0;             /* :413-415 */
    const int row0_cost_inc = a->start_in_query ? 0 : D;
    const int row0_score_inc = a->start_in_query ? 0 : -2;

    for (int j = min_n + 1; j <= max_n; j++) {                          /* :433 */
        int dc = C[0], ds = S[0], dor = O[0];
        O[0] += row0_origin_inc; C[0] += row0_cost_inc; S[0] += row0_score_inc;
        const unsigned char q = qtab[query[j - 1]];
        for (int i = 1; i <= last; i++) {                               /* :441 */
            int eq = cmp_equal ? (s1[i - 1] == q) : ((s1[i - 1] & q) != 0);
            if (eq) {                                                   /* :446-453 */
                cost = dc; origin = dor; score = ds + 1;
            } else {                                                    /* :455-476 */
                int c_diag = dc + 1, c_ins = C[i] + D, c_del = C[i - 1] + D;
                if (c_diag <= c_del && c_diag <= c_ins) {
                    cost = c_diag; origin = dor; score = ds - 1;
                } else if (c_del <= c_ins) {
                    cost = c_del; origin = O[i - 1]; score = S[i - 1] - 2;
                } else {
                    cost = c_ins; origin = O[i]; score = S[i] - 2;
                }
            }
            dc = C[i]; ds = S[i]; dor = O[i];                           /* :479 */
            C[i] = cost; O[i] = origin; S[i] = score;
        }
        last_filled = last;                                             /* :484 */
        while (last >= 0 && C[last] > k) last--;                        /* :490-491 */
        if (last < m) {
            last++;
        } else if (a->stop_in_query) {                                  /* :496-533 */
            cost = C[m]; score = S[m]; origin = O[m];
            int length = m + imin(origin, 0);
            int eff = budget_length(a, length, m - length, m);
            int ok = length >= a->min_overlap && (double)cost <= eff * a->rate;
            int best_len = m + imin(b_origin, 0);
            if (ok && (b_cost == SENTINEL
                       || (origin <= b_origin + m / 2 && score > b_score)
                       || (length > best_len && score > b_score))) {
                b_score = score; b_cost = cost; b_origin = origin;
                b_refstop = m; b_qstop = j;
                if (cost == 0 && origin >= 0) break;                    /* :531-533 */
            }
        }
    }

    if (max_n == n) {                                                   /* :536-572 */
        int first_i = a->stop_in_ref ? 0 : m;
        for (int i = last_filled; i >= first_i; i--) {
            int length = i + imin(O[i], 0);
            cost = C[i]; score = S[i];
            int lo = -imin(O[i], 0);
            int eff = budget_length(a, length, lo, i);
            int ok = length >= a->min_overlap && (double)cost <= eff * a->rate;
            int best_len = b_refstop + imin(b_origin, 0);
            /* NB: `origin` is the stale scalar, not O[i] (:565) */
            if (ok && (b_cost == SENTINEL
                       || (origin <= b_origin + m / 2 && score > b_score)
                       || (length > best_len && score > b_score))) {
                b_score = score; b_cost = cost; b_origin = O[i];
                b_refstop = i; b_qstop = n;
            }
        }
    }
    if (b_cost == SENTINEL) return 0;                                   /* :573-577 */
    if (b_origin >= 0) { out6[0] = 0; out6[2] = b_origin; }
    else { out6[0] = -b_origin; out6[2] = 0; }
    out6[1] = b_refstop; out6[3] = b_qstop; out6[4] = b_score; out6[5] = b_cost;
    return 1;
}

/* ------------------------------------------------------------------------------------
 * PrefixComparer / SuffixComparer (_align.pyx:594-714)
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int m, max_k, min_overlap, effective_length, wildcard_ref, wildcard_query, is_suffix;
    unsigned char *ref;   /* encoded; stored REVERSED for the suffix variant (:706) */
} orc_comparer;

/* err: 1 only-N, 4 rate outside [0,1], 5 min_overlap < 1, 3 non-ASCII */
orc_comparer *orc_comparer_new(const char *reference, int m, double rate, int wildcard_ref,
                               int wildcard_query, int min_overlap, int is_suffix, int *err) {
    orc_tables_init();
    *err = 0;
    if (!all_ascii((const unsigned char *)reference, m)) { *err = 3; return NULL; }
    int eff = m;
    if (wildcard_ref) {
        /* quirk: subtracts count('N') - count('n') (:628) */
        int nN = 0, nn = 0;
        for (int i = 0; i < m; i++) { nN += reference[i] == 'N'; nn += reference[i] == 'n'; }
        eff -= nN - nn;
        if (eff == 0) { *err = 1; return NULL; }
    }
    if (!(0.0 <= rate && rate <= 1.0)) { *err = 4; return NULL; }
    if (min_overlap < 1) { *err = 5; return NULL; }
    orc_comparer *c = (orc_comparer *)calloc(1, sizeof(orc_comparer));
    c->m = m; c->effective_length = eff; c->min_overlap = min_overlap;
    c->max_k = (int)(rate * eff);                                       /* :633 */
    c->wildcard_ref = wildcard_ref != 0; c->wildcard_query = wildcard_query != 0;
    c->is_suffix = is_suffix != 0;
    c->ref = (unsigned char *)malloc((size_t)m + 1);
    const unsigned char *tab = wildcard_ref ? T_IUPAC : (wildcard_query ? T_ACGT : T_UPPER);
    for (int i = 0; i < m; i++) {
        int src = is_suffix ? m - 1 - i : i;
        c->ref[i] = tab[(unsigned char)reference[src]];                 /* :637-642 */
    }
    return c;
}

void orc_comparer_free(orc_comparer *c) { if (c) { free(c->ref); free(c); } }
int orc_comparer_effective_length(const orc_comparer *c) { return c->effective_length; }

int orc_comparer_locate(const orc_comparer *c, const unsigned char *query, int n, int out6[6]) {
    if (!all_ascii(query, n)) return -1;
    const unsigned char *qtab;
    int cmp_equal = 0;
    if (c->wildcard_query) qtab = T_IUPAC;
    else if (c->wildcard_ref) qtab = T_ACGT;
    else { qtab = T_UPPER; cmp_equal = 1; }
    int length = imin(c->m, n), errors = 0;
    for (int i = 0; i < length; i++) {
        unsigned char q = qtab[query[c->is_suffix ? n - 1 - i : i]];
        int eq = cmp_equal ? (c->ref[i] == q) : ((c->ref[i] & q) != 0);
        errors += !eq;
    }
    if (errors > c->max_k || length < c->min_overlap) return 0;         /* :690-691 */
    int score = (length - errors) - errors;                             /* :692 */
    if (!c->is_suffix) {
        out6[0] = 0; out6[1] = length; out6[2] = 0; out6[3] = length;
    } else {                                                            /* :714 */
        out6[0] = c->m - length; out6[1] = c->m; out6[2] = n - length; out6[3] = n;
    }
    out6[4] = score; out6[5] = errors;
    return 1;
}

/* ------------------------------------------------------------------------------------
 * KmerFinder (_kmer_finder.pyx:106-257; match lists _match_tables.py:69-98)
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int64_t start, stop;          /* stop == 0 means "to the end" (:156-157) */
    uint64_t init_mask, found_mask;
    uint64_t mask[128];
} orc_kmer_entry;

typedef struct {
    int n_entries;
    orc_kmer_entry *entries;
} orc_kmer_finder;

/* does query character qc (ASCII code < 128, non-NUL) match needle character rc ?
 * (_match_tables.py:81-98 with all_matches_generator :69-78) */
static int kmer_char_matches(unsigned char rc, unsigned char qc, int ref_wc, int query_wc) {
    if (qc == 0 || qc >= 128) return 0;
    if (!ref_wc && !query_wc) return T_UPPER[rc] == T_UPPER[qc];
    if (ref_wc && !query_wc) return (T_IUPAC[rc] & T_ACGT[qc]) != 0;
    if (!ref_wc && query_wc) return (T_ACGT[rc] & T_IUPAC[qc]) != 0;
    return (T_IUPAC[rc] & T_IUPAC[qc]) != 0;
}

/*
 * sets are given flattened: set s has window (starts[s], stops[s]) (stops[s]==0 <=> None)
 * and kmers kmers[first[s] .. first[s+1]).  Returns NULL with *err = 6 when a k-mer is
 * longer than 64 characters (ValueError :138-140), 3 for non-ASCII.
 */
orc_kmer_finder *orc_kmer_finder_new(int n_sets, const int64_t *starts, const int64_t *stops,
                                     const int *first, const char *const *kmers,
                                     int ref_wc, int query_wc, int *err) {
    orc_tables_init();
    *err = 0;
    orc_kmer_finder *f = (orc_kmer_finder *)calloc(1, sizeof(orc_kmer_finder));
    int cap = 0;
    for (int s = 0; s < n_sets; s++) {
        int idx = first[s], end = first[s + 1];
        while (idx < end) {
            /* greedily pack k-mers into one 64-bit word (:131-149) */
            unsigned char word[64];
            memset(word, 0, 64);
            size_t off = 0;
            uint64_t init = 0, found = 0;
            while (idx < end) {
                size_t len = strlen(kmers[idx]);
                if (!all_ascii((const unsigned char *)kmers[idx], (int64_t)len)) { *err = 3; goto fail; }
                if (len > 64) { *err = 6; goto fail; }
                if (off + len > 64) break;
                init |= 1ULL << off;
                memcpy(word + off, kmers[idx], len);
                found |= 1ULL << (off + len - 1);
                off += len;
                idx++;
            }
            if (f->n_entries == cap) {
                cap = cap ? cap * 2 : 8;
                f->entries = (orc_kmer_entry *)realloc(f->entries, sizeof(orc_kmer_entry) * cap);
            }
            orc_kmer_entry *e = &f->entries[f->n_entries++];
            e->start = starts[s]; e->stop = stops[s];
            e->init_mask = init; e->found_mask = found;
            memset(e->mask, 0, sizeof(e->mask));
            for (size_t p = 0; p < off; p++) {                          /* :226-238 */
                if (word[p] == 0) continue;
                for (int qc = 0; qc < 128; qc++)
                    if (kmer_char_matches(word[p], (unsigned char)qc, ref_wc, query_wc))
                        e->mask[qc] |= 1ULL << p;
            }
        }
    }
    return f;
fail:
    free(f->entries); free(f);
    return NULL;
}

void orc_kmer_finder_free(orc_kmer_finder *f) { if (f) { free(f->entries); free(f); } }
int orc_kmer_finder_n_entries(const orc_kmer_finder *f) { return f->n_entries; }

/* 1 / 0, or -1 for a non-ASCII sequence (:182-183).  A positive stop beyond the end of the
 * sequence is clamped to the end (the reference reads past the buffer there, see
 * SURVEY.md section 7 "Reference UB"). */
int orc_kmers_present(const orc_kmer_finder *f, const unsigned char *seq, int64_t n) {
    if (!all_ascii(seq, n)) return -1;
    for (int e = 0; e < f->n_entries; e++) {
        const orc_kmer_entry *en = &f->entries[e];
        int64_t start = en->start, stop = en->stop;
        if (start < 0) { start += n; if (start < 0) start = 0; }        /* :190-195 */
        else if (start > n) continue;
        if (stop < 0) { stop += n; if (stop <= 0) continue; }           /* :196-201 */
        else if (stop == 0) stop = n;
        if (stop > n) stop = n;                                         /* clamp (UB in ref) */
        if (stop - start <= 0) continue;
        uint64_t R = 0;
        for (int64_t i = start; i < stop; i++) {                        /* :251-256 */
            R = ((R << 1) | en->init_mask) & en->mask[seq[i]];
            if (R & en->found_mask) return 1;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * Batch drivers over packed reads (seqs + int64 offsets[n+1]) -- used by tests to compare
 * whole batches and by bench.py's cpu_baseline "port" leg.
 * status[i]: 0 = None, 1 = match, 2 = invalid (non-ASCII) input
 * ---------------------------------------------------------------------------------- */
void orc_locate_batch(orc_aligner *a, const unsigned char *seqs, const int64_t *offsets,
                      int64_t n_reads, int32_t *out6, unsigned char *status) {
    for (int64_t r = 0; r < n_reads; r++) {
        int res[6] = {0, 0, 0, 0, 0, 0};
        int rc = orc_aligner_locate(a, seqs + offsets[r], (int)(offsets[r + 1] - offsets[r]), res);
        status[r] = rc == 1 ? 1 : (rc == 0 ? 0 : 2);
        for (int t = 0; t < 6; t++) out6[r * 6 + t] = rc == 1 ? res[t] : 0;
    }
}

void orc_comparer_batch(const orc_comparer *c, const unsigned char *seqs, const int64_t *offsets,
                        int64_t n_reads, int32_t *out6, unsigned char *status) {
    for (int64_t r = 0; r < n_reads; r++) {
        int res[6] = {0, 0, 0, 0, 0, 0};
        int rc = orc_comparer_locate(c, seqs + offsets[r], (int)(offsets[r + 1] - offsets[r]), res);
        status[r] = rc == 1 ? 1 : (rc == 0 ? 0 : 2);
        for (int t = 0; t < 6; t++) out6[r * 6 + t] = rc == 1 ? res[t] : 0;
    }
}

void orc_kmers_present_batch(const orc_kmer_finder *f, const unsigned char *seqs,
                             const int64_t *offsets, int64_t n_reads, unsigned char *present) {
    for (int64_t r = 0; r < n_reads; r++) {
        int rc = orc_kmers_present(f, seqs + offsets[r], offsets[r + 1] - offsets[r]);
        present[r] = rc == 1 ? 1 : (rc == 0 ? 0 : 2);
    }
}

/* filter -> locate, i.e. what BackAdapter/FrontAdapter.match_to does before wrapping the
 * tuple into a Match object (adapters.py:707-724, 815-832).  f may be NULL (MockKmerFinder). */
void orc_match_batch(orc_aligner *a, const orc_kmer_finder *f, const unsigned char *seqs,
                     const int64_t *offsets, int64_t n_reads, int32_t *out6,
                     unsigned char *status) {
    for (int64_t r = 0; r < n_reads; r++) {
        const unsigned char *q = seqs + offsets[r];
        int64_t n = offsets[r + 1] - offsets[r];
        int res[6] = {0, 0, 0, 0, 0, 0};
        int rc = f ? orc_kmers_present(f, q, n) : 1;
        if (rc == 1) rc = orc_aligner_locate(a, q, (int)n, res);
        status[r] = rc == 1 ? 1 : (rc == 0 ? 0 : 2);
        for (int t = 0; t < 6; t++) out6[r * 6 + t] = rc == 1 ? res[t] : 0;
    }
}
