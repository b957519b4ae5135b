/*
 * cutadapt_hip.h -- C ABI of libcutadapt_hip.so: cutadapt's error-tolerant adapter
 * matching (Aligner.locate + k-mer prefilter + match_to) on AMD Instinct MI355X (gfx950).
 *
 * The reference has no FFI/plugin seam of its own; the seam this library replaces is the
 * Cython extension-module API that reference src/cutadapt/adapters.py calls:
 *
 *   cutadapt._align.Aligner(...).locate(query)             _align.pyx:195-204, 298-587
 *   cutadapt._align.PrefixComparer / SuffixComparer        _align.pyx:594-714
 *   cutadapt._kmer_finder.KmerFinder(...).kmers_present()  _kmer_finder.pyx:106-213
 *   *Adapter.match_to() = kmers_present -> locate          adapters.py:707-724, 815-832
 *   MultipleAdapters.match_to() = argmax over adapters     adapters.py:1265-1286
 *
 * Conventions
 *   - plain C types only; every function returns an int status (CAH_OK == 0) and never
 *     throws; cah_last_error() returns the message of the calling thread's last failure.
 *   - a *plan* is immutable after creation and may be shared by any number of streams /
 *     threads; all per-launch scratch is passed in by the caller (workspace) so that calls
 *     on different streams never share state (the reference Aligner is NOT re-entrant,
 *     _align.pyx:172/317 -- this ABI is).
 *   - reads are *packed*: one uint8 buffer `seqs` with all sequences back to back and
 *     `offsets` (int64, n_reads + 1 entries); read r is seqs[offsets[r] : offsets[r+1]).
 *     If `lens` is non-NULL, read r is seqs[offsets[r] : offsets[r] + lens[r]) instead and
 *     `offsets` needs only n_reads entries (used for the second stage of linked adapters,
 *     adapters.py:1222-1224, where the 3' adapter is searched in a suffix of the read).
 *   - results: `out6` is int32[n_reads][6] = (ref_start, ref_stop, query_start, query_stop,
 *     score, errors), exactly the tuple Aligner.locate returns (_align.pyx:587);
 *     `status` is uint8[n_reads]: CAH_NONE (locate() -> None), CAH_MATCH, or CAH_INVALID
 *     (a byte >= 0x80 was met: the reference raises ValueError, _align.pyx:44-45).
 *     out6 rows of non-matching reads are zero.
 *   - pointers named d_* are DEVICE pointers (HBM) on the current HIP device and the call
 *     is asynchronous on `stream` (a hipStream_t passed as void*, NULL = default stream).
 *     The *_host variants take host pointers, stage through HBM and synchronise.
 */
#ifndef CUTADAPT_HIP_H
#define CUTADAPT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAH_ABI_VERSION 5   /* 2: CAH_PROF_N = 5, plan workspaces (cah_plan_workspace_bytes), no adapter length limit; 3: cah_plan_multi_kind; 4: cah_build_id, cah_last_multi_path, CAH_EINTERNAL; 5: cah_match_batch_frames, cah_set_deferred_errors, CAH_STATUS_INTERNAL, cah_mark_reads_device, cah_revcomp_in_place_device, cah_fastq_format_suffix_device, cah_info_format_device */

/* status codes */
#define CAH_OK 0
#define CAH_EINVAL 1        /* bad argument (ValueError in the reference) */
#define CAH_ETYPE 2         /* wrong type (TypeError in the reference) */
#define CAH_EHIP 3          /* HIP runtime failure, message has the hipError string */
#define CAH_ENOMEM 4
#define CAH_EUNSUPPORTED 5  /* outside this build's limits (read > 1e6 characters, device is not gfx950, ...) */
#define CAH_EINTERNAL 6     /* the library caught itself breaking one of its own invariants (e.g. the page pool of the
                               streaming multi-adapter path ran out although the gate should have prevented it): the
                               results of the call are not to be used; never returned for bad input */

/* per-read result status */
#define CAH_NONE 0
#define CAH_MATCH 1
#define CAH_INVALID 2

/* matcher kinds */
#define CAH_KIND_ALIGNER 0   /* Aligner: banded semi-global DP       (_align.pyx:93)  */
#define CAH_KIND_PREFIX 1    /* PrefixComparer: Hamming, 5' anchored (_align.pyx:594) */
#define CAH_KIND_SUFFIX 2    /* SuffixComparer: Hamming, 3' anchored (_align.pyx:696) */
#define CAH_KIND_KMER_ONLY 3 /* a bare KmerFinder, no aligner                          */

/* limits of this build */
/* adapters of any length are accepted, as in the reference (_align.pyx:250-257): up to 64 characters the DP
 * column lives in VGPRs and the match relation in 64-bit bitsets, longer ones run with the column in HBM
 * scratch (size it with cah_plan_workspace_bytes) */
#define CAH_MAX_READ_LEN 1000000 /* origin is carried in 20 bits of the packed DP cell */
#define CAH_MAX_INDEL_COST 10000000

/* One k-mer search set = (start, stop, [kmers]) of KmerFinder's positions_and_kmers
 * (_kmer_finder.pyx:106, :121); stop == 0 encodes Python's None (:156-157). */
typedef struct cah_kmer_set {
    int64_t start;
    int64_t stop;
    const char *const *kmers; /* n_kmers NUL-terminated ASCII strings, each <= 64 chars */
    int32_t n_kmers;
} cah_kmer_set;

/* One matcher = what one *Adapter object holds: an aligner (or comparer) plus an optional
 * k-mer prefilter.  Mirrors the constructor arguments of Aligner (_align.pyx:195-204),
 * PrefixComparer (:615-622) and KmerFinder (_kmer_finder.pyx:106). */
typedef struct cah_adapter_desc {
    const char *sequence;  /* the aligner's `reference` (adapter) string, not NUL-terminated */
    int32_t length;        /* m */
    double max_error_rate;
    int32_t flags;         /* EndSkip bits: 1 REFERENCE_START, 2 QUERY_START,
                              4 REFERENCE_END, 8 QUERY_STOP (align.py:24-34) */
    int32_t wildcard_ref;
    int32_t wildcard_query;
    int32_t indel_cost;    /* 1, or 100000 for "no indels" (adapters.py:605) */
    int32_t min_overlap;
    int32_t kind;          /* CAH_KIND_* */
    /* prefilter; n_kmer_sets < 0 means "no prefilter" (MockKmerFinder, adapters.py:29-31) */
    const cah_kmer_set *kmer_sets;
    int32_t n_kmer_sets;
    int32_t kmer_ref_wildcards;   /* KmerFinder(ref_wildcards=...) */
    int32_t kmer_query_wildcards; /* KmerFinder(query_wildcards=...) */
} cah_adapter_desc;

typedef struct cah_plan cah_plan;

/* ---- library / device ---------------------------------------------------------------- */
int cah_abi_version(void);
/* sha256 (hex) over the sources the library was built from (cutadapt_amd/csrc/ + this header), embedded at build time by
 * cutadapt_amd/build.py: ties a binary -- not just a source tree -- to a profile or a bench line */
const char *cah_build_id(void);
/* copies the calling thread's last error message (NUL-terminated) into buf */
void cah_last_error(char *buf, size_t buflen);
int cah_device_count(int *count);
int cah_set_device(int device);
/* name of the HIP device and its gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
int cah_device_info(int device, char *name, size_t name_len, char *arch, size_t arch_len,
                    int *compute_units, int64_t *hbm_bytes);

/* ---- plans ---------------------------------------------------------------------------- */
/* Validates like the reference constructors do (CAH_EINVAL for: only-N reference with
 * wildcard_ref, indel_cost < 1, comparer rate outside [0,1], comparer min_overlap < 1,
 * non-ASCII characters, k-mer longer than 64) and uploads the immutable tables to the
 * current device.  Replaces Aligner.__cinit__/_set_reference (_align.pyx:195-277),
 * PrefixComparer.__init__ (:615-642) and KmerFinder.__cinit__ (_kmer_finder.pyx:106-165). */
int cah_plan_create(const cah_adapter_desc *adapters, int32_t n_adapters, cah_plan **out);
void cah_plan_destroy(cah_plan *plan);
int cah_plan_n_adapters(const cah_plan *plan);
/* Aligner.effective_length / PrefixComparer.effective_length (_align.pyx:188, :611) */
int cah_plan_effective_length(const cah_plan *plan, int32_t adapter, int32_t *out);
/* number of packed 64-bit shift-and words (KmerFinder.number_of_searches) */
int cah_plan_n_kmer_entries(const cah_plan *plan, int32_t adapter, int32_t *out);
/* which prefilter kernel serves this adapter: none (MockKmerFinder), the general shift-and kernel, or
 * the lean kernel (all search sets are whole-read / last-L / first-L windows, k-mers <= 32) */
#define CAH_PREFILTER_NONE 0
#define CAH_PREFILTER_GENERAL 1
#define CAH_PREFILTER_LEAN 2
int cah_plan_prefilter_kind(const cah_plan *plan, int32_t adapter, int32_t *out);
/* how a batch of equally long reads of `read_len` characters is matched against ALL adapters of the plan
 * (MultipleAdapters.match_to, adapters.py:1265-1286): one adapter after the other, one fused prefilter pass over
 * (read, adapter) pairs, or its streaming form (k_multi_stream: LDS-staged reads, pairs in pages by window class) */
#define CAH_MULTI_SEQUENTIAL 0
#define CAH_MULTI_FUSED 1
#define CAH_MULTI_STREAM 2
int cah_plan_multi_kind(const cah_plan *plan, int32_t read_len, int32_t *out);
/* the form the calling thread's LAST cah_match_batch / cah_match_batch_uniform call actually took (one of the three
 * above; -1 before the first call): cah_plan_multi_kind says what a plan can do for a read length, this says what ran
 * -- a ragged batch or a workspace smaller than cah_plan_workspace_bytes falls back without an error */
int cah_last_multi_path(void);
/* Introspection for tests: copies the host-side matcher table of an adapter (struct CahMatcher of
 * cutadapt_amd/csrc/cah_device.h: DP constants, row bitsets, cost-scan tables) into buf; *need receives
 * its size.  The layout is internal to the library version. */
int cah_plan_debug_matcher(const cah_plan *plan, int32_t adapter, void *buf, size_t buflen,
                           size_t *need);
/* ... and the lean prefilter's word tables (struct CahLeanFilter) */
int cah_plan_debug_lean(const cah_plan *plan, int32_t adapter, void *buf, size_t buflen, size_t *need);

/* ---- batch entry points (device pointers, asynchronous) --------------------------------- */
/* Aligner.locate / PrefixComparer.locate / SuffixComparer.locate over a batch.
 * d_workspace: at least cah_workspace_bytes(n_reads) bytes of device scratch. */
int cah_locate_batch(const cah_plan *plan, int32_t adapter, const uint8_t *d_seqs,
                     const int64_t *d_offsets, const int32_t *d_lens, int64_t n_reads,
                     int32_t *d_out6, uint8_t *d_status, void *d_workspace,
                     size_t workspace_bytes, void *stream);

/* KmerFinder.kmers_present over a batch: d_present[r] = 0 / 1 (or CAH_INVALID). */
int cah_kmers_present_batch(const cah_plan *plan, int32_t adapter, const uint8_t *d_seqs,
                            const int64_t *d_offsets, const int32_t *d_lens, int64_t n_reads,
                            uint8_t *d_present, void *stream);

/* Fused hot path: for every adapter of the plan (in order) prefilter -> wave-compact the
 * survivors into a work queue -> DP on dense waves -> keep the best match per read with
 * MultipleAdapters' rule (higher score, then fewer errors, then first adapter;
 * adapters.py:1278-1285).  d_best_adapter (int32[n_reads], may be NULL) receives the index
 * of the winning adapter or -1. */
int cah_match_batch(const cah_plan *plan, const uint8_t *d_seqs, const int64_t *d_offsets,
                    const int32_t *d_lens, int64_t n_reads, int32_t *d_out6,
                    int32_t *d_best_adapter, uint8_t *d_status, void *d_workspace,
                    size_t workspace_bytes, void *stream);

/* The same for a batch of EQUALLY LONG reads stored back to back without an offsets array: read r is
 * d_seqs[r * read_len, (r + 1) * read_len) -- what a sequencer emits, what `cutadapt` sees for every untrimmed Illumina
 * run (the reference has no batch call to compare with; per read it is Adapter.match_to, adapters.py:707-724, :815-832).
 * Same results as cah_match_batch on the equivalent offsets; cheaper: nothing looks at 8 bytes of offset per read,
 * the prefilter is one launch, the scan / DP kernels compute a survivor's address instead of fetching it. */
int cah_match_batch_uniform(const cah_plan *plan, const uint8_t *d_seqs, int32_t read_len, int64_t n_reads,
                            int32_t *d_out6, int32_t *d_best_adapter, uint8_t *d_status, void *d_workspace,
                            size_t workspace_bytes, void *stream);
/* ... and for views that are SUFFIXES of the reads of such a batch -- the second stage of LinkedAdapter.match_to
 * (adapters.py:1222-1224: the 3' adapter is searched in read[front_match.rstop:]): view r = d_seqs[d_starts[r],
 * d_starts[r] + d_lens[r]) with d_starts[r] + d_lens[r] == (r + 1) * parent_read_len.  Same results as
 * cah_match_batch(plan, d_seqs, d_starts, d_lens, ...) (coordinates relative to the views); the prefilter streams
 * the parent's reads instead of fetching ragged views. */
int cah_match_batch_suffix_views(const cah_plan *plan, const uint8_t *d_seqs, const int64_t *d_starts,
                                 const int32_t *d_lens, int32_t parent_read_len, int64_t n_reads,
                                 int32_t *d_out6, int32_t *d_best_adapter, uint8_t *d_status,
                                 void *d_workspace, size_t workspace_bytes, void *stream);
/* ... and for views that lie ANYWHERE inside the reads of such a batch: r * parent_read_len <= d_starts[r],
 * d_starts[r] + d_lens[r] <= (r + 1) * parent_read_len -- the reads of a sequencer's batch after a modifier in front of
 * the adapter search has cut them (reference modifiers.py: QualityTrimmer, UnconditionalCutter, Shortener run before
 * AdapterCutter; per read the adapter then sees a shorter str, adapters.py:707-724, :815-832).  Same results as
 * cah_match_batch(plan, d_seqs, d_starts, d_lens, ...); the prefilter streams the parent's reads end-aligned. */
int cah_match_batch_views(const cah_plan *plan, const uint8_t *d_seqs, const int64_t *d_starts,
                          const int32_t *d_lens, int32_t parent_read_len, int64_t n_reads,
                          int32_t *d_out6, int32_t *d_best_adapter, uint8_t *d_status,
                          void *d_workspace, size_t workspace_bytes, void *stream);
/* Views ANYWHERE in d_seqs, none longer than frame_len characters: a packed batch with its offsets (d_starts[r] =
 * offsets[r], d_lens[r] = its length), the reads of a raw FASTQ chunk in place -- MultipleAdapters.match_to on reads of any
 * length (reference adapters.py:1265-1286; behind the quality trimmers a pipeline's reads are ragged, cli.py:938-954).
 * Same results as cah_match_batch(plan, d_seqs, d_starts, d_lens, ...).  What the frame length buys: a plan of several
 * adapters that has the streaming form takes the views end-aligned in frames of frame_len characters instead of the
 * per-lane kernels.  The caller guarantees d_lens[r] <= frame_len. */
int cah_match_batch_frames(const cah_plan *plan, const uint8_t *d_seqs, const int64_t *d_starts, const int32_t *d_lens,
                           int32_t frame_len, int64_t n_reads, int32_t *d_out6, int32_t *d_best_adapter,
                           uint8_t *d_status, void *d_workspace, size_t workspace_bytes, void *stream);

/* Deferred error check of the streaming multi-adapter path, per calling thread (returns the previous setting; default 0).
 * 0: every call synchronises once and answers CAH_EINTERNAL if a kernel flagged a broken invariant.  1: a call whose page
 * pool holds the batch's worst case returns without waiting; should a kernel have flagged something, EVERY status byte of
 * that batch is CAH_STATUS_INTERNAL (the rows are void) -- the caller must look.  For callers that issue many small batches
 * on several streams. */
#define CAH_STATUS_INTERNAL 255
int cah_set_deferred_errors(int on);

/* The views the second stage of a linked adapter searches (adapters.py:1222-1224): d_starts[r] = start of read r +
 * (front match ? its query_stop : 0), d_view_lens[r] = what is left of the read.  Reads as in cah_match_batch, or
 * read_len > 0: equally long reads back to back from byte 0 (d_offsets may be NULL). */
int cah_linked_views(const int32_t *d_out6_front, const uint8_t *d_status_front, const int64_t *d_offsets,
                     const int32_t *d_lens, int32_t read_len, int64_t n_reads, int64_t *d_starts,
                     int32_t *d_view_lens, void *stream);
/* LinkedAdapter.match_to (adapters.py:1215-1227) over a batch of equally long reads in ONE call: the 5' plan on the
 * reads, the 3' plan on read[front_match.rstop:] (the whole read where the 5' adapter was not found).  Writes both
 * stages' results as cah_match_batch does (the 3' stage's coordinates are relative to its view) and the views
 * (d_starts int64[n], d_view_lens int32[n]); `required` / `optional` is the caller's verdict over the two status
 * arrays.  d_best_* may be NULL.  When the 5' adapter is anchored and tolerates no error (e.g. ^NNNNNNNNACGTACGT) and
 * the 3' adapter has a streaming prefilter, one pass over the batch does the 5' comparison, the view arithmetic and
 * the 3' prefilter; otherwise the two stages run one after the other.  Workspace: cah_plan_workspace_bytes of the
 * larger of the two plans. */
int cah_linked_match_batch_uniform(const cah_plan *front_plan, const cah_plan *back_plan,
                                   const uint8_t *d_seqs, int32_t read_len, int64_t n_reads,
                                   int32_t *d_out6_front, int32_t *d_best_front, uint8_t *d_status_front,
                                   int32_t *d_out6_back, int32_t *d_best_back, uint8_t *d_status_back,
                                   int64_t *d_starts, int32_t *d_view_lens, void *d_workspace,
                                   size_t workspace_bytes, void *stream);

/* bytes of device scratch the calls above need for n_reads reads (17.7 bytes per read + 6 KiB: counters, the
 * prefilter's survivor queue with keys, the cell-DP work list with its column windows, the cost scan's
 * straggler list) */
size_t cah_workspace_bytes(int64_t n_reads);
/* ... for THIS plan: plans of many equally shaped 3' adapters (e.g. `-a file:` with 96 adapters) have a fused
 * multi-adapter path -- one prefilter pass for all adapters instead of one per adapter -- that needs more
 * scratch (pair queue + per-read best keys).  cah_match_batch takes that path when workspace_bytes is at
 * least this value and otherwise matches one adapter after the other (same results). */
size_t cah_plan_workspace_bytes(const cah_plan *plan, int64_t n_reads);

/* d_bad[0] (int32) is set to the number of reads holding a byte >= 0x80 (the reference
 * rejects such strings up front, _align.pyx:44-45 / _kmer_finder.pyx:182-183). */
int cah_validate_ascii_batch(const uint8_t *d_seqs, const int64_t *d_offsets,
                             const int32_t *d_lens, int64_t n_reads, int32_t *d_bad,
                             void *stream);

/* The per-read API of the reference -- Adapter.match_to(read) (adapters.py:684-1089) / Aligner.locate(read)
 * (_align.pyx:298) -- for a plan of ONE adapter and one read, without an offsets array.  Same results as the
 * *_batch_host calls with n_reads = 1; the library queues no copies for it (the kernels read the read from, and
 * write the tuple to, mapped pinned memory) and prepares its scratch for the next call while the caller is away. */
int cah_match_one_host(const cah_plan *plan, const uint8_t *seq, int64_t n, int32_t *out6, uint8_t *status);
int cah_locate_one_host(const cah_plan *plan, const uint8_t *seq, int64_t n, int32_t *out6, uint8_t *status);

/* Aligner.enable_debug() (_align.pyx:279-296): locate() of ONE read with the dynamic-programming matrices the
 * reference collects in DPMatrix objects (:58-92, filled at :385-390 and :485-489).  cost_matrix / score_matrix:
 * (length + 1) x (n + 1) int32, row-major (row = adapter position, column = read position); the call only writes
 * the cells the banded algorithm computes, so the caller pre-fills them with its "not computed" marker.
 * Host pointers, synchronous, any adapter length; runs the statement-by-statement kernel of long.hip. */
int cah_locate_debug_host(const cah_adapter_desc *adapter, const uint8_t *seq, int64_t n, int32_t *out6,
                          uint8_t *status, int32_t *cost_matrix, int32_t *score_matrix);

/* ---- host-pointer conveniences (synchronous; stage through HBM internally) -------------- */
int cah_locate_batch_host(const cah_plan *plan, int32_t adapter, const uint8_t *seqs,
                          const int64_t *offsets, int64_t n_reads, int32_t *out6,
                          uint8_t *status);
int cah_kmers_present_batch_host(const cah_plan *plan, int32_t adapter, const uint8_t *seqs,
                                 const int64_t *offsets, int64_t n_reads, uint8_t *present);
int cah_match_batch_host(const cah_plan *plan, const uint8_t *seqs, const int64_t *offsets,
                         int64_t n_reads, int32_t *out6, int32_t *best_adapter,
                         uint8_t *status);

/* ---- measurement ------------------------------------------------------------------------ */
/* When enabled, every batch call brackets its kernels with HIP events on the launch stream;
 * cah_profile_read() synchronises those events and returns the accumulated kernel time and
 * launch count per kernel family since the last cah_profile_reset().  Not thread-safe; meant
 * for bench.py (roofline.achieved needs the dominant kernel's own duration). */
#define CAH_PROF_FILTER 0
#define CAH_PROF_DP 1
#define CAH_PROF_COMPARER 2
#define CAH_PROF_SCAN 3   /* k_back_scan: bit-parallel cost scan in front of the cell DP */
#define CAH_PROF_MERGE 4  /* k_multi_decode: best keys -> tuples (fused multi-adapter path) */
#define CAH_PROF_N 5
int cah_profile_enable(int enable);
int cah_profile_reset(void);
int cah_profile_read(double ms[CAH_PROF_N], int64_t launches[CAH_PROF_N],
                     int64_t units[CAH_PROF_N]);

/* ---- synthetic workloads (benchmark utility) -------------------------------------------- */
/* Fills d_seqs (n_reads * read_len bytes) and d_offsets (n_reads + 1) with the synthetic
 * read model of SURVEY.md section 8(d); read r is a pure function of (seed, first_index + r).
 * Probabilities are fixed point: p_adapter and p_edit as p * 2^32, p_n as p * 2^16.
 * The CPU twin is oracle/synth_reads.c. */
int cah_synth_reads(uint64_t seed, int64_t first_index, int64_t n_reads, int32_t read_len,
                    uint32_t p_adapter_u32, uint32_t p_edit_u32, uint32_t p_n_u16,
                    const char *adapters, const int32_t *adapter_off, int32_t n_adapters,
                    uint8_t *d_seqs, int64_t *d_offsets, void *stream);

/* ---- "next" rows of SURVEY.md section 8(f): the data formats either side of the path ------- */
/* Host-side FASTQ chunk handling (no GPU needed).  Replaces, for the batch pipeline, what the
 * reference does through the third-party dnaio package: record-aligned chunks
 * (runners.py:116-126), per-read SequenceRecord parsing (files.py:108-114), string slicing for
 * trimming (adapters.py:453-454, :486-487) and FASTQ formatting of the output.
 *
 * cah_fastq_scan: index complete 4-line records of a chunk.  rec[i*6..i*6+5] = (name_beg,
 *   name_end, seq_beg, seq_end, qual_beg, qual_end) byte offsets into buf ('@', "\n" and "\r\n"
 *   excluded).  *consumed = bytes covered by complete records; the caller prepends the rest to the
 *   next chunk.  is_final: the chunk ends the file (a last line without newline is accepted,
 *   leftovers are a format error).  CAH_EINVAL + message on malformed records. */
int cah_fastq_scan(const uint8_t *buf, int64_t len, int is_final, int64_t max_records,
                   int64_t *rec, int64_t *n_records, int64_t *consumed);
/* sequence lines packed back to back + offsets[n+1]: the layout of cah_match_batch */
int cah_pack_sequences(const uint8_t *buf, const int64_t *rec, int64_t n_records,
                       uint8_t *out_seqs, int64_t *out_offsets);
/* "@name\nSEQ[keep_beg:keep_end]\n+\nQUAL[keep_beg:keep_end]\n" for every record with
 * keep[i] != 0 (keep may be NULL = all). */
int cah_fastq_write_trimmed(const uint8_t *buf, const int64_t *rec, int64_t n_records,
                            const int32_t *keep_beg, const int32_t *keep_end, const uint8_t *keep,
                            uint8_t *out, int64_t out_cap, int64_t *out_len);

/* FASTA twin of cah_fastq_scan (">name" + one or more sequence lines).  Same rec columns;
 * seq_beg..seq_end spans the raw sequence lines (cah_pack_sequences strips the line breaks) and
 * qual_beg = qual_end = -1 marks a record without qualities. */
int cah_fasta_scan(const uint8_t *buf, int64_t len, int is_final, int64_t max_records,
                   int64_t *rec, int64_t *n_records, int64_t *consumed);
/* Largest offset <= len at which a record starts (0 if only the first record start was seen): lets
 * a reader thread cut the byte stream without parsing it; the full scan runs in a worker. */
int cah_record_boundary(const uint8_t *buf, int64_t len, int is_fasta, int64_t *cut);
/* Output formatting for every AdapterCutter action (modifiers.py:170-198, :236-251): sequence from
 * the packed batch (seqs/offsets), names and qualities from the raw chunk; interval [beg,end) per
 * read.  SLICE: sequence+qualities sliced (trim, retain, crop; whole read for action None);
 * MASK: bases outside the interval become 'N'; LOWERCASE: outside lower-, inside upper-case. */
#define CAH_WRITE_SLICE 0
#define CAH_WRITE_MASK 1
#define CAH_WRITE_LOWERCASE 2
int cah_records_write(const uint8_t *buf, const int64_t *rec, int64_t n_records,
                      const uint8_t *seqs, const int64_t *offsets, const int32_t *beg,
                      const int32_t *end, const uint8_t *keep, int mode, uint8_t *out,
                      int64_t out_cap, int64_t *out_len);
/* --info-file rows (steps.py:232-253, adapters.py:395-417, linked :1157-1171).  rows[k*7..] =
 * (read, errors, rstart, rstop, wbeg, wend, name_idx) sorted by read then match order; rstart and
 * rstop are relative to original[wbeg:wend], the read as it was when the match was made. */
int cah_info_write(const uint8_t *buf, const int64_t *rec, int64_t n_records, const uint8_t *seqs,
                   const int64_t *offsets, const int64_t *rows, int64_t n_rows, const char *names,
                   const int64_t *name_off, int64_t n_names, uint8_t *out, int64_t out_cap,
                   int64_t *out_len);

/* ... with the reverse-complement column (steps.py:224, :243 RC_MAP): is_rc uint8[n_records], "1"/"0" per row
 * (NULL: the empty column cah_info_write writes, i.e. --revcomp was not given), and with final_beg/final_end
 * int32[n_records] (both or neither): what the other modifiers left of a read, shown on the "-1" line of a read
 * without a match (steps.py:248-251 prints the read as it is written; match rows show the read as it came in). */
int cah_info_write_rc(const uint8_t *buf, const int64_t *rec, int64_t n_records, const uint8_t *seqs,
                      const int64_t *offsets, const int64_t *rows, int64_t n_rows, const char *names,
                      const int64_t *name_off, int64_t n_names, const uint8_t *is_rc,
                      const int32_t *final_beg, const int32_t *final_end, uint8_t *out,
                      int64_t out_cap, int64_t *out_len);
/* The chunk as ReverseComplementer leaves it (modifiers.py:264-308; SequenceRecord.reverse_complement() is dnaio's,
 * restated in csrc/revcomp.h): one normalised record per input record written to out + its record table out_rec
 * int64[n_records*6]; records with is_rc[i] != 0 carry the reverse-complemented sequence, the reversed qualities
 * and name + suffix (" rc", modifiers.py:296; suffix_len 0 = none).  seqs/offsets as cah_pack_sequences made them. */
int cah_chunk_revcomp(const uint8_t *buf, const int64_t *rec, int64_t n_records, const uint8_t *seqs,
                      const int64_t *offsets, const uint8_t *is_rc, const char *suffix, int64_t suffix_len,
                      uint8_t *out, int64_t out_cap, int64_t *out_rec, int64_t *out_len);
/* One output chunk of PairedReverseComplementer (modifiers.py:311-405: the pair with R1 and R2 swapped): record i is
 * record i of chunk B where swap[i] != 0 (name + suffix) and of chunk A otherwise; out / out_rec as cah_chunk_revcomp. */
int cah_chunk_select(const uint8_t *buf_a, const int64_t *rec_a, const uint8_t *seqs_a, const int64_t *offsets_a,
                     const uint8_t *buf_b, const int64_t *rec_b, const uint8_t *seqs_b, const int64_t *offsets_b,
                     int64_t n_records, const uint8_t *swap, const char *suffix, int64_t suffix_len,
                     uint8_t *out, int64_t out_cap, int64_t *out_rec, int64_t *out_len);
/* Whole 4-line FASTQ records in buf (at most max_records) and the bytes they cover: lines are counted, not parsed --
 * what cutting two FASTQ streams into chunks with equal record counts needs (dnaio.read_paired_chunks,
 * runners.py:104-113).  is_final: a last record without a final line feed counts. */
int cah_fastq_span(const uint8_t *buf, int64_t len, int is_final, int64_t max_records, int64_t *n_records,
                   int64_t *consumed);

/* ---- the same formats ON THE DEVICE (fastq_gpu.hip): the raw FASTQ chunk goes to HBM as it is, records are
 * indexed and the trimmed records formatted there, so that the host does nothing per read.  All pointers are
 * device pointers, all calls asynchronous on `stream`.  Scratch: cah_fastq_device_scratch_bytes(chunk, records),
 * shared by the three steps of one chunk (step 2 and 3 read what the earlier steps left there).
 *   1. cah_fastq_count_lines_device   d_info[0] = number of line feeds; the host synchronises, derives
 *                                     n_records = n_lines / 4 (a last line without line feed counts) and sizes
 *                                     the per-record arrays
 *   2. cah_fastq_index_device         d_rec6 (the columns of cah_fastq_scan), d_seq_off / d_seq_len = the reads as an
 *                                     offsets + lens view INTO THE RAW CHUNK (what cah_match_batch takes: nothing is
 *                                     packed), d_info[1] = (first bad record + 1) << 8 | code or ~0 (1: no '@',
 *                                     2: no '+', 3: sequence and quality lengths differ)
 *   3. cah_fastq_format_device        "@name\nSEQ[beg:end]\n+\nQUAL[beg:end]\n" of every kept record in record
 *                                     order, d_info[3] = bytes written (<= chunk bytes + 4 * n_records)
 * d_info: int64[8]. */
size_t cah_fastq_device_scratch_bytes(int64_t chunk_bytes, int64_t max_records);
int cah_fastq_count_lines_device(const uint8_t *d_buf, int64_t len, void *d_scratch, size_t scratch_bytes,
                                 int64_t *d_info, void *stream);
int cah_fastq_index_device(const uint8_t *d_buf, int64_t len, int64_t n_newlines, int64_t n_records,
                           void *d_scratch, size_t scratch_bytes, int64_t *d_rec6, int64_t *d_seq_off,
                           int32_t *d_seq_len, int64_t *d_info, void *stream);
/* between 2 and 3: what is left of every read after its best match (Match.trimmed(), adapters.py:453-454, :486-487)
 * and whether the record is written (filters of cli.py:735-912: too short, too long, --discard-(un)trimmed).
 * d_adapter_kind[a]: 0 = 3' adapter, 1 = 5' adapter, 2 = anywhere (5' iff the match starts at 0); min_len / max_len
 * < 0 = no limit; d_counters: uint64[8], accumulated: reads, with adapters, bp in, bp out, too short, too long,
 * invalid reads. */
int cah_trim_decide_device(const int32_t *d_out6, const uint8_t *d_status, const int32_t *d_best_adapter,
                           const int32_t *d_seq_len, int64_t n_reads, const uint8_t *d_adapter_kind,
                           int32_t min_len, int32_t max_len, int32_t discard_trimmed,
                           int32_t discard_untrimmed, int32_t *d_beg, int32_t *d_end, uint8_t *d_keep,
                           uint64_t *d_counters, void *stream);
/* ... when modifiers ran in front of the adapter step (-u, --nextseq-trim, -q: cli.py:938-954): the matcher saw the
 * window [d_win_beg[r], d_win_beg[r] + d_win_len[r]) of read r (full length d_seq_len[r]; d_win_beg NULL: windows
 * start at 0); d_beg / d_end come back relative to the read, "bp in" counts the full reads.  intervals_only != 0:
 * no filter is applied and "bp out" is not counted -- cah_trim_filter_device follows (modifiers behind the adapter
 * step). */
int cah_trim_decide_window_device(const int32_t *d_out6, const uint8_t *d_status, const int32_t *d_best_adapter,
                                  const int32_t *d_win_beg, const int32_t *d_win_len, const int32_t *d_seq_len,
                                  int64_t n_reads, const uint8_t *d_adapter_kind, int32_t min_len,
                                  int32_t max_len, int32_t discard_trimmed, int32_t discard_untrimmed,
                                  int32_t intervals_only, int32_t *d_beg, int32_t *d_end, uint8_t *d_keep,
                                  uint64_t *d_counters, void *stream);
/* ... with the adapter step's other actions that only move the kept interval (reference modifiers.py:170-198, :225-251,
 * one round of matching): none -- the read is kept whole (the match still counts), retain -- the read is trimmed but keeps
 * the adapter (Match.retained_adapter_interval, adapters.py:446-447, :479-480), crop -- only the adapter is kept. */
#define CAH_ACTION_TRIM 0
#define CAH_ACTION_NONE 1
#define CAH_ACTION_RETAIN 2
#define CAH_ACTION_CROP 3
int cah_trim_decide_action_device(const int32_t *d_out6, const uint8_t *d_status, const int32_t *d_best_adapter,
                                  const int32_t *d_win_beg, const int32_t *d_win_len, const int32_t *d_seq_len,
                                  int64_t n_reads, const uint8_t *d_adapter_kind, int32_t action, int32_t min_len,
                                  int32_t max_len, int32_t discard_trimmed, int32_t discard_untrimmed,
                                  int32_t intervals_only, int32_t *d_beg, int32_t *d_end, uint8_t *d_keep,
                                  uint64_t *d_counters, void *stream);
/* ... and when modifiers run BEHIND the adapter step as well (--poly-a, -l; cli.py:956-973) or --max-ee is given: call
 * cah_trim_decide*_device without limits (min_len = max_len = -1, no discards), move d_beg / d_end with the later
 * modifiers, then let this apply the filters in the reference's order (too short, too long, too many expected
 * errors, discards; cli.py:735-912).  d_ee: expected errors of the kept intervals or NULL; max_ee < 0: no limit.
 * d_counters as above; [3] bp out, [4] too short, [5] too long, [7] too many expected errors are accumulated. */
int cah_trim_filter_device(const int32_t *d_beg, const int32_t *d_end, const uint8_t *d_status,
                           const double *d_ee, int64_t n_reads, int32_t min_len, int32_t max_len,
                           double max_ee, int32_t discard_trimmed, int32_t discard_untrimmed,
                           uint8_t *d_keep, uint64_t *d_counters, void *stream);
int cah_fastq_format_device(const uint8_t *d_buf, const int64_t *d_rec6, int64_t n_records, const int32_t *d_beg,
                            const int32_t *d_end, const uint8_t *d_keep, void *d_scratch, size_t scratch_bytes,
                            int64_t chunk_bytes, uint8_t *d_out, int64_t out_cap, int64_t *d_info, void *stream);
/* AdapterCutter's marking actions (reference modifiers.py:170-198: masked_read / lowercased_read) on a chunk in HBM, IN
 * PLACE: of the characters [d_beg[r], d_end[r]) of read r (the read as the adapter step saw it) those OUTSIDE [d_mark_beg[r],
 * d_mark_end[r]) -- what trimming would have removed; all relative to the read -- become 'N' (mode 1: --action=mask) or lower
 * case, with the characters inside in upper case (mode 2: --action=lowercase; a read without a match has its whole window
 * "inside": the reference upper-cases every read before it looks for adapters, modifiers.py:222-223).  The modifiers behind
 * the adapter step (--poly-a, -l) and cah_fastq_format_device then see the marked read, as the reference's do. */
int cah_mark_reads_device(uint8_t *d_buf, const int64_t *d_rec6, int64_t n_records, const int32_t *d_beg,
                          const int32_t *d_end, const int32_t *d_mark_beg, const int32_t *d_mark_end, int mode,
                          void *stream);
/* --revcomp on a chunk in HBM (ReverseComplementer, reference modifiers.py:264-308): the caller matches every read and its
 * reverse complement (cah_revcomp_reads_batch into a second buffer at the same offsets, two cah_match_batch calls), takes
 * the reverse complement where its score is HIGHER (modifiers.py:287) and then
 *   cah_revcomp_in_place_device     turns the records with d_flags[r] != 0 around IN PLACE inside the window the adapter
 *                                   step saw ([d_win_beg[r], + d_win_len[r]) relative to the read; d_win_beg NULL: from
 *                                   0): the sequence becomes its reverse complement (csrc/revcomp.h), the qualities are
 *                                   reversed -- the modifiers behind the adapter step and the formatter see that read;
 *   cah_fastq_format_suffix_device  cah_fastq_format_device with `suffix` (suffix_len <= CAH_MAX_NAME_SUFFIX bytes) behind
 *                                   the name of those records (modifiers.py:296-297: `trimmed_read.name += rc_suffix`);
 *                                   out_cap >= chunk length + (4 + suffix_len) * n_records always suffices. */
#define CAH_MAX_NAME_SUFFIX 32
int cah_revcomp_in_place_device(uint8_t *d_buf, const int64_t *d_rec6, int64_t n_records, const int32_t *d_win_beg,
                                const int32_t *d_win_len, const uint8_t *d_flags, void *stream);
int cah_fastq_format_suffix_device(const uint8_t *d_buf, const int64_t *d_rec6, int64_t n_records, const int32_t *d_beg,
                                   const int32_t *d_end, const uint8_t *d_keep, const uint8_t *d_flags,
                                   const char *suffix, int32_t suffix_len, void *d_scratch, size_t scratch_bytes,
                                   int64_t chunk_bytes, uint8_t *d_out, int64_t out_cap, int64_t *d_info, void *stream);
/* --info-file rows of a chunk in HBM (InfoFileWriter, reference steps.py:215-253; Match.get_info_records,
 * adapters.py:395-417), the lines of every record in record order into d_out, d_total[0] = bytes written:
 *   per match of a read (rounds of single adapters, --times N: d_out6 / d_status / d_best hold `rounds` arrays of n_records
 *   rows back to back, cah_match_batch's results round after round -- or the two parts of a linked adapter, its 5' part as
 *   round 0 and its 3' part as round 1 (LinkedMatch.get_info_records, adapters.py:1157-1171) --; a round without a match gives
 *   no row):
 *       name[suffix] TAB errors TAB rstart TAB rstop TAB seq[:rstart] TAB seq[rstart:rstop] TAB seq[rstop:] TAB adapter name
 *       TAB qual[:rstart] TAB qual[rstart:rstop] TAB qual[rstop:] TAB rc LF
 *     seq = the read as InfoFileWriter holds it at that round: the WHOLE read as it came in (turned around if its reverse
 *     complement won: cah_revcomp_in_place_device with the whole read as the window), trimmed the way every earlier match
 *     trims (`current_read = match.trimmed(current_read)`, steps.py:247; d_kinds: per adapter 0 = 3' -- keeps what is in
 *     front of the match --, 1 = 5', 2 = anywhere: 5' when the match starts at position 0; NULL: all 3'), cut at the
 *     coordinates of the match -- which were found on what the modifiers in front of the adapter step and the earlier rounds
 *     left of the read; the reference does exactly that (steps.py:233-236: info.original_read) --; rc = "" (d_is_rc NULL: no
 *     --revcomp), "0" or "1" (RC_MAP, steps.py:224), and the names of the reads with d_is_rc[r] != 0 carry `suffix`;
 *   a read without a match:  name TAB -1 TAB seq[fb:fe] TAB qual[fb:fe] LF   with [fb, fe) = [d_final_beg[r], d_final_end[r]):
 *     the read as it is written (steps.py:248-251).
 * d_names / d_name_off (int32[n_names + 1]): the adapters' names back to back in plan order (device memory).  d_scratch as
 * for cah_fastq_format_device (called after it: the formatter's arrays are reused).
 * out_cap >= rounds * (chunk length + n_records * (longest adapter name + suffix_len + 48)) always suffices. */
int cah_info_format_device(const uint8_t *d_buf, const int64_t *d_rec6, int64_t n_records, const int32_t *d_out6,
                           const uint8_t *d_status, const int32_t *d_best, int32_t rounds, const uint8_t *d_kinds,
                           const int32_t *d_final_beg, const int32_t *d_final_end, const uint8_t *d_names,
                           const int32_t *d_name_off, int32_t n_names, const uint8_t *d_is_rc, const char *suffix,
                           int32_t suffix_len, void *d_scratch, size_t scratch_bytes, int64_t chunk_bytes, uint8_t *d_out,
                           int64_t out_cap, int64_t *d_total, void *stream);

/* ---- AdapterIndex (adapters.py:1289-1551) on the GPU ---------- */
/* Many anchored adapters of one kind (all 5' "^ADAPTER" or all 3' "ADAPTER$", no wildcards, at most
 * 3 errors) are matched with one dictionary lookup per read: every string within k errors of any
 * adapter (_align.pyx:784-882 edit_environment with indels, :717-781 hamming_sphere without) maps to
 * (adapter, errors, matches); collisions keep the entry with more matches, equal matches make the
 * string ambiguous and remove it (adapters.py:1425-1464).  cah_index_create builds that dictionary
 * on the host and stores it as a hash table (2-bit packed strings when every adapter is plain ACGT
 * and at most 60 characters long, hashed and verified byte strings otherwise); cah_index_lookup_batch probes
 * it with one GPU lane per read (multi-length rule :1487-1530, 'N' re-alignment :1532-1551) and
 * writes the same outputs as cah_match_batch: out6 = (0, len(adapter), rstart, rstop, matches,
 * errors), best_adapter (index into the adapter list, -1 = none), status.
 * Limits of this build: adapters of 1..1000 ASCII characters, at most 64 different string lengths
 * per index (CAH_EUNSUPPORTED otherwise).
 * Errors: CAH_EINVAL "Adapter list is empty" / "Error rate too high" (adapters.py:1309, :1385). */
typedef struct cah_index cah_index;
typedef struct cah_index_adapter {
    const char *sequence;       /* upper-case, as SingleAdapter stores it */
    int32_t length;
    double max_error_rate;
    int32_t indels;             /* 0: Hamming distance only (--no-indels) */
    /* the adapter's own k-mer prefilter (what its match_to() asks first, adapters.py:707-724);
     * consulted only when a read's affix contains 'N' and the adapter has to be re-aligned
     * (:1543-1551).  n_kmer_sets < 0: no prefilter (MockKmerFinder, e.g. --no-indels). */
    const cah_kmer_set *kmer_sets;
    int32_t n_kmer_sets;
} cah_index_adapter;
int cah_index_create(const cah_index_adapter *adapters, int32_t n_adapters, int32_t prefix,
                     cah_index **out);
void cah_index_destroy(cah_index *index);
/* number of strings, number of ambiguous strings dropped, indexed string lengths (longest first;
 * lengths must have room for 64 entries) */
int cah_index_info(const cah_index *index, int64_t *n_strings, int32_t *n_ambiguous,
                   int32_t *lengths, int32_t *n_lengths);
/* host-side dictionary query of one string (introspection / tests) */
int cah_index_get(const cah_index *index, const char *s, int32_t len, int32_t *found,
                  int32_t *adapter, int32_t *errors, int32_t *matches);
int cah_index_lookup_batch(const cah_index *index, const uint8_t *d_seqs, const int64_t *d_offsets,
                           const int32_t *d_lens, int64_t n_reads, int32_t *d_out6,
                           int32_t *d_best_adapter, uint8_t *d_status, void *stream);
int cah_index_lookup_batch_host(const cah_index *index, const uint8_t *seqs, const int64_t *offsets,
                                int64_t n_reads, int32_t *out6, int32_t *best_adapter,
                                uint8_t *status);

/* Reversed copy of every read of a packed batch: Rightmost* adapters search the REVERSED read with the reversed
 * adapter (adapters.py:766 RightmostFrontAdapter.match_to, :870 RightmostBackAdapter: `sequence[::-1]`).
 * d_lens as in cah_match_batch; d_out_offsets int64[n_reads]: where read r starts in d_out (for a packed copy the
 * running sum of the lengths).  All device pointers, asynchronous on `stream`. */
int cah_reverse_reads_batch(const uint8_t *d_seqs, const int64_t *d_offsets, const int32_t *d_lens,
                            int64_t n_reads, const int64_t *d_out_offsets, uint8_t *d_out, void *stream);

/* The reverse complement of every read of a packed batch -- what ReverseComplementer (modifiers.py:264-308) searches
 * besides the read itself.  complement == 0 only reverses (quality strings).  d_select (uint8[n_reads], may be NULL):
 * only reads with d_select[r] != 0 are turned around, the others are copied -- one pass merges "the orientation that
 * scored better" of every read into one batch for the modifiers that follow.  Layout arguments as
 * cah_reverse_reads_batch; asynchronous on `stream`. */
int cah_revcomp_reads_batch(const uint8_t *d_seqs, const int64_t *d_offsets, const int32_t *d_lens,
                            int64_t n_reads, const int64_t *d_out_offsets, uint8_t *d_out,
                            int32_t complement, const uint8_t *d_select, void *stream);

/* ---- quality / NextSeq / poly-A trimming, expected errors ------ */
/* The O(n) per-read scans that run just before adapter matching (cli.py:938-954), for a batch
 * whose qualities are packed with the SAME offsets as the sequences.  All device pointers;
 * d_lens as in cah_match_batch (NULL = offsets[r+1]-offsets[r]).
 *   cah_quality_trim_batch   quality_trim_index (qualtrim.pyx:22-70)  -> d_start_stop int32[n,2]
 *   cah_nextseq_trim_batch   nextseq_trim_index (:73-113)             -> d_stop int32[n]
 *   cah_poly_a_trim_batch    poly_a_trim_index (:116-165)             -> d_index int32[n]
 *   cah_expected_errors_batch expected_errors (:168-190, expected_errors.h:103-140) -> double[n],
 *                            bit-identical (same accumulation order); an invalid phred value gives
 *                            -1.0 and d_status[r] = CAH_INVALID (the reference raises ValueError). */
int cah_quality_trim_batch(const uint8_t *d_quals, const int64_t *d_offsets, const int32_t *d_lens,
                           int64_t n_reads, int32_t cutoff_front, int32_t cutoff_back,
                           int32_t base, int32_t *d_start_stop, void *stream);
int cah_nextseq_trim_batch(const uint8_t *d_seqs, const uint8_t *d_quals, const int64_t *d_offsets,
                           const int32_t *d_lens, int64_t n_reads, int32_t cutoff, int32_t base,
                           int32_t *d_stop, void *stream);
/* ... when the qualities of read r start at d_quals[d_qual_offsets[r]] instead of d_quals[d_offsets[r]]: sequences
 * and qualities used in place in a raw FASTQ chunk that cah_fastq_index_device indexed (d_lens required). */
int cah_nextseq_trim_batch_q(const uint8_t *d_seqs, const uint8_t *d_quals, const int64_t *d_offsets,
                             const int64_t *d_qual_offsets, const int32_t *d_lens, int64_t n_reads,
                             int32_t cutoff, int32_t base, int32_t *d_stop, void *stream);
int cah_poly_a_trim_batch(const uint8_t *d_seqs, const int64_t *d_offsets, const int32_t *d_lens,
                          int64_t n_reads, int32_t revcomp, int32_t *d_index, void *stream);
int cah_expected_errors_batch(const uint8_t *d_quals, const int64_t *d_offsets,
                              const int32_t *d_lens, int64_t n_reads, int32_t base,
                              double *d_expected, uint8_t *d_status, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CUTADAPT_HIP_H */
