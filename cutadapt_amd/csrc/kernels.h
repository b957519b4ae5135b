// kernels.h -- launch interface between api.cpp (host) and kernels.hip (device).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cah_device.h"

#define CAH_KEY_SHIFT 2            // queue key resolution: 4 read columns
#define CAH_QUEUE_BINS 256         // survivor queue is ordered by key = min(first-hit position >> CAH_KEY_SHIFT, 255)

struct FilterArgs {
    const CahKmerWord* words;        // this adapter's packed shift-and words (HBM)
    int32_t n_words;
    const uint8_t* seqs;
    const int64_t* offsets;
    const int32_t* lens;             // may be NULL
    int64_t n_reads;
    int64_t max_read_len;
    unsigned long long* work_counter;   // zeroed before launch
    uint8_t* present;                // MODE 0
    uint8_t* status;                 // MODE 1: only written for invalid reads
    int32_t* queue;                  // MODE 1: surviving read indices, in runs ordered by hit position
    unsigned long long* queue_count; // MODE 1: zeroed before launch
    uint8_t* queue_keys;             // MODE 1: per queue entry, min(first-hit position >> CAH_KEY_SHIFT, 255)
    const unsigned long long* batch_flag;   // k_filter_lean: *batch_flag == 0 <=> all reads have one length (its
                                            // UNIFORM variant works, the ragged one leaves); NULL = no check made
    const CahLeanFilter* lean;       // k_filter_lean / k_filter_stream only
    int32_t stream_n_lo, stream_n_hi;   // equally long reads of these lengths are k_filter_stream's (an instance
                                        // takes its own range, the per-lane uniform kernel leaves the union alone)
    // equally long reads the HOST knows about (cah_match_batch_uniform): read r is seqs[uniform_first + r * uniform_len
    // ...) -- no offsets array is read, no batch check is needed.  uniform_len == 0: lengths come from offsets.
    int64_t uniform_first = 0;
    int32_t uniform_len = 0;
    // k_filter_stream2 only: offsets[r] is where the view of read r starts inside the uniform PARENT batch described by
    // (uniform_first, uniform_len); 1: the view ends where that read ends (second stage of a linked adapter); 2: it is
    // lens[r] characters long and ends anywhere inside the read (reads cut at their 3' end: the RV form)
    int32_t suffix_views = 0;
    // ... and, when `front` is set, the view starts are decided by the kernel itself: the anchored 5' adapter of a linked
    // adapter (no error tolerated, m <= 32) is compared with the read's head; the kernel writes the front stage's rows
    // and the views the later kernels read (cah_linked_match_batch_uniform)
    const CahMatcher* front = nullptr;
    int32_t* front_out6 = nullptr;
    uint8_t* front_status = nullptr;
    int32_t* front_best = nullptr;
    int64_t* view_starts = nullptr;
    int32_t* view_lens = nullptr;
    int32_t* clear_best;             // with clear_out6, may be NULL: best_adapter[r] = -1 for the same reads
    int32_t* clear_out6;             // MODE 1, may be NULL: the result rows (6 x int32 per read) of every read the
                                     // kernel looks at are zeroed on the way (rows of reads that match are written
                                     // later, by the scan / DP kernels) -- saves the caller a memset of 24 B per read
};

struct DpArgs {
    // equally long reads the HOST knows about (cah_match_batch_uniform): read r = seqs[uniform_first + r * uniform_len ..),
    // `offsets` is not read.  0: the packed layout (offsets / lens).
    int64_t uniform_first = 0;
    int32_t uniform_len = 0;

    const CahMatcher* matcher;
    const uint8_t* seqs;
    const int64_t* offsets;
    const int32_t* lens;             // may be NULL
    int64_t n_reads;
    int64_t max_read_len;
    const int32_t* queue;            // NULL: process reads 0..n_reads-1
    const unsigned long long* queue_count;   // NULL: n_reads
    const uint8_t* queue_keys;       // NULL, or per queue entry the first-hit chunk (lower bound of any k-mer hit)
    unsigned long long* work_counter;        // zeroed before launch
    int32_t* out6;
    uint8_t* status;
    int32_t* best_adapter;           // may be NULL
    int32_t adapter_index;
    int32_t merge_best;              // 0: overwrite (locate_batch); 1: keep best (match_batch)
    // Windowed work list written by k_back_scan (k_dp_packed only; all NULL/0 otherwise): `queue` is filled
    // from both ends -- [0, *queue_count) holds reads with a bounded column window, the last
    // *queue_count_back slots of its queue_cap slots hold reads whose DP runs to the read end -- and
    // win[2*slot] = first column, win[2*slot+1] = last column * 2 + (1: do the last-column scan).
    const int32_t* win;
    const unsigned long long* queue_count_back;
    int64_t queue_cap;
    // Fused multi-adapter mode (k_dp_packed<ROWS, true>): queue entries index `pairs` (read << 32 | adapter << 8
    // | key), the match table of the lane's adapter comes from tab[adapter * CAH_MULTI_TAB_STRIDE + (c & 31)]
    // and results are merged with one 64-bit atomic max per match (pack_best) instead of out6 rows.
    const uint64_t* pairs;
    const uint64_t* tab;
    int32_t n_adapters;
    unsigned long long* best_key;
    // Streaming form (multi2.h, round 6): a TAIL pair's match that reaches further back than its error class's last row
    // is checked against the reference's own tail k-mers before it is merged (m2_ref_present; NULL: no such check)
    const struct CahMulti2Header* m2_hdr = nullptr;
    const int32_t* m2_ref_begin = nullptr;
    const uint32_t* m2_ref_list = nullptr;
};

#define CAH_MULTI_TAB_STRIDE 33   // 32 entries + 1 of padding: spreads the adapters' tables over the LDS banks

// MultipleAdapters' order (adapters.py:1278-1285: higher score, then fewer errors, then the first adapter)
// as ONE unsigned 64-bit key with the match itself in the low bits, so that the best match of a read over
// any number of adapters processed in any order is atomicMax(key):
//   [score + 128 : 8 @54][127 - errors : 7 @47][4095 - adapter : 12 @35][ref_stop : 7 @28][query_start : 20 @8]
//   [query_stop - query_start : 8 @0]        (3' adapters: ref_start = 0; a match always gives key != 0)
__host__ __device__ inline unsigned long long pack_best(int score, int errors, int adapter, int ref_stop,
                                                         int query_start, int query_stop) {
    return ((unsigned long long)(unsigned)(score + 128) << 54) | ((unsigned long long)(unsigned)(127 - errors) << 47) |
           ((unsigned long long)(unsigned)(4095 - adapter) << 35) | ((unsigned long long)(unsigned)ref_stop << 28) |
           ((unsigned long long)(unsigned)query_start << 8) | (unsigned long long)(unsigned)(query_stop - query_start);
}

// k_back_scan: bit-parallel cost scan + classification (back_scan.h) of the reads of a work list
struct ScanArgs {
    // equally long reads the HOST knows about (cah_match_batch_uniform): read r = seqs[uniform_first + r * uniform_len ..),
    // `offsets` is not read.  0: the packed layout (offsets / lens).
    int64_t uniform_first = 0;
    int32_t uniform_len = 0;

    const CahMatcher* matcher;
    const uint8_t* seqs;
    const int64_t* offsets;
    const int32_t* lens;             // may be NULL
    int64_t n_reads;
    int64_t max_read_len;
    const int32_t* queue;            // NULL: reads 0..n_reads-1
    const unsigned long long* queue_count;
    const uint8_t* queue_keys;       // NULL, or the first-hit group of every queue entry (column skipping)
    unsigned long long* work_counter;        // zeroed before launch
    int32_t* out6;
    uint8_t* status;
    int32_t* best_adapter;           // may be NULL
    int32_t adapter_index;
    int32_t merge_best;
    // fused multi-adapter mode (k_back_scan<true>), see DpArgs; dp_queue then receives pair indices
    const uint64_t* pairs;
    const uint64_t* tab;
    int32_t n_adapters;
    int32_t multi_skip_ok;
    unsigned long long* best_key;
    int32_t* dp_queue;               // out: reads that need the cell DP (filled from both ends, see DpArgs)
    int32_t* dp_win;
    unsigned long long* dp_count_front;      // zeroed before launch
    unsigned long long* dp_count_back;
    int64_t dp_cap;
    // stragglers (single-adapter mode): when at most retry_threshold lanes of a wave are still scanning and have 32+
    // columns to go, they are appended here and scanned again by a second launch over this list (retry_threshold
    // 0 there).  retry_threshold 0: off.  The list may fill up (retry_cap); waves then run to the end.
    int32_t retry_threshold;
    int32_t* retry_queue;
    uint8_t* retry_keys;
    unsigned long long* retry_count;         // zeroed before launch
    int64_t retry_cap;
    int32_t kind;                    // form of the column, bs_kind_of(m) (back_scan.h); 0 = the 64-bit word (always valid)
    int32_t tile;                    // entries a workgroup takes per atomic (256, 512, 768 or 1024; 0 = 1024)
    int32_t early_stop;              // 1: the reads come from a prefilter that has looked at every character (invalid
                                     // bytes are flagged there), so the scan may stop before the read end (back_scan.h)
    int64_t queue_limit;             // > 0: at most this many entries of the queue are valid (the retry list's capacity)
};

// k_tiny: prefilter + cost scan of <= 64 reads in one launch of one wave (the per-read API)
struct TinyArgs {
    const CahLeanFilter* lean;       // NULL: no prefilter (Aligner.locate)
    const CahMatcher* matcher;       // scan_ok
    const uint8_t* seqs;
    const int64_t* offsets;
    int64_t n_reads;                 // <= 64
    int64_t max_read_len;
    int32_t* out6;
    uint8_t* status;
    int32_t* dp_queue;               // the cell DP's work list, as k_back_scan leaves it
    int32_t* dp_win;
    unsigned long long* dp_count_front;
    unsigned long long* dp_count_back;
    unsigned long long* dp_work;     // the cell-DP kernel's work counter: zeroed by k_tiny
    int64_t dp_cap;
    int32_t* need_dp;                // out: number of reads on that list
    int32_t* done;                   // out (mapped host memory): set to `ticket` when everything above is written
    int32_t ticket;
    const void* image;               // NULL, or the kernel's LDS tables as a former call with image_out left them
    void* image_out;                 // non-NULL: build the tables from lean / matcher, write them here, return
};
#define CAH_TINY_IMAGE_BYTES (20 * 1024)   // upper bound of the table image of any k_tiny class
hipError_t launch_tiny(const TinyArgs& a, int n_lead, int n_gated, int delay, hipStream_t s);
hipError_t launch_ticket(int32_t* done, int32_t ticket, hipStream_t s);
// tw_ok / n_tw: the plan's T-words (CahLeanFilter): equally long short reads then take k_filter_stream2 (stream2.hip)
hipError_t launch_filter_lean(const FilterArgs& a, int mode, int n_lead, int n_gated, int delay, int tw_ok, int n_tw,
                              int n_cus, hipStream_t s);
hipError_t launch_filter_stream2(const FilterArgs& a, int mode, int n_lead, int n_tw, int n_cus, hipStream_t s);
int stream2_max_len();
int stream2_long_max_len();      // ... of its LONG form (reads walked in segments of 160 characters)
bool stream2_class_ok(int n_lead, int n_tw);
hipError_t launch_uniform_check(const int64_t* offsets, int64_t n_reads, int64_t max_read_len, unsigned long long* flag,
                                int n_cus, hipStream_t s);
hipError_t launch_filter(const FilterArgs& a, int mode, bool narrow_words, int n_cus, hipStream_t s);
hipError_t launch_dp(const DpArgs& a, int m, bool unit_indel_cost, bool back_adapter, int64_t max_items, int n_cus,
                     hipStream_t s);
hipError_t launch_back_scan(const ScanArgs& a, int64_t max_items, int n_cus, hipStream_t s);

// multi.hip: the fused multi-adapter prefilter and the final decode of the per-read best keys
struct MultiFilterArgs {
    // equally long reads the HOST knows about (cah_match_batch_uniform): read r = seqs[uniform_first + r * uniform_len ..),
    // `offsets` is not read.  0: the packed layout (offsets / lens).
    int64_t uniform_first = 0;
    int32_t uniform_len = 0;

    const CahMultiHeader* hdr;
    const CahMultiDir* dir;
    const CahMultiEntry* entries;
    const uint32_t* bitmap;
    const uint8_t* seqs;
    const int64_t* offsets;
    const int32_t* lens;             // may be NULL
    int64_t first_read, n_reads;     // this launch handles reads [first_read, first_read + n_reads)
    int64_t max_read_len;
    unsigned long long* work_counter;        // zeroed before launch
    uint8_t* status;                 // only written for invalid reads
    uint64_t* pairs;                 // out: (read << 32 | adapter << 8 | key), runs ordered by key
    unsigned long long* pair_count;  // zeroed before launch
    int64_t pair_cap;
};
hipError_t launch_multi_filter(const MultiFilterArgs& a, const CahMultiHeader& host_hdr, int n_cus, hipStream_t s);
hipError_t launch_multi_decode(const unsigned long long* best_key, int64_t n_reads, int32_t* out6, uint8_t* status,
                               int32_t* best_adapter, int n_cus, hipStream_t s, const unsigned long long* err = nullptr);
// multi2.hip: the streaming form of the fused multi-adapter path (equally long short reads; tables: multi2.h)
struct CahMulti2Header;
struct CahM2Slot;
struct Multi2Args {
    int64_t uniform_first;           // read r of the BATCH = seqs[uniform_first + r * uniform_len ..)
    int32_t uniform_len;
    int32_t win_hi, win_lo;          // a tail pair's scan window starts at column n - win (CahMulti2Header::win_dist)
    const CahMulti2Header* hdr;
    const uint16_t* dir;
    const CahM2Slot* entries;
    const uint32_t* bitmap;
    const uint32_t* prefix;
    const uint8_t* seqs;
    int64_t first_read, n_reads;     // this launch handles reads [first_read, first_read + n_reads) of the batch
    uint8_t* status;                 // only written for invalid reads
    unsigned long long* best_key;    // zeroed before launch: pairs decided by the suffix compare are merged here
    uint64_t* pairs;                 // the page pool: max_pages * CAH_M2_PAGE pair records
    uint32_t* page_hdr;              // [max_pages]
    unsigned long long* page_counter;        // zeroed before launch: pages handed out
    int64_t max_pages;
    unsigned long long* err;         // set (never cleared) by the kernels when the pool or a wait gives out: bit 0 = a page beyond
                                     // max_pages was handed out (pairs were lost), bit 1 = a wave waited in vain for its tile
    // the tiles (M2_TILE reads) of [first_read, first_read + n_reads) are drawn from tile_counter, which lives through the
    // ROUNDS of a batch: a launch draws no further tile once more than gate_pages pages are handed out, and the next
    // round (after the scan and the cell DP have emptied the pool) goes on where it stopped
    unsigned long long* tile_counter;
    int64_t n_tiles, gate_pages;
    uint16_t* wmeta;                 // out, per read of the batch: the adapter whose pair saw a further hit (CAH_M2_NO_FLAG ..) | chunk of the earliest such hit << 8
    // RV form: read r of the batch is the VIEW seqs[view_starts[r], + view_lens[r]) inside seqs[uniform_first + r * uniform_len ..)
    // (NULL: the reads themselves)
    const int64_t* view_starts = nullptr;
    const int32_t* view_lens = nullptr;
    // 1: the views lie ANYWHERE in seqs (a packed batch with its offsets, the reads of a raw FASTQ chunk): no read around
    // them, uniform_len is the frame's length alone (>= every view's); the copy gathers every unit from its view's end
    int32_t view_general = 0;
};
struct Multi2ScanArgs {
    int64_t uniform_first;
    int32_t uniform_len;
    int32_t kind;                    // form of the column, bs_kind_of(m)
    int32_t rows_lo;                 // the longest overlap of the first error class: no higher row of a "lo" pair can match
    const CahMatcher* matcher;       // matcher 0: all adapters of the fused path have one shape
    const uint64_t* tab;             // [n_adapters][CAH_MULTI_TAB_STRIDE] padded match words
    int32_t n_adapters;
    const uint8_t* seqs;
    const uint64_t* pairs;
    const uint32_t* page_hdr;
    const unsigned long long* page_counter;
    int64_t max_pages;
    unsigned long long* err;         // (Multi2Args::err)
    unsigned long long* work_counter;        // zeroed before launch
    unsigned long long* best_key;
    int32_t* dp_queue;               // out: pair indices that need the cell DP (filled from both ends, see DpArgs)
    int32_t* dp_win;
    unsigned long long* dp_count_front;      // zeroed before launch
    unsigned long long* dp_count_back;
    int64_t dp_cap;
    const uint16_t* wmeta;           // per read: Multi2Args::wmeta
    const uint32_t* prefix;          // per adapter: its first ten characters (M2Tables::prefix), for the suffix compare
    int32_t lmax0;                   // the largest overlap without error tolerance
    // views inside the reads of the uniform batch (k_multi_stream's RV form; NULL: the reads themselves): the scan works on
    // the end-aligned frame of uniform_len characters and reports in the view's coordinates
    const int64_t* view_starts = nullptr;
    const int32_t* view_lens = nullptr;
    int32_t view_general = 0;        // (Multi2Args::view_general)
};
bool multi2_read_len_ok(const CahMulti2Header& h, int read_len);
size_t multi2_lds_bytes(const CahMulti2Header& h);
hipError_t launch_multi_stream(const Multi2Args& a, const CahMulti2Header& host_hdr, int grid, hipStream_t s);
int multi2_tile_reads();
hipError_t launch_multi_scan(const Multi2ScanArgs& a, int64_t max_pages, int n_cus, hipStream_t s);
// long.hip: adapters longer than 64 characters (column in HBM scratch)
struct LongArgs {
    // equally long reads the HOST knows about (cah_match_batch_uniform): read r = seqs[uniform_first + r * uniform_len ..),
    // `offsets` is not read.  0: the packed layout (offsets / lens).
    int64_t uniform_first = 0;
    int32_t uniform_len = 0;

    const CahLongMatcher* lm;
    const uint8_t* ref;              // encoded adapter, m bytes
    const int32_t* ncnt;             // n_counts, m + 1 entries
    const uint8_t* seqs;
    const int64_t* offsets;
    const int32_t* lens;             // may be NULL
    int64_t n_reads;
    int64_t max_read_len;
    const int32_t* queue;            // NULL: reads 0..n_reads-1
    const unsigned long long* queue_count;
    unsigned long long* work_counter;        // zeroed before launch
    int32_t* scratch;                // 3 * (m + 1) * lanes int32
    int32_t* out6;
    uint8_t* status;
    int32_t* best_adapter;
    int32_t adapter_index;
    int32_t merge_best;
    int32_t* dbg_cost;               // NULL, or (cah_locate_debug_host: ONE read) the DP cost / score matrices
    int32_t* dbg_score;
};
int64_t long_scratch_lanes(int64_t max_items, int n_cus);      // threads k_dp_long is launched with (each owns a column)
hipError_t launch_dp_long(const LongArgs& a, int64_t lanes, hipStream_t s);
hipError_t launch_comparer(const DpArgs& a, int64_t max_items, int n_cus, hipStream_t s);
hipError_t launch_anchored_exact(const DpArgs& a, int64_t max_items, int n_cus, hipStream_t s);
hipError_t launch_validate(const uint8_t* seqs, const int64_t* offsets, const int32_t* lens,
                           int64_t n_reads, int32_t* bad, int n_cus, hipStream_t s);
hipError_t launch_init_best(int32_t* best_adapter, int64_t n_reads, int n_cus, hipStream_t s);
hipError_t launch_synth(uint64_t seed, int64_t first_index, int64_t n_reads, int32_t read_len,
                        uint32_t p_adapter_u32, uint32_t p_edit_u32, uint32_t p_n_u16,
                        const char* d_adapters, const int32_t* d_adapter_off, int32_t n_adapters,
                        uint8_t* d_seqs, int64_t* d_offsets, int n_cus, hipStream_t s);
