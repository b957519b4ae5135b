// stream2.h -- the word machinery of k_filter_stream2 (stream2.hip), in a form the kernel AND a plain C++ host
// model (tests/host_model/stream_model.cpp, compiled with g++) share, like back_scan.h for the cost scan.
//
// KmerFinder.kmers_present (reference _kmer_finder.pyx:170-257) over equally long reads, for the search sets
// kmer_heuristic builds for a 3' adapter: whole-read sets (0, None) -> lead words, tail sets (-L, None) -> T-words
// (CahLeanFilter, cah_device.h).  Both kinds are shift-and words in which every k-mer is followed by
// CAH_LEAN_DELAY = 3 delay bits that pass every byte, and both advance FOUR characters per step:
//     R4 = ((R << 4) | S3) & T3[c1] & T2[c2] & T1[c3] & T0[c4],   Ts[c] = ((M[c] | PASS) << s) | S(s-1),
//     S(k) = START | START << 1 | .. | START << k
// (four single steps R' = ((R << 1) | START) & M[c] written out).  After a group of four characters whose last
// one is t, bit "end + d" of a k-mer says that the k-mer ended at t - d.  A lead k-mer counts wherever it ends; a
// k-mer of length q of the tail set (-L, None) counts iff it started at or after n - L, i.e. iff
// dist + d <= L - q with dist = n - 1 - t -- one found mask per word and dist (tw_found), the same for every read
// of a batch of equally long reads.
#pragma once
#include <stdint.h>

#include "cah_device.h"

#if defined(__HIPCC__)
#define S2_HD __host__ __device__ __forceinline__
#else
#define S2_HD inline
#endif

// S(s-1): the start bits of the s positions before the current one (0 for s == 0)
S2_HD uint32_t s2_fill(uint32_t init, int s) {
    uint32_t f = 0;
    for (int j = 0; j < s; ++j) f |= init << j;
    return f;
}
// table entry of byte value c for the character at distance s from the end of its group (s = 3: first character)
S2_HD uint32_t s2_entry(uint32_t mask_c, uint32_t pass, uint32_t init, int s) {
    return ((mask_c | pass) << s) | s2_fill(init, s);
}
S2_HD uint32_t s2_init4(uint32_t init) { return init | (init << 1) | (init << 2) | (init << 3); }
S2_HD uint32_t s2_step4(uint32_t R, uint32_t init4, uint32_t m0, uint32_t m1, uint32_t m2, uint32_t m3) {
    return ((R << 4) | init4) & m0 & m1 & m2 & m3;
}
// index into tw_found of the group whose last character is t, in a read of n characters.  In range for every group
// of a chunk that touches a window or the read's end (t <= n + 14, and n - t <= CAH_LEAN_SPAN + 16 there): no clamping.
S2_HD int s2_found_index(int n, int t) { return n - 1 - t + CAH_TW_DIST0; }
// T-words the 16-character chunk at `pos` must advance: word w is idle (state 0) until the chunk that holds
// position n - tw_span[w], the first at which one of its k-mers may start; the spans fall from word to word
S2_HD int s2_active_tw(const int32_t* tw_span, int n_tw, int n, int pos) {
    int na = 0;
    for (int w = 0; w < n_tw; ++w)
        if (pos + 16 > n - tw_span[w]) na = w + 1;
    return na;
}
