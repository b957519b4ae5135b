/*
 * synth_reads.c -- CPU twin of the device-side synthetic read generator.
 *
 * TEST INFRASTRUCTURE (see cutadapt_oracle.c header).  The benchmark workloads
 *  are defined as a pure function
 *     read(seed, global_read_index) -> bytes[read_len]
 * so that (a) every GPU rank can generate its own shard in HBM without a host round
 * trip, (b) the result does not depend on the number of GPUs, and (c) the CPU baseline
 * and the parity checks can regenerate exactly the same reads for any sub-range.
 * The HIP implementation is cutadapt_amd/csrc/synth_kernel.hip (cah_synth_reads); the
 * two must agree byte for byte (tests/test_synth.py).
 *
 * Read model: bases i.i.d. uniform over ACGT; with
 * probability p_adapter one adapter (chosen uniformly from the list) is written at a
 * uniform position in [0, read_len], each adapter character independently edited with
 * probability p_edit (50 % substitution, 25 % insertion, 25 % deletion), the copy is
 * truncated at the read end and what follows it stays random; finally every base turns
 * into 'N' with probability p_n.
 */
#include <stdint.h>

static inline uint64_t mix64(uint64_t z) {           /* splitmix64 finaliser */
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

#define GOLDEN 0x9E3779B97F4A7C15ULL

static inline uint64_t stream_key(uint64_t seed, uint64_t index, uint64_t stream) {
    return mix64(mix64(seed + GOLDEN * (stream + 1)) ^ (index * GOLDEN));
}

static inline uint64_t draw(uint64_t key, uint64_t counter) {
    return mix64(key + GOLDEN * (counter + 1));
}

void orc_synth_reads(uint64_t seed, int64_t first_index, int64_t n_reads, int read_len,
                     uint32_t p_adapter_u32, uint32_t p_edit_u32, uint32_t p_n_u16,
                     const char *adapters, const int *adapter_off, int n_adapters,
                     unsigned char *seqs) {
    static const char BASES[4] = {'A', 'C', 'G', 'T'};
    for (int64_t r = 0; r < n_reads; r++) {
        const uint64_t idx = (uint64_t)(first_index + r);
        unsigned char *out = seqs + r * (int64_t)read_len;
        /* stream 0: random bases, 32 per draw (2 bits each, low bits first) */
        const uint64_t k0 = stream_key(seed, idx, 0);
        for (int j = 0; j < read_len; j += 32) {
            uint64_t w = draw(k0, (uint64_t)(j >> 5));
            for (int t = 0; t < 32 && j + t < read_len; t++) out[j + t] = BASES[(w >> (2 * t)) & 3];
        }
        /* stream 1: adapter insertion */
        const uint64_t k1 = stream_key(seed, idx, 1);
        uint64_t u = draw(k1, 0);
        if (n_adapters > 0 && (uint32_t)(u >> 32) < p_adapter_u32) {
            int which = (int)((u & 0xFFFFFFFFULL) % (uint64_t)n_adapters);
            const char *ad = adapters + adapter_off[which];
            int m = adapter_off[which + 1] - adapter_off[which];
            int pos = (int)(draw(k1, 1) % (uint64_t)(read_len + 1));
            for (int i = 0; i < m && pos < read_len; i++) {
                uint64_t e = draw(k1, (uint64_t)(2 + i));
                char c = ad[i];
                if ((uint32_t)(e & 0xFFFFFFFFULL) < p_edit_u32) {
                    unsigned kind = (unsigned)(e >> 32) & 3u;
                    unsigned rb = (unsigned)(e >> 34) & 3u;       /* a random base */
                    if (kind < 2) {                               /* substitution */
                        unsigned cur = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u;
                        unsigned step = 1u + (unsigned)((e >> 36) % 3ULL);
                        out[pos++] = (unsigned char)BASES[(cur + step) & 3u];
                    } else if (kind == 2) {                       /* insertion before c */
                        out[pos++] = (unsigned char)BASES[rb];
                        if (pos < read_len) out[pos++] = (unsigned char)c;
                    } else {                                      /* deletion of c */
                    }
                } else {
                    out[pos++] = (unsigned char)c;
                }
            }
        }
        /* stream 2: N substitution, four 16-bit lots per draw */
        if (p_n_u16) {
            const uint64_t k2 = stream_key(seed, idx, 2);
            for (int j = 0; j < read_len; j += 4) {
                uint64_t w = draw(k2, (uint64_t)(j >> 2));
                for (int t = 0; t < 4 && j + t < read_len; t++)
                    if (((w >> (16 * t)) & 0xFFFFULL) < p_n_u16) out[j + t] = 'N';
            }
        }
    }
}
