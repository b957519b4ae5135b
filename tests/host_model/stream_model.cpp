// stream_model.cpp -- plain C++ restatement of k_filter_stream2's word machinery (cutadapt_amd/csrc/stream2.hip), built
// on the very header the kernel uses (stream2.h) and on the product's own CahLeanFilter tables (cah_plan_debug_lean).
// Test infrastructure only (tests/test_stream2_model.py): never loaded by the product.
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <vector>

#include "../../cutadapt_amd/csrc/stream2.h"

extern "C" {

size_t sm_lean_size() { return sizeof(CahLeanFilter); }
int sm_tw_ok(const CahLeanFilter* lf) { return lf->ok && lf->tw_ok; }
int sm_n_words(const CahLeanFilter* lf, int* n_lead, int* n_tw) { *n_lead = lf->n_lead; *n_tw = lf->n_tw; return 0; }

// n_reads reads of n characters each, back to back.  present[r]: 0 / 1 / 2 (a byte >= 0x80); hit_pos[r]: first position of
// the 4-character group in which the first k-mer that counts ends, -1 if none.
// every_word != 0: every T-word is advanced over the whole read (the result must not depend on when a word wakes up)
int sm_filter_batch(const CahLeanFilter* lf, const uint8_t* seqs, int64_t n_reads, int n, uint8_t* present,
                    int32_t* hit_pos, int every_word) {
    if (!lf->ok || !lf->tw_ok) return 1;
    const int NL = lf->n_lead, NT = lf->n_tw;
    // tables as the kernel builds them
    std::vector<uint32_t> lead(4 * 128 * (NL > 0 ? NL : 1)), tail(4 * 128 * (NT > 0 ? NT : 1));
    for (int sh = 0; sh < 4; ++sh)
        for (int c = 0; c < 128; ++c) {
            for (int w = 0; w < NL; ++w)
                lead[(sh * 128 + c) * NL + w] = s2_entry(lf->lead_mask[w][c], lf->lead_pass[w], lf->lead_init[w], sh);
            for (int w = 0; w < NT; ++w)
                tail[(sh * 128 + c) * NT + w] = s2_entry(lf->tw_mask[w][c], lf->tw_pass[w], lf->tw_init[w], sh);
        }
    for (int64_t r = 0; r < n_reads; ++r) {
        const uint8_t* q = seqs + r * (int64_t)n;
        uint32_t RL[CAH_LEAN_MAX_LEAD] = {0}, RT[CAH_LEAN_MAX_TW] = {0};
        int hp = -1;
        bool invalid = false;
        for (int i = 0; i < n; ++i) invalid |= q[i] >= 0x80;
        for (int pos = 0; pos < n && hp < 0 && !invalid; pos += 16) {
            // the kernel's chunk: every T-word from the chunk on in which the widest window opens, all four groups
            // (characters past the read's end are NUL); one look at the four groups' found words per chunk
            const bool tails = every_word || (NT > 0 && pos + 16 > n - lf->tw_span[0]);
            uint32_t f[4] = {0, 0, 0, 0};
            for (int g = 0; g < 4; ++g) {
                unsigned c[4];
                for (int i = 0; i < 4; ++i) { const int p = pos + 4 * g + i; c[i] = p < n ? q[p] : 0; }
                for (int w = 0; w < NL; ++w) {
                    RL[w] = s2_step4(RL[w], s2_init4(lf->lead_init[w]), lead[(3 * 128 + c[0]) * NL + w], lead[(2 * 128 + c[1]) * NL + w],
                                     lead[(1 * 128 + c[2]) * NL + w], lead[(0 * 128 + c[3]) * NL + w]);
                    f[g] |= RL[w] & lf->lead_found[w];
                }
                const int idx = s2_found_index(n, pos + 4 * g + 3);
                for (int w = 0; tails && w < NT; ++w) {
                    RT[w] = s2_step4(RT[w], s2_init4(lf->tw_init[w]), tail[(3 * 128 + c[0]) * NT + w], tail[(2 * 128 + c[1]) * NT + w],
                                     tail[(1 * 128 + c[2]) * NT + w], tail[(0 * 128 + c[3]) * NT + w]);
                    const bool in_table = idx >= 0 && idx < CAH_TW_DIST_LEN;
                    if (!in_table && !every_word) return 2;                                 // the kernel does not clamp
                    f[g] |= RT[w] & (in_table ? lf->tw_found[w][idx] : 0u);
                }
            }
            for (int g = 3; g >= 0; --g) if (f[g] != 0) hp = pos + 4 * g;
        }
        present[r] = invalid ? 2 : (hp >= 0 ? 1 : 0);
        hit_pos[r] = invalid ? -1 : hp;
    }
    return 0;
}

// u / H as the kernel computes it (multiply + shift): returns the first u < 320 for which it is wrong, -1 if none
int sm_check_unit_division(int H) {
    const unsigned magic = (65536u + (unsigned)H - 1u) / (unsigned)H;
    for (unsigned u = 0; u < 320; ++u)
        if (((u * magic) >> 16) != u / (unsigned)H) return (int)u;
    return -1;
}

}  // extern "C"
