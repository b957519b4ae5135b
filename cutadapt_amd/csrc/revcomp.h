// revcomp.h -- the complement of one sequence character, shared by the device kernel (qualtrim.hip: k_reverse_reads)
// and the host formatter (fastq.cpp: cah_chunk_revcomp).
//
// The reference's ReverseComplementer (modifiers.py:264-308) calls dnaio's SequenceRecord.reverse_complement().
// dnaio is a third-party dependency (pyproject.toml: dnaio >= 1.2.3) that is not part of /root/reference; its
// published behaviour is restated here: the IUPAC nucleotide codes are complemented in place, upper and lower
// case separately (A<->T, C<->G, U->A, M<->K, R<->Y, W, S and N stay, V<->B, H<->D), every other byte is left as
// it is, the sequence is reversed and the qualities are reversed.  Parity is anchored on the reference's own
// fixtures for this path: tests/cut/revcomp-single-normalize.fastq (upper- and lower-case ACGT) and
// tests/cut/info-rc.txt.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define CAH_RC_HD __host__ __device__
#else
#define CAH_RC_HD
#endif

CAH_RC_HD static inline uint8_t cah_complement(uint8_t c) {
    const uint8_t low = c & 0x20u;                          // ASCII letters differ in bit 5 only
    uint8_t u = (uint8_t)(c & ~0x20u), v;
    switch (u) {
        case 'A': v = 'T'; break;  case 'T': v = 'A'; break;  case 'U': v = 'A'; break;
        case 'C': v = 'G'; break;  case 'G': v = 'C'; break;
        case 'M': v = 'K'; break;  case 'K': v = 'M'; break;
        case 'R': v = 'Y'; break;  case 'Y': v = 'R'; break;
        case 'V': v = 'B'; break;  case 'B': v = 'V'; break;
        case 'H': v = 'D'; break;  case 'D': v = 'H'; break;
        case 'W': case 'S': case 'N': v = u; break;
        default: return c;                                  // not a nucleotide code (this also keeps '[' vs '{' apart)
    }
    return (uint8_t)(v | low);
}
