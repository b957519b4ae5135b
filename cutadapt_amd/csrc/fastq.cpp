// fastq.cpp -- host-side FASTQ chunk indexing, sequence packing and trimmed output
// (the data formats either side of the matching path).
//
// The reference gets record-aligned 4 MiB chunks from the third-party dnaio.read_chunks
// (reference src/cutadapt/runners.py:116-126, :306), parses them into SequenceRecord objects
// (files.py:108-114), trims Python strings (adapters.py:453-454, :486-487) and formats the output
// through dnaio again.  Here a chunk is scanned once into offset arrays, the sequence lines are
// packed back to back for the GPU (the layout cah_match_batch takes), and the trimmed records are
// emitted straight from the raw chunk -- no per-read objects.
//
// Plain C++ (no HIP): these entry points work without a GPU.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/cutadapt_hip.h"

extern int cah_set_error_(int code, const char* msg);   // api.cpp

namespace {
inline const uint8_t* find_nl(const uint8_t* p, const uint8_t* end) {
    return static_cast<const uint8_t*>(memchr(p, '\n', (size_t)(end - p)));
}
}  // namespace

extern "C" {

// Scans complete 4-line records.  rec[i*6 .. i*6+5] = (name_beg, name_end, seq_beg, seq_end,
// qual_beg, qual_end) as byte offsets into buf; name excludes the leading '@', line ends exclude
// "\n" / "\r\n".  *consumed = bytes covered by the complete records (the caller carries the rest
// over to the next chunk, like dnaio.read_chunks does).  With is_final != 0 a last record without
// trailing newline is accepted and leftover bytes are an error.
int cah_fastq_scan(const uint8_t* buf, int64_t len, int is_final, int64_t max_records, int64_t* rec,
                   int64_t* n_records, int64_t* consumed) {
    if (!n_records || !consumed || (len > 0 && !buf) || (max_records > 0 && !rec))
        return cah_set_error_(CAH_EINVAL, "cah_fastq_scan: NULL argument");
    const uint8_t* const end = buf + len;
    const uint8_t* p = buf;
    int64_t n = 0;
    *n_records = 0;
    *consumed = 0;
    while (p < end && n < max_records) {
        const uint8_t* line[4];
        const uint8_t* line_end[4];
        const uint8_t* q = p;
        bool complete = true;
        for (int l = 0; l < 4; l++) {
            if (q >= end && !(is_final && l == 3 && q == end)) { complete = false; break; }
            const uint8_t* nl = q < end ? find_nl(q, end) : nullptr;
            line[l] = q;
            if (nl) {
                line_end[l] = (nl > q && nl[-1] == '\r') ? nl - 1 : nl;
                q = nl + 1;
            } else if (is_final && l == 3) {           // last line of the file without newline
                line_end[l] = (end > q && end[-1] == '\r') ? end - 1 : end;
                q = end;
            } else {
                complete = false;
                break;
            }
        }
        if (!complete) break;
        if (line[0] == line_end[0] || *line[0] != '@') {
            char msg[160];
            snprintf(msg, sizeof(msg), "FASTQ format error in record %lld: line expected to start with '@'",
                     (long long)n);
            return cah_set_error_(CAH_EINVAL, msg);
        }
        if (line[2] == line_end[2] || *line[2] != '+') {
            char msg[160];
            snprintf(msg, sizeof(msg), "FASTQ format error in record %lld: third line expected to start with '+'",
                     (long long)n);
            return cah_set_error_(CAH_EINVAL, msg);
        }
        if (line_end[1] - line[1] != line_end[3] - line[3]) {
            char msg[200];
            snprintf(msg, sizeof(msg),
                     "FASTQ format error in record %lld: length of sequence and qualities differ (%lld vs %lld)",
                     (long long)n, (long long)(line_end[1] - line[1]), (long long)(line_end[3] - line[3]));
            return cah_set_error_(CAH_EINVAL, msg);
        }
        int64_t* r = rec + n * 6;
        r[0] = (line[0] + 1) - buf; r[1] = line_end[0] - buf;
        r[2] = line[1] - buf;       r[3] = line_end[1] - buf;
        r[4] = line[3] - buf;       r[5] = line_end[3] - buf;
        n++;
        p = q;
    }
    if (is_final && p < end && n < max_records)
        return cah_set_error_(CAH_EINVAL, "FASTQ format error: premature end of file (incomplete record)");
    *n_records = n;
    *consumed = p - buf;
    return CAH_OK;
}

// Packs the sequence lines back to back: out_seqs (at least sum of lengths bytes) and
// out_offsets[n+1] -- the layout cah_match_batch consumes.
int cah_pack_sequences(const uint8_t* buf, const int64_t* rec, int64_t n_records, uint8_t* out_seqs,
                       int64_t* out_offsets) {
    if (n_records < 0 || !out_offsets || (n_records > 0 && (!buf || !rec)))
        return cah_set_error_(CAH_EINVAL, "cah_pack_sequences: bad argument");
    int64_t pos = 0;
    for (int64_t i = 0; i < n_records; i++) {
        const int64_t b = rec[i * 6 + 2], e = rec[i * 6 + 3];
        out_offsets[i] = pos;
        if (e > b) {
            if (!out_seqs) return cah_set_error_(CAH_EINVAL, "cah_pack_sequences: out_seqs is NULL");
            memcpy(out_seqs + pos, buf + b, (size_t)(e - b));
        }
        pos += e - b;
    }
    out_offsets[n_records] = pos;
    return CAH_OK;
}

// Emits "@name\nSEQ[keep_beg:keep_end]\n+\nQUAL[keep_beg:keep_end]\n" for every record with
// keep[i] != 0 (trimming = slicing, reference adapters.py:453-454 / :486-487; the second header
// is written as a bare '+', as dnaio does).  Returns CAH_ENOMEM if out_cap is too small (an upper
// bound is the input length + 4 * n_records).
int cah_fastq_write_trimmed(const uint8_t* buf, const int64_t* rec, int64_t n_records,
                            const int32_t* keep_beg, const int32_t* keep_end, const uint8_t* keep,
                            uint8_t* out, int64_t out_cap, int64_t* out_len) {
    if (!out_len || (n_records > 0 && (!buf || !rec || !keep_beg || !keep_end || !out)))
        return cah_set_error_(CAH_EINVAL, "cah_fastq_write_trimmed: NULL argument");
    int64_t pos = 0;
    for (int64_t i = 0; i < n_records; i++) {
        if (keep && !keep[i]) continue;
        const int64_t* r = rec + i * 6;
        const int64_t name_len = r[1] - r[0], seq_len = r[3] - r[2];
        int64_t a = keep_beg[i], b = keep_end[i];
        if (a < 0) a = 0;
        if (b > seq_len) b = seq_len;
        if (b < a) b = a;
        const int64_t need = 1 + name_len + 1 + (b - a) + 1 + 2 + (b - a) + 1;
        if (pos + need > out_cap) return cah_set_error_(CAH_ENOMEM, "cah_fastq_write_trimmed: output buffer too small");
        out[pos++] = '@';
        memcpy(out + pos, buf + r[0], (size_t)name_len); pos += name_len;
        out[pos++] = '\n';
        memcpy(out + pos, buf + r[2] + a, (size_t)(b - a)); pos += b - a;
        out[pos++] = '\n'; out[pos++] = '+'; out[pos++] = '\n';
        memcpy(out + pos, buf + r[4] + a, (size_t)(b - a)); pos += b - a;
        out[pos++] = '\n';
    }
    *out_len = pos;
    return CAH_OK;
}

}  // extern "C"
