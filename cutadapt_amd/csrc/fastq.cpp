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
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "revcomp.h"
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
            if (rec[i * 6 + 4] >= 0) {                     // FASTQ: one sequence line
                memcpy(out_seqs + pos, buf + b, (size_t)(e - b));
                pos += e - b;
            } else {                                       // FASTA: the sequence may span several lines
                for (int64_t q = b; q < e; q++) {
                    const uint8_t c = buf[q];
                    if (c != '\n' && c != '\r') out_seqs[pos++] = c;
                }
            }
        }
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
    // Records that are kept whole and already have the output's form ("@name\nSEQ\n+\nQUAL\n": line feeds only, a
    // bare '+' line) are copied as they are, and neighbours among them as ONE run: with a 3' adapter in a quarter
    // of the reads, a run is about four records -- one memcpy of a kilobyte instead of a dozen small ones.
    int64_t run_beg = -1, run_end = -1;                          // pending run [run_beg, run_end) of buf
    auto flush = [&]() -> bool {
        if (run_beg < 0) return true;
        const int64_t len = run_end - run_beg;
        if (pos + len > out_cap) return false;
        memcpy(out + pos, buf + run_beg, (size_t)len);
        pos += len;
        run_beg = -1;
        return true;
    };
    for (int64_t i = 0; i < n_records; i++) {
        if (keep && !keep[i]) continue;
        const int64_t* r = rec + i * 6;
        const int64_t name_len = r[1] - r[0], seq_len = r[3] - r[2];
        int64_t a = keep_beg[i], b = keep_end[i];
        if (a < 0) a = 0;
        if (b > seq_len) b = seq_len;
        if (b < a) b = a;
        // whole and canonical?  (r[5] is followed by a line feed inside the record's chunk unless it ends the data:
        // the caller's buffer ends there, so the last record of a file without final line feed goes piecewise)
        if (a == 0 && b == seq_len && r[2] == r[1] + 1 && r[4] == r[3] + 3 && r[5] - r[4] == seq_len &&
            buf[r[3] + 1] == '+' && (i + 1 < n_records ? rec[(i + 1) * 6] == r[5] + 2 : false)) {
            const int64_t beg = r[0] - 1, end = r[5] + 1;        // '@' .. the line feed after the qualities
            if (run_beg >= 0 && run_end == beg) { run_end = end; continue; }
            if (!flush()) return cah_set_error_(CAH_ENOMEM, "cah_fastq_write_trimmed: output buffer too small");
            run_beg = beg; run_end = end;
            continue;
        }
        if (!flush()) return cah_set_error_(CAH_ENOMEM, "cah_fastq_write_trimmed: output buffer too small");
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
    if (!flush()) return cah_set_error_(CAH_ENOMEM, "cah_fastq_write_trimmed: output buffer too small");
    *out_len = pos;
    return CAH_OK;
}

// FASTA twin of cah_fastq_scan: records are ">name" + one or more sequence lines.  rec has the same
// six columns; seq_beg..seq_end spans the raw sequence lines (line breaks included,
// cah_pack_sequences strips them) and qual_beg = qual_end = -1 marks "no qualities".  A record is
// complete when the next '>' line (or, with is_final, the end of the data) has been seen.
int cah_fasta_scan(const uint8_t* buf, int64_t len, int is_final, int64_t max_records, int64_t* rec,
                   int64_t* n_records, int64_t* consumed) {
    if (!n_records || !consumed || (len > 0 && !buf) || (max_records > 0 && !rec))
        return cah_set_error_(CAH_EINVAL, "cah_fasta_scan: NULL argument");
    const uint8_t* const end = buf + len;
    const uint8_t* p = buf;
    int64_t n = 0;
    *n_records = 0;
    *consumed = 0;
    while (p < end && (*p == '\n' || *p == '\r')) p++;          // blank lines before a header
    while (p < end && n < max_records) {
        if (*p != '>') {
            char msg[160];
            snprintf(msg, sizeof(msg), "FASTA format error in record %lld: expected '>' at the start of a record",
                     (long long)n);
            return cah_set_error_(CAH_EINVAL, msg);
        }
        const uint8_t* nl = find_nl(p, end);
        if (!nl && !is_final) break;                            // header line incomplete
        const uint8_t* name_end = nl ? ((nl > p && nl[-1] == '\r') ? nl - 1 : nl) : end;
        const uint8_t* seq_beg = nl ? nl + 1 : end;
        // the sequence runs to the next line that starts with '>'
        const uint8_t* q = seq_beg;
        bool closed = false;
        while (q < end) {
            if (*q == '>') { closed = true; break; }
            const uint8_t* l = find_nl(q, end);
            if (!l) { q = end; break; }
            q = l + 1;
        }
        if (!closed && !is_final) break;                        // may continue in the next chunk
        const uint8_t* seq_end = q;
        while (seq_end > seq_beg && (seq_end[-1] == '\n' || seq_end[-1] == '\r')) seq_end--;
        int64_t* r = rec + n * 6;
        r[0] = (p + 1) - buf; r[1] = name_end - buf;
        r[2] = seq_beg - buf; r[3] = seq_end - buf;
        r[4] = -1;            r[5] = -1;
        n++;
        p = q;
    }
    *n_records = n;
    *consumed = p - buf;
    return CAH_OK;
}

// Output formatting for every AdapterCutter action (reference modifiers.py:170-198, :236-251).
// The sequence comes from the packed buffer (seqs/offsets, the batch the GPU matched), names and
// qualities from the raw chunk.  For record i with keep[i] != 0 (keep NULL = all) and interval
// [beg[i], end[i]) relative to the read:
//   mode CAH_WRITE_SLICE      sequence and qualities sliced to the interval (trim / retain / crop;
//                             the whole read for action None)
//   mode CAH_WRITE_MASK       bases outside the interval replaced by 'N', qualities unchanged
//   mode CAH_WRITE_LOWERCASE  bases outside the interval lower-cased, inside upper-cased
// FASTQ records are written as "@name\nSEQ\n+\nQUAL\n", FASTA records (qual_beg < 0) as ">name\nSEQ\n".
int cah_records_write(const uint8_t* buf, const int64_t* rec, int64_t n_records, const uint8_t* seqs,
                      const int64_t* offsets, const int32_t* beg, const int32_t* end, const uint8_t* keep,
                      int mode, uint8_t* out, int64_t out_cap, int64_t* out_len) {
    if (!out_len || (n_records > 0 && (!buf || !rec || !offsets || !beg || !end || !out)))
        return cah_set_error_(CAH_EINVAL, "cah_records_write: NULL argument");
    if (mode < CAH_WRITE_SLICE || mode > CAH_WRITE_LOWERCASE)
        return cah_set_error_(CAH_EINVAL, "cah_records_write: unknown mode");
    int64_t pos = 0;
    for (int64_t i = 0; i < n_records; i++) {
        if (keep && !keep[i]) continue;
        const int64_t* r = rec + i * 6;
        const bool fastq = r[4] >= 0;
        const int64_t name_len = r[1] - r[0], seq_len = offsets[i + 1] - offsets[i];
        int64_t a = beg[i], b = end[i];
        if (a < 0) a = 0;
        if (b > seq_len) b = seq_len;
        if (b < a) b = a;
        const int64_t body = mode == CAH_WRITE_SLICE ? b - a : seq_len;
        const int64_t need = 1 + name_len + 1 + body + 1 + (fastq ? 2 + body + 1 : 0);
        if (pos + need > out_cap) return cah_set_error_(CAH_ENOMEM, "cah_records_write: output buffer too small");
        const uint8_t* s = seqs + offsets[i];
        out[pos++] = fastq ? '@' : '>';
        memcpy(out + pos, buf + r[0], (size_t)name_len); pos += name_len;
        out[pos++] = '\n';
        if (mode == CAH_WRITE_SLICE) {
            memcpy(out + pos, s + a, (size_t)(b - a)); pos += b - a;
        } else if (mode == CAH_WRITE_MASK) {
            memset(out + pos, 'N', (size_t)a);
            memcpy(out + pos + a, s + a, (size_t)(b - a));
            memset(out + pos + b, 'N', (size_t)(seq_len - b));
            pos += seq_len;
        } else {
            for (int64_t q = 0; q < seq_len; q++) {
                const uint8_t c = s[q];
                const bool inside = q >= a && q < b;
                out[pos + q] = inside ? ((c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c)
                                      : ((c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c);
            }
            pos += seq_len;
        }
        out[pos++] = '\n';
        if (fastq) {
            out[pos++] = '+'; out[pos++] = '\n';
            const int64_t qa = mode == CAH_WRITE_SLICE ? a : 0;
            memcpy(out + pos, buf + r[4] + qa, (size_t)body); pos += body;
            out[pos++] = '\n';
        }
    }
    *out_len = pos;
    return CAH_OK;
}

// Info-file rows (reference steps.py:232-253 + adapters.py:395-417, linked records :1157-1171).
// rows[k*7..] = (read, errors, rstart, rstop, wbeg, wend, name_idx), sorted by read and, within a
// read, in match order; rstart/rstop are relative to the read as it was when that match was made,
// i.e. to original[wbeg:wend].  Reads without a row get the "-1" line.  names = concatenated adapter
// names, name_off[n_names+1].  Every match row ends with an empty reverse-complement column.
static int info_write_impl(const uint8_t* buf, const int64_t* rec, int64_t n_records, const uint8_t* seqs,
                           const int64_t* offsets, const int64_t* rows, int64_t n_rows, const char* names,
                           const int64_t* name_off, int64_t n_names, const uint8_t* is_rc, const int32_t* final_beg,
                           const int32_t* final_end, uint8_t* out, int64_t out_cap, int64_t* out_len) {
    if ((final_beg == nullptr) != (final_end == nullptr))
        return cah_set_error_(CAH_EINVAL, "cah_info_write: final_beg and final_end go together");
    if (!out_len || (n_records > 0 && (!buf || !rec || !offsets || !out)) || (n_rows > 0 && (!rows || !names || !name_off)))
        return cah_set_error_(CAH_EINVAL, "cah_info_write: NULL argument");
    int64_t pos = 0, k = 0;
    auto put = [&](const uint8_t* src, int64_t len) { memcpy(out + pos, src, (size_t)len); pos += len; };
    auto put_int = [&](long long v) { pos += snprintf(reinterpret_cast<char*>(out + pos), 24, "%lld", v); };
    for (int64_t i = 0; i < n_records; i++) {
        const int64_t* r = rec + i * 6;
        const bool fastq = r[4] >= 0;
        const int64_t name_len = r[1] - r[0], seq_len = offsets[i + 1] - offsets[i];
        const uint8_t* s = seqs + offsets[i];
        const uint8_t* qual = fastq ? buf + r[4] : nullptr;
        if (k >= n_rows || rows[k * 7] != i) {
            if (k < n_rows && rows[k * 7] < i) return cah_set_error_(CAH_EINVAL, "cah_info_write: rows are not sorted by read");
            if (pos + name_len + 2 * seq_len + 16 > out_cap) return cah_set_error_(CAH_ENOMEM, "cah_info_write: output buffer too small");
            // no match: the read as the modifiers left it (steps.py:248-251)
            int64_t fa = final_beg ? final_beg[i] : 0, fb = final_end ? final_end[i] : seq_len;
            if (fa < 0) fa = 0;
            if (fb > seq_len) fb = seq_len;
            if (fb < fa) fb = fa;
            put(buf + r[0], name_len); out[pos++] = '\t'; out[pos++] = '-'; out[pos++] = '1'; out[pos++] = '\t';
            put(s + fa, fb - fa); out[pos++] = '\t';
            if (fastq) put(qual + fa, fb - fa);
            out[pos++] = '\n';
            continue;
        }
        for (; k < n_rows && rows[k * 7] == i; k++) {
            const int64_t* m = rows + k * 7;
            const int64_t wb = m[4], we = m[5], ni = m[6];
            if (wb < 0 || we > seq_len || wb > we || ni < 0 || ni >= n_names || m[2] < 0 || m[3] < m[2] || m[3] > we - wb)
                return cah_set_error_(CAH_EINVAL, "cah_info_write: row outside its read");
            const int64_t nlen = name_off[ni + 1] - name_off[ni];
            if (pos + name_len + nlen + 2 * (we - wb) + 96 > out_cap) return cah_set_error_(CAH_ENOMEM, "cah_info_write: output buffer too small");
            const int64_t a = wb + m[2], b = wb + m[3];
            put(buf + r[0], name_len); out[pos++] = '\t';
            put_int(m[1]); out[pos++] = '\t'; put_int(m[2]); out[pos++] = '\t'; put_int(m[3]); out[pos++] = '\t';
            put(s + wb, a - wb); out[pos++] = '\t'; put(s + a, b - a); out[pos++] = '\t'; put(s + b, we - b); out[pos++] = '\t';
            put(reinterpret_cast<const uint8_t*>(names) + name_off[ni], nlen); out[pos++] = '\t';
            if (fastq) { put(qual + wb, a - wb); out[pos++] = '\t'; put(qual + a, b - a); out[pos++] = '\t'; put(qual + b, we - b); }
            else { out[pos++] = '\t'; out[pos++] = '\t'; }
            out[pos++] = '\t';                                  // reverse-complement flag: "" (not searched), 0 or 1
            if (is_rc) out[pos++] = is_rc[i] ? '1' : '0';
            out[pos++] = '\n';
        }
    }
    if (k != n_rows) return cah_set_error_(CAH_EINVAL, "cah_info_write: rows refer to reads outside the chunk");
    *out_len = pos;
    return CAH_OK;
}

int cah_info_write(const uint8_t* buf, const int64_t* rec, int64_t n_records, const uint8_t* seqs,
                   const int64_t* offsets, const int64_t* rows, int64_t n_rows, const char* names,
                   const int64_t* name_off, int64_t n_names, uint8_t* out, int64_t out_cap, int64_t* out_len) {
    return info_write_impl(buf, rec, n_records, seqs, offsets, rows, n_rows, names, name_off, n_names, nullptr, nullptr,
                           nullptr, out, out_cap, out_len);
}

// ... with the last column filled in: is_rc[i] says whether record i was reverse-complemented (the reference's
// InfoFileWriter.RC_MAP, steps.py:224, :243: "" without --revcomp, else 0 or 1), and with the window
// [final_beg[i], final_end[i]) that the other modifiers left of a read WITHOUT a match (its "-1" line shows the read
// as it is written, steps.py:248-251; match rows show the read as it came in, :233-247).  Each may be NULL.
int cah_info_write_rc(const uint8_t* buf, const int64_t* rec, int64_t n_records, const uint8_t* seqs,
                      const int64_t* offsets, const int64_t* rows, int64_t n_rows, const char* names,
                      const int64_t* name_off, int64_t n_names, const uint8_t* is_rc, const int32_t* final_beg,
                      const int32_t* final_end, uint8_t* out, int64_t out_cap, int64_t* out_len) {
    return info_write_impl(buf, rec, n_records, seqs, offsets, rows, n_rows, names, name_off, n_names, is_rc, final_beg,
                           final_end, out, out_cap, out_len);
}

// The chunk after ReverseComplementer (reference modifiers.py:264-308): a second raw buffer with one normalised
// record per input record ("@name\nSEQ\n+\nQUAL\n" or ">name\nSEQ\n") and its record table, in which the records
// with is_rc[i] != 0 hold the reverse complement of the sequence (revcomp.h), the reversed qualities and the name
// followed by `suffix` (" rc"; suffix_len 0: none).  Every writer then works on it unchanged.  seqs/offsets: the
// packed sequences of the input chunk (cah_pack_sequences).  out_rec: int64[n_records*6].
int cah_chunk_revcomp(const uint8_t* buf, const int64_t* rec, int64_t n_records, const uint8_t* seqs,
                      const int64_t* offsets, const uint8_t* is_rc, const char* suffix, int64_t suffix_len,
                      uint8_t* out, int64_t out_cap, int64_t* out_rec, int64_t* out_len) {
    if (!out_len || suffix_len < 0 || (suffix_len > 0 && !suffix)
        || (n_records > 0 && (!buf || !rec || !offsets || !is_rc || !out || !out_rec)))
        return cah_set_error_(CAH_EINVAL, "cah_chunk_revcomp: NULL argument");
    int64_t pos = 0;
    for (int64_t i = 0; i < n_records; i++) {
        const int64_t* r = rec + i * 6;
        int64_t* w = out_rec + i * 6;
        const bool fastq = r[4] >= 0, flip = is_rc[i] != 0;
        const int64_t name_len = r[1] - r[0], n = offsets[i + 1] - offsets[i];
        const int64_t need = 1 + name_len + (flip ? suffix_len : 0) + 1 + n + 1 + (fastq ? 2 + n + 1 : 0);
        if (pos + need > out_cap) return cah_set_error_(CAH_ENOMEM, "cah_chunk_revcomp: output buffer too small");
        const uint8_t* s = seqs + offsets[i];
        out[pos++] = fastq ? '@' : '>';
        w[0] = pos;
        memcpy(out + pos, buf + r[0], (size_t)name_len); pos += name_len;
        if (flip && suffix_len) { memcpy(out + pos, suffix, (size_t)suffix_len); pos += suffix_len; }
        w[1] = pos;
        out[pos++] = '\n';
        w[2] = pos;
        if (flip) for (int64_t q = 0; q < n; q++) out[pos + q] = cah_complement(s[n - 1 - q]);
        else memcpy(out + pos, s, (size_t)n);
        pos += n;
        w[3] = pos;
        out[pos++] = '\n';
        w[4] = w[5] = -1;
        if (fastq) {
            out[pos++] = '+'; out[pos++] = '\n';
            w[4] = pos;
            const uint8_t* qual = buf + r[4];
            if (flip) for (int64_t q = 0; q < n; q++) out[pos + q] = qual[n - 1 - q];
            else memcpy(out + pos, qual, (size_t)n);
            pos += n;
            w[5] = pos;
            out[pos++] = '\n';
        }
    }
    *out_len = pos;
    return CAH_OK;
}

// The two chunks of a read pair as PairedReverseComplementer leaves them (reference modifiers.py:311-405: "R1 and R2
// swapped (equivalent to reverse complementing)"): record i of the output is record i of chunk B where swap[i] != 0
// -- with `suffix` behind its name -- and of chunk A otherwise, normalised like cah_chunk_revcomp's output.  Called
// once per output file with the roles of the chunks exchanged.
int cah_chunk_select(const uint8_t* buf_a, const int64_t* rec_a, const uint8_t* seqs_a, const int64_t* offsets_a,
                     const uint8_t* buf_b, const int64_t* rec_b, const uint8_t* seqs_b, const int64_t* offsets_b,
                     int64_t n_records, const uint8_t* swap, const char* suffix, int64_t suffix_len, uint8_t* out,
                     int64_t out_cap, int64_t* out_rec, int64_t* out_len) {
    if (!out_len || suffix_len < 0 || (suffix_len > 0 && !suffix)
        || (n_records > 0 && (!buf_a || !rec_a || !offsets_a || !buf_b || !rec_b || !offsets_b || !swap || !out || !out_rec)))
        return cah_set_error_(CAH_EINVAL, "cah_chunk_select: NULL argument");
    int64_t pos = 0;
    for (int64_t i = 0; i < n_records; i++) {
        const bool sw = swap[i] != 0;
        const uint8_t* buf = sw ? buf_b : buf_a;
        const int64_t* r = (sw ? rec_b : rec_a) + i * 6;
        const int64_t* offsets = sw ? offsets_b : offsets_a;
        const uint8_t* s = (sw ? seqs_b : seqs_a) + offsets[i];
        int64_t* w = out_rec + i * 6;
        const bool fastq = r[4] >= 0;
        const int64_t name_len = r[1] - r[0], n = offsets[i + 1] - offsets[i];
        const int64_t need = 1 + name_len + (sw ? suffix_len : 0) + 1 + n + 1 + (fastq ? 2 + n + 1 : 0);
        if (pos + need > out_cap) return cah_set_error_(CAH_ENOMEM, "cah_chunk_select: output buffer too small");
        out[pos++] = fastq ? '@' : '>';
        w[0] = pos;
        memcpy(out + pos, buf + r[0], (size_t)name_len); pos += name_len;
        if (sw && suffix_len) { memcpy(out + pos, suffix, (size_t)suffix_len); pos += suffix_len; }
        w[1] = pos;
        out[pos++] = '\n';
        w[2] = pos;
        memcpy(out + pos, s, (size_t)n); pos += n;
        w[3] = pos;
        out[pos++] = '\n';
        w[4] = w[5] = -1;
        if (fastq) {
            out[pos++] = '+'; out[pos++] = '\n';
            w[4] = pos;
            memcpy(out + pos, buf + r[4], (size_t)n); pos += n;
            w[5] = pos;
            out[pos++] = '\n';
        }
    }
    *out_len = pos;
    return CAH_OK;
}

// How many whole 4-line records (at most max_records) does buf hold, and where do they end?  Lines are counted, not
// parsed (memchr): what pairing two FASTQ streams needs -- the chunks handed on must hold the same number of records
// (the job of dnaio.read_paired_chunks, reference runners.py:104-113).  is_final: a last record without a final line
// feed counts.
static inline int64_t count_line_feeds(const uint8_t* p, int64_t len) {
    int64_t c = 0, i = 0;
#if defined(__SSE2__)
    // sixteen bytes per compare (SSE2 is part of x86-64): matches are 0xFF = -1 per byte, subtracted into sixteen byte
    // counters that are summed (psadbw) before they can overflow
    const __m128i lf = _mm_set1_epi8('\n'), zero = _mm_setzero_si128();
    while (i + 16 <= len) {
        __m128i acc = zero;
        const int64_t rounds = (len - i) / 16 < 255 ? (len - i) / 16 : 255;
        for (int64_t r = 0; r < rounds; r++, i += 16)
            acc = _mm_sub_epi8(acc, _mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + i)), lf));
        const __m128i sums = _mm_sad_epu8(acc, zero);
        c += _mm_cvtsi128_si64(sums) + _mm_cvtsi128_si64(_mm_srli_si128(sums, 8));
    }
#endif
    for (; i < len; i++) c += p[i] == '\n';
    return c;
}

int cah_fastq_span(const uint8_t* buf, int64_t len, int is_final, int64_t max_records, int64_t* n_records,
                   int64_t* consumed) {
    if (!n_records || !consumed || (len > 0 && !buf) || len < 0 || max_records < 0)
        return cah_set_error_(CAH_EINVAL, "cah_fastq_span: bad argument");
    *n_records = 0;
    *consumed = 0;
    if (max_records == 0 || len == 0) return CAH_OK;
    const int64_t want = max_records > (INT64_MAX >> 2) ? INT64_MAX : 4 * max_records;    // line feeds of max_records records
    int64_t lines = 0, pos = 0;
    while (pos < len) {                                                // blocks of 4 KiB: counted, not searched
        const int64_t blk = len - pos < 4096 ? len - pos : 4096;
        const int64_t c = count_line_feeds(buf + pos, blk);
        if (lines + c >= want) {                                       // the want-th line feed lies in this block
            int64_t need = want - lines;
            const uint8_t* q = buf + pos;
            while (need--) q = static_cast<const uint8_t*>(memchr(q, '\n', (size_t)(buf + len - q))) + 1;
            *n_records = max_records;
            *consumed = q - buf;
            return CAH_OK;
        }
        lines += c;
        pos += blk;
    }
    int64_t rec = lines / 4;
    const int rem = (int)(lines % 4);
    if (is_final && rem == 3 && buf[len - 1] != '\n') {                // the last record's last line has no line feed
        *n_records = rec + 1;
        *consumed = len;
        return CAH_OK;
    }
    // the records end behind the (4 rec)-th line feed = the (rem + 1)-th from the end
    const uint8_t* e = buf + len;
    for (int k = 0; k <= rem && rec > 0; k++) {
        const uint8_t* nl = static_cast<const uint8_t*>(memrchr(buf, '\n', (size_t)(e - buf)));
        e = (k == rem) ? nl + 1 : nl;
    }
    *n_records = rec;
    *consumed = rec > 0 ? e - buf : 0;
    return CAH_OK;
}

// Cheap record boundary for the threaded pipeline: the reader thread only has to cut the byte
// stream at a record start, the full scan (cah_fastq_scan / cah_fasta_scan) runs in a worker
// (dnaio.read_chunks does the same kind of backwards search for the reference's reader process,
// runners.py:116-126).  Returns in *cut the largest offset <= len that starts a record, 0 if none
// beyond the first was found.  FASTQ: a line starting with '@' whose next-but-one line starts
// with '+' (a quality line that happens to start with '@' is followed, two lines later, by a
// sequence line, never by '+').  FASTA: a line starting with '>'.
int cah_record_boundary(const uint8_t* buf, int64_t len, int is_fasta, int64_t* cut) {
    if (!cut || (len > 0 && !buf)) return cah_set_error_(CAH_EINVAL, "cah_record_boundary: NULL argument");
    *cut = 0;
    int64_t p = len;
    while (p > 0) {
        // start of the line that contains byte p-1
        const void* nl = memrchr(buf, '\n', (size_t)(p - 1));
        const int64_t ls = nl ? (const uint8_t*)nl - buf + 1 : 0;
        if (ls == 0) return CAH_OK;                         // the first record start is not a useful cut
        if (is_fasta) {
            if (buf[ls] == '>') { *cut = ls; return CAH_OK; }
        } else if (buf[ls] == '@') {
            const uint8_t* l1 = find_nl(buf + ls, buf + len);
            const uint8_t* l2 = l1 ? find_nl(l1 + 1, buf + len) : nullptr;
            if (l2 && l2 + 1 < buf + len && l2[1] == '+') { *cut = ls; return CAH_OK; }
        }
        p = ls;                                             // continue with the previous line
    }
    return CAH_OK;
}

}  // extern "C"
