"""Batch-aware pipeline stage: FASTQ chunks in, adapter-trimmed FASTQ out
(SURVEY.md section 8(f), rows 1 and 2 -- the callers and data formats either side of the
matching path).

The reference walks reads one at a time (reference src/cutadapt/pipeline.py:60-69 ->
modifiers.py:200-261 ``AdapterCutter`` -> adapters ``match_to`` -> ``match.trimmed(read)``),
getting record-aligned 4 MiB chunks from dnaio.read_chunks (runners.py:116-126, :306).  Here a
chunk is indexed once by the C++ scanner (csrc/fastq.cpp), its sequences are packed and matched
in ONE GPU batch call, and the trimmed records are written straight from the raw chunk; no
per-read Python objects are created.

Scope of this slice: single-end FASTQ, ``AdapterCutter`` with ``times=1`` and ``action='trim'``
(the reference's fast path, modifiers.py:118-119, :253-261) over single or multiple adapters.
"""
import ctypes as C
import gzip
import io
from typing import BinaryIO, Dict, Iterator, Optional, Tuple, Union

import numpy as np

from . import _lib
from .adapters import MultipleAdapters, SingleAdapter
from .sharding import MatchHistogram

DEFAULT_CHUNK_BYTES = 4 * 1024 * 1024     # reference runners.py:306 buffer_size


def _open_maybe_gz(path_or_file: Union[str, BinaryIO]) -> BinaryIO:
    if hasattr(path_or_file, "read"):
        return path_or_file  # type: ignore[return-value]
    f = open(path_or_file, "rb")
    magic = f.read(2)
    f.seek(0)
    if magic == b"\x1f\x8b":
        return gzip.GzipFile(fileobj=f)  # type: ignore[return-value]
    return f


class FastqChunk:
    """One record-aligned chunk: the raw bytes plus rec[n,6] = (name_beg, name_end, seq_beg,
    seq_end, qual_beg, qual_end) byte offsets (csrc/fastq.cpp: cah_fastq_scan)."""

    def __init__(self, buf: np.ndarray, rec: np.ndarray):
        self.buf = buf
        self.rec = rec

    def __len__(self):
        return len(self.rec)

    def pack_sequences(self) -> Tuple[np.ndarray, np.ndarray]:
        n = len(self.rec)
        total = int((self.rec[:, 3] - self.rec[:, 2]).sum()) if n else 0
        seqs = np.empty(total, dtype=np.uint8)
        offsets = np.zeros(n + 1, dtype=np.int64)
        _lib.check(_lib.lib().cah_pack_sequences(
            self.buf.ctypes.data, self.rec.ctypes.data, n, seqs.ctypes.data if total else None,
            offsets.ctypes.data))
        return seqs, offsets

    def write_trimmed(self, keep_beg: np.ndarray, keep_end: np.ndarray, keep: Optional[np.ndarray] = None) -> bytes:
        n = len(self.rec)
        cap = int(len(self.buf)) + 4 * n + 16
        out = np.empty(cap, dtype=np.uint8)
        out_len = C.c_int64(0)
        kb = np.ascontiguousarray(keep_beg, dtype=np.int32)
        ke = np.ascontiguousarray(keep_end, dtype=np.int32)
        kp = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
        _lib.check(_lib.lib().cah_fastq_write_trimmed(
            self.buf.ctypes.data, self.rec.ctypes.data, n, kb.ctypes.data, ke.ctypes.data,
            kp.ctypes.data if kp is not None else None, out.ctypes.data, cap, C.byref(out_len)))
        return out[:out_len.value].tobytes()


def read_fastq_chunks(path_or_file: Union[str, BinaryIO], chunk_bytes: int = DEFAULT_CHUNK_BYTES) -> Iterator[FastqChunk]:
    """Record-aligned chunks of a (possibly gzip-compressed) FASTQ file: the job dnaio.read_chunks
    does for the reference's ReaderProcess (runners.py:116-126).  A partial record at the end of a
    buffer is carried over to the next one.  Each chunk gets its own buffer (read straight into it;
    only the short carried tail is copied)."""
    f = _open_maybe_gz(path_or_file)
    L = _lib.lib()
    carry = b""
    while True:
        buf = np.empty(len(carry) + chunk_bytes, dtype=np.uint8)
        if carry:
            buf[:len(carry)] = np.frombuffer(carry, dtype=np.uint8)
        got = f.readinto(memoryview(buf)[len(carry):]) if hasattr(f, "readinto") else None
        if got is None:                                   # file objects without readinto
            block = f.read(chunk_bytes)
            got = len(block)
            buf[len(carry):len(carry) + got] = np.frombuffer(block, dtype=np.uint8)
        total = len(carry) + got
        final = got == 0
        if total == 0:
            break
        data = buf[:total]
        max_rec = int(np.count_nonzero(data == 10)) // 4 + 2
        rec = np.empty((max_rec, 6), dtype=np.int64)
        n = C.c_int64(0)
        consumed = C.c_int64(0)
        _lib.check(L.cah_fastq_scan(data.ctypes.data, total, int(final), max_rec, rec.ctypes.data,
                                    C.byref(n), C.byref(consumed)))
        carry = data[consumed.value:].tobytes()
        if n.value:
            yield FastqChunk(data[:consumed.value], rec[:n.value])
        if final:
            break
        if not n.value and len(carry) > 64 * chunk_bytes:
            raise ValueError("FASTQ record larger than 64 chunks: not a FASTQ file?")


class BatchAdapterCutter:
    """``AdapterCutter(adapters, times=1, action='trim')`` over whole chunks
    (reference modifiers.py:82-261, fast path :253-261): best match of all adapters per read,
    5' matches keep read[rstop:], 3' matches keep read[:rstart] (adapters.py:453-454, :486-487)."""

    def __init__(self, adapters, device=None):
        if isinstance(adapters, MultipleAdapters):
            self.adapters = adapters
        else:
            adapters = list(adapters) if not isinstance(adapters, SingleAdapter) else [adapters]
            self.adapters = MultipleAdapters(adapters)
        self.device = device
        self.histogram = MatchHistogram(len(self.adapters))
        self.reads = 0
        self.with_adapters = 0
        self.bp_in = 0
        self.bp_out = 0

    def cut_intervals(self, seqs: np.ndarray, offsets: np.ndarray):
        """-> (keep_beg, keep_end, BatchMatches) for packed reads"""
        from .batch import ReadBatch
        n = len(offsets) - 1
        lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
        if n == 0:
            z = np.zeros(0, dtype=np.int32)
            return z, z, None
        batch = ReadBatch.from_host(seqs, offsets, device=self.device)
        batch.validate_ascii()
        bm = self.adapters.match_to_batch(batch)
        beg = np.zeros(n, dtype=np.int64)
        end = lens.copy()
        f = bm.found
        before = f & bm.remove_before
        after = f & ~bm.remove_before
        beg[before] = bm.coords[before, 3]          # rstop
        end[after] = bm.coords[after, 2]            # rstart
        self.histogram.add_batch(bm.coords, f, bm.adapter_index)
        self.reads += n
        self.with_adapters += int(f.sum())
        self.bp_in += int(lens.sum())
        self.bp_out += int((end - beg).sum())
        return beg.astype(np.int32), end.astype(np.int32), bm

    def process_chunk(self, chunk: FastqChunk) -> bytes:
        seqs, offsets = chunk.pack_sequences()
        beg, end, _ = self.cut_intervals(seqs, offsets)
        return chunk.write_trimmed(beg, end)


def trim_fastq(inpath: Union[str, BinaryIO], outpath: Union[str, BinaryIO], adapters,
               chunk_bytes: int = DEFAULT_CHUNK_BYTES, device=None) -> Dict[str, int]:
    """``cutadapt <adapter options> -o outpath inpath`` for the supported slice; returns the
    read/basepair counters the reference reports (reference report.py:62-80)."""
    cutter = BatchAdapterCutter(adapters, device=device)
    out = outpath if hasattr(outpath, "write") else open(outpath, "wb")
    try:
        for chunk in read_fastq_chunks(inpath, chunk_bytes):
            out.write(cutter.process_chunk(chunk))
    finally:
        if out is not outpath:
            out.close()
    return {"reads": cutter.reads, "with_adapters": cutter.with_adapters,
            "bp_in": cutter.bp_in, "bp_out": cutter.bp_out}
