"""Batch-aware pipeline stage: FASTA/FASTQ chunks in, adapter-processed records out
(SURVEY.md section 8(f), rows 1 and 2 -- the callers and data formats either side of the
matching path).

The reference walks reads one at a time (reference src/cutadapt/pipeline.py:60-69 ->
modifiers.py:200-261 ``AdapterCutter`` -> adapters ``match_to`` -> ``match.trimmed(read)``),
getting record-aligned 4 MiB chunks from dnaio.read_chunks (runners.py:116-126, :306).  Here a
chunk is indexed once by the C++ scanner (csrc/fastq.cpp), its sequences are packed and matched
in GPU batch calls -- one per search round -- and the output records, the info file and the
statistics are produced from arrays; no per-read Python objects are created.

Scope of this slice: single-end FASTA/FASTQ, ``AdapterCutter`` with every action
(trim / retain / crop / mask / lowercase / None, modifiers.py:170-198, :236-251), ``times``
rounds (:225-231), single, multiple and linked adapters, ``--discard-untrimmed`` /
``--discard-trimmed`` (steps.py DiscardUntrimmed/DiscardTrimmed), the ``--info-file`` rows
(steps.py:232-253) and the per-adapter ``errors[removed_length][errors]`` statistics
(adapters.py:185-199, :233-247).
"""
import ctypes as C
import gzip
from typing import BinaryIO, Dict, Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from .adapters import (AdapterIndex, BatchMatches, IndexedPrefixAdapters, IndexedSuffixAdapters, LinkedAdapter,
                       LinkedBatchMatches, MultipleAdapters, SingleAdapter, AnywhereAdapter, BackAdapter,
                       FrontAdapter, NonInternalBackAdapter, NonInternalFrontAdapter, PrefixAdapter,
                       RightmostBackAdapter, RightmostFrontAdapter, SuffixAdapter)
from .sharding import MatchHistogram

DEFAULT_CHUNK_BYTES = 4 * 1024 * 1024     # reference runners.py:306 buffer_size
ACTIONS = ("trim", "mask", "lowercase", "retain", "crop", None)


def _open_maybe_gz(path_or_file: Union[str, BinaryIO]) -> BinaryIO:
    if hasattr(path_or_file, "read"):
        return path_or_file  # type: ignore[return-value]
    f = open(path_or_file, "rb")
    magic = f.read(2)
    f.seek(0)
    if magic == b"\x1f\x8b":
        return gzip.GzipFile(fileobj=f)  # type: ignore[return-value]
    return f


class _BufferPool:
    """Recycles the large byte buffers of the threaded pipeline (input chunks, packed sequences,
    formatted output).  Fresh 32 MiB allocations are mmap + first-touch page faults, and page
    faults of threads of one process serialise in the kernel; recycled buffers cost nothing."""

    def __init__(self, max_free: int = 96):
        import threading
        self._free: List[np.ndarray] = []
        self._lock = threading.Lock()
        self._max_free = max_free

    def get(self, nbytes: int) -> np.ndarray:
        with self._lock:
            best = -1
            for i, a in enumerate(self._free):
                if len(a) >= nbytes and (best < 0 or len(a) < len(self._free[best])):
                    best = i
            if best >= 0:
                return self._free.pop(best)
        return np.empty(max(int(nbytes), 1), dtype=np.uint8)

    def put(self, arr) -> None:
        while isinstance(arr, np.ndarray) and arr.base is not None and isinstance(arr.base, np.ndarray):
            arr = arr.base                                  # views go back as their owning buffer
        if not isinstance(arr, np.ndarray) or arr.dtype != np.uint8 or arr.ndim != 1:
            return
        with self._lock:
            if len(self._free) < self._max_free:
                self._free.append(arr)


POOL = _BufferPool()


class FastqChunk:
    """One record-aligned chunk: the raw bytes plus rec[n,6] = (name_beg, name_end, seq_beg,
    seq_end, qual_beg, qual_end) byte offsets (csrc/fastq.cpp: cah_fastq_scan / cah_fasta_scan;
    FASTA records have qual_beg = qual_end = -1)."""

    def __init__(self, buf: np.ndarray, rec: np.ndarray):
        self.buf = buf
        self.rec = rec
        self._packed: Optional[Tuple[np.ndarray, np.ndarray]] = None
        self.pooled = False            # threaded pipeline: big buffers come from / go back to POOL
        self.derived: Optional["FastqChunk"] = None          # reverse_complemented(): lives and dies with this chunk

    def release(self) -> None:
        """hand the input and packed buffers back to the pool (the chunk is dead afterwards)"""
        if self.derived is not None:
            self.derived.release()
            self.derived = None
        if self.pooled:
            POOL.put(self.buf)
            if self._packed is not None:
                POOL.put(self._packed[0])
        self.buf = self._packed = None

    def __len__(self):
        return len(self.rec)

    def pack_sequences(self) -> Tuple[np.ndarray, np.ndarray]:
        if self._packed is None:
            n = len(self.rec)
            cap = int((self.rec[:, 3] - self.rec[:, 2]).sum()) if n else 0     # upper bound for FASTA
            seqs = POOL.get(cap) if self.pooled else np.empty(cap, dtype=np.uint8)
            offsets = np.zeros(n + 1, dtype=np.int64)
            _lib.check(_lib.lib().cah_pack_sequences(
                self.buf.ctypes.data, self.rec.ctypes.data, n, seqs.ctypes.data if cap else None,
                offsets.ctypes.data))
            self._packed = (seqs[:int(offsets[-1])], offsets)
        return self._packed

    # -- what the modifiers need of a chunk (gpu_pipeline.DeviceFastqChunk offers the same three, in place in HBM) --
    def lengths(self) -> np.ndarray:
        offsets = self.pack_sequences()[1]
        return (offsets[1:] - offsets[:-1]).astype(np.int64)

    def reads(self, device=None):
        """the reads of the chunk in HBM (a packed ReadBatch)"""
        from .batch import ReadBatch
        seqs, offsets = self.pack_sequences()
        return ReadBatch.from_host(seqs, offsets, device=device)

    def qualities(self, base):
        """-> (byte tensor on base's device, int64 start of every read's qualities in it or None = packed like
        ``base``); None for FASTA"""
        import torch
        q = self.pack_qualities()
        if q is None:
            return None
        return torch.from_numpy(q).to(base.device), None

    def pack_qualities(self) -> Optional[np.ndarray]:
        """quality lines packed with the same offsets as the sequences; None for FASTA"""
        n = len(self.rec)
        if n == 0 or self.rec[0, 4] < 0:
            return None
        seqs, offsets = self.pack_sequences()
        out = np.empty(len(seqs), dtype=np.uint8)
        idx = np.repeat(self.rec[:, 4] - offsets[:-1], offsets[1:] - offsets[:-1]) + np.arange(len(seqs))
        np.take(self.buf, idx, out=out)
        return out

    def write_trimmed(self, keep_beg: np.ndarray, keep_end: np.ndarray, keep: Optional[np.ndarray] = None) -> bytes:
        """FASTQ only, straight from the raw chunk (the times=1 / action='trim' fast path)."""
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

    def write_records(self, beg: np.ndarray, end: np.ndarray, keep: Optional[np.ndarray] = None,
                      mode: int = 0) -> bytes:
        """Any action, FASTA or FASTQ (cah_records_write): mode 0 slice, 1 mask, 2 lowercase."""
        n = len(self.rec)
        seqs, offsets = self.pack_sequences()
        cap = int(len(self.buf)) + 4 * n + 16
        out = POOL.get(cap) if self.pooled else np.empty(cap, dtype=np.uint8)
        out_len = C.c_int64(0)
        kb = np.ascontiguousarray(beg, dtype=np.int32)
        ke = np.ascontiguousarray(end, dtype=np.int32)
        kp = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
        _lib.check(_lib.lib().cah_records_write(
            self.buf.ctypes.data, self.rec.ctypes.data, n, seqs.ctypes.data if len(seqs) else None,
            offsets.ctypes.data, kb.ctypes.data, ke.ctypes.data,
            kp.ctypes.data if kp is not None else None, int(mode), out.ctypes.data, len(out), C.byref(out_len)))
        return memoryview(out)[:out_len.value]            # no copy; bytes-like (write(), ==, b"".join all take it)

    def reverse_complemented(self, is_rc: np.ndarray, suffix: Optional[str] = " rc") -> "FastqChunk":
        """The chunk as ReverseComplementer leaves it (reference modifiers.py:264-308): records with ``is_rc`` carry
        the reverse complement of the sequence, the reversed qualities and the name + ``suffix``
        (cah_chunk_revcomp).  The new chunk is released together with this one."""
        n = len(self.rec)
        seqs, offsets = self.pack_sequences()
        flags = np.ascontiguousarray(is_rc, dtype=np.uint8)
        sfx = (suffix or "").encode("ascii")
        names = int((self.rec[:, 1] - self.rec[:, 0]).sum()) if n else 0
        cap = names + 2 * int(len(seqs)) + (6 + len(sfx)) * n + 16
        out = POOL.get(cap) if self.pooled else np.empty(cap, dtype=np.uint8)
        rec = np.empty((n, 6), dtype=np.int64)
        out_len = C.c_int64(0)
        _lib.check(_lib.lib().cah_chunk_revcomp(
            self.buf.ctypes.data, self.rec.ctypes.data, n, seqs.ctypes.data if len(seqs) else None,
            offsets.ctypes.data, flags.ctypes.data, sfx, len(sfx), out.ctypes.data, len(out),
            rec.ctypes.data, C.byref(out_len)))
        chunk = FastqChunk(out[:out_len.value] if not self.pooled else out, rec)
        chunk.pooled = self.pooled
        self.derived = chunk
        return chunk

    def selected(self, other: "FastqChunk", swap: np.ndarray, suffix: Optional[str] = " rc") -> "FastqChunk":
        """One output chunk of PairedReverseComplementer (reference modifiers.py:311-405): record i of ``other`` where
        ``swap[i]`` (its name followed by ``suffix``), of this chunk otherwise (cah_chunk_select).  Released together
        with this chunk."""
        n = len(self.rec)
        if len(other.rec) != n:
            raise ValueError("Reads are improperly paired")
        sa, oa = self.pack_sequences()
        sb, ob = other.pack_sequences()
        flags = np.ascontiguousarray(swap, dtype=np.uint8)
        sfx = (suffix or "").encode("ascii")
        names = int(max((self.rec[:, 1] - self.rec[:, 0]).sum(), (other.rec[:, 1] - other.rec[:, 0]).sum())) if n else 0
        cap = 2 * names + 2 * (int(len(sa)) + int(len(sb))) + (6 + len(sfx)) * n + 16
        out = POOL.get(cap) if self.pooled else np.empty(cap, dtype=np.uint8)
        rec = np.empty((n, 6), dtype=np.int64)
        out_len = C.c_int64(0)
        _lib.check(_lib.lib().cah_chunk_select(
            self.buf.ctypes.data, self.rec.ctypes.data, sa.ctypes.data if len(sa) else None, oa.ctypes.data,
            other.buf.ctypes.data, other.rec.ctypes.data, sb.ctypes.data if len(sb) else None, ob.ctypes.data,
            n, flags.ctypes.data, sfx, len(sfx), out.ctypes.data, len(out), rec.ctypes.data, C.byref(out_len)))
        chunk = FastqChunk(out[:out_len.value] if not self.pooled else out, rec)
        chunk.pooled = self.pooled
        if self.derived is not None:
            self.derived.release()
        self.derived = chunk
        return chunk

    def write_info(self, rows: np.ndarray, names: Sequence[str], is_rc: Optional[np.ndarray] = None,
                   final: Optional[Tuple[np.ndarray, np.ndarray]] = None) -> bytes:
        """--info-file rows for this chunk; rows int64[k,7] as cah_info_write takes them.  ``is_rc``: per read, fills
        the reverse-complement column (--revcomp).  ``final`` = (beg, end): what is written of every read -- shown
        on the line of a read without a match."""
        n = len(self.rec)
        seqs, offsets = self.pack_sequences()
        rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1, 7)
        blob = "".join(names).encode("ascii")
        name_off = np.zeros(len(names) + 1, dtype=np.int64)
        np.cumsum([len(x) for x in names], out=name_off[1:])
        longest = max([len(x) for x in names], default=0)
        cap = 3 * int(len(self.buf)) * max(1, 1 + len(rows) // max(n, 1)) + (len(rows) + n) * (96 + longest) + 64
        out = np.empty(cap, dtype=np.uint8)
        out_len = C.c_int64(0)
        flags = None if is_rc is None else np.ascontiguousarray(is_rc, dtype=np.uint8)
        fb = fe = None
        if final is not None:
            fb, fe = (np.ascontiguousarray(x, dtype=np.int32) for x in final)
        _lib.check(_lib.lib().cah_info_write_rc(
            self.buf.ctypes.data, self.rec.ctypes.data, n, seqs.ctypes.data if len(seqs) else None,
            offsets.ctypes.data, rows.ctypes.data if len(rows) else None, len(rows), blob,
            name_off.ctypes.data, len(names), flags.ctypes.data if flags is not None else None,
            fb.ctypes.data if fb is not None else None, fe.ctypes.data if fe is not None else None,
            out.ctypes.data, cap, C.byref(out_len)))
        return out[:out_len.value].tobytes()


def read_fastq_chunks(path_or_file: Union[str, BinaryIO], chunk_bytes: int = DEFAULT_CHUNK_BYTES) -> Iterator[FastqChunk]:
    """Record-aligned chunks of a (possibly gzip-compressed) FASTA or FASTQ file: the job
    dnaio.read_chunks does for the reference's ReaderProcess (runners.py:116-126).  The format is
    taken from the first byte ('@' FASTQ, '>' FASTA).  A partial record at the end of a buffer is
    carried over to the next one.  Each chunk gets its own buffer (read straight into it; only the
    short carried tail is copied)."""
    f = _open_maybe_gz(path_or_file)
    L = _lib.lib()
    carry = b""
    scan = None
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
        if scan is None:
            scan = L.cah_fasta_scan if data[0] == ord(">") else L.cah_fastq_scan
        fasta = scan is L.cah_fasta_scan
        max_rec = (int(np.count_nonzero(data == ord(">"))) + 2) if fasta else int(np.count_nonzero(data == 10)) // 4 + 2
        rec = np.empty((max_rec, 6), dtype=np.int64)
        n = C.c_int64(0)
        consumed = C.c_int64(0)
        _lib.check(scan(data.ctypes.data, total, int(final), max_rec, rec.ctypes.data,
                        C.byref(n), C.byref(consumed)))
        carry = data[consumed.value:].tobytes()
        if n.value:
            yield FastqChunk(data[:consumed.value], rec[:n.value])
        if final:
            break
        if not n.value and len(carry) > 64 * chunk_bytes:
            raise ValueError("record larger than 64 chunks: not a FASTA/FASTQ file?")


def read_raw_chunks(path_or_file: Union[str, BinaryIO], chunk_bytes: int = DEFAULT_CHUNK_BYTES) -> Iterator[Tuple[np.ndarray, bool]]:
    """Record-aligned but UNPARSED chunks: (bytes, is_fasta).  The reader only looks for a record
    start near the end of each buffer (cah_record_boundary); parsing is left to the consumer
    (``scan_chunk``), so that several worker threads can do it."""
    f = _open_maybe_gz(path_or_file)
    L = _lib.lib()
    carry = b""
    fasta = None
    while True:
        buf = POOL.get(len(carry) + chunk_bytes)
        if carry:
            buf[:len(carry)] = np.frombuffer(carry, dtype=np.uint8)
        got = f.readinto(memoryview(buf)[len(carry):len(carry) + chunk_bytes]) if hasattr(f, "readinto") else None
        if got is None:
            block = f.read(chunk_bytes)
            got = len(block)
            buf[len(carry):len(carry) + got] = np.frombuffer(block, dtype=np.uint8)
        total = len(carry) + got
        if total == 0:
            POOL.put(buf)
            break
        data = buf[:total]
        if fasta is None:
            fasta = bool(data[0] == ord(">"))
        if got == 0:                                      # end of input: everything that is left
            yield data, fasta
            break
        cut = C.c_int64(0)
        _lib.check(L.cah_record_boundary(data.ctypes.data, total, int(fasta), C.byref(cut)))
        carry = data[cut.value:].tobytes()
        if cut.value:
            yield data[:cut.value], fasta
        elif len(carry) > 64 * chunk_bytes:
            raise ValueError("record larger than 64 chunks: not a FASTA/FASTQ file?")


def scan_chunk(data: np.ndarray, fasta: bool) -> FastqChunk:
    """Full record index of a record-aligned chunk (the whole chunk must parse)."""
    L = _lib.lib()
    total = len(data)
    max_rec = (int(np.count_nonzero(data == ord(">"))) + 2) if fasta else int(np.count_nonzero(data == 10)) // 4 + 2
    rec = np.empty((max_rec, 6), dtype=np.int64)
    n = C.c_int64(0)
    consumed = C.c_int64(0)
    _lib.check((L.cah_fasta_scan if fasta else L.cah_fastq_scan)(
        data.ctypes.data, total, 1, max_rec, rec.ctypes.data, C.byref(n), C.byref(consumed)))
    return FastqChunk(data, rec[:n.value])


# -------------------------------------------------------------------------------------------------
# one search round in array form
# -------------------------------------------------------------------------------------------------
class _Round:
    """Result of matching one Matchable unit against n (current) reads, reduced to what the
    pipeline needs: per read a found flag, score/errors for the best-of rule, the three
    intervals an action may use (all relative to the current read) and the info/statistics
    rows (k rows: read, errors, rstart, rstop, window offset inside the current read, name id,
    statistics slot, removed length)."""

    def __init__(self, n: int):
        z = lambda: np.zeros(n, dtype=np.int64)
        self.found = np.zeros(n, dtype=bool)
        self.score, self.errors = z(), z()
        self.rem_beg, self.rem_end = z(), z()
        self.ret_beg, self.ret_end = z(), z()
        self.crop_beg, self.crop_end = z(), z()
        self.can_crop = np.ones(n, dtype=bool)
        # rows: columns (read, errors, rstart, rstop, rel_wbeg, rel_wend, name_id, stat_slot, removed_len, order,
        # remove_before)
        self.rows = np.zeros((0, 11), dtype=np.int64)

    def take_better(self, other: "_Round") -> None:
        """MultipleAdapters' rule (reference adapters.py:1278-1285): higher score, then fewer
        errors; the unit that comes first keeps ties."""
        better = other.found & (~self.found | (other.score > self.score)
                                | ((other.score == self.score) & (other.errors < self.errors)))
        if not better.any():
            return
        for name in ("score", "errors", "rem_beg", "rem_end", "ret_beg", "ret_end", "crop_beg", "crop_end", "can_crop"):
            getattr(self, name)[better] = getattr(other, name)[better]
        self.found |= better
        keep_own = ~better[self.rows[:, 0]] if len(self.rows) else np.zeros(0, dtype=bool)
        take = better[other.rows[:, 0]] if len(other.rows) else np.zeros(0, dtype=bool)
        self.rows = np.concatenate([self.rows[keep_own], other.rows[take]])


def _rows_for(idx: np.ndarray, errors, rstart, rstop, rel_wbeg, rel_wend, name_id, stat_slot, removed, order: int,
              before=0):
    k = len(idx)
    rows = np.empty((k, 11), dtype=np.int64)
    rows[:, 10] = before                                     # Match.trimmed() keeps what FOLLOWS the match
    rows[:, 0] = idx
    rows[:, 1] = errors
    rows[:, 2] = rstart
    rows[:, 3] = rstop
    rows[:, 4] = rel_wbeg
    rows[:, 5] = rel_wend
    rows[:, 6] = name_id
    rows[:, 7] = stat_slot
    rows[:, 8] = removed
    rows[:, 9] = order
    return rows


class BatchAdapterCutter:
    """``AdapterCutter(adapters, times, action, index)`` over whole chunks (reference
    modifiers.py:82-261).  With ``index=True`` (the reference default) two or more indexable
    anchored 5' (3') adapters are regrouped behind one ``IndexedPrefixAdapters``
    (``IndexedSuffixAdapters``) exactly as :124-141 does, which also moves them behind the
    remaining adapters.  In every round the best match of all adapters is taken per read
    (``MultipleAdapters``), the read shrinks to the remainder and only reads that matched go into
    the next round.  Results are intervals on the original read + info rows + statistics."""

    def __init__(self, adapters, times: int = 1, action: Optional[str] = "trim", index: bool = True, device=None):
        if isinstance(adapters, MultipleAdapters):
            adapters = list(adapters._adapters)
        elif isinstance(adapters, (SingleAdapter, LinkedAdapter)):
            adapters = [adapters]
        else:
            adapters = list(adapters)
        if action not in ACTIONS:
            raise ValueError(f"unknown action {action!r}")
        if action in ("retain", "crop") and times > 1:
            raise ValueError("'retain' and 'crop' cannot be combined with times > 1")
        groups: List[Tuple[str, list]] = [("plain", adapters)]
        if index:                                           # _regroup_into_indexed_adapters (:124-141)
            prefix = [a for a in adapters if isinstance(a, SingleAdapter) and AdapterIndex.is_acceptable(a, True)]
            suffix = [a for a in adapters if isinstance(a, SingleAdapter) and a not in prefix
                      and AdapterIndex.is_acceptable(a, False)]
            if len(prefix) > 1 or len(suffix) > 1:
                single = [a for a in adapters if a not in prefix and a not in suffix]
                groups = [("plain", single)]
                groups.append(("prefix", prefix) if len(prefix) > 1 else ("plain", prefix))
                groups.append(("suffix", suffix) if len(suffix) > 1 else ("plain", suffix))
                adapters = single + prefix + suffix
        self.all_adapters = adapters
        self.adapters = MultipleAdapters(adapters)          # the reference attribute (un-indexed view)
        self.times = int(times)
        self.action = action
        self.device = device
        # units: runs of consecutive fusable single adapters become one fused plan; anything else
        # (linked adapters, rightmost adapters) is matched on its own.  Unit order = adapter order,
        # so "first adapter wins ties" is preserved.
        self._units: List[Tuple[str, object, List[int]]] = []
        pos = 0
        for gkind, members in groups:
            ids = list(range(pos, pos + len(members)))
            pos += len(members)
            if gkind == "prefix":
                self._units.append(("fused", IndexedPrefixAdapters(members), ids))
                continue
            if gkind == "suffix":
                self._units.append(("fused", IndexedSuffixAdapters(members), ids))
                continue
            run: List[int] = []
            for i, a in zip(ids, members):
                if isinstance(a, SingleAdapter) and not a._reverse_reads:
                    run.append(i)
                    continue
                if run:
                    self._units.append(("fused", MultipleAdapters([adapters[j] for j in run]), run))
                    run = []
                self._units.append(("linked" if isinstance(a, LinkedAdapter) else "single", a, [i]))
            if run:
                self._units.append(("fused", MultipleAdapters([adapters[j] for j in run]), run))
        # info-file names and statistics slots: one slot per single adapter, two per linked one
        self.names: List[str] = []
        self._slot: Dict[Tuple[int, int], int] = {}
        self.stat_labels: List[Tuple[str, str]] = []
        for i, a in enumerate(adapters):
            if isinstance(a, LinkedAdapter):
                base = "none" if a.name is None else a.name
                for part, suffix in ((0, ";1"), (1, ";2")):
                    self._slot[(i, part)] = len(self.names)
                    self.names.append(base + suffix)
                    self.stat_labels.append((a.name, "front" if part == 0 else "back"))
            else:
                self._slot[(i, 0)] = len(self.names)
                self.names.append(a.name)
                self.stat_labels.append((a.name, "end"))
        self._slot_of_adapter = np.array([self._slot[(i, 0)] for i in range(len(adapters))], dtype=np.int64)
        self.histogram = MatchHistogram(len(self.names))
        self.reverse_complemented = np.zeros(len(self.names), dtype=np.int64)      # per statistics slot (modifiers.py:305)
        self.reads = 0
        self.with_adapters = 0
        self.bp_in = 0
        self.bp_out = 0

    # ---- one unit, one round ----------------------------------------------------------------
    def _single_round(self, bm: BatchMatches, lens: np.ndarray, adapter_ids: Sequence[int]) -> _Round:
        n = len(lens)
        r = _Round(n)
        f = bm.found
        c = bm.coords
        before = bm.remove_before
        r.found = f.copy()
        r.score = np.where(f, c[:, 4], 0)
        r.errors = np.where(f, c[:, 5], 0)
        rstart, rstop = c[:, 2], c[:, 3]
        r.rem_beg = np.where(before, rstop, 0)
        r.rem_end = np.where(before, lens, rstart)
        r.ret_beg = np.where(before, rstart, 0)
        r.ret_end = np.where(before, lens, rstop)
        r.crop_beg, r.crop_end = rstart.copy(), rstop.copy()
        idx = np.flatnonzero(f)
        ids = np.asarray(adapter_ids, dtype=np.int64)[bm.adapter_index[idx].astype(np.int64)]
        slots = self._slot_of_adapter[ids]
        removed = np.where(before[idx], rstop[idx], lens[idx] - rstart[idx])     # removed_sequence_length()
        r.rows = _rows_for(idx, c[idx, 5], rstart[idx], rstop[idx], 0, lens[idx], slots, slots, removed, 0, before[idx])
        return r

    def _linked_round(self, lm: LinkedBatchMatches, lens: np.ndarray, adapter_id: int) -> _Round:
        n = len(lens)
        r = _Round(n)
        ff, bf = lm.front.found & lm.found, lm.back.found & lm.found
        fc, bc = lm.front.coords, lm.back.coords
        off = np.where(ff, fc[:, 3], 0)                      # the back stage saw read[off:]
        r.found = lm.found.copy()
        r.score = np.where(ff, fc[:, 4], 0) + np.where(bf, bc[:, 4], 0)
        r.errors = np.where(ff, fc[:, 5], 0) + np.where(bf, bc[:, 5], 0)
        # remainder([front, back]) (reference adapters.py:1588-1602, :1139-1143)
        r.rem_beg = off
        r.rem_end = np.where(bf, off + bc[:, 2], lens)
        # retained_adapter_interval (:1145-1155)
        r.ret_beg = np.where(ff, fc[:, 2], 0)
        r.ret_end = np.where(bf, bc[:, 3] + off, lens)
        r.can_crop[:] = False                                # LinkedMatch has no rstart/rstop
        fi, bi = np.flatnonzero(ff), np.flatnonzero(bf)
        s1, s2 = self._slot[(adapter_id, 0)], self._slot[(adapter_id, 1)]
        rows1 = _rows_for(fi, fc[fi, 5], fc[fi, 2], fc[fi, 3], 0, lens[fi], s1, s1, fc[fi, 3], 0, 1)
        rows2 = _rows_for(bi, bc[bi, 5], bc[bi, 2], bc[bi, 3], off[bi], lens[bi], s2, s2,
                          (lens[bi] - off[bi]) - bc[bi, 2], 1)
        r.rows = np.concatenate([rows1, rows2])
        return r

    def _match_round(self, batch, lens: np.ndarray) -> _Round:
        best = None
        for kind, unit, ids in self._units:
            if kind == "linked":
                rr = self._linked_round(unit.match_to_batch(batch), lens, ids[0])
            else:
                rr = self._single_round(unit.match_to_batch(batch), lens, ids if kind == "fused" else [ids[0]])
            if best is None:
                best = rr
            else:
                best.take_better(rr)
        return best

    # ---- all rounds of a chunk --------------------------------------------------------------
    def process_arrays(self, seqs: np.ndarray, offsets: np.ndarray, base=None, window=None):
        """-> dict(beg, end, matched, rows): the interval of every read the chosen action keeps
        or marks (relative to the original read), whether any adapter was found, and the info
        rows int64[k,7] = (read, errors, rstart, rstop, wbeg, wend, name_id).
        ``base``: the chunk already in HBM (ReadBatch); ``window`` = (beg, end) arrays: the part of
        every read earlier modifiers (quality trimming ...) left, which is what gets searched."""
        from .batch import ReadBatch
        n = len(offsets) - 1
        lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
        if n and base is None:
            base = ReadBatch.from_host(seqs, offsets, device=self.device)
        return self.process_batch(base, lens, window)

    def process_batch(self, base, lens: np.ndarray, window=None):
        """process_arrays for reads that are in HBM already (``base``: any ReadBatch, packed or a view)"""
        return self.commit(self.search(base, lens, window), lens)

    def search(self, base, lens: np.ndarray, window=None) -> Dict[str, np.ndarray]:
        """``match_and_trim`` (reference modifiers.py:209-251) for every read of ``base``: all rounds, no statistics.
        -> beg/end/matched as process_arrays returns them, ``rows`` int64[k,9] (process_arrays' seven columns, the
        order of the match within its read, whether the match removes what is in front of it), ``stats`` int64[k,3] = (statistics slot, removed length, errors) of
        the same k matches and ``score``: the sum of the scores of a read's matches (what ReverseComplementer
        compares, modifiers.py:287-289)."""
        import torch
        from .batch import ReadBatch
        n = len(lens)
        if window is None:
            wbeg, wend = np.zeros(n, dtype=np.int64), lens.copy()
        else:
            wbeg, wend = np.asarray(window[0], dtype=np.int64).copy(), np.asarray(window[1], dtype=np.int64).copy()
        w0beg, w0end = wbeg.copy(), wend.copy()
        matched = np.zeros(n, dtype=bool)
        score = np.zeros(n, dtype=np.int64)
        last_ret = (w0beg.copy(), w0end.copy())
        last_crop = (w0beg.copy(), w0end.copy())
        all_rows, all_stats = [], []
        if n:
            base.validate_ascii()
            dev_off = base.offsets[:n]
            active = np.arange(n)
            for rnd in range(self.times):
                if rnd == 0 and window is None:
                    batch, cur_len = base, lens
                else:
                    ia = torch.from_numpy(active).to(base.device)
                    starts = dev_off[ia] + torch.from_numpy(wbeg[active]).to(base.device)
                    cur_len = wend[active] - wbeg[active]
                    batch = ReadBatch(base.seqs, starts, torch.from_numpy(cur_len.astype(np.int32)).to(base.device),
                                      n_reads=len(active), validated=True)
                    if rnd == 0 and base.uniform_len and base.lens is None:
                        batch.within_uniform = int(base.uniform_len)     # every read of a sequencer's batch, cut: streamed
                rr = self._match_round(batch, cur_len)
                hit = rr.found
                if not hit.any():
                    break
                g = active[hit]
                if self.action == "crop" and not rr.can_crop[hit].all():
                    raise AttributeError("'LinkedMatch' object has no attribute 'rstart'")     # as the reference
                rows = rr.rows
                if len(rows):
                    if len(rows) > 1 and not (rows[1:, 0] > rows[:-1, 0]).all():     # (one unit, one row per read: sorted as it is)
                        rows = rows[np.lexsort((rows[:, 9], rows[:, 0]))]
                    out = np.empty((len(rows), 9), dtype=np.int64)
                    gi = active[rows[:, 0]]
                    out[:, 8] = rows[:, 10]
                    out[:, 0] = gi
                    out[:, 1:4] = rows[:, 1:4]
                    out[:, 4] = wbeg[gi] + rows[:, 4]
                    out[:, 5] = wbeg[gi] + rows[:, 5]
                    out[:, 6] = rows[:, 6]
                    out[:, 7] = rnd * 2 + rows[:, 9]
                    all_rows.append(out)
                    all_stats.append(rows[:, [7, 8, 1]])
                score[g] += rr.score[hit]
                if self.action == "retain":
                    last_ret[0][g], last_ret[1][g] = wbeg[g] + rr.ret_beg[hit], wbeg[g] + rr.ret_end[hit]
                elif self.action == "crop":
                    last_crop[0][g], last_crop[1][g] = wbeg[g] + rr.crop_beg[hit], wbeg[g] + rr.crop_end[hit]
                new_beg = wbeg[g] + rr.rem_beg[hit]
                new_end = wbeg[g] + rr.rem_end[hit]
                wbeg[g], wend[g] = new_beg, new_end
                matched[g] = True
                active = g
        if self.action == "retain":
            beg, end = last_ret
        elif self.action == "crop":
            beg, end = last_crop
        elif self.action is None:
            beg, end = w0beg, w0end
        else:                                                # trim, mask, lowercase: the remainder
            beg, end = wbeg, wend
        rows = np.concatenate(all_rows) if all_rows else np.zeros((0, 9), dtype=np.int64)
        stats = np.concatenate(all_stats) if all_stats else np.zeros((0, 3), dtype=np.int64)
        if len(all_rows) > 1:                                # (one round: sorted as it is)
            order = np.lexsort((rows[:, 7], rows[:, 0]))
            rows, stats = rows[order], stats[order]
        return {"beg": beg, "end": end, "matched": matched, "rows": rows, "stats": stats, "score": score}

    def commit(self, found: Dict[str, np.ndarray], lens: np.ndarray, reverse_complemented=None):
        """statistics of one searched chunk (AdapterCutter.__call__, reference modifiers.py:199-206) + the result
        dict of process_arrays.  ``reverse_complemented``: bool per read, counted per adapter like
        ``stats.reverse_complemented`` (modifiers.py:305)."""
        beg, end, matched, rows, stats = (found[k] for k in ("beg", "end", "matched", "rows", "stats"))
        if len(stats):
            self.histogram.add_rows(stats[:, 0], stats[:, 1], stats[:, 2])
            if reverse_complemented is not None:
                np.add.at(self.reverse_complemented, stats[reverse_complemented[rows[:, 0]], 0], 1)
        self.reads += len(lens)
        self.with_adapters += int(matched.sum())
        self.bp_in += int(lens.sum())
        self.bp_out += int((end - beg).sum()) if self.action in ("trim", "retain", "crop") else int(lens.sum())
        return {"beg": beg.astype(np.int32), "end": end.astype(np.int32), "matched": matched,
                "rows": np.ascontiguousarray(rows[:, :7]), "before": rows[:, 8].astype(bool)}

    # ---- compatibility with the first slice (times=1, trim) ---------------------------------
    def cut_intervals(self, seqs: np.ndarray, offsets: np.ndarray):
        """-> (keep_beg, keep_end, matched) for packed reads"""
        res = self.process_arrays(seqs, offsets)
        return res["beg"], res["end"], res["matched"]

    def process_chunk(self, chunk: FastqChunk, discard_untrimmed: bool = False, discard_trimmed: bool = False,
                      info: Optional[list] = None) -> bytes:
        seqs, offsets = chunk.pack_sequences()
        res = self.process_arrays(seqs, offsets)
        if info is not None:
            info.append(chunk.write_info(res["rows"], self.names))
        keep = None
        if discard_untrimmed:
            keep = res["matched"]
        elif discard_trimmed:
            keep = ~res["matched"]
        mode = {"mask": 1, "lowercase": 2}.get(self.action, 0)
        return chunk.write_records(res["beg"], res["end"], keep, mode)


class BatchReverseComplementer:
    """``ReverseComplementer(adapter_cutter, rc_suffix)`` over whole chunks (reference modifiers.py:264-308): every
    read and its reverse complement go through the cutter's search (two passes of the batch matcher over two copies
    of the chunk in HBM, the second made by cah_revcomp_reads_batch); per read the reverse complement is taken when
    the scores of its matches add up to MORE than the forward read's (a tie keeps the read as it is)."""

    def __init__(self, adapter_cutter: BatchAdapterCutter, rc_suffix: Optional[str] = " rc"):
        self.adapter_cutter = adapter_cutter
        self.reverse_complemented = 0
        self._suffix = rc_suffix

    def process_arrays(self, seqs: np.ndarray, offsets: np.ndarray, base=None, window=None):
        """BatchAdapterCutter.process_arrays + ``rc``: bool per read.  Intervals and info rows of a reverse-complemented
        read are relative to the reverse complement of the read.  ``window`` is given on the forward read."""
        from .batch import ReadBatch
        n = len(offsets) - 1
        lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
        if n and base is None:
            base = ReadBatch.from_host(seqs, offsets, device=self.adapter_cutter.device)
        return self.process_batch(base, lens, window)

    def process_batch(self, base, lens: np.ndarray, window=None):
        from .adapters import _reverse_batch
        cutter = self.adapter_cutter
        n = len(lens)
        if n == 0:
            out = cutter.commit(cutter.search(None, lens, window), lens)
            out["rc"] = np.zeros(0, dtype=bool)
            return out
        fwd = cutter.search(base, lens, window)
        rbase = _reverse_batch(base, complement=True)
        rbase.validated = True                               # search(base) just validated the same bytes
        rwindow = None if window is None else (lens - np.asarray(window[1], dtype=np.int64),
                                               lens - np.asarray(window[0], dtype=np.int64))
        rev = cutter.search(rbase, lens, rwindow)
        use = rev["score"] > fwd["score"]
        merged = {k: np.where(use, rev[k], fwd[k]) for k in ("beg", "end", "matched")}
        keep_f = ~use[fwd["rows"][:, 0]]
        keep_r = use[rev["rows"][:, 0]]
        rows = np.concatenate([fwd["rows"][keep_f], rev["rows"][keep_r]])
        stats = np.concatenate([fwd["stats"][keep_f], rev["stats"][keep_r]])
        order = np.lexsort((rows[:, 7], rows[:, 0]))
        merged["rows"], merged["stats"] = rows[order], stats[order]
        self.reverse_complemented += int(use.sum())
        out = cutter.commit(merged, lens, reverse_complemented=use)
        out["rc"] = use
        return out


def _info_rows_on_original(rows: np.ndarray, before: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """Info rows as the reference prints them when other modifiers ran in front of the adapter step
    (steps.py:232-247): InfoFileWriter starts from the read AS IT CAME IN (``info.original_read``), cuts it at the
    coordinates of every match -- which were found on the read the earlier modifiers left -- and trims it the way the
    match trims (``match.trimmed(current_read)``) before the next match.  rows: (read, errors, rstart, rstop, wbeg,
    wend, name) sorted by read and match order; columns 4:6 are replaced by that window on the whole read."""
    rows = rows.copy()
    k = len(rows)
    reads = rows[:, 0]
    qb, qe = np.zeros(len(lens), dtype=np.int64), lens.astype(np.int64).copy()
    first = np.ones(k, dtype=bool)
    first[1:] = reads[1:] != reads[:-1]
    rank = np.arange(k) - np.maximum.accumulate(np.where(first, np.arange(k), 0))
    for j in range(int(rank.max()) + 1 if k else 0):
        sel = np.flatnonzero(rank == j)
        r = reads[sel]
        b, e = qb[r], qe[r]
        rows[sel, 4], rows[sel, 5] = b, e
        rows[sel, 3] = np.minimum(rows[sel, 3], e - b)
        rows[sel, 2] = np.minimum(rows[sel, 2], rows[sel, 3])
        front = before[sel].astype(bool)
        qb[r] = np.where(front, b + rows[sel, 3], b)
        qe[r] = np.where(front, e, b + rows[sel, 2])
    return rows


def _materialized(chunk: "FastqChunk", obeg, oend, ibeg, iend, mode: int) -> "FastqChunk":
    """The chunk with every record cut to [obeg, oend) and marked (mode 1 mask, 2 lowercase) outside [ibeg, iend)."""
    if hasattr(chunk, "host_chunk"):
        chunk = chunk.host_chunk()                          # (a device-indexed chunk: the host writers take over)
    n = len(chunk)
    fasta = bool(n and chunk.rec[0, 4] < 0)
    sliced = scan_chunk(np.frombuffer(bytes(chunk.write_records(obeg, oend, None, 0)), dtype=np.uint8).copy(), fasta)
    marked = sliced.write_records((ibeg - obeg), (iend - obeg), None, mode)
    return scan_chunk(np.frombuffer(bytes(marked), dtype=np.uint8).copy(), fasta)


class BatchTrimmer:
    """The read-modifying part of a single-end pipeline in the reference's order (cli.py:938-987):
    NextSeq trimming, quality trimming, adapter cutting, poly-A trimming -- all on windows
    (offset, length) into ONE copy of the chunk in HBM.  Statistics as the reference keeps them
    (``trimmed_bases`` of modifiers.py:829/845, PolyATrimmer's length histogram :865)."""

    def __init__(self, adapters=(), times: int = 1, action: Optional[str] = "trim", index: bool = True,
                 nextseq_trim: Optional[int] = None, quality_cutoff: Optional[Tuple[int, int]] = None,
                 quality_base: int = 33, poly_a: bool = False, max_expected_errors: Optional[float] = None,
                 cut: Sequence[int] = (), length: Optional[int] = None, device=None, poly_a_revcomp: bool = False,
                 revcomp: bool = False, rc_suffix: Optional[str] = " rc"):
        adapters = list(adapters._adapters) if isinstance(adapters, MultipleAdapters) else \
            ([adapters] if isinstance(adapters, (SingleAdapter, LinkedAdapter)) else list(adapters))
        self.cutter = BatchAdapterCutter(adapters, times=times, action=action, index=index, device=device) if adapters else None
        # --revcomp: ReverseComplementer takes the adapter cutter's place in the chain (reference cli.py:1113-1118)
        self.rc = BatchReverseComplementer(self.cutter, rc_suffix) if (revcomp and self.cutter is not None) else None
        self.action = action
        self.nextseq_trim = nextseq_trim
        self.quality_cutoff = quality_cutoff
        self.quality_base = quality_base
        self.poly_a = poly_a
        # the second mate of a pair loses a poly-T HEAD instead of a poly-A tail: the reference builds
        # (PolyATrimmer(), PolyATrimmer(revcomp=True)) for paired-end data (cli.py, modifiers.py:861-879)
        self.poly_a_revcomp = poly_a_revcomp
        self.max_expected_errors = max_expected_errors
        self.too_many_expected_errors = 0
        self.device = device
        self.cut = [int(c) for c in cut if int(c) != 0]       # UnconditionalCutter(s), modifiers.py:55-79
        if len(self.cut) > 2:
            raise ValueError("You cannot remove bases from more than two ends.")
        if len(self.cut) == 2 and self.cut[0] * self.cut[1] > 0:
            raise ValueError("You cannot remove bases from the same end twice.")
        self.length = length                                   # Shortener, modifiers.py:882-899
        self.filtered: Dict[str, int] = {}
        self.nextseq_trimmed_bases = 0
        self.quality_trimmed_bases = 0
        self.poly_a_trimmed_lengths: Dict[int, int] = {}
        self.reads = 0
        self.bp_in = 0
        self.bp_out = 0

    def merge(self, other: "BatchTrimmer") -> None:
        """add another worker's statistics (the reference sums Statistics objects, report.py:81-126)"""
        for name in ("nextseq_trimmed_bases", "quality_trimmed_bases", "too_many_expected_errors", "reads", "bp_in", "bp_out"):
            setattr(self, name, getattr(self, name) + getattr(other, name))
        for k, c in other.poly_a_trimmed_lengths.items():
            self.poly_a_trimmed_lengths[k] = self.poly_a_trimmed_lengths.get(k, 0) + c
        for k, c in other.filtered.items():
            self.filtered[k] = self.filtered.get(k, 0) + c
        if self.rc is not None and other.rc is not None:
            self.rc.reverse_complemented += other.rc.reverse_complemented
        if self.cutter is not None and other.cutter is not None:
            self.cutter.histogram += other.cutter.histogram
            self.cutter.reverse_complemented += other.cutter.reverse_complemented
            for name in ("reads", "with_adapters", "bp_in", "bp_out"):
                setattr(self.cutter, name, getattr(self.cutter, name) + getattr(other.cutter, name))

    def modify(self, chunk: FastqChunk, info: Optional[list] = None) -> Dict[str, object]:
        """All read-modifying steps for one chunk -> per-read arrays: the window (beg, end) of the
        original read that is left, whether an adapter was found, the output mode and (with
        --max-ee) the expected errors of what is left."""
        gen = self._modify_steps(chunk, info)
        request = next(gen)                                 # everything in front of the adapter step is done
        res = None
        if request is not None and self.cutter is not None:
            res = (self.rc or self.cutter).process_batch(request["base"], request["lens"], request["window"])
        try:
            gen.send(res)
        except StopIteration as stop:
            return stop.value
        raise RuntimeError("modify: the step generator did not finish")

    def _modify_steps(self, chunk: FastqChunk, info: Optional[list] = None):
        """modify() as a two-phase generator: runs the modifiers in front of the adapter step, yields what the
        adapter step needs (packed reads, the chunk in HBM, the window left so far), receives the adapter step's
        result (BatchAdapterCutter.process_arrays, or the paired cutter's for --pair-adapters, where both
        mates must reach this point before either is matched) and finishes with the remaining modifiers."""
        import torch
        from . import qualtrim as qt
        if self.rc is not None and hasattr(chunk, "host_chunk"):
            chunk = chunk.host_chunk()                      # the orientations are merged into a host-side chunk
        n = len(chunk)
        lens = chunk.lengths()
        bp_in = int(lens.sum())                              # of the reads as they came in (reference PairedEndPipeline.process_reads:
                                                             # len(read1) / len(read2) BEFORE any modifier, swapped mates or not)
        wbeg, wend = np.zeros(n, dtype=np.int64), lens.copy()
        pre = self.nextseq_trim is not None or self.quality_cutoff is not None or bool(self.cut)
        base = chunk.reads(self.device) if n else None
        quals = qoff = None

        def load_qualities(why: str):
            nonlocal quals, qoff
            if quals is None:
                q = chunk.qualities(base)
                if q is None:
                    raise qt.HasNoQualities(why)
                quals, qoff = q

        if (self.nextseq_trim is not None or self.quality_cutoff is not None) and n:
            load_qualities("Cannot do quality trimming when no qualities are available")

        def view_args():
            o = base.offsets[:n] + torch.from_numpy(wbeg).to(base.device)
            l = torch.from_numpy((wend - wbeg).astype(np.int32)).to(base.device)
            return o, l

        def qual_view(o):
            """where the current window's qualities start: packed like the reads, or at their own offsets"""
            return o if qoff is None else qoff[:n] + torch.from_numpy(wbeg).to(base.device)

        for c in self.cut:                                   # read[c:] / read[:c]
            cur = wend - wbeg
            if c > 0:
                wbeg = wbeg + np.minimum(c, cur)
            else:
                wend = wbeg + np.maximum(cur + c, 0)
        if self.nextseq_trim is not None and n:
            o, l = view_args()
            stop = qt.nextseq_trim_batch(base.seqs, quals, o, l, n, self.nextseq_trim, self.quality_base,
                                         qual_offsets=None if qoff is None else qual_view(o)).astype(np.int64)
            self.nextseq_trimmed_bases += int(((wend - wbeg) - stop).sum())
            wend = wbeg + stop
        if self.quality_cutoff is not None and n:
            o, l = view_args()
            ss = qt.quality_trim_batch(quals, qual_view(o), l, n, self.quality_cutoff[0], self.quality_cutoff[1],
                                       self.quality_base).astype(np.int64)
            self.quality_trimmed_bases += int(((wend - wbeg) - (ss[:, 1] - ss[:, 0])).sum())
            wbeg, wend = wbeg + ss[:, 0], wbeg + ss[:, 1]
        matched = np.zeros(n, dtype=bool)
        mode = 0
        res = yield {"base": base, "window": (wbeg, wend) if pre else None, "lens": lens, "n": n}
        out_chunk, is_rc = chunk, None
        info_rows, info_chunk, info_shift = np.zeros((0, 7), np.int64), None, 0
        if res is not None:
            w0beg, w0end = wbeg, wend
            wbeg, wend, matched = res["beg"].astype(np.int64), res["end"].astype(np.int64), res["matched"]
            mode = {"mask": 1, "lowercase": 2}.get(self.action, 0)
            is_rc = res.get("rc")
            if res.get("chunk") is not None:
                # the adapter step exchanged records (PairedReverseComplementer: the mates of some pairs are swapped):
                # from here on this mate IS that chunk
                out_chunk = res["chunk"]
                lens = out_chunk.lengths()
                if pre:
                    w0beg, w0end = res["w0"]
                if n and (self.poly_a or self.max_expected_errors is not None):
                    base = out_chunk.reads(self.device)
                quals = qoff = None
                is_rc_done = True
            else:
                is_rc_done = False
            if is_rc is not None and n and not is_rc_done:
                # from here on every read is the orientation that won: one merged copy in HBM for the modifiers
                # that follow, one merged chunk for the writers; the qualities turn around with their reads
                from .adapters import _reverse_batch
                sel = torch.from_numpy(is_rc).to(base.device)
                if quals is not None:
                    quals = _reverse_batch(base, select=sel, data=quals)
                if self.poly_a:
                    base = _reverse_batch(base, complement=True, select=sel)
                out_chunk = chunk.reverse_complemented(is_rc, self.rc._suffix)
                if pre:                                      # the window the search saw, in the winner's coordinates
                    w0beg, w0end = np.where(is_rc, lens - w0end, w0beg), np.where(is_rc, lens - w0beg, w0end)
            if info is not None:
                info_rows = res["rows"]
                if pre and len(info_rows):
                    info_rows = _info_rows_on_original(info_rows, res["before"], lens)
            if mode and n and (pre or self.poly_a or self.length is not None):
                # mask / lowercase change characters, and the modifiers that follow look at characters: the reads
                # are written out as they stand now (the window the earlier modifiers left, marked outside the
                # adapter step's interval) and become the chunk everything else works on -- a rare combination
                # of options, served by two formatting passes on the host instead of kernels of its own
                info_chunk, info_shift = out_chunk, w0beg      # (the info file shows reads without the marking)
                out_chunk = _materialized(out_chunk, w0beg, w0end, wbeg, wend, mode)
                base = out_chunk.reads(self.device)
                quals = qoff = None
                wbeg, wend = np.zeros(n, dtype=np.int64), (w0end - w0beg).astype(np.int64)
                mode = 0
        if self.poly_a and n:
            o, l = view_args()
            idx = qt.poly_a_trim_batch(base.seqs, o, l, n, self.poly_a_revcomp).astype(np.int64)
            # tail: record[:index], removed length len - index; head (revcomp): record[index:], removed index
            removed, counts = np.unique(idx if self.poly_a_revcomp else (wend - wbeg) - idx, return_counts=True)
            for k, c in zip(removed.tolist(), counts.tolist()):
                self.poly_a_trimmed_lengths[k] = self.poly_a_trimmed_lengths.get(k, 0) + c
            if self.poly_a_revcomp:
                wbeg = wbeg + idx
            else:
                wend = wbeg + idx
        if self.length is not None:
            cur = wend - wbeg
            if self.length >= 0:
                wend = wbeg + np.minimum(cur, self.length)
            else:
                wbeg = wend - np.minimum(cur, -self.length)
        ee = None
        if self.max_expected_errors is not None and n:
            # TooManyExpectedErrors (reference predicates.py:55-71): the filter sees the read as trimmed so far
            if quals is None:
                if out_chunk is not chunk:                   # --revcomp / materialised: the chunk that is written
                    q = out_chunk.pack_qualities()
                    if q is None:
                        raise qt.HasNoQualities("expected errors need qualities")
                    quals = torch.from_numpy(q).to(base.device)
                else:
                    load_qualities("expected errors need qualities")
            o, l = view_args()
            o = qual_view(o)
            ee, valid = qt.expected_errors_batch(quals, o, l, n, 33)      # the reference calls expected_errors(q) with its default base
            if not valid.all():
                bad = int(np.flatnonzero(~valid)[0])
                raise ValueError(f"Not a valid phred value in the qualities of read {bad} of the chunk")
        if info is not None:
            # match rows show the read as it came in, the line of a read without a match shows it as it is written
            # (reference steps.py:232-253)
            info.append((info_chunk or out_chunk).write_info(
                info_rows, self.cutter.names if self.cutter is not None else [], is_rc,
                final=(info_shift + wbeg, info_shift + wend)))
        self.reads += n
        self.bp_in += bp_in
        return {"beg": wbeg, "end": wend, "matched": matched, "mode": mode, "ee": ee, "lens": lens, "chunk": out_chunk,
                "rc": is_rc}

    def process_chunk(self, chunk: FastqChunk, discard_untrimmed: bool = False, discard_trimmed: bool = False,
                      info: Optional[list] = None, minimum_length: Optional[int] = None,
                      maximum_length: Optional[int] = None) -> bytes:
        res = self.modify(chunk, info)
        keep = filter_reads([res], [self], discard_untrimmed, discard_trimmed, (minimum_length,), (maximum_length,), "any")
        return self.write(chunk, res, keep)

    def write(self, chunk: FastqChunk, res: Dict[str, object], keep: Optional[np.ndarray]):
        beg, end = res["beg"], res["end"]
        out_len = (end - beg) if res["mode"] == 0 else res["lens"]
        self.bp_out += int(out_len.sum() if keep is None else out_len[keep].sum())
        return res.get("chunk", chunk).write_records(beg.astype(np.int32), end.astype(np.int32), keep, res["mode"])


def filter_reads(results: Sequence[Dict[str, object]], trimmers: Sequence["BatchTrimmer"], discard_untrimmed: bool,
                 discard_trimmed: bool, minimum_length: Sequence[Optional[int]], maximum_length: Sequence[Optional[int]],
                 pair_filter: str = "any", untrimmed_pair_filter: Optional[str] = None) -> Optional[np.ndarray]:
    """The filtering steps in the reference's order (cli.py:735-912): too short, too long, too many
    expected errors, then --discard-trimmed / --discard-untrimmed.  ``results`` holds one entry
    (single-end) or two (a read pair: both mates are kept or dropped together; a predicate combines
    over the mates by ``pair_filter`` = any / both / first like PairedEndFilter, steps.py).  A mate
    whose limit is None takes no part in that predicate.  Returns the keep mask (None = keep all)."""
    n = len(results[0]["beg"])
    keep = np.ones(n, dtype=bool)

    def combine(preds, mode):
        preds = [p for p in preds if p is not None]
        if not preds:
            return None
        if mode == "first":
            return preds[0]
        out = preds[0].copy()
        for p in preds[1:]:
            out = (out | p) if mode == "any" else (out & p)
        return out

    lens = [r["end"] - r["beg"] if r["mode"] == 0 else r["lens"] for r in results]
    short = combine([None if m is None else (l < m) for l, m in zip(lens, minimum_length)], pair_filter)
    long_ = combine([None if m is None else (l > m) for l, m in zip(lens, maximum_length)], pair_filter)
    ee = combine([None if (t.max_expected_errors is None or r["ee"] is None) else (r["ee"] > t.max_expected_errors)
                  for r, t in zip(results, trimmers)], pair_filter)
    for name, pred in (("too_short", short), ("too_long", long_), ("too_many_expected_errors", ee)):
        if pred is not None:
            hit = keep & pred
            if name == "too_many_expected_errors":
                trimmers[0].too_many_expected_errors += int(hit.sum())
            else:
                trimmers[0].filtered[name] = trimmers[0].filtered.get(name, 0) + int(hit.sum())
            keep &= ~pred
    if discard_trimmed:
        keep &= ~combine([r["matched"] for r in results], pair_filter)
    elif discard_untrimmed:
        keep &= ~combine([~r["matched"] for r in results], untrimmed_pair_filter or pair_filter)
    return None if keep.all() else keep


def trim_fastq(inpath: Union[str, BinaryIO], outpath: Union[str, BinaryIO], adapters=(), times: int = 1,
               action: Optional[str] = "trim", discard_untrimmed: bool = False, discard_trimmed: bool = False,
               info_file: Union[None, str, BinaryIO] = None, chunk_bytes: int = DEFAULT_CHUNK_BYTES,
               index: bool = True, nextseq_trim: Optional[int] = None,
               quality_cutoff: Optional[Tuple[int, int]] = None, quality_base: int = 33, poly_a: bool = False,
               max_expected_errors: Optional[float] = None, threads: int = 1, cut: Sequence[int] = (),
               length: Optional[int] = None, minimum_length: Optional[int] = None,
               maximum_length: Optional[int] = None, device=None, devices=None, revcomp: bool = False,
               rc_suffix: Optional[str] = " rc") -> Dict[str, object]:
    """``revcomp``: --revcomp, search every read and its reverse complement and keep the better one (``rc_suffix`` is
    appended to the names of the reads that were turned around; None for none).
    ``devices``: with ``threads`` > 1 the GPUs the worker threads are dealt to, round-robin -- a list of device
    indices or "all" for every visible GPU (the reference's reader -> workers -> ordered writer layout,
    runners.py:116-134, :224-245, with one HIP stream per worker and the workers spread over the node's GPUs;
    plans replicate their tables on each device on first use).  Default: the one ``device``.

    ``cutadapt [--nextseq-trim N] [-q [FRONT,]BACK] [--quality-base B] <adapter options> [--times N]
    [--action A] [--poly-a] [--max-ee E] [--discard-(un)trimmed] [--info-file F] -o outpath inpath`` for the
    supported slice; returns the read/basepair counters the reference reports (reference
    report.py:62-80), the adapter cutter (statistics) and the trimmer."""
    def make_trimmer(dev=device):
        return BatchTrimmer(adapters, times=times, action=action, index=index, nextseq_trim=nextseq_trim,
                            quality_cutoff=quality_cutoff, quality_base=quality_base, poly_a=poly_a,
                            max_expected_errors=max_expected_errors, cut=cut, length=length, device=dev,
                            revcomp=revcomp, rc_suffix=rc_suffix)

    trimmer = make_trimmer()
    out = outpath if hasattr(outpath, "write") else open(outpath, "wb")
    inf = None if info_file is None else (info_file if hasattr(info_file, "write") else open(info_file, "wb"))
    try:
        if threads <= 1:
            for chunk in read_fastq_chunks(inpath, chunk_bytes):
                info: Optional[list] = [] if inf is not None else None
                out.write(trimmer.process_chunk(chunk, discard_untrimmed, discard_trimmed, info, minimum_length,
                                                maximum_length))
                if inf is not None:
                    inf.write(b"".join(info))
        else:
            _trim_threaded(inpath, out, inf, trimmer, make_trimmer, threads, chunk_bytes, discard_untrimmed,
                           discard_trimmed, minimum_length, maximum_length, devices)
    finally:
        if out is not outpath:
            out.close()
        if inf is not None and inf is not info_file:
            inf.close()
    cutter = trimmer.cutter
    return {"reads": trimmer.reads, "with_adapters": cutter.with_adapters if cutter else 0,
            "bp_in": trimmer.bp_in, "bp_out": trimmer.bp_out, "cutter": cutter, "trimmer": trimmer,
            "reverse_complemented": trimmer.rc.reverse_complemented if trimmer.rc is not None else None}


def _trim_threaded(inpath, out, inf, trimmer: "BatchTrimmer", make_trimmer, threads: int, chunk_bytes: int,
                   discard_untrimmed: bool, discard_trimmed: bool, minimum_length: Optional[int] = None,
                   maximum_length: Optional[int] = None, devices=None) -> None:
    """The reference's reader -> workers -> ordered writer layout (runners.py:96-245) with threads
    instead of processes: the calling thread cuts the input into record-aligned raw chunks and
    writes results in chunk order, ``threads`` workers parse, pack, match (each on its own HIP
    stream) and format.  Parsing, packing and formatting are C calls that release the GIL.
    Workers are dealt round-robin to ``devices`` (chunks therefore go round-robin to the GPUs, like the
    reference deals chunks to its worker processes, runners.py:116-134); results come back in chunk order."""
    import threading
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    import torch
    local = threading.local()
    workers: List["BatchTrimmer"] = []
    lock = threading.Lock()
    if devices == "all":
        devices = list(range(torch.cuda.device_count()))
    elif devices is None:
        devices = [trimmer.device]
    devices = list(devices)
    if not devices:
        raise ValueError("devices must name at least one GPU")

    def work(data: np.ndarray, fasta: bool):
        if not hasattr(local, "trimmer"):
            with lock:
                dev = devices[len(workers) % len(devices)]
                workers.append(None)                        # reserve the slot: the next worker gets the next device
                slot = len(workers) - 1
            if dev is not None:
                dev = torch.device("cuda", dev) if isinstance(dev, int) else torch.device(dev)
                torch.cuda.set_device(dev)                  # per thread: the C ABI works on the current device
            local.trimmer = make_trimmer(dev)
            local.stream = torch.cuda.Stream(device=dev)
            with lock:
                workers[slot] = local.trimmer
        chunk = scan_chunk(data, fasta)
        chunk.pooled = True
        info: Optional[list] = [] if inf is not None else None
        try:
            with torch.cuda.stream(local.stream):
                res = local.trimmer.process_chunk(chunk, discard_untrimmed, discard_trimmed, info, minimum_length,
                                                  maximum_length)
        finally:
            chunk.release()
        return res, (b"".join(info) if info is not None else None)

    pending: deque = deque()
    with ThreadPoolExecutor(max_workers=threads) as pool:
        def drain(limit: int) -> None:
            while len(pending) > limit:
                res, info = pending.popleft().result()
                out.write(res)
                POOL.put(res.obj)                           # the formatted chunk's buffer is free again
                if inf is not None:
                    inf.write(info)
        for data, fasta in read_raw_chunks(inpath, chunk_bytes):
            pending.append(pool.submit(work, data, fasta))
            drain(2 * threads)
        drain(0)
    for w in workers:                                       # merge the workers' statistics
        if w is not None:
            trimmer.merge(w)
    trimmer.devices_used = sorted({str(w.device) for w in workers if w is not None})


# -------------------------------------------------------------------------------------------------
# --pair-adapters (reference PairedAdapterCutter, modifiers.py:412-503)
# -------------------------------------------------------------------------------------------------
class BatchPairedAdapterCutter:
    """Trim adapters in pairs: adapters1[i] must be found in R1 AND adapters2[i] in R2; of all pairs that are,
    the one with the highest total score wins, then the fewest total errors, then the first
    (reference ``_find_best_match_pair``, modifiers.py:480-503).  One fused library call per adapter and mate;
    the best pair is kept per read pair ON THE DEVICE (score / error sums, strict improvement so that the
    first pair wins ties), and only the winning tuples come back."""

    def __init__(self, adapters1, adapters2, action: Optional[str] = "trim"):
        adapters1, adapters2 = list(adapters1), list(adapters2)
        if len(adapters1) != len(adapters2):
            raise ValueError("The number of adapters to trim from R1 and R2 must be the same. "
                             "Given: {} for R1, {} for R2".format(len(adapters1), len(adapters2)))
        if not adapters1:
            raise ValueError("No adapters given")
        for a in adapters1 + adapters2:
            if not isinstance(a, SingleAdapter):
                raise ValueError("--pair-adapters works with single (not linked) adapters")
        if action not in ("trim", "mask", "lowercase", "retain", None):
            raise ValueError(f"action {action!r} is not available with paired adapters")
        self.pairs = list(zip(adapters1, adapters2))
        self.action = action
        self.with_adapters = 0
        self.histograms = (MatchHistogram(len(adapters1)), MatchHistogram(len(adapters2)))
        self.names = ([a.name for a in adapters1], [a.name for a in adapters2])

    def _views(self, request):
        """the ReadBatch an adapter step searches: the chunk, or the windows earlier modifiers left"""
        import torch
        from .batch import ReadBatch
        base, n = request["base"], request["n"]
        if request["window"] is None:
            return base, np.zeros(n, dtype=np.int64), request["lens"]
        wbeg, wend = request["window"]
        starts = base.offsets[:n] + torch.from_numpy(wbeg).to(base.device)
        cur = (wend - wbeg).astype(np.int64)
        views = ReadBatch(base.seqs, starts, torch.from_numpy(cur.astype(np.int32)).to(base.device), n_reads=n, validated=True)
        if base.uniform_len and base.lens is None:
            views.within_uniform = int(base.uniform_len)
        return views, wbeg.astype(np.int64), cur

    def best_pairs(self, batch1, batch2):
        """-> (found[n], pair index[n], coords1[n,6], coords2[n,6]) as numpy; the merge runs on the device"""
        import torch
        from . import batch as _b
        n = batch1.n_reads
        dev = batch1.device
        have = torch.zeros(n, dtype=torch.bool, device=dev)
        best_score = torch.zeros(n, dtype=torch.int32, device=dev)
        best_err = torch.zeros(n, dtype=torch.int32, device=dev)
        best_idx = torch.zeros(n, dtype=torch.int32, device=dev)
        c1 = torch.zeros((n, 6), dtype=torch.int32, device=dev)
        c2 = torch.zeros((n, 6), dtype=torch.int32, device=dev)
        for i, (a1, a2) in enumerate(self.pairs):
            r1 = self._match(a1, batch1)
            r2 = self._match(a2, batch2)
            if (r1.status == _lib.INVALID).any() or (r2.status == _lib.INVALID).any():
                _lib.raise_invalid_reads(max(int(batch1.lengths().max().item()), int(batch2.lengths().max().item())))
            both = (r1.status == _lib.MATCH) & (r2.status == _lib.MATCH)
            score = r1.out6[:, 4] + r2.out6[:, 4]
            err = r1.out6[:, 5] + r2.out6[:, 5]
            better = both & (~have | (score > best_score) | ((score == best_score) & (err < best_err)))
            have |= better
            best_score = torch.where(better, score, best_score)
            best_err = torch.where(better, err, best_err)
            best_idx = torch.where(better, torch.full_like(best_idx, i), best_idx)
            c1 = torch.where(better[:, None], r1.out6, c1)
            c2 = torch.where(better[:, None], r2.out6, c2)
        return (have.cpu().numpy(), best_idx.cpu().numpy().astype(np.int64), c1.cpu().numpy().astype(np.int64),
                c2.cpu().numpy().astype(np.int64))

    @staticmethod
    def _match(adapter, batch):
        """device-resident (out6, status) of one adapter in match coordinates"""
        import torch
        from . import batch as _b
        if not adapter._reverse_reads:
            return _b.match_batch(adapter._fused_plan, batch)
        bm = adapter.match_to_batch(batch)                   # Rightmost*: mirrored, on the device
        return _b.BatchResult(bm.device_coords(), bm.device_found().to(torch.uint8))

    def process(self, request1, request2):
        """the adapter step of both mates: -> (res1, res2) like BatchAdapterCutter.process_arrays"""
        n = request1["n"]
        if n != request2["n"]:
            raise ValueError("Reads are improperly paired")
        if n == 0:
            empty = {"beg": np.zeros(0, np.int32), "end": np.zeros(0, np.int32), "matched": np.zeros(0, bool),
                     "rows": np.zeros((0, 7), np.int64)}
            return empty, dict(empty)
        v1, w1, l1 = self._views(request1)
        v2, w2, l2 = self._views(request2)
        found, idx, c1, c2 = self.best_pairs(v1, v2)
        self.with_adapters += int(found.sum())
        out = []
        for k, (c, wbeg, cur) in enumerate(((c1, w1, l1), (c2, w2, l2))):
            ads = [p[k] for p in self.pairs]
            before = np.array([a._remove_before for a in ads], dtype=bool)[idx]
            anywhere = np.array([isinstance(a, AnywhereAdapter) for a in ads], dtype=bool)[idx]
            before = np.where(anywhere, c[:, 2] == 0, before)
            rstart, rstop = c[:, 2], c[:, 3]
            if self.action == "retain":                      # trim_but_retain_adapter (modifiers.py:184-198)
                beg = np.where(before, rstart, 0)
                end = np.where(before, cur, rstop)
            elif self.action is None:
                beg, end = np.zeros(n, np.int64), cur.copy()
            else:                                            # trim / mask / lowercase: what Match.trimmed() leaves
                beg = np.where(before, rstop, 0)
                end = np.where(before, cur, rstart)
            beg = np.where(found, beg, 0)
            end = np.where(found, end, cur)
            f = np.flatnonzero(found)
            removed = np.where(before[f], rstop[f], cur[f] - rstart[f])
            self.histograms[k].add_rows(idx[f], removed, c[f, 5])
            out.append({"beg": (wbeg + beg).astype(np.int32), "end": (wbeg + end).astype(np.int32),
                        "matched": found.copy(), "rows": np.zeros((0, 7), np.int64)})
        return out[0], out[1]


# -------------------------------------------------------------------------------------------------
# paired-end (reference PairedEndPipeline, pipeline.py:76-158; PairedEndModifierWrapper,
# modifiers.py:51-79; PairedEndFilter, steps.py)
# -------------------------------------------------------------------------------------------------
def read_paired_chunks(path1, path2, chunk_bytes: int = DEFAULT_CHUNK_BYTES) -> Iterator[Tuple[FastqChunk, FastqChunk]]:
    """Chunks of two files with the SAME number of records (the job of dnaio.read_paired_chunks,
    runners.py:104-113): each side is read and indexed on its own, the shorter record count wins and
    the surplus records of the other side are carried over to the next round."""
    L = _lib.lib()
    files = [_open_maybe_gz(path1), _open_maybe_gz(path2)]
    carry = [b"", b""]
    eof = [False, False]
    fasta: List[Optional[bool]] = [None, None]
    while True:
        scanned = []
        for k in (0, 1):
            block = b"" if eof[k] else files[k].read(chunk_bytes)
            if not block:
                eof[k] = True
            data = np.frombuffer(carry[k] + block, dtype=np.uint8)
            if fasta[k] is None and len(data):
                fasta[k] = bool(data[0] == ord(">"))
            if len(data) == 0:
                scanned.append((data, np.zeros((0, 6), np.int64), 0))
                continue
            max_rec = (int(np.count_nonzero(data == ord(">"))) + 2) if fasta[k] else int(np.count_nonzero(data == 10)) // 4 + 2
            rec = np.empty((max_rec, 6), dtype=np.int64)
            n = C.c_int64(0)
            consumed = C.c_int64(0)
            _lib.check((L.cah_fasta_scan if fasta[k] else L.cah_fastq_scan)(
                data.ctypes.data, len(data), int(eof[k]), max_rec, rec.ctypes.data, C.byref(n), C.byref(consumed)))
            scanned.append((data, rec[:n.value], consumed.value))
        n = min(len(scanned[0][1]), len(scanned[1][1]))
        if n == 0:
            if eof[0] and eof[1]:
                if len(scanned[0][1]) != len(scanned[1][1]):
                    raise ValueError("Reads are improperly paired. There are more reads in one file than in the other.")
                break
            if all(len(sc[0]) > 64 * chunk_bytes for sc in scanned):
                raise ValueError("record larger than 64 chunks: not a FASTA/FASTQ file?")
            carry = [sc[0].tobytes() for sc in scanned]
            continue
        pair = []
        for k in (0, 1):
            data, rec, consumed = scanned[k]
            # bytes covered by the first n records: up to the start of record n (its marker character)
            cut = consumed if n == len(rec) else int(rec[n, 0]) - 1
            carry[k] = data[cut:].tobytes()
            pair.append(FastqChunk(data[:cut], rec[:n]))
        yield pair[0], pair[1]


def trim_fastq_paired(in1, in2, out1, out2, r1: Optional[dict] = None, r2: Optional[dict] = None,
                      pair_filter: Optional[str] = None, minimum_length=None, maximum_length=None,
                      discard_untrimmed: bool = False, discard_trimmed: bool = False,
                      chunk_bytes: int = DEFAULT_CHUNK_BYTES, device=None, pair_adapters: bool = False,
                      revcomp: bool = False, rc_suffix: Optional[str] = " rc") -> Dict[str, object]:
    """``cutadapt <R1 options> <R2 options: -A/-G/-B, -U, -Q, -L ...> -o out1 -p out2 in1 in2``:
    ``r1`` / ``r2`` are BatchTrimmer keyword arguments for the two mates (adapters, times, action,
    cut, nextseq_trim, quality_cutoff, quality_base, poly_a, length, max_expected_errors).
    ``minimum_length`` / ``maximum_length``: one number for both mates or a pair (None = no limit
    for that mate, reference ``-m 5:7``).  Pairs are filtered as a unit with ``--pair-filter``
    any (default) / both / first; like the reference (cli.py:861-892), --discard-untrimmed uses
    'both' when only one mate has adapters."""
    job = PairedJob(r1, r2, pair_filter, minimum_length, maximum_length, discard_untrimmed, discard_trimmed, device,
                    pair_adapters, revcomp, rc_suffix)
    o1 = out1 if hasattr(out1, "write") else open(out1, "wb")
    o2 = out2 if hasattr(out2, "write") else open(out2, "wb")
    try:
        for c1, c2 in read_paired_chunks(in1, in2, chunk_bytes):
            b1, b2 = job.process_pair(c1, c2)
            o1.write(b1)
            o2.write(b2)
    finally:
        if o1 is not out1:
            o1.close()
        if o2 is not out2:
            o2.close()
    return job.result()


class PairedJob:
    """Everything ``trim_fastq_paired`` decides once (the two trimmers, the pair filter's modes) + the work per
    pair of chunks; shared by the host-parsed pipeline above and gpu_pipeline.trim_fastq_gpu_paired."""

    def __init__(self, r1: Optional[dict] = None, r2: Optional[dict] = None, pair_filter: Optional[str] = None,
                 minimum_length=None, maximum_length=None, discard_untrimmed: bool = False,
                 discard_trimmed: bool = False, device=None, pair_adapters: bool = False, revcomp: bool = False,
                 rc_suffix: Optional[str] = " rc"):
        r1, r2 = dict(r1 or {}), dict(r2 or {})
        if revcomp and pair_adapters:
            raise ValueError("Cannot use --revcomp with --pair-adapters")       # reference cli.py:1086-1087
        if r2.get("poly_a"):
            r2.setdefault("poly_a_revcomp", True)      # --poly-a on paired data: poly-T head of R2
        self.paired_cutter = None
        if pair_adapters:
            # --pair-adapters (cli.py:594-632): adapter i of R1 is only removed together with adapter i of R2
            if r1.get("times", 1) != 1 or r2.get("times", 1) != 1:
                raise ValueError("--pair-adapters cannot be used with --times")
            action = r1.get("action", "trim")
            self.paired_cutter = BatchPairedAdapterCutter(r1.pop("adapters", ()), r2.pop("adapters", ()), action)
            r1["action"] = r2["action"] = action
        self.t1, self.t2 = BatchTrimmer(device=device, **r1), BatchTrimmer(device=device, **r2)

        def both(v):
            return tuple(v) if isinstance(v, (tuple, list)) else (v, v)

        self.min_len, self.max_len = both(minimum_length), both(maximum_length)
        self.mode = "any" if pair_filter is None else pair_filter
        if self.mode not in ("any", "both", "first"):
            raise ValueError("pair_filter must be any, both or first")
        one_sided = (self.t1.cutter is None or self.t2.cutter is None) and self.paired_cutter is None
        self.untrimmed_mode = "both" if (one_sided and discard_untrimmed) else self.mode
        self.discard_untrimmed, self.discard_trimmed = discard_untrimmed, discard_trimmed
        self.pairs = self.kept = 0
        # --revcomp on read pairs: PairedReverseComplementer takes the adapter cutters' place when at least one mate has
        # adapters (reference cli.py:1102-1110)
        self.revcomp = bool(revcomp) and (self.t1.cutter is not None or self.t2.cutter is not None)
        self.rc_suffix = rc_suffix
        self.reverse_complemented = 0

    def process_pair(self, c1, c2):
        """two chunks with the same number of records -> the bytes to write for R1 and R2"""
        t1, t2, paired_cutter = self.t1, self.t2, self.paired_cutter
        if len(c1) != len(c2):
            raise ValueError("Reads are improperly paired")
        if self.revcomp:
            g1, g2 = t1._modify_steps(c1), t2._modify_steps(c2)
            a1, a2 = self._swap_step(next(g1), next(g2), c1, c2)
            res = []
            for g, a in ((g1, a1), (g2, a2)):
                try:
                    g.send(a)
                    raise RuntimeError("modify: the step generator did not finish")
                except StopIteration as stop:
                    res.append(stop.value)
            res1, res2 = res
        elif paired_cutter is None:
            res1, res2 = t1.modify(c1), t2.modify(c2)
        else:
            g1, g2 = t1._modify_steps(c1), t2._modify_steps(c2)
            a1, a2 = paired_cutter.process(next(g1), next(g2))
            res = []
            for g, a in ((g1, a1), (g2, a2)):
                try:
                    g.send(a)
                    raise RuntimeError("modify: the step generator did not finish")
                except StopIteration as stop:
                    res.append(stop.value)
            res1, res2 = res
        keep = filter_reads([res1, res2], [t1, t2], self.discard_untrimmed, self.discard_trimmed, self.min_len,
                            self.max_len, self.mode, self.untrimmed_mode)
        self.pairs += len(c1)
        self.kept += len(c1) if keep is None else int(keep.sum())
        return t1.write(c1, res1, keep), t2.write(c2, res2, keep)

    def _swap_step(self, q1, q2, c1, c2):
        """PairedReverseComplementer (reference modifiers.py:311-405): both cutters on the pair as it is and on the pair
        with its mates exchanged ("equivalent to reverse complementing"); the exchanged pair is taken when the scores
        of its matches add up to MORE.  -> the adapter step's results for the two OUTPUT mates, each with the chunk
        that holds its records from here on (FastqChunk.selected)."""
        t = (self.t1, self.t2)
        n = q1["n"]
        if n != q2["n"]:
            raise ValueError("Reads are improperly paired")
        q = (q1, q2)
        lens = (q1["lens"], q2["lens"])

        def window(k):
            w = q[k]["window"]
            if w is None:
                return np.zeros(n, dtype=np.int64), lens[k].copy()
            return np.asarray(w[0], dtype=np.int64), np.asarray(w[1], dtype=np.int64)

        def search(cutter, k):
            """``cutter`` on input mate k"""
            if cutter is None or n == 0:
                b, e = window(k)
                return {"beg": b, "end": e, "matched": np.zeros(n, dtype=bool), "rows": np.zeros((0, 9), dtype=np.int64),
                        "stats": np.zeros((0, 3), dtype=np.int64), "score": np.zeros(n, dtype=np.int64)}
            return cutter.search(q[k]["base"], lens[k], q[k]["window"])

        plain = (search(t[0].cutter, 0), search(t[1].cutter, 1))
        swapped = (search(t[0].cutter, 1), search(t[1].cutter, 0))       # cutter 1 on R2, cutter 2 on R1
        use = (swapped[0]["score"] + swapped[1]["score"]) > (plain[0]["score"] + plain[1]["score"])
        self.reverse_complemented += int(use.sum())
        hc = tuple(c.host_chunk() if hasattr(c, "host_chunk") else c for c in (c1, c2))
        out = []
        for k in (0, 1):
            f, s_ = plain[k], swapped[k]
            merged = {name: np.where(use, s_[name], f[name]) for name in ("beg", "end", "matched")}
            keep_f, keep_s = ~use[f["rows"][:, 0]], use[s_["rows"][:, 0]]
            rows = np.concatenate([f["rows"][keep_f], s_["rows"][keep_s]])
            stats = np.concatenate([f["stats"][keep_f], s_["stats"][keep_s]])
            order = np.lexsort((rows[:, 7], rows[:, 0]))
            merged["rows"], merged["stats"] = rows[order], stats[order]
            lens_out = np.where(use, lens[1 - k], lens[k])
            if t[k].cutter is not None:
                res = t[k].cutter.commit(merged, lens_out, reverse_complemented=use)
            else:
                res = {"beg": merged["beg"].astype(np.int32), "end": merged["end"].astype(np.int32),
                       "matched": merged["matched"], "rows": np.zeros((0, 7), dtype=np.int64), "before": np.zeros(0, dtype=bool)}
            res["rc"] = use
            res["chunk"] = hc[k].selected(hc[1 - k], use, self.rc_suffix)
            wa, wb = window(k), window(1 - k)
            res["w0"] = (np.where(use, wb[0], wa[0]), np.where(use, wb[1], wa[1]))
            out.append(res)
        return out[0], out[1]

    def merge(self, other: "PairedJob") -> None:
        self.t1.merge(other.t1)
        self.t2.merge(other.t2)
        self.pairs += other.pairs
        self.kept += other.kept
        self.reverse_complemented += other.reverse_complemented
        if self.paired_cutter is not None and other.paired_cutter is not None:
            self.paired_cutter.with_adapters += other.paired_cutter.with_adapters
            for a, b in zip(self.paired_cutter.histograms, other.paired_cutter.histograms):
                a += b

    def result(self) -> Dict[str, object]:
        t1, t2, paired_cutter = self.t1, self.t2, self.paired_cutter
        if paired_cutter is not None:
            with_adapters = (paired_cutter.with_adapters, paired_cutter.with_adapters)
        else:
            with_adapters = (t1.cutter.with_adapters if t1.cutter else 0, t2.cutter.with_adapters if t2.cutter else 0)
        return {"pairs": self.pairs, "pairs_written": self.kept, "trimmers": (t1, t2), "filtered": dict(t1.filtered),
                "with_adapters": with_adapters, "paired_cutter": paired_cutter,
                "reverse_complemented": self.reverse_complemented if self.revcomp else None}


# -------------------------------------------------------------------------------------------------
# adapter specifications ("-a name=^FRONT...BACK$;e=0.2;o=5" etc.): the grammar of reference parser.py
# (:28-86 search parameters, :89-126 brace expansion, :203-300 placement restrictions, :322-361 class choice,
# :472-522 linked adapters, :525-551 single adapters)
# -------------------------------------------------------------------------------------------------
_PARAMETER_ALIASES = {"e": "max_errors", "error_rate": "max_errors", "max_error_rate": "max_errors", "o": "min_overlap"}
_PARAMETER_NAMES = {"max_errors", "min_overlap", "anywhere", "required", "optional", "indels", "noindels", "rightmost"}


def parse_search_parameters(text: str) -> dict:
    """``key=value;key;...`` -> dict (abbreviations resolved, flags True, numbers int or float)"""
    result = {}
    for field in text.split(";"):
        field = field.strip()
        if not field:
            continue
        key, eq, value = field.partition("=")
        key = _PARAMETER_ALIASES.get(key.strip(), key.strip())
        if key not in _PARAMETER_NAMES:
            raise KeyError(f"Unknown parameter '{field.partition('=')[0].strip()}'")
        if eq and value == "":
            raise ValueError(f"No value given for key '{key}'")
        value = value.strip()
        if value == "":
            parsed = True
        else:
            try:
                parsed = int(value)
            except ValueError:
                parsed = float(value)
        if key in result:
            raise KeyError(f"Key '{key}' specified twice")
        result[key] = parsed
    if "optional" in result and "required" in result:
        raise ValueError("'optional' and 'required' cannot be specified at the same time")
    if "indels" in result and "noindels" in result:
        raise ValueError("'indels' and 'noindels' cannot be specified at the same time")
    if result.pop("optional", None) is not None:
        result["required"] = False
    if result.pop("noindels", None) is not None:
        result["indels"] = False
    return result


def expand_braces(sequence: str) -> str:
    """``TGA{5}CT`` -> ``TGAAAAACT``"""
    out = []
    i = 0
    repeatable = False                      # the previous token is a plain character
    while i < len(sequence):
        c = sequence[i]
        if c == "}":
            raise ValueError('"}" cannot be used here')
        if c == "{":
            if not repeatable:
                raise ValueError('"{" must be used after a character')
            close = sequence.find("}", i)
            if close < 0:
                raise ValueError("Unterminated expression")
            count = int(sequence[i + 1:close])
            if not 0 <= count <= 10000:
                raise ValueError(f"Value {count} invalid")
            last = out.pop()
            out.append(last * count)
            i = close + 1
            repeatable = False
            continue
        out.append(c)
        repeatable = True
        i += 1
    return "".join(out)


def _parse_single_spec(spec: str, adapter_type: str):
    """-> (name, restriction, sequence, parameters, rightmost) of one adapter (not linked)"""
    spec, _, parameter_text = spec.partition(";")
    name = None
    if "=" in spec:
        name, spec = spec.split("=", 1)
        name = name.strip()
    parameters = parse_search_parameters(parameter_text)     # (the reference's order: a specification with two faults
    spec = expand_braces(spec.strip())                       #  raises what its first check raises, parser.py:234-238)
    rightmost = bool(parameters.pop("rightmost", False))
    if len(spec.strip("X")) == 0:                  # only X characters: a plain adapter (parser.py:243-246)
        return name, None, spec, {}, False
    front = back = None
    if spec.startswith("^"):
        front, spec = "anchored", spec[1:]
    if spec.upper().startswith("X"):
        if front is not None:
            front = "conflict"
        else:
            front, spec = "noninternal", spec.lstrip("xX")
    if spec.endswith("$"):
        back, spec = "anchored", spec[:-1]
    if spec.upper().endswith("X"):
        if back is not None:
            back = "conflict"
        else:
            back, spec = "noninternal", spec.rstrip("xX")
    if "conflict" in (front, back) or (front and back):
        raise ValueError("You cannot use multiple placement restrictions for an adapter at the same time. "
                         "Choose one of ^ADAPTER, ADAPTER$, XADAPTER or ADAPTERX")
    if adapter_type == "front" and back:
        raise ValueError("Allowed placement restrictions for a 5' adapter are XADAPTER and ^ADAPTER")
    if adapter_type == "back" and front:
        raise ValueError("Allowed placement restrictions for a 3' adapter are ADAPTERX and ADAPTER$")
    restriction = front if front is not None else back
    if adapter_type == "anywhere" and restriction is not None:
        raise ValueError("Placement restrictions (with X, ^, $) not supported for 'anywhere' (-b) adapters")
    if "min_overlap" in parameters and restriction == "anchored":
        raise ValueError("Setting 'min_overlap=' (or 'o=') for anchored adapters is not possible because "
                         "anchored adapters always need to match in full.")
    if parameters.get("min_overlap", 0) > len(spec):
        parameters["min_overlap"] = len(spec)
    if rightmost and (adapter_type not in ("front", "back") or restriction is not None):
        raise ValueError("'rightmost' only allowed with regular 5' and 3' adapters")
    return name, restriction, spec, parameters, rightmost


def _adapter_class(adapter_type: str, restriction, rightmost: bool):
    if adapter_type == "front":
        if rightmost:
            return RightmostFrontAdapter
        return {None: FrontAdapter, "anchored": PrefixAdapter, "noninternal": NonInternalFrontAdapter}[restriction]
    if adapter_type == "back":
        if rightmost:
            return RightmostBackAdapter
        return {None: BackAdapter, "anchored": SuffixAdapter, "noninternal": NonInternalBackAdapter}[restriction]
    return AnywhereAdapter


def adapter_from_spec(spec: str, adapter_type: str = "back", **params):
    """``-a SPEC`` (adapter_type 'back'), ``-g SPEC`` ('front') or ``-b SPEC`` ('anywhere') -> adapter object.
    Understands ``name=``, ``^``/``$`` anchoring, ``X`` non-internal markers, ``{n}`` repeats, ``FRONT...BACK``
    linked adapters and the per-adapter search parameters ``;e=`` / ``;max_error_rate=``, ``;o=`` /
    ``;min_overlap=``, ``;noindels``, ``;anywhere``, ``;rightmost`` and, inside linked adapters, ``;required`` /
    ``;optional``; ``params`` are the command line's defaults, which the spec's own parameters override."""
    if adapter_type not in ("front", "back", "anywhere"):
        raise ValueError("adapter_type must be front, back or anywhere")
    spec1, middle, spec2 = spec.partition("...")
    if middle and spec1 and spec2:
        if adapter_type == "anywhere":
            raise ValueError("'anywhere' (-b) adapters may not be linked")
        fname, frestr, fseq, fparams, fright = _parse_single_spec(spec1, "front")
        _bname, brestr, bseq, bparams, bright = _parse_single_spec(spec2, "back")
        front_parameters = dict(params, **fparams)
        back_parameters = dict(params, **bparams)
        if adapter_type == "front":            # -g requires both parts
            front_required = back_required = True
        else:                                  # -a requires only the anchored parts
            front_required, back_required = frestr is not None, brestr is not None
        front_required = front_parameters.pop("required", front_required)
        back_required = back_parameters.pop("required", back_required)
        # (";anywhere" inside a linked adapter reaches the adapter class as an unknown keyword: TypeError, as in the reference)
        front = _adapter_class("front", frestr, fright)(fseq, name="linked_front", **front_parameters)
        back = _adapter_class("back", brestr, bright)(bseq, name="linked_back", **back_parameters)
        return LinkedAdapter(front, back, front_required, back_required, fname)
    if middle:
        # "ADAPTER..." is a 5' adapter, "...ADAPTER" a 3' adapter (parser.py:129-152)
        if adapter_type == "anywhere":
            raise ValueError('No ellipsis ("...") allowed in "anywhere" adapters')
        if spec1:
            if adapter_type == "back":
                adapter_type = "front"         # -a ADAPTER...  ->  -g ADAPTER
            spec = spec1
        else:
            if adapter_type == "front":
                raise ValueError("Invalid adapter specification")
            spec = spec2
    name, restriction, seq, parameters, rightmost = _parse_single_spec(spec, adapter_type)
    cls = _adapter_class(adapter_type, restriction, rightmost)
    if parameters.pop("anywhere", False) and cls in (FrontAdapter, BackAdapter, RightmostFrontAdapter, RightmostBackAdapter):
        parameters["force_anywhere"] = True
    if "required" in parameters:
        raise ValueError("'required' and 'optional' can only be used within linked adapters")
    return cls(seq, name=name, **dict(params, **parameters))
