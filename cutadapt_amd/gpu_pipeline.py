"""FASTQ in -> trimmed FASTQ out with parsing, matching AND formatting on the GPU.

The batch pipeline of ``pipeline.py`` parses and formats records on host cores (fastq.cpp), which bounds it
at ~20 Mreads/s however fast the matcher is.  Here a raw, record-aligned chunk goes to HBM as it is
(fastq_gpu.hip): the device finds the line feeds, builds the record index, matches the reads *in place*
(an offsets + lens view into the raw chunk: nothing is packed), computes the kept interval of every read,
filters, and formats the trimmed records; only raw FASTQ bytes go in over PCIe and only trimmed FASTQ bytes
come out.  The host cuts the input at record starts (``cah_record_boundary``, no parsing), deals chunks to
worker threads -- each with its own HIP stream and pinned staging buffers, spread round-robin over the
visible GPUs -- and writes the results back in chunk order: the reference's reader -> workers -> ordered
writer layout (reference src/cutadapt/runners.py:96-245) with GPUs as the workers.

Two ways through a chunk once it is indexed:
  * the all-device way (any number of single adapters with ``--times N`` -- rightmost ones only among themselves --, or linked
    adapters (``--times 1``); action ``trim``
    -- or, with one round of single adapters, ``none`` / ``retain`` / ``crop``: other intervals from the same matches --;
    the marking actions ``mask`` / ``lowercase`` (marked in place in HBM); ``--revcomp`` with action ``trim``, one round and
    single adapters (both orientations matched, the better one turned around in place in HBM); ``--info-file`` with single
    adapters -- action ``trim`` with any number of rounds, or one round of an action that leaves the characters alone -- or
    linked adapters (the rows formatted on the device);
    no adapter at all: the other modifiers and the filters alone;
    ``-u`` / ``--nextseq-trim`` / ``-q`` in front of the adapter step, ``--poly-a`` / ``-l`` / ``--max-ee`` / ``-m`` /
    ``-M`` / ``--discard-(un)trimmed`` behind it -- the usual ``cutadapt -q 20 -a ADAPTER -m 20`` and more): trim,
    match, decide and format without a byte of per-read data touching the host (``cah_trim_decide_device`` /
    ``cah_trim_decide_window_device`` / ``cah_trim_decide_action_device`` / ``cah_trim_filter_device``);
  * the general way (everything else ``pipeline.BatchTrimmer`` does: rightmost or linked adapters among others,
    ``--revcomp`` with several rounds, ``--revcomp`` / ``--info-file`` with a marking action, ``--pair-adapters``, adapter sets regrouped behind an
    ``AdapterIndex``): the
    modifiers run as kernels on windows into the raw chunk in HBM (``DeviceFastqChunk``: reads AND qualities are used in place), the window arithmetic between them is
    numpy on 4-byte-per-read arrays, and plain slicing is formatted on the device again.  What cannot be expressed
    as a slice of the raw chunk (there: mask / lowercase, reverse-complemented records, info files) is formatted by the
    host writers from the device's record index -- no host parsing in either way.
FASTA input is parsed on the host (``pipeline.trim_fastq``; a FASTA sequence may span lines and cannot be matched
in place), through the same trimmer.

Feeding: one feeder per GPU -- ``threads`` worker threads with a HIP stream each, pinned staging and output
buffers allocated while that GPU is current (HIP places them on the NUMA node next to it) and, when several GPUs are
fed, pinned to the CPUs of that node.  Chunks are dealt round-robin over the feeders and written in chunk order.
``feeder="process"`` makes every feeder a process of its own (own interpreter, own HIP context: the reference's
ParallelPipelineRunner shape): each reads its own byte ranges of the input file and writes its chunks into the output
file at the offsets the parent hands out (``_trim_fastq_gpu_processes``); ``profiles/r04/feeder_scaling.json`` has what
either shape moves on one box.
For a plain file the dealer only looks for record starts around the nominal cut points (1 MiB windows); the bytes
themselves are read by the feeders (``preadv`` straight into their pinned buffers), so no single thread touches
all the data.
"""
import ctypes as C
import os
import threading
import time
from collections import deque, namedtuple
from concurrent.futures import ThreadPoolExecutor
from typing import BinaryIO, Dict, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from .adapters import AnywhereAdapter, LinkedAdapter, MultipleAdapters, SingleAdapter

DEFAULT_GPU_CHUNK_BYTES = 64 * 1024 * 1024


class _PinnedPool:
    """pinned host buffers for the formatted output: a worker fills one per chunk, the ordered writer returns it
    after writing (pinning memory is expensive; the pool stays small because the writer drains in order)"""

    def __init__(self):
        self._free: list = []
        self._lock = threading.Lock()

    def get(self, nbytes: int):
        import torch
        with self._lock:
            for i, b in enumerate(self._free):
                if b.numel() >= nbytes:
                    return self._free.pop(i)
        return torch.empty(int(nbytes * 1.2) + 4096, dtype=torch.uint8).pin_memory()

    def put(self, buf) -> None:
        with self._lock:
            if len(self._free) < 64:
                self._free.append(buf)


class _PinnedInputPool:
    """pinned INPUT buffers as numpy arrays (what a file reader fills), each backed by a pinned torch tensor: a reader that
    fills them hands the device copy its source directly -- no staging copy through a worker's own pinned buffer.  Same
    interface as pipeline.POOL (views go back as their owning buffer)."""

    def __init__(self, max_free: int = 64):
        self._free: list = []
        self._owner: dict = {}                                # id(root array) -> (root array, pinned tensor)
        self._lock = threading.Lock()
        self._max_free = max_free

    @staticmethod
    def _root(arr):
        while isinstance(arr, np.ndarray) and isinstance(arr.base, np.ndarray):
            arr = arr.base
        return arr

    def get(self, nbytes: int) -> np.ndarray:
        import torch
        with self._lock:
            best = -1
            for i, a in enumerate(self._free):
                if len(a) >= nbytes and (best < 0 or len(a) < len(self._free[best])):
                    best = i
            if best >= 0:
                return self._free.pop(best)
        t = torch.empty(max(int(nbytes), 1), dtype=torch.uint8).pin_memory()
        a = t.numpy()
        with self._lock:
            self._owner[id(a)] = (a, t)
        return a

    def put(self, arr) -> None:
        root = self._root(arr)
        with self._lock:
            if id(root) not in self._owner:
                return
            if len(self._free) >= self._max_free:
                # full: the smallest buffer goes (a job with larger blocks would otherwise pin new memory for every block)
                small = min(range(len(self._free)), key=lambda i: len(self._free[i]))
                if len(self._free[small]) >= len(root):
                    del self._owner[id(root)]
                    return
                del self._owner[id(self._free.pop(small))]
            self._free.append(root)

    def trim(self, keep: int = 8) -> None:
        """let go of all but ``keep`` free buffers (a job's end: dozens of blocks were in flight, pinned memory is scarce)"""
        with self._lock:
            while len(self._free) > keep:
                del self._owner[id(self._free.pop())]

    def tensor_of(self, arr):
        """the pinned tensor view over the bytes of ``arr`` (an array, or a view of one, this pool handed out)"""
        root = self._root(arr)
        with self._lock:
            owner = self._owner.get(id(root))
        if owner is None:
            return arr
        off = arr.ctypes.data - root.ctypes.data
        return owner[1][off:off + len(arr)]


_PINNED_INPUT = _PinnedInputPool()


class _Worker:
    """one stream + its (grow-only) buffers on one device; a chunk costs a dozen library calls, no allocation"""

    def __init__(self, plan, adapter_kinds, device, opts, pool: _PinnedPool):
        import torch
        self.torch = torch
        self.plan = plan
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.kinds = torch.tensor(adapter_kinds, dtype=torch.uint8, device=self.device)
        self.opts = opts
        self.pool = pool
        self.cap = 0
        self.rcap = 0
        self.d_in = self.d_out = self.d_scratch = self.h_in = None
        self.d_info = torch.zeros(8, dtype=torch.int64, device=self.device)
        self.h_info = torch.zeros(8, dtype=torch.int64).pin_memory()
        self.counters = torch.zeros(8, dtype=torch.int64, device=self.device)   # see cah_trim_decide_device
        self.pre_counts = torch.zeros(2, dtype=torch.int64, device=self.device)  # bases removed by NextSeq / quality trimming
        # reads by length of the poly-A tail removed (+ 4096 spare bins that are bin 0 spread out, see _run)
        self.polya_hist = torch.zeros(_lib.MAX_READ_LEN + 1 + 4096, dtype=torch.int64, device=self.device)
        self.ee_invalid = torch.zeros((), dtype=torch.bool, device=self.device)  # a quality value outside the phred range seen
        self.rc_count = torch.zeros((), dtype=torch.int64, device=self.device)   # --revcomp: reads turned around
        self._ws = None
        self.n = self.n_bytes = 0
        self.busy_s, self.chunks, self.bytes_in, self.reads_in = 0.0, 0, 0, 0   # per-device rates (trim_fastq_gpu's "per_device")
        self.stream.wait_stream(torch.cuda.current_stream(self.device))   # (the zeroing / uploads above ran on it)

    def _ensure(self, nbytes: int):
        torch = self.torch
        if nbytes <= self.cap:
            return
        cap = int(nbytes * 1.25) + 4096
        self.d_in = torch.empty(cap, dtype=torch.uint8, device=self.device)
        self.d_out = torch.empty(cap + cap // 2, dtype=torch.uint8, device=self.device)
        self.h_in = None                                      # pinned staging for pageable input: made on demand
        self.cap = cap

    def _ensure_records(self, n: int):
        torch = self.torch
        if n <= self.rcap:
            return
        cap = int(n * 1.25) + 1024
        dev = self.device
        from .batch import BatchResult
        self.rec6 = torch.empty((cap, 6), dtype=torch.int64, device=dev)
        self.seq_off = torch.empty(cap, dtype=torch.int64, device=dev)
        self.seq_len = torch.empty(cap, dtype=torch.int32, device=dev)
        self.res = BatchResult(torch.empty((cap, 6), dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.uint8, device=dev),
                               torch.empty(cap, dtype=torch.int32, device=dev))
        self.beg = torch.empty(cap, dtype=torch.int32, device=dev)
        self.end = torch.empty(cap, dtype=torch.int32, device=dev)
        self.keep = torch.empty(cap, dtype=torch.uint8, device=dev)
        self.rcap = cap
        self._ws = None

    def read_range(self, fd: int, offset: int, length: int):
        """bytes [offset, offset + length) of an open file, read straight into this worker's pinned staging buffer"""
        torch = self.torch
        torch.cuda.set_device(self.device)
        self._ensure(length)
        if self.h_in is None or self.h_in.numel() < length:
            self.h_in = torch.empty(self.cap, dtype=torch.uint8).pin_memory()
        view = memoryview(self.h_in.numpy())[:length]
        got = 0
        while got < length:
            k = os.preadv(fd, [view[got:]], offset + got)
            if k <= 0:
                raise IOError("the input file shrank while it was being read")
            got += k
        return self.h_in[:length]

    def load(self, data) -> int:
        """steps 0-2 of a chunk: raw bytes -> HBM, line count, record index.  -> number of records (the stream is
        synchronised once, after the line count)."""
        torch = self.torch
        L = _lib.lib()
        n_bytes = int(len(data))
        self._ensure(n_bytes)
        sp = self.stream.cuda_stream
        # ---- raw chunk -> HBM (from pinned memory: the input itself if it is pinned, else staged) ----------
        if isinstance(data, torch.Tensor) and data.is_pinned():
            src = data
        else:
            if self.h_in is None or self.h_in.numel() < n_bytes:
                self.h_in = torch.empty(self.cap, dtype=torch.uint8).pin_memory()
            src = self.h_in[:n_bytes]
            src.numpy()[:] = data if isinstance(data, np.ndarray) else data.numpy()
        self.d_in[:n_bytes].copy_(src, non_blocking=True)
        last_byte = int(data[n_bytes - 1])
        # ---- step 1: count lines, size the record arrays ------------------------------------------------
        need = int(L.cah_fastq_device_scratch_bytes(n_bytes, n_bytes // 64 + 1024))
        if self.d_scratch is None or self.d_scratch.numel() < need:
            self.d_scratch = torch.empty(need + need // 4, dtype=torch.uint8, device=self.device)
        _lib.check(L.cah_fastq_count_lines_device(self.d_in.data_ptr(), n_bytes, self.d_scratch.data_ptr(),
                                                  self.d_scratch.numel(), self.d_info.data_ptr(), sp))
        self.h_info.copy_(self.d_info, non_blocking=True)
        self.stream.synchronize()
        n_newlines = int(self.h_info[0])
        n_lines = n_newlines + (1 if last_byte != 10 else 0)
        if n_lines % 4 != 0:
            raise ValueError("FASTQ format error: premature end of file (incomplete record)")
        n = n_lines // 4
        need = int(L.cah_fastq_device_scratch_bytes(n_bytes, n))
        if self.d_scratch.numel() < need:
            # the tile counts of step 1 live at the front of the scratch: keep them
            grown = torch.empty(need + need // 4, dtype=torch.uint8, device=self.device)
            grown[: self.d_scratch.numel()].copy_(self.d_scratch)
            self.d_scratch = grown
        self._ensure_records(max(n, 1))
        _lib.check(L.cah_fastq_index_device(self.d_in.data_ptr(), n_bytes, n_newlines, n, self.d_scratch.data_ptr(),
                                            self.d_scratch.numel(), self.rec6.data_ptr(), self.seq_off.data_ptr(),
                                            self.seq_len.data_ptr(), self.d_info.data_ptr(), sp))
        self.n, self.n_bytes = n, n_bytes
        self.reads_in += int(n)
        return n

    def check_index(self) -> None:
        """raise the format error the index step left in d_info, if any (synchronises the stream)"""
        self.h_info.copy_(self.d_info, non_blocking=True)
        self.stream.synchronize()
        self._raise_format_error(int(self.h_info[1]))

    @staticmethod
    def _raise_format_error(err: int) -> None:
        if err != -1:
            code, record = err & 0xFF, (err >> 8) - 1
            what = {1: "line expected to start with '@'", 2: "third line expected to start with '+'",
                    3: "length of sequence and qualities differ"}.get(code, "malformed record")
            raise ValueError(f"FASTQ format error in record {record} of the chunk: {what}")

    def run(self, data, is_final: bool):
        """data: uint8 numpy array or torch tensor on the host holding whole records -> (pinned buffer, bytes)"""
        torch = self.torch
        L = _lib.lib()
        if int(len(data)) == 0:
            return None, 0
        torch.cuda.set_device(self.device)
        t0 = time.perf_counter()
        try:
            with torch.cuda.stream(self.stream):
                return self._run(data, L)
        finally:
            self.busy_s += time.perf_counter() - t0
            self.chunks += 1
            self.bytes_in += int(len(data))

    def run_general(self, data, job):
        """the general way (module docstring): index on the device, then ``job`` (a callable taking the
        DeviceFastqChunk) does the rest -> whatever job returns, plus the pinned buffers to recycle"""
        torch = self.torch
        if int(len(data)) == 0:
            return None, []
        torch.cuda.set_device(self.device)
        t0 = time.perf_counter()
        try:
            with torch.cuda.stream(self.stream):
                n = self.load(data)
                self.check_index()
                chunk = DeviceFastqChunk(self, data, n, self.n_bytes)
                return job(chunk), chunk.out_bufs
        finally:
            self.busy_s += time.perf_counter() - t0
            self.chunks += 1
            self.bytes_in += int(len(data))

    def _run(self, data, L):
        n = self.load(data)
        n_bytes = self.n_bytes
        # ---- step 3: modifiers, matching in place, what is kept ---------------------------------------------
        o = self.opts
        if n:
            limits = (-1 if o["minimum_length"] is None else int(o["minimum_length"]),
                      -1 if o["maximum_length"] is None else int(o["maximum_length"]),
                      int(bool(o["discard_trimmed"])), int(bool(o["discard_untrimmed"])))
            self.modify(n, o.get("pre"), o.get("post"), limits)
        return self.finish(data, n, n_bytes)

    def modify(self, n: int, pre, post, limits=None):
        """The read-modifying steps of a loaded chunk on this worker's stream (reference cli.py:938-973): ``pre`` (-u,
        --nextseq-trim, -q) -> adapter step (``self.plan``; None: no adapters) -> ``post`` (--poly-a, -l) and the
        expected errors for --max-ee.  Leaves the kept intervals in self.beg / self.end (relative to the read) and the
        matches' status in self.res.status.  ``limits`` = (min_len, max_len, discard_trimmed, discard_untrimmed): the
        filters are applied here too (self.keep, counters); None: the caller filters (read pairs are filtered as a
        unit).  -> the expected errors (float64 tensor) or None."""
        torch = self.torch
        L = _lib.lib()
        sp = self.stream.cuda_stream
        seq_len, seq_off = self.seq_len[:n], self.seq_off[:n]
        qual_off = self.rec6[:n, 4]
        wbeg = None                                          # None: the matcher sees whole reads
        wlen = seq_len
        keepalive = []                                       # (tensors kernels read: alive until the stream is through)
        if pre:
            # the modifiers in front of the adapter step (-u, --nextseq-trim, -q: reference cli.py:938-954) as kernels
            # and a few element-wise operations on this stream; the matcher then sees a window of every read
            wbeg = torch.zeros(n, dtype=torch.int32, device=self.device)
            wlen = seq_len.clone()
            for c in pre["cut"]:                              # UnconditionalCutter: read[c:] / read[:c]
                if c > 0:
                    d = torch.clamp(wlen, max=c)
                    wbeg += d
                    wlen -= d
                else:
                    wlen = torch.clamp(wlen + c, min=0)
            if pre["nextseq_trim"] is not None:
                stop = torch.empty(n, dtype=torch.int32, device=self.device)
                w64 = wbeg.to(torch.int64)
                so, qo = seq_off + w64, qual_off + w64       # (named: a temporary's block could be handed out again
                _lib.check(L.cah_nextseq_trim_batch_q(       #  before the kernel has read it)
                    self.d_in.data_ptr(), self.d_in.data_ptr(), so.data_ptr(), qo.data_ptr(),
                    wlen.data_ptr(), n, int(pre["nextseq_trim"]), int(pre["quality_base"]), stop.data_ptr(), sp))
                self.pre_counts[0] += (wlen - stop).sum()
                keepalive += [so, qo, wlen]
                wlen = stop
            if pre["quality_cutoff"] is not None:
                ss = torch.empty((n, 2), dtype=torch.int32, device=self.device)
                qo = qual_off + wbeg.to(torch.int64)
                _lib.check(L.cah_quality_trim_batch(
                    self.d_in.data_ptr(), qo.data_ptr(), wlen.data_ptr(), n, int(pre["quality_cutoff"][0]),
                    int(pre["quality_cutoff"][1]), int(pre["quality_base"]), ss.data_ptr(), sp))
                kept = ss[:, 1] - ss[:, 0]
                self.pre_counts[1] += (wlen - kept).sum()
                keepalive += [qo, wlen]
                wbeg = wbeg + ss[:, 0]
                wlen = kept.contiguous()
            wbeg = wbeg.contiguous()
        voff = seq_off if wbeg is None else (seq_off + wbeg.to(torch.int64)).contiguous()
        linked = isinstance(self.plan, (_LinkedPlans, list))
        if linked:
            # linked adapters (reference adapters.py:1215-1227): per adapter the 5' plan on the window, the 3' plan on what is
            # behind the 5' match (cah_linked_views), the required / optional verdict and the kept interval as a handful of
            # element-wise operations on this stream; several of them (round 6): MultipleAdapters' rule over the LinkedMatches
            # (adapters.py:1278-1285: the higher score -- the parts' scores added up --, then fewer errors, then the first)
            plans = self.plan if isinstance(self.plan, list) else [self.plan]
            ws_need = max(int(L.cah_plan_workspace_bytes(h.handle, n)) for lp in plans for h in (lp.front, lp.back))
            if self._ws is None or self._ws.numel() < ws_need:
                self._ws = torch.empty(ws_need + ws_need // 4, dtype=torch.uint8, device=self.device)
            wl = wlen.contiguous()
            w0 = wbeg if wbeg is not None else torch.zeros(n, dtype=torch.int32, device=self.device)
            zero = torch.zeros_like(wl)
            best = None
            invalid = None
            want_info = self.opts.get("info") is not None
            parts = None                                          # --info-file: the winner's parts as two "rounds" of results
            for li, lp in enumerate(plans):
                f_out6 = torch.empty((n, 6), dtype=torch.int32, device=self.device)
                f_status = torch.empty(n, dtype=torch.uint8, device=self.device)
                f_best = torch.empty(n, dtype=torch.int32, device=self.device)
                b_out6 = torch.empty((n, 6), dtype=torch.int32, device=self.device)
                b_status = torch.empty(n, dtype=torch.uint8, device=self.device)
                b_best = torch.empty(n, dtype=torch.int32, device=self.device)
                starts = torch.empty(n, dtype=torch.int64, device=self.device)
                vlens = torch.empty(n, dtype=torch.int32, device=self.device)
                _lib.check(L.cah_match_batch(lp.front.handle, self.d_in.data_ptr(), voff.data_ptr(), wl.data_ptr(), n,
                                             f_out6.data_ptr(), f_best.data_ptr(), f_status.data_ptr(),
                                             self._ws.data_ptr(), self._ws.numel(), sp))
                _lib.check(L.cah_linked_views(f_out6.data_ptr(), f_status.data_ptr(), voff.data_ptr(), wl.data_ptr(), 0, n,
                                              starts.data_ptr(), vlens.data_ptr(), sp))
                _lib.check(L.cah_match_batch(lp.back.handle, self.d_in.data_ptr(), starts.data_ptr(), vlens.data_ptr(), n,
                                             b_out6.data_ptr(), b_best.data_ptr(), b_status.data_ptr(),
                                             self._ws.data_ptr(), self._ws.numel(), sp))
                F, B = f_status == 1, b_status == 1
                ok = B if lp.back_required else (B | F)           # LinkedAdapter.match_to: None unless ...
                if lp.front_required:
                    ok = ok & F
                b = w0 + torch.where(ok & F, f_out6[:, 3], zero)
                e = torch.where(ok & B, b + b_out6[:, 2], w0 + wl)
                score = torch.where(F, f_out6[:, 4], zero) + torch.where(B, b_out6[:, 4], zero)
                errors = torch.where(F, f_out6[:, 5], zero) + torch.where(B, b_out6[:, 5], zero)
                bad = (f_status == 2) | (b_status == 2)
                invalid = bad if invalid is None else (invalid | bad)
                if best is None:
                    better = ok
                    best = [ok, b, e, score, errors]
                else:
                    better = ok & (~best[0] | (score > best[3]) | ((score == best[3]) & (errors < best[4])))
                    best = [best[0] | ok, torch.where(better, b, best[1]), torch.where(better, e, best[2]),
                            torch.where(better, score, best[3]), torch.where(better, errors, best[4])]
                if want_info:
                    # (the rows of a LinkedMatch: its 5' part on the read, its 3' part on what that left -- LinkedMatch.
                    # get_info_records, adapters.py:1157-1171; names "<adapter>;1" / ";2": entries 2 li and 2 li + 1)
                    if parts is None:
                        parts = (torch.zeros((2, n, 6), dtype=torch.int32, device=self.device),
                                 torch.zeros((2, n), dtype=torch.uint8, device=self.device),
                                 torch.zeros((2, n), dtype=torch.int32, device=self.device))
                    bm = better[:, None]
                    parts[0][0].copy_(torch.where(bm, f_out6, parts[0][0])); parts[0][1].copy_(torch.where(bm, b_out6, parts[0][1]))
                    parts[1][0].copy_(torch.where(better, (ok & F).to(torch.uint8), parts[1][0]))
                    parts[1][1].copy_(torch.where(better, (ok & B).to(torch.uint8), parts[1][1]))
                    parts[2][0].copy_(torch.where(better, torch.full_like(f_best, 2 * li), parts[2][0]))
                    parts[2][1].copy_(torch.where(better, torch.full_like(f_best, 2 * li + 1), parts[2][1]))
                keepalive += [f_out6, f_status, f_best, b_out6, b_status, b_best, starts, vlens]
            ok, b, e = best[0], best[1], best[2]
            self.counters[0] += n
            self.counters[1] += ok.sum()
            self.counters[2] += seq_len.sum()
            self.counters[6] += invalid.sum()
            self.beg[:n].copy_(torch.where(ok, b, w0))
            self.end[:n].copy_(torch.where(ok, e, w0 + wl))
            self.res.status[:n].copy_(ok.to(torch.uint8))     # "with adapters", as the filters read it
            self._linked_info = parts
            keepalive += [wl, w0]
        elif self.plan is not None:
            ws_need = int(L.cah_plan_workspace_bytes(self.plan.handle, n))
            if self._ws is None or self._ws.numel() < ws_need:
                self._ws = torch.empty(ws_need + ws_need // 4, dtype=torch.uint8, device=self.device)
            reversed_reads = bool(self.opts.get("reversed"))

            def match_views(off, ln):
                """the plan on the views (off, ln) of the chunk -> self.res.  Rightmost* adapters (reference adapters.py:766,
                :870: the reversed adapter on the reversed read): the views reversed into a second buffer at the same offsets
                (cah_reverse_reads_batch), matched there, the read coordinates mirrored back (:777-785)"""
                if not reversed_reads:
                    _lib.check(L.cah_match_batch(self.plan.handle, self.d_in.data_ptr(), off.data_ptr(), ln.data_ptr(), n,
                                                 self.res.out6.data_ptr(), self.res.best_adapter.data_ptr(),
                                                 self.res.status.data_ptr(), self._ws.data_ptr(), self._ws.numel(), sp))
                    keepalive.append(ln)
                    return
                if getattr(self, "d_rc", None) is None or self.d_rc.numel() < self.d_in.numel():
                    self.d_rc = torch.empty(self.d_in.numel(), dtype=torch.uint8, device=self.device)
                _lib.check(L.cah_reverse_reads_batch(self.d_in.data_ptr(), off.data_ptr(), ln.data_ptr(), n, off.data_ptr(),
                                                     self.d_rc.data_ptr(), sp))
                _lib.check(L.cah_match_batch(self.plan.handle, self.d_rc.data_ptr(), off.data_ptr(), ln.data_ptr(), n,
                                             self.res.out6.data_ptr(), self.res.best_adapter.data_ptr(),
                                             self.res.status.data_ptr(), self._ws.data_ptr(), self._ws.numel(), sp))
                o6 = self.res.out6[:n]
                q0, q1 = ln - o6[:, 3], ln - o6[:, 2]
                o6[:, 2].copy_(q0)
                o6[:, 3].copy_(q1)
                keepalive.append(ln)
            match_views(voff, wlen.contiguous())
            if self.opts.get("revcomp"):
                # --revcomp (ReverseComplementer, reference modifiers.py:264-308, in the adapter cutter's place): the
                # reverse complement of every window into a second buffer at the same offsets, matched with the same plan;
                # where ITS match scores higher (:287; no match: score 0) the record is turned around in place -- sequence
                # and qualities -- and the match results are the second call's; the formatter adds the suffix to the name
                from .batch import BatchResult
                if getattr(self, "d_rc", None) is None or self.d_rc.numel() < self.d_in.numel():
                    self.d_rc = torch.empty(self.d_in.numel(), dtype=torch.uint8, device=self.device)
                if getattr(self, "res_rc", None) is None or self.res_rc.status.numel() < self.rcap:
                    dev = self.device
                    self.res_rc = BatchResult(torch.empty((self.rcap, 6), dtype=torch.int32, device=dev),
                                              torch.empty(self.rcap, dtype=torch.uint8, device=dev),
                                              torch.empty(self.rcap, dtype=torch.int32, device=dev))
                    self.rc_flags = torch.empty(self.rcap, dtype=torch.uint8, device=dev)
                wl = wlen.contiguous()
                _lib.check(L.cah_revcomp_reads_batch(self.d_in.data_ptr(), voff.data_ptr(), wl.data_ptr(), n, voff.data_ptr(),
                                                     self.d_rc.data_ptr(), 1, None, sp))
                r2 = self.res_rc
                _lib.check(L.cah_match_batch(self.plan.handle, self.d_rc.data_ptr(), voff.data_ptr(), wl.data_ptr(), n,
                                             r2.out6.data_ptr(), r2.best_adapter.data_ptr(), r2.status.data_ptr(),
                                             self._ws.data_ptr(), self._ws.numel(), sp))
                fwd, rev = self.res.status[:n] == 1, r2.status[:n] == 1
                zero = torch.zeros((), dtype=torch.int32, device=self.device)
                use = torch.where(rev, r2.out6[:n, 4], zero) > torch.where(fwd, self.res.out6[:n, 4], zero)
                self.counters[6] += (r2.status[:n] == 2).sum()
                self.res.out6[:n].copy_(torch.where(use[:, None], r2.out6[:n], self.res.out6[:n]))
                self.res.best_adapter[:n].copy_(torch.where(use, r2.best_adapter[:n], self.res.best_adapter[:n]))
                self.res.status[:n].copy_(torch.where(use, r2.status[:n], self.res.status[:n]))
                self.rc_flags[:n].copy_(use)
                self.rc_count += use.sum()
                # (the WHOLE read is turned, as the reference's is -- its info rows show that --: the window the modifiers in
                # front left of it is then [len - wend, len - wbeg) of the turned read)
                _lib.check(L.cah_revcomp_in_place_device(self.d_in.data_ptr(), self.rec6.data_ptr(), n, None,
                                                         self.seq_len.data_ptr(), self.rc_flags.data_ptr(), sp))
                if wbeg is not None:
                    keepalive.append(wbeg)
                    wbeg = torch.where(use, seq_len - wbeg - wl, wbeg).contiguous()
                keepalive += [wl, use]
        else:
            self.res.status[:n].zero_()                      # a mate without adapters: nothing is found
        rounds = int(self.opts.get("times", 1)) if (self.plan is not None and not linked) else 1
        action = int(self.opts.get("action", 0))
        # the marking actions (4 mask, 5 lowercase; reference modifiers.py:170-198): the rounds run as for trim -- what
        # they would keep is remainder(matches) --, then the record stays whole and is marked in place around that interval
        marking = action in (4, 5) and self.plan is not None
        if marking:
            action = 0
        final_here = limits is not None and not post and rounds == 1 and not linked and not marking
        lim = limits if final_here else (-1, -1, 0, 0)
        if linked:
            pass                                             # (intervals and status are in place)
        elif action != 0:
            # --action none / retain / crop: other intervals from the same match results (one round)
            wl = wlen.contiguous()
            _lib.check(L.cah_trim_decide_action_device(
                self.res.out6.data_ptr(), self.res.status.data_ptr(), self.res.best_adapter.data_ptr(),
                wbeg.data_ptr() if wbeg is not None else None, wl.data_ptr(), self.seq_len.data_ptr(), n,
                self.kinds.data_ptr(), action, *lim, 0 if final_here else 1,
                self.beg.data_ptr(), self.end.data_ptr(), self.keep.data_ptr(), self.counters.data_ptr(), sp))
            keepalive.append(wl)
        elif not pre and final_here:
            _lib.check(L.cah_trim_decide_device(
                self.res.out6.data_ptr(), self.res.status.data_ptr(), self.res.best_adapter.data_ptr(),
                self.seq_len.data_ptr(), n, self.kinds.data_ptr(), *lim,
                self.beg.data_ptr(), self.end.data_ptr(), self.keep.data_ptr(), self.counters.data_ptr(), sp))
        else:
            _lib.check(L.cah_trim_decide_window_device(
                self.res.out6.data_ptr(), self.res.status.data_ptr(), self.res.best_adapter.data_ptr(),
                wbeg.data_ptr() if wbeg is not None else None, wlen.data_ptr(), self.seq_len.data_ptr(), n,
                self.kinds.data_ptr(), *lim, 0 if final_here else 1,
                self.beg.data_ptr(), self.end.data_ptr(), self.keep.data_ptr(), self.counters.data_ptr(), sp))
        self._info_rounds = None
        if rounds > 1:
            # --times N (reference modifiers.py:367-380: match, trim, match what is left, ... until nothing is found):
            # every further round matches the interval the last one kept.  A read without a match keeps its interval
            # and cannot match later either, so "with adapters" is round 1's status (restored below for the filters)
            # and every read can take every round; the later rounds' read / bp counters go to a scratch array.
            first_status = self.res.status[:n].clone()
            scratch = torch.zeros_like(self.counters)
            info_rounds = None
            if self.opts.get("info") is not None:
                # --info-file: every round's match of every read is a row (cah_info_format_device takes them round after round)
                info_rounds = (torch.empty((rounds, n, 6), dtype=torch.int32, device=self.device),
                               torch.empty((rounds, n), dtype=torch.uint8, device=self.device),
                               torch.empty((rounds, n), dtype=torch.int32, device=self.device))
                info_rounds[0][0].copy_(self.res.out6[:n]); info_rounds[1][0].copy_(first_status)
                info_rounds[2][0].copy_(self.res.best_adapter[:n])
            self._info_rounds = info_rounds
            for rnd in range(1, rounds):
                rb = self.beg[:n].clone()
                rl = (self.end[:n] - rb).contiguous()
                ro = (seq_off + rb.to(torch.int64)).contiguous()
                match_views(ro, rl)
                _lib.check(L.cah_trim_decide_window_device(
                    self.res.out6.data_ptr(), self.res.status.data_ptr(), self.res.best_adapter.data_ptr(),
                    rb.data_ptr(), rl.data_ptr(), self.seq_len.data_ptr(), n, self.kinds.data_ptr(), -1, -1, 0, 0, 1,
                    self.beg.data_ptr(), self.end.data_ptr(), self.keep.data_ptr(), scratch.data_ptr(), sp))
                if info_rounds is not None:
                    info_rounds[0][rnd].copy_(self.res.out6[:n]); info_rounds[1][rnd].copy_(self.res.status[:n])
                    info_rounds[2][rnd].copy_(self.res.best_adapter[:n])
                keepalive += [rb, rl, ro]
            self.res.status[:n].copy_(first_status)
            keepalive += [first_status, scratch]
        if marking:
            if getattr(self, "mark_beg", None) is None or self.mark_beg.numel() < self.beg.numel():
                self.mark_beg = torch.empty_like(self.beg)
                self.mark_end = torch.empty_like(self.end)
            self.mark_beg[:n].copy_(self.beg[:n])
            self.mark_end[:n].copy_(self.end[:n])
            if wbeg is None:
                self.beg[:n].zero_()
                self.end[:n].copy_(seq_len)
            else:
                self.beg[:n].copy_(wbeg)
                self.end[:n].copy_(wbeg + wlen)
            _lib.check(L.cah_mark_reads_device(self.d_in.data_ptr(), self.rec6.data_ptr(), n, self.beg.data_ptr(),
                                               self.end.data_ptr(), self.mark_beg.data_ptr(), self.mark_end.data_ptr(),
                                               1 if int(self.opts.get("action", 0)) == 4 else 2, sp))
        ee = None
        if post:
            # the modifiers behind the adapter step (--poly-a, -l: cli.py:956-973) move the kept interval, then the
            # filters (--max-ee among them) decide what is written: cah_trim_filter_device
            beg, end = self.beg[:n], self.end[:n]
            if post["poly_a"]:
                head = bool(post.get("poly_a_revcomp"))      # the second mate of a pair loses a poly-T HEAD
                idx = torch.empty(n, dtype=torch.int32, device=self.device)
                cur = (end - beg).contiguous()
                po = (seq_off + beg.to(torch.int64)).contiguous()
                _lib.check(L.cah_poly_a_trim_batch(self.d_in.data_ptr(), po.data_ptr(), cur.data_ptr(), n, int(head),
                                                   idx.data_ptr(), sp))
                removed = (idx if head else cur - idx).to(torch.int64)
                # (index_add_, not bincount: bincount asks the device for the largest value first; the reads without
                # a tail -- nearly all -- are counted in 4096 spare bins behind the histogram instead of all in bin 0)
                spare = _lib.MAX_READ_LEN + 1 + (torch.arange(n, device=self.device) & 4095)
                self.polya_hist.index_add_(0, torch.where(removed != 0, removed, spare), torch.ones_like(removed))
                if head:
                    beg.copy_(beg + idx)
                else:
                    end.copy_(beg + idx)
                keepalive += [cur, po, idx]
            if post["length"] is not None:
                cur = end - beg
                if post["length"] >= 0:
                    end.copy_(beg + torch.clamp(cur, max=int(post["length"])))
                else:
                    beg.copy_(end - torch.clamp(cur, max=-int(post["length"])))
            if post["max_expected_errors"] is not None:
                ee = torch.empty(n, dtype=torch.float64, device=self.device)
                ee_status = torch.zeros(n, dtype=torch.uint8, device=self.device)
                cur = (end - beg).contiguous()
                qo = (qual_off + beg.to(torch.int64)).contiguous()
                _lib.check(L.cah_expected_errors_batch(self.d_in.data_ptr(), qo.data_ptr(), cur.data_ptr(), n, 33,
                                                       ee.data_ptr(), ee_status.data_ptr(), sp))
                self.ee_invalid |= (ee_status == _lib.INVALID).any()
                keepalive += [ee_status, cur, qo]
        if limits is not None and not final_here:
            _lib.check(L.cah_trim_filter_device(
                self.beg.data_ptr(), self.end.data_ptr(), self.res.status.data_ptr(),
                ee.data_ptr() if ee is not None else None, n, limits[0], limits[1],
                float(post["max_expected_errors"]) if ee is not None else -1.0, limits[2], limits[3],
                self.keep.data_ptr(), self.counters.data_ptr(), sp))
        self._keepalive = keepalive + [voff, wlen, wbeg, ee]
        return ee

    def finish(self, data, n: int, n_bytes: int):
        """step 4 of a chunk whose intervals and keep flags are final -> (pinned buffer, bytes); with --info-file on the
        all-device way the info rows are formatted on the device behind the records (``self.last_info`` = (pinned buffer,
        bytes), cah_info_format_device)"""
        info = self.opts.get("info")
        self.last_info = None
        if info is None:
            return self._finish_records(data, n, n_bytes)
        torch = self.torch
        L = _lib.lib()
        sp = self.stream.cuda_stream
        out = self._finish_records(data, n, n_bytes)
        if n == 0:
            return out
        names = info["names"]
        if getattr(self, "_info_names", None) is not names:
            blob = "".join(names).encode("ascii")
            off = np.zeros(len(names) + 1, dtype=np.int32)
            np.cumsum([len(x) for x in names], out=off[1:])
            self.d_names = torch.from_numpy(np.frombuffer(blob + b"\0", dtype=np.uint8).copy()).to(self.device)
            self.d_name_off = torch.from_numpy(off).to(self.device)
            self.d_info_total = torch.zeros(1, dtype=torch.int64, device=self.device)
            self.h_info_total = torch.zeros(1, dtype=torch.int64).pin_memory()
            self._info_names = names
        rc = bool(self.opts.get("revcomp")) and self.plan is not None
        suffix = (self.opts.get("rc_suffix") or "").encode() if rc else b""
        rounds_info = getattr(self, "_info_rounds", None)
        kinds = self.kinds
        if isinstance(self.plan, (_LinkedPlans, list)):
            rounds_info = self._linked_info
            if getattr(self, "_linked_kinds", None) is None or self._linked_kinds.numel() != len(names):
                self._linked_kinds = torch.tensor([1, 0] * (len(names) // 2), dtype=torch.uint8, device=self.device)
            kinds = self._linked_kinds
        if rounds_info is not None:
            r6, rst, rbest, n_rounds = rounds_info[0], rounds_info[1], rounds_info[2], int(rounds_info[1].shape[0])
        else:
            r6, rst, rbest, n_rounds = self.res.out6, self.res.status, self.res.best_adapter, 1
        cap = n_rounds * (n_bytes + n * (max([len(x) for x in names], default=0) + len(suffix) + 48)) + 64
        if getattr(self, "d_info_out", None) is None or self.d_info_out.numel() < cap:
            self.d_info_out = torch.empty(cap + cap // 4, dtype=torch.uint8, device=self.device)
        _lib.check(L.cah_info_format_device(
            self.d_in.data_ptr(), self.rec6.data_ptr(), n, r6.data_ptr(), rst.data_ptr(), rbest.data_ptr(), n_rounds,
            kinds.data_ptr() if self.plan is not None else None, self.beg.data_ptr(), self.end.data_ptr(), self.d_names.data_ptr(),
            self.d_name_off.data_ptr(), len(names), self.rc_flags.data_ptr() if rc else None, suffix if suffix else None,
            len(suffix), self.d_scratch.data_ptr(), self.d_scratch.numel(), n_bytes, self.d_info_out.data_ptr(),
            self.d_info_out.numel(), self.d_info_total.data_ptr(), sp))
        self.h_info_total.copy_(self.d_info_total, non_blocking=True)
        self.stream.synchronize()
        total = int(self.h_info_total[0])
        h = self.pool.get(total)
        h[:total].copy_(self.d_info_out[:total], non_blocking=True)
        self.stream.synchronize()
        self.last_info = (h, total)
        return out

    def _finish_records(self, data, n: int, n_bytes: int):
        torch = self.torch
        L = _lib.lib()
        sp = self.stream.cuda_stream
        o = self.opts
        if o.get("assemble") == "host":
            return self._assemble_on_host(data, n_bytes, n)
        # ---- step 4: format on the device, bring the bytes back ---------------------------------------------
        suffix = (o.get("rc_suffix") or "").encode() if (o.get("revcomp") and self.plan is not None and n) else b""
        if suffix:
            if self.d_out.numel() < n_bytes + (4 + len(suffix)) * n:
                self.d_out = torch.empty(n_bytes + (4 + len(suffix)) * n + 4096, dtype=torch.uint8, device=self.device)
            _lib.check(L.cah_fastq_format_suffix_device(
                self.d_in.data_ptr(), self.rec6.data_ptr(), n, self.beg.data_ptr(), self.end.data_ptr(), self.keep.data_ptr(),
                self.rc_flags.data_ptr(), suffix, len(suffix), self.d_scratch.data_ptr(), self.d_scratch.numel(), n_bytes,
                self.d_out.data_ptr(), self.d_out.numel(), self.d_info.data_ptr(), sp))
        else:
            _lib.check(L.cah_fastq_format_device(self.d_in.data_ptr(), self.rec6.data_ptr(), n, self.beg.data_ptr(),
                                                 self.end.data_ptr(), self.keep.data_ptr(), self.d_scratch.data_ptr(),
                                                 self.d_scratch.numel(), n_bytes, self.d_out.data_ptr(),
                                                 self.d_out.numel(), self.d_info.data_ptr(), sp))
        self.d_info[4:5].copy_(self.counters[6:7], non_blocking=True)
        self.d_info[5:6].copy_(self.ee_invalid.to(torch.int64).reshape(1), non_blocking=True)
        self.h_info.copy_(self.d_info, non_blocking=True)
        self.stream.synchronize()
        self._raise_format_error(int(self.h_info[1]))
        if int(self.h_info[4]) != 0:
            _lib.raise_invalid_reads(int(self.seq_len[:max(n, 1)].max().item()))
        if int(self.h_info[5]) != 0:
            raise ValueError("Not a valid phred value in the qualities of a read of the chunk")
        total = int(self.h_info[3])
        h_out = self.pool.get(total)
        h_out[:total].copy_(self.d_out[:total], non_blocking=True)
        self.stream.synchronize()
        return h_out, total


    def _assemble_on_host(self, data, n_bytes: int, n: int):
        """Step 4, other half of the trade: only the record index and the kept intervals (57 bytes per record
        instead of the ~280 of a formatted record) come back over PCIe, and the trimmed FASTQ is put together from
        the chunk, which is in host memory anyway -- records kept whole are copied in runs
        (``cah_fastq_write_trimmed``).  Byte-identical to the device formatter; leaves the outbound PCIe direction
        almost idle at the price of one memcpy pass per chunk on a host core."""
        torch = self.torch
        L = _lib.lib()
        if getattr(self, "_h_cap", 0) < n:
            cap = int(n * 1.25) + 1024
            self._h_rec6 = torch.empty((cap, 6), dtype=torch.int64).pin_memory()
            self._h_beg = torch.empty(cap, dtype=torch.int32).pin_memory()
            self._h_end = torch.empty(cap, dtype=torch.int32).pin_memory()
            self._h_keep = torch.empty(cap, dtype=torch.uint8).pin_memory()
            self._h_cap = cap
        self.d_info[4:5].copy_(self.counters[6:7], non_blocking=True)
        self.d_info[5:6].copy_(self.ee_invalid.to(torch.int64).reshape(1), non_blocking=True)
        self.h_info.copy_(self.d_info, non_blocking=True)
        if n:
            self._h_rec6[:n].copy_(self.rec6[:n], non_blocking=True)
            self._h_beg[:n].copy_(self.beg[:n], non_blocking=True)
            self._h_end[:n].copy_(self.end[:n], non_blocking=True)
            self._h_keep[:n].copy_(self.keep[:n], non_blocking=True)
        self.stream.synchronize()
        self._raise_format_error(int(self.h_info[1]))
        if int(self.h_info[4]) != 0:
            _lib.raise_invalid_reads(int(self.seq_len[:max(n, 1)].max().item()))
        if int(self.h_info[5]) != 0:
            raise ValueError("Not a valid phred value in the qualities of a read of the chunk")
        h_out = self.pool.get(n_bytes + 4 * n + 64)
        out_len = C.c_int64(0)
        src_ptr = data.data_ptr() if isinstance(data, torch.Tensor) else data.ctypes.data
        _lib.check(L.cah_fastq_write_trimmed(src_ptr, self._h_rec6.data_ptr(), n, self._h_beg.data_ptr(),
                                             self._h_end.data_ptr(), self._h_keep.data_ptr(), h_out.data_ptr(),
                                             h_out.numel(), C.byref(out_len)))
        return h_out, int(out_len.value)


class DeviceFastqChunk:
    """A raw FASTQ chunk in HBM with its record index, as a worker's ``load()`` left it: what ``pipeline.FastqChunk``
    offers the modifiers (lengths / reads / qualities / writers), without packing anything -- the reads and the
    qualities are offsets + lengths INTO the raw chunk.  Lives until the worker loads its next chunk."""

    def __init__(self, worker: "_Worker", data, n: int, n_bytes: int):
        self.w = worker
        self.data = data
        self.n = n
        self.n_bytes = n_bytes
        self.out_bufs: list = []                             # pinned output buffers handed out by write_records
        self._host = None
        self._lens = None

    def __len__(self):
        return self.n

    def lengths(self) -> np.ndarray:
        if self._lens is None:
            self._lens = self.w.seq_len[:self.n].cpu().numpy().astype(np.int64) if self.n else np.zeros(0, np.int64)
        return self._lens

    def reads(self, device=None):
        from .batch import ReadBatch
        w = self.w
        return ReadBatch(w.d_in, w.seq_off[:self.n], w.seq_len[:self.n], n_reads=self.n)

    def qualities(self, base):
        w = self.w
        return w.d_in, w.rec6[:self.n, 4].contiguous()

    def host_chunk(self):
        """the same chunk for the host-side writers: the raw bytes (they are in host memory anyway) + the device's
        record index brought back (48 bytes per record; nothing is parsed on the host)"""
        if self._host is None:
            from .pipeline import FastqChunk
            torch = self.w.torch
            buf = self.data.numpy() if isinstance(self.data, torch.Tensor) else np.asarray(self.data)
            rec = self.w.rec6[:self.n].cpu().numpy() if self.n else np.zeros((0, 6), np.int64)
            self._host = FastqChunk(buf, rec)
        return self._host

    def write_records(self, beg, end, keep=None, mode: int = 0):
        if mode != 0:
            return self.host_chunk().write_records(beg, end, keep, mode)
        w, n = self.w, self.n
        torch = w.torch
        L = _lib.lib()
        if n:
            w.beg[:n].copy_(torch.from_numpy(np.ascontiguousarray(beg, dtype=np.int32)), non_blocking=False)
            w.end[:n].copy_(torch.from_numpy(np.ascontiguousarray(end, dtype=np.int32)), non_blocking=False)
            if keep is None:
                w.keep[:n].fill_(1)
            else:
                w.keep[:n].copy_(torch.from_numpy(np.ascontiguousarray(keep, dtype=np.uint8)), non_blocking=False)
        sp = torch.cuda.current_stream(w.device).cuda_stream
        _lib.check(L.cah_fastq_format_device(w.d_in.data_ptr(), w.rec6.data_ptr(), n, w.beg.data_ptr(), w.end.data_ptr(),
                                             w.keep.data_ptr(), w.d_scratch.data_ptr(), w.d_scratch.numel(), self.n_bytes,
                                             w.d_out.data_ptr(), w.d_out.numel(), w.d_info.data_ptr(), sp))
        w.h_info.copy_(w.d_info, non_blocking=True)
        torch.cuda.current_stream(w.device).synchronize()
        total = int(w.h_info[3])
        h_out = w.pool.get(total)
        h_out[:total].copy_(w.d_out[:total], non_blocking=True)
        torch.cuda.current_stream(w.device).synchronize()
        self.out_bufs.append(h_out)
        return memoryview(h_out.numpy())[:total]

    def write_info(self, rows, names, is_rc=None, final=None):
        return self.host_chunk().write_info(rows, names, is_rc, final)

    def reverse_complemented(self, is_rc, suffix=" rc"):
        return self.host_chunk().reverse_complemented(is_rc, suffix)


# buffers outlive a call: pinning host memory and growing device buffers cost tens of milliseconds, more than a
# whole chunk -- workers (device buffers, stream) and pinned output buffers are recycled between calls
_PINNED: Dict[str, _PinnedPool] = {}                        # one per device: its buffers sit on that GPU's NUMA node
_IDLE_WORKERS: Dict[str, list] = {}
_IDLE_LOCK = threading.Lock()


def _take_worker(plan, kinds, dev, opts) -> "_Worker":
    import torch
    key = str(torch.device(dev))
    with _IDLE_LOCK:
        idle = _IDLE_WORKERS.get(key, [])
        w = idle.pop() if idle else None
    if w is None:
        with _IDLE_LOCK:
            pinned = _PINNED.setdefault(key, _PinnedPool())
        return _Worker(plan, kinds, dev, opts, pinned)
    torch.cuda.set_device(w.device)
    w.plan, w.opts = plan, opts
    with torch.cuda.stream(w.stream):                           # ordered in front of the worker's next kernels
        w.kinds = torch.tensor(kinds, dtype=torch.uint8, device=w.device)
        w.counters.zero_()
        w.pre_counts.zero_()
        w.polya_hist.zero_()
        w.ee_invalid.zero_()
        w.rc_count.zero_()
    w._ws = None
    w.busy_s, w.chunks, w.bytes_in, w.reads_in = 0.0, 0, 0, 0
    return w


def _give_back(w: "_Worker") -> None:
    with _IDLE_LOCK:
        idle = _IDLE_WORKERS.setdefault(str(w.device), [])
        if len(idle) < 16:
            idle.append(w)


def _adapter_list(adapters) -> list:
    if adapters is None:
        return []
    if isinstance(adapters, MultipleAdapters):
        return list(adapters._adapters)
    if isinstance(adapters, (list, tuple)):
        return list(adapters)
    return [adapters]


def _index_regroups(adapters) -> bool:
    """would ``BatchAdapterCutter(index=True)`` put these adapters behind an ``AdapterIndex`` (reference
    modifiers.py:124-141: more than one anchored 5' or more than one anchored 3' adapter the index accepts)?  Then the
    adapters are tried in another order and looked up, not aligned: the general way's business."""
    from .adapters import AdapterIndex
    prefix = [a for a in adapters if isinstance(a, SingleAdapter) and AdapterIndex.is_acceptable(a, True)]
    suffix = [a for a in adapters if isinstance(a, SingleAdapter) and a not in prefix and AdapterIndex.is_acceptable(a, False)]
    return len(prefix) > 1 or len(suffix) > 1


_LinkedPlans = namedtuple("_LinkedPlans", "front back front_required back_required")


def _all_device_adapters(adapters, times: int, index: bool) -> bool:
    """adapter sets ``_Worker.modify`` serves: single adapters (any number, any --times; rightmost ones only among themselves), or
    linked adapters of non-rightmost parts (any number of them, --times 1)"""
    if adapters and all(isinstance(a, LinkedAdapter) for a in adapters):
        return times == 1 and not any(a.front_adapter._reverse_reads or a.back_adapter._reverse_reads for a in adapters)
    # (single adapters: none of them rightmost, or -- round 6 -- all of them: those are matched on reversed views)
    return (times >= 1 and all(isinstance(a, SingleAdapter) for a in adapters)
            and len({bool(a._reverse_reads) for a in adapters}) == 1
            and not (index and _index_regroups(adapters)))


def _plan_for(adapters):
    """the fused plan + adapter kinds of the all-device way"""
    if all(isinstance(a, LinkedAdapter) for a in adapters):
        plans = [_LinkedPlans(a.front_adapter._fused_plan, a.back_adapter._fused_plan, bool(a.front_required),
                              bool(a.back_required)) for a in adapters]
        return (plans[0] if len(plans) == 1 else plans), [0]
    plan = adapters[-1]._fused_plan if len(adapters) == 1 else _lib.Plan([a.matcher_spec() for a in adapters])
    kinds = [2 if isinstance(a, AnywhereAdapter) else (1 if a._remove_before else 0) for a in adapters]
    return plan, kinds


def _device_cpus(device) -> Optional[set]:
    """the CPUs next to a GPU (its PCI device's local_cpulist), or None when the system does not say"""
    import torch
    try:
        p = torch.cuda.get_device_properties(device)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as f:
            text = f.read().strip()
        cpus = set()
        for part in text.split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        return cpus or None
    except Exception:
        return None


class _Feeder:
    """One per GPU: ``threads`` worker threads (a HIP stream and a set of device + pinned buffers each), created
    while this GPU is current and, when ``pin_cpus``, running on the CPUs next to it."""

    def __init__(self, device, threads: int, pin_cpus: bool, make_worker):
        self.device = device
        self.make_worker = make_worker
        self.cpus = _device_cpus(device) if pin_cpus else None
        self.local = threading.local()
        self.workers: list = []
        self.lock = threading.Lock()
        self.pool = ThreadPoolExecutor(max_workers=threads, initializer=self._enter_thread)

    def _enter_thread(self) -> None:
        import torch
        if self.cpus:
            try:
                os.sched_setaffinity(0, self.cpus)          # pid 0: the calling thread
            except OSError:
                pass
        torch.cuda.set_device(self.device)

    def worker(self):
        if not hasattr(self.local, "w"):
            with self.lock:
                slot = len(self.workers)
                self.workers.append(None)
            self.local.w = self.make_worker(self.device, slot)
            with self.lock:
                self.workers[slot] = self.local.w
        return self.local.w

    def submit(self, fn, *args):
        return self.pool.submit(lambda: fn(self.worker(), *args))

    def close(self) -> None:
        self.pool.shutdown(wait=True)


def _file_ranges(path: str, chunk_bytes: int):
    """(offset, length, is_final) of record-aligned pieces of a plain FASTQ file; only 1 MiB windows around the
    nominal cut points are read here (``cah_record_boundary``), the pieces themselves by whoever gets them"""
    L = _lib.lib()
    size = os.path.getsize(path)
    fd = os.open(path, os.O_RDONLY)
    try:
        if size and os.pread(fd, 1, 0) == b">":
            raise ValueError("the device-side path reads 4-line FASTQ; use pipeline.trim_fastq for FASTA")
        pos = 0
        while pos < size:
            stop = min(size, pos + chunk_bytes)
            window = 1 << 20
            while stop < size:
                w = min(window, stop - pos)
                win = np.frombuffer(os.pread(fd, w, stop - w), dtype=np.uint8)
                cut = C.c_int64(0)
                _lib.check(L.cah_record_boundary(win.ctypes.data, len(win), 0, C.byref(cut)))
                if cut.value:
                    stop = stop - w + cut.value
                    break
                if w < stop - pos:
                    window *= 8                              # no record start in the window: look further back
                elif stop - pos >= 64 * chunk_bytes:
                    raise ValueError("record larger than 64 chunks: not a FASTQ file?")
                else:
                    stop = min(size, stop + chunk_bytes)     # ... or the record is longer than the piece
            yield pos, stop - pos, stop >= size
            pos = stop
    finally:
        os.close(fd)


def _is_plain_file(source) -> bool:
    if not isinstance(source, (str, os.PathLike)) or not os.path.isfile(source):
        return False
    with open(source, "rb") as f:
        return f.read(2) != b"\x1f\x8b"


def _chunks(source, chunk_bytes: int):
    """record-aligned pieces of the input, unparsed: (array, is_final).  ``source``: a path / binary file (read
    into pooled buffers) or an in-memory uint8 array / pinned torch tensor (sliced without copying)."""
    import torch
    L = _lib.lib()
    if isinstance(source, (np.ndarray, torch.Tensor)):
        total = int(len(source))
        arr = source.numpy() if isinstance(source, torch.Tensor) else source
        if total and arr[0] == ord(">"):
            raise ValueError("the device-side path reads 4-line FASTQ; use pipeline.trim_fastq for FASTA")
        pos = 0
        while pos < total:
            stop = min(total, pos + chunk_bytes)
            if stop < total:
                cut = C.c_int64(0)
                _lib.check(L.cah_record_boundary(arr[pos:].ctypes.data, stop - pos, 0, C.byref(cut)))
                if cut.value == 0:
                    stop = min(total, pos + 64 * chunk_bytes) if stop - pos < 64 * chunk_bytes else None
                    if stop is None:
                        raise ValueError("record larger than 64 chunks: not a FASTQ file?")
                    if stop < total:
                        _lib.check(L.cah_record_boundary(arr[pos:].ctypes.data, stop - pos, 0, C.byref(cut)))
                        stop = pos + cut.value if cut.value else total
                else:
                    stop = pos + cut.value
            yield source[pos:stop], stop >= total
            pos = stop
        return
    from .pipeline import read_raw_chunks
    prev = None
    for data, fasta in read_raw_chunks(source, chunk_bytes):
        if fasta:
            raise ValueError("the device-side path reads 4-line FASTQ; use pipeline.trim_fastq for FASTA")
        if prev is not None:
            yield prev, False
        prev = data
    if prev is not None:
        yield prev, True


def _resolve_devices(devices) -> list:
    import torch
    if devices == "all":
        devices = list(range(torch.cuda.device_count()))
    elif devices is None:
        devices = [torch.cuda.current_device()]
    devices = [torch.device("cuda", d) if isinstance(d, int) else torch.device(d) for d in devices]
    if not devices:
        raise ValueError("devices must name at least one GPU")
    return devices


def _is_fasta(source) -> bool:
    import torch
    if isinstance(source, (np.ndarray, torch.Tensor)):
        return bool(len(source)) and int(source[0]) == ord(">")
    if isinstance(source, (str, os.PathLike)):
        from .pipeline import _open_maybe_gz
        f = _open_maybe_gz(source)
        try:
            return f.read(1) == b">"
        finally:
            f.close()
    if hasattr(source, "peek"):
        return source.peek(1)[:1] == b">"
    if hasattr(source, "seek") and hasattr(source, "tell"):
        at = source.tell()
        first = source.read(1)
        source.seek(at)
        return first == b">"
    return False


def _per_device(feeders, wall: float) -> Dict[str, dict]:
    out = {}
    for f in feeders:
        ws = [w for w in f.workers if w is not None]
        nbytes = sum(w.bytes_in for w in ws)
        out[str(f.device)] = {"workers": len(ws), "chunks": sum(w.chunks for w in ws), "bytes_in": int(nbytes),
                              "reads_in": int(sum(getattr(w, "reads_in", 0) for w in ws)),
                              "GB_per_s_in": nbytes / wall / 1e9 if wall > 0 else 0.0,
                              "busy_fraction": (sum(w.busy_s for w in ws) / (wall * len(ws))) if ws and wall > 0 else 0.0,
                              "cpus": len(f.cpus) if f.cpus else None}
    return out


def trim_fastq_gpu(source: Union[str, BinaryIO, np.ndarray], out: Union[str, BinaryIO, None], adapters=(),
                   discard_untrimmed: bool = False, discard_trimmed: bool = False,
                   minimum_length: Optional[int] = None, maximum_length: Optional[int] = None,
                   chunk_bytes: int = DEFAULT_GPU_CHUNK_BYTES, threads: int = 3, devices=None,
                   assemble: str = "device", times: int = 1, action: Optional[str] = "trim", index: bool = True,
                   nextseq_trim: Optional[int] = None, quality_cutoff: Optional[Tuple[int, int]] = None,
                   quality_base: int = 33, poly_a: bool = False, max_expected_errors: Optional[float] = None,
                   cut: Sequence[int] = (), length: Optional[int] = None, revcomp: bool = False,
                   rc_suffix: Optional[str] = " rc", info_file: Union[None, str, BinaryIO] = None,
                   feeder: str = "thread", _ranges=None, _deliver=None, _stub_device: bool = False,
                   _repeat: int = 1, _general: bool = False) -> Dict[str, object]:
    """``cutadapt [-u N] [--nextseq-trim N] [-q [F,]B] <adapter options> [--times N] [--action A] [--revcomp]
    [--poly-a] [-l N] [--max-ee E] [-m N] [-M N] [--discard-(un)trimmed] [--info-file F] -o out in.fastq`` with the
    records indexed, matched, filtered and formatted on the GPU(s) (module docstring: which option sets take the
    all-device way and which the general way).  ``out``: path, binary file, or None (the output is produced and
    counted but not kept: measures the pipeline without a sink).
    ``devices``: list of indices, "all", or None = the current device; every device gets a feeder of ``threads``
    worker threads (a stream and pinned buffers each), chunks are dealt round-robin over the devices and written
    in chunk order (reference runners.py:116-134, :224-245).
    ``assemble`` (all-device way): "device" formats the trimmed records on the GPU and brings the bytes back; "host"
    brings back only the record index and kept intervals and copies the records together on the worker's host core
    (same bytes; trades the outbound PCIe traffic for host memcpy); "mixed" lets every other worker do that.
    ``feeder``: "thread" -- one process, a feeder (thread pool) per GPU; "process" -- one feeder PROCESS per GPU, the
    shape of the reference's ParallelPipelineRunner (runners.py:275-412: a reader that deals chunks, N worker
    processes, an ordered writer) without its pipes for bulk data: every process reads its own byte ranges of the input
    file and writes its trimmed chunks straight into the output file at the offset the parent hands out once the sizes
    of all earlier chunks are known (``_trim_fastq_gpu_processes``).  Input must be a plain file, output a path or None.
    (``_ranges`` / ``_deliver`` / ``_stub_device`` are that mode's and the feeder-scaling measurement's internals:
    explicit (index, offset, length, is_final) pieces, a callback that takes the finished chunks in order, and "do no
    device work, hand the input back" -- profiles/scripts/r04_feeder_scaling.py.)
    Returns the reference's counters (report.py:62-80) + ``devices_used`` and ``per_device`` (chunks, bytes, input
    rate and busy fraction of every feeder)."""
    import torch
    if feeder not in ("thread", "process"):
        raise ValueError("feeder must be 'thread' or 'process'")
    if feeder == "process":
        return _trim_fastq_gpu_processes(source, out, dict(
            adapters=adapters, discard_untrimmed=discard_untrimmed, discard_trimmed=discard_trimmed,
            minimum_length=minimum_length, maximum_length=maximum_length, chunk_bytes=chunk_bytes, threads=threads,
            assemble=assemble, times=times, action=action, index=index, nextseq_trim=nextseq_trim,
            quality_cutoff=quality_cutoff, quality_base=quality_base, poly_a=poly_a,
            max_expected_errors=max_expected_errors, cut=cut, length=length, revcomp=revcomp, rc_suffix=rc_suffix,
            _stub_device=_stub_device), devices, info_file, _repeat)
    if assemble not in ("device", "host", "mixed", "mixed3"):
        raise ValueError("assemble must be 'device', 'host', 'mixed' or 'mixed3'")
    adapters = _adapter_list(adapters)
    general_opts = dict(times=times, action=action, index=index, nextseq_trim=nextseq_trim, quality_cutoff=quality_cutoff,
                        quality_base=quality_base, poly_a=poly_a, max_expected_errors=max_expected_errors, cut=cut,
                        length=length, revcomp=revcomp, rc_suffix=rc_suffix)
    if _is_fasta(source):
        # a FASTA sequence may span lines: parsed and packed on the host, same trimmer (module docstring)
        from .pipeline import trim_fastq
        if isinstance(source, (np.ndarray, torch.Tensor)):
            import io
            source = io.BytesIO((source.numpy() if isinstance(source, torch.Tensor) else source).tobytes())
        sink = out if out is not None else open(os.devnull, "wb")
        try:
            res = trim_fastq(source, sink, adapters, discard_untrimmed=discard_untrimmed, discard_trimmed=discard_trimmed,
                             info_file=info_file, minimum_length=minimum_length, maximum_length=maximum_length,
                             threads=threads, devices=devices, **general_opts)
        finally:
            if out is None:
                sink.close()
        res["devices_used"] = getattr(res["trimmer"], "devices_used", [str(_resolve_devices(devices)[0])])
        res["way"] = "host-parsed (FASTA)"
        return res
    cut = [int(c) for c in cut if int(c) != 0]
    if len(cut) > 2:
        raise ValueError("You cannot remove bases from more than two ends.")
    if len(cut) == 2 and cut[0] * cut[1] > 0:
        raise ValueError("You cannot remove bases from the same end twice.")
    # the all-device way: action trim with any number of rounds, or -- one round, single adapters -- the actions that only
    # move the kept interval (none / retain / crop); no adapters at all (-q / --nextseq-trim / --poly-a / -l / --max-ee / -m
    # alone) is the same way with an empty adapter step
    act = {"trim": 0, None: 1, "none": 1, "retain": 2, "crop": 3, "mask": 4, "lowercase": 5}.get(action, -1)
    no_linked = not any(isinstance(a, LinkedAdapter) for a in adapters)
    single_round_action = act in (1, 2, 3) and int(times) == 1 and bool(adapters) and no_linked
    # (round 6: the marking actions -- mask / lowercase, reference modifiers.py:170-198 -- too: the rounds run as for trim,
    # then the reads are marked IN PLACE in the device's copy of the chunk around what the rounds would keep
    # (cah_mark_reads_device): the modifiers behind the adapter step and the formatter see the marked read)
    marking = act in (4, 5) and bool(adapters)
    # (... and --revcomp with action trim, one round, single adapters: both orientations matched, the better one kept in
    # place -- _Worker.modify; without adapters --revcomp does nothing, reference cli.py:1113-1118)
    rc_device = bool(revcomp) and bool(adapters)
    rightmost = any(isinstance(a, SingleAdapter) and a._reverse_reads for a in adapters)
    rc_ok = not rc_device or (act == 0 and int(times) == 1 and no_linked and not rightmost and len((rc_suffix or "").encode()) <= _lib.MAX_NAME_SUFFIX)
    # (... and --info-file with single adapters -- action trim with any number of rounds, or one round of an action that leaves
    # the characters alone: the rows are formatted on the device too, cah_info_format_device)
    info_ok = info_file is None or ((no_linked and (act == 0 or (int(times) == 1 and act in (1, 2, 3)))) or
                                    (not no_linked and act == 0 and int(times) == 1))
    # (_general: tests only -- the general way for an option set the all-device way serves, to compare the two)
    all_device = (not _general and rc_ok and info_ok and
                  ((not adapters and action in ("trim", None, "none", "retain", "crop", "mask", "lowercase")) or
                   (bool(adapters) and (act == 0 or single_round_action or marking) and _all_device_adapters(adapters, int(times), index))))
    pre = post = None
    if all_device and (cut or nextseq_trim is not None or quality_cutoff is not None):
        pre = {"cut": cut, "nextseq_trim": nextseq_trim, "quality_cutoff": quality_cutoff, "quality_base": quality_base}
    if all_device and (poly_a or length is not None or max_expected_errors is not None):
        post = {"poly_a": bool(poly_a), "length": length, "max_expected_errors": max_expected_errors}
    devices = _resolve_devices(devices)
    threads = max(1, int(threads))
    opts = {"discard_untrimmed": discard_untrimmed, "discard_trimmed": discard_trimmed,
            "minimum_length": minimum_length, "maximum_length": maximum_length, "assemble": assemble, "pre": pre, "post": post,
            "times": int(times), "action": max(act, 0) if adapters else 0, "revcomp": all_device and rc_device,
            "rc_suffix": rc_suffix,
            "reversed": bool(adapters) and all(isinstance(a, SingleAdapter) and a._reverse_reads for a in adapters),
            "info": {"names": [nm for a in adapters for nm in (
                [("none" if a.name is None else str(a.name)) + ";1", ("none" if a.name is None else str(a.name)) + ";2"]
                if isinstance(a, LinkedAdapter) else [str(a.name)])]} if (all_device and info_file is not None) else None}
    if all_device and adapters and (act in (4, 5) or rc_device):
        assemble = opts["assemble"] = "device"               # (the host-side assembler copies slices of the INPUT: it cannot mark or turn)
    from .pipeline import BatchTrimmer
    if all_device:
        plan, kinds = _plan_for(adapters) if adapters else (None, [0])
    else:
        plan, kinds = None, []
        BatchTrimmer(adapters, device=devices[0], **general_opts)       # option errors surface here, not in a worker

    def make_worker(dev, slot):
        # "mixed": every other worker assembles on its host core -- the outbound PCIe traffic and the host's
        # memcpy work are both halved (measured best when neither direction of the link is to be the bound)
        # ("mixed3": two of three workers)
        wopts = opts
        if assemble == "mixed":
            wopts = dict(opts, assemble="host" if slot % 2 else "device")
        elif assemble == "mixed3":
            wopts = dict(opts, assemble="host" if slot % 3 else "device")
        w = _take_worker(plan, kinds, dev, wopts)
        w.trimmer = None if all_device else BatchTrimmer(adapters, device=dev, **general_opts)
        return w

    ranged = _is_plain_file(source)
    # (stub mode: a worker's staging buffer is reused by its next chunk -- the echo is copied when somebody will read it)
    sink_wants_copy = out is not None or (_deliver is not None and getattr(_deliver, "reads_body", True))
    fd = os.open(source, os.O_RDONLY) if ranged else None
    from_pool = not ranged and not isinstance(source, (np.ndarray, torch.Tensor))
    want_info = info_file is not None

    def work(w: _Worker, item, is_final):
        data = w.read_range(fd, item[0], item[1]) if ranged else item
        try:
            if _stub_device:
                # measurement only: the host side of a feeder (range read into pinned memory, hand-over, ordered sink)
                # without any device work -- the input comes back as the "output"
                w.chunks += 1
                w.bytes_in += int(len(data))
                body = bytes(memoryview(data.numpy() if hasattr(data, "numpy") else data)) if sink_wants_copy else memoryview(data.numpy() if hasattr(data, "numpy") else data)
                return body, [], None, w
            if all_device:
                buf, total = w.run(data, is_final)
                bufs = [buf] if buf is not None else []
                rows = None
                if want_info:
                    rows = b""
                    if buf is not None and w.last_info is not None:
                        rows = memoryview(w.last_info[0].numpy())[:w.last_info[1]]
                        bufs.append(w.last_info[0])
                return (memoryview(buf.numpy())[:total] if buf is not None else b""), bufs, rows, w
            info: Optional[list] = [] if want_info else None
            body, bufs = w.run_general(data, lambda chunk: w.trimmer.process_chunk(
                chunk, discard_untrimmed, discard_trimmed, info, minimum_length, maximum_length))
            return (body if body is not None else b""), bufs, (b"".join(info) if info is not None else None), w
        finally:
            if from_pool:
                from .pipeline import POOL
                POOL.put(data)                              # the reader's buffer is free again

    feeders = [_Feeder(dev, threads, len(devices) > 1, make_worker) for dev in devices]
    sink = None if out is None else (out if hasattr(out, "write") else open(out, "wb"))
    inf = None if info_file is None else (info_file if hasattr(info_file, "write") else open(info_file, "wb"))
    bytes_out = 0
    t_start = time.perf_counter()
    try:
        pending: deque = deque()

        def drain(limit: int) -> None:
            nonlocal bytes_out
            while len(pending) > limit:
                chunk_index, fut = pending.popleft()
                body, bufs, info, w = fut.result()
                bytes_out += len(body)
                if _deliver is not None:
                    _deliver(chunk_index, body)             # (process mode: the chunk goes where the parent says)
                elif sink is not None and len(body):
                    sink.write(body)                        # straight from the pinned buffer
                if inf is not None and info is not None and len(info):
                    inf.write(info)
                body = info = None
                for b in bufs:
                    w.pool.put(b)
        if _ranges is not None:
            if not ranged:
                raise ValueError("explicit ranges need a plain input file")
            items = ((idx, ((off, ln), fin)) for idx, off, ln, fin in _ranges)
        elif ranged:
            base_ranges = list(_file_ranges(source, chunk_bytes))
            # (_repeat: measurement only -- the file's pieces over and over, for runs long enough to show a steady rate)
            items = ((i, ((it[0], it[1]), it[2])) for i, it in enumerate(base_ranges * max(1, int(_repeat))))
        else:
            items = enumerate(_chunks(source, chunk_bytes))
        for i, (chunk_index, (item, is_final)) in enumerate(items):
            pending.append((chunk_index, feeders[i % len(feeders)].submit(work, item, is_final)))
            drain(2 * threads * len(feeders))
        drain(0)
    finally:
        for f in feeders:
            f.close()
        if fd is not None:
            os.close(fd)
        if sink is not None and sink is not out:
            sink.close()
        if inf is not None and inf is not info_file:
            inf.close()
    wall = time.perf_counter() - t_start
    workers = [w for f in feeders for w in f.workers if w is not None]
    result: Dict[str, object] = {"bytes_out": int(bytes_out), "per_device": _per_device(feeders, wall),
                                 "devices_used": sorted({str(w.device) for w in workers}),
                                 "way": "all-device" if all_device else "general", "wall_s": wall}
    if _stub_device:
        for w in workers:
            _give_back(w)
        result["way"] = "stub (no device work)"
        return result
    if all_device:
        stats = np.zeros(8, dtype=np.int64)
        removed = np.zeros(2, dtype=np.int64)
        polya: Dict[int, int] = {}
        turned = 0
        for w in workers:
            stats += w.counters.cpu().numpy()
            turned += int(w.rc_count.item())
            removed += w.pre_counts.cpu().numpy()
            if post and post["poly_a"]:
                h = w.polya_hist.cpu().numpy()
                h = np.concatenate([[h[0] + h[_lib.MAX_READ_LEN + 1:].sum()], h[1:_lib.MAX_READ_LEN + 1]])
                for k in np.flatnonzero(h).tolist():
                    polya[k] = polya.get(k, 0) + int(h[k])
        result.update({"reads": int(stats[0]), "with_adapters": int(stats[1]), "bp_in": int(stats[2]),
                       "bp_out": int(stats[3]), "filtered": {"too_short": int(stats[4]), "too_long": int(stats[5])},
                       "too_many_expected_errors": int(stats[7]), "poly_a_trimmed_lengths": polya,
                       "nextseq_trimmed_bases": int(removed[0]), "quality_trimmed_bases": int(removed[1]),
                       "reverse_complemented": turned if opts["revcomp"] else None})
    else:
        total = BatchTrimmer(adapters, device=devices[0], **general_opts)
        for w in workers:
            total.merge(w.trimmer)
        cutter = total.cutter
        result.update({"reads": total.reads, "with_adapters": cutter.with_adapters if cutter else 0,
                       "bp_in": total.bp_in, "bp_out": total.bp_out, "filtered": dict(total.filtered),
                       "too_many_expected_errors": total.too_many_expected_errors,
                       "poly_a_trimmed_lengths": {k: v for k, v in total.poly_a_trimmed_lengths.items() if v},
                       "nextseq_trimmed_bases": total.nextseq_trimmed_bases,
                       "quality_trimmed_bases": total.quality_trimmed_bases,
                       "cutter": cutter, "trimmer": total,
                       "reverse_complemented": total.rc.reverse_complemented if total.rc is not None else None})
    for w in workers:
        w.trimmer = None
        _give_back(w)
    return result


_NUMERIC_RESULT_KEYS = ("reads", "with_adapters", "bp_in", "bp_out", "too_many_expected_errors", "nextseq_trimmed_bases",
                        "quality_trimmed_bases", "bytes_out")


def _process_feeder_main(conn, source, out_path, device_index, rank, options, ranges):
    """One feeder process of ``trim_fastq_gpu(feeder="process")``: its own interpreter, its own HIP context on its GPU,
    the thread-mode pipeline over ITS pieces of the input file; a finished chunk is written at the offset the parent
    sends back for it."""
    try:
        import torch
        torch.cuda.set_device(device_index)
        out_fd = os.open(out_path, os.O_WRONLY) if out_path is not None else None

        def deliver(index, body):
            conn.send(("size", index, len(body)))
            if out_fd is not None:
                tag, idx, offset = conn.recv()
                assert tag == "offset" and idx == index
                view, done = memoryview(body), 0
                while done < len(view):
                    done += os.pwrite(out_fd, view[done:], offset + done)
        deliver.reads_body = out_fd is not None              # (without a sink only the size of a chunk is looked at)
        try:
            res = trim_fastq_gpu(source, None, devices=[device_index], _ranges=ranges, _deliver=deliver, **options)
        finally:
            if out_fd is not None:
                os.close(out_fd)
        slim = {k: res[k] for k in _NUMERIC_RESULT_KEYS if k in res}
        if res.get("reverse_complemented") is not None:
            slim["reverse_complemented"] = int(res["reverse_complemented"])
        for k in ("filtered", "poly_a_trimmed_lengths", "per_device", "devices_used", "way", "wall_s"):
            if k in res:
                slim[k] = res[k]
        conn.send(("done", rank, slim))
    except BaseException as exc:                             # the parent must hear about it, whatever it was
        import traceback
        try:
            conn.send(("error", rank, f"{type(exc).__name__}: {exc}\n{traceback.format_exc()[-1500:]}"))
        except Exception:
            pass
    finally:
        conn.close()


def _trim_fastq_gpu_processes(source, out, options: dict, devices, info_file, repeat: int = 1) -> Dict[str, object]:
    """``trim_fastq_gpu(feeder="process")``: the parent cuts the file into record-aligned pieces (only 1 MiB windows are
    read here), deals them round-robin to one process per GPU and plays the ordered writer WITHOUT touching the data:
    a process reports the size of a finished chunk, the parent answers with the chunk's offset in the output as soon as
    all earlier sizes are known (reference runners.py:224-245, OrderedChunkWriter), the process writes it there itself."""
    import multiprocessing as mp
    from multiprocessing.connection import wait
    if info_file is not None:
        raise ValueError("feeder='process' does not write info files (use feeder='thread')")
    if not _is_plain_file(source):
        raise ValueError("feeder='process' needs a plain (uncompressed) input file: every process reads its own byte ranges")
    if out is not None and not isinstance(out, (str, os.PathLike)):
        raise ValueError("feeder='process' writes to a path (or to nothing): the processes write their chunks themselves")
    devs = _resolve_devices(devices)
    n = len(devs)
    t0 = time.perf_counter()
    ranges = [(i, off, ln, fin) for i, (off, ln, fin) in
              enumerate(list(_file_ranges(source, options["chunk_bytes"])) * max(1, int(repeat)))]
    if out is not None:
        with open(out, "wb"):
            pass                                             # created / truncated here, opened for pwrite by every process
    ctx = mp.get_context("spawn")                            # (a forked child cannot use the parent's HIP context)
    conns, procs = [], []
    for rank, dev in enumerate(devs):
        pc, cc = ctx.Pipe()
        p = ctx.Process(target=_process_feeder_main,
                        args=(cc, str(source), None if out is None else str(out), dev.index if dev.index is not None else 0,
                              rank, options, ranges[rank::n]), daemon=True)
        p.start()
        cc.close()
        conns.append(pc)
        procs.append(p)
    sizes: Dict[int, int] = {}
    results: Dict[int, dict] = {}
    expected, next_offset, error = 0, 0, None
    live = {c: r for r, c in enumerate(conns)}
    try:
        while live and error is None:
            for c in wait(list(live)):
                try:
                    msg = c.recv()
                except EOFError:
                    if live[c] not in results:
                        error = f"feeder process {live[c]} ended without a result"
                    del live[c]
                    continue
                if msg[0] == "size":
                    sizes[msg[1]] = msg[2]
                    while expected in sizes:
                        if out is not None:
                            conns[expected % n].send(("offset", expected, next_offset))
                        next_offset += sizes.pop(expected)
                        expected += 1
                elif msg[0] == "done":
                    results[msg[1]] = msg[2]
                    del live[c]
                else:
                    error = f"feeder process {msg[1]}: {msg[2]}"
    finally:
        for p in procs:
            p.join(timeout=30 if error is None else 1)
            if p.is_alive():
                p.kill()
    if error is not None:
        raise RuntimeError(error)
    if expected != len(ranges):
        raise RuntimeError(f"only {expected} of {len(ranges)} chunks were delivered")
    wall = time.perf_counter() - t0
    total: Dict[str, object] = {k: sum(int(r.get(k, 0)) for r in results.values()) for k in _NUMERIC_RESULT_KEYS}
    if any("reverse_complemented" in r for r in results.values()):
        total["reverse_complemented"] = sum(int(r.get("reverse_complemented", 0)) for r in results.values())
    filtered: Dict[str, int] = {}
    polya: Dict[int, int] = {}
    per_device: Dict[str, dict] = {}
    for rank, r in sorted(results.items()):
        for k, v in (r.get("filtered") or {}).items():
            filtered[k] = filtered.get(k, 0) + int(v)
        for k, v in (r.get("poly_a_trimmed_lengths") or {}).items():
            polya[int(k)] = polya.get(int(k), 0) + int(v)
        for k, v in (r.get("per_device") or {}).items():
            per_device[f"{k} (process {rank})"] = v
    total.update({"filtered": filtered, "poly_a_trimmed_lengths": polya, "per_device": per_device,
                  "devices_used": sorted({d for r in results.values() for d in r.get("devices_used", [])}),
                  "way": next(iter(results.values()))["way"] + ", one feeder process per GPU" if results else "process",
                  "feeder_processes": n, "wall_s": wall})
    return total


class _LineFeedIndex:
    """Line feeds of a block counted by several threads (``cah_fastq_span`` on sub-ranges; ctypes releases the
    interpreter lock), then: how many whole records the block holds and where the first ``n`` of them end -- what one
    sequential ``cah_fastq_span`` over the block answers, from a handful of short calls."""

    MIN_PART = 4 << 20

    def __init__(self, d: np.ndarray, pool: ThreadPoolExecutor, parts: int):
        self.d = d
        step = max(self.MIN_PART, -(-len(d) // max(1, parts)))
        self.bounds = list(range(0, len(d), step)) + [len(d)]
        self.counts = list(pool.map(self._count, zip(self.bounds[:-1], self.bounds[1:])))

    def _span(self, a: int, b: int, limit: int):
        n, used = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.lib().cah_fastq_span(self.d[a:b].ctypes.data if b > a else None, b - a, 0, limit, C.byref(n), C.byref(used)))
        return n.value, used.value

    def _count(self, ab) -> int:
        a, b = ab
        n, used = self._span(a, b, 1 << 62)                  # 4 n line feeds in front of a + used, fewer than 4 behind
        return 4 * n + int(np.count_nonzero(self.d[a + used:b] == 10))

    @property
    def records(self) -> int:
        return sum(self.counts) // 4

    def end_of(self, n: int) -> int:
        """bytes the first n records take (n <= self.records)"""
        if n <= 0:
            return 0
        target, before = 4 * n, 0
        for (a, b), c in zip(zip(self.bounds[:-1], self.bounds[1:]), self.counts):
            if before + c >= target:
                t = target - before                          # the t-th line feed of this part ends record n
                _, used = self._span(a, b, t // 4) if t >= 4 else (0, 0)
                pos = a + used
                for _ in range(t % 4):
                    width = 1 << 12
                    while True:
                        hit = np.flatnonzero(self.d[pos:min(b, pos + width)] == 10)
                        if len(hit):
                            pos += int(hit[0]) + 1
                            break
                        width *= 8
                return pos
            before += c
        raise AssertionError("fewer records in the block than asked for")


def _paired_pieces(source1, source2, chunk_bytes: int, threads: int = 4, pool=None):
    """pairs of raw 4-line-FASTQ pieces with the SAME number of records (the job of dnaio.read_paired_chunks,
    reference runners.py:104-113) -- found by counting line feeds (cah_fastq_span: nothing is parsed): each side is
    read in blocks into a pooled buffer, the side with fewer complete records decides, the surplus of the other side
    is carried over.  The two sides are loaded side by side, and each side's block is read (``preadv`` at explicit
    offsets, plain files) and counted by ``threads`` threads: one thread reads the page cache at a few GB/s and counts
    at ~10, which made this reader the bound of the paired pipeline.
    The pieces are views of pooled buffers: hand them back (``pool.put``; pipeline.POOL unless another pool is given --
    the all-device way passes pinned buffers, _PinnedInputPool) when done."""
    from .pipeline import POOL as _DEFAULT_POOL, _open_maybe_gz
    POOL = pool if pool is not None else _DEFAULT_POOL
    L = _lib.lib()
    plain = [_is_plain_file(source1), _is_plain_file(source2)]
    files = [open(src, "rb", buffering=0) if pl else _open_maybe_gz(src) for src, pl in zip((source1, source2), plain)]
    sizes = [os.fstat(f.fileno()).st_size if pl else None for f, pl in zip(files, plain)]
    offsets = [0, 0]
    carry = [np.zeros(0, np.uint8), np.zeros(0, np.uint8)]
    eof = [False, False]
    threads = max(1, int(threads))
    helpers = ThreadPoolExecutor(max_workers=2 * threads)    # sub-range reads and counts of both sides

    def span(d, final, limit):
        n, used = C.c_int64(0), C.c_int64(0)
        _lib.check(L.cah_fastq_span(d.ctypes.data if len(d) else None, len(d), int(final), limit, C.byref(n), C.byref(used)))
        return n.value, used.value

    def read_range(k, view, offset):
        done = 0
        while done < len(view):
            got = os.preadv(files[k].fileno(), [view[done:]], offset + done)
            if got <= 0:
                raise OSError("short read: the file shrank while it was read")
            done += got

    # Plain files: the NEXT block of a side is read (into a fresh buffer, behind HEAD spare bytes) while the current one is
    # counted and cut -- the carried-over bytes are put in front of it afterwards.  Reading and counting took turns before
    # (profiles/r04/paired_stages.json: the reader alone was the pair pipeline's bound, 25-30 GB/s of 40).
    HEAD = min(4 << 20, max(4096, chunk_bytes))
    fetchers = ThreadPoolExecutor(max_workers=2)
    ahead = [None, None]
    waiting = [0, 0]                                         # whole records in what was carried over

    def fetch(k):
        """-> (buffer, bytes read behind HEAD, that was the file's last block)"""
        want = min(chunk_bytes, sizes[k] - offsets[k])
        if want <= 0:
            return None, 0, True
        buf = POOL.get(HEAD + chunk_bytes)
        view = memoryview(buf)[HEAD:HEAD + want]
        step = max(1 << 20, -(-want // threads))
        list(helpers.map(lambda a: read_range(k, view[a:a + step], offsets[k] + a), range(0, want, step)))
        offsets[k] += want
        return buf, want, offsets[k] >= sizes[k]

    def load_side(k):
        """the next block of file k behind what was carried over, and how many whole records that is"""
        if plain[k] and not eof[k] and waiting[k] > 0 and len(carry[k]) >= chunk_bytes:
            # (this side is a block ahead already, with whole records waiting -- its records are smaller than the other
            # file's: no new block now)
            d = carry[k]
            fill = len(d)
        elif plain[k]:
            buf, want, last = (ahead[k] or fetchers.submit(fetch, k)).result() if not eof[k] else (None, 0, True)
            eof[k] = last
            ahead[k] = None if last else fetchers.submit(fetch, k)
            keep = len(carry[k])
            if buf is None:
                d = carry[k]
            elif keep <= HEAD:
                buf[HEAD - keep:HEAD] = carry[k]
                d = buf[HEAD - keep:HEAD + want]
            else:
                # (the two files' records differ so much in size that one side piles up: copy)
                big = POOL.get(keep + want)
                big[:keep] = carry[k]
                big[keep:keep + want] = buf[HEAD:HEAD + want]
                POOL.put(buf)
                d = big[:keep + want]
            fill = len(d)
        else:
            buf = POOL.get(len(carry[k]) + chunk_bytes)
            fill = len(carry[k])
            buf[:fill] = carry[k]
            if not eof[k]:
                view = memoryview(buf)[fill:fill + chunk_bytes]
                got = files[k].readinto(view) if hasattr(files[k], "readinto") else None
                if got is None:
                    block = files[k].read(chunk_bytes)
                    got = len(block)
                    buf[fill:fill + got] = np.frombuffer(block, dtype=np.uint8)
                if got == 0:
                    eof[k] = True
                fill += got
            d = buf[:fill]
        if fill and d[0] == ord(">"):
            raise ValueError("the device-side path reads 4-line FASTQ; use pipeline.trim_fastq_paired for FASTA")
        if eof[k] or fill < 2 * _LineFeedIndex.MIN_PART:
            return d, span(d, eof[k], 1 << 62), None         # (the end of a file has its own rule: one plain call)
        index = _LineFeedIndex(d, helpers, threads)
        n = index.records
        return d, (n, index.end_of(n)), index

    sides = ThreadPoolExecutor(max_workers=2)                # the two files are read and counted side by side
    try:
        while True:
            loaded = list(sides.map(load_side, (0, 1)))
            data = [x[0] for x in loaded]
            counts = [x[1] for x in loaded]
            used = [c[1] for c in counts]
            counts = [c[0] for c in counts]
            n = min(counts)
            if n == 0:
                if eof[0] and eof[1]:
                    for d in data:
                        POOL.put(d)
                    if len(data[0]) or len(data[1]):
                        if counts[0] != counts[1]:
                            raise ValueError("Reads are improperly paired. There are more reads in one file than in the other.")
                        raise ValueError("FASTQ format error: premature end of file (incomplete record)")
                    return
                # one file is at its end with nothing left while the other still has records (or unread data): the
                # reference fails at once (dnaio.read_paired_chunks); reading the longer file to ITS end first would
                # pull all of it into memory, recopied chunk after chunk
                for k in (0, 1):
                    if eof[k] and counts[k] == 0 and len(data[k]) == 0 and (counts[1 - k] > 0 or len(data[1 - k]) or not eof[1 - k]):
                        for d in data:
                            POOL.put(d)
                        raise ValueError("Reads are improperly paired. There are more reads in one file than in the other.")
                if any(len(d) > 64 * chunk_bytes for k, d in enumerate(data) if not eof[k]):
                    for d in data:
                        POOL.put(d)
                    raise ValueError("record larger than 64 chunks: not a FASTQ file?")
                carry = [d.copy() for d in data]
                waiting = list(counts)
                for d in data:
                    POOL.put(d)
                continue
            cuts = []
            for k, d in enumerate(data):
                if counts[k] == n:
                    cuts.append(used[k])
                elif loaded[k][2] is not None:
                    cuts.append(loaded[k][2].end_of(n))
                else:
                    cuts.append(span(d, eof[k], n)[1])
            carry = [data[0][cuts[0]:].copy(), data[1][cuts[1]:].copy()]
            waiting = [counts[0] - n, counts[1] - n]
            yield data[0][:cuts[0]], data[1][:cuts[1]]
    finally:
        for fut in ahead:
            if fut is not None:
                try:
                    POOL.put(fut.result()[0])
                except Exception:
                    pass
        sides.shutdown(wait=True)
        fetchers.shutdown(wait=True)
        helpers.shutdown(wait=True)
        for f, src in zip(files, (source1, source2)):
            if f is not src:
                f.close()


def _mate_all_device(opts: Optional[dict]):
    """BatchTrimmer options of one mate -> (adapters, pre, post) if the all-device way can serve them, else None"""
    o = dict(opts or {})
    adapters = _adapter_list(o.pop("adapters", ()))
    times = int(o.pop("times", 1))
    if times < 1 or o.pop("action", "trim") != "trim" or o.pop("revcomp", False):
        return None
    index = o.pop("index", True)
    o.pop("rc_suffix", None)
    cut = [int(c) for c in o.pop("cut", ()) if int(c) != 0]
    nextseq, qcut, qbase = o.pop("nextseq_trim", None), o.pop("quality_cutoff", None), o.pop("quality_base", 33)
    poly_a, head = o.pop("poly_a", False), o.pop("poly_a_revcomp", False)
    length, max_ee = o.pop("length", None), o.pop("max_expected_errors", None)
    if o:                                                    # an option this way does not know
        return None
    if adapters and not _all_device_adapters(adapters, times, index):
        return None                                          # (rightmost parts, adapters regrouped behind an index, ...)
    if len(cut) > 2 or (len(cut) == 2 and cut[0] * cut[1] > 0):
        return None                                          # (BatchTrimmer raises the reference's error for these)
    pre = post = None
    if cut or nextseq is not None or qcut is not None:
        pre = {"cut": cut, "nextseq_trim": nextseq, "quality_cutoff": qcut, "quality_base": qbase}
    if poly_a or length is not None or max_ee is not None:
        post = {"poly_a": bool(poly_a), "poly_a_revcomp": bool(head), "length": length, "max_expected_errors": max_ee}
    return adapters, pre, post, times


def _paired_all_device(in1, in2, out1, out2, mates, job, discard_untrimmed, discard_trimmed, chunk_bytes, threads, devices):
    """Read pairs on the all-device way: both mates' chunks are indexed, trimmed and matched by ``_Worker.modify`` (no
    filter), the pair filter (reference PairedEndFilter / cli.py:735-912, as pipeline.filter_reads combines it) is a
    dozen element-wise operations on the two mates' intervals, both outputs are formatted on the device with the
    shared keep flags.  ``job``: a PairedJob built from the same options (its filter modes are used; nothing is run
    through it)."""
    import torch
    min_len, max_len, mode, untrimmed_mode = job.min_len, job.max_len, job.mode, job.untrimmed_mode
    plans = []
    for adapters, pre, post, times in mates:
        rev = bool(adapters) and all(isinstance(a, SingleAdapter) and a._reverse_reads for a in adapters)
        plans.append((_plan_for(adapters) if adapters else (None, [0])) + (times, rev))

    def make_worker(dev, slot):
        ws = []
        for (plan, kinds, times, rev) in plans:
            ws.append(_take_worker(plan, kinds, dev, {"times": times, "reversed": rev}))
        w, mate = ws
        # one stream for the pair.  What _take_worker queued on the mate's own stream (its counters' zeroing) must be
        # through before anything of the pair's stream touches them, and the pair's counters are born on that stream
        w.stream.wait_stream(mate.stream)
        mate.stream = w.stream
        w.mate = mate
        with torch.cuda.stream(w.stream):
            w.pair_counts = torch.zeros(8, dtype=torch.int64, device=w.device)   # kept, too short, too long, too many ee, bp out 1 / 2
        w.stage_s = [0.0, 0.0, 0.0]        # host seconds in: load (copy to HBM, line count, index) / trim + match + pair filter (queued) / format + copy back
        return w

    def combine(preds, how):
        preds = [p for p in preds if p is not None]
        if not preds:
            return None
        if how == "first":
            return preds[0]
        out = preds[0]
        for p in preds[1:]:
            out = (out | p) if how == "any" else (out & p)
        return out

    def work(w: _Worker, d1, d2):
        if len(d1) == 0:
            return b"", b"", [], w
        torch.cuda.set_device(w.device)
        t0 = time.perf_counter()
        try:
            with torch.cuda.stream(w.stream):
                ws = (w, w.mate)
                # (the reader filled pinned buffers: they are the device copies' sources as they are)
                n = w.load(_PINNED_INPUT.tensor_of(d1))
                if w.mate.load(_PINNED_INPUT.tensor_of(d2)) != n:
                    raise ValueError("Reads are improperly paired")
                t1_ = time.perf_counter()
                w.stage_s[0] += t1_ - t0
                if n == 0:
                    return b"", b"", [], w
                ee = [ww.modify(n, pre, post, None) for ww, (_, pre, post, _t) in zip(ws, mates)]
                lens = [ww.end[:n] - ww.beg[:n] for ww in ws]
                found = [ww.res.status[:n] == 1 for ww in ws]
                keep = torch.ones(n, dtype=torch.bool, device=w.device)
                short = combine([None if m is None else (l < int(m)) for l, m in zip(lens, min_len)], mode)
                long_ = combine([None if m is None else (l > int(m)) for l, m in zip(lens, max_len)], mode)
                many = combine([None if (e is None) else (e > float(post["max_expected_errors"]))
                                for e, (_, pre, post, _t) in zip(ee, mates)], mode)
                for slot, pred in ((1, short), (2, long_), (3, many)):
                    if pred is not None:
                        w.pair_counts[slot] += (keep & pred).sum()
                        keep = keep & ~pred
                if discard_trimmed:
                    keep = keep & ~combine(found, mode)
                elif discard_untrimmed:
                    keep = keep & ~combine([~f for f in found], untrimmed_mode)
                w.pair_counts[0] += keep.sum()
                for k, ww in enumerate(ws):
                    w.pair_counts[4 + k] += (lens[k] * keep).sum()
                    ww.keep[:n].copy_(keep.to(torch.uint8))
                t2_ = time.perf_counter()
                w.stage_s[1] += t2_ - t1_
                h1, t1 = w.finish(d1, n, w.n_bytes)
                h2, t2 = w.mate.finish(d2, n, w.mate.n_bytes)
                w.stage_s[2] += time.perf_counter() - t2_
                return memoryview(h1.numpy())[:t1], memoryview(h2.numpy())[:t2], [h1, h2], w
        finally:
            w.busy_s += time.perf_counter() - t0
            w.chunks += 1
            w.bytes_in += len(d1) + len(d2)
            _PINNED_INPUT.put(d1)
            _PINNED_INPUT.put(d2)                            # the reader's buffers are free again

    feeders = [_Feeder(dev, threads, len(devices) > 1, make_worker) for dev in devices]
    o1 = out1 if hasattr(out1, "write") else open(out1, "wb")
    o2 = out2 if hasattr(out2, "write") else open(out2, "wb")
    t_start = time.perf_counter()
    try:
        pending: deque = deque()

        def drain(limit: int) -> None:
            while len(pending) > limit:
                b1, b2, bufs, w = pending.popleft().result()
                o1.write(b1)
                o2.write(b2)
                b1 = b2 = None
                for b in bufs:
                    w.pool.put(b)
        reader_s = drain_s = 0.0
        pieces = _paired_pieces(in1, in2, chunk_bytes, pool=_PINNED_INPUT)
        i = 0
        while True:
            t_a = time.perf_counter()
            piece = next(pieces, None)
            t_b = time.perf_counter()
            reader_s += t_b - t_a
            if piece is None:
                break
            pending.append(feeders[i % len(feeders)].submit(work, *piece))
            i += 1
            drain(2 * threads * len(feeders))
            drain_s += time.perf_counter() - t_b
        t_b = time.perf_counter()
        drain(0)
        drain_s += time.perf_counter() - t_b
    finally:
        for f in feeders:
            f.close()
        if o1 is not out1:
            o1.close()
        if o2 is not out2:
            o2.close()
        _PINNED_INPUT.trim()
    wall = time.perf_counter() - t_start
    workers = [w for f in feeders for w in f.workers if w is not None]
    pc = np.zeros(8, dtype=np.int64)
    c = [np.zeros(8, dtype=np.int64), np.zeros(8, dtype=np.int64)]
    removed = [np.zeros(2, dtype=np.int64), np.zeros(2, dtype=np.int64)]
    for w in workers:
        pc += w.pair_counts.cpu().numpy()
        for k, ww in enumerate((w, w.mate)):
            c[k] += ww.counters.cpu().numpy()
            removed[k] += ww.pre_counts.cpu().numpy()
    result = {"pairs": int(c[0][0]), "pairs_written": int(pc[0]), "trimmers": None, "paired_cutter": None,
              "filtered": {k: v for k, v in (("too_short", int(pc[1])), ("too_long", int(pc[2]))) if v},
              "too_many_expected_errors": int(pc[3]), "with_adapters": (int(c[0][1]), int(c[1][1])),
              "bp_in": (int(c[0][2]), int(c[1][2])), "bp_out": (int(pc[4]), int(pc[5])),
              "quality_trimmed_bases": (int(removed[0][1]), int(removed[1][1])),
              "nextseq_trimmed_bases": (int(removed[0][0]), int(removed[1][0])), "reverse_complemented": None,
              "devices_used": sorted({str(w.device) for w in workers}), "per_device": _per_device(feeders, wall),
              "way": "all-device",
              # where the wall time goes: the main thread alternates between the reader (both files read and cut into
              # pieces of equal record counts) and handing over / writing results; the workers' host seconds by stage
              "stages": {"wall_s": wall, "reader_s": reader_s, "submit_and_write_s": drain_s, "workers": len(workers),
                         "worker_load_s": sum(w.stage_s[0] for w in workers), "worker_trim_s": sum(w.stage_s[1] for w in workers),
                         "worker_format_s": sum(w.stage_s[2] for w in workers)}}
    for w in workers:
        mate, w.mate = w.mate, None
        mate.stream = torch.cuda.Stream(device=mate.device)      # (it shared its partner's)
        _give_back(mate)
        _give_back(w)
    return result


def trim_fastq_gpu_paired(in1, in2, out1, out2, r1: Optional[dict] = None, r2: Optional[dict] = None,
                          pair_filter: Optional[str] = None, minimum_length=None, maximum_length=None,
                          discard_untrimmed: bool = False, discard_trimmed: bool = False,
                          chunk_bytes: int = DEFAULT_GPU_CHUNK_BYTES, threads: int = 2, devices=None,
                          pair_adapters: bool = False, revcomp: bool = False,
                          rc_suffix: Optional[str] = " rc") -> Dict[str, object]:
    """``pipeline.trim_fastq_paired`` (same arguments and result) with both mates' chunks indexed and formatted on
    the GPU(s): a worker holds a pair of chunks (two raw buffers in HBM, one stream).  When both mates' options are
    of the all-device kind (``_mate_all_device``: single non-rightmost adapters with ``times`` rounds or one linked
    adapter, action ``trim``, the simple modifiers; no ``pair_adapters`` / ``revcomp``) nothing per read touches the
    host (``_paired_all_device``; result key ``way`` = "all-device"); otherwise the worker runs
    ``PairedJob.process_pair`` on the two device chunks (``way`` = "general").  The writer keeps both outputs in chunk
    order.  FASTA input goes to the host-parsed pipeline."""
    import torch
    from .pipeline import PairedJob, trim_fastq_paired
    if _is_fasta(in1) or _is_fasta(in2):
        return trim_fastq_paired(in1, in2, out1, out2, r1, r2, pair_filter, minimum_length, maximum_length,
                                 discard_untrimmed, discard_trimmed, device=_resolve_devices(devices)[0],
                                 pair_adapters=pair_adapters, revcomp=revcomp, rc_suffix=rc_suffix)
    devices = _resolve_devices(devices)
    threads = max(1, int(threads))

    def make_job(dev):
        return PairedJob(r1, r2, pair_filter, minimum_length, maximum_length, discard_untrimmed, discard_trimmed, dev,
                         pair_adapters, revcomp, rc_suffix)

    total = make_job(devices[0])                             # option errors surface here, not in a worker
    r2_eff = dict(r2 or {})
    if r2_eff.get("poly_a"):
        r2_eff.setdefault("poly_a_revcomp", True)            # --poly-a on read pairs: the poly-T head of R2
    mates = None if (pair_adapters or revcomp) else (_mate_all_device(r1), _mate_all_device(r2_eff))
    if mates is not None and all(m is not None for m in mates):
        return _paired_all_device(in1, in2, out1, out2, mates, total, discard_untrimmed, discard_trimmed, chunk_bytes,
                                  threads, devices)

    def make_worker(dev, slot):
        w = _take_worker(None, [], dev, {})
        w.mate = _take_worker(None, [], dev, {})
        w.mate.stream = w.stream                             # one stream for the pair
        w.job = make_job(dev)
        return w

    def work(w: _Worker, d1, d2):
        if len(d1) == 0:
            return b"", b"", [], w
        torch.cuda.set_device(w.device)
        t0 = time.perf_counter()
        try:
            with torch.cuda.stream(w.stream):
                chunks = []
                for ww, d in ((w, d1), (w.mate, d2)):
                    n = ww.load(d)
                    ww.check_index()
                    chunks.append(DeviceFastqChunk(ww, d, n, ww.n_bytes))
                b1, b2 = w.job.process_pair(chunks[0], chunks[1])
                return b1, b2, chunks[0].out_bufs + chunks[1].out_bufs, w
        finally:
            w.busy_s += time.perf_counter() - t0
            w.chunks += 1
            w.bytes_in += len(d1) + len(d2)
            _PINNED_INPUT.put(d1)
            _PINNED_INPUT.put(d2)                            # the reader's (pinned) buffers are free again

    feeders = [_Feeder(dev, threads, len(devices) > 1, make_worker) for dev in devices]
    o1 = out1 if hasattr(out1, "write") else open(out1, "wb")
    o2 = out2 if hasattr(out2, "write") else open(out2, "wb")
    t_start = time.perf_counter()
    try:
        pending: deque = deque()

        def drain(limit: int) -> None:
            while len(pending) > limit:
                b1, b2, bufs, w = pending.popleft().result()
                o1.write(b1)
                o2.write(b2)
                b1 = b2 = None
                for b in bufs:
                    w.pool.put(b)
        reader_s = drain_s = 0.0
        pieces = _paired_pieces(in1, in2, chunk_bytes, pool=_PINNED_INPUT)
        i = 0
        while True:
            t_a = time.perf_counter()
            piece = next(pieces, None)
            t_b = time.perf_counter()
            reader_s += t_b - t_a
            if piece is None:
                break
            pending.append(feeders[i % len(feeders)].submit(work, *piece))
            i += 1
            drain(2 * threads * len(feeders))
            drain_s += time.perf_counter() - t_b
        t_b = time.perf_counter()
        drain(0)
        drain_s += time.perf_counter() - t_b
    finally:
        for f in feeders:
            f.close()
        if o1 is not out1:
            o1.close()
        if o2 is not out2:
            o2.close()
        _PINNED_INPUT.trim()
    wall = time.perf_counter() - t_start
    workers = [w for f in feeders for w in f.workers if w is not None]
    for w in workers:
        total.merge(w.job)
    result = total.result()
    result["devices_used"] = sorted({str(w.device) for w in workers})
    result["per_device"] = _per_device(feeders, wall)
    result["way"] = "general"
    for w in workers:
        mate, w.mate, w.job = w.mate, None, None
        mate.stream = torch.cuda.Stream(device=mate.device)      # (it shared its partner's)
        _give_back(mate)
        _give_back(w)
    return result
