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

Scope of this stage: single-end 4-line FASTQ, any number of single (not linked, not rightmost) adapters of
every type, ``--times 1``, action ``trim``, ``--discard-trimmed`` / ``--discard-untrimmed``, ``-m`` / ``-M``.
Everything else (FASTA, info files, other actions, quality trimming in the same pass ...) is served by
``pipeline.trim_fastq``; ``trim_fastq_gpu`` refuses such options instead of silently doing something else.
"""
import ctypes as C
import threading
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from typing import BinaryIO, Dict, List, Optional, Sequence, Union

import numpy as np

from . import _lib
from .adapters import AnywhereAdapter, MultipleAdapters, SingleAdapter

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
        self._ws = None
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

    def run(self, data, is_final: bool):
        """data: uint8 numpy array or torch tensor on the host holding whole records -> (pinned buffer, bytes)"""
        torch = self.torch
        L = _lib.lib()
        n_bytes = int(len(data))
        if n_bytes == 0:
            return None, 0
        torch.cuda.set_device(self.device)
        self._ensure(n_bytes)
        sp = self.stream.cuda_stream
        with torch.cuda.stream(self.stream):
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
            # ---- step 3: match the reads in place, decide what is kept ------------------------------------------
            o = self.opts
            if n:
                ws_need = int(L.cah_plan_workspace_bytes(self.plan.handle, n))
                if self._ws is None or self._ws.numel() < ws_need:
                    self._ws = torch.empty(ws_need + ws_need // 4, dtype=torch.uint8, device=self.device)
                _lib.check(L.cah_match_batch(self.plan.handle, self.d_in.data_ptr(), self.seq_off.data_ptr(),
                                             self.seq_len.data_ptr(), n, self.res.out6.data_ptr(),
                                             self.res.best_adapter.data_ptr(), self.res.status.data_ptr(),
                                             self._ws.data_ptr(), self._ws.numel(), sp))
                _lib.check(L.cah_trim_decide_device(
                    self.res.out6.data_ptr(), self.res.status.data_ptr(), self.res.best_adapter.data_ptr(),
                    self.seq_len.data_ptr(), n, self.kinds.data_ptr(),
                    -1 if o["minimum_length"] is None else int(o["minimum_length"]),
                    -1 if o["maximum_length"] is None else int(o["maximum_length"]),
                    int(bool(o["discard_trimmed"])), int(bool(o["discard_untrimmed"])),
                    self.beg.data_ptr(), self.end.data_ptr(), self.keep.data_ptr(), self.counters.data_ptr(), sp))
            if o.get("assemble") == "host":
                return self._assemble_on_host(data, n_bytes, n)
            # ---- step 4: format on the device, bring the bytes back ---------------------------------------------
            _lib.check(L.cah_fastq_format_device(self.d_in.data_ptr(), self.rec6.data_ptr(), n, self.beg.data_ptr(),
                                                 self.end.data_ptr(), self.keep.data_ptr(), self.d_scratch.data_ptr(),
                                                 self.d_scratch.numel(), n_bytes, self.d_out.data_ptr(),
                                                 self.d_out.numel(), self.d_info.data_ptr(), sp))
            self.d_info[4:5].copy_(self.counters[6:7], non_blocking=True)
            self.h_info.copy_(self.d_info, non_blocking=True)
            self.stream.synchronize()
            err = int(self.h_info[1])
            if err != -1:
                code, record = err & 0xFF, (err >> 8) - 1
                what = {1: "line expected to start with '@'", 2: "third line expected to start with '+'",
                        3: "length of sequence and qualities differ"}.get(code, "malformed record")
                raise ValueError(f"FASTQ format error in record {record} of the chunk: {what}")
            if int(self.h_info[4]) != 0:
                _lib.raise_invalid_reads(int(self.seq_len[:max(n, 1)].max().item()))
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
        self.h_info.copy_(self.d_info, non_blocking=True)
        if n:
            self._h_rec6[:n].copy_(self.rec6[:n], non_blocking=True)
            self._h_beg[:n].copy_(self.beg[:n], non_blocking=True)
            self._h_end[:n].copy_(self.end[:n], non_blocking=True)
            self._h_keep[:n].copy_(self.keep[:n], non_blocking=True)
        self.stream.synchronize()
        err = int(self.h_info[1])
        if err != -1:
            code, record = err & 0xFF, (err >> 8) - 1
            what = {1: "line expected to start with '@'", 2: "third line expected to start with '+'",
                    3: "length of sequence and qualities differ"}.get(code, "malformed record")
            raise ValueError(f"FASTQ format error in record {record} of the chunk: {what}")
        if int(self.h_info[4]) != 0:
            _lib.raise_invalid_reads(int(self.seq_len[:max(n, 1)].max().item()))
        h_out = self.pool.get(n_bytes + 4 * n + 64)
        out_len = C.c_int64(0)
        src_ptr = data.data_ptr() if isinstance(data, torch.Tensor) else data.ctypes.data
        _lib.check(L.cah_fastq_write_trimmed(src_ptr, self._h_rec6.data_ptr(), n, self._h_beg.data_ptr(),
                                             self._h_end.data_ptr(), self._h_keep.data_ptr(), h_out.data_ptr(),
                                             h_out.numel(), C.byref(out_len)))
        return h_out, int(out_len.value)


# buffers outlive a call: pinning host memory and growing device buffers cost tens of milliseconds, more than a
# whole chunk -- workers (device buffers, stream) and pinned output buffers are recycled between calls
_PINNED = _PinnedPool()
_IDLE_WORKERS: Dict[str, list] = {}
_IDLE_LOCK = threading.Lock()


def _take_worker(plan, kinds, dev, opts) -> "_Worker":
    import torch
    key = str(torch.device(dev))
    with _IDLE_LOCK:
        idle = _IDLE_WORKERS.get(key, [])
        w = idle.pop() if idle else None
    if w is None:
        return _Worker(plan, kinds, dev, opts, _PINNED)
    torch.cuda.set_device(w.device)
    w.plan, w.opts = plan, opts
    with torch.cuda.stream(w.stream):                           # ordered in front of the worker's next kernels
        w.kinds = torch.tensor(kinds, dtype=torch.uint8, device=w.device)
        w.counters.zero_()
    w._ws = None
    return w


def _give_back(w: "_Worker") -> None:
    with _IDLE_LOCK:
        idle = _IDLE_WORKERS.setdefault(str(w.device), [])
        if len(idle) < 16:
            idle.append(w)


def _plan_for(adapters):
    if isinstance(adapters, MultipleAdapters):
        adapters = list(adapters._adapters)
    elif isinstance(adapters, SingleAdapter):
        adapters = [adapters]
    adapters = list(adapters)
    if not adapters:
        raise ValueError("trim_fastq_gpu needs at least one adapter")
    for a in adapters:
        if not isinstance(a, SingleAdapter) or a._reverse_reads:
            raise ValueError("the device-side FASTQ path takes single, non-rightmost adapters; use pipeline.trim_fastq")
    plan = a._fused_plan if len(adapters) == 1 else _lib.Plan([a.matcher_spec() for a in adapters])
    kinds = [2 if isinstance(a, AnywhereAdapter) else (1 if a._remove_before else 0) for a in adapters]
    return adapters, plan, kinds


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


def trim_fastq_gpu(source: Union[str, BinaryIO, np.ndarray], out: Union[str, BinaryIO, None], adapters,
                   discard_untrimmed: bool = False, discard_trimmed: bool = False,
                   minimum_length: Optional[int] = None, maximum_length: Optional[int] = None,
                   chunk_bytes: int = DEFAULT_GPU_CHUNK_BYTES, threads: int = 3, devices=None,
                   assemble: str = "device") -> Dict[str, object]:
    """``cutadapt <adapter options> [-m N] [-M N] [--discard-(un)trimmed] -o out in.fastq`` with the records
    indexed, matched, filtered and formatted on the GPU(s).  ``out``: path, binary file, or None (the output is
    produced and counted but not kept: measures the pipeline without a sink).  ``threads`` workers, each with
    its own stream and pinned buffers, dealt round-robin over ``devices`` (list of indices, "all", or None =
    current device).  ``assemble``: "device" formats the trimmed records on the GPU and brings the bytes back;
    "host" brings back only the record index and kept intervals and copies the records together on the worker's
    host core (same bytes; trades the outbound PCIe traffic for host memcpy); "mixed" lets every other worker do
    that.  Returns the reference's counters
    (report.py:62-80)."""
    import torch
    if assemble not in ("device", "host", "mixed", "mixed3"):
        raise ValueError("assemble must be 'device', 'host', 'mixed' or 'mixed3'")
    adapters, plan, kinds = _plan_for(adapters)
    pinned = _PINNED
    opts = {"discard_untrimmed": discard_untrimmed, "discard_trimmed": discard_trimmed,
            "minimum_length": minimum_length, "maximum_length": maximum_length, "assemble": assemble}
    if devices == "all":
        devices = list(range(torch.cuda.device_count()))
    elif devices is None:
        devices = [torch.cuda.current_device()]
    devices = [torch.device("cuda", d) if isinstance(d, int) else torch.device(d) for d in devices]
    threads = max(1, int(threads))
    local = threading.local()
    workers: List[_Worker] = []
    lock = threading.Lock()

    def work(data, is_final):
        if not hasattr(local, "w"):
            with lock:
                dev = devices[len(workers) % len(devices)]
                workers.append(None)
                slot = len(workers) - 1
            # "mixed": every other worker assembles on its host core -- the outbound PCIe traffic and the host's
            # memcpy work are both halved (measured best when neither direction of the link is to be the bound)
            # ("mixed3": two of three workers)
            wopts = opts
            if assemble == "mixed":
                wopts = dict(opts, assemble="host" if slot % 2 else "device")
            elif assemble == "mixed3":
                wopts = dict(opts, assemble="host" if slot % 3 else "device")
            local.w = _take_worker(plan, kinds, dev, wopts)
            with lock:
                workers[slot] = local.w
        res = local.w.run(data, is_final)
        if from_file:
            from .pipeline import POOL
            POOL.put(data)                                  # the reader's buffer is free again
        return res

    from_file = not isinstance(source, (np.ndarray, torch.Tensor))
    sink = None if out is None else (out if hasattr(out, "write") else open(out, "wb"))
    bytes_out = 0
    try:
        pending: deque = deque()
        with ThreadPoolExecutor(max_workers=threads) as pool:
            def drain(limit: int) -> None:
                nonlocal bytes_out
                while len(pending) > limit:
                    buf, total = pending.popleft().result()
                    bytes_out += total
                    if buf is not None:
                        if sink is not None and total:
                            sink.write(memoryview(buf.numpy())[:total])      # straight from the pinned buffer
                        pinned.put(buf)
            for data, is_final in _chunks(source, chunk_bytes):
                pending.append(pool.submit(work, data, is_final))
                drain(2 * threads)
            drain(0)
    finally:
        if sink is not None and sink is not out:
            sink.close()
    stats = np.zeros(8, dtype=np.int64)
    for w in workers:
        if w is not None:
            stats += w.counters.cpu().numpy()
            _give_back(w)
    return {"reads": int(stats[0]), "with_adapters": int(stats[1]), "bp_in": int(stats[2]), "bp_out": int(stats[3]),
            "bytes_out": int(bytes_out), "filtered": {"too_short": int(stats[4]), "too_long": int(stats[5])},
            "devices_used": sorted({str(w.device) for w in workers if w is not None})}
