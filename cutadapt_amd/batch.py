"""Packed read batches resident in HBM, and batch results.

The reference hands reads to the matcher one Python ``str`` at a time
(reference src/cutadapt/pipeline.py:60-69 -> modifiers.py:200-261 -> adapters.py:815-832).
The GPU path works on *batches*: all sequences of a chunk back to back in one uint8 buffer
plus an int64 offsets array (the same shape a 4 MiB FASTQ chunk of
reference runners.py:116-126 has once its sequence lines are concatenated).

PyTorch is used only as the owner of device memory and streams; the kernels are reached
through the C ABI with raw pointers.
"""
import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib


def _torch():
    import torch
    return torch


def _stream_ptr() -> int:
    return int(_torch().cuda.current_stream().cuda_stream)


def pack_strings(reads: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    """list[str] -> (uint8[total], int64[n+1]).  Non-ASCII characters raise ValueError like
    the reference's translate() (reference _align.pyx:44-45)."""
    encoded = []
    for r in reads:
        if not isinstance(r, str):
            raise TypeError(f"sequence must be str, not {type(r).__name__}")
        try:
            encoded.append(r.encode("ascii"))
        except UnicodeEncodeError:
            raise ValueError("String must contain only ASCII characters")
    offsets = np.zeros(len(encoded) + 1, dtype=np.int64)
    if encoded:
        np.cumsum([len(e) for e in encoded], out=offsets[1:])
    seqs = np.frombuffer(b"".join(encoded), dtype=np.uint8)
    return seqs, offsets


class ReadBatch:
    """n reads packed in HBM: ``seqs`` uint8[total], ``offsets`` int64[n+1] (or int64[n] plus
    ``lens`` int32[n] for sub-sequence views)."""

    def __init__(self, seqs, offsets, lens=None, n_reads: Optional[int] = None, validated: bool = False,
                 uniform_len: Optional[int] = None):
        torch = _torch()
        if not (seqs.is_cuda and offsets.is_cuda):
            raise ValueError("ReadBatch needs CUDA/HIP tensors; use from_strings()/from_host()")
        if seqs.dtype != torch.uint8 or offsets.dtype != torch.int64:
            raise TypeError("seqs must be uint8 and offsets int64")
        if lens is not None and lens.dtype != torch.int32:
            raise TypeError("lens must be int32")
        self.seqs = seqs.contiguous()
        self.offsets = offsets.contiguous()
        self.lens = None if lens is None else lens.contiguous()
        if n_reads is None:
            n_reads = int(lens.numel()) if lens is not None else int(offsets.numel()) - 1
        self.n_reads = int(n_reads)
        self.validated = validated
        # every read has this length and read 0 starts at seqs[0] (None: not known): the batch then takes
        # cah_match_batch_uniform -- no offsets array is read on the device
        self.uniform_len = None if (uniform_len is None or lens is not None) else int(uniform_len)
        self._workspace = None

    # ---- constructors ---------------------------------------------------------------------
    @classmethod
    def from_host(cls, seqs: np.ndarray, offsets: np.ndarray, device=None, validated: bool = False):
        torch = _torch()
        device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        # .to(device) copies synchronously out of the (pageable) host arrays: no defensive host copy
        s_host = np.ascontiguousarray(seqs, dtype=np.uint8)
        o_host = np.ascontiguousarray(offsets, dtype=np.int64)
        if len(s_host) == 0:
            s_host = np.zeros(1, dtype=np.uint8)          # an empty batch still needs a valid pointer
        if not s_host.flags.writeable:
            s_host = s_host.copy()                        # torch.from_numpy wants a writeable array
        if not o_host.flags.writeable:
            o_host = o_host.copy()
        s = torch.from_numpy(s_host).to(device)
        o = torch.from_numpy(o_host).to(device)
        # equally long reads starting at byte 0 (one vectorised look at the host offsets)
        uniform = None
        if len(o_host) >= 2 and o_host[0] == 0:
            step = int(o_host[1])
            if step >= 1 and int(o_host[-1]) == step * (len(o_host) - 1) and bool((np.diff(o_host) == step).all()):
                uniform = step
        b = cls(s, o, validated=validated, uniform_len=uniform)
        if len(o_host) >= 2:
            b.max_len = int(np.diff(o_host).max())            # (known here for free: match_batch's frame length, no device look-up)
        return b

    @classmethod
    def from_strings(cls, reads: Sequence[str], device=None):
        seqs, offsets = pack_strings(reads)      # str.encode already rejected non-ASCII
        return cls.from_host(seqs, offsets, device=device, validated=True)

    @classmethod
    def synthetic(cls, n_reads: int, read_len: int, adapters: Sequence[str], seed: int = 2,
                  first_index: int = 0, p_adapter: float = 0.25, p_edit: float = 0.02,
                  p_n: float = 0.005, device=None):
        """Generate the synthetic workload of SURVEY.md section 8(d) directly in HBM."""
        torch = _torch()
        device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        seqs = torch.empty(n_reads * read_len, dtype=torch.uint8, device=device)
        offsets = torch.empty(n_reads + 1, dtype=torch.int64, device=device)
        ads = [a.encode("ascii") for a in adapters]
        off = (C.c_int32 * (len(ads) + 1))()
        for i, a in enumerate(ads):
            off[i + 1] = off[i] + len(a)
        with torch.cuda.device(device):
            _lib.check(_lib.lib().cah_synth_reads(
                seed, first_index, n_reads, read_len, prob_u32(p_adapter), prob_u32(p_edit),
                prob_u16(p_n), b"".join(ads), off, len(ads), seqs.data_ptr(), offsets.data_ptr(),
                _stream_ptr()))
        return cls(seqs, offsets, validated=True, uniform_len=read_len if read_len >= 1 else None)

    # ---- helpers ----------------------------------------------------------------------------
    def __len__(self):
        return self.n_reads

    @property
    def device(self):
        return self.seqs.device

    def _lens_ptr(self):
        return self.lens.data_ptr() if self.lens is not None else None

    def workspace(self, plan=None):
        """Per-batch device scratch for the work queue / counters (never shared between
        batches, so batches on different streams do not interfere).  With ``plan`` the scratch is sized
        for that plan (cah_plan_workspace_bytes: the fused multi-adapter path needs more)."""
        torch = _torch()
        if plan is not None:
            need = int(_lib.lib().cah_plan_workspace_bytes(plan.handle, self.n_reads))
        else:
            need = int(_lib.lib().cah_workspace_bytes(self.n_reads))
        if self._workspace is None or (self._workspace.numel() < need and not getattr(self, "_ws_is_fallback", False)):
            try:
                self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
            except RuntimeError:
                # the plan-sized scratch (fused multi-adapter path: up to ~20 GB) does not fit beside the batch: the base
                # size does, and cah_match_batch then matches one adapter after the other (same results)
                base = int(_lib.lib().cah_workspace_bytes(self.n_reads))
                if plan is None or base >= need:
                    raise
                torch.cuda.empty_cache()
                self._workspace = torch.empty(base, dtype=torch.uint8, device=self.device)
                self._ws_is_fallback = True
        return self._workspace

    def validate_ascii(self) -> None:
        """Raise ValueError if any read holds a byte >= 0x80 (the reference rejects such
        strings before matching, reference _align.pyx:44-45, _kmer_finder.pyx:182-183)."""
        if self.validated or self.n_reads == 0:
            self.validated = True
            return
        torch = _torch()
        bad = torch.zeros(1, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().cah_validate_ascii_batch(
                self.seqs.data_ptr(), self.offsets.data_ptr(), self._lens_ptr(), self.n_reads,
                bad.data_ptr(), _stream_ptr()))
        if int(bad.item()) != 0:
            raise ValueError("String must contain only ASCII characters")
        self.validated = True

    def view(self, starts, lens, check: bool = True) -> "ReadBatch":
        """Sub-sequence view: read r becomes seqs[offsets[r]+starts[r] : ... + lens[r]).
        Used for the second stage of linked adapters (reference adapters.py:1222-1224).
        ``check``: views of a uniform batch are streamed by the library (cah_match_batch_views), which needs every view to
        lie inside its read -- looked at here once, on the device (one synchronisation); a batch with a view that does
        not takes the plain-views entry point instead (round-5 advisor: the streaming prefilter clamps such a view, the
        aligner does not).  check=False: the caller vouches for 0 <= starts, starts + lens <= read length."""
        torch = _torch()
        base = self.offsets[: self.n_reads]
        v = ReadBatch(self.seqs, base + starts.to(torch.int64), lens.to(torch.int32),
                      n_reads=self.n_reads, validated=self.validated)
        v._workspace = self._workspace       # same reads, same stream order: the scratch can be shared
        if self.uniform_len and self.lens is None:
            inside = True
            if check and self.n_reads:
                inside = bool(((starts >= 0) & (lens >= 0) & (starts + lens <= int(self.uniform_len))).all().item())
            if inside:
                # views inside the reads of a sequencer's batch: the library streams the parent's reads instead of
                # fetching ragged views (cah_match_batch_views)
                v.within_uniform = int(self.uniform_len)
        return v

    def lengths(self):
        torch = _torch()
        if self.lens is not None:
            return self.lens.to(torch.int64)
        return self.offsets[1:] - self.offsets[:-1]

    def to_strings(self) -> List[str]:
        seqs = self.seqs.cpu().numpy().tobytes()
        offs = self.offsets.cpu().numpy()
        if self.lens is not None:
            ls = self.lens.cpu().numpy()
            return [seqs[int(o):int(o) + int(l)].decode("latin-1") for o, l in zip(offs, ls)]
        return [seqs[int(offs[i]):int(offs[i + 1])].decode("latin-1") for i in range(self.n_reads)]


class BatchResult:
    """Per-read results of a batch call: ``out6`` int32[n,6] =
    (ref_start, ref_stop, query_start, query_stop, score, errors) -- the tuple
    Aligner.locate returns (reference _align.pyx:587) -- ``status`` uint8[n]
    (0 None, 1 match, 2 invalid input) and optionally ``best_adapter`` int32[n]."""

    def __init__(self, out6, status, best_adapter=None):
        self.out6 = out6
        self.status = status
        self.best_adapter = best_adapter

    def cpu(self):
        out6 = self.out6.cpu().numpy()
        status = self.status.cpu().numpy()
        best = None if self.best_adapter is None else self.best_adapter.cpu().numpy()
        return out6, status, best

    def tuples(self) -> List[Optional[Tuple[int, int, int, int, int, int]]]:
        out6, status, _ = self.cpu()
        if (status == _lib.INVALID).any():
            raise ValueError("String must contain only ASCII characters")
        return [tuple(int(v) for v in out6[i]) if status[i] == _lib.MATCH else None
                for i in range(len(status))]


def prob_u32(p: float) -> int:
    return min(int(round(p * 4294967296.0)), 4294967295)


def prob_u16(p: float) -> int:
    return min(int(round(p * 65536.0)), 65535)


# ---- thin launch helpers (device pointers) ------------------------------------------------------
def locate_batch(plan: "_lib.Plan", adapter: int, batch: ReadBatch) -> BatchResult:
    torch = _torch()
    n = batch.n_reads
    out6 = torch.empty((n, 6), dtype=torch.int32, device=batch.device)
    status = torch.empty(n, dtype=torch.uint8, device=batch.device)
    if n:
        ws = batch.workspace(plan)
        with torch.cuda.device(batch.device):
            _lib.check(_lib.lib().cah_locate_batch(
                plan.handle, adapter, batch.seqs.data_ptr(), batch.offsets.data_ptr(),
                batch._lens_ptr(), n, out6.data_ptr(), status.data_ptr(), ws.data_ptr(),
                ws.numel(), _stream_ptr()))
    return BatchResult(out6, status)


def kmers_present_batch(plan: "_lib.Plan", adapter: int, batch: ReadBatch):
    torch = _torch()
    n = batch.n_reads
    present = torch.empty(n, dtype=torch.uint8, device=batch.device)
    if n:
        with torch.cuda.device(batch.device):
            _lib.check(_lib.lib().cah_kmers_present_batch(
                plan.handle, adapter, batch.seqs.data_ptr(), batch.offsets.data_ptr(),
                batch._lens_ptr(), n, present.data_ptr(), _stream_ptr()))
    return present


# ---- ragged batches of plans with several adapters -----------------------------------------------------------------
FRAME_MIN_READS = 65536       # smaller batches: the per-lane kernels do (and a batch built on the device would need a look-up of its longest read)


def _frame_len(plan: "_lib.Plan", batch: ReadBatch) -> int:
    """the frame length a ragged batch is streamed in (0: the batch takes the plain entry points).  *Adapter.match_to /
    MultipleAdapters.match_to on reads of any length (reference adapters.py:815-832, :1265-1286): the streaming kernels take
    every read end-aligned in a frame of ``frame_len`` characters (cah_match_batch_frames) -- the longest read of the batch,
    known where the batch was built (``ReadBatch.max_len``) or looked up once on the device."""
    L = _lib.lib()
    if os.environ.get("CAH_NO_FRAMES") or not hasattr(L, "cah_match_batch_frames"):
        return 0
    if batch.n_reads < FRAME_MIN_READS or (batch.uniform_len and batch.lens is None):
        return 0
    if getattr(batch, "within_uniform", None) or getattr(batch, "suffix_of_uniform", None):
        return 0                                              # (views of a uniform batch have entry points of their own)
    n = getattr(batch, "max_len", None)
    if n is None:
        n = int(batch.lengths().max().item()) if batch.n_reads else 0       # (one synchronisation, once per batch)
        batch.max_len = n
    n = max(int(n), 16)
    if plan.n_adapters >= 2:
        return n if plan.multi_kind(n) == "stream" else 0
    # one adapter: its streaming prefilter takes frames of up to 160 characters (the library falls back by itself where the
    # plan has no streaming form)
    return n if n <= 160 else 0


def match_batch(plan: "_lib.Plan", batch: ReadBatch, out: Optional[BatchResult] = None) -> BatchResult:
    """Fused prefilter -> compaction -> DP -> best-adapter for every matcher of ``plan``."""
    torch = _torch()
    n = batch.n_reads
    if out is None:
        out = BatchResult(torch.empty((n, 6), dtype=torch.int32, device=batch.device),
                          torch.empty(n, dtype=torch.uint8, device=batch.device),
                          torch.empty(n, dtype=torch.int32, device=batch.device))
    if n:
        ws = batch.workspace(plan)
        best_ptr = out.best_adapter.data_ptr() if out.best_adapter is not None else None
        with torch.cuda.device(batch.device):
            if batch.uniform_len and batch.lens is None and not os.environ.get("CAH_NO_UNIFORM"):
                # equally long reads (what a sequencer emits): the entry point without an offsets array
                _lib.check(_lib.lib().cah_match_batch_uniform(
                    plan.handle, batch.seqs.data_ptr(), batch.uniform_len, n, out.out6.data_ptr(), best_ptr,
                    out.status.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr()))
            elif getattr(batch, "suffix_of_uniform", None) and batch.lens is not None and not os.environ.get("CAH_NO_UNIFORM"):
                # views that end where the reads of a uniform batch end (second stage of a linked adapter)
                _lib.check(_lib.lib().cah_match_batch_suffix_views(
                    plan.handle, batch.seqs.data_ptr(), batch.offsets.data_ptr(), batch.lens.data_ptr(),
                    int(batch.suffix_of_uniform), n, out.out6.data_ptr(), best_ptr, out.status.data_ptr(), ws.data_ptr(),
                    ws.numel(), _stream_ptr()))
            elif (getattr(batch, "within_uniform", None) and batch.lens is not None and not os.environ.get("CAH_NO_UNIFORM")
                  and hasattr(_lib.lib(), "cah_match_batch_views")):
                # views anywhere inside the reads of a uniform batch (reads cut by a modifier in front of the adapter search)
                _lib.check(_lib.lib().cah_match_batch_views(
                    plan.handle, batch.seqs.data_ptr(), batch.offsets.data_ptr(), batch.lens.data_ptr(),
                    int(batch.within_uniform), n, out.out6.data_ptr(), best_ptr, out.status.data_ptr(), ws.data_ptr(),
                    ws.numel(), _stream_ptr()))
            elif _frame_len(plan, batch):
                # a ragged batch of a plan with several adapters: every read end-aligned in a frame of the longest one's length
                if batch.lens is None:
                    batch._frame_lens = getattr(batch, "_frame_lens", None)
                    if batch._frame_lens is None:
                        batch._frame_lens = (batch.offsets[1:] - batch.offsets[:-1]).to(_torch().int32)
                lens = batch.lens if batch.lens is not None else batch._frame_lens
                _lib.check(_lib.lib().cah_match_batch_frames(
                    plan.handle, batch.seqs.data_ptr(), batch.offsets.data_ptr(), lens.data_ptr(), _frame_len(plan, batch), n,
                    out.out6.data_ptr(), best_ptr, out.status.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr()))
            else:
                _lib.check(_lib.lib().cah_match_batch(
                    plan.handle, batch.seqs.data_ptr(), batch.offsets.data_ptr(), batch._lens_ptr(), n,
                    out.out6.data_ptr(), best_ptr, out.status.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr()))
    return out


def linked_match_batch(front_plan: "_lib.Plan", back_plan: "_lib.Plan", batch: ReadBatch,
                       out_front: Optional[BatchResult] = None, out_back: Optional[BatchResult] = None):
    """LinkedAdapter.match_to over a batch, entirely on the device (reference adapters.py:1215-1227):
    the 5' adapter on the reads, then the 3' adapter on the part after the 5' match -- a view
    (start = rstop of the front match, or 0) into the same HBM buffer.  Returns (front, back, view);
    the required/optional verdict is the caller's (it needs only the two status arrays)."""
    torch = _torch()
    n = batch.n_reads
    starts = torch.empty(n, dtype=torch.int64, device=batch.device)
    vlens = torch.empty(n, dtype=torch.int32, device=batch.device)
    uniform = batch.uniform_len if (batch.uniform_len and batch.lens is None and not os.environ.get("CAH_NO_UNIFORM")) else 0

    def result(out):
        if out is not None:
            return out
        return BatchResult(torch.empty((n, 6), dtype=torch.int32, device=batch.device),
                           torch.empty(n, dtype=torch.uint8, device=batch.device),
                           torch.empty(n, dtype=torch.int32, device=batch.device))

    if uniform and n:
        # equally long reads: the library runs both stages (and folds the 5' comparison into the 3' prefilter's pass
        # over the batch when the 5' adapter is anchored and tolerates no error)
        front, back = result(out_front), result(out_back)
        batch.workspace(front_plan)
        ws = batch.workspace(back_plan)                      # (grow-only: large enough for either plan now)
        with torch.cuda.device(batch.device):
            _lib.check(_lib.lib().cah_linked_match_batch_uniform(
                front_plan.handle, back_plan.handle, batch.seqs.data_ptr(), int(uniform), n,
                front.out6.data_ptr(), front.best_adapter.data_ptr() if front.best_adapter is not None else None,
                front.status.data_ptr(), back.out6.data_ptr(),
                back.best_adapter.data_ptr() if back.best_adapter is not None else None, back.status.data_ptr(),
                starts.data_ptr(), vlens.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr()))
        view = ReadBatch(batch.seqs, starts, vlens, n_reads=n, validated=batch.validated)
        view._workspace = batch._workspace
        view.suffix_of_uniform = uniform
        return front, back, view
    front = match_batch(front_plan, batch, out_front)
    batch.workspace()
    # the views in one pass over the front stage's results (cah_linked_views)
    if n:
        with torch.cuda.device(batch.device):
            _lib.check(_lib.lib().cah_linked_views(
                front.out6.data_ptr(), front.status.data_ptr(), batch.offsets.data_ptr(), batch._lens_ptr(), 0, n,
                starts.data_ptr(), vlens.data_ptr(), _stream_ptr()))
    view = ReadBatch(batch.seqs, starts, vlens, n_reads=n, validated=batch.validated)
    view._workspace = batch._workspace                       # same reads, same stream order: the scratch can be shared
    back = match_batch(back_plan, view, out_back)
    return front, back, view
