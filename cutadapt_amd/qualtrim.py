"""Quality trimming, NextSeq trimming, poly-A trimming and expected errors for whole batches
(SURVEY.md section 8(f), row 4).

Mirrors reference src/cutadapt/qualtrim.pyx (``quality_trim_index`` :22-70, ``nextseq_trim_index``
:73-113, ``poly_a_trim_index`` :116-165, ``expected_errors`` :168-190) -- same names, arguments and
errors for single strings (a batch of one) -- and adds ``*_batch`` forms over a ReadBatch whose
qualities are packed with the same offsets as the sequences (HIP kernels in csrc/qualtrim.hip).
"""
from typing import Optional, Tuple

import numpy as np

from . import _lib


class HasNoQualities(Exception):
    pass


def _torch():
    import torch
    return torch


def _stream(dev) -> int:
    return _torch().cuda.current_stream(dev).cuda_stream


def _device_bytes(data: bytes, device=None):
    torch = _torch()
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    t = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(device) if data else \
        torch.zeros(1, dtype=torch.uint8, device=device)
    offsets = torch.tensor([0, len(data)], dtype=torch.int64, device=device)
    return t, offsets


# ---- batch forms: everything stays in HBM, results come back as numpy ---------------------------
def quality_trim_batch(quals, offsets, lens, n_reads: int, cutoff_front: int, cutoff_back: int, base: int = 33) -> np.ndarray:
    """-> int32[n,2] (start, stop) of the good-quality segment of every read"""
    torch = _torch()
    out = torch.zeros((n_reads, 2), dtype=torch.int32, device=quals.device)
    if n_reads:
        with torch.cuda.device(quals.device):
            _lib.check(_lib.lib().cah_quality_trim_batch(
                quals.data_ptr(), offsets.data_ptr(), lens.data_ptr() if lens is not None else None, n_reads,
                int(cutoff_front), int(cutoff_back), int(base), out.data_ptr(), _stream(quals.device)))
    return out.cpu().numpy()


def nextseq_trim_batch(seqs, quals, offsets, lens, n_reads: int, cutoff: int, base: int = 33,
                       qual_offsets=None) -> np.ndarray:
    """``qual_offsets``: where the qualities of every read start in ``quals`` when they are not packed like the
    sequences (a raw FASTQ chunk indexed on the device)"""
    torch = _torch()
    out = torch.zeros(n_reads, dtype=torch.int32, device=quals.device)
    if n_reads:
        with torch.cuda.device(quals.device):
            if qual_offsets is None:
                _lib.check(_lib.lib().cah_nextseq_trim_batch(
                    seqs.data_ptr(), quals.data_ptr(), offsets.data_ptr(), lens.data_ptr() if lens is not None else None,
                    n_reads, int(cutoff), int(base), out.data_ptr(), _stream(quals.device)))
            else:
                _lib.check(_lib.lib().cah_nextseq_trim_batch_q(
                    seqs.data_ptr(), quals.data_ptr(), offsets.data_ptr(), qual_offsets.data_ptr(), lens.data_ptr(),
                    n_reads, int(cutoff), int(base), out.data_ptr(), _stream(quals.device)))
    return out.cpu().numpy()


def poly_a_trim_batch(seqs, offsets, lens, n_reads: int, revcomp: bool = False) -> np.ndarray:
    torch = _torch()
    out = torch.zeros(n_reads, dtype=torch.int32, device=seqs.device)
    if n_reads:
        with torch.cuda.device(seqs.device):
            _lib.check(_lib.lib().cah_poly_a_trim_batch(
                seqs.data_ptr(), offsets.data_ptr(), lens.data_ptr() if lens is not None else None, n_reads,
                int(bool(revcomp)), out.data_ptr(), _stream(seqs.device)))
    return out.cpu().numpy()


def expected_errors_batch(quals, offsets, lens, n_reads: int, base: int = 33) -> Tuple[np.ndarray, np.ndarray]:
    """-> (float64[n] expected errors, bool[n] valid); invalid reads hold -1.0"""
    torch = _torch()
    out = torch.zeros(n_reads, dtype=torch.float64, device=quals.device)
    status = torch.zeros(n_reads, dtype=torch.uint8, device=quals.device)
    if n_reads:
        with torch.cuda.device(quals.device):
            _lib.check(_lib.lib().cah_expected_errors_batch(
                quals.data_ptr(), offsets.data_ptr(), lens.data_ptr() if lens is not None else None, n_reads,
                int(base), out.data_ptr(), status.data_ptr(), _stream(quals.device)))
    return out.cpu().numpy(), status.cpu().numpy() != _lib.INVALID


# ---- the reference's single-string functions (a batch of one) -----------------------------------
def _latin1(s: str, what: str) -> bytes:
    try:
        return s.encode("latin-1")                # PyUnicode_1BYTE_KIND (qualtrim.pyx:47-48)
    except UnicodeEncodeError:
        raise ValueError(what)


def quality_trim_index(qualities: Optional[str], cutoff_front: int, cutoff_back: int, base: int = 33) -> Tuple[int, int]:
    if qualities is None:
        raise HasNoQualities("Cannot do quality trimming when no qualities are available")
    q, off = _device_bytes(_latin1(qualities, "Quality data is not ASCII"))
    r = quality_trim_batch(q, off, None, 1, cutoff_front, cutoff_back, base)
    return int(r[0, 0]), int(r[0, 1])


def nextseq_trim_index(sequence, cutoff: int, base: int = 33) -> int:
    """``sequence`` is a record with ``.sequence`` and ``.qualities`` (reference :84-85)"""
    bases, qualities = sequence.sequence, sequence.qualities
    if qualities is None:
        raise HasNoQualities()
    qb = _latin1(qualities, "Quality data is not ASCII")
    sb = bases.encode("latin-1", errors="replace")
    if len(sb) < len(qb):
        raise IndexError("string index out of range")
    q, off = _device_bytes(qb)
    s, _ = _device_bytes(sb[:len(qb)] if len(qb) else b"")
    return int(nextseq_trim_batch(s, q, off, None, 1, cutoff, base)[0])


def poly_a_trim_index(s: str, revcomp: bool = False) -> int:
    b, off = _device_bytes(_latin1(s, "Sequence is not ASCII"))
    return int(poly_a_trim_batch(b, off, None, 1, revcomp)[0])


def expected_errors(qualities: str, base: int = 33) -> float:
    try:
        data = qualities.encode("ascii")
    except UnicodeEncodeError:
        raise ValueError(f"Quality string contains non-ASCII values: {qualities}")
    q, off = _device_bytes(data)
    e, ok = expected_errors_batch(q, off, None, 1, base)
    if not ok[0]:
        for c in qualities:
            if ord(c) < base or ord(c) > 126:
                raise ValueError(f"Not a valid phred value {ord(c)} for character {c}")
    return float(e[0])
