"""Aligner / PrefixComparer / SuffixComparer with the reference's Python API, backed by the
HIP kernels (no CPU fallback).

Mirrors reference src/cutadapt/_align.pyx (Aligner :93-591, PrefixComparer :594-693,
SuffixComparer :696-714; type stubs _align.pyi:10-27) and src/cutadapt/align.py:24-34
(EndSkip).  ``locate(str)`` keeps the per-read signature by sending a batch of one through
the same kernel; ``locate_batch(ReadBatch)`` is the form the pipeline should use.
"""
from enum import IntFlag
from typing import Optional, Tuple

from . import _lib

__all__ = ["EndSkip", "Aligner", "PrefixComparer", "SuffixComparer"]

AlignmentTuple = Tuple[int, int, int, int, int, int]


class EndSkip(IntFlag):
    """Which ends may be skipped at no cost (reference align.py:24-34)."""

    REFERENCE_START = 1
    QUERY_START = 2
    REFERENCE_END = 4
    QUERY_STOP = 8
    SEMIGLOBAL = 15


def _locate_one(plan: _lib.Plan, query: str) -> Optional[AlignmentTuple]:
    """one read through the library's one-read entry point (no arrays are built for it)"""
    return _lib.one_read(_lib.lib().cah_locate_one_host, plan.handle, query)


class Aligner:
    """Find a full or partial occurrence of ``reference`` (the adapter) in a query (the read)
    allowing mismatches and indels; see reference _align.pyx:94-169 for the semantics.

    Result of ``locate``: (ref_start, ref_stop, query_start, query_stop, score, errors) or None.
    """

    def __init__(self, reference: str, max_error_rate: float, flags: int = 15,
                 wildcard_ref: bool = False, wildcard_query: bool = False,
                 indel_cost: int = 1, min_overlap: int = 1):
        self.reference = reference
        self.max_error_rate = float(max_error_rate)
        self._flags = int(flags) & 15
        self.wildcard_ref = bool(wildcard_ref)
        self.wildcard_query = bool(wildcard_query)
        if not reference.isascii():
            # the reference translates the adapter through a table when wildcards are on (ValueError) and encodes it
            # otherwise (UnicodeEncodeError): _align.pyx:40-46, :265-277
            if self.wildcard_ref or self.wildcard_query:
                raise ValueError("String must contain only ASCII characters")
            reference.encode("ascii")
        self._indel_cost = int(indel_cost)
        self._min_overlap = int(min_overlap)
        self._plan = _lib.Plan([self.spec()])
        self.effective_length = self._plan.effective_length(0)
        self._debug = False
        self._dpmatrix = None
        self._scorematrix = None

    def spec(self, kmer_sets=None, kmer_ref_wildcards=False, kmer_query_wildcards=False) -> _lib.MatcherSpec:
        return _lib.MatcherSpec(self.reference, self.max_error_rate, self._flags, self.wildcard_ref,
                                self.wildcard_query, self._indel_cost, self._min_overlap,
                                _lib.KIND_ALIGNER, kmer_sets, kmer_ref_wildcards, kmer_query_wildcards)

    def __reduce__(self):
        # pipelines are pickled into worker processes (reference _align.pyx:239-240)
        return (Aligner, (self.reference, self.max_error_rate, self._flags, self.wildcard_ref,
                          self.wildcard_query, self._indel_cost, self._min_overlap))

    def __repr__(self):
        return (f"Aligner(reference='{self.reference}', max_error_rate={self.max_error_rate}, "
                f"flags={self._flags}, wildcard_ref={self.wildcard_ref}, "
                f"wildcard_query={self.wildcard_query}, indel_cost={self._indel_cost}, "
                f"min_overlap={self._min_overlap})")

    def locate(self, query: str) -> Optional[AlignmentTuple]:
        if self._debug:
            # one read through the statement-by-statement kernel, which writes the matrices out (same tuple)
            result, cost, score = _lib.locate_debug(self.spec(), query)
            self._dpmatrix = DPMatrix(self.reference, query, cost)
            self._scorematrix = DPMatrix(self.reference, query, score)
            return result
        return _locate_one(self._plan, query)

    def locate_batch(self, batch):
        from . import batch as _b
        return _b.locate_batch(self._plan, 0, batch)

    def enable_debug(self):
        """Store the dynamic programming matrices while running the locate() method and make them available in
        the .dpmatrix and .scorematrix attributes (reference _align.pyx:291-296).  The register-resident kernels
        never hold a matrix, so a debugging aligner sends its reads one at a time through the kernel that keeps
        the column in memory (csrc/long.hip), which writes every computed cell out."""
        self._debug = True

    @property
    def dpmatrix(self):
        """The dynamic programming matrix as a DPMatrix object; None unless debugging has been enabled with
        enable_debug() (reference _align.pyx:279-285)."""
        return self._dpmatrix

    @property
    def scorematrix(self):
        return self._scorematrix


class DPMatrix:
    """Representation of the dynamic-programming matrix (reference _align.pyx:58-92): entries may be None, in
    which case that value was not computed."""

    def __init__(self, reference, query, rows=None):
        m, n = len(reference), len(query)
        self._rows = rows if rows is not None else [[None] * (n + 1) for _ in range(m + 1)]
        self.reference = reference
        self.query = query

    def set_entry(self, i: int, j: int, cost):
        self._rows[i][j] = cost

    def __str__(self):
        rows = ["     " + " ".join(c.rjust(2) for c in self.query)]
        for c, row in zip(" " + self.reference, self._rows):
            rows.append(c + " " + " ".join("  " if v is None else "{:2d}".format(v) for v in row))
        return "\n".join(rows)


_IUPAC_BITS = {"X": 0, "A": 1, "C": 2, "G": 4, "T": 8, "U": 8}
for _code, _members in (("R", "AG"), ("Y", "CT"), ("S", "GC"), ("W", "AT"), ("K", "GT"), ("M", "AC"), ("B", "CGT"), ("D", "AGT"),
                        ("H", "ACT"), ("V", "ACG")):
    _IUPAC_BITS[_code] = sum(_IUPAC_BITS[x] for x in _members)
_IUPAC_BITS["N"] = 15 | 0x80


class PrefixComparer:
    """Hamming-distance comparison of an anchored 5' adapter with the start of the read
    (reference _align.pyx:594-693)."""

    _kind = _lib.KIND_PREFIX

    def __init__(self, reference: str, max_error_rate: float, wildcard_ref: bool = False,
                 wildcard_query: bool = False, min_overlap: int = 1):
        self._reference = reference
        self.max_error_rate = float(max_error_rate)
        self.wildcard_ref = bool(wildcard_ref)
        self.wildcard_query = bool(wildcard_query)
        self.min_overlap = int(min_overlap)
        self._plan = _lib.Plan([self.spec()])
        self.effective_length = self._plan.effective_length(0)
        self.max_k = int(self.max_error_rate * self.effective_length)

    def spec(self, kmer_sets=None, kmer_ref_wildcards=False, kmer_query_wildcards=False) -> _lib.MatcherSpec:
        return _lib.MatcherSpec(self._reference, self.max_error_rate, 0, self.wildcard_ref,
                                self.wildcard_query, 1, self.min_overlap, self._kind, kmer_sets,
                                kmer_ref_wildcards, kmer_query_wildcards)

    def __reduce__(self):
        return (type(self), (self._reference, self.max_error_rate, self.wildcard_ref,
                             self.wildcard_query, self.min_overlap))

    def _shown_reference(self) -> bytes:
        """what the reference's repr shows for `reference`: the adapter as ITS comparer stores it (_align.pyx:637-642, :707) --
        one byte per character: IUPAC bit sets when the adapter has wildcards (A 1, C 2, G 4, T 8, their unions, N with the
        top bit as well, anything else 0), A/C/G/T bits with every other character 0x80 when only the read has, upper-case
        text otherwise; an anchored 3' adapter back to front"""
        text = self._reference[::-1] if self._kind == _lib.KIND_SUFFIX else self._reference
        if self.wildcard_ref:
            return bytes(_IUPAC_BITS.get(c, 0) for c in text.upper())
        if self.wildcard_query:
            return bytes(_IUPAC_BITS[c] if c in "ACGTU" else 0x80 for c in text.upper())
        return text.encode("ascii").upper()

    def __repr__(self):
        return "{}(reference={!r}, max_k={}, wildcard_ref={}, wildcard_query={})".format(
            self.__class__.__name__, self._shown_reference(), self.max_k, self.wildcard_ref,
            self.wildcard_query)

    def locate(self, query: str) -> Optional[AlignmentTuple]:
        return _locate_one(self._plan, query)

    def locate_batch(self, batch):
        from . import batch as _b
        return _b.locate_batch(self._plan, 0, batch)


class SuffixComparer(PrefixComparer):
    """Anchored 3' variant (reference _align.pyx:696-714)."""

    _kind = _lib.KIND_SUFFIX
