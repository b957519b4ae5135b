"""KmerFinder with the reference's Python API, backed by the HIP shift-and kernel.

Mirrors reference src/cutadapt/_kmer_finder.pyx:66-213 (stub _kmer_finder.pyi:5-12).
"""
from typing import List, Optional, Tuple

import numpy as np

from . import _lib

MAXIMUM_WORD_SIZE = 64   # reference _kmer_finder.pyx:51-52


class KmerFinder:
    """Find k-mers in strings; case-independent, optional IUPAC matching.

    ``positions_and_kmers`` is a list of ``(start, stop, [kmers])``: at least one of the k-mers
    has to occur inside ``sequence[start:stop]`` (Python slice semantics for negative start;
    ``stop`` None = end).  ``kmers_present`` answers whether any search set hits.
    """

    def __init__(self, positions_and_kmers: List[Tuple[int, Optional[int], List[str]]],
                 ref_wildcards: bool = False, query_wildcards: bool = False):
        self.positions_and_kmers = positions_and_kmers
        self.ref_wildcards = bool(ref_wildcards)
        self.query_wildcards = bool(query_wildcards)
        # An EMPTY k-mer -- kmer_heuristic emits one when an adapter may have as many errors as it has characters -- is
        # accepted by the reference and never found (_kmer_finder.pyx:121-160: its start bit sits on the next k-mer's, its
        # "found" bit on the previous one's): the search sets the library sees are the given ones without them.
        self.searched_positions_and_kmers = [
            (start, stop, [k for k in kmers if not (isinstance(k, str) and k == "")]) for start, stop, kmers in positions_and_kmers]
        spec = _lib.MatcherSpec(kind=_lib.KIND_KMER_ONLY, kmer_sets=list(self.searched_positions_and_kmers),
                                kmer_ref_wildcards=ref_wildcards,
                                kmer_query_wildcards=query_wildcards)
        self._plan = _lib.Plan([spec])
        self.number_of_searches = self._plan.n_kmer_entries(0)

    def __reduce__(self):
        return KmerFinder, (self.positions_and_kmers, self.ref_wildcards, self.query_wildcards)

    def kmers_present(self, sequence: str) -> bool:
        if not isinstance(sequence, str):
            raise TypeError("sequence must be a str")
        try:
            q = sequence.encode("ascii")
        except UnicodeEncodeError:
            raise ValueError("Only ASCII strings are supported")
        seqs = np.frombuffer(q, dtype=np.uint8)
        offsets = np.array([0, len(q)], dtype=np.int64)
        present = np.zeros(1, dtype=np.uint8)
        _lib.check(_lib.lib().cah_kmers_present_batch_host(
            self._plan.handle, 0, seqs.ctypes.data if len(q) else None, offsets.ctypes.data, 1,
            present.ctypes.data))
        if present[0] == _lib.INVALID:
            raise ValueError("Only ASCII strings are supported")
        return bool(present[0])

    def kmers_present_batch(self, batch):
        from . import batch as _b
        return _b.kmers_present_batch(self._plan, 0, batch)
