"""Adapter classes with the reference's ``match_to()`` API plus batch forms, on the HIP path.

Mirrors the matching part of reference src/cutadapt/adapters.py:
  Where flag table :39-53, Match types :292-493, SingleAdapter :533-681 and its subclasses
  :684-1089 (Front/Back/Anywhere/NonInternal*/Prefix/Suffix/Rightmost*), LinkedAdapter
  :1181-1243 with LinkedMatch :1092-1178, MultipleAdapters :1246-1286.
Out of scope here: statistics classes (:71-289) and AdapterIndex
(:1289-1571).

``match_to(sequence: str)`` keeps the reference's per-read contract (returns a Match or
None) by running a batch of one through the fused kernel path; ``match_to_batch(ReadBatch)``
is the form that feeds the GPU properly and returns array results, Match objects are
created on demand only.
"""
from abc import ABC, abstractmethod
from enum import IntFlag
from typing import List, Optional, Sequence, Tuple, Union

import collections
import numpy as np

from . import _lib
from ._kmer_finder import KmerFinder
from .align import Aligner, EndSkip, PrefixComparer, SuffixComparer
from .kmer_heuristic import create_positions_and_kmers


class MockKmerFinder:
    """Stand-in used when the prefilter cannot be applied (reference adapters.py:29-31)."""

    def kmers_present(self, sequence: str):
        return True


class InvalidCharacter(Exception):
    pass


class Where(IntFlag):
    """Aligner flag combinations per adapter type (reference adapters.py:39-53)."""

    BACK = EndSkip.QUERY_START | EndSkip.QUERY_STOP | EndSkip.REFERENCE_END
    FRONT = EndSkip.QUERY_START | EndSkip.QUERY_STOP | EndSkip.REFERENCE_START
    PREFIX = EndSkip.QUERY_STOP
    SUFFIX = EndSkip.QUERY_START
    FRONT_NOT_INTERNAL = EndSkip.REFERENCE_START | EndSkip.QUERY_STOP
    BACK_NOT_INTERNAL = EndSkip.QUERY_START | EndSkip.REFERENCE_END
    ANYWHERE = EndSkip.SEMIGLOBAL


# -------------------------------------------------------------------------------------------------
# Match objects (reference adapters.py:292-493)
# -------------------------------------------------------------------------------------------------
class Match(ABC):
    """What a modifier asks of any match (reference adapters.py:292-313): the part of the read that stays, the part that
    stays when the adapter is retained, the trimmed read, the matched characters."""
    adapter: "Adapter"

    remainder_interval = abstractmethod(lambda self: None)            # -> (start, stop) of what is left of the read
    retained_adapter_interval = abstractmethod(lambda self: None)     # ... when the adapter itself is kept (--action=retain)
    trimmed = abstractmethod(lambda self, read: None)
    match_sequence = abstractmethod(lambda self: None)


class SingleMatch(Match, ABC):
    """One adapter matched to one string (reference adapters.py:316-424): the tuple Aligner.locate returns -- adapter
    interval [astart, astop), read interval [rstart, rstop), score, errors -- with the adapter and the string.  Which side
    of the match goes is the subclass's one bit (_BEFORE); every interval below follows from it."""

    __slots__ = ["astart", "astop", "rstart", "rstop", "score", "errors", "adapter", "sequence", "length"]
    _BEFORE = property(abstractmethod(lambda self: None))     # (a plain class attribute in the two concrete classes)

    def __init__(self, astart: int, astop: int, rstart: int, rstop: int, score: int, errors: int,
                 adapter: "SingleAdapter", sequence: str):
        self.astart, self.astop, self.rstart, self.rstop, self.score, self.errors = astart, astop, rstart, rstop, score, errors
        self.adapter, self.sequence = adapter, sequence
        self.length = astop - astart   # aligned adapter characters

    def astuple(self) -> Tuple[int, int, int, int, int, int]:
        return (self.astart, self.astop, self.rstart, self.rstop, self.score, self.errors)

    def __repr__(self):
        names = self.__slots__[:6]
        return f"{type(self).__name__}({', '.join(f'{k}={v}' for k, v in zip(names, self.astuple()))})"

    def __eq__(self, other) -> bool:
        return (type(other) is type(self) and self.astuple() == other.astuple()
                and self.adapter is other.adapter and self.sequence == other.sequence)

    def wildcards(self, wildcard_char: str = "N") -> str:
        """Characters of the read aligned to wildcard characters of the adapter
        (reference adapters.py:378-393; unreliable with indels, as in the reference)."""
        ref, seq = self.adapter.sequence, self.sequence
        return "".join(seq[self.rstart + i] for i in range(self.length)
                       if ref[self.astart + i] == wildcard_char and self.rstart + i < len(seq))

    def get_info_records(self, read) -> List[List]:
        """one row of the info file (reference :395-417): the read cut at the match's two ends, qualities likewise"""
        cuts = (slice(0, self.rstart), slice(self.rstart, self.rstop), slice(self.rstop, None))
        quals = [read.qualities[c] for c in cuts] if read.qualities else ["", "", ""]
        return [["", self.errors, self.rstart, self.rstop] + [read.sequence[c] for c in cuts] + [self.adapter.name] + quals]

    def match_sequence(self):
        return self.sequence[self.rstart:self.rstop]

    # -- the side that goes (reference :427-493) ------------------------------------------------------
    def remainder_interval(self) -> Tuple[int, int]:
        return (self.rstop, len(self.sequence)) if self._BEFORE else (0, self.rstart)

    def retained_adapter_interval(self) -> Tuple[int, int]:
        return (self.rstart, len(self.sequence)) if self._BEFORE else (0, self.rstop)

    def trim_slice(self):
        return slice(self.rstop, None) if self._BEFORE else slice(None, self.rstart)

    def trimmed(self, read):
        return read[self.trim_slice()]

    def rest(self) -> str:
        """the part of the string on the removed side of the match"""
        return self.sequence[:self.rstart] if self._BEFORE else self.sequence[self.rstop:]

    def removed_sequence_length(self) -> int:
        return self.rstop if self._BEFORE else len(self.sequence) - self.rstart


class RemoveBeforeMatch(SingleMatch):
    """A match that removes the sequence before it (5' adapters; reference :427-457)."""
    _BEFORE = True


class RemoveAfterMatch(SingleMatch):
    """A match that removes the sequence after it (3' adapters; reference :460-493)."""
    _BEFORE = False

    def adjacent_base(self) -> str:
        return self.sequence[self.rstart - 1:self.rstart]


_name_counter = [1]


def _generate_adapter_name() -> str:
    name = str(_name_counter[0])
    _name_counter[0] += 1
    return name


# -------------------------------------------------------------------------------------------------
# batch results
# -------------------------------------------------------------------------------------------------
class BatchMatches:
    """Array-form result of ``match_to_batch``: per read the 6-tuple in *match coordinates*
    (astart, astop, rstart, rstop, score, errors), a found flag, which adapter won and whether
    the match removes the sequence before (True) or after (False) it.

    The result of a single adapter STAYS ON THE DEVICE: ``device_coords()`` / ``device_found()`` are the tensors a
    device-side pipeline goes on with (cah_trim_decide_device ...), and the numpy views ``coords`` / ``found`` /
    ``adapter_index`` / ``remove_before`` are made on first access (24 B per read over PCIe plus a host copy -- what
    used to bound ``match_to_batch`` at a fraction of the kernels' rate).  ``match(i)`` builds the reference-style
    Match object for one read on demand."""

    def __init__(self, coords=None, found=None, adapter_index=None, adapters: Sequence["SingleAdapter"] = (),
                 remove_before=None, reads=None, device_result=None, mirror_m: Optional[int] = None,
                 remove_before_fn=None):
        self._coords, self._found = coords, found
        self._adapter_index, self._remove_before = adapter_index, remove_before
        self.adapters = list(adapters)
        self._reads = reads          # ReadBatch or list[str]; needed only for Match objects
        self._strings = None
        self._res = device_result    # batch.BatchResult (device tensors) or None
        self._mirror_m = mirror_m    # Rightmost* adapters: the tuples refer to the reversed read (length of the adapter)
        self._remove_before_fn = remove_before_fn
        self._dev_coords = None

    # ---- device side -------------------------------------------------------------------------------
    def device_found(self):
        """bool tensor on the device (single-adapter results only)"""
        return self._res.status == _lib.MATCH

    def device_coords(self):
        """int32 [n, 6] tensor on the device in match coordinates (single-adapter results only)"""
        if self._dev_coords is None:
            c = self._res.out6
            if self._mirror_m is not None:
                import torch
                lens = self._reads.lengths().to(torch.int32)
                m = self._mirror_m
                mirrored = torch.stack([m - c[:, 1], m - c[:, 0], lens - c[:, 3], lens - c[:, 2], c[:, 4], c[:, 5]], dim=1)
                c = torch.where(self.device_found()[:, None], mirrored, torch.zeros_like(c))
            self._dev_coords = c
        return self._dev_coords

    # ---- host side, on demand ------------------------------------------------------------------------
    @property
    def coords(self) -> np.ndarray:
        if self._coords is None:
            self._coords = self.device_coords().cpu().numpy().astype(np.int64)
        return self._coords

    @coords.setter
    def coords(self, value) -> None:
        self._coords = value

    @property
    def found(self) -> np.ndarray:
        if self._found is None:
            self._found = (self._res.status == _lib.MATCH).cpu().numpy()
        return self._found

    @found.setter
    def found(self, value) -> None:
        self._found = value

    @property
    def adapter_index(self) -> np.ndarray:
        if self._adapter_index is None:
            self._adapter_index = np.zeros(len(self), dtype=np.int32)
        return self._adapter_index

    @property
    def remove_before(self) -> np.ndarray:
        if self._remove_before is None:
            self._remove_before = self._remove_before_fn(self.coords, self.found)
        return self._remove_before

    def __len__(self):
        return len(self._found) if self._found is not None else int(self._res.status.shape[0])

    def _sequence(self, i: int) -> str:
        if self._strings is None:
            self._strings = self._reads if isinstance(self._reads, list) else self._reads.to_strings()
        return self._strings[i]

    def match(self, i: int) -> Optional[SingleMatch]:
        if not self.found[i]:
            return None
        adapter = self.adapters[int(self.adapter_index[i])]
        cls = RemoveBeforeMatch if self.remove_before[i] else RemoveAfterMatch
        return cls(*(int(v) for v in self.coords[i]), adapter=adapter, sequence=self._sequence(i))

    def matches(self) -> List[Optional[SingleMatch]]:
        return [self.match(i) for i in range(len(self))]

    def tuples(self):
        return [tuple(int(v) for v in self.coords[i]) if self.found[i] else None for i in range(len(self))]


def _raise_if_invalid(status: np.ndarray, batch=None):
    """status 2 = the kernels refused a read: a byte >= 0x80 (the reference raises the same ValueError,
    _align.pyx:44-45) or -- a limit of this build, said as such -- a read longer than CAH_MAX_READ_LEN"""
    if (status == _lib.INVALID).any():
        _lib.raise_invalid_reads(int(batch.lengths().max().item()) if batch is not None and batch.n_reads else 0)


# -------------------------------------------------------------------------------------------------
# adapters
# -------------------------------------------------------------------------------------------------
class Matchable(ABC):
    def __init__(self, name: Optional[str], *args, **kwargs):
        self.name = name

    @abstractmethod
    def match_to(self, sequence: str):
        pass


class Adapter(Matchable, ABC):
    description = "adapter with one component"

    @abstractmethod
    def spec(self) -> str:
        pass

    @abstractmethod
    def descriptive_identifier(self) -> str:
        pass


_ACGT = frozenset("ACGT")
_IUPAC = frozenset("ABCDGHKMNRSTUVWXY")


def _canonical_adapter(sequence: str) -> str:
    """upper case, RNA and inosine spelled as DNA / N (reference adapters.py:579-581); empty adapters are an error"""
    seq = sequence.upper().replace("U", "T").replace("I", "N")
    if seq == "":
        raise ValueError("Adapter sequence is empty")
    return seq


def _error_rate(max_errors: float, seq: str) -> float:
    """values from 1 on are absolute error counts, spread over the characters that are not N (reference :582-584)"""
    informative = len(seq) - seq.count("N")
    return max_errors / informative if (max_errors >= 1 and informative) else max_errors


def _require_iupac(seq: str, letters) -> None:
    """the reference's complaint about the first character that is no IUPAC code (:586-592)"""
    if letters <= _IUPAC:
        return
    bad = next(c for c in seq if c not in _IUPAC)
    raise InvalidCharacter(f"Character '{bad}' in adapter sequence '{seq}' is not a valid IUPAC code. "
                           f"Use only characters 'ABCDGHIKMNRSTUVWXY'.")


class SingleAdapter(Adapter, ABC):
    """One adapter: sequence, error rate, type (reference adapters.py:533-681).

    ``max_errors`` < 1 is a rate, otherwise an absolute number divided by the number of non-N
    characters; ``min_overlap`` is clamped to the adapter length; ``indels=False`` is expressed
    as indel_cost 100000 (reference :605)."""

    allows_partial_matches: bool = True
    _reverse_reads = False           # Rightmost* adapters match the reversed read
    _remove_before = False           # which Match type a hit produces

    def __init__(self, sequence: str, max_errors: float = 0.1, min_overlap: int = 3,
                 read_wildcards: bool = False, adapter_wildcards: bool = True,
                 name: Optional[str] = None, indels: bool = True):
        super().__init__(name if name is not None else _generate_adapter_name())
        self._debug = False
        self.sequence = _canonical_adapter(sequence)
        self.max_error_rate = _error_rate(max_errors, self.sequence)
        self.min_overlap = min(min_overlap, len(self.sequence))
        letters = set(self.sequence)
        if adapter_wildcards:
            _require_iupac(self.sequence, letters)
        # a plain ACGT adapter is matched without the wildcard tables (reference :592-595)
        self.adapter_wildcards = bool(adapter_wildcards) and not letters <= _ACGT
        self.read_wildcards = read_wildcards
        self.indels = indels
        self.aligner = self._aligner()
        self.kmer_finder = self._kmer_finder()                # (may look at the aligner: anchored adapters without indels)
        # fused plan: this adapter's aligner + its prefilter as ONE matcher, so that
        # match_to()/match_to_batch() are a single library call
        self._fused_plan = _lib.Plan([self.matcher_spec()])

    # -- pickling: adapters travel to worker processes (reference runners.py:345-356 pickles the whole pipeline) ------
    # The library handles (aligner, prefilter, fused plan) stay behind; the receiving process builds its own from the
    # plain attributes, in its own HIP context.
    _HANDLES = ("aligner", "kmer_finder", "_fused_plan")

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in self._HANDLES}

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.aligner = self._aligner()
        self.kmer_finder = self._kmer_finder()
        if self._debug:
            self.aligner.enable_debug()
        self._fused_plan = _lib.Plan([self.matcher_spec()])

    # -- construction helpers ---------------------------------------------------------------
    def _make_aligner(self, sequence: str, flags: int) -> Aligner:
        indel_cost = 1 if self.indels else 100000
        return Aligner(sequence, self.max_error_rate, flags=flags, wildcard_ref=self.adapter_wildcards,
                       wildcard_query=self.read_wildcards, indel_cost=indel_cost,
                       min_overlap=self.min_overlap)

    def _make_kmer_finder(self, sequence: str, back_adapter: bool, front_adapter: bool,
                          internal: bool = True) -> Union[KmerFinder, MockKmerFinder]:
        positions_and_kmers = create_positions_and_kmers(
            sequence, self.min_overlap, self.max_error_rate, back_adapter, front_adapter, internal)
        try:
            return KmerFinder(positions_and_kmers, self.adapter_wildcards, self.read_wildcards)
        except ValueError:
            return MockKmerFinder()     # k-mers too long (reference :633-639)

    def matcher_spec(self) -> _lib.MatcherSpec:
        """aligner + prefilter as one cah_adapter_desc"""
        if isinstance(self.kmer_finder, KmerFinder):
            return self.aligner.spec(self.kmer_finder.searched_positions_and_kmers,
                                     self.kmer_finder.ref_wildcards, self.kmer_finder.query_wildcards)
        return self.aligner.spec(None)

    def __repr__(self):
        fields = ", ".join(f"{k}={getattr(self, k)!r}" if k in ("name", "sequence") else f"{k}={getattr(self, k)}"
                           for k in ("name", "sequence", "max_error_rate", "min_overlap", "read_wildcards",
                                     "adapter_wildcards", "indels"))
        return f"<{type(self).__name__}({fields})>"

    @property
    def effective_length(self) -> int:
        return self.aligner.effective_length

    def enable_debug(self) -> None:
        self._debug = True
        self.aligner.enable_debug()

    @abstractmethod
    def _aligner(self):
        pass

    @abstractmethod
    def _kmer_finder(self):
        pass

    def __len__(self) -> int:
        return len(self.sequence)

    # -- matching ---------------------------------------------------------------------------------
    def _wrap(self, alignment, sequence: str):
        cls = RemoveBeforeMatch if self._remove_before else RemoveAfterMatch
        return cls(*alignment, adapter=self, sequence=sequence)

    def _mirror(self, t, seq_len: int):
        """map a hit on the reversed read / reversed adapter back (reference :777-785)"""
        ref_start, ref_end, query_start, query_end, score, errors = t
        m = len(self.sequence)
        return (m - ref_end, m - ref_start, seq_len - query_end, seq_len - query_start, score, errors)

    def _locate_fused(self, sequence: str):
        """kmers_present -> locate for one read in a single library call"""
        return _lib.one_read(_lib.lib().cah_match_one_host, self._fused_plan.handle, sequence)

    def _locate_debug(self, sequence: str):
        """enable_debug(): the reference's two steps, kmers_present then locate, with the aligner collecting its
        matrices, which are printed like the reference does (adapters.py:62-67, :772-773, :876-877)"""
        if not self.kmer_finder.kmers_present(sequence):
            return None
        alignment = self.aligner.locate(sequence)
        print("Edit distances:")
        print(self.aligner.dpmatrix)
        print("Scores:")
        print(self.aligner.scorematrix)
        return alignment

    def match_to(self, sequence: str):
        """Match this adapter to one read; a Match or None (reference e.g. :707-724, :815-832)."""
        if self._debug:
            work = sequence[::-1] if self._reverse_reads else sequence
            alignment = self._locate_debug(work)
            if alignment is None:
                return None
            return self._wrap(self._mirror(alignment, len(sequence)) if self._reverse_reads else alignment, sequence)
        if self._reverse_reads:
            alignment = self._locate_fused(sequence[::-1])
            if alignment is None:
                return None
            alignment = self._mirror(alignment, len(sequence))
        else:
            alignment = self._locate_fused(sequence)
            if alignment is None:
                return None
        return self._wrap(alignment, sequence)

    def match_to_batch(self, batch) -> BatchMatches:
        """Match this adapter to every read of a ReadBatch."""
        from . import batch as _b
        work = _reverse_batch(batch) if self._reverse_reads else batch
        return self._batch_matches(_b.match_batch(self._fused_plan, work), batch)

    def _batch_matches(self, res, batch) -> BatchMatches:
        """device result of this adapter's fused plan -> BatchMatches in match coordinates; nothing but one flag (was
        a read refused?) crosses PCIe here"""
        if batch.n_reads and bool((res.status == _lib.INVALID).any().item()):
            _lib.raise_invalid_reads(int(batch.lengths().max().item()))
        return BatchMatches(adapters=[self], reads=batch, device_result=res,
                            mirror_m=len(self.sequence) if self._reverse_reads else None,
                            remove_before_fn=self._remove_before_array)

    def _remove_before_array(self, coords, found) -> np.ndarray:
        return np.full(len(found), self._remove_before, dtype=bool)


def _reverse_batch(batch, complement: bool = False, select=None, data=None):
    """Per-read reversed copy of a ReadBatch (``cah_reverse_reads_batch``: one pass on the device; Rightmost*
    adapters search the reversed read with the reversed adapter, reference adapters.py:766, :870).
    ``complement``: the reverse complement (``cah_revcomp_reads_batch``; ReverseComplementer, reference
    modifiers.py:264-308).  ``select`` (bool/uint8 device tensor): only these reads are turned around, the others are
    copied.  ``data``: another byte tensor with the batch's layout to read instead of the sequences (qualities)."""
    import torch
    from .batch import ReadBatch, _stream_ptr
    lens = batch.lengths()
    n = batch.n_reads
    new_offsets = torch.zeros(n + 1, dtype=torch.int64, device=batch.device)
    torch.cumsum(lens, 0, out=new_offsets[1:])
    total = int(new_offsets[-1].item()) if n else 0
    out = torch.empty(total, dtype=torch.uint8, device=batch.device)
    src = batch.seqs if data is None else data
    if total:
        with torch.cuda.device(batch.device):
            if complement or select is not None or data is not None:
                sel = None if select is None else select.to(device=batch.device, dtype=torch.uint8).contiguous()
                _lib.check(_lib.lib().cah_revcomp_reads_batch(
                    src.data_ptr(), batch.offsets.data_ptr(), batch._lens_ptr(), n,
                    new_offsets.data_ptr(), out.data_ptr(), int(bool(complement)),
                    sel.data_ptr() if sel is not None else None, _stream_ptr()))
            else:
                _lib.check(_lib.lib().cah_reverse_reads_batch(
                    src.data_ptr(), batch.offsets.data_ptr(), batch._lens_ptr(), n,
                    new_offsets.data_ptr(), out.data_ptr(), _stream_ptr()))
    if data is not None:
        return out
    return ReadBatch(out, new_offsets, validated=batch.validated, uniform_len=batch.uniform_len)


# The nine adapter types as rows of ONE table: what the aligner is anchored to (Where), which sides of the read the k-mer
# prefilter's search sets cover, whether the read is matched reversed, how the type is written in an adapter
# specification.  SingleAdapter builds aligner, prefilter and names from the row (_KIND); the classes below carry the
# reference's names and hierarchy (adapters.py:684-1089: isinstance checks in AdapterIndex, LinkedAdapter, the parser and
# the modifiers go by it) and nothing else.  `force_anywhere` (the linked adapter's non-anchored parts, :1166-1183) lets a
# match start or end anywhere: Where.ANYWHERE, and the search sets of the other side as well -- "force" in the table.
_Kind = collections.namedtuple("_Kind", "description identifier where reverse remove_before kmer_back kmer_front internal spec "
                                        "force_where", defaults=(False,))
_FORCE = "force"


class _TableAdapter(SingleAdapter):
    _KIND: _Kind = None
    _COMPARER = None                 # anchored types without indels: a Hamming comparer instead of the aligner

    def __init__(self, sequence: str, *args, **kwargs):
        self._force_anywhere = bool(kwargs.pop("force_anywhere", False)) if self._KIND.where is not Where.ANYWHERE else False
        if not self.allows_partial_matches:
            kwargs["min_overlap"] = len(sequence)             # anchored: the whole adapter or nothing (:1029, :1066)
        super().__init__(sequence, *args, **kwargs)

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        kind = cls._KIND
        cls.description, cls._reverse_reads, cls._remove_before = kind.description, kind.reverse, kind.remove_before

    def descriptive_identifier(self) -> str:
        return self._KIND.identifier

    def spec(self) -> str:
        return self._KIND.spec.format(self.sequence)

    def _side(self, flag) -> bool:
        return self._force_anywhere if flag == _FORCE else bool(flag)

    def _aligner(self):
        kind = self._KIND
        if self._COMPARER is not None and not self.indels:
            return self._COMPARER(self.sequence, self.max_error_rate, wildcard_ref=self.adapter_wildcards,
                                  wildcard_query=self.read_wildcards, min_overlap=self.min_overlap)
        # (force_anywhere frees the aligner's ends for the regular and rightmost types only; the non-internal and anchored
        # ones keep their ends and merely search the other side's k-mers as well, reference :944-1089)
        where = Where.ANYWHERE if (self._force_anywhere and kind.force_where) else kind.where
        return self._make_aligner(self.sequence[::-1] if kind.reverse else self.sequence, where.value)

    def _kmer_finder(self):
        kind = self._KIND
        if self._COMPARER is not None and isinstance(self.aligner, self._COMPARER):
            return MockKmerFinder()                           # no DP to save (reference :1043-1049, :1080-1086)
        return self._make_kmer_finder(self.sequence[::-1] if kind.reverse else self.sequence,
                                      back_adapter=self._side(kind.kmer_back), front_adapter=self._side(kind.kmer_front),
                                      internal=kind.internal)


class FrontAdapter(_TableAdapter):
    """A 5' adapter (reference adapters.py:684-730)."""
    _KIND = _Kind("regular 5'", "regular_five_prime", Where.FRONT, False, True, _FORCE, True, True, "{}...", force_where=True)


class RightmostFrontAdapter(FrontAdapter):
    """A 5' adapter that prefers rightmost matches: the reversed adapter on the reversed read (reference :733-789)."""
    _KIND = _Kind("rightmost 5'", "rightmost_five_prime", Where.BACK, True, True, True, _FORCE, True, "{}...;rightmost", force_where=True)


class BackAdapter(_TableAdapter):
    """A 3' adapter (reference adapters.py:792-838)."""
    _KIND = _Kind("regular 3'", "regular_three_prime", Where.BACK, False, False, True, _FORCE, True, "{}", force_where=True)


class RightmostBackAdapter(BackAdapter):
    """A 3' adapter that prefers rightmost matches (reference :841-893)."""
    _KIND = _Kind("rightmost 3'", "rightmost_three_prime", Where.FRONT, True, False, _FORCE, True, True, "{};rightmost", force_where=True)


class AnywhereAdapter(_TableAdapter):
    """5' or 3': a match that starts at read position 0 is treated as a 5' adapter (reference adapters.py:896-941)."""
    _KIND = _Kind("variable 5'/3'", "anywhere", Where.ANYWHERE, False, False, True, True, True, "...{}...")

    def _wrap(self, alignment, sequence: str):
        # the reference upper-cases the read before locate (:925); every table used by the
        # kernels is case-insensitive on the read side, so that step is a no-op here
        cls = RemoveBeforeMatch if alignment[2] == 0 else RemoveAfterMatch   # index 2 is rstart
        return cls(*alignment, adapter=self, sequence=sequence)

    def _remove_before_array(self, coords, found) -> np.ndarray:
        return found & (coords[:, 2] == 0)


class NonInternalFrontAdapter(FrontAdapter):
    """A non-internal 5' adapter (reference adapters.py:944-978)."""
    _KIND = _Kind("non-internal 5'", "noninternal_five_prime", Where.FRONT_NOT_INTERNAL, False, True, _FORCE, True, False, "X{}...")


class NonInternalBackAdapter(BackAdapter):
    """A non-internal 3' adapter (reference adapters.py:981-1015)."""
    _KIND = _Kind("non-internal 3'", "noninternal_three_prime", Where.BACK_NOT_INTERNAL, False, False, True, _FORCE, False, "{}X")


class PrefixAdapter(NonInternalFrontAdapter):
    """An anchored 5' adapter (reference adapters.py:1018-1052)."""
    _KIND = _Kind("anchored 5'", "anchored_five_prime", Where.PREFIX, False, True, _FORCE, True, False, "^{}...")
    _COMPARER = PrefixComparer
    allows_partial_matches = False


class SuffixAdapter(NonInternalBackAdapter):
    """An anchored 3' adapter (reference adapters.py:1055-1089)."""
    _KIND = _Kind("anchored 3'", "anchored_three_prime", Where.SUFFIX, False, False, True, _FORCE, False, "{}$")
    _COMPARER = SuffixComparer
    allows_partial_matches = False


# -------------------------------------------------------------------------------------------------
# linked adapters (reference adapters.py:1092-1243)
# -------------------------------------------------------------------------------------------------
class LinkedMatch(Match):
    """The 5' and / or 3' part of a linked adapter found in a read (reference adapters.py:1092-1153); the 3' match's
    coordinates are relative to what the 5' match left."""

    def __init__(self, front_match: Optional[RemoveBeforeMatch], back_match: Optional[RemoveAfterMatch],
                 adapter: "LinkedAdapter"):
        assert front_match is not None or back_match is not None
        self.front_match, self.back_match, self.adapter = front_match, back_match, adapter

    def _parts(self):
        return [m for m in (self.front_match, self.back_match) if m is not None]

    def __repr__(self):
        return f"<LinkedMatch(front_match={self.front_match!r}, back_match={self.back_match}, adapter={self.adapter})>"

    score = property(lambda self: sum(m.score for m in self._parts()))
    errors = property(lambda self: sum(m.errors for m in self._parts()))

    def trimmed(self, read):
        for m in self._parts():                               # 5' first: the 3' match was found in what that leaves
            read = m.trimmed(read)
        return read

    def remainder_interval(self) -> Tuple[int, int]:
        return remainder(self._parts())

    def retained_adapter_interval(self) -> Tuple[int, int]:
        front, back = self.front_match, self.back_match
        start, shift = (front.rstart, front.rstop) if front else (0, 0)
        return start, (back.rstop + shift if back else len(front.sequence))

    def match_sequence(self):
        return ",".join(m.match_sequence() if m else "" for m in (self.front_match, self.back_match))


def remainder(matches: Sequence[Match]) -> Tuple[int, int]:
    """Interval of the original read that remains after applying the matches in order
    (reference adapters.py:1588-1602)."""
    if not matches:
        raise ValueError("matches must not be empty")
    start = 0
    for match in matches:
        match_start, match_stop = match.remainder_interval()
        start += match_start
    length = match_stop - match_start
    return (start, start + length)


class LinkedBatchMatches:
    """Array-form result of LinkedAdapter.match_to_batch: front and back stage results plus
    the per-read verdict of the required/optional rule (reference adapters.py:1219-1227).
    Back coordinates are relative to the read *after* front trimming, like the reference's
    second match_to call."""

    def __init__(self, front: BatchMatches, back: BatchMatches, found: np.ndarray,
                 adapter: "LinkedAdapter", reads):
        self.front = front
        self.back = back
        self.found = found
        self.adapter = adapter
        self._reads = reads
        self._strings = None

    def __len__(self):
        return len(self.found)

    def match(self, i: int) -> Optional[LinkedMatch]:
        if not self.found[i]:
            return None
        if self._strings is None:
            self._strings = self._reads.to_strings()
        seq = self._strings[i]
        fm = bm = None
        if self.front.found[i]:
            fm = RemoveBeforeMatch(*(int(v) for v in self.front.coords[i]),
                                   adapter=self.adapter.front_adapter, sequence=seq)
            seq = seq[fm.trim_slice()]
        if self.back.found[i]:
            bm = RemoveAfterMatch(*(int(v) for v in self.back.coords[i]),
                                  adapter=self.adapter.back_adapter, sequence=seq)
        return LinkedMatch(fm, bm, self.adapter)

    def matches(self):
        return [self.match(i) for i in range(len(self))]


class LinkedAdapter(Adapter):
    """A 5' adapter combined with a 3' adapter (reference adapters.py:1181-1243)."""

    description = "linked"

    def __init__(self, front_adapter: SingleAdapter, back_adapter: SingleAdapter,
                 front_required: bool, back_required: bool, name: Optional[str]):
        super().__init__(name)
        self.name = _generate_adapter_name() if name is None else name
        self.where = "linked"
        self.front_adapter, self.back_adapter = front_adapter, back_adapter
        self.front_required, self.back_required = front_required, back_required
        front_adapter.name = self.name                       # (the 5' part reports under the pair's name, reference :1199)

    def __repr__(self):
        return f"{type(self).__name__}(front_adapter={self.front_adapter}, back_adapter={self.back_adapter})"

    def descriptive_identifier(self) -> str:
        return "linked"

    def match_to(self, sequence: str) -> Optional[LinkedMatch]:
        """reference :1215-1227: the 3' part is searched in what the 5' match leaves; a required part that is missing, or
        no part at all, is no match"""
        front = self.front_adapter.match_to(sequence)
        if front is None and self.front_required:
            return None
        back = self.back_adapter.match_to(sequence if front is None else sequence[front.trim_slice()])
        if back is None and (front is None or self.back_required):
            return None
        return LinkedMatch(front, back, self)

    def match_to_batch(self, batch) -> LinkedBatchMatches:
        """Two dependent stages: the back adapter is searched in the suffix after the front
        match (device-side view, no copy)."""
        fa, ba = self.front_adapter, self.back_adapter
        if fa._reverse_reads or ba._reverse_reads:
            # Rightmost* parts work on reversed copies: stage by stage through the adapter classes
            import torch
            front = fa.match_to_batch(batch)
            lens = batch.lengths()
            rstop = torch.from_numpy(np.where(front.found, front.coords[:, 3], 0)).to(batch.device)
            view = batch.view(rstop, lens - rstop, check=False)       # (inside its read by construction)
            back = ba.match_to_batch(view)
        else:
            from . import batch as _b
            f_res, b_res, view = _b.linked_match_batch(fa._fused_plan, ba._fused_plan, batch)
            front = fa._batch_matches(f_res, batch)
            back = ba._batch_matches(b_res, view)
        ok = np.ones(len(front), dtype=bool)
        if self.front_required:
            ok &= front.found
        ok &= back.found | ~(np.full(len(front), self.back_required) | ~front.found)
        # reads rejected by the front_required rule never had a meaningful back stage
        back.found = back.found & ok
        front.found = front.found & ok
        return LinkedBatchMatches(front, back, ok, self, batch)

    @property
    def sequence(self):
        return self.front_adapter.sequence + "..." + self.back_adapter.sequence

    @property
    def remove(self):
        return None

    def spec(self) -> str:
        return f"{self.front_adapter.spec()}...{self.back_adapter.spec()}"


# -------------------------------------------------------------------------------------------------
# index of anchored adapters (reference adapters.py:1289-1567)
# -------------------------------------------------------------------------------------------------
class AdapterIndex:
    """Index of multiple anchored adapters of one kind (reference adapters.py:1289-1551).

    The dictionary {string within k errors of an adapter: (adapter, errors, matches)} is built and
    held by the library (``cah_index_create``; hash table in HBM), reads are matched with one lookup
    kernel (``cah_index_lookup_batch``).  Restrictions as in the reference (:1296-1299): at most 3
    errors, no wildcards in adapters or reads.  Plain ACGT adapters of up to 60 characters get the
    2-bit packed table; longer ones (up to 1000 characters) and adapters with other characters
    (possible with ``adapter_wildcards=False``) a table of hashed byte strings."""

    def __init__(self, adapters, prefix: bool):
        if not adapters:
            raise ValueError("Adapter list is empty")
        for adapter in adapters:
            self._accept(adapter, prefix)
        self._adapters = list(adapters)
        self._prefix = bool(prefix)
        self._h = _lib.Index(
            [(a.sequence, a.max_error_rate, a.indels,
              a.kmer_finder.searched_positions_and_kmers if isinstance(a.kmer_finder, KmerFinder) else None)
             for a in self._adapters], prefix)
        self._lengths = list(self._h.lengths)
        self._ambiguous = self._h.n_ambiguous

    def __repr__(self):
        return f"{self.__class__.__name__}(adapters={self._adapters!r})"

    def __len__(self):
        return self._h.n_strings

    @classmethod
    def _accept(cls, adapter, prefix: bool):
        """Raise a ValueError if the adapter is not acceptable (reference :1373-1386)"""
        if prefix and not isinstance(adapter, PrefixAdapter):
            raise ValueError("Only 5' anchored adapters are allowed")
        elif not prefix and not isinstance(adapter, SuffixAdapter):
            raise ValueError("Only 3' anchored adapters are allowed")
        if adapter.read_wildcards:
            raise ValueError("Wildcards in the read not supported")
        if adapter.adapter_wildcards:
            raise ValueError("Wildcards in the adapter not supported")
        k = int(len(adapter) * adapter.max_error_rate)
        if k > 3:
            raise ValueError("Error rate too high")
        if len(adapter.sequence) > 1000:
            raise ValueError("Adapters of more than 1000 characters cannot be indexed by this build")

    @classmethod
    def is_acceptable(cls, adapter, prefix: bool) -> bool:
        try:
            cls._accept(adapter, prefix)
        except ValueError:
            return False
        return True

    def lookup(self, string: str):
        """The dictionary entry of one string: (adapter, errors, matches) or None."""
        r = self._h.get(string)
        return None if r is None else (self._adapters[r[0]], r[1], r[2])

    def match_to_batch(self, batch) -> BatchMatches:
        import torch
        n = batch.n_reads
        dev = batch.device
        out6 = torch.zeros((n, 6), dtype=torch.int32, device=dev)
        status = torch.zeros(n, dtype=torch.uint8, device=dev)
        best = torch.zeros(n, dtype=torch.int32, device=dev)
        if n:
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().cah_index_lookup_batch(
                    self._h.handle, batch.seqs.data_ptr(), batch.offsets.data_ptr(), batch._lens_ptr(), n,
                    out6.data_ptr(), best.data_ptr(), status.data_ptr(),
                    torch.cuda.current_stream(dev).cuda_stream))
        status = status.cpu().numpy()
        _raise_if_invalid(status)
        found = status == _lib.MATCH
        coords = out6.cpu().numpy().astype(np.int64)
        best = np.where(found, best.cpu().numpy(), 0).astype(np.int32)
        return BatchMatches(coords, found, best, self._adapters, np.full(n, self._prefix, dtype=bool), reads=batch)

    def match_to(self, sequence: str):
        """Best match over all indexed adapters or None (reference :1468-1530)"""
        seqs = np.frombuffer(_lib._ascii(sequence), dtype=np.uint8)
        offsets = np.array([0, len(seqs)], dtype=np.int64)
        out6 = np.zeros(6, dtype=np.int32)
        best = np.zeros(1, dtype=np.int32)
        status = np.zeros(1, dtype=np.uint8)
        _lib.check(_lib.lib().cah_index_lookup_batch_host(
            self._h.handle, seqs.ctypes.data if len(seqs) else None, offsets.ctypes.data, 1,
            out6.ctypes.data, best.ctypes.data, status.ctypes.data))
        if status[0] != _lib.MATCH:
            return None
        cls = RemoveBeforeMatch if self._prefix else RemoveAfterMatch
        return cls(*(int(v) for v in out6), adapter=self._adapters[int(best[0])], sequence=sequence)


class IndexedPrefixAdapters(Matchable):
    def __init__(self, adapters):
        super().__init__(name="indexed_prefix_adapters")
        self._index = AdapterIndex(adapters, prefix=True)
        self._adapters = self._index._adapters

    def match_to(self, sequence: str):
        return self._index.match_to(sequence)

    def match_to_batch(self, batch) -> BatchMatches:
        return self._index.match_to_batch(batch)


class IndexedSuffixAdapters(Matchable):
    def __init__(self, adapters):
        super().__init__(name="indexed_suffix_adapters")
        self._index = AdapterIndex(adapters, prefix=False)
        self._adapters = self._index._adapters

    def match_to(self, sequence: str):
        return self._index.match_to(sequence)

    def match_to_batch(self, batch) -> BatchMatches:
        return self._index.match_to_batch(batch)


# -------------------------------------------------------------------------------------------------
# several adapters (reference adapters.py:1246-1286)
# -------------------------------------------------------------------------------------------------
class MultipleAdapters(Matchable):
    """Best match over several adapters: higher score wins, then fewer errors, then the
    adapter that comes first (reference adapters.py:1278-1285)."""

    def __getstate__(self):
        return dict(self.__dict__, _plan=None)               # (the fused plan is a library handle: rebuilt on first use)

    def __init__(self, adapters: Sequence[Matchable]):
        super().__init__(name="multiple_adapters")
        self._adapters = list(adapters)
        self._fusable = all(isinstance(a, SingleAdapter) and not a._reverse_reads for a in self._adapters)
        self._plan = None

    def __getitem__(self, item):
        return self._adapters[item]

    def __len__(self):
        return len(self._adapters)

    def match_to(self, sequence: str):
        best_match = None
        for adapter in self._adapters:
            match = adapter.match_to(sequence)
            if match is None:
                continue
            if (best_match is None or match.score > best_match.score
                    or (match.score == best_match.score and match.errors < best_match.errors)):
                best_match = match
        return best_match

    def match_to_batch(self, batch) -> BatchMatches:
        """All adapters in one fused library call (one plan holding every matcher; the kernels
        keep the best match per read on the device)."""
        from . import batch as _b
        if not self._fusable:
            return self._match_to_batch_unfused(batch)
        if self._plan is None:
            self._plan = _lib.Plan([a.matcher_spec() for a in self._adapters])
        res = _b.match_batch(self._plan, batch)
        out6, status, best = res.cpu()
        _raise_if_invalid(status, batch)
        found = status == _lib.MATCH
        coords = out6.astype(np.int64)
        best = np.where(found, best, 0).astype(np.int32)
        before_by_adapter = np.array([a._remove_before for a in self._adapters], dtype=bool)
        remove_before = before_by_adapter[best]
        anywhere = np.array([isinstance(a, AnywhereAdapter) for a in self._adapters], dtype=bool)
        remove_before = np.where(anywhere[best], found & (coords[:, 2] == 0), remove_before)
        return BatchMatches(coords, found, best, self._adapters, remove_before, reads=batch)

    def _match_to_batch_unfused(self, batch) -> BatchMatches:
        n = batch.n_reads
        coords = np.zeros((n, 6), dtype=np.int64)
        found = np.zeros(n, dtype=bool)
        best = np.zeros(n, dtype=np.int32)
        remove_before = np.zeros(n, dtype=bool)
        for idx, adapter in enumerate(self._adapters):
            r = adapter.match_to_batch(batch)
            better = r.found & (~found | (r.coords[:, 4] > coords[:, 4])
                                | ((r.coords[:, 4] == coords[:, 4]) & (r.coords[:, 5] < coords[:, 5])))
            coords[better] = r.coords[better]
            best[better] = idx
            remove_before[better] = r.remove_before[better]
            found |= better
        return BatchMatches(coords, found, best, self._adapters, remove_before, reads=batch)
