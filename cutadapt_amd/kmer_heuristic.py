"""Search-set construction for the k-mer prefilter (host-side setup, pure Python).

Restates reference src/cutadapt/kmer_heuristic.py:6-164 (pigeonhole principle: an adapter
occurrence with at most e errors must contain at least one of e+1 disjoint chunks of the
adapter verbatim).  The output format is the reference's ``positions_and_kmers`` list
``[(start, stop, [kmers]), ...]`` that KmerFinder consumes; ``stop`` None means "end of read",
negative ``start`` counts from the end of the read.

k-mer order inside one search set is not significant (kmers_present is an OR over all of
them); tests compare the sets order-insensitively against the reference.
"""
from collections import OrderedDict
from typing import Dict, List, Optional, Set, Tuple

SearchSet = Tuple[int, Optional[int], Set[str]]
PositionsAndKmers = List[Tuple[int, Optional[int], List[str]]]

# below this overlap length even exact k-mers are too unspecific to be chunked
# (reference kmer_heuristic.py:104-113)
_SHORT_OVERLAP_KMER = 5


def kmer_chunks(sequence: str, chunks: int) -> Set[str]:
    """Split ``sequence`` into ``chunks`` nearly equal consecutive pieces, longer pieces first
    (reference kmer_heuristic.py:6-21)."""
    base, extra = divmod(len(sequence), chunks)
    pieces = set()
    pos = 0
    for i in range(chunks):
        size = base + (1 if i < extra else 0)
        pieces.add(sequence[pos:pos + size])
        pos += size
    return pieces


def _error_length_classes(adapter_length: int, error_rate: float) -> List[Tuple[int, int]]:
    """[(allowed_errors, longest overlap length with that allowance), ...]
    (reference kmer_heuristic.py:94-98)."""
    classes = []
    allowed = 0
    for length in range(adapter_length + 1):
        if int(length * error_rate) > allowed:
            classes.append((allowed, length - 1))
            allowed += 1
    classes.append((allowed, adapter_length))
    return classes


def create_back_overlap_searchsets(adapter: str, min_overlap: int, error_rate: float) -> List[SearchSet]:
    """Search sets for a 3' adapter that may run off the end of the read
    (reference kmer_heuristic.py:87-117)."""
    search_sets: List[SearchSet] = []
    shortest = min_overlap
    for allowed_errors, longest in _error_length_classes(len(adapter), error_rate):
        if shortest > longest:
            continue
        if allowed_errors == 0 and shortest < _SHORT_OVERLAP_KMER:
            # very short overlaps: look for the exact prefix at exactly its position
            for size in range(shortest, _SHORT_OVERLAP_KMER):
                search_sets.append((-size, None, {adapter[:size]}))
            shortest = _SHORT_OVERLAP_KMER
        search_sets.append((-longest, None, kmer_chunks(adapter[:shortest], allowed_errors + 1)))
        shortest = longest + 1
    return search_sets


def minimize_kmer_search_list(kmer_search_list):
    """A k-mer that is searched in several windows only needs the widest one
    (reference kmer_heuristic.py:29-64)."""
    windows: Dict[str, List[Tuple[int, Optional[int]]]] = OrderedDict()
    for kmer, start, stop in kmer_search_list:
        windows.setdefault(kmer, []).append((start, stop))
    result = []
    for kmer, positions in windows.items():
        if len(positions) == 1:
            result.append((kmer,) + positions[0])
            continue
        if (0, None) in positions:
            result.append((kmer, 0, None))
            continue
        front = [stop for start, stop in positions if start == 0]
        back = [start for start, stop in positions if stop is None]
        if any(start != 0 and stop is not None for start, stop in positions):
            raise NotImplementedError(
                "Situations with searches starting in the middle have not been considered.")
        if front:
            result.append((kmer, 0, max(front)))
        if back:
            result.append((kmer, min(back), None))
    return result


def remove_redundant_kmers(search_sets: List[SearchSet]) -> PositionsAndKmers:
    """reference kmer_heuristic.py:67-84"""
    flat = [(kmer, start, stop) for start, stop, kmers in search_sets for kmer in kmers]
    grouped: Dict[Tuple[int, Optional[int]], List[str]] = OrderedDict()
    for kmer, start, stop in minimize_kmer_search_list(flat):
        grouped.setdefault((start, stop), []).append(kmer)
    return [(start, stop, kmers) for (start, stop), kmers in grouped.items()]


def create_positions_and_kmers(adapter: str, min_overlap: int, error_rate: float,
                               back_adapter: bool, front_adapter: bool,
                               internal: bool = True) -> PositionsAndKmers:
    """reference kmer_heuristic.py:120-164"""
    max_errors = int(len(adapter) * error_rate)
    search_sets: List[SearchSet] = []
    if back_adapter:
        search_sets.extend(create_back_overlap_searchsets(adapter, min_overlap, error_rate))
    if front_adapter:
        # a 5' adapter is the mirror image: build the sets for the reversed adapter and flip
        # both the k-mers and the windows (reference :149-160)
        for start, _stop, kmers in create_back_overlap_searchsets(adapter[::-1], min_overlap, error_rate):
            search_sets.append((0, -start, {kmer[::-1] for kmer in kmers}))
    if internal:
        search_sets.append((0, None, kmer_chunks(adapter, max_errors + 1)))
    return remove_redundant_kmers(search_sets)
