"""cutadapt_amd -- MI355X-native (gfx950) adapter matching for cutadapt's hot path.

Only the error-tolerant adapter matcher is here: ``Aligner.locate`` (banded semi-global DP),
the k-mer prefilter ``KmerFinder.kmers_present`` and the ``*Adapter.match_to`` wrappers, with
the reference's Python API (reference src/cutadapt/_align.pyx, _kmer_finder.pyx, adapters.py)
and batch forms that keep reads and results in HBM.  The work runs in hand-written HIP kernels
behind the C ABI of include/cutadapt_hip.h; there is no CPU fallback.
"""
__version__ = "0.1.0"

__all__ = ["Aligner", "PrefixComparer", "SuffixComparer", "EndSkip", "KmerFinder",
           "create_positions_and_kmers", "ReadBatch", "adapters"]


def __getattr__(name):
    # lazy: importing the package must not require the built library (build() imports it first)
    if name in ("Aligner", "PrefixComparer", "SuffixComparer", "EndSkip"):
        from . import align
        return getattr(align, name)
    if name == "KmerFinder":
        from ._kmer_finder import KmerFinder
        return KmerFinder
    if name == "create_positions_and_kmers":
        from .kmer_heuristic import create_positions_and_kmers
        return create_positions_and_kmers
    if name == "ReadBatch":
        from .batch import ReadBatch
        return ReadBatch
    if name == "adapters":
        import importlib
        return importlib.import_module(".adapters", __name__)
    raise AttributeError(name)
