"""Import the compiled reference hot path from oracle/_ref/ (test infrastructure only).

TEST INFRASTRUCTURE -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may use this.  The product package (cutadapt_amd) never imports it.

``load()`` returns a namespace with the reference's own classes:
    ref.Aligner, ref.PrefixComparer, ref.SuffixComparer      (reference _align.pyx)
    ref.KmerFinder                                            (reference _kmer_finder.pyx)
    ref.create_positions_and_kmers                            (reference kmer_heuristic.py)
    ref.adapters                                              (reference adapters.py module)
    ref.qualtrim                                              (reference qualtrim.pyx module, or None)
or None when oracle/_ref has not been built (see oracle/build_ref.py).
"""
import importlib
import os
import sys
import types

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_ROOT = os.path.join(_HERE, "_ref")
_cached = None


def available() -> bool:
    from . import build_ref
    return build_ref.is_built()


def load():
    global _cached
    if _cached is not None:
        return _cached
    if not os.path.exists(os.path.join(_REF_ROOT, "cutadapt", "__init__.py")):
        return None
    # reference align.py does ``from cutadapt._align import ...`` (absolute), so the
    # compiled package has to be importable under its real name.
    if _REF_ROOT not in sys.path:
        sys.path.insert(0, _REF_ROOT)
    if "dnaio" not in sys.modules:
        # reference adapters.py does not import dnaio, but be defensive: nothing on the
        # matching path needs it.
        pass
    try:
        adapters = importlib.import_module("cutadapt.adapters")
        align = importlib.import_module("cutadapt.align")
        kf = importlib.import_module("cutadapt._kmer_finder")
        kh = importlib.import_module("cutadapt.kmer_heuristic")
        mt = importlib.import_module("cutadapt._match_tables")
    except ImportError:
        return None
    try:
        qualtrim = importlib.import_module("cutadapt.qualtrim")
    except ImportError:
        qualtrim = None
    ns = types.SimpleNamespace(
        adapters=adapters,
        align=align,
        Aligner=align.Aligner,
        PrefixComparer=align.PrefixComparer,
        SuffixComparer=align.SuffixComparer,
        EndSkip=align.EndSkip,
        KmerFinder=kf.KmerFinder,
        create_positions_and_kmers=kh.create_positions_and_kmers,
        match_tables=mt,
        qualtrim=qualtrim,
    )
    _cached = ns
    return ns
