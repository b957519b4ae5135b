"""ctypes front-end of the CPU oracle (oracle/cutadapt_oracle.c, oracle/synth_reads.c).

TEST INFRASTRUCTURE -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module.  The product package never does.

The classes mirror the reference's signatures (reference src/cutadapt/_align.pyi:10-27,
_kmer_finder.pyi:5-12) so that parity tests can run the same call on the oracle, on the
compiled reference (oracle/ref_loader.py) and on the HIP path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcutadapt_oracle.so")
_lib = None

_ERRORS = {
    1: "Cannot have only N wildcards in the sequence",
    2: "indel_cost must be at least 1",
    3: "String must contain only ASCII characters",
    4: "max_error_rate must be between 0 and 1",
    5: "min_overlap must be at least 1",
    6: "kmer is longer than the maximum of 64",
}


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("cutadapt_oracle.c", "synth_reads.c")]
    if force or not os.path.exists(_LIB_PATH) or \
            any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.run(["make", "-s", "-C", _HERE, "-B", "libcutadapt_oracle.so"], check=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    L.orc_aligner_new.restype = vp
    L.orc_aligner_new.argtypes = [C.c_char_p, i32, dbl, i32, i32, i32, i32, i32, C.POINTER(i32)]
    L.orc_aligner_free.argtypes = [vp]
    L.orc_aligner_effective_length.argtypes = [vp]
    L.orc_aligner_locate.argtypes = [vp, C.c_char_p, i32, C.POINTER(i32)]
    L.orc_comparer_new.restype = vp
    L.orc_comparer_new.argtypes = [C.c_char_p, i32, dbl, i32, i32, i32, i32, C.POINTER(i32)]
    L.orc_comparer_free.argtypes = [vp]
    L.orc_comparer_effective_length.argtypes = [vp]
    L.orc_comparer_locate.argtypes = [vp, C.c_char_p, i32, C.POINTER(i32)]
    L.orc_kmer_finder_new.restype = vp
    L.orc_kmer_finder_new.argtypes = [i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32),
                                      C.POINTER(C.c_char_p), i32, i32, C.POINTER(i32)]
    L.orc_kmer_finder_free.argtypes = [vp]
    L.orc_kmer_finder_n_entries.argtypes = [vp]
    L.orc_kmers_present.argtypes = [vp, C.c_char_p, i64]
    L.orc_locate_batch.argtypes = [vp, vp, vp, i64, vp, vp]
    L.orc_comparer_batch.argtypes = [vp, vp, vp, i64, vp, vp]
    L.orc_kmers_present_batch.argtypes = [vp, vp, vp, i64, vp]
    L.orc_match_batch.argtypes = [vp, vp, vp, vp, i64, vp, vp]
    L.orc_synth_reads.argtypes = [C.c_uint64, i64, i64, i32, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_char_p, C.POINTER(i32), i32, vp]
    _lib = L
    return L


def _encode(s: str) -> bytes:
    try:
        return s.encode("ascii")
    except UnicodeEncodeError:
        raise ValueError("String must contain only ASCII characters")


def _raise(err):
    raise ValueError(_ERRORS.get(err, f"oracle error {err}"))


def pack_reads(reads):
    """list[str|bytes] -> (uint8[total], int64[n+1]) in the packed layout used everywhere."""
    bs = [r if isinstance(r, (bytes, bytearray)) else r.encode("latin-1") for r in reads]
    offsets = np.zeros(len(bs) + 1, dtype=np.int64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs])
    seqs = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return seqs, offsets


def _batch_out(n):
    return np.zeros((n, 6), dtype=np.int32), np.zeros(n, dtype=np.uint8)


class Aligner:
    def __init__(self, reference, max_error_rate, flags=15, wildcard_ref=False,
                 wildcard_query=False, indel_cost=1, min_overlap=1):
        self._args = (reference, max_error_rate, int(flags), bool(wildcard_ref),
                      bool(wildcard_query), int(indel_cost), int(min_overlap))
        err = C.c_int(0)
        ref = _encode(reference)
        self._h = lib().orc_aligner_new(ref, len(ref), float(max_error_rate), int(flags),
                                        int(bool(wildcard_ref)), int(bool(wildcard_query)),
                                        int(indel_cost), int(min_overlap), C.byref(err))
        if not self._h:
            _raise(err.value)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_aligner_free(self._h)
            self._h = None

    @property
    def effective_length(self):
        return lib().orc_aligner_effective_length(self._h)

    def locate(self, query: str):
        q = _encode(query)
        out = (C.c_int * 6)()
        rc = lib().orc_aligner_locate(self._h, q, len(q), out)
        if rc < 0:
            _raise(3)
        return tuple(out) if rc == 1 else None

    def locate_batch(self, seqs: np.ndarray, offsets: np.ndarray):
        n = len(offsets) - 1
        out6, status = _batch_out(n)
        lib().orc_locate_batch(self._h, seqs.ctypes.data, offsets.ctypes.data, n,
                               out6.ctypes.data, status.ctypes.data)
        return out6, status


class _Comparer:
    _suffix = 0

    def __init__(self, reference, max_error_rate, wildcard_ref=False, wildcard_query=False,
                 min_overlap=1):
        err = C.c_int(0)
        ref = _encode(reference)
        self._h = lib().orc_comparer_new(ref, len(ref), float(max_error_rate),
                                         int(bool(wildcard_ref)), int(bool(wildcard_query)),
                                         int(min_overlap), self._suffix, C.byref(err))
        if not self._h:
            _raise(err.value)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_comparer_free(self._h)
            self._h = None

    @property
    def effective_length(self):
        return lib().orc_comparer_effective_length(self._h)

    def locate(self, query: str):
        q = _encode(query)
        out = (C.c_int * 6)()
        rc = lib().orc_comparer_locate(self._h, q, len(q), out)
        if rc < 0:
            _raise(3)
        return tuple(out) if rc == 1 else None

    def locate_batch(self, seqs, offsets):
        n = len(offsets) - 1
        out6, status = _batch_out(n)
        lib().orc_comparer_batch(self._h, seqs.ctypes.data, offsets.ctypes.data, n,
                                 out6.ctypes.data, status.ctypes.data)
        return out6, status


class PrefixComparer(_Comparer):
    _suffix = 0


class SuffixComparer(_Comparer):
    _suffix = 1


class KmerFinder:
    def __init__(self, positions_and_kmers, ref_wildcards=False, query_wildcards=False):
        sets = list(positions_and_kmers)
        n = len(sets)
        starts = (C.c_int64 * max(n, 1))()
        stops = (C.c_int64 * max(n, 1))()
        first = (C.c_int * (n + 1))()
        flat = []
        for i, (start, stop, kmers) in enumerate(sets):
            starts[i] = start
            stops[i] = 0 if stop is None else stop
            first[i] = len(flat)
            for k in kmers:
                if not isinstance(k, str):
                    raise TypeError(f"Kmer should be a string not {type(k)}")
                flat.append(_encode(k))
        first[n] = len(flat)
        arr = (C.c_char_p * max(len(flat), 1))(*flat)
        err = C.c_int(0)
        self._h = lib().orc_kmer_finder_new(n, starts, stops, first, arr,
                                            int(bool(ref_wildcards)), int(bool(query_wildcards)),
                                            C.byref(err))
        if not self._h:
            _raise(err.value)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_kmer_finder_free(self._h)
            self._h = None

    def kmers_present(self, sequence: str) -> bool:
        q = _encode(sequence)
        rc = lib().orc_kmers_present(self._h, q, len(q))
        if rc < 0:
            _raise(3)
        return bool(rc)

    def kmers_present_batch(self, seqs, offsets):
        n = len(offsets) - 1
        present = np.zeros(n, dtype=np.uint8)
        lib().orc_kmers_present_batch(self._h, seqs.ctypes.data, offsets.ctypes.data, n,
                                      present.ctypes.data)
        return present


def match_batch(aligner: Aligner, finder, seqs, offsets):
    """kmers_present -> locate, the body of BackAdapter/FrontAdapter.match_to
    (reference adapters.py:707-724, 815-832) before the tuple is wrapped in a Match."""
    n = len(offsets) - 1
    out6, status = _batch_out(n)
    lib().orc_match_batch(aligner._h, finder._h if finder is not None else None,
                          seqs.ctypes.data, offsets.ctypes.data, n,
                          out6.ctypes.data, status.ctypes.data)
    return out6, status


def prob_u32(p: float) -> int:
    return min(int(round(p * 4294967296.0)), 4294967295)


def prob_u16(p: float) -> int:
    return min(int(round(p * 65536.0)), 65535)


def synth_reads(seed, first_index, n_reads, read_len, adapters, p_adapter=0.25, p_edit=0.02,
                p_n=0.005):
    """CPU twin of cah_synth_reads; returns (uint8[n*read_len], int64[n+1])."""
    ads = [a.encode("ascii") for a in adapters]
    off = (C.c_int * (len(ads) + 1))()
    for i, a in enumerate(ads):
        off[i + 1] = off[i] + len(a)
    seqs = np.empty(n_reads * read_len, dtype=np.uint8)
    lib().orc_synth_reads(seed, first_index, n_reads, read_len, prob_u32(p_adapter),
                          prob_u32(p_edit), prob_u16(p_n), b"".join(ads), off, len(ads),
                          seqs.ctypes.data)
    offsets = np.arange(n_reads + 1, dtype=np.int64) * read_len
    return seqs, offsets
