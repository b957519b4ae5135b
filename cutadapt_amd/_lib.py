"""ctypes binding of libcutadapt_hip.so (the C ABI in include/cutadapt_hip.h).

There is no CPU fallback: if the library is missing or no HIP device is usable, every
operation raises.  (cffi is not installed in this image, hence ctypes.)
"""
import ctypes as C
import threading
import os
from typing import Optional, Sequence, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
# CAH_LIB_PATH: developer knob to A/B-test a differently built kernel library
LIB_PATH = os.environ.get("CAH_LIB_PATH") or os.path.join(_HERE, "libcutadapt_hip.so")

CAH_OK, CAH_EINVAL, CAH_ETYPE, CAH_EHIP, CAH_ENOMEM, CAH_EUNSUPPORTED, CAH_EINTERNAL = 0, 1, 2, 3, 4, 5, 6
NONE, MATCH, INVALID = 0, 1, 2
STATUS_INTERNAL = 255         # CAH_STATUS_INTERNAL (deferred error check: the batch's rows are void)
MAX_READ_LEN = 1000000        # CAH_MAX_READ_LEN of include/cutadapt_hip.h
MAX_NAME_SUFFIX = 32           # CAH_MAX_NAME_SUFFIX
KIND_ALIGNER, KIND_PREFIX, KIND_SUFFIX, KIND_KMER_ONLY = 0, 1, 2, 3
ABI_VERSION = 5              # CAH_ABI_VERSION of include/cutadapt_hip.h this binding was written against
PROF_FILTER, PROF_DP, PROF_COMPARER, PROF_SCAN, PROF_MERGE, PROF_N = 0, 1, 2, 3, 4, 5

# every symbol include/cutadapt_hip.h declares (tests check the library exports them all)
EXPORTED_SYMBOLS = [
    "cah_abi_version", "cah_build_id", "cah_last_error", "cah_device_count", "cah_set_device", "cah_device_info",
    "cah_plan_create", "cah_plan_destroy", "cah_plan_n_adapters", "cah_plan_effective_length",
    "cah_plan_n_kmer_entries", "cah_plan_prefilter_kind", "cah_plan_multi_kind", "cah_last_multi_path", "cah_plan_debug_matcher", "cah_plan_debug_lean", "cah_locate_batch", "cah_kmers_present_batch", "cah_match_batch", "cah_match_batch_uniform", "cah_match_batch_suffix_views", "cah_match_batch_views", "cah_match_batch_frames", "cah_set_deferred_errors", "cah_linked_views", "cah_linked_match_batch_uniform",
    "cah_workspace_bytes", "cah_plan_workspace_bytes", "cah_validate_ascii_batch", "cah_reverse_reads_batch", "cah_revcomp_reads_batch", "cah_locate_batch_host",
    "cah_kmers_present_batch_host", "cah_match_batch_host", "cah_locate_debug_host", "cah_match_one_host", "cah_locate_one_host", "cah_profile_enable",
    "cah_profile_reset", "cah_profile_read", "cah_synth_reads",
    "cah_fastq_scan", "cah_pack_sequences", "cah_fastq_write_trimmed",
    "cah_fasta_scan", "cah_records_write", "cah_info_write", "cah_info_write_rc", "cah_chunk_revcomp", "cah_chunk_select", "cah_fastq_span", "cah_record_boundary",
    "cah_fastq_device_scratch_bytes", "cah_fastq_count_lines_device", "cah_fastq_index_device", "cah_fastq_format_device", "cah_mark_reads_device", "cah_revcomp_in_place_device", "cah_fastq_format_suffix_device", "cah_info_format_device", "cah_trim_decide_device", "cah_trim_decide_window_device", "cah_trim_decide_action_device", "cah_trim_filter_device",
    "cah_index_create", "cah_index_destroy", "cah_index_info", "cah_index_get",
    "cah_index_lookup_batch", "cah_index_lookup_batch_host",
    "cah_quality_trim_batch", "cah_nextseq_trim_batch", "cah_nextseq_trim_batch_q", "cah_poly_a_trim_batch", "cah_expected_errors_batch",
]


class HipLibraryMissing(RuntimeError):
    pass


class UnsupportedByHipPath(ValueError):
    """Input outside the limits of this build (e.g. a read longer than MAX_READ_LEN characters)."""


class KmerSetC(C.Structure):
    _fields_ = [("start", C.c_int64), ("stop", C.c_int64),
                ("kmers", C.POINTER(C.c_char_p)), ("n_kmers", C.c_int32)]


class AdapterDescC(C.Structure):
    _fields_ = [("sequence", C.c_char_p), ("length", C.c_int32), ("max_error_rate", C.c_double),
                ("flags", C.c_int32), ("wildcard_ref", C.c_int32), ("wildcard_query", C.c_int32),
                ("indel_cost", C.c_int32), ("min_overlap", C.c_int32), ("kind", C.c_int32),
                ("kmer_sets", C.POINTER(KmerSetC)), ("n_kmer_sets", C.c_int32),
                ("kmer_ref_wildcards", C.c_int32), ("kmer_query_wildcards", C.c_int32)]


class IndexAdapterC(C.Structure):
    _fields_ = [("sequence", C.c_char_p), ("length", C.c_int32), ("max_error_rate", C.c_double),
                ("indels", C.c_int32), ("kmer_sets", C.POINTER(KmerSetC)), ("n_kmer_sets", C.c_int32)]


_lib = None


def _share_hip_runtime_with_torch() -> None:
    """PyTorch's ROCm wheel bundles its own libamdhip64.so (same SONAME libamdhip64.so.7 as
    /opt/rocm's).  Two HIP runtimes in one process fight over the device ("no ROCm-capable
    device is detected" in whichever initialises second), so make sure torch's copy is mapped
    first: the dynamic loader then resolves this library's NEEDED libamdhip64.so.7 to the
    already-loaded runtime and device pointers / streams can be exchanged with torch.  All HIP
    symbols this library uses are the long-stable hip_4.2 / hip_6.0 versions."""
    try:
        import torch  # noqa: F401
    except Exception:   # torch absent: the system runtime is used on its own
        pass


def lib():
    """Load the shared library (once).  Raises HipLibraryMissing if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -m cutadapt_amd.build` "
            "(the HIP path has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.cah_abi_version.restype = C.c_int
    # (CAH_LIB_ANY_ABI=1 with CAH_LIB_PATH: developer A/B runs against an older round's build of the library)
    any_abi = bool(os.environ.get("CAH_LIB_PATH")) and os.environ.get("CAH_LIB_ANY_ABI") == "1"
    if L.cah_abi_version() != ABI_VERSION and not any_abi:
        raise HipLibraryMissing(f"{LIB_PATH} has ABI version {L.cah_abi_version()}, this binding needs {ABI_VERSION}: "
                                "rebuild it with `python -m cutadapt_amd.build --force`")
    if hasattr(L, "cah_build_id"):
        L.cah_build_id.restype = C.c_char_p
    L.cah_last_error.argtypes = [C.c_char_p, C.c_size_t]
    L.cah_last_error.restype = None
    L.cah_device_count.argtypes = [C.POINTER(C.c_int)]
    L.cah_set_device.argtypes = [C.c_int]
    L.cah_device_info.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                  C.POINTER(C.c_int), C.POINTER(i64)]
    L.cah_plan_create.argtypes = [C.POINTER(AdapterDescC), i32, C.POINTER(vp)]
    L.cah_plan_destroy.argtypes = [vp]
    L.cah_plan_destroy.restype = None
    L.cah_plan_n_adapters.argtypes = [vp]
    L.cah_plan_effective_length.argtypes = [vp, i32, C.POINTER(i32)]
    L.cah_plan_n_kmer_entries.argtypes = [vp, i32, C.POINTER(i32)]
    L.cah_plan_prefilter_kind.argtypes = [vp, i32, C.POINTER(i32)]
    L.cah_plan_multi_kind.argtypes = [vp, i32, C.POINTER(i32)]
    L.cah_plan_debug_matcher.argtypes = [vp, i32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.cah_plan_debug_lean.argtypes = [vp, i32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.cah_locate_batch.argtypes = [vp, i32, vp, vp, vp, i64, vp, vp, vp, C.c_size_t, vp]
    L.cah_kmers_present_batch.argtypes = [vp, i32, vp, vp, vp, i64, vp, vp]
    L.cah_match_batch.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, vp, C.c_size_t, vp]
    L.cah_match_batch_uniform.argtypes = [vp, vp, i32, i64, vp, vp, vp, vp, C.c_size_t, vp]
    L.cah_match_batch_suffix_views.argtypes = [vp, vp, vp, vp, i32, i64, vp, vp, vp, vp, C.c_size_t, vp]
    if hasattr(L, "cah_match_batch_views") or not any_abi:
        L.cah_match_batch_views.argtypes = [vp, vp, vp, vp, i32, i64, vp, vp, vp, vp, C.c_size_t, vp]
    if hasattr(L, "cah_match_batch_frames") or not any_abi:
        L.cah_match_batch_frames.argtypes = [vp, vp, vp, vp, i32, i64, vp, vp, vp, vp, C.c_size_t, vp]
        L.cah_set_deferred_errors.argtypes = [C.c_int]
    L.cah_linked_views.argtypes = [vp, vp, vp, vp, i32, i64, vp, vp, vp]
    L.cah_linked_match_batch_uniform.argtypes = [vp, vp, vp, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    L.cah_workspace_bytes.argtypes = [i64]
    L.cah_workspace_bytes.restype = C.c_size_t
    L.cah_plan_workspace_bytes.argtypes = [vp, i64]
    L.cah_plan_workspace_bytes.restype = C.c_size_t
    L.cah_validate_ascii_batch.argtypes = [vp, vp, vp, i64, vp, vp]
    L.cah_reverse_reads_batch.argtypes = [vp, vp, vp, i64, vp, vp, vp]
    L.cah_revcomp_reads_batch.argtypes = [vp, vp, vp, i64, vp, vp, C.c_int32, vp, vp]
    L.cah_locate_batch_host.argtypes = [vp, i32, vp, vp, i64, vp, vp]
    L.cah_kmers_present_batch_host.argtypes = [vp, i32, vp, vp, i64, vp]
    L.cah_match_batch_host.argtypes = [vp, vp, vp, i64, vp, vp, vp]
    L.cah_locate_debug_host.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    L.cah_match_one_host.argtypes = [vp, C.c_char_p, i64, vp, vp]
    L.cah_locate_one_host.argtypes = [vp, C.c_char_p, i64, vp, vp]
    L.cah_profile_enable.argtypes = [C.c_int]
    L.cah_profile_reset.argtypes = []
    L.cah_profile_read.argtypes = [C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(i64)]
    L.cah_synth_reads.argtypes = [C.c_uint64, i64, i64, i32, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_char_p, C.POINTER(i32), i32, vp, vp, vp]
    L.cah_fastq_scan.argtypes = [vp, i64, C.c_int, i64, vp, C.POINTER(i64), C.POINTER(i64)]
    L.cah_pack_sequences.argtypes = [vp, vp, i64, vp, vp]
    L.cah_fastq_write_trimmed.argtypes = [vp, vp, i64, vp, vp, vp, vp, i64, C.POINTER(i64)]
    L.cah_fasta_scan.argtypes = [vp, i64, C.c_int, i64, vp, C.POINTER(i64), C.POINTER(i64)]
    L.cah_records_write.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, C.c_int, vp, i64, C.POINTER(i64)]
    L.cah_info_write.argtypes = [vp, vp, i64, vp, vp, vp, i64, vp, vp, i64, vp, i64, C.POINTER(i64)]
    L.cah_info_write_rc.argtypes = [vp, vp, i64, vp, vp, vp, i64, vp, vp, i64, vp, vp, vp, vp, i64, C.POINTER(i64)]
    L.cah_chunk_revcomp.argtypes = [vp, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, C.POINTER(i64)]
    L.cah_chunk_select.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, i64, vp, i64, vp, C.POINTER(i64)]
    L.cah_fastq_span.argtypes = [vp, i64, C.c_int, i64, C.POINTER(i64), C.POINTER(i64)]
    L.cah_record_boundary.argtypes = [vp, i64, C.c_int, C.POINTER(i64)]
    L.cah_fastq_device_scratch_bytes.argtypes = [i64, i64]
    L.cah_fastq_device_scratch_bytes.restype = C.c_size_t
    L.cah_fastq_count_lines_device.argtypes = [vp, i64, vp, C.c_size_t, vp, vp]
    L.cah_fastq_index_device.argtypes = [vp, i64, i64, i64, vp, C.c_size_t, vp, vp, vp, vp, vp]
    L.cah_fastq_format_device.argtypes = [vp, vp, i64, vp, vp, vp, vp, C.c_size_t, i64, vp, i64, vp, vp]
    if hasattr(L, "cah_mark_reads_device") or not any_abi:
        L.cah_mark_reads_device.argtypes = [vp, vp, i64, vp, vp, vp, vp, C.c_int, vp]
    if hasattr(L, "cah_revcomp_in_place_device") or not any_abi:
        L.cah_revcomp_in_place_device.argtypes = [vp, vp, i64, vp, vp, vp, vp]
        L.cah_fastq_format_suffix_device.argtypes = [vp, vp, i64, vp, vp, vp, vp, C.c_char_p, C.c_int32, vp, C.c_size_t, i64, vp,
                                                     i64, vp, vp]
        L.cah_info_format_device.argtypes = [vp, vp, i64, vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp, C.c_int32, vp, C.c_char_p,
                                             C.c_int32, vp, C.c_size_t, i64, vp, i64, vp, vp]
    L.cah_trim_decide_device.argtypes = [vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    L.cah_trim_decide_window_device.argtypes = [vp, vp, vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    L.cah_trim_decide_action_device.argtypes = [vp, vp, vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    L.cah_trim_filter_device.argtypes = [vp, vp, vp, vp, i64, i32, i32, C.c_double, i32, i32, vp, vp, vp]
    L.cah_index_create.argtypes = [C.POINTER(IndexAdapterC), i32, i32, C.POINTER(vp)]
    L.cah_index_destroy.argtypes = [vp]
    L.cah_index_destroy.restype = None
    L.cah_index_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.cah_index_get.argtypes = [vp, C.c_char_p, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.cah_index_lookup_batch.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, vp]
    L.cah_index_lookup_batch_host.argtypes = [vp, vp, vp, i64, vp, vp, vp]
    L.cah_quality_trim_batch.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, vp]
    L.cah_nextseq_trim_batch.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp, vp]
    L.cah_nextseq_trim_batch_q.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, vp, vp]
    L.cah_poly_a_trim_batch.argtypes = [vp, vp, vp, i64, i32, vp, vp]
    L.cah_expected_errors_batch.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp]
    for name in EXPORTED_SYMBOLS:
        if not (any_abi and not hasattr(L, name)):                 # (an older round's build lacks the newer entry points)
            getattr(L, name)
    _lib = L
    return L


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    lib().cah_last_error(buf, len(buf))
    return buf.value.decode("utf-8", "replace")


class HipInternalError(RuntimeError):
    """CAH_EINTERNAL: the library caught itself breaking one of its own invariants; the call's results are void"""


def build_id() -> str:
    """sha256 over the sources the LOADED library was built from (cah_build_id; cutadapt_amd.build.source_hash() is the
    same hash over the sources on disk)"""
    L = lib()
    return L.cah_build_id().decode() if hasattr(L, "cah_build_id") else ""


def last_multi_path() -> str:
    """'sequential', 'fused' or 'stream': the form this thread's last match_batch call actually took ('' before the first)"""
    v = lib().cah_last_multi_path()
    return ("sequential", "fused", "stream")[v] if 0 <= v <= 2 else ""


def check(rc: int) -> None:
    """Map a C status to the exception the reference would raise."""
    if rc == CAH_OK:
        return
    msg = last_error()
    if rc == CAH_EINVAL:
        raise ValueError(msg)
    if rc == CAH_ETYPE:
        raise TypeError(msg)
    if rc == CAH_EUNSUPPORTED:
        raise UnsupportedByHipPath(msg)
    if rc == CAH_ENOMEM:
        raise MemoryError(msg)
    if rc == CAH_EINTERNAL:
        raise HipInternalError(msg)
    raise RuntimeError(msg)


def raise_invalid_reads(max_len: int):
    """Status 2 (CAH_INVALID) was met: a byte >= 0x80 -- the reference raises this very ValueError (_align.pyx:44-45)
    -- or, a limit of this build and said as such, a read longer than CAH_MAX_READ_LEN (max_len: the batch's longest)."""
    if max_len > MAX_READ_LEN:
        raise UnsupportedByHipPath(f"reads longer than {MAX_READ_LEN} characters are not supported by this build")
    raise ValueError("String must contain only ASCII characters")


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().cah_device_count(C.byref(n))
    return n.value if rc == CAH_OK else 0


def device_info(device: int = 0) -> dict:
    name = C.create_string_buffer(256)
    arch = C.create_string_buffer(256)
    cus = C.c_int(0)
    mem = C.c_int64(0)
    check(lib().cah_device_info(device, name, 256, arch, 256, C.byref(cus), C.byref(mem)))
    return {"name": name.value.decode(), "arch": arch.value.decode(),
            "compute_units": cus.value, "hbm_bytes": mem.value}


def _ascii(s: str, what: str = "String") -> bytes:
    if not isinstance(s, str):
        raise TypeError(f"{what} must be a str, not {type(s).__name__}")
    try:
        return s.encode("ascii")
    except UnicodeEncodeError:
        raise ValueError("String must contain only ASCII characters")


class MatcherSpec:
    """Python-side description of one matcher (one cah_adapter_desc)."""

    def __init__(self, sequence: str = "", max_error_rate: float = 0.0, flags: int = 15,
                 wildcard_ref: bool = False, wildcard_query: bool = False, indel_cost: int = 1,
                 min_overlap: int = 1, kind: int = KIND_ALIGNER,
                 kmer_sets: Optional[Sequence[Tuple[int, Optional[int], Sequence[str]]]] = None,
                 kmer_ref_wildcards: bool = False, kmer_query_wildcards: bool = False):
        self.sequence = sequence
        self.max_error_rate = float(max_error_rate)
        self.flags = int(flags)
        self.wildcard_ref = bool(wildcard_ref)
        self.wildcard_query = bool(wildcard_query)
        self.indel_cost = int(indel_cost)
        self.min_overlap = int(min_overlap)
        self.kind = int(kind)
        self.kmer_sets = None if kmer_sets is None else [
            (int(start), stop, list(kmers)) for start, stop, kmers in kmer_sets]
        self.kmer_ref_wildcards = bool(kmer_ref_wildcards)
        self.kmer_query_wildcards = bool(kmer_query_wildcards)


class Plan:
    """An immutable device plan (cah_plan): tables of one or more matchers in HBM."""

    def __init__(self, specs: Sequence[MatcherSpec]):
        self._h = None
        self.specs = list(specs)
        n = len(self.specs)
        descs = (AdapterDescC * n)()
        keep = []   # keep ctypes buffers alive during the call
        for i, sp in enumerate(self.specs):
            d = descs[i]
            seq = _ascii(sp.sequence)
            keep.append(seq)
            d.sequence = seq
            d.length = len(seq)
            d.max_error_rate = sp.max_error_rate
            d.flags = sp.flags
            d.wildcard_ref = int(sp.wildcard_ref)
            d.wildcard_query = int(sp.wildcard_query)
            d.indel_cost = sp.indel_cost
            d.min_overlap = sp.min_overlap
            d.kind = sp.kind
            d.kmer_ref_wildcards = int(sp.kmer_ref_wildcards)
            d.kmer_query_wildcards = int(sp.kmer_query_wildcards)
            if sp.kmer_sets is None:
                d.n_kmer_sets = -1
                d.kmer_sets = None
            else:
                sets = (KmerSetC * max(len(sp.kmer_sets), 1))()
                for j, (start, stop, kmers) in enumerate(sp.kmer_sets):
                    enc = []
                    for k in kmers:
                        if type(k) is not str:
                            raise TypeError(f"Kmer should be a string not {type(k)}")
                        try:
                            enc.append(k.encode("ascii"))
                        except UnicodeEncodeError:
                            raise ValueError("Only ASCII strings are supported")
                    arr = (C.c_char_p * max(len(enc), 1))(*enc)
                    keep.extend([enc, arr])
                    sets[j].start = start
                    sets[j].stop = 0 if stop is None else int(stop)
                    sets[j].kmers = arr
                    sets[j].n_kmers = len(enc)
                keep.append(sets)
                d.kmer_sets = sets
                d.n_kmer_sets = len(sp.kmer_sets)
        h = C.c_void_p()
        check(lib().cah_plan_create(descs, n, C.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.cah_plan_destroy(h)

    @property
    def handle(self):
        return self._h

    @property
    def n_adapters(self) -> int:
        return int(lib().cah_plan_n_adapters(self._h))

    def effective_length(self, adapter: int = 0) -> int:
        out = C.c_int32(0)
        check(lib().cah_plan_effective_length(self._h, adapter, C.byref(out)))
        return out.value

    def prefilter_kind(self, adapter: int = 0) -> str:
        """'none', 'general' or 'lean': the prefilter kernel that serves this adapter"""
        out = C.c_int32(0)
        check(lib().cah_plan_prefilter_kind(self._h, adapter, C.byref(out)))
        return ("none", "general", "lean")[out.value]

    def multi_kind(self, read_len: int) -> str:
        """'sequential', 'fused' or 'stream': how equally long reads of this length are matched against all adapters"""
        out = C.c_int32(0)
        check(lib().cah_plan_multi_kind(self._h, read_len, C.byref(out)))
        return ("sequential", "fused", "stream")[out.value]

    def n_kmer_entries(self, adapter: int = 0) -> int:
        out = C.c_int32(0)
        check(lib().cah_plan_n_kmer_entries(self._h, adapter, C.byref(out)))
        return out.value


class _OneReadBuffers(threading.local):
    """per-thread output buffers of the one-read calls"""

    def __init__(self):
        self.out6 = (C.c_int32 * 6)()
        self.status = (C.c_uint8 * 1)()
        self.p6 = C.addressof(self.out6)
        self.ps = C.addressof(self.status)


_one = _OneReadBuffers()


def one_read(fn, plan_handle, sequence: str):
    """fn = lib().cah_match_one_host / cah_locate_one_host: the tuple of one read or None"""
    if type(sequence) is not str:
        raise TypeError(f"sequence must be a str, not {type(sequence).__name__}")
    try:
        q = sequence.encode("ascii")
    except UnicodeEncodeError:
        raise ValueError("String must contain only ASCII characters")
    b = _one
    rc = fn(plan_handle, q, len(q), b.p6, b.ps)
    if rc != CAH_OK:
        check(rc)
    st = b.status[0]
    if st == MATCH:
        o = b.out6
        return (o[0], o[1], o[2], o[3], o[4], o[5])
    if st == INVALID:
        raise ValueError("String must contain only ASCII characters")
    return None


def locate_debug(spec: MatcherSpec, query: str):
    """Aligner.locate(query) with the DP matrices of Aligner.enable_debug() (reference _align.pyx:279-296):
    returns (tuple or None, cost rows, score rows); rows[i][j] is None where the banded algorithm computed nothing."""
    import numpy as np
    seq = _ascii(spec.sequence)
    q = _ascii(query)
    d = AdapterDescC()
    d.sequence = seq
    d.length = len(seq)
    d.max_error_rate = spec.max_error_rate
    d.flags = spec.flags
    d.wildcard_ref = int(spec.wildcard_ref)
    d.wildcard_query = int(spec.wildcard_query)
    d.indel_cost = spec.indel_cost
    d.min_overlap = spec.min_overlap
    d.kind = spec.kind
    d.n_kmer_sets = -1
    d.kmer_sets = None
    m, n = len(seq), len(q)
    none = np.iinfo(np.int32).min
    cost = np.full((m + 1, n + 1), none, dtype=np.int32)
    score = np.full((m + 1, n + 1), none, dtype=np.int32)
    out6 = np.zeros(6, dtype=np.int32)
    status = np.zeros(1, dtype=np.uint8)
    qbuf = np.frombuffer(q, dtype=np.uint8) if n else np.zeros(1, dtype=np.uint8)
    check(lib().cah_locate_debug_host(C.byref(d), qbuf.ctypes.data, n, out6.ctypes.data, status.ctypes.data,
                                      cost.ctypes.data, score.ctypes.data))
    if status[0] == INVALID:
        raise ValueError("String must contain only ASCII characters")

    def rows(a):
        return [[None if v == none else int(v) for v in row] for row in a]
    return (tuple(int(v) for v in out6) if status[0] == MATCH else None), rows(cost), rows(score)


class Index:
    """An immutable adapter index (cah_index): the AdapterIndex dictionary as a hash table."""

    def __init__(self, adapters: Sequence[tuple], prefix: bool):
        """adapters: (sequence, max_error_rate, indels[, kmer_sets]) per adapter; kmer_sets is the
        adapter's positions_and_kmers list or None (no prefilter)"""
        self._h = None
        n = len(adapters)
        descs = (IndexAdapterC * max(n, 1))()
        keep = []
        for i, spec in enumerate(adapters):
            seq, rate, indels = spec[:3]
            kmer_sets = spec[3] if len(spec) > 3 else None
            b = _ascii(seq)
            keep.append(b)
            descs[i].sequence = b
            descs[i].length = len(b)
            descs[i].max_error_rate = float(rate)
            descs[i].indels = int(bool(indels))
            if kmer_sets is None:
                descs[i].kmer_sets = None
                descs[i].n_kmer_sets = -1
            else:
                sets = (KmerSetC * max(len(kmer_sets), 1))()
                for j, (start, stop, kmers) in enumerate(kmer_sets):
                    enc = [_ascii(k, "Kmer") for k in kmers]
                    arr = (C.c_char_p * max(len(enc), 1))(*enc)
                    keep.extend([enc, arr])
                    sets[j].start = start
                    sets[j].stop = 0 if stop is None else int(stop)
                    sets[j].kmers = arr
                    sets[j].n_kmers = len(enc)
                keep.append(sets)
                descs[i].kmer_sets = sets
                descs[i].n_kmer_sets = len(kmer_sets)
        h = C.c_void_p()
        check(lib().cah_index_create(descs, n, int(bool(prefix)), C.byref(h)))
        self._h = h
        n_strings, n_amb, n_len = C.c_int64(0), C.c_int32(0), C.c_int32(0)
        lengths = (C.c_int32 * 64)()
        check(lib().cah_index_info(h, C.byref(n_strings), C.byref(n_amb), lengths, C.byref(n_len)))
        self.n_strings = n_strings.value
        self.n_ambiguous = n_amb.value
        self.lengths = [int(lengths[i]) for i in range(n_len.value)]

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.cah_index_destroy(h)

    @property
    def handle(self):
        return self._h

    def get(self, s: str):
        """-> (adapter index, errors, matches) or None: the dictionary entry of one string"""
        try:
            b = s.encode("ascii")
        except UnicodeEncodeError:
            return None
        found, ad, e, m = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        check(lib().cah_index_get(self._h, b, len(b), C.byref(found), C.byref(ad), C.byref(e), C.byref(m)))
        return (ad.value, e.value, m.value) if found.value else None
