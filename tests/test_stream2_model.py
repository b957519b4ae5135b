"""Host-model fuzz of k_filter_stream2's word machinery (cutadapt_amd/csrc/stream2.h) against the oracle.

No GPU: the header the kernel is built from is compiled with g++ (tests/host_model/stream_model.cpp) and run on the
product's own CahLeanFilter tables (cah_plan_debug_lean).  For every read:
  * present must equal the oracle's KmerFinder.kmers_present (reference _kmer_finder.pyx:170-257),
  * the first-hit group (the survivor queue's key) must equal a brute-force restatement of "the first 4-character
    group in which a k-mer that lies in its window ends",
  * neither may depend on when a T-word wakes up (every_word = 1 advances every word over the whole read).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from cutadapt_amd import _lib
from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_model", "stream_model.cpp")
SO = os.path.join(HERE, "host_model", "libstream_model.so")
TRUSEQ = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
SEED0 = int(os.environ.get("CAH_TEST_SEED_OFFSET", "0"))


@pytest.fixture(scope="module")
def model():
    deps = [SRC, os.path.join(HERE, "..", "cutadapt_amd", "csrc", "stream2.h"),
            os.path.join(HERE, "..", "cutadapt_amd", "csrc", "cah_device.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", SRC, "-o", SO], check=True)
    L = C.CDLL(SO)
    vp = C.c_void_p
    L.sm_lean_size.restype = C.c_size_t
    L.sm_filter_batch.argtypes = [vp, vp, C.c_int64, C.c_int, vp, vp, C.c_int]
    L.sm_tw_ok.argtypes = [vp]
    L.sm_n_words.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.sm_check_unit_division.argtypes = [C.c_int]
    return L


def lean_blob(sets, ref_wc=False, query_wc=False):
    spec = _lib.MatcherSpec(kind=_lib.KIND_KMER_ONLY, kmer_sets=sets, kmer_ref_wildcards=ref_wc,
                            kmer_query_wildcards=query_wc)
    plan = _lib.Plan([spec])
    L = _lib.lib()
    need = C.c_size_t(0)
    _lib.check(L.cah_plan_debug_lean(plan.handle, 0, None, 0, C.byref(need)))
    buf = (C.c_uint8 * need.value)()
    _lib.check(L.cah_plan_debug_lean(plan.handle, 0, buf, need.value, C.byref(need)))
    return buf, need.value




def brute_first_group(finder_for, sets, read, n):
    """first 4-character group in which a k-mer of `sets` that lies in its window ends (-1: none); character matching
    is delegated to single-k-mer oracle finders on the window's substring"""
    best = -1
    for start, stop, kmers in sets:
        lo = max(0, n + start) if start < 0 else start
        hi = n if stop is None else stop
        for k in kmers:
            q = len(k)
            f = finder_for(k)
            for p in range(lo, hi - q + 1):
                if f.kmers_present(read[p:p + q]):
                    e = p + q - 1
                    if best < 0 or e // 4 < best:
                        best = e // 4
                    break
    return best


def random_reads(rng, n_reads, n, adapter, alphabet="ACGT", p_n=0.01):
    reads = []
    m = len(adapter)
    for _ in range(n_reads):
        s = rng.choice(list(alphabet), size=n)
        u = rng.random()
        if u < 0.7 and n > 0:
            # a piece of the adapter's head somewhere, mostly near the read's end
            ln = int(rng.integers(1, m + 1))
            piece = list(adapter[:ln]) if rng.random() < 0.7 else list(adapter[int(rng.integers(0, m)):][:ln])
            if rng.random() < 0.6:
                at = n - len(piece) + int(rng.integers(-3, 4))
            else:
                at = int(rng.integers(0, n))
            for i, ch in enumerate(piece):
                if 0 <= at + i < n:
                    s[at + i] = ch
            # a sprinkle of edits
            for _e in range(int(rng.integers(0, 3))):
                s[int(rng.integers(0, n))] = rng.choice(list("ACGTN"))
        mask = rng.random(n) < p_n
        s[mask] = "N"
        if rng.random() < 0.1 and n > 0:
            s[int(rng.integers(0, n))] = rng.choice(list("acgtnRYX."))
        reads.append("".join(s))
    return reads


def run(model, blob, reads, n, every_word=0):
    nr = len(reads)
    seqs = np.frombuffer("".join(reads).encode("ascii"), dtype=np.uint8).copy() if nr * n else np.zeros(1, np.uint8)
    present = np.zeros(nr, dtype=np.uint8)
    hit = np.zeros(nr, dtype=np.int32)
    rc = model.sm_filter_batch(blob, seqs.ctypes.data, nr, n, present.ctypes.data, hit.ctypes.data, every_word)
    assert rc == 0, rc
    return present, hit


def check_plan(model, sets, reads, n, ref_wc=False, query_wc=False, brute=True):
    blob, size = lean_blob(sets, ref_wc, query_wc)
    assert size == model.sm_lean_size()
    if not model.sm_tw_ok(blob):
        return False
    of = orc.KmerFinder(sets, ref_wc, query_wc)
    want = np.array([of.kmers_present(r) for r in reads], dtype=np.uint8)
    present, hit = run(model, blob, reads, n)
    assert np.array_equal(present, want), (sets, [r for r, a, b in zip(reads, present, want) if a != b][:3])
    present2, hit2 = run(model, blob, reads, n, every_word=1)
    assert np.array_equal(present2, want)
    assert np.array_equal(hit2, hit)
    if brute:
        cache = {}

        def finder_for(k):
            if k not in cache:
                cache[k] = orc.KmerFinder([(0, None, [k])], ref_wc, query_wc)
            return cache[k]
        for r, p, h in zip(reads, present, hit):
            g = brute_first_group(finder_for, sets, r, n)
            assert (h // 4 if p else -1) == g, (sets, r, h, g)
    return True


def test_unit_division(model):
    for H in range(1, 6):
        assert model.sm_check_unit_division(H) == -1


def test_truseq_every_length(model):
    rng = np.random.default_rng(101 + SEED0)
    sets = create_positions_and_kmers(TRUSEQ, 3, 0.1, back_adapter=True, front_adapter=False, internal=True)
    used = 0
    for n in list(range(1, 40)) + list(range(60, 162)):
        reads = random_reads(rng, 60, n, TRUSEQ)
        used += check_plan(model, sets, reads, n, brute=(n % 7 == 0 or n == 150))
    assert used > 100
    blob, _ = lean_blob(sets)
    nl, nt = C.c_int(0), C.c_int(0)
    model.sm_n_words(blob, C.byref(nl), C.byref(nt))
    assert (nl.value, nt.value) == (2, 4)          # the class the headline configuration runs in


def test_random_back_adapters(model):
    rng = np.random.default_rng(202 + SEED0)
    used = tried = 0
    for _ in range(160):
        m = int(rng.integers(6, 41))
        alphabet = "ACGT" if rng.random() < 0.8 else "AC"
        adapter = "".join(rng.choice(list(alphabet), size=m))
        rate = float(rng.choice([0.0, 0.05, 0.1, 0.1, 0.15, 0.2]))
        min_overlap = int(rng.integers(1, 8))
        sets = create_positions_and_kmers(adapter, min_overlap, rate, back_adapter=True, front_adapter=False, internal=True)
        n = int(rng.choice([150, 150, 100, 76, 50, 36, 151, 160, 33, 17, 75, 81]))
        reads = random_reads(rng, 80, n, adapter)
        tried += 1
        used += check_plan(model, sets, reads, n, brute=(tried % 4 == 0))
    assert used >= tried // 2, (used, tried)


def test_wildcards(model):
    rng = np.random.default_rng(303 + SEED0)
    used = 0
    for i in range(40):
        m = int(rng.integers(10, 34))
        adapter = "".join(rng.choice(list("ACGTACGTACGTNRY"), size=m))
        sets = create_positions_and_kmers(adapter, 3, 0.1, back_adapter=True, front_adapter=False, internal=True)
        ref_wc, query_wc = bool(i & 1), bool(i & 2)
        n = int(rng.choice([150, 100, 51]))
        reads = random_reads(rng, 60, n, adapter.replace("N", "A").replace("R", "G").replace("Y", "C"), p_n=0.05)
        used += check_plan(model, sets, reads, n, ref_wc, query_wc, brute=(i % 5 == 0))
    assert used >= 10


def brute_first_end(finder_for, sets, read, n):
    """end position of the first-ending k-mer of `sets` that lies in its window (-1: none)"""
    best = -1
    for start, stop, kmers in sets:
        lo = max(0, n + start) if start < 0 else start
        hi = n if stop is None else stop
        for k in kmers:
            q = len(k)
            f = finder_for(k)
            for p in range(lo, hi - q + 1):
                if f.kmers_present(read[p:p + q]):
                    if best < 0 or p + q - 1 < best:
                        best = p + q - 1
                    break
    return best


def test_end_aligned_views(model):
    """What k_filter_stream2's RV form rests on (views inside the reads of a uniform batch, cah_match_batch_views): a view of
    len characters, streamed as a read of n characters that ENDS where the view ends and holds NUL in its first n - len
    positions, has the view's own kmers_present -- tail search sets count from the end (reference _kmer_finder.pyx:186-204),
    also when the view is shorter than a set's window -- and its first-hit group is that of the view's first hit moved by
    n - len (the queue key the kernel derives from it is a lower bound of the hit within the view)."""
    rng = np.random.default_rng(404 + SEED0)
    used = tried = 0
    for i in range(60):
        if i == 0:
            adapter, rate, min_overlap = TRUSEQ, 0.1, 3
        else:
            m = int(rng.integers(6, 41))
            adapter = "".join(rng.choice(list("ACGT" if rng.random() < 0.7 else "ACGTNRY"), size=m))
            rate = float(rng.choice([0.0, 0.1, 0.1, 0.2]))
            min_overlap = int(rng.integers(1, 8))
        ref_wc, query_wc = bool(i & 1) and i > 0, bool(i & 2)
        sets = create_positions_and_kmers(adapter, min_overlap, rate, back_adapter=True, front_adapter=False, internal=True)
        blob, size = lean_blob(sets, ref_wc, query_wc)
        tried += 1
        if not model.sm_tw_ok(blob):
            continue
        used += 1
        n = int(rng.choice([150, 150, 100, 76, 40, 151, 160, 33, 17]))
        plain = adapter.replace("N", "A").replace("R", "G").replace("Y", "C")
        lens = [int(x) for x in rng.choice([0, 1, 2, 3, 5, 8, 15, 16, 17, 30, n - 1, n] + list(range(n + 1)), size=70)]
        lens = [min(max(x, 0), n) for x in lens]
        views = [random_reads(rng, 1, ln, plain)[0] if ln else "" for ln in lens]
        padded = ["\0" * (n - len(v)) + v for v in views]
        of = orc.KmerFinder(sets, ref_wc, query_wc)
        want = np.array([of.kmers_present(v) for v in views], dtype=np.uint8)
        present, hit = run(model, blob, padded, n)
        assert np.array_equal(present, want), (sets, n, [(v, a, b) for v, a, b in zip(views, present, want) if a != b][:3])
        if i % 3 == 0:
            cache = {}

            def finder_for(k):
                if k not in cache:
                    cache[k] = orc.KmerFinder([(0, None, [k])], ref_wc, query_wc)
                return cache[k]
            for v, p_, h in zip(views, present, hit):
                e = brute_first_end(finder_for, sets, v, len(v))
                assert (h // 4 if p_ else -1) == ((e + n - len(v)) // 4 if e >= 0 else -1), (sets, v, n, h, e)
    assert used >= tried // 2, (used, tried)


def test_invalid_bytes(model):
    sets = create_positions_and_kmers(TRUSEQ, 3, 0.1, back_adapter=True, front_adapter=False, internal=True)
    blob, _ = lean_blob(sets)
    n = 150
    read = bytearray(b"A" * n)
    read[77] = 0xC3
    seqs = np.frombuffer(bytes(read), dtype=np.uint8).copy()
    present = np.zeros(1, dtype=np.uint8)
    hit = np.zeros(1, dtype=np.int32)
    assert model.sm_filter_batch(blob, seqs.ctypes.data, 1, n, present.ctypes.data, hit.ctypes.data, 0) == 0
    assert present[0] == 2
