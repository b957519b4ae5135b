"""Host-model fuzz of the cost scan's classification (cutadapt_amd/csrc/back_scan.h) against the oracle.

No GPU: the header k_back_scan is built from is compiled with g++ (tests/host_model/back_model.cpp,
together with a plain restatement of the windowed cell DP) and every read's result --
NONE / EXACT_FULL / EXACT_TAIL shortcut or windowed DP -- must equal the oracle's Aligner.locate on the
whole read.  The matcher tables come from the product's own plan builder (cah_plan_debug_matcher)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from cutadapt_amd import _lib
from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_model", "back_model.cpp")
SO = os.path.join(HERE, "host_model", "libback_model.so")
TRUSEQ = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"


@pytest.fixture(scope="module")
def model():
    deps = [SRC, os.path.join(HERE, "..", "cutadapt_amd", "csrc", "back_scan.h"),
            os.path.join(HERE, "..", "cutadapt_amd", "csrc", "cah_device.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", SRC, "-o", SO], check=True)
    L = C.CDLL(SO)
    vp, i64 = C.c_void_p, C.c_int64
    L.bm_locate_batch.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp, vp, C.c_int32, vp, C.c_int32]
    L.bm_skip_columns.argtypes = [C.c_char_p, C.c_int, C.c_int, vp, vp, i64, vp]
    L.bm_refined_columns.argtypes = [C.c_char_p, C.c_int, C.c_int, vp, vp, i64, vp]
    L.bm_keys.argtypes = [C.c_char_p, C.c_int, C.c_int, vp, vp, i64, vp]
    L.bm_matcher_size.restype = C.c_size_t
    return L


def matcher_blob(adapter, rate, min_overlap, wildcard_ref=False, wildcard_query=False):
    spec = _lib.MatcherSpec(adapter, rate, 14, wildcard_ref, wildcard_query, 1, min_overlap)
    plan = _lib.Plan([spec])
    L = _lib.lib()
    need = C.c_size_t(0)
    _lib.check(L.cah_plan_debug_matcher(plan.handle, 0, None, 0, C.byref(need)))
    buf = (C.c_uint8 * need.value)()
    _lib.check(L.cah_plan_debug_matcher(plan.handle, 0, buf, need.value, C.byref(need)))
    return buf, need.value


STATS = {"stopped": 0, "reads": 0}


def run_model(L, blob, seqs, offsets, j0s=None, stop_every=16, form=-1):
    n = len(offsets) - 1
    out6 = np.zeros((n, 6), dtype=np.int32)
    status = np.zeros(n, dtype=np.uint8)
    cls = np.zeros(n, dtype=np.uint8)
    rc = L.bm_locate_batch(blob, seqs.ctypes.data, offsets.ctypes.data, n,
                           None if j0s is None else j0s.ctypes.data, out6.ctypes.data, status.ctypes.data,
                           cls.ctypes.data, None, stop_every, None, form)
    STATS["stopped"] += int(((cls & 8) != 0).sum())
    STATS["reads"] += len(cls)
    return rc, out6, status, cls & 7


def compare(L, adapter, rate, min_overlap, seqs, offsets, wr=False, wq=False, skip=False, label=""):
    blob, size = matcher_blob(adapter, rate, min_overlap, wr, wq)
    assert size == L.bm_matcher_size()
    oa = orc.Aligner(adapter, rate, 14, wr, wq, 1, min_overlap)
    want6, want_st = oa.locate_batch(seqs, offsets)
    j0s = None
    if skip:
        n = len(offsets) - 1
        j0s = np.zeros(n, dtype=np.int32)
        fn = L.bm_refined_columns if skip == "refined" else L.bm_skip_columns
        fn(adapter.encode(), len(adapter), int(rate * len(adapter)), seqs.ctypes.data, offsets.ctypes.data, n, j0s.ctypes.data)
    # the early stop looked for once per 16-column chunk and in its middle (what the kernel does), after every column (the tightest
    # use of the rule) and never (the scan always reaches the read end)
    # ... each in the form the launcher picks for the adapter (32-bit words, + explicit rows for 33 / 34 characters)
    # and in the 64-bit form
    for stop_every, form in ((16, -1), (8, -1), (1, -1), (0, -1), (16, 0), (0, 0)):
        rc, out6, status, cls = run_model(L, blob, seqs, offsets, j0s, stop_every, form)
        if rc == 1:
            return None                                  # matcher not scan-eligible: nothing to check
        bad = np.nonzero((status != want_st) | (out6 != want6).any(axis=1))[0]
        if len(bad):
            r = int(bad[0])
            read = bytes(seqs[offsets[r]:offsets[r + 1]]).decode("latin-1")
            raise AssertionError(f"{label} (stop_every {stop_every}, form {form}): {len(bad)} of {len(want_st)} reads differ; first: read {r} {read!r} adapter "
                                 f"{adapter} rate {rate} O {min_overlap} wr {wr} wq {wq} j0 {None if j0s is None else j0s[r]} "
                                 f"class {cls[r]} model {status[r]} {out6[r].tolist()} oracle {want_st[r]} {want6[r].tolist()}")
        if stop_every == 16 and form == -1:
            counts = np.bincount(cls, minlength=6)
    return counts


def random_reads(rng, adapter, n_reads, max_len, p_edit, p_n, alphabet="ACGT"):
    """ragged reads with 0-2 (edited, possibly truncated) adapter copies"""
    reads = []
    bases = np.array(list(alphabet))
    for _ in range(n_reads):
        n = int(rng.integers(0, max_len + 1))
        s = list(rng.choice(bases, size=n))
        for _copy in range(int(rng.integers(0, 3))):
            ad = []
            for c in adapter:
                u = rng.random()
                if u < p_edit / 2:
                    ad.append(str(rng.choice(bases)))
                elif u < p_edit * 0.75:
                    ad.append(str(rng.choice(bases))); ad.append(c)
                elif u < p_edit:
                    pass
                else:
                    ad.append(c)
            pos = int(rng.integers(0, n + 1))
            s[pos:pos + len(ad)] = ad
            s = s[:n]
        if p_n:
            for i in range(len(s)):
                if rng.random() < p_n:
                    s[i] = "N"
        reads.append("".join(s))
    return orc.pack_reads(reads)


def test_truseq_bulk(model):
    """the benchmark shape: TruSeq, e=0.1, O=3, 150 bp synthetic reads (with and without window skipping)"""
    for seed, gen in ((2, dict(p_adapter=0.25, p_edit=0.02, p_n=0.005)), (5, dict(p_adapter=0.9, p_edit=0.08, p_n=0.02)),
                      (6, dict(p_adapter=0.0, p_edit=0.0, p_n=0.0))):
        seqs, offsets = orc.synth_reads(seed, 0, 150_000, 150, [TRUSEQ], **gen)
        for skip in (False, True):
            counts = compare(model, TRUSEQ, 0.1, 3, seqs, offsets, skip=skip, label=f"truseq seed {seed} skip {skip}")
            assert counts is not None
    # most reads must be finished without the cell DP
    seqs, offsets = orc.synth_reads(2, 0, 100_000, 150, [TRUSEQ])
    counts = compare(model, TRUSEQ, 0.1, 3, seqs, offsets)
    assert counts[3] < 0.15 * counts.sum(), counts


def test_random_adapters(model):
    rng = np.random.default_rng(11)
    done = 0
    for it in range(160):
        m = int(rng.integers(1, 65))
        adapter = "".join(rng.choice(list("ACGT"), size=m))
        rate = float(rng.choice([0.0, 0.05, 0.1, 0.15, 0.2, 0.3, 0.5]))
        min_overlap = int(rng.choice([1, 2, 3, 5, m, m + 2]))
        seqs, offsets = random_reads(rng, adapter, 700, int(rng.choice([12, 40, 90, 170])),
                                     float(rng.choice([0.0, 0.03, 0.1, 0.2])), float(rng.choice([0.0, 0.01, 0.1])))
        for skip in (False, True):
            if compare(model, adapter, rate, min_overlap, seqs, offsets, skip=skip, label=f"random {it} skip {skip}") is not None:
                done += 1
    assert done > 150


def test_wildcards_and_case(model):
    rng = np.random.default_rng(12)
    for it in range(80):
        m = int(rng.integers(3, 50))
        letters = list("ACGT") * 4 + list("NRYSWKMBDHV") + ["n", "a", "X"]
        adapter = "".join(rng.choice(letters, size=m))
        wr = bool(rng.integers(0, 2))
        wq = bool(rng.integers(0, 2))
        if wr and all(c in "Nn" for c in adapter):
            continue
        rate = float(rng.choice([0.05, 0.1, 0.2, 0.34]))
        seqs, offsets = random_reads(rng, adapter.upper() if not wr else adapter.upper().replace("X", "A"), 500,
                                     int(rng.choice([30, 80, 160])), float(rng.choice([0.0, 0.05, 0.15])),
                                     float(rng.choice([0.0, 0.05, 0.3])), alphabet="ACGTacgtNRY")
        compare(model, adapter, rate, int(rng.choice([1, 3, 8])), seqs, offsets, wr=wr, wq=wq, label=f"wild {it}")


def test_two_copies_and_repeats(model):
    """several candidate clusters per read: the shortcut must give way to the DP when its bound fails"""
    rng = np.random.default_rng(13)
    for it in range(60):
        unit = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 5))))
        m = int(rng.integers(6, 40))
        adapter = (unit * 40)[:m] if it % 2 else "".join(rng.choice(list("ACGT"), size=m))
        reads = []
        for _ in range(400):
            parts = []
            for _c in range(int(rng.integers(1, 4))):
                parts.append("".join(rng.choice(list("ACGT"), size=int(rng.integers(0, 30)))))
                ad = list(adapter)
                for _e in range(int(rng.integers(0, 4))):
                    if ad:
                        ad[int(rng.integers(0, len(ad)))] = str(rng.choice(list("ACGT")))
                cut = int(rng.integers(0, len(ad) + 1)) if rng.random() < 0.3 else len(ad)
                parts.append("".join(ad[:cut]))
            reads.append("".join(parts))
        seqs, offsets = orc.pack_reads(reads)
        for skip in (False, True):
            compare(model, adapter, float(rng.choice([0.1, 0.2, 0.3])), int(rng.choice([1, 3])), seqs, offsets, skip=skip,
                    label=f"copies {it}")


def test_substitution_class(model):
    """SUBS_FULL (the adapter with substitutions only): adapters of every kind -- random, low-complexity (where an
    insertion + deletion can cost as little as two substitutions), with a second, possibly better copy or a partial
    copy at the read end -- must either get exactly the reference's tuple or fall back to the DP."""
    rng = np.random.default_rng(21)
    subs_total = 0
    for it in range(120):
        m = int(rng.integers(4, 65))
        kind = it % 4
        if kind == 0:
            adapter = "".join(rng.choice(list("ACGT"), size=m))
        elif kind == 1:
            unit = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 4))))
            adapter = (unit * 70)[:m]
        elif kind == 2:
            adapter = "".join(rng.choice(list("AC"), size=m))
        else:
            adapter = "".join(rng.choice(list("ACGT"), size=m // 2)) * 2 + "A" * (m % 2)
        rate = float(rng.choice([0.05, 0.1, 0.15, 0.2, 0.3]))
        min_overlap = int(rng.choice([1, 3, 5, m]))
        reads = []
        for _ in range(600):
            n = int(rng.integers(m, 3 * m + 40))
            s = list(rng.choice(list("ACGT") if kind != 2 else list("ACCA"), size=n))
            for _copy in range(int(rng.integers(1, 3))):
                ad = list(adapter)
                for _e in range(int(rng.integers(0, 5))):
                    ad[int(rng.integers(0, m))] = str(rng.choice(list("ACGT")))
                if rng.random() < 0.15:
                    del ad[int(rng.integers(0, len(ad)))]
                if rng.random() < 0.15:
                    ad.insert(int(rng.integers(0, len(ad) + 1)), str(rng.choice(list("ACGT"))))
                pos = int(rng.integers(0, n + 1)) if rng.random() < 0.7 else n - int(rng.integers(1, m + 1))
                pos = max(pos, 0)
                s[pos:pos + len(ad)] = ad
                s = s[:n]
            if rng.random() < 0.1:
                s[int(rng.integers(0, len(s)))] = "N"
            reads.append("".join(s))
        seqs, offsets = orc.pack_reads(reads)
        for skip in (False, True):
            counts = compare(model, adapter, rate, min_overlap, seqs, offsets, skip=skip, label=f"subs {it} skip {skip}")
            if counts is not None:
                subs_total += int(counts[4])
    assert subs_total > 5000, subs_total


def test_early_stop_and_shadowed_tails(model):
    """The two rules of round 2, session 3 (back_scan.h): (1) the scan stops `gap` columns after the last acceptable
    candidate -- reads with a second copy of the adapter (better, worse, partial at the read end) at every distance
    around that gap; (2) EXACT_TAIL although longer rows of the last column are acceptable -- reads that end with a
    prefix of the adapter, behind text that makes longer rows cheap (shifted copies, low-complexity adapters)."""
    rng = np.random.default_rng(33)
    STATS["stopped"] = STATS["reads"] = 0
    tails = dp = total = 0
    for it in range(150):
        kind = it % 5
        m = int(rng.integers(6, 65)) if kind else 33
        if kind == 0:
            adapter = TRUSEQ
        elif kind == 1:
            adapter = "".join(rng.choice(list("ACGT"), size=m))
        elif kind == 2:
            unit = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 4))))
            adapter = (unit * 70)[:m]
        elif kind == 3:
            adapter = "".join(rng.choice(list("AC"), size=m))
        else:
            adapter = "".join(rng.choice(list("ACGT"), size=(m + 1) // 2)) * 2
            adapter = adapter[:m]
        rate = float(rng.choice([0.05, 0.1, 0.1, 0.15, 0.2, 0.3]))
        k = int(rate * m)
        gap = k + 1 + k + m // 2
        min_overlap = int(rng.choice([1, 3, 3, 5]))
        alpha = list("ACGT") if kind != 3 else list("ACCA")

        def edited(piece, n_edits):
            ad = list(piece)
            for _e in range(n_edits):
                if not ad:
                    break
                u, pos = rng.random(), int(rng.integers(0, len(ad)))
                if u < 0.6:
                    ad[pos] = str(rng.choice(alpha))
                elif u < 0.8:
                    del ad[pos]
                else:
                    ad.insert(pos, str(rng.choice(alpha)))
            return "".join(ad)

        reads = []
        for _ in range(500):
            u = rng.random()
            head = "".join(rng.choice(alpha, size=int(rng.integers(0, 40))))
            if u < 0.5:
                # full copy, a gap around the stop distance, a second (full or partial) copy, maybe a tail
                first = edited(adapter, int(rng.integers(0, k + 2)))
                g = max(0, gap - m + int(rng.integers(-12, 13))) if rng.random() < 0.7 else int(rng.integers(0, 80))
                second = edited(adapter, int(rng.integers(0, k + 2)))
                if rng.random() < 0.4:
                    second = second[:int(rng.integers(0, len(second) + 1))]
                tail = "".join(rng.choice(alpha, size=int(rng.integers(0, 50)))) if rng.random() < 0.6 else ""
                reads.append(head + first + "".join(rng.choice(alpha, size=g)) + second + tail)
            else:
                # the read ends with adapter[0:i]; in front of it a shifted / edited piece of the adapter
                i = int(rng.integers(1, m + 1))
                shift = int(rng.integers(0, 4))
                front = edited(adapter[shift:shift + int(rng.integers(0, m))], int(rng.integers(0, 3))) if rng.random() < 0.6 else ""
                body = "".join(rng.choice(alpha, size=int(rng.integers(0, 60))))
                end = adapter[:i] if rng.random() < 0.7 else edited(adapter[:i], 1)
                reads.append(head + body + front + end)
        seqs, offsets = orc.pack_reads(reads)
        for skip in (False, True):
            counts = compare(model, adapter, rate, min_overlap, seqs, offsets, skip=skip, label=f"stop/tail {it} skip {skip}")
            if counts is not None:
                tails += int(counts[2]); dp += int(counts[3]); total += int(counts.sum())
    assert STATS["stopped"] > 20000, STATS
    assert tails > 15000 and total > 100000, (tails, dp, total)


def test_word_forms(model):
    """Adapters around the 32-bit word: 31 / 32 characters (one 32-bit word), 33 / 34 (32-bit word + 1 / 2 explicit
    rows -- the TruSeq adapter's case), 35 (64-bit word): every form against the oracle and, inside compare(), against
    the 64-bit form of the same reads."""
    rng = np.random.default_rng(77)
    done = 0
    for it in range(120):
        m = int(rng.choice([1, 2, 31, 32, 33, 33, 34, 34, 35]))
        kind = it % 3
        if kind == 0:
            adapter = "".join(rng.choice(list("ACGT"), size=m))
        elif kind == 1:
            unit = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 4))))
            adapter = (unit * 40)[:m]
        else:
            adapter = "".join(rng.choice(list("AC"), size=m))
        rate = float(rng.choice([0.0, 0.05, 0.1, 0.15, 0.2, 0.3]))
        min_overlap = int(rng.choice([1, 3, 5, m]))
        seqs, offsets = random_reads(rng, adapter, 600, int(rng.choice([40, 90, 170])),
                                     float(rng.choice([0.0, 0.03, 0.1, 0.2])), float(rng.choice([0.0, 0.01])),
                                     alphabet="ACGT" if kind != 2 else "ACCA")
        for skip in (False, True):
            if compare(model, adapter, rate, min_overlap, seqs, offsets, skip=skip, label=f"forms {it} m {m} skip {skip}") is not None:
                done += 1
    assert done > 200


def test_one_indel_class(model):
    """INDEL1_FULL (the adapter with one insertion or one deletion and, possibly, substitutions; 32-bit forms): every position of the edit --
    inside runs of equal characters, at the adapter's ends --, adapters of every kind (random, low-complexity,
    two-letter: where a mismatch path, a deletion path and an insertion path can cost the same), a second copy
    or a partial copy behind it, further substitutions (the DP's case).  Either the reference's tuple or the DP."""
    rng = np.random.default_rng(55)
    indel1 = 0
    for it in range(160):
        m = int(rng.choice([4, 6, 9, 12, 17, 20, 25, 31, 32, 33, 33, 34]))
        kind = it % 4
        if kind == 0:
            adapter = "".join(rng.choice(list("ACGT"), size=m))
        elif kind == 1:
            unit = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 4))))
            adapter = (unit * 40)[:m]
        elif kind == 2:
            adapter = "".join(rng.choice(list("AC"), size=m))
        else:
            adapter = "".join(rng.choice(list("ACGT"), size=1)[0] * int(rng.integers(1, 4)) for _ in range(m))[:m]
        rate = float(rng.choice([0.1, 0.1, 0.15, 0.2, 0.3]))
        min_overlap = int(rng.choice([1, 3, 5]))
        alpha = list("ACGT") if kind != 2 else list("ACCA")
        reads = []
        for _ in range(600):
            ad = list(adapter)
            pos = int(rng.integers(0, m))
            if rng.random() < 0.5:
                del ad[pos]
            else:
                ad.insert(pos, str(rng.choice(alpha)))
            for _e in range(int(rng.choice([0, 0, 1, 1, 2]))):           # + substitutions: still one indel
                ad[int(rng.integers(0, len(ad)))] = str(rng.choice(alpha))
            if rng.random() < 0.1:                                       # a second indel: the DP's case
                del ad[int(rng.integers(0, len(ad)))]
            head = "".join(rng.choice(alpha, size=int(rng.integers(0, 60))))
            tail = "".join(rng.choice(alpha, size=int(rng.integers(0, 60)))) if rng.random() < 0.7 else ""
            if rng.random() < 0.15:
                second = list(adapter)
                if rng.random() < 0.5:
                    second[int(rng.integers(0, m))] = str(rng.choice(alpha))
                tail = tail[:int(rng.integers(0, 30))] + "".join(second)[:int(rng.integers(1, m + 1))] + tail[30:]
            reads.append(head + "".join(ad) + tail)
        seqs, offsets = orc.pack_reads(reads)
        for skip in (False, True):
            counts = compare(model, adapter, rate, min_overlap, seqs, offsets, skip=skip, label=f"indel1 {it} m {m} skip {skip}")
            if counts is not None and len(counts) > 5:
                indel1 += int(counts[5])
    assert indel1 > 30000, indel1


def test_tails_with_substitutions(model):
    """EXACT_TAIL generalised: the read ends with adapter[0:i] carrying substitutions (rows of the last column with
    clean diagonals have exact scores), next to rows reached by insertions / deletions (bounds only), behind
    shifted or partial copies; adapters of every kind and length (all forms)."""
    rng = np.random.default_rng(66)
    tails = 0
    for it in range(150):
        m = int(rng.choice([6, 10, 16, 20, 25, 31, 32, 33, 34, 40, 50, 64]))
        kind = it % 3
        if kind == 0:
            adapter = "".join(rng.choice(list("ACGT"), size=m))
        elif kind == 1:
            unit = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 4))))
            adapter = (unit * 70)[:m]
        else:
            adapter = "".join(rng.choice(list("AC"), size=m))
        rate = float(rng.choice([0.1, 0.1, 0.15, 0.2, 0.3]))
        alpha = list("ACGT") if kind != 2 else list("ACCA")
        reads = []
        for _ in range(500):
            i = int(rng.integers(1, m + 1))
            end = list(adapter[:i])
            for _e in range(int(rng.choice([0, 1, 1, 2, 3]))):
                end[int(rng.integers(0, len(end)))] = str(rng.choice(alpha))
            u = rng.random()
            if u < 0.15 and len(end) > 1:
                del end[int(rng.integers(0, len(end)))]
            elif u < 0.3:
                end.insert(int(rng.integers(0, len(end) + 1)), str(rng.choice(alpha)))
            front = ""
            if rng.random() < 0.4:
                s0 = int(rng.integers(0, 4))
                front = adapter[s0:s0 + int(rng.integers(0, m))]
            body = "".join(rng.choice(alpha, size=int(rng.integers(0, 80))))
            reads.append(body + front + "".join(end))
        seqs, offsets = orc.pack_reads(reads)
        for skip in (False, True):
            counts = compare(model, adapter, rate, int(rng.choice([1, 3, 5])), seqs, offsets, skip=skip,
                             label=f"tails {it} m {m} skip {skip}")
            if counts is not None:
                tails += int(counts[2])
    assert tails > 40000, tails


def test_refined_window_start(model):
    """A window start the kernels do not use yet (bm_refined_columns in the host model): the chunks' first
    positions instead of the first hit of any chunk.  Exactness of the rule, fuzzed here so that the kernel work can
    rely on it: random adapters and rates, ragged reads with up to two edited copies (indels included), the TruSeq
    shapes of the benchmark, and second copies / partial copies at every distance."""
    rng = np.random.default_rng(77)
    done = 0
    for it in range(200):
        m = int(rng.integers(4, 65))
        adapter = "".join(rng.choice(list("ACGT"), size=m)) if it % 5 else "".join(rng.choice(list("AC"), size=m))
        rate = float(rng.choice([0.0, 0.05, 0.1, 0.15, 0.2, 0.3]))
        min_overlap = int(rng.choice([1, 3, 5, m]))
        seqs, offsets = random_reads(rng, adapter, 600, int(rng.choice([20, 60, 100, 170])),
                                     float(rng.choice([0.0, 0.03, 0.1, 0.2])), float(rng.choice([0.0, 0.01])))
        if compare(model, adapter, rate, min_overlap, seqs, offsets, skip="refined", label=f"refined {it}") is not None:
            done += 1
    assert done > 150
    for seed, gen in ((12, dict(p_adapter=0.25, p_edit=0.02, p_n=0.005)), (15, dict(p_adapter=0.9, p_edit=0.1, p_n=0.02))):
        seqs, offsets = orc.synth_reads(seed, 0, 100_000, 150, [TRUSEQ], **gen)
        assert compare(model, TRUSEQ, 0.1, 3, seqs, offsets, skip="refined", label=f"refined truseq {seed}") is not None
    # a second (partial, shifted, edited) copy in front of / behind a full one, at every distance
    reads = []
    body = "".join(rng.choice(list("ACGT"), size=60))
    for d in range(0, 70):
        for cut in (33, 20, 12, 9):
            for edit in (0, 1, 2):
                second = list(TRUSEQ[:cut])
                for _ in range(edit):
                    second[int(rng.integers(0, len(second)))] = str(rng.choice(list("ACGT")))
                reads.append(body[:20] + TRUSEQ + body[20:20 + d] + "".join(second))
                reads.append(body[:10] + "".join(second) + body[10:10 + d] + TRUSEQ + body[40:])
    seqs, offsets = orc.pack_reads(reads)
    assert compare(model, TRUSEQ, 0.1, 3, seqs, offsets, skip="refined", label="refined copies") is not None
