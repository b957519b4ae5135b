"""Host-model fuzz of the streaming multi-adapter path (cutadapt_amd/csrc/multi2.h: the k-mer table of the REF and
WIDE families, pair classes, scan windows, the error-free suffix compare) against the oracle.

No GPU: tests/host_model/multi2_model.cpp compiles the product's table builder and rules with g++ and replays what
k_multi_stream / k_multi_scan do, read by read; the merged result must equal MultipleAdapters.match_to of the
reference (adapters.py:1265-1286) = the oracle's kmers_present + locate of every adapter on the WHOLE read, best
match by (score, errors, first adapter)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from cutadapt_amd import _lib, workloads
from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_model", "multi2_model.cpp")
SO = os.path.join(HERE, "host_model", "libmulti2_model.so")
CSRC = os.path.join(HERE, "..", "cutadapt_amd", "csrc")


@pytest.fixture(scope="module")
def model():
    deps = [SRC, os.path.join(HERE, "host_model", "back_model.cpp")] + [os.path.join(CSRC, h) for h in
                                                                          ("back_scan.h", "cah_device.h", "multi2.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", SRC, "-o", SO], check=True)
    L = C.CDLL(SO)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
    L.m2m_match_batch.argtypes = [C.c_char_p, i32, i32, vp, i32, vp, vp, C.c_char_p, vp, vp, i64, vp, vp, vp, i32, vp, vp]
    return L


def matcher_blobs(adapters, rate, min_overlap):
    specs = [_lib.MatcherSpec(a, rate, 14, False, False, 1, min_overlap) for a in adapters]
    L = _lib.lib()
    need = C.c_size_t(0)
    blobs = b""
    # (one plan per adapter: the debug entry hands out matcher 0's tables as cah_plan_create built them)
    for s in specs:
        plan = _lib.Plan([s])
        _lib.check(L.cah_plan_debug_matcher(plan.handle, 0, None, 0, C.byref(need)))
        buf = (C.c_uint8 * need.value)()
        _lib.check(L.cah_plan_debug_matcher(plan.handle, 0, buf, need.value, C.byref(need)))
        blobs += bytes(buf)
    return blobs


def ref_sets(adapters, rate, min_overlap, sets=None):
    """the reference's search sets (kmer_heuristic), flattened for the model; also returned as the oracle wants them"""
    ra, rw, rk, per_adapter = [], [], b"", []
    for a, ad in enumerate(adapters):
        pk = sets[a] if sets is not None else create_positions_and_kmers(ad, min_overlap, rate, True, False)
        per_adapter.append(pk)
        for start, stop, kmers in pk:
            assert stop is None and start <= 0
            for kmer in kmers:
                ra.append(a); rw.append(255 if start == 0 else -start); rk += kmer.encode() + b"\0"
    return np.array(ra, dtype=np.int32), np.array(rw, dtype=np.int32), rk, per_adapter


def oracle_multiple(adapters, rate, min_overlap, per_adapter, seqs, offsets):
    n = len(offsets) - 1
    want6 = np.zeros((n, 6), dtype=np.int32)
    want_st = np.zeros(n, dtype=np.uint8)
    want_best = np.full(n, -1, dtype=np.int32)
    for idx, ad in enumerate(adapters):
        oa = orc.Aligner(ad, rate, 14, False, False, 1, min_overlap)
        of = orc.KmerFinder(per_adapter[idx])
        c6, st = orc.match_batch(oa, of, seqs, offsets)
        f = st == 1
        better = f & ((want_st == 0) | (c6[:, 4] > want6[:, 4]) | ((c6[:, 4] == want6[:, 4]) & (c6[:, 5] < want6[:, 5])))
        want6[better] = c6[better]
        want_best[better] = idx
        want_st[better] = 1
    return want6, want_st, want_best


def run(model, adapters, rate, min_overlap, seqs, offsets, label, sets=None, must_build=True, pads=None):
    m = len(adapters[0])
    blobs = matcher_blobs(adapters, rate, min_overlap)
    ra, rw, rk, per_adapter = ref_sets(adapters, rate, min_overlap, sets)
    n = len(offsets) - 1
    want6, want_st, want_best = oracle_multiple(adapters, rate, min_overlap, per_adapter, seqs, offsets)
    totals = None
    for subs in (0, 1):
        out6 = np.zeros((n, 6), dtype=np.int32)
        status = np.zeros(n, dtype=np.uint8)
        best = np.zeros(n, dtype=np.int32)
        stats = np.zeros(16, dtype=np.int64)
        rc = model.m2m_match_batch("".join(adapters).encode(), len(adapters), m, blobs, len(ra), ra.ctypes.data,
                                   rw.ctypes.data, rk, seqs.ctypes.data, offsets.ctypes.data, n, out6.ctypes.data,
                                   status.ctypes.data, best.ctypes.data, subs, stats.ctypes.data,
                                   None if pads is None else pads.ctypes.data)
        if rc == 1:
            assert not must_build, f"{label}: the tables were not built"
            return None
        # (reads with bytes >= 0x80: the reference raises; the oracle's batch form flags them 2 as well)
        bad = np.nonzero((status != want_st) | (out6 != want6).any(axis=1) | ((status == 1) & (best != want_best)))[0]
        if len(bad):
            r = int(bad[0])
            read = bytes(seqs[offsets[r]:offsets[r + 1]]).decode("latin-1")
            raise AssertionError(f"{label} (subs {subs}): {len(bad)} of {n} reads differ; first: read {r} {read!r} rate {rate} O "
                                 f"{min_overlap} model {status[r]} {out6[r].tolist()} adapter {best[r]} "
                                 f"({adapters[best[r]] if best[r] >= 0 else None}) oracle {want_st[r]} {want6[r].tolist()} "
                                 f"adapter {want_best[r]} ({adapters[want_best[r]] if want_best[r] >= 0 else None})")
        if subs == 0:
            totals = stats
    return totals


def rand_adapters(rng, count, m):
    return ["".join(rng.choice(list("ACGT"), size=m)) for _ in range(count)]


def test_c4_workload(model):
    """BASELINE C4: 96 random 33-mers, the synthetic reads of the benchmark"""
    ads = workloads.SPECS["C4"]["adapters"]
    seqs, offsets = orc.synth_reads(4, 0, 6000, 150, ads)
    st = run(model, ads, 0.1, 3, seqs, offsets, "C4")
    n = 6000
    # what the design counts on: most pairs are decided by the suffix compare or scanned over a short window
    # (round 6: sub-classes of the tail rows with their own windows -- 4.5 pairs per read before, ~2 now)
    assert st[0] + st[1] + st[2] < 2.5 * n and st[3] > 0.8 * n and st[7] > 0.4 * n, st.tolist()
    for seed, gen in ((41, dict(p_adapter=0.9, p_edit=0.08, p_n=0.02)), (42, dict(p_adapter=0.0, p_edit=0.0, p_n=0.0))):
        seqs, offsets = orc.synth_reads(seed, 0, 2500, 150, ads, **gen)
        run(model, ads, 0.1, 3, seqs, offsets, f"C4 seed {seed}")


def tail_reads(rng, adapters, n_reads, read_len, p_n=0.0):
    """reads that END with a (possibly edited) prefix of some adapter: every overlap length, errors placed so that they
    land in the margins of the windows (insertions near the front of the overlap), second partial copies"""
    reads = []
    bases = list("ACGT")
    for _ in range(n_reads):
        ad = adapters[int(rng.integers(0, len(adapters)))]
        L = int(rng.integers(1, len(ad) + 1))
        part = list(ad[:L])
        for _e in range(int(rng.integers(0, 5))):
            if not part:
                break
            u = rng.random()
            pos = int(rng.integers(0, len(part)))
            if rng.random() < 0.5:
                pos = min(pos, int(rng.integers(0, 4)))           # near the front of the overlap: shifts its first chunk out
            if u < 0.4:
                part[pos] = str(rng.choice(bases))
            elif u < 0.75:
                part.insert(pos, str(rng.choice(bases)))
            else:
                del part[pos]
        body = list(rng.choice(bases, size=read_len))
        if rng.random() < 0.3:                                    # something else of an adapter further in
            other = adapters[int(rng.integers(0, len(adapters)))]
            cut = other[int(rng.integers(0, 10)):][:int(rng.integers(3, len(other) + 1))]
            at = int(rng.integers(0, read_len))
            body[at:at + len(cut)] = list(cut)
            body = body[:read_len]
        s = body[:max(0, read_len - len(part))] + part
        s = s[-read_len:] if len(s) > read_len else s
        if p_n:
            s = [("N" if rng.random() < p_n else c) for c in s]
        reads.append("".join(s))
    return reads


def test_tails_margins_and_second_copies(model):
    """partial adapters at the read end with indels that move their chunks into and out of the reference's windows"""
    rng = np.random.default_rng(77)
    for it in range(30):
        m = int(rng.choice([20, 25, 30, 33, 34, 40, 50, 64]))
        count = int(rng.choice([2, 8, 24]))
        ads = rand_adapters(rng, count, m)
        if it % 3 == 0:                                           # near-duplicates: shared k-mers between adapters
            ads = [ads[0]] + [ads[0][:i] + ("A" if ads[0][i] != "A" else "C") + ads[0][i + 1:] for i in rng.integers(0, m, size=count - 1)]
        rate = float(rng.choice([0.1, 0.12, 0.15, 0.2]))
        O = int(rng.choice([1, 3, 5, 8]))
        n_len = int(rng.choice([60, 100, 150]))
        reads = tail_reads(rng, ads, 1500, n_len, p_n=float(rng.choice([0.0, 0.01])))
        seqs, offsets = orc.pack_reads(reads)
        run(model, ads, rate, O, seqs, offsets, f"tails {it} m {m} x {count}", must_build=False)


def test_adapter_families_that_share_their_kmers(model):
    """near-duplicate adapters (a family that differs in one position each): homes of the directory with more entries than
    its count field holds are walked to their end (multi2.h: m2_home_of)"""
    rng = np.random.default_rng(80)
    for count, m in ((20, 33), (30, 24), (12, 40)):
        base = rand_adapters(rng, 1, m)[0]
        ads = [base]
        for i in range(1, count):
            p = (7 * i) % m
            ads.append(base[:p] + "ACGT"[("ACGT".index(base[p]) + 1 + i % 3) % 4] + base[p + 1:])
        ads = list(dict.fromkeys(ads))
        reads = tail_reads(rng, ads, 1200, 150)
        seqs, offsets = orc.pack_reads(reads)
        run(model, ads, 0.1, 3, seqs, offsets, f"family of {len(ads)}, m {m}")
        seqs, offsets = orc.synth_reads(12, 0, 1500, 150, ads, p_adapter=0.7, p_edit=0.04)
        run(model, ads, 0.1, 3, seqs, offsets, f"family of {len(ads)}, m {m}, synthetic")


def test_views_in_a_padded_frame(model):
    """Round 6: the views of a uniform batch through the streaming form -- every view END-aligned in a frame of the parent's
    length (NULs in front): prefilter and scan on the frame, the cell DP and every coordinate on the view, a shortcut whose
    alignment would begin inside the pad handed to the cell DP.  Views cut at either end (a cut at the 5' end leaves adapter
    TAILS at the view's first characters: the pad's case), every length from 0 on, against the oracle on the views."""
    rng = np.random.default_rng(81)
    built = 0
    for it in range(16):
        m = int(rng.choice([20, 24, 30, 33, 34, 40, 64]))
        count = int(rng.choice([2, 8, 24]))
        ads = rand_adapters(rng, count, m)
        rate = float(rng.choice([0.1, 0.1, 0.15, 0.2]))
        O = int(rng.choice([1, 3, 5]))
        N = int(rng.choice([100, 150, 151, 160]))
        reads = tail_reads(rng, ads, 700, N, p_n=0.005)
        reads = [r if len(r) == N else (r + "A" * N)[:N] for r in reads]
        sq2, of2 = orc.synth_reads(int(rng.integers(1, 10 ** 6)), 0, 900, N, ads, p_adapter=0.85,
                                   p_edit=float(rng.choice([0.02, 0.06, 0.1])), p_n=0.005)
        reads += [bytes(sq2[of2[i]:of2[i + 1]]).decode("latin-1") for i in range(900)]
        views = []
        for i, r in enumerate(reads):
            mode = i % 4
            a = int(rng.integers(0, N + 1)) if mode in (1, 3) else 0           # cut at the 5' end
            b = int(rng.integers(a, N + 1)) if mode in (2, 3) else N           # ... and at the 3' end
            if mode == 1 and rng.random() < 0.5:
                # a cut that lands inside an adapter copy, near its head (the alignment would begin in the pad)
                for ad in ads:
                    at = r.find(ad[4:14])
                    if at >= 0:
                        a = min(N, at + int(rng.integers(0, 4)))
                        break
            views.append(r[a:b])
        seqs, offsets = orc.pack_reads(views)
        pads = (N - np.diff(offsets)).astype(np.int32)
        if run(model, ads, rate, O, seqs, offsets, f"views it {it} m {m} x {count} rate {rate} O {O} N {N}", must_build=False, pads=pads) is not None:
            built += 1
    assert built >= 8, built


def test_low_complexity_and_lowercase(model):
    rng = np.random.default_rng(78)
    done = 0
    for it in range(20):
        unit = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 4))))
        m = int(rng.choice([24, 33, 40]))
        ads = [(unit * 40)[:m], (unit[::-1] * 40)[:m]] + rand_adapters(rng, 6, m)
        ads = list(dict.fromkeys(ads))
        reads = tail_reads(rng, ads, 800, 150)
        reads = [r.lower() if i % 5 == 0 else r for i, r in enumerate(reads)]
        reads += [(unit * 80)[:150], "A" * 150, "ACGT" * 37 + "AC"]
        seqs, offsets = orc.pack_reads(reads)
        if run(model, ads, 0.1, 3, seqs, offsets, f"lowcomp {it}", must_build=False) is not None:
            done += 1
    assert done >= 10


def test_custom_search_sets(model):
    """search sets that are NOT what kmer_heuristic builds (the C ABI takes any).  Round 6: the streaming form proves the
    reference's kmers_present from its own k-mer family, which needs three properties of the sets (multi2.h: (i), (iii),
    (iv)).  Sets that have them -- the heuristic's with further k-mers, wider windows -- take the form and stay exact (the
    cell DP's corner check evaluates whatever the sets hold); sets without them are not built (older kernels)."""
    rng = np.random.default_rng(79)
    checked = 0
    for m, rate in ((15, 0.1), (33, 0.1), (24, 0.15)):
        ads = rand_adapters(rng, 6, m)
        sets = []
        for ad in ads:
            pk = [(start, stop, list(kmers)) for start, stop, kmers in create_positions_and_kmers(ad, 3, rate, True, False)]
            # wider windows for the tail sets, k-mers of our own in a window of their own and in the whole read
            pk = [((start - 2) if start < -4 else start, stop, kmers) for start, stop, kmers in pk]
            pk.append((-12, None, [ad[2:8]]))
            pk.append((0, None, [ad[1:8]]))
            sets.append(pk)
        for n_len in (40, 150):
            reads = tail_reads(rng, ads, 3000, n_len)
            seqs, offsets = orc.pack_reads(reads)
            st = run(model, ads, rate, 3, seqs, offsets, f"custom sets m {m}, n {n_len}", sets=sets)
            checked += int(st[15])
        seqs, offsets = orc.synth_reads(9, 0, 3000, 150, ads, p_adapter=0.6, p_edit=0.05)
        run(model, ads, rate, 3, seqs, offsets, "custom sets, synthetic", sets=sets)
    assert checked > 0, "no match reached further back than its error class: the corner check was not exercised"
    # ... and sets without the properties are not built (the plan takes the older kernels): (i) a chunk of the whole adapter
    # that is no whole-read k-mer, (iv) an error class without its chunks, (iii) no prefix for the exact overlaps
    ads = rand_adapters(rng, 4, 33)
    seqs, offsets = orc.pack_reads(tail_reads(rng, ads, 200, 150))
    for label, make in (("(i)", lambda ad: [(-3, None, [ad[:3]]), (-25, None, [ad[2:8], ad[9:15]]), (0, None, [ad[0:9], ad[20:28]])]),
                        ("(iv)", lambda ad: [s for s in create_positions_and_kmers(ad, 3, 0.1, True, False) if s[0] != -19]),
                        ("(iii)", lambda ad: [s for s in create_positions_and_kmers(ad, 3, 0.1, True, False) if s[0] != -4])):
        sets = [[(a, b, list(c)) for a, b, c in make(ad)] for ad in ads]
        assert run(model, ads, 0.1, 3, seqs, offsets, f"custom sets without {label}", sets=sets, must_build=False) is None, label


def test_occurrence_windows_regressions_and_mixed_plans(model):
    """What a soak with other seeds found in round 4, and its generator as a test.
    (1) A whole-adapter chunk whose STRING is a chunk of a tail class too (20-character adapters at rate 0.15: chunks of
        five either way): its single occurrence near the read's end stands for rows of the last column as well, so the pair
        must not take the window of that occurrence alone (multi2.h: tail_role).
    (2) Reads shorter than the adapter, whole-read pairs with the adapter INSIDE the read (one and several chunk hits,
        many edits), repetitive adapters, rates 0.08-0.25."""
    import random
    ads = ["CTTTATATAGTCCCCCACAC"[1:] + "T", "GGTCAATGCCGATTGACTTA"]
    assert ads[0] == "TTTATATAGTCCCCCACACT"
    reads = ["CTCCTCAGAAGGCCCCGGAAACCGAGCGCCCATATGAGTTAAATACTCTAGGGTCATCTGTATATAGTCCGCCACAC",
             "CTCCTCAGAAGGCCCCGGAAACCGAGCGCCCATATGAGTTAAATACTCTAGGGTCATCTGTATATAGTCCCCCACAC",
             "CTCCTCAGAAGGCCCCGGAATTTATATAGTCCGCCACACTGAGTTAAATACTCTAGGGTCATCTGAAAAAAAAAAAA"]
    seqs, offsets = orc.pack_reads(reads)
    run(model, ads, 0.15, 8, seqs, offsets, "chunk that is a tail chunk too")
    rng = np.random.default_rng(31337)
    prng = random.Random(31338)
    built = 0
    for it in range(14):
        m = int(rng.choice([16, 20, 24, 28, 30, 33, 35, 40, 64]))
        count = int(rng.choice([2, 3, 8, 24]))
        ads = ["".join(prng.choice("ACGT") for _ in range(m)) for _ in range(count)]
        if it % 5 == 3:
            unit = "".join(prng.choice("ACGT") for _ in range(int(rng.choice([2, 3, 5]))))
            ads[0] = (unit * m)[:m]
        rate = float(rng.choice([0.08, 0.1, 0.12, 0.15, 0.2, 0.25]))
        O = int(rng.choice([1, 3, 5, 8]))
        n = int(rng.integers(16, 161))
        reads = tail_reads(rng, ads, 700, n, p_n=float(rng.choice([0.0, 0.01])))
        reads = [r if len(r) == n else (r + "A" * n)[:n] for r in reads]
        sq2, of2 = orc.synth_reads(int(rng.integers(1, 10 ** 6)), 0, 700, n, ads, p_adapter=float(rng.choice([0.3, 0.8])),
                                   p_edit=float(rng.choice([0.03, 0.08, 0.12])), p_n=0.005)
        sq, offs = orc.pack_reads(reads)
        sq = np.concatenate([sq, sq2])
        offs = np.concatenate([offs, of2[1:] + offs[-1]])
        st = run(model, ads, rate, O, sq, offs, f"mixed it {it} m {m} x {count} rate {rate} O {O} n {n}", must_build=False)
        built += st is not None
    assert built >= 6, built
