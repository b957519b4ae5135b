"""INTEGRATION.md section 1 executed: the REFERENCE's own adapters.py (compiled unmodified into oracle/_ref
by oracle/build_ref.py) with its module globals Aligner / PrefixComparer / SuffixComparer / KmerFinder
rebound to the cutadapt_amd classes -- the two-import change a cutadapt maintainer would make
(reference src/cutadapt/adapters.py:15-24) -- must reproduce the golden results through the reference's
own match_to() code.  Also measures the cost of a per-read match_to() through that seam.  GPU only."""
import json
import os
import time

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def patched(ref):
    assert ref is not None, "oracle/_ref (the compiled reference) is needed: python -m oracle.build_ref"
    from cutadapt_amd._kmer_finder import KmerFinder
    from cutadapt_amd.align import Aligner, PrefixComparer, SuffixComparer
    A = ref.adapters
    names = ("Aligner", "PrefixComparer", "SuffixComparer", "KmerFinder")
    saved = {n: getattr(A, n) for n in names}
    A.Aligner, A.PrefixComparer, A.SuffixComparer, A.KmerFinder = Aligner, PrefixComparer, SuffixComparer, KmerFinder
    try:
        yield A
    finally:
        for n, v in saved.items():
            setattr(A, n, v)


def test_reference_adapters_on_hip_classes_illumina(hip, patched, golden):
    """tests/cut/illumina.info.txt coordinates (reference tests/test_info_file.py:14-32) through the
    reference's BackAdapter.match_to on top of the HIP Aligner / KmerFinder"""
    g = golden("illumina_info.json")
    ad = patched.BackAdapter(g["adapter"], max_errors=g["max_errors"], min_overlap=g["min_overlap"])
    assert type(ad.aligner).__module__.startswith("cutadapt_amd")
    assert type(ad.kmer_finder).__module__.startswith("cutadapt_amd")
    n_match = 0
    for read, exp in zip(g["reads"], g["expected"]):
        m = ad.match_to(read)
        if exp is None:
            assert m is None, read
        else:
            assert m is not None and [m.errors, m.rstart, m.rstop] == exp, (read, exp)
            n_match += 1
    assert n_match == 56


def test_reference_adapters_on_hip_classes_all_adapter_types(hip, patched, golden):
    """a slice of the adapter-class golden vectors (all nine classes, wildcards, no-indel comparers) through
    the reference's own classes"""
    cases = golden("adapters.json")[:300]
    n_reads = n_found = 0
    classes = set()
    for c in cases:
        cls = getattr(patched, c["cls"])
        ad = cls(c["sequence"], **c["kwargs"])
        classes.add(c["cls"])
        for read, exp in c["reads"]:
            m = ad.match_to(read)
            n_reads += 1
            if exp is None:
                assert m is None, (c["cls"], c["sequence"], read)
            else:
                assert m is not None and type(m).__name__ == exp["cls"], (c["cls"], c["sequence"], read)
                assert [m.astart, m.astop, m.rstart, m.rstop, m.score, m.errors] == exp["t"], (c, read)
                n_found += 1
    assert len(classes) >= 8 and n_reads >= 1000 and n_found >= 100


def test_reference_linked_and_multiple_on_hip_classes(hip, patched, ref):
    """LinkedAdapter / MultipleAdapters of the reference on the HIP classes == the unpatched reference"""
    import random
    rng = random.Random(77)
    A = patched
    front = A.PrefixAdapter("NNNNACGTACGT", max_errors=0.1)
    back = A.BackAdapter("AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", max_errors=0.1, min_overlap=3)
    linked = A.LinkedAdapter(front, back, front_required=True, back_required=False, name="l")
    multi = A.MultipleAdapters([A.BackAdapter(s, max_errors=0.1, min_overlap=3) for s in
                                ("AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", "CTGTCTCTTATACACATCT", "TGGAATTCTCGGGTGCCAAGG")])
    reads = []
    for _ in range(300):
        r = ("".join(rng.choice("ACGT") for _ in range(4)) + "ACGTACGT") if rng.random() < 0.7 else ""
        r += "".join(rng.choice("ACGT") for _ in range(rng.randint(10, 80)))
        if rng.random() < 0.6:
            r += rng.choice(["AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", "CTGTCTCTTATACACATCT", "TGGAATTCTCGGGTGCCAAGG"])[:rng.randint(3, 33)]
        reads.append(r)
    got_l = [linked.match_to(r) for r in reads]
    got_m = [multi.match_to(r) for r in reads]
    sig_l = [None if m is None else (m.front_match.rstop, None if m.back_match is None else
                                    (m.back_match.rstart, m.back_match.rstop, m.back_match.errors)) for m in got_l]
    sig_m = [None if m is None else (m.adapter.sequence, m.rstart, m.rstop, m.score, m.errors) for m in got_m]
    # the same objects built from the unpatched reference classes
    # the same objects built from the reference's own classes
    from oracle import ref_loader
    Aref = ref_loader.load().adapters
    saved = (Aref.Aligner, Aref.PrefixComparer, Aref.SuffixComparer, Aref.KmerFinder)
    Aref.Aligner, Aref.PrefixComparer, Aref.SuffixComparer, Aref.KmerFinder = (
        ref.Aligner, ref.PrefixComparer, ref.SuffixComparer, ref.KmerFinder)
    try:
        f2 = Aref.PrefixAdapter("NNNNACGTACGT", max_errors=0.1)
        b2 = Aref.BackAdapter("AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", max_errors=0.1, min_overlap=3)
        assert not type(b2.aligner).__module__.startswith("cutadapt_amd")
        l2 = Aref.LinkedAdapter(f2, b2, front_required=True, back_required=False, name="l")
        m2 = Aref.MultipleAdapters([Aref.BackAdapter(s, max_errors=0.1, min_overlap=3) for s in
                                    ("AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", "CTGTCTCTTATACACATCT", "TGGAATTCTCGGGTGCCAAGG")])
        want_l = [l2.match_to(r) for r in reads]
        want_m = [m2.match_to(r) for r in reads]
    finally:
        Aref.Aligner, Aref.PrefixComparer, Aref.SuffixComparer, Aref.KmerFinder = saved
    ref_l = [None if m is None else (m.front_match.rstop, None if m.back_match is None else
                                    (m.back_match.rstart, m.back_match.rstop, m.back_match.errors)) for m in want_l]
    ref_m = [None if m is None else (m.adapter.sequence, m.rstart, m.rstop, m.score, m.errors) for m in want_m]
    assert sig_l == ref_l
    assert sig_m == ref_m
    assert sum(x is not None for x in ref_l) > 100 and sum(x is not None for x in ref_m) > 100


def test_per_read_match_to_latency(hip):
    """a batch of one through the persistent per-thread staging (no hipMalloc/hipFree/device sync per call);
    the measured cost is written to gpurun_out/ for profiles/"""
    from cutadapt_amd.adapters import BackAdapter
    ad = BackAdapter("AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", max_errors=0.1, min_overlap=3)
    reads = ["ACGT" * 30 + "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"[:k] for k in range(3, 33)] + ["ACGTTGCA" * 18] * 30
    for r in reads:
        ad.match_to(r)                                   # warm-up: plan upload, scratch, code objects
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        for r in reads:
            ad.match_to(r)
    dt = (time.perf_counter() - t0) / (reps * len(reads))
    t0 = time.perf_counter()
    for _ in range(reps):
        for r in reads:
            ad.aligner.locate(r)
    dt_locate = (time.perf_counter() - t0) / (reps * len(reads))
    out = {"match_to_us": dt * 1e6, "aligner_locate_us": dt_locate * 1e6, "reads": len(reads), "reps": reps,
           "note": "Python BackAdapter.match_to(str) / Aligner.locate(str): one cah_*_batch_host call of one read each"}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "per_read_latency.json"), "w") as f:
        json.dump(out, f)
    print(out)
    assert dt < 500e-6, out        # round 1: ~1 ms per call (5-6 hipMalloc/hipFree + device syncs)
