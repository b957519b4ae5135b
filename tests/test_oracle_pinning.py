"""Pin the CPU oracle (oracle/cutadapt_oracle.c) -- CPU only.

1. against the committed golden vectors that tests/golden/make_golden.py generated from the
   reference itself (these travel to the GPU box, the reference does not);
2. against the reference's own known-answer tests (restated from reference
   tests/test_align.py, tests/test_kmer_finder.py);
3. when oracle/_ref is present (build container), by a randomized differential run against
   the compiled reference.
"""
import random

import numpy as np
import pytest

BACK, FRONT, PREFIX, SUFFIX, ANYWHERE = 14, 11, 8, 2, 15


def test_golden_locate(golden, orc):
    cases = golden("locate.json")
    assert len(cases) >= 2000
    cache = {}
    for c in cases:
        key = (c["ref"], c["rate"], c["flags"], c["wr"], c["wq"], c["indel_cost"], c["min_overlap"])
        if key not in cache:
            cache[key] = orc.Aligner(*key)
        a = cache[key]
        assert a.effective_length == c["effective_length"]
        got = a.locate(c["query"])
        want = tuple(c["result"]) if c["result"] is not None else None
        assert got == want, c


def test_golden_truseq(golden, orc):
    g = golden("truseq.json")
    a = orc.Aligner(g["ref"], g["rate"], g["flags"], False, False, 1, g["min_overlap"])
    n_hit = 0
    for c in g["cases"]:
        want = tuple(c["result"]) if c["result"] is not None else None
        assert a.locate(c["query"]) == want
        n_hit += want is not None
    assert n_hit > 300


def test_golden_comparers(golden, orc):
    for c in golden("comparers.json"):
        cls = orc.PrefixComparer if c["kind"] == "prefix" else orc.SuffixComparer
        cmp_ = cls(c["ref"], c["rate"], c["wr"], c["wq"], c["min_overlap"])
        assert cmp_.effective_length == c["effective_length"]
        want = tuple(c["result"]) if c["result"] is not None else None
        assert cmp_.locate(c["query"]) == want, c


def test_golden_kmers(golden, orc):
    for c in golden("kmers.json"):
        f = orc.KmerFinder([(a, b, k) for a, b, k in c["sets"]], c["wr"], c["wq"])
        for read, want in c["reads"]:
            assert f.kmers_present(read) == want, (c["sets"], read)


def test_golden_illumina_info(golden, orc):
    """reference tests/cut/illumina.info.txt coordinates via filter -> locate"""
    from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
    g = golden("illumina_info.json")
    a = orc.Aligner(g["adapter"], g["max_errors"], BACK, False, False, 1, g["min_overlap"])
    f = orc.KmerFinder(create_positions_and_kmers(g["adapter"], g["min_overlap"], g["max_errors"], True, False))
    seqs, offsets = orc.pack_reads(g["reads"])
    out6, status = orc.match_batch(a, f, seqs, offsets)
    n_found = 0
    for i, want in enumerate(g["expected"]):
        if want is None:
            assert status[i] == 0
        else:
            assert status[i] == 1
            assert [out6[i, 5], out6[i, 2], out6[i, 3]] == want
            n_found += 1
    assert n_found == 56


# ---- the reference's known answers (reference tests/test_align.py) ----------------------------
def test_known_answers_aligner(orc):
    A = orc.Aligner
    assert A("", 0, flags=0, min_overlap=0).locate("") == (0, 0, 0, 0, 0, 0)             # :65-68
    assert A("CCAGTCCTCT", 0.3, flags=PREFIX).locate("CCAGTCCTTTCCTGAGAGT") == (0, 10, 0, 10, 8, 1)  # :79-82
    assert A("TCGATC", 1.5 / 6, flags=PREFIX).locate("TCGATGC") == (0, 6, 0, 6, 4, 1)    # :88-90
    assert A("GCCGAACTTCTTAGACTGCCTTAAGGACGT", 0.1, flags=BACK).locate(
        "CAAATCACCAGAAGGCGCCTAACTTCTTAGACTGCC") == (0, 20, 16, 36, 18, 1)                 # :94-107
    assert A("TTTT", 0.25, flags=BACK).locate("CCTTTT") == (0, 4, 2, 6, 4, 0)            # :110-118
    assert A("TTTTTT", 0.25, flags=BACK).locate("CCTTTT") == (0, 4, 2, 6, 4, 0)          # :121-129
    assert A("TTT", 1 / 3, flags=BACK).locate("CCTTTT")[:4] == (0, 3, 2, 5)              # :132-138
    s, t = "A" * 17, "ACAG" + "A" * 42
    assert A(s, 0.0, BACK).locate(t) == (0, 17, 4, 21, 17, 0)                            # :141-146
    assert A("CTGATCTGGCCG", 0.1, BACK).locate("AAAAGGG") is None                        # :410-412
    with pytest.raises(ValueError):
        A("NNNNN", 0.1, wildcard_ref=True)                                               # :59-63
    with pytest.raises(ValueError):
        A("ACGT", 0.1, indel_cost=0)
    assert A("NNACGT", 0, BACK, wildcard_ref=True).locate("AAANTACGTAAA") == (0, 6, 3, 9, 6, 0)  # :262-267


def test_known_answers_n_wildcards(orc):
    ref_seq = "AGGNNNNNNNNNNNNNNTTC"
    a = orc.Aligner(ref_seq, 0.1, BACK, wildcard_ref=True, min_overlap=3)                # :326-349
    assert a.effective_length == 6
    assert a.locate("TTC") is None
    assert a.locate("AGG")[:4] == (0, 3, 0, 3)
    assert a.locate("AGGCCCCCCC")[:4] == (0, 10, 0, 10)
    assert a.locate("ATGCCCCCCC") is None
    assert a.locate("AGGCCCCCCCCCCCCCCATC") is None
    assert a.locate("CCC" + ref_seq.replace("N", "G") + "AAA") == (0, 20, 3, 23, 20, 0)
    a = orc.Aligner(ref_seq, 0.1, FRONT, wildcard_ref=True, min_overlap=3)               # :352-376
    assert a.locate("TTC")[:4] == (17, 20, 0, 3)
    assert a.locate("TGC") is None
    assert a.locate("CCCCCCCTTC")[:4] == (10, 20, 0, 10)
    assert a.locate("CCCCCCCGTC") is None


WILDCARD_SEQUENCES = ["CCCATTGATC", "CCCRTTRATC", "YCCATYGATC", "CSSATTSATC", "CCCWWWGATC",
                      "CCCATKKATC", "CCMATTGMTC", "BCCATTBABC", "BCCATTBABC", "CCCDTTDADC",
                      "CHCATHGATC", "CVCVTTVATC", "CCNATNGATC", "CCCNTTNATC"]


def test_known_answers_wildcards(orc):                                                   # :379-408
    r = "CATCTGTCC" + WILDCARD_SEQUENCES[0] + "GCCAGGGTTGATTCGGCTGATCTGGCCG"
    for a in WILDCARD_SEQUENCES:
        assert orc.Aligner(a, 0.0, BACK, wildcard_ref=True).locate(r) == (0, 10, 9, 19, 10, 0)
    assert orc.Aligner("CCCXTTXATC", 0.0, BACK, wildcard_ref=True).locate(r) is None
    a = WILDCARD_SEQUENCES[0]
    for s in WILDCARD_SEQUENCES + ["CCCXTTXATC"]:
        r = "CATCTGTCC" + s + "GCCAGGGTTGATTCGGCTGATCTGGCCG"
        res = orc.Aligner(a, 0.0, BACK, wildcard_query=True).locate(r)
        assert res is None if "X" in s else res == (0, 10, 9, 19, 10, 0)
    for a in WILDCARD_SEQUENCES:
        for s in WILDCARD_SEQUENCES:
            r = "CATCTGTCC" + s + "GCCAGGGTTGATTCGGCTGATCTGGCCG"
            assert orc.Aligner(a, 0.0, BACK, wildcard_ref=True, wildcard_query=True).locate(r) == (0, 10, 9, 19, 10, 0)


def test_known_answers_comparers(orc):                                                   # :191-323
    P, S = orc.PrefixComparer, orc.SuffixComparer
    assert P("AAXAA", 0.9).locate("AAAAATTTTTTTTT") == (0, 5, 0, 5, 3, 1)
    assert P("AANAA", 0.9, wildcard_ref=True).locate("AACAATTTTTTTTT") == (0, 5, 0, 5, 5, 0)
    assert P("XAAAAA", 0.9).locate("AAAAATTTTTTTTT") == (0, 6, 0, 6, 2, 2)
    assert P("NNACGT", 0.9, wildcard_ref=True).locate("NTACGTAA") == (0, 6, 0, 6, 6, 0)
    assert P("NNACGT", 0.9, wildcard_ref=True).locate("YTACGTAA") == (0, 6, 0, 6, 6, 0)
    assert S("AAXAA", 0.9).locate("TTTTTTTAAAAA") == (0, 5, 7, 12, 3, 1)
    assert S("AANAA", 0.9, wildcard_ref=True).locate("TTTTTTTAACAA") == (0, 5, 7, 12, 5, 0)
    assert S("AAAAAX", 0.9).locate("TTTTTTTAAAAA") == (0, 6, 6, 12, 2, 2)
    for ref_seq in ("axcgt", "AXCGT"):
        c = P(ref_seq, 0.4)
        assert c.locate("TTG") is None and c.locate("AGT") is not None and c.locate("agt") is not None
        assert c.locate("CGT") is None
        c = S(ref_seq, 0.4)
        assert c.locate("TTG") is None and c.locate("AGT") is not None and c.locate("CGT") is not None
    for cls in (P, S):
        c = cls("CNNNNNNNNGTT", 0.25, wildcard_ref=True)
        assert c.locate("CAAAAAAAAGTT") is not None and c.locate("CAAAAAAAAGTA") is not None
        assert c.locate("CAAAAAAAAGAA") is None


def test_known_answers_kmer_finder(orc):                                                 # test_kmer_finder.py
    table = [
        ("ACGT", [(0, None, ["ACGT"])], True), ("ACGA", [(0, None, ["ACGT"])], False),
        ("ACGTACG", [(0, 6, ["ACGTAC"])], True), ("ACGTACG", [(0, 5, ["ACGTAC"])], False),
        ("GGGGACGT", [(-4, None, ["ACGT"])], True), ("GGGACGTG", [(-4, None, ["ACGT"])], False),
        ("acgt", [(0, None, ["ACGT"])], True), ("ACGT", [(0, None, ["acgt"])], True),
    ]
    for seq, sets, want in table:
        assert orc.KmerFinder(sets).kmers_present(seq) == want, (seq, sets)
    assert orc.KmerFinder([(0, None, ["ACGN"])], ref_wildcards=True).kmers_present("ACGT")
    assert not orc.KmerFinder([(0, None, ["ACGN"])]).kmers_present("ACGT")
    assert orc.KmerFinder([(0, None, ["ACGT"])], query_wildcards=True).kmers_present("ACGN")
    with pytest.raises(ValueError):
        orc.KmerFinder([(0, None, ["A" * 65])])
    # more than 64 characters in one search set are split over several words
    rng = random.Random(7)
    kmers = ["".join(rng.choice("ACG") for _ in range(30)) for i in range(5)]
    f = orc.KmerFinder([(0, None, kmers)])
    for k in kmers:
        assert f.kmers_present("TT" + k + "GG")
    assert not f.kmers_present("T" * 100)


# ---- differential run against the compiled reference (build container only) ------------------
def test_differential_vs_reference(orc, ref):
    if ref is None:
        pytest.skip("oracle/_ref not built here (the reference only exists in the build container)")
    rng = random.Random(99)
    alphabets = ["ACGT", "ACGTN", "ACGTNRYacgtn", "ACGTXNSWKMBDHVU"]
    n = 0
    for _ in range(6000):
        al = rng.choice(alphabets)
        m = rng.randint(1, 64) if rng.random() < 0.85 else rng.choice([65, 80, 130])   # the whole range the kernels serve
        adapter = "".join(rng.choice(al) for _ in range(m))
        args = (adapter, rng.choice([0, 0.1, 0.2, 0.35, 1.0, rng.random()]), rng.randint(0, 15),
                rng.random() < 0.3, rng.random() < 0.3, rng.choice([1, 1, 3, 100000]), rng.randint(1, min(m, 5)))
        try:
            ra = ref.Aligner(*args)
        except ValueError:
            with pytest.raises(ValueError):
                orc.Aligner(*args)
            continue
        oa = orc.Aligner(*args)
        for _ in range(3):
            q = "".join(rng.choice(al) for _ in range(rng.randint(0, 170)))
            if rng.random() < 0.6 and q:
                p = rng.randint(0, len(q))
                q = q[:p] + adapter[rng.randint(0, m - 1):] + q[p:]
            assert ra.locate(q) == oa.locate(q), (args, q)
            n += 1
    assert n > 10000


def test_differential_kmer_finder_and_comparers_vs_reference(orc, ref):
    """KmerFinder.kmers_present and both comparers of the oracle against the compiled reference"""
    if ref is None:
        pytest.skip("oracle/_ref not built here (the reference only exists in the build container)")
    rng = random.Random(199)
    alphabets = ["ACGT", "ACGTN", "ACGTNRYacgtn", "ACGTXNSWKMBDHVU"]
    n = 0
    for _ in range(1500):
        al = rng.choice(alphabets)
        rw, qw = rng.random() < 0.3, rng.random() < 0.3
        sets = []
        for _s in range(rng.randint(1, 5)):
            kind = rng.random()
            kmers = ["".join(rng.choice(al) for _ in range(rng.randint(1, 64 if rng.random() < 0.1 else 12)))
                     for _k in range(rng.randint(1, 9))]
            if kind < 0.4:
                sets.append((0, None, kmers))
            elif kind < 0.7:
                sets.append((-rng.randint(1, 70), None, kmers))
            else:
                start = rng.randint(0, 20)
                sets.append((start, start + rng.randint(1, 60), kmers))
        rf, of = ref.KmerFinder(sets, rw, qw), orc.KmerFinder(sets, rw, qw)
        for _q in range(6):
            # windows are kept inside the read: beyond it the reference reads out of bounds
            q = "".join(rng.choice(al) for _ in range(rng.randint(80, 200)))
            if rng.random() < 0.5:
                kmer = rng.choice(rng.choice(sets)[2])
                pos = rng.randint(0, len(q))
                q = q[:pos] + kmer + q[pos:]
            assert rf.kmers_present(q) == of.kmers_present(q), (sets, rw, qw, q)
            n += 1
    for _ in range(3000):
        al = rng.choice(alphabets)
        m = rng.randint(1, 80)
        adapter = "".join(rng.choice(al) for _ in range(m))
        args = (adapter, rng.choice([0, 0.1, 0.2, 0.5, 1.0]), rng.random() < 0.3, rng.random() < 0.3, rng.randint(1, 6))
        for rcls, ocls in ((ref.PrefixComparer, orc.PrefixComparer), (ref.SuffixComparer, orc.SuffixComparer)):
            try:
                rc = rcls(*args)
            except ValueError:
                with pytest.raises(ValueError):
                    ocls(*args)
                continue
            oc = ocls(*args)
            assert rc.effective_length == oc.effective_length
            for _q in range(3):
                q = "".join(rng.choice(al) for _ in range(rng.randint(0, 100)))
                if rng.random() < 0.6:
                    q = (adapter[:rng.randint(1, m)] + q) if rcls is ref.PrefixComparer else (q + adapter[rng.randint(0, m - 1):])
                assert rc.locate(q) == oc.locate(q), (args, q)
                n += 1
    assert n > 20000


def test_reference_own_tests_pass_on_the_compiled_reference(ref):
    """DESIGN.md section 6 as a committed fact: the reference's OWN tests of this path
    (/root/reference/tests/test_align.py, test_kmer_finder.py, test_kmer_heuristic.py) pass against
    oracle/_ref, the build the oracle and the goldens are pinned to.  Build container only."""
    import os
    import subprocess
    import sys
    tests_dir = "/root/reference/tests"
    if ref is None or not os.path.isdir(tests_dir):
        pytest.skip("needs /root/reference and oracle/_ref (build container only)")
    ref_root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
    env = dict(os.environ, PYTHONPATH=ref_root + os.pathsep + tests_dir)
    # --noconftest: the reference conftest imports cutadapt.cli (needs dnaio, not installed); these three test
    # modules use none of its fixtures
    cmd = [sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "--noconftest", "--rootdir", "/tmp", "-c", "/dev/null",
           os.path.join(tests_dir, "test_align.py"), os.path.join(tests_dir, "test_kmer_finder.py"),
           os.path.join(tests_dir, "test_kmer_heuristic.py")]
    out = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    tail = out.stdout.decode(errors="replace")[-1500:]
    assert out.returncode == 0, tail
    import re
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 85, tail


def test_synth_reads_are_deterministic_and_shardable(orc):
    ad = ["AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"]
    full, offs = orc.synth_reads(2, 0, 1000, 150, ad)
    assert full.shape == (150000,) and offs[-1] == 150000
    part, _ = orc.synth_reads(2, 400, 100, 150, ad)
    assert np.array_equal(full[400 * 150:500 * 150], part)      # shard independence
    other, _ = orc.synth_reads(3, 0, 1000, 150, ad)
    assert not np.array_equal(full, other)
    assert set(np.unique(full).tolist()) <= set(b"ACGTN")
    # roughly a quarter of the reads carry (a prefix of) the adapter
    reads = [bytes(full[i * 150:(i + 1) * 150]).decode() for i in range(1000)]
    frac = sum(ad[0][:12] in r for r in reads) / 1000
    assert 0.1 < frac < 0.3, frac
    n_frac = (full == ord("N")).mean()
    assert 0.003 < n_frac < 0.007
