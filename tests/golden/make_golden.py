#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference and oracle/_ref, see
oracle/build_ref.py); the JSON it writes is committed so that the GPU box -- where the
reference does not exist -- can check both the oracle and the HIP path against it.

    python tests/golden/make_golden.py

Sources of truth used here:
  * the reference's compiled Cython classes (Aligner, PrefixComparer, SuffixComparer,
    KmerFinder), its kmer_heuristic and adapters modules, imported from oracle/_ref;
  * the reference's own golden coordinates: tests/cut/illumina.info.txt for
    tests/data/illumina.fastq.gz with -a GCCGAACTTCTTAGACTGCCTTAAGGACGT
    (reference tests/test_info_file.py:14-32).
"""
import gzip
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import build_ref, ref_loader  # noqa: E402

REFERENCE = os.environ.get("CUTADAPT_REFERENCE", "/root/reference")
ALPHABETS = ["ACGT", "ACGTN", "ACGTNRYacgtn", "ACGTXNSWKMBDHVU"]
TRUSEQ = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"


def rs(rng, n, alphabet):
    return "".join(rng.choice(alphabet) for _ in range(n))


def mutate(rng, s, alphabet, n_edits):
    s = list(s)
    for _ in range(n_edits):
        if not s:
            break
        p = rng.randrange(len(s))
        op = rng.randint(0, 2)
        if op == 0:
            s[p] = rng.choice(alphabet)
        elif op == 1:
            s.insert(p, rng.choice(alphabet))
        else:
            del s[p]
    return "".join(s)


def read_with_adapter(rng, adapter, alphabet, n):
    """random read that (usually) carries a mutated piece of the adapter"""
    if n == 0:
        return ""
    m = len(adapter)
    if rng.random() < 0.65:
        a = rng.randint(0, m - 1)
        piece = mutate(rng, adapter[a:a + rng.randint(1, m)], alphabet, rng.randint(0, 3))
        read = rs(rng, rng.randint(0, n), alphabet) + piece + rs(rng, rng.randint(0, 20), alphabet)
        return read[:n] if rng.random() < 0.5 else read
    return rs(rng, n, alphabet)


def gen_locate(ref, rng, count):
    cases = []
    while len(cases) < count:
        al = rng.choice(ALPHABETS)
        m = rng.randint(1, 64)
        adapter = rs(rng, m, al)
        rate = rng.choice([0, 0.05, 0.1, 0.1, 0.2, 0.3, 0.5, 1.0, round(rng.random(), 3)])
        flags = rng.randint(0, 15)
        wr, wq = rng.random() < 0.3, rng.random() < 0.3
        ic = rng.choice([1, 1, 1, 2, 100000])
        mo = rng.randint(1, min(m, 6))
        try:
            aligner = ref.Aligner(adapter, rate, flags, wr, wq, ic, mo)
        except ValueError:
            continue
        for _ in range(3):
            query = read_with_adapter(rng, adapter, al, rng.randint(0, 160))
            res = aligner.locate(query)
            cases.append({"ref": adapter, "rate": rate, "flags": flags, "wr": wr, "wq": wq,
                          "indel_cost": ic, "min_overlap": mo, "query": query,
                          "result": list(res) if res is not None else None,
                          "effective_length": aligner.effective_length})
    return cases


def gen_truseq(ref, rng, count):
    """the benchmark configuration: TruSeq 3' adapter, e=0.1, O=3, 150 bp reads"""
    aligner = ref.Aligner(TRUSEQ, 0.1, 14, False, False, 1, 3)
    cases = []
    for _ in range(count):
        read = rs(rng, 150, "ACGT")
        if rng.random() < 0.6:
            pos = rng.randint(0, 150)
            copy = mutate(rng, TRUSEQ, "ACGT", rng.choice([0, 0, 0, 1, 1, 2, 3, 4]))
            read = (read[:pos] + copy + read[pos:])[:150]
        if rng.random() < 0.3:
            read = "".join(c if rng.random() > 0.01 else "N" for c in read)
        res = aligner.locate(read)
        cases.append({"query": read, "result": list(res) if res is not None else None})
    return {"ref": TRUSEQ, "rate": 0.1, "flags": 14, "min_overlap": 3, "cases": cases}


def gen_comparers(ref, rng, count):
    cases = []
    while len(cases) < count:
        al = rng.choice(ALPHABETS)
        m = rng.randint(1, 40)
        adapter = rs(rng, m, al)
        rate = rng.choice([0, 0.1, 0.2, 0.4, 0.9, 1.0])
        wr, wq = rng.random() < 0.4, rng.random() < 0.4
        mo = rng.randint(1, 5)
        n = rng.randint(0, 60)
        if rng.random() < 0.7:
            read = adapter[:n] if rng.random() < 0.5 else (rs(rng, max(0, n - m), al) + adapter)
            read = "".join(c if rng.random() > 0.1 else rng.choice(al) for c in read)
        else:
            read = rs(rng, n, al)
        for name, cls in (("prefix", ref.PrefixComparer), ("suffix", ref.SuffixComparer)):
            try:
                cmp_ = cls(adapter, rate, wr, wq, mo)
            except ValueError:
                continue
            res = cmp_.locate(read)
            cases.append({"kind": name, "ref": adapter, "rate": rate, "wr": wr, "wq": wq,
                          "min_overlap": mo, "query": read,
                          "result": list(res) if res is not None else None,
                          "effective_length": cmp_.effective_length})
    return cases


def gen_kmers(ref, rng, count):
    cases = []
    while len(cases) < count:
        al = rng.choice(ALPHABETS)
        sets = []
        for _ in range(rng.randint(1, 5)):
            kind = rng.randint(0, 2)
            if kind == 0:
                st, sp = -rng.randint(1, 40), None
            elif kind == 1:
                st, sp = 0, rng.randint(1, 40)
            else:
                st, sp = 0, None
            sets.append([st, sp, [rs(rng, rng.randint(1, 24), al) for _ in range(rng.randint(1, 6))]])
        wr, wq = rng.random() < 0.4, rng.random() < 0.4
        finder = ref.KmerFinder([(a, b, c) for a, b, c in sets], wr, wq)
        reads = []
        for _ in range(6):
            n = rng.randint(0, 100)
            read = rs(rng, n, al)
            if rng.random() < 0.5 and n:
                km = rng.choice(rng.choice(sets)[2])
                p = rng.randint(0, n)
                read = (read[:p] + km + read[p:])[:max(n, len(km))]
            # the reference reads out of bounds when stop > len(read): keep to defined inputs
            if any(sp is not None and sp > len(read) for _, sp, _ in sets):
                continue
            reads.append([read, bool(finder.kmers_present(read))])
        if reads:
            cases.append({"sets": sets, "wr": wr, "wq": wq, "reads": reads})
    return cases


def gen_heuristic(ref, rng, count):
    cases = []
    adapters = [TRUSEQ, "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT", "GCCGAACTTCTTAGACTGCCTTAAGGACGT", "AAAAAAAAAA"]
    while len(cases) < count:
        if adapters:
            ad = adapters.pop()
        else:
            ad = rs(rng, rng.randint(1, 70), "ACGT" if rng.random() < 0.8 else "ACGTN")
        mo = rng.randint(1, min(len(ad), 10))
        er = rng.choice([0, 0.05, 0.1, 0.15, 0.2, 0.33, 0.5])
        for back, front, internal in [(1, 0, 1), (0, 1, 1), (1, 1, 1), (1, 0, 0), (0, 1, 0)]:
            res = ref.create_positions_and_kmers(ad, mo, er, bool(back), bool(front), bool(internal))
            norm = sorted([[s, e, sorted(k)] for s, e, k in res], key=lambda x: (x[0], -1 if x[1] is None else x[1]))
            cases.append({"adapter": ad, "min_overlap": mo, "error_rate": er, "back": bool(back),
                          "front": bool(front), "internal": bool(internal), "result": norm})
    return cases


ADAPTER_CLASSES = ["FrontAdapter", "BackAdapter", "AnywhereAdapter", "NonInternalFrontAdapter",
                   "NonInternalBackAdapter", "PrefixAdapter", "SuffixAdapter",
                   "RightmostFrontAdapter", "RightmostBackAdapter"]


class _NeverPresent:
    def kmers_present(self, sequence):
        return False


def safe_match_to(ref, adapter, read):
    """adapter.match_to(read) restricted to DEFINED behaviour of the reference: its
    kmers_present reads past the end of the string when a search window has a positive stop
    beyond len(read) (reference _kmer_finder.pyx:196-205; front-adapter sets use positive
    stops, kmer_heuristic.py:159) and the result then depends on heap garbage.  For such
    reads the windows are clamped to the read first -- what the memory-safe reading of the
    algorithm gives and what the HIP kernel and the oracle implement."""
    kf = getattr(adapter, "kmer_finder", None)
    sets = getattr(kf, "positions_and_kmers", None)
    n = len(read)
    if sets is not None and any(stop is not None and stop > n for _, stop, _ in sets):
        clamped = [(s, (min(e, n) if e is not None else None), k) for s, e, k in sets]
        clamped = [(s, e, k) for s, e, k in clamped if e is None or e > 0]   # empty windows never match
        adapter.kmer_finder = ref.KmerFinder(clamped, kf.ref_wildcards, kf.query_wildcards) \
            if clamped else _NeverPresent()
        try:
            return adapter.match_to(read)
        finally:
            adapter.kmer_finder = kf
    return adapter.match_to(read)


def safe_linked_match_to(ref, linked, read):
    """LinkedAdapter.match_to (reference adapters.py:1215-1227) with safe_match_to stages"""
    front = safe_match_to(ref, linked.front_adapter, read)
    if linked.front_required and front is None:
        return None, None, False
    seq = read[front.trim_slice()] if front is not None else read
    back = safe_match_to(ref, linked.back_adapter, seq)
    if back is None and (linked.back_required or front is None):
        return None, None, False
    return front, back, True


def match_to_json(match):
    if match is None:
        return None
    return {"cls": type(match).__name__,
            "t": [match.astart, match.astop, match.rstart, match.rstop, match.score, match.errors]}


def gen_adapters(ref, rng, count):
    """match_to() of every adapter class (reference adapters.py:684-1089)"""
    ad_mod = ref.adapters
    cases = []
    while len(cases) < count:
        cls_name = rng.choice(ADAPTER_CLASSES)
        al = rng.choice(["ACGT", "ACGT", "ACGTN", "ACGTRYN"])
        m = rng.randint(3, 40)
        seq = rs(rng, m, al)
        kwargs = {"max_errors": rng.choice([0, 0.1, 0.1, 0.2, 0.3, 2]),
                  "min_overlap": rng.randint(1, 8),
                  "read_wildcards": rng.random() < 0.3,
                  "adapter_wildcards": rng.random() < 0.8,
                  "indels": rng.random() < 0.7}
        try:
            adapter = getattr(ad_mod, cls_name)(seq, **kwargs)
        except Exception:
            continue
        reads = []
        for _ in range(4):
            n = rng.randint(0, 120)
            read = read_with_adapter(rng, adapter.sequence, "ACGTN" if rng.random() < 0.3 else "ACGT", n)
            if rng.random() < 0.2:
                read = read.lower()
            reads.append([read, match_to_json(safe_match_to(ref, adapter, read))])
        cases.append({"cls": cls_name, "sequence": seq, "kwargs": kwargs, "reads": reads})
    return cases


def gen_linked_and_multiple(ref, rng):
    ad_mod = ref.adapters
    out = {"linked": [], "multiple": []}
    for _ in range(60):
        fseq, bseq = rs(rng, rng.randint(5, 16), "ACGT"), rs(rng, rng.randint(8, 33), "ACGT")
        anchored = rng.random() < 0.5
        freq, breq = rng.random() < 0.7, rng.random() < 0.5
        fcls = "PrefixAdapter" if anchored else "FrontAdapter"
        front = getattr(ad_mod, fcls)(fseq, max_errors=0.1)
        back = ad_mod.BackAdapter(bseq, max_errors=0.1, min_overlap=3)
        linked = ad_mod.LinkedAdapter(front, back, front_required=freq, back_required=breq, name="linked")
        reads = []
        for _ in range(6):
            core = rs(rng, rng.randint(0, 60), "ACGT")
            f = mutate(rng, fseq, "ACGT", rng.choice([0, 0, 1])) if rng.random() < 0.7 else ""
            b = mutate(rng, bseq, "ACGT", rng.choice([0, 0, 1, 2])) if rng.random() < 0.7 else ""
            pre = "" if anchored else rs(rng, rng.randint(0, 5), "ACGT")
            read = pre + f + core + b + rs(rng, rng.randint(0, 10), "ACGT")
            fm, bm, ok = safe_linked_match_to(ref, linked, read)
            reads.append([read, None if not ok else
                          {"front": match_to_json(fm), "back": match_to_json(bm)}])
        out["linked"].append({"front_cls": fcls, "front": fseq, "back": bseq, "front_required": freq,
                              "back_required": breq, "reads": reads})
    for _ in range(40):
        seqs = [rs(rng, rng.randint(10, 33), "ACGT") for _ in range(rng.randint(2, 6))]
        if rng.random() < 0.5:
            seqs.append(seqs[0][:-2] + rs(rng, 2, "ACGT"))     # near-duplicates -> ties
        adapters = [ad_mod.BackAdapter(s, max_errors=0.15, min_overlap=3) for s in seqs]
        multi = ad_mod.MultipleAdapters(adapters)
        reads = []
        for _ in range(8):
            which = rng.choice(seqs)
            read = rs(rng, rng.randint(10, 80), "ACGT") + mutate(rng, which, "ACGT", rng.choice([0, 1, 2]))
            read = read[:rng.randint(20, len(read))] + rs(rng, rng.randint(0, 10), "ACGT")
            mt = multi.match_to(read)
            reads.append([read, None if mt is None else
                          {"adapter": adapters.index(mt.adapter), "m": match_to_json(mt)}])
        out["multiple"].append({"seqs": seqs, "reads": reads})
    return out


def gen_illumina_info(ref):
    """reference tests/cut/illumina.info.txt: (errors, rstart, rstop) per read of
    tests/data/illumina.fastq.gz for -a GCCGAACTTCTTAGACTGCCTTAAGGACGT"""
    fastq = os.path.join(REFERENCE, "tests", "data", "illumina.fastq.gz")
    info = os.path.join(REFERENCE, "tests", "cut", "illumina.info.txt")
    with gzip.open(fastq, "rt") as f:
        lines = f.read().split("\n")
    seqs = [lines[i + 1] for i in range(0, len(lines) - 1, 4) if lines[i].startswith("@")]
    expected = []
    with open(info) as f:
        for line in f:
            fields = line.rstrip("\n").split("\t")
            if fields[1] == "-1":
                expected.append(None)
            else:
                expected.append([int(fields[1]), int(fields[2]), int(fields[3])])
    assert len(seqs) == len(expected), (len(seqs), len(expected))
    # sanity: the compiled reference reproduces its own golden file
    adapter = ref.adapters.BackAdapter("GCCGAACTTCTTAGACTGCCTTAAGGACGT", max_errors=0.1, min_overlap=3)
    for s, e in zip(seqs, expected):
        mt = adapter.match_to(s)
        got = None if mt is None else [mt.errors, mt.rstart, mt.rstop]
        assert got == e, (s, got, e)
    return {"adapter": "GCCGAACTTCTTAGACTGCCTTAAGGACGT", "max_errors": 0.1, "min_overlap": 3,
            "reads": seqs, "expected": expected}


def main():
    assert build_ref.build(verbose=False), "oracle/_ref could not be built"
    ref = ref_loader.load()
    assert ref is not None
    rng = random.Random(20260924)
    out = {
        "locate.json": gen_locate(ref, rng, 2400),
        "truseq.json": gen_truseq(ref, rng, 1500),
        "comparers.json": gen_comparers(ref, rng, 800),
        "kmers.json": gen_kmers(ref, rng, 300),
        "heuristic.json": gen_heuristic(ref, rng, 300),
        "adapters.json": gen_adapters(ref, rng, 500),
        "linked_multiple.json": gen_linked_and_multiple(ref, rng),
        "illumina_info.json": gen_illumina_info(ref),
    }
    for name, data in out.items():
        path = os.path.join(HERE, name)
        with open(path, "w") as f:
            json.dump(data, f, separators=(",", ":"))
        print(f"wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
