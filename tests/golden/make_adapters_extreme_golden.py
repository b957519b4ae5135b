#!/usr/bin/env python3
"""Golden vectors for Adapter.match_to at the EDGES of the parameter space, generated from the REFERENCE's adapter classes
(build container only):

    python tests/golden/make_adapters_extreme_golden.py        ->  tests/golden/adapters_extreme.json

All nine classes (the rightmost ones included), adapters of 1 .. 33 characters, error rates up to 1.0 (as many errors as
characters), min_overlap from 1 to beyond the adapter, `force_anywhere` on every class that takes it, reads of only N, empty
reads, reads that hold pieces of the adapter.  One thing is avoided: a read shorter than a positive `stop` of the adapter's
k-mer search sets -- the reference's KmerFinder then reads behind the string (undefined behaviour; this package and its
oracle clamp the window).  tests/test_match_to_host_logic.py replays the file on cutadapt_amd.adapters with the ORACLE in
the kernels' place (no GPU: what is checked is the host logic around them -- classes, flags, search sets, reversal)."""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import build_ref, ref_loader  # noqa: E402

CLASSES = ["FrontAdapter", "RightmostFrontAdapter", "BackAdapter", "RightmostBackAdapter", "AnywhereAdapter",
           "NonInternalFrontAdapter", "NonInternalBackAdapter", "PrefixAdapter", "SuffixAdapter"]


def rs(rng, n, alphabet):
    return "".join(rng.choice(alphabet) for _ in range(n))


def main():
    assert build_ref.build(verbose=False), "oracle/_ref could not be built"
    R = ref_loader.load().adapters
    rng = random.Random(57721)
    cases = []
    while len(cases) < 260:
        cls = rng.choice(CLASSES)
        m = rng.choice([1, 2, 3, 4, 6, 10, 20, 33])
        seq = rs(rng, m, rng.choice(["ACGT", "ACGT", "ACGTN", "ACGTRY"]))
        if set(seq) <= set("N"):
            seq = "A" + seq[1:]
        kwargs = {"max_errors": rng.choice([0, 0.1, 0.25, 0.5, 0.9, 1, 2, m]), "min_overlap": rng.choice([1, 2, 3, m, m + 3]),
                  "read_wildcards": rng.random() < 0.3, "indels": rng.random() < 0.7}
        if cls != "AnywhereAdapter" and rng.random() < 0.35:
            kwargs["force_anywhere"] = True
        try:
            ad = getattr(R, cls)(seq, **kwargs)
        except ValueError:
            continue
        sets = getattr(ad.kmer_finder, "positions_and_kmers", [])
        floor = max([stop for _, stop, _ in sets if stop is not None and stop > 0] + [0])
        reads = []
        for _ in range(9):
            reads.append(rs(rng, rng.randint(floor, floor + 40), rng.choice(["ACGT", "ACGTN", "N", "ACGTNacgt"])))
        for _ in range(5):
            r = rs(rng, rng.randint(0, 10), "ACGT") + seq[:rng.randint(1, m)] + rs(rng, rng.randint(0, 10), "ACGT")
            reads.append(r + rs(rng, max(0, floor - len(r)), "ACGT"))
        if floor == 0:
            reads.append("")
        out = []
        for r in reads:
            mt = ad.match_to(r)
            out.append([r, None if mt is None else {"cls": type(mt).__name__,
                                                    "t": [mt.astart, mt.astop, mt.rstart, mt.rstop, mt.score, mt.errors]}])
        cases.append({"cls": cls, "sequence": seq, "kwargs": kwargs, "reads": out})
    path = os.path.join(HERE, "adapters_extreme.json")
    with open(path, "w") as f:
        json.dump(cases, f, indent=0)
    n = sum(len(c["reads"]) for c in cases)
    print("wrote", path, os.path.getsize(path), "bytes;", n, "reads,", sum(w is not None for c in cases for _, w in c["reads"]), "matches")


if __name__ == "__main__":
    main()
