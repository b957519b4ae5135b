#!/usr/bin/env python3
"""Golden vectors for what the nine adapter classes ARE before any read is matched (reference adapters.py:496-1089),
generated from the REFERENCE itself (build container only: /root/reference + oracle/_ref):

    python tests/golden/make_adapter_attr_golden.py        ->  tests/golden/adapter_attrs.json

Per class and parameter set: the normalised sequence / error rate / overlap, description, spec(), descriptive_identifier(),
repr, effective length, which aligner class and flags it builds, and the k-mer search sets of its prefilter -- including
the `force_anywhere` parts of linked adapters and the anchored classes without indels (comparers, no prefilter).
tests/test_adapter_classes.py replays them on cutadapt_amd.adapters (no GPU: construction is host work)."""
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
FORCEABLE = {"FrontAdapter", "RightmostFrontAdapter", "BackAdapter", "RightmostBackAdapter",
             "NonInternalFrontAdapter", "NonInternalBackAdapter", "PrefixAdapter", "SuffixAdapter"}


def rs(rng, n, alphabet):
    return "".join(rng.choice(alphabet) for _ in range(n))


def describe(ad):
    al = ad.aligner
    kf = ad.kmer_finder
    out = {
        "sequence": ad.sequence, "max_error_rate": ad.max_error_rate, "min_overlap": ad.min_overlap,
        "read_wildcards": bool(ad.read_wildcards), "adapter_wildcards": bool(ad.adapter_wildcards), "indels": bool(ad.indels),
        "description": ad.description, "spec": ad.spec(), "identifier": ad.descriptive_identifier(),
        "repr": repr(ad), "len": len(ad), "effective_length": ad.effective_length,
        "allows_partial_matches": bool(ad.allows_partial_matches),
        "aligner": type(al).__name__, "finder": type(kf).__name__,
    }
    if type(al).__name__ == "Aligner":
        # Aligner.__reduce__: (reference, max_error_rate, flags, wildcard_ref, wildcard_query, indel_cost, min_overlap) --
        # the ends the aligner may skip (force_anywhere frees them for the regular and rightmost types only)
        out["aligner_args"] = list(al.__reduce__()[1])
    if type(kf).__name__ == "KmerFinder":
        out["kmer_sets"] = [[a, b, sorted(k)] for a, b, k in kf.positions_and_kmers]
        out["kmer_wildcards"] = [bool(kf.ref_wildcards), bool(kf.query_wildcards)]
    return out


def main():
    assert build_ref.build(verbose=False), "oracle/_ref could not be built"
    R = ref_loader.load().adapters
    rng = random.Random(314159)
    cases = []
    for cls in CLASSES:
        for i in range(14):
            m = rng.choice([5, 8, 12, 20, 33, 40])
            seq = rs(rng, m, rng.choice(["ACGT", "ACGT", "ACGTN", "acgtuiN", "ACGTRYKM"]))
            if set(seq.upper()) <= set("NI"):
                seq = "A" + seq[1:]
            kwargs = {"max_errors": rng.choice([0, 0.1, 0.1, 0.2, 1, 2]), "min_overlap": rng.choice([1, 3, 3, 5, 50]),
                      "read_wildcards": rng.random() < 0.3, "indels": rng.random() < 0.7, "name": f"n{i}"}
            if rng.random() < 0.5:
                kwargs["adapter_wildcards"] = rng.random() < 0.5
            if cls in FORCEABLE and rng.random() < 0.45:
                kwargs["force_anywhere"] = True
            try:
                ad = getattr(R, cls)(seq, **kwargs)
            except Exception as exc:            # the error a class raises for a parameter set is part of its behaviour
                cases.append({"cls": cls, "seq": seq, "kwargs": kwargs, "error": type(exc).__name__})
                continue
            want = describe(ad)
            # which adapters the index of anchored adapters takes (reference :1373-1392)
            want["indexable"] = [bool(R.AdapterIndex.is_acceptable(ad, True)), bool(R.AdapterIndex.is_acceptable(ad, False))]
            cases.append({"cls": cls, "seq": seq, "kwargs": kwargs, "want": want})
    for cls in ("PrefixAdapter", "SuffixAdapter"):
        for i in range(24):
            seq = rs(rng, rng.choice([6, 10, 20, 33, 50]), rng.choice(["ACGT", "ACGT", "ACGTN"]))
            kwargs = {"max_errors": rng.choice([0, 0.05, 0.1, 0.2, 1, 3, 4]), "read_wildcards": rng.random() < 0.2,
                      "adapter_wildcards": rng.random() < 0.4, "indels": rng.random() < 0.7, "name": f"x{i}"}
            ad = getattr(R, cls)(seq, **kwargs)
            want = describe(ad)
            want["indexable"] = [bool(R.AdapterIndex.is_acceptable(ad, True)), bool(R.AdapterIndex.is_acceptable(ad, False))]
            cases.append({"cls": cls, "seq": seq, "kwargs": kwargs, "want": want})
    path = os.path.join(HERE, "adapter_attrs.json")
    with open(path, "w") as f:
        json.dump(cases, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes;", sum("error" in c for c in cases), "error cases of", len(cases))


if __name__ == "__main__":
    main()
