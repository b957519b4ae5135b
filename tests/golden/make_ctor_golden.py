#!/usr/bin/env python3
"""Golden vectors for the CONSTRUCTORS of Aligner / PrefixComparer / SuffixComparer (reference _align.pyx:200-277, :607-640),
generated from the reference's compiled classes (build container only):

    python tests/golden/make_ctor_golden.py        ->  tests/golden/ctors.json

Random parameter sets, valid and not: the exception class the reference raises, or repr(), effective_length (comparers) and
the pickle arguments.  tests/test_ctor_golden.py replays them on cutadapt_amd.align (no GPU: plans are built on the host)."""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import build_ref, ref_loader  # noqa: E402


def rs(rng, n, alphabet):
    return "".join(rng.choice(alphabet) for _ in range(n))


def main():
    assert build_ref.build(verbose=False), "oracle/_ref could not be built"
    R = ref_loader.load()
    rng = random.Random(161803)
    cases = []
    for i in range(500):
        kind = rng.choice(["Aligner", "Aligner", "PrefixComparer", "SuffixComparer"])
        m = rng.choice([0, 1, 3, 8, 20, 33, 64, 70])
        seq = rs(rng, m, rng.choice(["ACGT", "ACGT", "ACGTN", "N", "acgtn", "ACGTRYXKM", "ACGé"]))
        rate = rng.choice([0, 0.0, 0.1, 0.25, 0.5, 1, 1.0, 1.5, -0.1, 2])
        wr, wq = rng.random() < 0.4, rng.random() < 0.3
        ov = rng.choice([1, 1, 3, 10, 0, -1, 100])
        if kind == "Aligner":
            args = [seq, rate, rng.choice([0, 2, 8, 11, 14, 15, 9, 6, 31]), wr, wq, rng.choice([1, 1, 2, 100000, 0, -3]), ov]
        else:
            args = [seq, rate, wr, wq, ov]
        case = {"cls": kind, "args": args}
        try:
            obj = getattr(R, kind)(*args)
        except Exception as exc:
            case["error"] = type(exc).__name__
        else:
            case["repr"] = repr(obj)
            if kind == "Aligner":                  # (the comparers pickle through Cython's generated helper: internal)
                red = obj.__reduce__()
                case["reduce"] = [red[0].__name__, list(red[1])]
            else:
                case["effective_length"] = obj.effective_length
        cases.append(case)
    path = os.path.join(HERE, "ctors.json")
    with open(path, "w") as f:
        json.dump(cases, f, indent=0, ensure_ascii=True)
    print("wrote", path, os.path.getsize(path), "bytes;", sum("error" in c for c in cases), "error cases of", len(cases))


if __name__ == "__main__":
    main()
