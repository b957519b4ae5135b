#!/usr/bin/env python3
"""Golden vectors for quality / NextSeq / poly-A trimming and expected errors,
generated with the reference's own qualtrim.pyx (built by oracle/build_ref.py).  Run in the build
container; writes tests/golden/qualtrim.json.  expected_errors values are stored as float.hex()."""
import json
import os
import random
import sys
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import ref_loader  # noqa: E402

q = ref_loader.load().qualtrim
rng = random.Random(77)


def rqual(n, lo=0, hi=41, base=33):
    return "".join(chr(base + rng.randint(lo, hi)) for _ in range(n))


def rseq(n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


out = {"quality_trim": [], "nextseq": [], "poly_a": [], "expected_errors": []}
for _ in range(400):
    n = rng.choice([0, 1, 2, 5, 20, 50, 100, 150])
    base = rng.choice([33, 33, 64])
    style = rng.random()
    if style < 0.4:                                   # good middle, bad ends
        a, b = rng.randint(0, n), rng.randint(0, n)
        a, b = min(a, b), max(a, b)
        qual = rqual(a, 0, 12, base) + rqual(b - a, 25, 40, base) + rqual(n - b, 0, 12, base)
    else:
        qual = rqual(n, 0, 40, base)
    cf, cb = rng.choice([0, 0, 5, 10, 20, 30]), rng.choice([0, 5, 10, 20, 30])
    out["quality_trim"].append([qual, cf, cb, base, list(q.quality_trim_index(qual, cf, cb, base))])
for _ in range(300):
    n = rng.choice([0, 1, 3, 20, 65, 150])
    seq = rseq(n - min(n, rng.randint(0, 30))) if n else ""
    seq = seq + "G" * (n - len(seq))
    if rng.random() < 0.5:
        seq = "".join(c if rng.random() > 0.1 else rng.choice("ACTN") for c in seq)
    qual = rqual(n, 0, 40)
    cutoff = rng.choice([10, 20, 22, 30])
    rec = types.SimpleNamespace(sequence=seq, qualities=qual)
    out["nextseq"].append([seq, qual, cutoff, 33, q.nextseq_trim_index(rec, cutoff, 33)])
for _ in range(400):
    body = rseq(rng.choice([0, 3, 12, 40, 100]))
    tail = "".join("A" if rng.random() > rng.choice([0.0, 0.05, 0.2, 0.4]) else rng.choice("CGT") for _ in range(rng.choice([0, 2, 3, 8, 30, 60])))
    revcomp = rng.random() < 0.5
    s = (tail.replace("A", "t").replace("T", "A").replace("t", "T") + body) if revcomp else body + tail
    out["poly_a"].append([s, revcomp, q.poly_a_trim_index(s, revcomp)])
for _ in range(300):
    n = rng.choice([0, 1, 2, 3, 4, 5, 7, 50, 150, 151])
    qual = rqual(n, 0, rng.choice([10, 41, 93]))
    out["expected_errors"].append([qual, 33, q.expected_errors(qual, 33).hex()])
out["expected_errors"].append(["".join(chr(33 + i) for i in range(94)), 33, q.expected_errors("".join(chr(33 + i) for i in range(94))).hex()])
out["error_table"] = [q.expected_errors(chr(33 + i)).hex() for i in range(94)]
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qualtrim.json")
with open(path, "w") as f:
    json.dump(out, f)
print("wrote", path, os.path.getsize(path), "bytes", {k: len(v) for k, v in out.items()})
