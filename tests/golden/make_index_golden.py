#!/usr/bin/env python3
"""Golden vectors for the AdapterIndex path, generated with the reference's own
AdapterIndex / IndexedPrefixAdapters / IndexedSuffixAdapters (adapters.py:1289-1567) as built by
oracle/build_ref.py.  Run in the build container; writes tests/golden/index.json.

Per adapter set: the index size, its string lengths, the number of ambiguous strings, a SHA-256 over
the sorted "string adapter errors matches" lines of the whole dictionary, a sample of entries, and
the match (adapter, astart, astop, rstart, rstop, score, errors) of every read of a small read set
that covers exact hits, 1-3 errors, indels, 'N' in the affix, lower case, reads shorter than the
indexed lengths, empty reads and foreign characters.
"""
import hashlib
import json
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import ref_loader  # noqa: E402

ref = ref_loader.load()
RA = ref.adapters
rng = random.Random(20260924)


def rnd(n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def mutate(s, n_edits):
    s = list(s)
    for _ in range(n_edits):
        op = rng.choice("sid")
        p = rng.randrange(len(s) + 1) if s else 0
        if op == "s" and s:
            p = min(p, len(s) - 1)
            s[p] = rng.choice("ACGTN")
        elif op == "i":
            s.insert(p, rng.choice("ACGTN"))
        elif s:
            del s[min(p, len(s) - 1)]
    return "".join(s)


SETS = [
    # name, prefix?, [(sequence, max_errors, indels)]
    ("barcodes8_e1", True, [(rnd(8), 0.125, True) for _ in range(12)]),
    ("barcodes8_hamming", True, [(rnd(8), 0.25, False) for _ in range(12)]),
    ("mixed_lengths_prefix", True, [(rnd(n), 0.2, n % 2 == 0) for n in (6, 7, 9, 10, 12, 12, 15)]),
    ("mixed_lengths_suffix", False, [(rnd(n), 0.2, n % 2 == 1) for n in (6, 7, 9, 10, 12, 12, 15)]),
    ("suffix10_e2", False, [(rnd(10), 0.2, True) for _ in range(6)]),
    ("similar_ambiguous", True, [("ACGTACGTAC", 0.2, True), ("ACGTACGTAG", 0.2, True), ("ACGTTCGTAC", 0.2, False)]),
    ("long33_e3", True, [(rnd(33), 0.1, False), (rnd(33), 0.1, False)]),
    ("exact_only", False, [(rnd(5), 0.0, True), (rnd(7), 0.0, False), (rnd(5), 0.0, True)]),
]

out = []
for name, prefix, specs in SETS:
    cls = RA.PrefixAdapter if prefix else RA.SuffixAdapter
    ads = [cls(s, max_errors=e, indels=i, name=f"a{j}") for j, (s, e, i) in enumerate(specs)]
    index = RA.AdapterIndex(ads, prefix=prefix)
    lines = sorted(f"{s} {ads.index(a)} {e} {m}" for s, (a, e, m) in index._index.items())
    digest = hashlib.sha256("\n".join(lines).encode()).hexdigest()
    sample = [lines[i] for i in sorted(rng.sample(range(len(lines)), min(40, len(lines))))]
    matcher = (RA.IndexedPrefixAdapters if prefix else RA.IndexedSuffixAdapters)(ads)
    reads = ["", "A", "N", "NNNNNNNNNNNN", "ACGU", "acgtacgtacgtacgtacgt", "ACGT ACGT", "RYKMACGTACGT"]
    for s, _, _ in specs:
        for n_edits in (0, 0, 1, 1, 2, 3):
            core = mutate(s, n_edits)
            pad = rnd(rng.randint(0, 12))
            read = core + pad if prefix else pad + core
            if rng.random() < 0.2:
                read = read.lower()
            if rng.random() < 0.15:
                read = read[:rng.randint(0, len(read))] if prefix else read[rng.randint(0, len(read)):]
            reads.append(read)
    reads += [rnd(rng.randint(0, 25)) for _ in range(40)]
    results = []
    for r in reads:
        m = matcher.match_to(r)
        results.append(None if m is None else [ads.index(m.adapter), m.astart, m.astop, m.rstart, m.rstop, m.score, m.errors])
    out.append({"name": name, "prefix": prefix, "adapters": [list(s) for s in specs], "n_strings": len(index._index),
                "lengths": index._lengths, "n_ambiguous": index._ambiguous, "sha256": digest, "sample": sample,
                "reads": reads, "results": results})
    print(name, len(index._index), "strings,", sum(r is not None for r in results), "of", len(reads), "reads match")

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "index.json")
with open(path, "w") as f:
    json.dump(out, f, indent=0)
print("wrote", path, os.path.getsize(path), "bytes")
