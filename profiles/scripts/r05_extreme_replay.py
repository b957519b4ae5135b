#!/usr/bin/env python3
"""tests/golden/adapters_extreme.json (reference results at the edges of the parameter space) through the batch API on the GPU:
prints the number of reads compared and the first mismatches.  PYTHONPATH=. python profiles/scripts/r05_extreme_replay.py"""
import json
import sys
from cutadapt_amd import adapters as A
from cutadapt_amd.batch import ReadBatch

cases = json.load(open("tests/golden/adapters_extreme.json"))
limit = int(sys.argv[1]) if len(sys.argv) > 1 else len(cases)
n = bad = 0
for c in cases[:limit]:
    ad = getattr(A, c["cls"])(c["sequence"], **c["kwargs"])
    reads = [r for r, _ in c["reads"]]
    bm = ad.match_to_batch(ReadBatch.from_strings(reads))
    for i, (read, want) in enumerate(c["reads"]):
        mt = bm.match(i)
        got = None if mt is None else {"cls": type(mt).__name__, "t": list(mt.astuple())}
        n += 1
        if got != want:
            bad += 1
            if bad <= 5:
                print("MISMATCH", c["cls"], c["sequence"], c["kwargs"], repr(read), got, want)
print("reads", n, "mismatches", bad)
