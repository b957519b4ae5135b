#!/usr/bin/env python3
"""Golden vectors for the Match objects of the adapter API (reference adapters.py:292-493, :1092-1153, :1588-1602),
generated from the REFERENCE itself (build container only: /root/reference + oracle/_ref, see oracle/build_ref.py):

    python tests/golden/make_match_golden.py        ->  tests/golden/matches.json

For random alignments (astart, astop, rstart, rstop, score, errors) on random reads: everything a modifier or the
report asks of RemoveBeforeMatch / RemoveAfterMatch / LinkedMatch -- intervals, slices, the trimmed read, info rows,
wildcards, repr.  tests/test_match_objects.py replays them on cutadapt_amd.adapters (no GPU: the objects are plain data).
"""
import json
import os
import random
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import build_ref, ref_loader  # noqa: E402


def rs(rng, n, alphabet="ACGTN"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def single_case(rng, R, before):
    n = rng.randint(0, 40)
    seq = rs(rng, n)
    m = rng.randint(1, 20)
    ad_seq = "A" + rs(rng, m - 1, "ACGTNN")              # (an adapter of N only is refused by the Aligner)
    rstart = rng.randint(0, n)
    rstop = rng.randint(rstart, n)
    astart = rng.randint(0, m)
    astop = min(m, astart + (rstop - rstart) if rng.random() < 0.7 else rng.randint(astart, m))
    score, errors = rng.randint(0, 20), rng.randint(0, 3)
    cls = R.RemoveBeforeMatch if before else R.RemoveAfterMatch
    adapter = (R.FrontAdapter if before else R.BackAdapter)(ad_seq, name="ad")
    match = cls(astart, astop, rstart, rstop, score, errors, adapter, seq)
    quals = "".join(chr(33 + rng.randint(0, 40)) for _ in range(n)) if rng.random() < 0.7 else None
    read = types.SimpleNamespace(sequence=seq, qualities=quals)
    sl = match.trim_slice()
    out = {
        "before": before, "adapter": ad_seq, "sequence": seq, "qualities": quals,
        "tuple": [astart, astop, rstart, rstop, score, errors],
        "repr": repr(match), "length": match.length,
        "remainder_interval": list(match.remainder_interval()),
        "retained_adapter_interval": list(match.retained_adapter_interval()),
        "trim_slice": [sl.start, sl.stop, sl.step], "trimmed": match.trimmed(seq), "rest": match.rest(),
        "removed_sequence_length": match.removed_sequence_length(), "match_sequence": match.match_sequence(),
        "wildcards": match.wildcards(), "info": match.get_info_records(read),
    }
    if not before:
        out["adjacent_base"] = match.adjacent_base()
    return out


def linked_case(rng, R):
    seq = rs(rng, rng.randint(6, 50))
    fa, ba = R.FrontAdapter(rs(rng, 5, "ACGT"), name="f"), R.BackAdapter(rs(rng, 5, "ACGT"), name="b")
    la = R.LinkedAdapter(fa, ba, front_required=True, back_required=False, name="la")
    front = back = None
    rest = seq
    spec = {}
    if rng.random() < 0.75:
        a = rng.randint(0, len(seq)); b = rng.randint(a, len(seq))
        t = [0, min(5, b - a), a, b, rng.randint(0, 9), rng.randint(0, 2)]
        front = R.RemoveBeforeMatch(*t, fa, seq)
        rest = seq[b:]
        spec["front"] = t
    if front is None or rng.random() < 0.7:
        a = rng.randint(0, len(rest)); b = rng.randint(a, len(rest))
        t = [0, min(5, b - a), a, b, rng.randint(0, 9), rng.randint(0, 2)]
        back = R.RemoveAfterMatch(*t, ba, rest)
        spec["back"] = t
    lm = R.LinkedMatch(front, back, la)
    spec.update({
        "sequence": seq, "score": lm.score, "errors": lm.errors, "trimmed": lm.trimmed(seq),
        "remainder_interval": list(lm.remainder_interval()),
        "retained_adapter_interval": list(lm.retained_adapter_interval()),
        "match_sequence": lm.match_sequence(),
        "remainder_of_parts": list(R.remainder([m for m in (front, back) if m is not None])),
    })
    return spec


def main():
    assert build_ref.build(verbose=False), "oracle/_ref could not be built"
    ref = ref_loader.load()
    R = ref.adapters
    rng = random.Random(20260926)
    out = {"single": [single_case(rng, R, rng.random() < 0.5) for _ in range(250)],
           "linked": [linked_case(rng, R) for _ in range(120)]}
    path = os.path.join(HERE, "matches.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
