#!/usr/bin/env python3
"""Golden vectors for the adapter-specification grammar (-a / -g / -b SPEC; reference parser.py:28-151, :441-551), generated
from the REFERENCE's parser itself (build container only):

    python tests/golden/make_parser_golden.py        ->  tests/golden/parser.json

parser.py is loaded from /root/reference/src/cutadapt as a module of the package under oracle/_ref (whose adapters module
is the reference's, compiled); its file readers (xopen, dnaio -- not installed here, not used by make_adapter) are stubbed.
Random specifications over the whole grammar -- names, anchors, X markers, brace repeats, ellipsis forms, linked adapters,
every search parameter, valid and invalid combinations; per specification the exception class the reference raises or what
it builds.  tests/test_parser_golden.py replays them on cutadapt_amd.pipeline.adapter_from_spec (no GPU)."""
import importlib.util
import json
import os
import random
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import build_ref, ref_loader  # noqa: E402

REFERENCE = os.environ.get("CUTADAPT_REFERENCE", "/root/reference")
DEFAULTS = dict(max_errors=0.1, min_overlap=3, read_wildcards=False, adapter_wildcards=True, indels=True)


def load_parser():
    assert build_ref.build(verbose=False), "oracle/_ref could not be built"
    ref_loader.load()                                   # puts oracle/_ref on sys.path: `cutadapt` is the compiled reference
    for name, attrs in (("xopen", {"xopen": None}), ("dnaio", {}), ("dnaio.readers", {"FastaReader": None})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("cutadapt.parser", os.path.join(REFERENCE, "src", "cutadapt", "parser.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cutadapt.parser"] = mod
    spec.loader.exec_module(mod)
    return mod


def rs(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


VALID = ["e=0.2", "max_error_rate=0.15", "error_rate=0.3", "max_errors=2", "e=1", "o=4", "min_overlap=5", "o=40", "noindels",
         "indels", "anywhere", "rightmost"]
LINKED_ONLY = ["required", "optional"]
BROKEN = ["e=abc", "foo=1", "o=", "e=0.1;e=0.2", "bar", "indels;noindels", "required;optional"]


def one_part(rng, side, sane, linked):
    """side: 'front' / 'back' / 'anywhere' -- which markers make sense there; sane: keep to them"""
    seq = rs(rng, rng.randint(4, 12), rng.choice(["ACGT", "ACGT", "ACGTN", "acgt", "ACGTRY"]))
    if rng.random() < 0.15:
        p = rng.randrange(len(seq))
        seq = seq[:p + 1] + "{%d}" % rng.choice([0, 2, 5]) + seq[p + 1:]
    if not sane and rng.random() < 0.2:
        seq = seq + rng.choice(["{", "}", "{x}", "Z"])
    if sane:
        pre = rng.choice(["", "", "^", "X"]) if side == "front" else ""
        post = rng.choice(["", "", "$", "X"]) if side == "back" else ""
    else:
        pre = rng.choice(["", "^", "X", "^X"])
        post = rng.choice(["", "$", "X", "X$"])
    out = pre + seq + post
    pool = VALID + (LINKED_ONLY if linked or not sane else []) + ([] if sane else BROKEN)
    chosen, keys = [], set()
    for p in rng.sample(pool, rng.choice([0, 0, 1, 1, 2, 3])):
        key = {"e": "err", "max_error_rate": "err", "error_rate": "err", "max_errors": "err", "o": "ov", "min_overlap": "ov",
               "noindels": "ind", "indels": "ind", "required": "req", "optional": "req"}.get(p.split("=")[0], p)
        if sane and key in keys:
            continue
        if sane and p == "rightmost" and (pre or post or side == "anywhere"):
            continue
        if sane and p.split("=")[0] in ("o", "min_overlap") and (pre == "^" or post == "$"):
            continue
        keys.add(key)
        chosen.append(p)
    for p in chosen:
        out += rng.choice([";", "; ", " ;"]) + p
    return out


def random_spec(rng, kind):
    sane = rng.random() < 0.75
    shape = rng.random()
    if shape < 0.55:
        s = one_part(rng, kind, sane, False)
    elif shape < 0.65:
        s = one_part(rng, "front" if sane else kind, sane, False) + "..."
    elif shape < 0.75:
        s = "..." + one_part(rng, "back" if sane else kind, sane, False)
    else:
        s = one_part(rng, "front", sane, True) + "..." + one_part(rng, "back", sane, True)
    if rng.random() < 0.3:
        s = rng.choice(["nm", "my_adapter", "a b"]) + "=" + s
    return s


def describe_single(ad):
    return {"cls": type(ad).__name__, "sequence": ad.sequence, "max_error_rate": float(ad.max_error_rate),
            "min_overlap": ad.min_overlap, "indels": bool(ad.indels), "read_wildcards": bool(ad.read_wildcards),
            "adapter_wildcards": bool(ad.adapter_wildcards), "spec": ad.spec(), "flags": int(ad.aligner.flags)
            if hasattr(ad.aligner, "flags") else None, "aligner": type(ad.aligner).__name__}


def main():
    P = load_parser()
    rng = random.Random(27182818)
    cases = []
    for _ in range(900):
        kind = rng.choice(["back", "back", "front", "front", "anywhere"])
        spec = random_spec(rng, kind)
        case = {"spec": spec, "type": kind}
        try:
            ad = P.make_adapter(spec, kind, dict(DEFAULTS))
        except Exception as exc:
            case["error"] = type(exc).__name__
            cases.append(case)
            continue
        named = "=" in spec.split(";")[0].split("...")[0] and not spec.split("=")[0].strip().isdigit()
        if type(ad).__name__ == "LinkedAdapter":
            case["want"] = {"cls": "LinkedAdapter", "front": describe_single(ad.front_adapter), "back": describe_single(ad.back_adapter),
                            "front_required": bool(ad.front_required), "back_required": bool(ad.back_required)}
        else:
            case["want"] = describe_single(ad)
        if named:
            case["want"]["name"] = ad.name
        cases.append(case)
    path = os.path.join(HERE, "parser.json")
    with open(path, "w") as f:
        json.dump({"defaults": DEFAULTS, "cases": cases}, f, indent=0)
    n_err = sum("error" in c for c in cases)
    print("wrote", path, os.path.getsize(path), "bytes;", n_err, "error cases,", sum(c.get("want", {}).get("cls") == "LinkedAdapter" for c in cases), "linked, of", len(cases))


if __name__ == "__main__":
    main()
