#!/usr/bin/env python3
"""Copy the small FASTQ inputs and expected outputs of the reference's command-line tests into
tests/golden/fastq/ (run in the build container; the GPU box has no /root/reference).

Each case below is one `run(params, expected, input)` of reference tests/test_commandline.py
(file:line given); `expected` comes from tests/cut/, `input` from tests/data/.
"""
import json
import os
import shutil

REF = os.environ.get("CUTADAPT_REFERENCE", "/root/reference") + "/tests"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "fastq")

CASES = [
    # (name, reference test line, adapter kind, adapter sequence(s), extra, input, expected)
    ("small", "test_commandline.py:79", "back", ["TTAGACATATCTCCGTCG"], {}, "small.fastq", "small.fastq"),
    ("empty", "test_commandline.py:91", "back", ["TTAGACATATCTCCGTCG"], {}, "empty.fastq", "empty.fastq"),
    ("dos", "test_commandline.py:104", "back", ["TTAGACATATCTCCGTCG"], {"max_errors": 0.12}, "dos.fastq", "dos.fastq"),
    ("lowercase", "test_commandline.py:109", "back", ["ttagacatatctccgtcg"], {}, "small.fastq", "lowercase.fastq"),
    ("illumina_iupac", "test_commandline.py:376", "back", ["VCCGAMCYUCKHRKDCUBBCNUWNSGHCGU"], {}, "illumina.fastq.gz", "illumina.fastq"),
    ("illumina_u", "test_commandline.py:456", "back", ["GCCGAACUUCUUAGACUGCCUUAAGGACGU"], {}, "illumina.fastq.gz", "illumina.fastq"),
    ("small_anywhere_gz", "test_commandline.py:776", "anywhere", ["TTAGACATATCTCCGTCG"], {}, "small.fastq.gz", "small.fastq"),
]

os.makedirs(OUT, exist_ok=True)
manifest = []
for name, where, kind, seqs, extra, inp, exp in CASES:
    shutil.copyfile(os.path.join(REF, "data", inp), os.path.join(OUT, "in_" + inp))
    shutil.copyfile(os.path.join(REF, "cut", exp), os.path.join(OUT, "out_" + exp))
    manifest.append({"name": name, "reference_test": where, "kind": kind, "adapters": seqs, "extra": extra,
                     "input": "in_" + inp, "expected": "out_" + exp})
with open(os.path.join(OUT, "manifest.json"), "w") as f:
    json.dump(manifest, f, indent=1)
print("wrote", len(manifest), "cases to", OUT)
