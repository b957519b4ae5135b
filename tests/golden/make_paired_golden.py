#!/usr/bin/env python3
"""Copy the inputs and expected outputs of the reference's paired-end command-line tests
(tests/test_paired.py) into tests/golden/paired/ and restate their options (run in the build
container).  Only test DATA is copied."""
import json
import os
import shutil

REF = os.environ.get("CUTADAPT_REFERENCE", "/root/reference") + "/tests"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "paired")
P = "test_paired.py"

def case(name, line, in1, in2, exp1, exp2, r1=None, r2=None, **top):
    return {"name": name, "reference_test": f"{P}:{line}", "in1": in1, "in2": in2, "exp1": exp1, "exp2": exp2,
            "r1": r1 or {}, "r2": r2 or {}, "top": top}

A1 = [["-a", "TTAGACATAT"]]
A2 = [["-a", "CAGTGGAGTA"]]
CASES = [
    case("no_legacy", 43, "paired.1.fastq", "paired.2.fastq", "paired.m14.1.fastq", "paired.m14.2.fastq",
         {"adapters": A1, "quality_cutoff": [0, 10]}, {"quality_cutoff": [0, 10]}, minimum_length=14),
    case("paired_end", 270, "paired.1.fastq", "paired.2.fastq", "paired.1.fastq", "paired.2.fastq",
         {"adapters": A1}, {"adapters": A2}, minimum_length=14),
    case("anchored_back_no_indels", 282, "anchored-back.fasta", "anchored-back.fasta", "anchored-back.fasta", "anchored-back.fasta",
         {"adapters": [["-a", "BACKADAPTER$"]], "params": {"indels": False, "adapter_wildcards": False}},
         {"adapters": [["-a", "BACKADAPTER$"]], "params": {"indels": False, "adapter_wildcards": False}}),
    case("qualtrim", 293, "paired.1.fastq", "paired.2.fastq", "pairedq.1.fastq", "pairedq.2.fastq",
         {"adapters": A1, "quality_cutoff": [0, 20]}, {"adapters": A2, "quality_cutoff": [0, 20]}, minimum_length=14, maximum_length=90),
    case("qualtrim_swapped", 305, "paired.2.fastq", "paired.1.fastq", "pairedq.2.fastq", "pairedq.1.fastq",
         {"adapters": A2, "quality_cutoff": [0, 20]}, {"adapters": A1, "quality_cutoff": [0, 20]}, minimum_length=14),
    case("qualtrim_r2_none", 318, "lowqual.fastq", "lowqual.fastq", "lowqual.unchanged.fastq", "lowqual.unchanged.fastq"),
    case("qualtrim_r2_q", 318, "lowqual.fastq", "lowqual.fastq", "lowqual.fastq", "lowqual.fastq",
         {"quality_cutoff": [0, 10]}, {"quality_cutoff": [0, 10]}),
    case("qualtrim_r2_Q", 318, "lowqual.fastq", "lowqual.fastq", "lowqual.unchanged.fastq", "lowqual.fastq",
         {}, {"quality_cutoff": [0, 10]}),
    case("qualtrim_r2_q_only", 318, "lowqual.fastq", "lowqual.fastq", "lowqual.fastq", "lowqual.unchanged.fastq",
         {"quality_cutoff": [0, 10]}, {}),
    case("cut", 339, "paired.1.fastq", "paired.2.fastq", "pairedu.1.fastq", "pairedu.2.fastq",
         {"cut": [3, -1]}, {"cut": [4, -2]}),
    case("length", 350, "paired.1.fastq", "paired.2.fastq", "length5.1.fastq", "length5.2.fastq", {"length": 5}, {"length": 5}),
    case("negative_length", 361, "paired.1.fastq", "paired.2.fastq", "length-5.1.fastq", "length-5.2.fastq", {"length": -5}, {"length": -5}),
    case("length_l_L", 372, "paired.1.fastq", "paired.2.fastq", "length5.1.fastq", "length-5.2.fastq", {"length": 5}, {"length": -5}),
    case("length_only_L", 383, "paired.1.fastq", "paired.2.fastq", "paired-unchanged.1.fastq", "length5.2.fastq", {}, {"length": 5}),
    case("upper_a_only", 394, "paired.1.fastq", "paired.2.fastq", "paired-onlyA.1.fastq", "paired-onlyA.2.fastq", {}, {"adapters": A2}),
    case("discard_untrimmed", 405, "paired.1.fastq", "paired.2.fastq", "empty.fastq", "empty.fastq",
         {"adapters": [["-a", "CTCCAGCTTAGACATATC"]]}, {"adapters": [["-a", "XXXXXXXX"]]}, discard_untrimmed=True),
    case("discard_trimmed", 418, "paired.1.fastq", "paired.2.fastq", "empty.fastq", "empty.fastq",
         {}, {"adapters": [["-a", "C"]], "params": {"min_overlap": 1}}, discard_trimmed=True),
    case("pair_filter_both", 492, "paired.1.fastq", "paired.2.fastq", "paired-filterboth.1.fastq", "paired-filterboth.2.fastq",
         {"adapters": A1}, {"adapters": [["-a", "GGAGTA"]]}, minimum_length=14, pair_filter="both"),
    case("pair_filter_first", 503, "paired.1.fastq", "paired.2.fastq", "paired-filterfirst.1.fastq", "paired-filterfirst.2.fastq",
         {"adapters": A1}, {"adapters": [["-a", "GGAGTA"]]}, minimum_length=14, pair_filter="first"),
    case("poly_a_poly_t", 775, "polya.1.fasta", "polya.2.fasta", "polya.1.fasta", "polya.2.fasta", {"poly_a": True}, {"poly_a": True}),
    case("pair_adapters", 668, "paired.1.fastq", "paired.2.fastq", "pair-adapters.1.fastq", "pair-adapters.2.fastq",
         {"adapters": [["-a", "GTCTCCAGCT"]]}, {"adapters": [["-a", "GACAAATAAC"]]}, pair_adapters=True),
    case("revcomp_only_r1", 786, "revcomp.1.fastq", "revcomp.2.fastq", "revcomp.1.fastq", "revcomp.2.fastq",
         {"adapters": [["-g", "^TTATTTGTCT"], ["-g", "^TCCGCACTGGC"]]}, {}, revcomp=True),
    case("revcomp_only_r2", 803, "revcomp.2.fastq", "revcomp.1.fastq", "revcomp.2.fastq", "revcomp.1.fastq",
         {}, {"adapters": [["-g", "^TTATTTGTCT"], ["-g", "^TCCGCACTGGC"]]}, revcomp=True),
    case("revcomp_r1_and_r2", 820, "revcomp.1.fastq", "revcomp.2.fastq", "revcomp-r1r2.1.fastq", "revcomp-r1r2.2.fastq",
         {"adapters": [["-g", "^TTATTTGTCT"]]}, {"adapters": [["-g", "^TCCGCACTGGC"]]}, revcomp=True),
    case("nextseq", 561, "nextseq.fastq", "nextseq.fastq", "nextseq.fastq", "nextseq.fastq", {"nextseq_trim": 22}, {"nextseq_trim": 22}),
]

os.makedirs(OUT, exist_ok=True)
for c in CASES:
    for key, sub in (("in1", "data"), ("in2", "data"), ("exp1", "cut"), ("exp2", "cut")):
        src = os.path.join(REF, sub, c[key])
        dst = ("in_" if sub == "data" else "out_") + c[key]
        shutil.copyfile(src, os.path.join(OUT, dst))
        c[key] = dst
with open(os.path.join(OUT, "manifest.json"), "w") as f:
    json.dump(CASES, f, indent=1)
print("wrote", len(CASES), "cases to", OUT)
