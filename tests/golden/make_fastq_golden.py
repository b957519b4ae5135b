#!/usr/bin/env python3
"""Copy the small FASTA/FASTQ inputs and expected outputs of the reference's command-line tests
into tests/golden/fastq/ (run in the build container; the GPU box has no /root/reference).

Each case below is one `run(params, expected, input)` of reference tests/test_commandline.py or
tests/test_info_file.py (file:line given); `expected` (and `info`) come from tests/cut/, `input`
from tests/data/.  Only test DATA is copied; the command lines are restated as
(option, adapter specification) pairs plus the AdapterCutter options they set.
"""
import json
import os
import shutil

REF = os.environ.get("CUTADAPT_REFERENCE", "/root/reference") + "/tests"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "fastq")

T = "test_commandline.py"
I = "test_info_file.py"
CASES = [
    # name, reference test, adapters [(option, spec)], options, input, expected, expected info file
    ("small", f"{T}:79", [("-a", "TTAGACATATCTCCGTCG")], {}, "small.fastq", "small.fastq", None),
    ("empty", f"{T}:91", [("-a", "TTAGACATATCTCCGTCG")], {}, "empty.fastq", "empty.fastq", None),
    ("dos", f"{T}:104", [("-a", "TTAGACATATCTCCGTCG")], {"max_errors": 0.12}, "dos.fastq", "dos.fastq", None),
    ("lowercase", f"{T}:109", [("-a", "ttagacatatctccgtcg")], {}, "small.fastq", "lowercase.fastq", None),
    ("discard_trimmed", f"{T}:127", [("-b", "TTAGACATATCTCCGTCG")], {"discard_trimmed": True}, "small.fastq", "discard.fastq", None),
    ("discard_untrimmed", f"{T}:132", [("-b", "CAAGAT")], {"discard_untrimmed": True}, "small.fastq", "discard-untrimmed.fastq", None),
    ("twoadapters", f"{T}:261", [("-a", "AATTTCAGGAATT"), ("-a", "GTTCTCTAGTTCT")], {}, "twoadapters.fasta", "twoadapters.fasta", None),
    ("action_none", f"{T}:289", [("-a", "CCCTAGTTAAAC")], {"action": None, "discard_untrimmed": True}, "small.fastq", "no-trim.fastq", None),
    ("action_mask", f"{T}:303", [("-b", "CAAG")], {"times": 3, "action": "mask"}, "anywhere_repeat.fastq", "anywhere_repeat.fastq", None),
    ("action_lowercase", f"{T}:308", [("-b", "CAAG")], {"times": 3, "action": "lowercase"}, "action_lowercase.fasta", "action_lowercase.fasta", None),
    ("action_retain", f"{T}:316", [("-g", "GGTTAACC"), ("-a", "CAAG")], {"action": "retain"}, "action_retain.fasta", "action_retain.fasta", None),
    ("action_crop", f"{T}:329", [("-g", "GGTTAA"), ("-a", "CAAG")], {"action": "crop", "discard_untrimmed": True}, "action_retain.fasta", "action_crop.fasta", None),
    ("illumina_iupac", f"{T}:376", [("-a", "VCCGAMCYUCKHRKDCUBBCNUWNSGHCGU")], {}, "illumina.fastq.gz", "illumina.fastq", None),
    ("illumina_u", f"{T}:456", [("-a", "GCCGAACUUCUUAGACUGCCUUAAGGACGU")], {}, "illumina.fastq.gz", "illumina.fastq", None),
    ("linked_explicitly_anchored", f"{T}:668", [("-a", "^AAAAAAAAAA...TTTTTTTTTT")], {}, "linked.fasta", "linked.fasta", None),
    ("linked_multiple", f"{T}:672", [("-a", "^AAAAAAAAAA...TTTTTTTTTT"), ("-a", "^AAAAAAAAAA...GCGCGCGCGC")], {}, "linked.fasta", "linked.fasta", None),
    ("linked_both_anchored", f"{T}:680", [("-a", "^AAAAAAAAAA...TTTTT$")], {}, "linked.fasta", "linked-anchored.fasta", None),
    ("linked_5p_not_anchored", f"{T}:684", [("-g", "AAAAAAAAAA...TTTTTTTTTT")], {}, "linked.fasta", "linked-not-anchored.fasta", None),
    ("linked_discard_untrimmed", f"{T}:688", [("-a", "^AAAAAAAAAA...TTTTTTTTTT")], {"discard_untrimmed": True}, "linked.fasta", "linked-discard.fasta", None),
    ("linked_discard_untrimmed_g", f"{T}:696", [("-g", "AAAAAAAAAA...TTTTTTTTTT")], {"discard_untrimmed": True}, "linked.fasta", "linked-discard-g.fasta", None),
    ("linked_lowercase", f"{T}:705", [("-a", "^AACCGGTTTT...GGGGGGG$"), ("-a", "^AAAA...TTTT$")], {"times": 2, "action": "lowercase"}, "linked.fasta", "linked-lowercase.fasta", None),
    ("small_anywhere_gz", f"{T}:776", [("-b", "TTAGACATATCTCCGTCG")], {}, "small.fastq.gz", "small.fastq", None),
    ("qualtrim", f"{T}:246", [("-a", "XXXXXX")], {"quality_cutoff": [0, 10]}, "lowqual.fastq", "lowqual.fastq", None),
    ("qualbase", f"{T}:251", [("-a", "XXXXXX")], {"quality_cutoff": [0, 10], "quality_base": 64}, "illumina64.fastq", "illumina64.fastq", None),
    ("quality_trim_only", f"{T}:256", [], {"quality_cutoff": [0, 10], "quality_base": 64}, "illumina64.fastq", "illumina64.fastq", None),
    ("poly_a", f"{T}:280", [], {"poly_a": True}, "polya.1.fasta", "polya.1.fasta", None),
    ("nextseq", f"{T}:666", [], {"nextseq_trim": 22}, "nextseq.fastq", "nextseq.fastq", None),
    ("max_expected_errors", f"{T}:837", [], {"max_expected_errors": 0.9}, "maxee.fastq", "maxee.fastq", None),
    ("info_file", f"{I}:14", [("-a", "adapt=GCCGAACTTCTTAGACTGCCTTAAGGACGT")], {}, "illumina.fastq.gz", "illumina.fastq", "illumina.info.txt"),
    ("info_file_times", f"{I}:35", [("-a", "adapt=GCCGAACTTCTTA"), ("-a", "adapt2=GACTGCCTTAAGGACGT")], {"times": 2}, "illumina5.fastq", "illumina5.fastq", "illumina5.info.txt"),
    ("revcomp_normalized", f"{T}:827", [("-g", "^TTATTTGTCT"), ("-g", "^TCCGCACTGG")], {"revcomp": True, "index": False}, "revcomp.1.fastq", "revcomp-single-normalize.fastq", None),
    ("info_file_revcomp", f"{I}:78", [("-a", "adapt=GAGTCG")], {"revcomp": True, "rc_suffix": None}, "info-rc.fasta", None, "info-rc.txt"),
    ("linked_info_file", f"{I}:119", [("-a", "linkedadapter=^AAAAAAAAAA...TTTTTTTTTT")], {}, "linked.fasta", None, "linked-info.txt"),
]

os.makedirs(OUT, exist_ok=True)
manifest = []
for name, where, adapters, options, inp, exp, info in CASES:
    shutil.copyfile(os.path.join(REF, "data", inp), os.path.join(OUT, "in_" + inp))
    entry = {"name": name, "reference_test": where, "adapters": [list(a) for a in adapters], "options": options,
             "input": "in_" + inp, "expected": None, "info": None}
    if exp is not None:
        shutil.copyfile(os.path.join(REF, "cut", exp), os.path.join(OUT, "out_" + exp))
        entry["expected"] = "out_" + exp
    if info is not None:
        shutil.copyfile(os.path.join(REF, "cut", info), os.path.join(OUT, "info_" + info))
        entry["info"] = "info_" + info
    manifest.append(entry)
with open(os.path.join(OUT, "manifest.json"), "w") as f:
    json.dump(manifest, f, indent=1)
print("wrote", len(manifest), "cases to", OUT)
