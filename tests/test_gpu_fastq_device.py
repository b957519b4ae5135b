"""FASTQ parsed, matched and formatted on the device (cutadapt_amd/gpu_pipeline.py, csrc/fastq_gpu.hip) must
write the same bytes as the host-side batch pipeline (pipeline.trim_fastq, itself pinned to the reference's
command-line goldens) and as the reference's golden files themselves.  GPU only."""
import io
import json
import os
import random

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FQ = os.path.join(HERE, "golden", "fastq")


def _fastq(rng, n, adapters, crlf=False, final_newline=True, lower=False, twice=False, lead=None, turned=False):
    recs = []
    for i in range(n):
        L = rng.randint(0, 160)
        s = "".join(rng.choice("ACGT") for _ in range(L))
        if twice and rng.random() < 0.4 and adapters:                   # (something for a second --times round to find)
            pos = rng.randint(0, L)
            s = s[:pos] + rng.choice(adapters) + s[pos:]
            L = len(s)
        if rng.random() < 0.6 and adapters:
            ad = rng.choice(adapters)
            ad = ad[:rng.randint(1, len(ad))] if rng.random() < 0.5 else ad
            pos = rng.randint(0, L)
            s = (s[:pos] + ad + s[pos:])[:max(L, 1)] if rng.random() < 0.5 else s[:pos] + ad
        if lead and rng.random() < 0.5:                                 # (a 5' adapter in front: linked adapters)
            s = (lead if rng.random() < 0.7 else lead[1:]) + s
        if rng.random() < 0.15:
            s += "A" * rng.randint(2, 30)                               # (a poly-A tail, sometimes with an error in it)
            if rng.random() < 0.3 and len(s) > 4:
                s = s[:-3] + "C" + s[-2:]
        if rng.random() < 0.1:
            s = "".join(c if rng.random() > 0.05 else "N" for c in s)
        if lower and rng.random() < 0.3:
            s = s.lower()
        if turned and rng.random() < 0.45:                              # (--revcomp: the other strand of the same fragment)
            s = s[::-1].translate(str.maketrans("ACGTNacgtn", "TGCANtgcan"))
        q = "".join(chr(rng.randint(33, 73)) for _ in s)
        name = f"read{i}" + (" some comment:" + "x" * rng.randint(0, 30) if rng.random() < 0.3 else "")
        recs.append(f"@{name}\n{s}\n+{name if rng.random() < 0.1 else ''}\n{q}\n")
    text = "".join(recs)
    if crlf:
        text = text.replace("\n", "\r\n")
    if not final_newline and text.endswith("\n"):
        text = text[:-2] if crlf else text[:-1]
    return text.encode()


def test_device_fastq_equals_host_pipeline(hip):
    import numpy as np
    from cutadapt_amd import adapters as A
    from cutadapt_amd.gpu_pipeline import trim_fastq_gpu
    from cutadapt_amd.pipeline import trim_fastq
    rng = random.Random(2024)
    ad_seqs = ["AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", "TGGAATTCTCGGGTGCCAAGG", "ACGTACGTTT"]
    cases = [
        ([A.BackAdapter(ad_seqs[0])], {}),
        ([A.BackAdapter(ad_seqs[0]), A.FrontAdapter(ad_seqs[1]), A.AnywhereAdapter(ad_seqs[2])], {}),
        ([A.PrefixAdapter(ad_seqs[2]), A.SuffixAdapter(ad_seqs[1])], {"discard_untrimmed": True}),
        ([A.BackAdapter(ad_seqs[0])], {"minimum_length": 30, "maximum_length": 120}),
        ([A.NonInternalBackAdapter(ad_seqs[1]), A.BackAdapter(ad_seqs[0])], {"discard_trimmed": True}),
        # modifiers in front of the adapter step, still without a byte of per-read data on the host
        ([A.BackAdapter(ad_seqs[0])], {"quality_cutoff": (0, 20), "minimum_length": 20}),
        ([A.BackAdapter(ad_seqs[0]), A.FrontAdapter(ad_seqs[1])], {"quality_cutoff": (15, 25), "cut": [3, -2]}),
        ([A.AnywhereAdapter(ad_seqs[2])], {"nextseq_trim": 20, "cut": [-5], "maximum_length": 140}),
        # ... and behind it: --poly-a, -l, --max-ee, with the filters applied after them
        ([A.BackAdapter(ad_seqs[0])], {"poly_a": True, "minimum_length": 25}),
        ([A.BackAdapter(ad_seqs[0]), A.FrontAdapter(ad_seqs[1])], {"quality_cutoff": (0, 15), "poly_a": True, "length": 80,
                                                               "max_expected_errors": 2.5, "minimum_length": 10}),
        ([A.BackAdapter(ad_seqs[0])], {"length": -40, "max_expected_errors": 1.0, "discard_untrimmed": True}),
        # --times N: every further round matches what the last one kept, still on the device
        ([A.BackAdapter(ad_seqs[0]), A.FrontAdapter(ad_seqs[1]), A.AnywhereAdapter(ad_seqs[2])], {"times": 3, "minimum_length": 10}),
        ([A.BackAdapter(ad_seqs[0]), A.BackAdapter(ad_seqs[1])], {"times": 2, "quality_cutoff": (0, 15), "poly_a": True,
                                                              "discard_untrimmed": True}),
        # one linked adapter: 5' plan, view, 3' plan, verdict and interval on the device (every required / optional mix)
        ([A.LinkedAdapter(A.PrefixAdapter(ad_seqs[2]), A.BackAdapter(ad_seqs[0]), True, False, "l1")], {"minimum_length": 5}),
        ([A.LinkedAdapter(A.PrefixAdapter(ad_seqs[2]), A.BackAdapter(ad_seqs[0]), True, True, "l2")], {"discard_untrimmed": True}),
        ([A.LinkedAdapter(A.FrontAdapter(ad_seqs[2]), A.BackAdapter(ad_seqs[0]), False, True, "l3")],
         {"quality_cutoff": (5, 15), "poly_a": True, "maximum_length": 100}),
        ([A.LinkedAdapter(A.FrontAdapter(ad_seqs[2]), A.SuffixAdapter(ad_seqs[1]), False, False, "l4")], {"cut": [1], "length": 50}),
        # rightmost adapters (round 6; reference adapters.py:733-789, :841-893): the reversed adapter on the reversed views,
        # coordinates mirrored back -- one, two of them, several rounds, modifiers on both sides
        ([A.RightmostBackAdapter(ad_seqs[0])], {"minimum_length": 10}),
        ([A.RightmostFrontAdapter(ad_seqs[1]), A.RightmostBackAdapter(ad_seqs[2])], {"times": 2, "quality_cutoff": (5, 15), "poly_a": True}),
        ([A.RightmostFrontAdapter(ad_seqs[2])], {"cut": [2, -1], "length": 100, "discard_untrimmed": True}),
        # several linked adapters (round 6): MultipleAdapters' rule over the LinkedMatches -- scores and errors of the parts
        # added up --, every required / optional mix, modifiers on both sides
        ([A.LinkedAdapter(A.PrefixAdapter(ad_seqs[2]), A.BackAdapter(ad_seqs[0]), True, False, "m1"),
          A.LinkedAdapter(A.PrefixAdapter(ad_seqs[2]), A.BackAdapter(ad_seqs[1]), True, False, "m2")], {"minimum_length": 5}),
        ([A.LinkedAdapter(A.FrontAdapter(ad_seqs[2]), A.BackAdapter(ad_seqs[0]), False, True, "m3"),
          A.LinkedAdapter(A.PrefixAdapter(ad_seqs[2][:6]), A.BackAdapter(ad_seqs[1]), True, True, "m4"),
          A.LinkedAdapter(A.FrontAdapter(ad_seqs[1][:10]), A.SuffixAdapter(ad_seqs[2]), False, False, "m5")],
         {"quality_cutoff": (5, 15), "poly_a": True, "maximum_length": 120, "discard_untrimmed": True}),
        ([A.LinkedAdapter(A.FrontAdapter(ad_seqs[2]), A.BackAdapter(ad_seqs[0]), True, True, "m6"),
          A.LinkedAdapter(A.FrontAdapter(ad_seqs[2]), A.BackAdapter(ad_seqs[0][:20]), False, False, "m7")], {"action": "mask", "cut": [1]}),
        # --action none / retain / crop (one round): other intervals from the same matches (round 4)
        ([A.BackAdapter(ad_seqs[0]), A.FrontAdapter(ad_seqs[1]), A.AnywhereAdapter(ad_seqs[2])], {"action": None, "discard_untrimmed": True}),
        ([A.BackAdapter(ad_seqs[0]), A.FrontAdapter(ad_seqs[1]), A.AnywhereAdapter(ad_seqs[2])], {"action": "retain", "minimum_length": 12}),
        ([A.FrontAdapter(ad_seqs[1]), A.BackAdapter(ad_seqs[0])], {"action": "crop", "quality_cutoff": (0, 15), "discard_untrimmed": True}),
        ([A.AnywhereAdapter(ad_seqs[2]), A.SuffixAdapter(ad_seqs[1])], {"action": "retain", "cut": [2, -1], "poly_a": True, "length": 90}),
        ([A.BackAdapter(ad_seqs[0])], {"action": "crop", "max_expected_errors": 3.0, "maximum_length": 40}),
        # --action mask / lowercase (round 6; reference modifiers.py:170-198): the rounds as for trim, the record written whole
        # and marked around what they would have kept -- with modifiers in front, several rounds, -l and the filters behind
        ([A.BackAdapter(ad_seqs[0])], {"action": "mask"}),
        ([A.BackAdapter(ad_seqs[0]), A.FrontAdapter(ad_seqs[1]), A.AnywhereAdapter(ad_seqs[2])], {"action": "lowercase", "minimum_length": 40}),
        ([A.BackAdapter(ad_seqs[0]), A.FrontAdapter(ad_seqs[1]), A.AnywhereAdapter(ad_seqs[2])], {"action": "mask", "times": 3, "discard_untrimmed": True}),
        ([A.FrontAdapter(ad_seqs[1]), A.BackAdapter(ad_seqs[0])], {"action": "lowercase", "quality_cutoff": (10, 15), "cut": [2, -3], "times": 2,
                                                              "length": 100, "max_expected_errors": 4.0, "maximum_length": 130}),
        ([A.NonInternalBackAdapter(ad_seqs[1]), A.PrefixAdapter(ad_seqs[2])], {"action": "mask", "nextseq_trim": 20, "discard_trimmed": True}),
        # ... with --poly-a behind them (the trimmer sees the MARKED read) and on one linked adapter
        ([A.BackAdapter(ad_seqs[0])], {"action": "mask", "poly_a": True, "minimum_length": 20}),
        ([A.BackAdapter(ad_seqs[0]), A.FrontAdapter(ad_seqs[1])], {"action": "lowercase", "poly_a": True, "times": 2, "quality_cutoff": (0, 12)}),
        ([A.LinkedAdapter(A.FrontAdapter(ad_seqs[2]), A.BackAdapter(ad_seqs[0]), False, True, "l5")], {"action": "lowercase", "minimum_length": 10}),
        ([A.LinkedAdapter(A.PrefixAdapter(ad_seqs[2]), A.BackAdapter(ad_seqs[0]), True, False, "l6")], {"action": "mask", "poly_a": True}),
        # --revcomp (round 6; reference modifiers.py:264-308): both orientations matched, the better one turned around in place
        # in HBM and named with the suffix -- alone, with several adapter types, with modifiers on both sides, no suffix
        ([A.BackAdapter(ad_seqs[0])], {"revcomp": True}),
        ([A.BackAdapter(ad_seqs[0]), A.FrontAdapter(ad_seqs[1]), A.AnywhereAdapter(ad_seqs[2])], {"revcomp": True, "rc_suffix": None, "minimum_length": 20}),
        ([A.FrontAdapter(ad_seqs[1]), A.BackAdapter(ad_seqs[0])], {"revcomp": True, "rc_suffix": "/turned around", "quality_cutoff": (10, 15), "cut": [2, -3],
                                                              "poly_a": True, "length": 100, "max_expected_errors": 4.0, "discard_untrimmed": True}),
        ([A.NonInternalBackAdapter(ad_seqs[1]), A.PrefixAdapter(ad_seqs[2])], {"revcomp": True, "nextseq_trim": 20, "discard_trimmed": True}),
        ([A.AnywhereAdapter(ad_seqs[0]), A.SuffixAdapter(ad_seqs[1])], {"revcomp": True, "maximum_length": 120, "length": -90}),
        # no adapter at all: the modifiers and filters alone, still on the device
        ([], {"quality_cutoff": (10, 20), "minimum_length": 30}),
        ([], {"nextseq_trim": 20, "poly_a": True, "length": 100, "max_expected_errors": 2.0}),
        ([], {"cut": [4, -3], "maximum_length": 150}),
    ]
    for ci, (ads, opts) in enumerate(cases):
        for crlf, final_nl, chunk in ((False, True, 1 << 20), (True, True, 4096), (False, False, 700)):
            data = _fastq(rng, 3000, ad_seqs, crlf=crlf, final_newline=final_nl, lower=ci == 1 or opts.get("action") == "lowercase" or opts.get("rc_suffix", "") is None, twice="times" in opts,
                          turned=bool(opts.get("revcomp")),
                          lead=ad_seqs[2] if (ads and isinstance(ads[0], A.LinkedAdapter)) else None)
            want = io.BytesIO()
            ws = trim_fastq(io.BytesIO(data), want, ads, index=False, **opts)
            for source, assemble in ((io.BytesIO(data), "device"), (np.frombuffer(data, dtype=np.uint8), "device"),
                                     (io.BytesIO(data), "host"), (np.frombuffer(data, dtype=np.uint8), "mixed")):
                got = io.BytesIO()
                gs = trim_fastq_gpu(source, got, ads, chunk_bytes=chunk, threads=2, devices="all", assemble=assemble,
                                    **opts)
                assert got.getvalue() == want.getvalue(), (ci, crlf, final_nl, chunk, assemble)
                assert (gs["reads"], gs["with_adapters"], gs["bp_in"], gs["bp_out"]) == \
                       (ws["reads"], ws["with_adapters"], ws["bp_in"], ws["bp_out"]), (ci, gs, ws["reads"])
                assert gs["way"] == "all-device"
                if opts.get("revcomp"):
                    assert gs["reverse_complemented"] == ws["reverse_complemented"] > 100, (ci, gs["reverse_complemented"])
                assert (gs["quality_trimmed_bases"], gs["nextseq_trimmed_bases"]) == \
                       (ws["trimmer"].quality_trimmed_bases, ws["trimmer"].nextseq_trimmed_bases), ci
                assert gs["too_many_expected_errors"] == ws["trimmer"].too_many_expected_errors, ci
                if opts.get("poly_a"):
                    assert gs["poly_a_trimmed_lengths"] == {k: v for k, v in ws["trimmer"].poly_a_trimmed_lengths.items() if v}, ci
                assert gs["filtered"] == {"too_short": ws["trimmer"].filtered.get("too_short", 0),
                                          "too_long": ws["trimmer"].filtered.get("too_long", 0)}, ci
            assert gs["bytes_out"] == len(want.getvalue())
    # --info-file on the all-device way (round 6): single adapters -- action trim with any number of rounds, or one round of an
    # action that leaves the characters alone --, linked adapters (two rows per match: the parts),
    # --revcomp included -- the rows are formatted on the device (cah_info_format_device): the same bytes as the host writer's
    # (reference steps.py:232-253: match rows cut the read AS IT CAME IN at the match's coordinates, "-1" rows show it as written)
    with_info = 0
    for ci, (ads, opts) in enumerate(cases):
        if opts.get("action", "trim") in ("mask", "lowercase"):
            continue
        if opts.get("times", 1) != 1 and opts.get("action", "trim") != "trim":
            continue
        for crlf, chunk in ((False, 1 << 20), (True, 3000)):
            data = _fastq(rng, 1500, ad_seqs, crlf=crlf, lower=ci == 1 or opts.get("rc_suffix", "") is None, turned=bool(opts.get("revcomp")),
                          twice="times" in opts, lead=ad_seqs[2] if (ads and isinstance(ads[0], A.LinkedAdapter)) else None)
            want, want_info = io.BytesIO(), io.BytesIO()
            trim_fastq(io.BytesIO(data), want, ads, index=False, info_file=want_info, **opts)
            got, got_info = io.BytesIO(), io.BytesIO()
            gs = trim_fastq_gpu(np.frombuffer(data, dtype=np.uint8), got, ads, chunk_bytes=chunk, threads=2, info_file=got_info, **opts)
            assert gs["way"] == "all-device", (ci, gs["way"])
            assert got.getvalue() == want.getvalue(), (ci, crlf)
            assert got_info.getvalue() == want_info.getvalue(), (ci, crlf)
            assert got_info.getvalue().count(b"\n") >= 1500
        with_info += 1
    assert with_info >= 29, with_info
    # malformed input is reported, not silently processed
    from cutadapt_amd.gpu_pipeline import trim_fastq_gpu as g
    for bad in (b"@r\nACGT\n-\nIIII\n", b"@r\nACGT\n+\nIII\n", b"r\nACGT\n+\nIIII\n", b"@r\nACGT\n+\n"):
        with pytest.raises(ValueError):
            g(np.frombuffer(bad, dtype=np.uint8), io.BytesIO(), [A.BackAdapter(ad_seqs[0])])
    assert g(np.zeros(0, dtype=np.uint8), io.BytesIO(), [A.BackAdapter(ad_seqs[0])])["reads"] == 0


def test_device_fastq_reference_goldens(hip):
    """the reference's command-line goldens that fall into this stage's scope (plain FASTQ, --times 1, action
    trim, no info file): byte for byte from the device-side path"""
    from cutadapt_amd.gpu_pipeline import trim_fastq_gpu
    from cutadapt_amd.pipeline import adapter_from_spec
    manifest = json.load(open(os.path.join(FQ, "manifest.json")))
    kinds = {"-a": "back", "-g": "front", "-b": "anywhere"}
    done = 0
    for case in manifest:
        opts = dict(case.get("options", {}))
        params = {"max_errors": opts.pop("max_errors")} if "max_errors" in opts else {}
        if case.get("info") or not case.get("expected") or not case["input"].endswith((".fastq", ".fq")):
            continue
        if opts.get("times", 1) != 1 or opts.get("action", "trim") != "trim":
            continue
        allowed = {"times", "action", "discard_untrimmed", "discard_trimmed", "minimum_length", "maximum_length"}
        if not set(opts) <= allowed:
            continue
        try:
            ads = [adapter_from_spec(spec, kinds[o], **params) for o, spec in case["adapters"]]
        except Exception:
            continue
        from cutadapt_amd.adapters import SingleAdapter
        if not ads or not all(isinstance(a, SingleAdapter) and not a._reverse_reads for a in ads):
            continue
        from cutadapt_amd.adapters import PrefixAdapter, SuffixAdapter
        if sum(isinstance(a, (PrefixAdapter, SuffixAdapter)) for a in ads) > 1:
            continue                     # the reference regroups several anchored adapters behind an index
        opts.pop("times", None); opts.pop("action", None)
        expected = open(os.path.join(FQ, case["expected"]), "rb").read()
        for chunk in (1 << 20, 512):
            out = io.BytesIO()
            trim_fastq_gpu(os.path.join(FQ, case["input"]), out, ads, chunk_bytes=chunk, threads=2, **opts)
            assert out.getvalue() == expected, (case["name"], chunk)
        done += 1
    assert done >= 5, done


def test_feeder_processes_write_the_same_bytes_as_feeder_threads(hip, tmp_path):
    """trim_fastq_gpu(feeder="process") -- one feeder process per GPU, each reading its own byte ranges and writing its
    chunks at the offsets the parent hands out (reference runners.py:275-412: reader, N worker processes, ordered
    writer) -- against the one-process thread mode: the same output file byte for byte, the same counters.  (Two and
    three processes on the one device of the test box; small chunks: many hand-overs.)"""
    from cutadapt_amd.adapters import BackAdapter, FrontAdapter
    from cutadapt_amd.gpu_pipeline import trim_fastq_gpu
    rng = random.Random(404)
    ads = ["AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", "GCCGAACTTCTTAGACTGCCTTAAGGACGT"]
    src = tmp_path / "in.fastq"
    src.write_bytes(_fastq(rng, 40_000, ads))
    cases = [
        (dict(adapters=[BackAdapter(ads[0], max_errors=0.1, min_overlap=3)], minimum_length=20), "all-device"),
        (dict(adapters=[BackAdapter(a, max_errors=0.1, min_overlap=3) for a in ads], times=2, quality_cutoff=(0, 20),
              discard_untrimmed=True), "all-device"),
        (dict(adapters=[FrontAdapter(ads[1][:12], max_errors=0.1)], action="mask"), "all-device"),
        (dict(adapters=[FrontAdapter(ads[1][:12], max_errors=0.1)], action="mask", poly_a=True), "all-device"),
        (dict(adapters=[FrontAdapter(ads[1][:12], max_errors=0.1)], action="mask", revcomp=True), "general"),
        (dict(adapters=[BackAdapter(ads[0], max_errors=0.1, min_overlap=3)], revcomp=True, minimum_length=20), "all-device"),
    ]
    for k, (opts, way) in enumerate(cases):
        for n_proc in (2, 3):
            a, b = tmp_path / f"thread{k}.fastq", tmp_path / f"process{k}_{n_proc}.fastq"
            ra = trim_fastq_gpu(str(src), str(a), chunk_bytes=1 << 20, threads=2, devices=[0], **opts)
            rb = trim_fastq_gpu(str(src), str(b), chunk_bytes=1 << 20, threads=2, devices=[0] * n_proc, feeder="process", **opts)
            assert ra["way"] == way and rb["way"].startswith(way), (ra["way"], rb["way"])
            assert a.read_bytes() == b.read_bytes(), (k, n_proc)
            for key in ("reads", "with_adapters", "bp_in", "bp_out", "bytes_out", "too_many_expected_errors"):
                assert ra[key] == rb[key], (k, n_proc, key, ra[key], rb[key])
            assert dict(ra["filtered"]) == dict(rb["filtered"]), (k, n_proc)
            assert rb["feeder_processes"] == n_proc
    # no sink: the sizes still add up
    r = trim_fastq_gpu(str(src), None, chunk_bytes=1 << 20, threads=2, devices=[0, 0], feeder="process", **cases[0][0])
    assert r["bytes_out"] == os.path.getsize(tmp_path / "thread0.fastq")
    with pytest.raises(ValueError):
        trim_fastq_gpu(io.BytesIO(src.read_bytes()), None, feeder="process", **cases[0][0])


def test_revcomp_suffix_and_info_entries_at_the_c_abi(hip):
    """cah_revcomp_in_place_device / cah_fastq_format_suffix_device / cah_info_format_device called directly on a small chunk
    in HBM, against a per-record restatement of ReverseComplementer + InfoFileWriter (reference modifiers.py:280-297,
    steps.py:232-253, adapters.py:395-417); argument errors are reported, not executed"""
    import ctypes as C
    import numpy as np
    import torch
    from cutadapt_amd import _lib
    L = _lib.lib()
    recs = [("r0 c", "ACGTNacgtn", "IIIIIHHHHH"), ("r1", "A", "#"), ("r2", "", ""), ("r3 xyz", "GATTACAGATTACA", "ABCDEFGHIJKLMN")]
    chunk = "".join(f"@{n}\n{s}\n+\n{q}\n" for n, s, q in recs).encode()
    dev = torch.device("cuda", 0)
    n, nb = len(recs), len(chunk)
    d_in = torch.from_numpy(np.frombuffer(chunk, dtype=np.uint8).copy()).to(dev)
    scratch = torch.empty(int(L.cah_fastq_device_scratch_bytes(nb, n)) + 4096, dtype=torch.uint8, device=dev)
    info = torch.zeros(8, dtype=torch.int64, device=dev)
    rec6 = torch.empty((n, 6), dtype=torch.int64, device=dev)
    soff = torch.empty(n, dtype=torch.int64, device=dev)
    slen = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.check(L.cah_fastq_count_lines_device(d_in.data_ptr(), nb, scratch.data_ptr(), scratch.numel(), info.data_ptr(), None))
    torch.cuda.synchronize()
    _lib.check(L.cah_fastq_index_device(d_in.data_ptr(), nb, int(info[0]), n, scratch.data_ptr(), scratch.numel(), rec6.data_ptr(),
                                        soff.data_ptr(), slen.data_ptr(), info.data_ptr(), None))
    flags = torch.tensor([1, 1, 1, 0], dtype=torch.uint8, device=dev)
    _lib.check(L.cah_revcomp_in_place_device(d_in.data_ptr(), rec6.data_ptr(), n, None, slen.data_ptr(), flags.data_ptr(), None))
    comp = str.maketrans("ACGTNacgtn", "TGCANtgcan")
    turned = [(nm, s[::-1].translate(comp) if f else s, q[::-1] if f else q) for (nm, s, q), f in zip(recs, [1, 1, 1, 0])]
    # the formatter with the suffix, every record whole
    beg = torch.zeros(n, dtype=torch.int32, device=dev)
    end = slen.clone()
    keep = torch.ones(n, dtype=torch.uint8, device=dev)
    out = torch.empty(nb + 64 * n, dtype=torch.uint8, device=dev)
    _lib.check(L.cah_fastq_format_suffix_device(d_in.data_ptr(), rec6.data_ptr(), n, beg.data_ptr(), end.data_ptr(), keep.data_ptr(),
                                                flags.data_ptr(), b" rc", 3, scratch.data_ptr(), scratch.numel(), nb, out.data_ptr(),
                                                out.numel(), info.data_ptr(), None))
    torch.cuda.synchronize()
    got = bytes(out[: int(info[3])].cpu().numpy())
    want = "".join(f"@{nm}{' rc' if f else ''}\n{s}\n+\n{q}\n" for (nm, s, q), f in zip(turned, [1, 1, 1, 0])).encode()
    assert got == want, (got, want)
    # info rows: record 0 and 3 matched (errors, rstart, rstop), 1 and 2 not; the rc column says which were turned
    out6 = torch.zeros((n, 6), dtype=torch.int32, device=dev)
    out6[0] = torch.tensor([0, 4, 2, 6, 4, 1], dtype=torch.int32)
    out6[3] = torch.tensor([0, 7, 7, 14, 7, 0], dtype=torch.int32)
    status = torch.tensor([1, 0, 0, 1], dtype=torch.uint8, device=dev)
    best = torch.tensor([1, 0, 0, 0], dtype=torch.int32, device=dev)
    names = torch.from_numpy(np.frombuffer(b"firstsecond2", dtype=np.uint8).copy()).to(dev)
    name_off = torch.tensor([0, 5, 12], dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    iout = torch.empty(nb + n * 64, dtype=torch.uint8, device=dev)
    _lib.check(L.cah_info_format_device(d_in.data_ptr(), rec6.data_ptr(), n, out6.data_ptr(), status.data_ptr(), best.data_ptr(), 1, None,
                                        beg.data_ptr(), end.data_ptr(), names.data_ptr(), name_off.data_ptr(), 2, flags.data_ptr(),
                                        b" rc", 3, scratch.data_ptr(), scratch.numel(), nb, iout.data_ptr(), iout.numel(),
                                        total.data_ptr(), None))
    torch.cuda.synchronize()
    got = bytes(iout[: int(total[0])].cpu().numpy()).decode()
    s0, q0 = turned[0][1], turned[0][2]
    s3, q3 = turned[3][1], turned[3][2]
    want = (f"r0 c rc\t1\t2\t6\t{s0[:2]}\t{s0[2:6]}\t{s0[6:]}\tsecond2\t{q0[:2]}\t{q0[2:6]}\t{q0[6:]}\t1\n"
            f"r1\t-1\t{turned[1][1]}\t{turned[1][2]}\n"
            f"r2\t-1\t\t\n"
            f"r3 xyz\t0\t7\t14\t{s3[:7]}\t{s3[7:14]}\t{s3[14:]}\tfirst\t{q3[:7]}\t{q3[7:14]}\t{q3[14:]}\t0\n")
    assert got == want, (got, want)
    # two rounds (--times 2): record 3 matches a 3' adapter, then a 5' adapter in what that left; record 0 only in round 1
    o2 = torch.zeros((2, n, 6), dtype=torch.int32, device=dev)
    o2[0, 0] = out6[0]
    o2[0, 3] = torch.tensor([0, 4, 10, 14, 4, 0], dtype=torch.int32)          # round 1: read[10:14], 3' -> keeps read[:10]
    o2[1, 3] = torch.tensor([0, 3, 1, 4, 3, 1], dtype=torch.int32)            # round 2 on read[:10]: [1:4], 5' -> keeps [4:10]
    st2 = torch.tensor([[1, 0, 0, 1], [0, 0, 0, 1]], dtype=torch.uint8, device=dev)
    b2 = torch.tensor([[1, 0, 0, 0], [0, 0, 0, 1]], dtype=torch.int32, device=dev)
    kinds = torch.tensor([0, 1], dtype=torch.uint8, device=dev)               # "first": 3', "second2": 5'
    _lib.check(L.cah_info_format_device(d_in.data_ptr(), rec6.data_ptr(), n, o2.data_ptr(), st2.data_ptr(), b2.data_ptr(), 2,
                                        kinds.data_ptr(), beg.data_ptr(), end.data_ptr(), names.data_ptr(), name_off.data_ptr(), 2,
                                        None, None, 0, scratch.data_ptr(), scratch.numel(), nb, iout.data_ptr(), iout.numel(),
                                        total.data_ptr(), None))
    torch.cuda.synchronize()
    got = bytes(iout[: int(total[0])].cpu().numpy()).decode()
    cur = s3[:10]
    T, NL = chr(9), chr(10)
    want = (T.join(["r0 c", "1", "2", "6", s0[:2], s0[2:6], s0[6:], "second2", q0[:2], q0[2:6], q0[6:], ""]) + NL
            + T.join(["r1", "-1", turned[1][1], turned[1][2]]) + NL
            + T.join(["r2", "-1", "", ""]) + NL
            + T.join(["r3 xyz", "0", "10", "14", s3[:10], s3[10:14], "", "first", q3[:10], q3[10:14], "", ""]) + NL
            + T.join(["r3 xyz", "1", "1", "4", cur[:1], cur[1:4], cur[4:], "second2", q3[:1], q3[1:4], q3[4:10], ""]) + NL)
    assert got == want, (got, want)
    # argument errors: a suffix longer than CAH_MAX_NAME_SUFFIX, missing pointers, negative counts, no round
    for rc in (L.cah_fastq_format_suffix_device(d_in.data_ptr(), rec6.data_ptr(), n, beg.data_ptr(), end.data_ptr(), keep.data_ptr(),
                                                flags.data_ptr(), b"x" * 40, 40, scratch.data_ptr(), scratch.numel(), nb,
                                                out.data_ptr(), out.numel(), info.data_ptr(), None),
               L.cah_revcomp_in_place_device(d_in.data_ptr(), rec6.data_ptr(), n, None, None, flags.data_ptr(), None),
               L.cah_revcomp_in_place_device(d_in.data_ptr(), rec6.data_ptr(), -1, None, slen.data_ptr(), flags.data_ptr(), None),
               L.cah_info_format_device(d_in.data_ptr(), rec6.data_ptr(), n, None, status.data_ptr(), best.data_ptr(), 1, None, beg.data_ptr(),
                                        end.data_ptr(), names.data_ptr(), name_off.data_ptr(), 2, None, None, 0, scratch.data_ptr(),
                                        scratch.numel(), nb, iout.data_ptr(), iout.numel(), total.data_ptr(), None),
               L.cah_info_format_device(d_in.data_ptr(), rec6.data_ptr(), n, out6.data_ptr(), status.data_ptr(), best.data_ptr(), 0, None,
                                        beg.data_ptr(), end.data_ptr(), names.data_ptr(), name_off.data_ptr(), 2, None, None, 0,
                                        scratch.data_ptr(), scratch.numel(), nb, iout.data_ptr(), iout.numel(), total.data_ptr(), None)):
        assert rc == _lib.CAH_EINVAL, rc
    assert L.cah_revcomp_in_place_device(d_in.data_ptr(), rec6.data_ptr(), 0, None, None, None, None) == 0      # nothing to do
