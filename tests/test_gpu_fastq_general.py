"""The device-side FASTQ path beyond adapter trimming (gpu_pipeline.py, "the general way"): the modifiers of
reference cli.py:938-973 chained on windows into the raw chunk in HBM, read pairs, several GPUs.

  * every single-end and paired command-line golden of the reference (tests/golden/fastq, tests/golden/paired)
    through trim_fastq_gpu / trim_fastq_gpu_paired, byte for byte;
  * 5000 random reads through -u, -q, two adapters with --times 2, --poly-a, -l, --max-ee and -m against a per-read
    restatement of the reference's modifier chain whose adapter step is driven by the ORACLE (not by
    pipeline.trim_fastq) and whose quality / poly-A steps are restated here and pinned to tests/golden/qualtrim.json;
  * devices="all": every visible GPU gets chunks, the bytes equal the one-device run.
GPU only."""
import io
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FQ = os.path.join(HERE, "golden", "fastq")
PD = os.path.join(HERE, "golden", "paired")
KINDS = {"-a": "back", "-g": "front", "-b": "anywhere"}


def _strip_trailing_space(b: bytes) -> bytes:
    return b"\n".join(line.rstrip(b" \t") for line in b.split(b"\n"))


def test_single_end_goldens_through_the_device_path(hip):
    from cutadapt_amd.gpu_pipeline import trim_fastq_gpu
    from cutadapt_amd.pipeline import adapter_from_spec
    manifest = json.load(open(os.path.join(FQ, "manifest.json")))
    ways, by_name = {}, {}
    for case in manifest:
        opts = dict(case["options"])
        params = {"max_errors": opts.pop("max_errors")} if "max_errors" in opts else {}
        if "quality_cutoff" in opts:
            opts["quality_cutoff"] = tuple(opts["quality_cutoff"])
        for chunk_bytes in (4 << 20, 512):
            ads = [adapter_from_spec(spec, KINDS[opt], **params) for opt, spec in case["adapters"]]
            out, info = io.BytesIO(), io.BytesIO()
            stats = trim_fastq_gpu(os.path.join(FQ, case["input"]), out, ads, chunk_bytes=chunk_bytes, threads=2,
                                   info_file=info if case["info"] else None, **opts)
            if case["expected"]:
                assert out.getvalue() == open(os.path.join(FQ, case["expected"]), "rb").read(), (case["name"], chunk_bytes)
            if case["info"]:
                expected = open(os.path.join(FQ, case["info"]), "rb").read()
                assert _strip_trailing_space(info.getvalue()) == _strip_trailing_space(expected), (case["name"], chunk_bytes)
        ways[stats["way"]] = ways.get(stats["way"], 0) + 1
        by_name[case["name"]] = stats["way"]
        if case["name"] == "max_expected_errors":
            assert stats["too_many_expected_errors"] == 2
        if case["name"] == "revcomp_normalized":
            assert stats["reverse_complemented"] == 2                   # reference test_commandline.py:834
    # what still takes the general way: info files of several rounds or of linked adapters
    # (round 6: mask / lowercase with single adapters are marked in place on the device, cah_mark_reads_device; --revcomp is
    # matched in both orientations and turned around there, cah_revcomp_in_place_device; the info rows of one round of single
    # adapters are formatted there, cah_info_format_device)
    general = {k for k, v in by_name.items() if v == "general"}
    assert general <= {"linked_info_file", "linked_multiple"}, general
    assert by_name.get("revcomp_normalized") == "all-device" and by_name.get("info_file") == "all-device", by_name
    assert by_name.get("info_file_times") == "all-device", by_name
    assert by_name.get("action_mask") == "all-device", by_name          # (action_lowercase is a FASTA golden: host-parsed)
    # (14 of the 33 goldens are FASTA files: parsed on the host -- a sequence may span lines -- and matched in batches)
    fastq = {k for k, v in by_name.items() if v != "host-parsed (FASTA)"}
    assert len(fastq) >= 19 and ways.get("all-device", 0) >= len(fastq) - 4, (ways, general)


def test_paired_goldens_through_the_device_path(hip):
    from cutadapt_amd.gpu_pipeline import trim_fastq_gpu_paired
    from cutadapt_amd.pipeline import adapter_from_spec
    manifest = json.load(open(os.path.join(PD, "manifest.json")))
    assert len(manifest) >= 20

    def mate(opts):
        opts = dict(opts)
        params = opts.pop("params", {})
        ads = [adapter_from_spec(spec, KINDS[o], **params) for o, spec in opts.pop("adapters", [])]
        if "quality_cutoff" in opts:
            opts["quality_cutoff"] = tuple(opts["quality_cutoff"])
        return dict(adapters=ads, **opts)

    on_device = 0
    for case in manifest:
        for chunk_bytes in (4 << 20, 300):
            o1, o2 = io.BytesIO(), io.BytesIO()
            stats = trim_fastq_gpu_paired(os.path.join(PD, case["in1"]), os.path.join(PD, case["in2"]), o1, o2,
                                          mate(case["r1"]), mate(case["r2"]), chunk_bytes=chunk_bytes, devices="all",
                                          **case["top"])
            assert o1.getvalue() == open(os.path.join(PD, case["exp1"]), "rb").read(), (case["name"], chunk_bytes, 1)
            assert o2.getvalue() == open(os.path.join(PD, case["exp2"]), "rb").read(), (case["name"], chunk_bytes, 2)
        exp = open(os.path.join(PD, case["exp1"]), "rb").read()
        n_out = exp.count(b">") if exp.startswith(b">") else exp.count(b"\n") // 4
        assert stats["pairs_written"] == n_out, case["name"]
        on_device += "per_device" in stats
    assert on_device >= 18, on_device                        # (two of the cases are FASTA: parsed on the host)
    # mates that do not pair up are reported
    r1 = open(os.path.join(PD, "in_paired.1.fastq"), "rb").read()
    r2 = open(os.path.join(PD, "in_paired.2.fastq"), "rb").read()
    with pytest.raises(ValueError, match="improperly paired"):
        trim_fastq_gpu_paired(io.BytesIO(r1), io.BytesIO(r2[: len(r2) // 2].rsplit(b"@", 1)[0]), io.BytesIO(), io.BytesIO())


# ---- per-read restatements of the modifiers around the adapter step (pinned to the reference's known answers) ----
def quality_trim_index(q: str, cutoff_front: int, cutoff_back: int, base: int = 33):
    """reference qualtrim.pyx:22-70"""
    n = len(q)
    start, stop = 0, n
    s = max_qual = 0
    for i in range(n):
        s += cutoff_front - (ord(q[i]) - base)
        if s < 0:
            break
        if s > max_qual:
            max_qual, start = s, i + 1
    s = max_qual = 0
    for i in reversed(range(n)):
        s += cutoff_back - (ord(q[i]) - base)
        if s < 0:
            break
        if s > max_qual:
            max_qual, stop = s, i
    if start >= stop:
        start, stop = 0, 0
    return start, stop


def poly_a_trim_index(s: str, revcomp: bool = False) -> int:
    """reference qualtrim.pyx:116-165"""
    n = len(s)
    if revcomp:
        best_index, best_score, score, errors = 0, 0, 0, 0
        for i in range(n):
            if s[i] == "T":
                score += 1
            else:
                score -= 2
                errors += 1
            if score > best_score and errors * 5 <= i + 1:
                best_score, best_index = score, i + 1
        return best_index
    best_index, best_score, score, errors = n, 0, 0, 0
    for i in reversed(range(n)):
        if s[i] == "A":
            score += 1
        else:
            score -= 2
            errors += 1
        if score > best_score and errors * 5 <= n - i:
            best_score, best_index = score, i
    return best_index


def test_restatements_agree_with_the_reference_known_answers(golden):
    g = golden("qualtrim.json")
    for q, front, back, base, want in g["quality_trim"]:
        assert list(quality_trim_index(q, front, back, base)) == want
    for s, revcomp, want in g["poly_a"]:
        got = poly_a_trim_index(s, revcomp)
        # the reference ignores tails shorter than three characters
        if revcomp:
            got = got if got > 2 else 0
        else:
            got = got if got < len(s) - 2 else len(s)
        assert got == want, (s, revcomp)


def _poly_a(s, revcomp=False):
    i = poly_a_trim_index(s, revcomp)
    if revcomp:
        return i if i > 2 else 0
    return i if i < len(s) - 2 else len(s)


def test_random_reads_against_the_modifier_chain_over_oracle_results(hip, orc, golden):
    from cutadapt_amd import adapters as A
    from cutadapt_amd.gpu_pipeline import trim_fastq_gpu
    from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
    rng = random.Random(77)
    table = [float.fromhex(x) for x in golden("qualtrim.json")["error_table"]]
    ad_back, ad_front = "ACGTTGCAAGTC", "GGATCCAATC"
    recs = []
    for i in range(5000):
        s = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 90)))
        for _ in range(rng.randint(0, 2)):
            p = rng.randint(0, len(s))
            piece = rng.choice((ad_back, ad_front))
            if rng.random() < 0.3:
                q = rng.randrange(len(piece))
                piece = piece[:q] + rng.choice("ACGT") + piece[q + 1:]
            s = s[:p] + piece + s[p:]
        if rng.random() < 0.3:
            p = rng.randint(0, len(s))
            s = s[:p] + "A" * rng.randint(3, 25) + s[p:]
        lo = rng.choice((33, 40, 53))
        q = "".join(chr(rng.randint(lo, 73)) for _ in s)
        if rng.random() < 0.5 and len(q) > 6:                  # a bad tail (and sometimes a bad head)
            k = rng.randint(1, 6)
            q = q[:-k] + "".join(chr(rng.randint(33, 38)) for _ in range(k))
            if rng.random() < 0.3:
                q = "".join(chr(rng.randint(33, 38)) for _ in range(3)) + q[3:]
        recs.append((f"r{i}", s, q))
    data = "".join(f"@{n}\n{s}\n+\n{q}\n" for n, s, q in recs).encode()
    finders = {
        "back": orc.KmerFinder(create_positions_and_kmers(ad_back, 3, 0.1, True, False), False, False),
        "front": orc.KmerFinder(create_positions_and_kmers(ad_front, 3, 0.1, False, True), False, False),
    }
    CUT, QCUT, LENGTH, MAXEE, MINLEN = 2, (8, 12), 60, 1.5, 10
    # both ways through a chunk (an info file makes it the general one: host-side window arithmetic between the kernels)
    for TIMES, way in ((2, "general"), (2, "all-device"), (1, "all-device")):

        def chain(name, s, q):
            """cli.py:938-973: -u, -q, adapters (--times 2), --poly-a, -l; then --max-ee and -m (cli.py:735-912)"""
            s, q = s[CUT:], q[CUT:]
            a, b = quality_trim_index(q, QCUT[0], QCUT[1])
            s, q = s[a:b], q[a:b]
            matched = False
            for _ in range(TIMES):
                best = None
                for seq, kind in ((ad_back, "back"), (ad_front, "front")):
                    if not finders[kind].kmers_present(s):
                        continue
                    t = orc.Aligner(seq, 0.1, flags=14 if kind == "back" else 11, wildcard_ref=False, min_overlap=3).locate(s)
                    if t is not None and (best is None or t[4] > best[0][4] or (t[4] == best[0][4] and t[5] < best[0][5])):
                        best = (t, kind)
                if best is None:
                    break
                matched = True
                t, kind = best
                s, q = (s[:t[2]], q[:t[2]]) if kind == "back" else (s[t[3]:], q[t[3]:])
            i = _poly_a(s)
            s, q = s[:i], q[:i]
            s, q = s[:LENGTH], q[:LENGTH]
            ee = 0.0
            for c in q:
                ee += table[ord(c) - 33]
            return s, q, ee, matched

        want, near = [], 0
        n_short = n_ee = n_matched = 0
        for name, s, q in recs:
            s2, q2, ee, matched = chain(name, s, q)
            n_matched += matched
            if len(s2) < MINLEN:                                     # too short is checked first (cli.py:735-912)
                n_short += 1
                continue
            if abs(ee - MAXEE) < 1e-9:
                near += 1
            if ee > MAXEE:
                n_ee += 1
                continue
            want.append(f"@{name}\n{s2}\n+\n{q2}\n")
        assert near == 0 and n_short > 100 and n_ee > 100 and n_matched > 1000
        ads = [A.BackAdapter(ad_back), A.FrontAdapter(ad_front)]
        for chunk_bytes, threads in ((1 << 20, 2), (40000, 3)):
            out = io.BytesIO()
            stats = trim_fastq_gpu(np.frombuffer(data, dtype=np.uint8), out, ads, cut=[CUT], quality_cutoff=QCUT,
                                   times=TIMES, poly_a=True, length=LENGTH, max_expected_errors=MAXEE,
                                   minimum_length=MINLEN, chunk_bytes=chunk_bytes, threads=threads,
                                   _general=way == "general")
            assert stats["way"] == way
            assert out.getvalue() == "".join(want).encode(), chunk_bytes
            assert stats["reads"] == len(recs) and stats["with_adapters"] == n_matched
            assert stats["filtered"].get("too_short", 0) == n_short
            assert stats["too_many_expected_errors"] == n_ee


def test_every_visible_gpu_is_fed(hip, tmp_path):
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd.gpu_pipeline import trim_fastq_gpu
    from test_gpu_fastq_device import _fastq
    rng = random.Random(5)
    ad = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
    data = _fastq(rng, 20000, [ad])
    path = tmp_path / "reads.fastq"
    path.write_bytes(data)
    n_dev = torch.cuda.device_count()
    visible = sorted(f"cuda:{i}" for i in range(n_dev))
    chunk = len(data) // (4 * n_dev)                            # at least four chunks per device
    for way, opts in (("all-device", {}), ("all-device", {"quality_cutoff": (0, 10), "times": 2, "action": "mask"}),
                      ("general", {"quality_cutoff": (0, 10), "times": 2, "action": "mask", "revcomp": True})):
        one = io.BytesIO()
        s1 = trim_fastq_gpu(str(path), one, [A.BackAdapter(ad)], chunk_bytes=chunk, threads=2, devices=[0], **opts)
        assert s1["way"] == way
        for source in (str(path), np.frombuffer(data, dtype=np.uint8), io.BytesIO(data)):
            every = io.BytesIO()
            sa = trim_fastq_gpu(source, every, [A.BackAdapter(ad)], chunk_bytes=chunk, threads=2, devices="all", **opts)
            assert sa["devices_used"] == visible
            assert every.getvalue() == one.getvalue()
            assert sorted(sa["per_device"]) == visible
            per = sa["per_device"]
            assert all(d["chunks"] >= 4 for d in per.values()), per     # round-robin: nobody is left out
            assert sum(d["bytes_in"] for d in per.values()) == len(data)
            # every device indexed reads of its own, and together they indexed each record once
            assert all(d["reads_in"] > 0 for d in per.values()), per
            assert sum(d["reads_in"] for d in per.values()) == 20000 == sa["reads"], per
            assert (sa["reads"], sa["with_adapters"], sa["bp_out"]) == (s1["reads"], s1["with_adapters"], s1["bp_out"])


def _random_records(rng, n, ads, polya=0.3):
    recs = []
    for i in range(n):
        s = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 80)))
        for _ in range(rng.randint(0, 2)):
            p = rng.randint(0, len(s))
            s = s[:p] + rng.choice(ads) + s[p:]
        if rng.random() < polya:
            s += "A" * rng.randint(3, 20)
        q = "".join(chr(rng.randint(40, 73)) for _ in s)
        if rng.random() < 0.6 and len(q) > 8:
            k = rng.randint(1, 6)
            q = q[:-k] + "".join(chr(rng.randint(33, 37)) for _ in range(k))
            if rng.random() < 0.5:
                q = "".join(chr(rng.randint(33, 37)) for _ in range(2)) + q[2:]
        recs.append((f"r{i} c", s, q))
    return recs


def _oracle_rounds(orc, finders, ad_back, ad_front, s, times):
    """AdapterCutter.match_and_trim's search loop (reference modifiers.py:209-251) over oracle results:
    -> [(out6, kind)] of every round, coordinates relative to the read that round saw"""
    rounds = []
    for _ in range(times):
        best = None
        for seq, kind in ((ad_back, "back"), (ad_front, "front")):
            if not finders[kind].kmers_present(s):
                continue
            t = orc.Aligner(seq, 0.1, flags=14 if kind == "back" else 11, wildcard_ref=False, min_overlap=3).locate(s)
            if t is not None and (best is None or t[4] > best[0][4] or (t[4] == best[0][4] and t[5] < best[0][5])):
                best = (t, kind)
        if best is None:
            break
        rounds.append(best)
        t, kind = best
        s = s[:t[2]] if kind == "back" else s[t[3]:]
    return rounds


def test_info_file_and_marking_actions_next_to_other_modifiers(hip, orc):
    """The two combinations that used to be refused.  --info-file with -q / --poly-a / -l: InfoFileWriter (reference
    steps.py:232-253) cuts the read AS IT CAME IN at the coordinates the matches have on the quality-trimmed read and
    prints unmatched reads as they are written.  --action mask / lowercase with -q and --poly-a: the poly-A trimmer
    sees the marked characters (modifiers.py:170-198 then :861-879)."""
    from cutadapt_amd import adapters as A
    from cutadapt_amd.gpu_pipeline import trim_fastq_gpu
    from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
    from cutadapt_amd.pipeline import trim_fastq
    rng = random.Random(88)
    ad_back, ad_front = "ACGTTGCAAGTC", "GGATCCAATC"
    finders = {
        "back": orc.KmerFinder(create_positions_and_kmers(ad_back, 3, 0.1, True, False), False, False),
        "front": orc.KmerFinder(create_positions_and_kmers(ad_front, 3, 0.1, False, True), False, False),
    }
    recs = _random_records(rng, 1500, [ad_back, ad_front])
    data = "".join(f"@{n}\n{s}\n+\n{q}\n" for n, s, q in recs).encode()
    QCUT, TIMES, LENGTH = (6, 10), 2, 50
    # ---- info file -----------------------------------------------------------------------------------------------
    want_info, want_out = [], []
    for name, s, q in recs:
        a, b = quality_trim_index(q, *QCUT)
        ts, tq = s[a:b], q[a:b]
        rounds = _oracle_rounds(orc, finders, ad_back, ad_front, ts, TIMES)
        cur_s, cur_q = s, q                                    # info.original_read
        for t, kind in rounds:
            ts, tq = (ts[:t[2]], tq[:t[2]]) if kind == "back" else (ts[t[3]:], tq[t[3]:])
            want_info.append("\t".join([name, str(t[5]), str(t[2]), str(t[3]), cur_s[:t[2]], cur_s[t[2]:t[3]], cur_s[t[3]:],
                                        "b" if kind == "back" else "f", cur_q[:t[2]], cur_q[t[2]:t[3]], cur_q[t[3]:], ""]))
            cur_s, cur_q = (cur_s[:t[2]], cur_q[:t[2]]) if kind == "back" else (cur_s[t[3]:], cur_q[t[3]:])
        i = _poly_a(ts)
        ts, tq = ts[:i][:LENGTH], tq[:i][:LENGTH]
        if not rounds:
            want_info.append("\t".join([name, "-1", ts, tq]))
        want_out.append(f"@{name}\n{ts}\n+\n{tq}\n")
    ads = [A.BackAdapter(ad_back, name="b"), A.FrontAdapter(ad_front, name="f")]
    for fn, kw in ((trim_fastq_gpu, dict(threads=2)), (trim_fastq, {})):
        for chunk_bytes in (1 << 20, 9000):
            out, info = io.BytesIO(), io.BytesIO()
            fn(io.BytesIO(data), out, ads, quality_cutoff=QCUT, times=TIMES, poly_a=True, length=LENGTH, info_file=info,
               chunk_bytes=chunk_bytes, **kw)
            assert out.getvalue() == "".join(want_out).encode(), (fn.__name__, chunk_bytes)
            got = info.getvalue().decode().split("\n")[:-1]
            assert got == want_info, (fn.__name__, [(g, w) for g, w in zip(got, want_info) if g != w][:2])
    # ---- mask / lowercase ----------------------------------------------------------------------------------------
    for action in ("mask", "lowercase"):
        want_out = []
        for name, s, q in recs:
            a, b = quality_trim_index(q, *QCUT)
            ts, tq = s[a:b], q[a:b]
            if action == "lowercase":
                ts = ts.upper()
            rounds = _oracle_rounds(orc, finders, ad_back, ad_front, ts, TIMES)
            lo, hi = 0, len(ts)                                # remainder(matches), modifiers.py:170-198
            for t, kind in rounds:
                if kind == "back":
                    hi = lo + t[2]
                else:
                    lo = lo + t[3]
            if rounds and action == "mask":
                ts = "N" * lo + ts[lo:hi] + "N" * (len(ts) - hi)
            elif rounds:
                ts = ts[:lo].lower() + ts[lo:hi].upper() + ts[hi:].lower()
            i = _poly_a(ts)
            ts, tq = ts[:i][:LENGTH], tq[:i][:LENGTH]
            want_out.append(f"@{name}\n{ts}\n+\n{tq}\n")
        for fn, kw in ((trim_fastq_gpu, dict(threads=2)), (trim_fastq, {})):
            out = io.BytesIO()
            fn(io.BytesIO(data), out, ads, quality_cutoff=QCUT, times=TIMES, poly_a=True, length=LENGTH, action=action,
               chunk_bytes=20000, **kw)
            got, want = out.getvalue().decode().split("\n"), "".join(want_out).split("\n")
            assert got == want, (action, fn.__name__, [(g, w) for g, w in zip(got, want) if g != w][:2])


def test_read_pairs_on_the_all_device_way_equal_the_host_pipeline(hip):
    """trim_fastq_gpu_paired without actions / linked adapters runs on the all-device way: both mates trimmed,
    matched and formatted on the GPU, the pair filter as element-wise operations.  Bytes and counters against
    pipeline.trim_fastq_paired (which the reference's paired goldens pin) over option sets that exercise every
    modifier, every --pair-filter mode, per-mate length limits and the discards."""
    from cutadapt_amd import adapters as A
    from cutadapt_amd.gpu_pipeline import trim_fastq_gpu_paired
    from cutadapt_amd.pipeline import trim_fastq_paired
    from test_gpu_fastq_device import _fastq
    rng = random.Random(99)
    ad1, ad2 = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"
    n = 4000
    f1 = _fastq(rng, n, [ad1], twice=True)
    f2 = _fastq(rng, n, [ad2], twice=True).replace(b"@read", b"@mate")
    cases = [
        (dict(adapters=[A.BackAdapter(ad1)]), dict(adapters=[A.BackAdapter(ad2)]), dict(minimum_length=20)),
        (dict(adapters=[A.BackAdapter(ad1)], quality_cutoff=(0, 20)), dict(adapters=[A.BackAdapter(ad2)], quality_cutoff=(5, 15)),
         dict(minimum_length=(30, 20), maximum_length=(150, None), pair_filter="both")),
        (dict(adapters=[A.BackAdapter(ad1), A.FrontAdapter(ad2[:15])], poly_a=True, cut=[2]), dict(poly_a=True, length=70),
         dict(minimum_length=25, pair_filter="first")),
        (dict(adapters=[A.BackAdapter(ad1)], max_expected_errors=2.0), dict(adapters=[A.BackAdapter(ad2)], max_expected_errors=4.0, nextseq_trim=15),
         dict(discard_untrimmed=True)),
        (dict(adapters=[A.BackAdapter(ad1)], length=-60), dict(adapters=[A.AnywhereAdapter(ad2[:20])], cut=[-3, 4]),
         dict(discard_trimmed=True, pair_filter="both")),
        (dict(quality_cutoff=(0, 25)), dict(adapters=[A.BackAdapter(ad2)]), dict(discard_untrimmed=True, minimum_length=10)),
        (dict(adapters=[A.BackAdapter(ad1), A.FrontAdapter(ad2[:15])], times=2),
         dict(adapters=[A.BackAdapter(ad2), A.BackAdapter(ad1[:12])], times=3, quality_cutoff=(0, 20)), dict(minimum_length=15)),
    ]
    for ci, (r1, r2, top) in enumerate(cases):
        w1, w2 = io.BytesIO(), io.BytesIO()
        ws = trim_fastq_paired(io.BytesIO(f1), io.BytesIO(f2), w1, w2, dict(r1), dict(r2), **top)
        for chunk_bytes in (1 << 20, 50_000):
            g1, g2 = io.BytesIO(), io.BytesIO()
            gs = trim_fastq_gpu_paired(io.BytesIO(f1), io.BytesIO(f2), g1, g2, dict(r1), dict(r2), chunk_bytes=chunk_bytes,
                                       threads=2, devices="all", **top)
            assert gs["way"] == "all-device", ci
            assert g1.getvalue() == w1.getvalue() and g2.getvalue() == w2.getvalue(), (ci, chunk_bytes)
            assert (gs["pairs"], gs["pairs_written"]) == (ws["pairs"], ws["pairs_written"]) == (n, gs["pairs_written"]), ci
            assert gs["with_adapters"] == ws["with_adapters"], ci
            assert gs["filtered"] == {k: v for k, v in ws["filtered"].items() if v}, (ci, gs["filtered"], ws["filtered"])
            assert gs["too_many_expected_errors"] == ws["trimmers"][0].too_many_expected_errors, ci
            assert gs["bp_out"] == (ws["trimmers"][0].bp_out, ws["trimmers"][1].bp_out), ci
            assert gs["quality_trimmed_bases"] == tuple(t.quality_trimmed_bases for t in ws["trimmers"]), ci
    # what the all-device way does not serve goes the general way and says so
    gs = trim_fastq_gpu_paired(io.BytesIO(f1), io.BytesIO(f2), io.BytesIO(), io.BytesIO(),
                               dict(adapters=[A.BackAdapter(ad1)], action="mask"), dict(adapters=[A.BackAdapter(ad2)]))
    assert gs["way"] == "general"


def test_general_paired_way_returns_its_pinned_input_buffers(hip):
    """Round-4 review: the general paired way read into pinned buffers (_PINNED_INPUT) but handed them back to the
    pageable pool, so every block pinned fresh memory and the pinned pool kept all of it for the life of the process.
    Many small blocks through --pair-adapters: the pool's bookkeeping stays bounded and shrinks at the job's end."""
    from cutadapt_amd import adapters as A
    from cutadapt_amd import gpu_pipeline as G
    from test_gpu_fastq_device import _fastq
    rng = random.Random(5)
    ad1, ad2 = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"
    n = 6000
    f1 = _fastq(rng, n, [ad1])
    f2 = _fastq(rng, n, [ad2]).replace(b"@read", b"@mate")
    before = len(G._PINNED_INPUT._owner)
    gs = G.trim_fastq_gpu_paired(io.BytesIO(f1), io.BytesIO(f2), io.BytesIO(), io.BytesIO(),
                                 dict(adapters=[A.BackAdapter(ad1)]), dict(adapters=[A.BackAdapter(ad2)]),
                                 pair_adapters=True, chunk_bytes=20_000, threads=2)
    assert gs["way"] == "general" and gs["pairs"] == n
    blocks = len(f1) // 20_000
    assert blocks > 40
    # far fewer pinned buffers than blocks, and the job's end trimmed the free list (trim keeps 8)
    assert len(G._PINNED_INPUT._owner) - before <= 16, (len(G._PINNED_INPUT._owner), before, blocks)
    assert len(G._PINNED_INPUT._free) <= 8
