"""FASTQ chunk scanner / packer / writer (CPU) and the batch pipeline against the reference's
command-line golden files (GPU)."""
import ctypes as C
import gzip
import io
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FQ = os.path.join(ROOT, "tests", "golden", "fastq")


def scan(data: bytes, final=True, max_rec=None):
    from cutadapt_amd import _lib
    buf = np.frombuffer(data, dtype=np.uint8)
    max_rec = max_rec if max_rec is not None else data.count(b"\n") // 4 + 2
    rec = np.empty((max_rec, 6), dtype=np.int64)
    n, consumed = C.c_int64(0), C.c_int64(0)
    _lib.check(_lib.lib().cah_fastq_scan(buf.ctypes.data if len(data) else None, len(data), int(final), max_rec,
                                         rec.ctypes.data, C.byref(n), C.byref(consumed)))
    return rec[:n.value], consumed.value


def test_scan_records_and_line_endings():
    data = b"@r1 comment\nACGT\n+\nIIII\n@r2\r\nAC\r\n+r2\r\n#!\r\n@r3\n\n+\n\n@r4\nTT\n+\nAB"
    rec, consumed = scan(data)
    assert consumed == len(data) and len(rec) == 4
    fields = [[data[r[0]:r[1]], data[r[2]:r[3]], data[r[4]:r[5]]] for r in rec]
    assert fields == [[b"r1 comment", b"ACGT", b"IIII"], [b"r2", b"AC", b"#!"], [b"r3", b"", b""], [b"r4", b"TT", b"AB"]]
    # a chunk that ends inside a record: only complete records are consumed
    rec, consumed = scan(data[:30], final=False)
    assert len(rec) == 1 and data[consumed:consumed + 3] == b"@r2"
    rec, consumed = scan(b"", final=True)
    assert len(rec) == 0 and consumed == 0


@pytest.mark.parametrize("bad", [b"r1\nACGT\n+\nIIII\n", b"@r1\nACGT\n-\nIIII\n", b"@r1\nACGT\n+\nIII\n", b"@r1\nACGT\n+\n"])
def test_scan_rejects_malformed(bad):
    with pytest.raises(ValueError):
        scan(bad, final=True)


def test_pack_and_write_roundtrip():
    from cutadapt_amd.pipeline import read_fastq_chunks
    rng = np.random.default_rng(5)
    recs = []
    for i in range(2000):
        n = int(rng.integers(0, 90))
        s = "".join(rng.choice(list("ACGTN"), n))
        q = "".join(chr(33 + int(x)) for x in rng.integers(0, 40, n))
        recs.append((f"read{i} x", s, q))
    data = "".join(f"@{a}\n{s}\n+\n{q}\n" for a, s, q in recs).encode()
    chunks = list(read_fastq_chunks(io.BytesIO(data), chunk_bytes=4096))      # many chunks, carried tails
    assert sum(len(c) for c in chunks) == 2000 and len(chunks) > 20
    out, k = b"", 0
    for c in chunks:
        seqs, offsets = c.pack_sequences()
        for j in range(len(c)):
            assert bytes(seqs[offsets[j]:offsets[j + 1]]).decode() == recs[k + j][1]
        lens = (offsets[1:] - offsets[:-1]).astype(np.int32)
        out += c.write_trimmed(np.zeros(len(c), np.int32), lens)
        k += len(c)
    assert out == data                                                       # identity trim
    c = chunks[0]
    seqs, offsets = c.pack_sequences()
    lens = (offsets[1:] - offsets[:-1]).astype(np.int32)
    half = c.write_trimmed(lens // 4, lens // 2, keep=(np.arange(len(c)) % 2).astype(np.uint8))
    lines = half.decode().split("\n")
    assert lines[0] == "@" + recs[1][0] and lines[1] == recs[1][1][len(recs[1][1]) // 4:len(recs[1][1]) // 2]
    assert lines[3] == recs[1][2][len(recs[1][1]) // 4:len(recs[1][1]) // 2]


def _strip_trailing_space(data: bytes) -> bytes:
    """diff --ignore-trailing-space, as the reference's info-file comparisons use"""
    return b"\n".join(line.rstrip() for line in data.split(b"\n"))


def test_fasta_scan_and_writers():
    """multi-line FASTA records, chunk boundaries, the three output modes and the info rows
    (host-only entry points: no GPU needed)"""
    from cutadapt_amd.pipeline import read_fastq_chunks
    data = b">r1 first\nACGTAC\nGTAA\n>r2\n\n>r3\r\nttgacc\r\n>r4\nAC"
    for chunk_bytes in (1 << 20, 7):
        chunks = list(read_fastq_chunks(io.BytesIO(data), chunk_bytes=chunk_bytes))
        names, seqs_ = [], []
        for c in chunks:
            seqs, offsets = c.pack_sequences()
            for j in range(len(c)):
                names.append(bytes(c.buf[c.rec[j, 0]:c.rec[j, 1]]).decode())
                seqs_.append(bytes(seqs[offsets[j]:offsets[j + 1]]).decode())
        assert names == ["r1 first", "r2", "r3", "r4"]
        assert seqs_ == ["ACGTACGTAA", "", "ttgacc", "AC"]
    c = list(read_fastq_chunks(io.BytesIO(data)))[0]        # r4 completes only at EOF: second chunk
    assert len(c) == 3
    beg, end = np.array([2, 0, 1], np.int32), np.array([5, 0, 4], np.int32)
    assert c.write_records(beg, end, mode=0) == b">r1 first\nGTA\n>r2\n\n>r3\ntga\n"
    assert c.write_records(beg, end, mode=1) == b">r1 first\nNNGTANNNNN\n>r2\n\n>r3\nNtgaNN\n"
    assert c.write_records(beg, end, mode=2) == b">r1 first\nacGTAcgtaa\n>r2\n\n>r3\ntTGAcc\n"
    assert c.write_records(beg, end, keep=np.array([0, 1, 1], np.uint8)) == b">r2\n\n>r3\ntga\n"
    rows = np.array([[0, 1, 2, 5, 0, 10, 0], [0, 0, 1, 2, 0, 2, 1], [2, 0, 0, 2, 1, 5, 0]], np.int64)
    info = c.write_info(rows, ["ad", "other"])
    assert info == (b"r1 first\t1\t2\t5\tAC\tGTA\tCGTAA\tad\t\t\t\t\n"
                    b"r1 first\t0\t1\t2\tA\tC\t\tother\t\t\t\t\n"
                    b"r2\t-1\t\t\n" b"r3\t0\t0\t2\t\ttg\tac\tad\t\t\t\t\n")
    fq = list(read_fastq_chunks(io.BytesIO(b"@q\nACGTT\n+\nIIHGF\n")))[0]
    assert fq.write_records(np.array([1], np.int32), np.array([3], np.int32), mode=1) == b"@q\nNCGNN\n+\nIIHGF\n"
    assert fq.write_info(np.array([[0, 0, 1, 3, 0, 5, 0]], np.int64), ["x"]) == b"q\t0\t1\t3\tA\tCG\tTT\tx\tI\tIH\tGF\t\n"
    with pytest.raises(ValueError):
        fq.write_info(np.array([[0, 0, 1, 9, 0, 5, 0]], np.int64), ["x"])


def test_raw_chunks_cut_at_record_starts():
    """the reader thread's cheap boundary search (cah_record_boundary) must give the same records
    as the parsing reader, also with '@' and '+' as first quality characters"""
    import random
    from cutadapt_amd.pipeline import read_fastq_chunks, read_raw_chunks, scan_chunk
    rng = random.Random(3)
    recs = []
    for i in range(300):
        n = rng.randint(0, 40)
        seq = "".join(rng.choice("ACGT") for _ in range(n))
        qual = "".join(rng.choice("@+I#5") for _ in range(n))
        recs.append((f"r{i} x", seq, qual))
    fq = "".join(f"@{h}\n{s}\n+\n{q}\n" for h, s, q in recs).encode()
    fa = "".join(f">{h}\n{s[:len(s) // 2]}\n{s[len(s) // 2:]}\n" for h, s, _ in recs).encode()
    for data in (fq, fa, fq[:-1]):
        for chunk_bytes in (64, 333, 1 << 20):
            ref_recs = []
            for c in read_fastq_chunks(io.BytesIO(data), chunk_bytes):
                seqs, off = c.pack_sequences()
                ref_recs += [(bytes(c.buf[c.rec[j, 0]:c.rec[j, 1]]), bytes(seqs[off[j]:off[j + 1]])) for j in range(len(c))]
            got, total = [], 0
            for raw, fasta in read_raw_chunks(io.BytesIO(data), chunk_bytes):
                total += len(raw)
                c = scan_chunk(raw, fasta)
                seqs, off = c.pack_sequences()
                got += [(bytes(c.buf[c.rec[j, 0]:c.rec[j, 1]]), bytes(seqs[off[j]:off[j + 1]])) for j in range(len(c))]
            assert total == len(data) and got == ref_recs and len(got) == 300


def test_adapter_specs():
    from cutadapt_amd import adapters as A
    from cutadapt_amd.pipeline import adapter_from_spec
    a = adapter_from_spec("name=ACGT", "back")
    assert isinstance(a, A.BackAdapter) and a.name == "name" and a.sequence == "ACGT"
    assert isinstance(adapter_from_spec("^ACGT", "front"), A.PrefixAdapter)
    assert isinstance(adapter_from_spec("ACGT$", "back"), A.SuffixAdapter)
    assert isinstance(adapter_from_spec("XACGT", "front"), A.NonInternalFrontAdapter)
    assert isinstance(adapter_from_spec("ACGTX", "back"), A.NonInternalBackAdapter)
    assert isinstance(adapter_from_spec("ACGT", "anywhere"), A.AnywhereAdapter)
    assert isinstance(adapter_from_spec("ACGT...", "back"), A.FrontAdapter)
    la = adapter_from_spec("l=^AAAA...TTTT", "back")
    assert isinstance(la, A.LinkedAdapter) and la.name == "l" and la.front_required and not la.back_required
    assert isinstance(la.front_adapter, A.PrefixAdapter) and isinstance(la.back_adapter, A.BackAdapter)
    lg = adapter_from_spec("AAAA...TTTT", "front")
    assert lg.front_required and lg.back_required and isinstance(lg.front_adapter, A.FrontAdapter)
    with pytest.raises(ValueError):
        adapter_from_spec("AAA...TTT", "anywhere")


@pytest.mark.gpu
def test_reference_commandline_goldens(hip):
    """The single-end adapter-trimming command lines of reference tests/test_commandline.py and
    tests/test_info_file.py (manifest.json lists test line, adapter options, AdapterCutter options,
    input, expected output and expected info file): every action, --times, linked and multiple
    adapters, --discard-(un)trimmed, FASTA and FASTQ, with one big and many tiny chunks."""
    from cutadapt_amd.pipeline import adapter_from_spec, trim_fastq
    manifest = json.load(open(os.path.join(FQ, "manifest.json")))
    assert len(manifest) >= 31
    kinds = {"-a": "back", "-g": "front", "-b": "anywhere"}
    for case in manifest:
        opts = dict(case["options"])
        params = {"max_errors": opts.pop("max_errors")} if "max_errors" in opts else {}
        if "quality_cutoff" in opts:
            opts["quality_cutoff"] = tuple(opts["quality_cutoff"])
        for chunk_bytes in (4 << 20, 512):          # 512 forces many chunks (reference --buffer-size=512 tests)
            ads = [adapter_from_spec(spec, kinds[opt], **params) for opt, spec in case["adapters"]]
            out, info = io.BytesIO(), io.BytesIO()
            stats = trim_fastq(os.path.join(FQ, case["input"]), out, ads, chunk_bytes=chunk_bytes,
                               info_file=info if case["info"] else None, **opts)
            if case["expected"]:
                expected = open(os.path.join(FQ, case["expected"]), "rb").read()
                assert out.getvalue() == expected, (case["name"], chunk_bytes)
            if case["info"]:
                expected = open(os.path.join(FQ, case["info"]), "rb").read()
                assert _strip_trailing_space(info.getvalue()) == _strip_trailing_space(expected), (case["name"], chunk_bytes)
            if chunk_bytes == 512:                      # reader + 3 worker threads + ordered writer
                out3, info3 = io.BytesIO(), io.BytesIO()
                # ... with the workers dealt round-robin over every visible GPU
                stats3 = trim_fastq(os.path.join(FQ, case["input"]), out3, ads, chunk_bytes=chunk_bytes,
                                    info_file=info3 if case["info"] else None, threads=3, devices="all", **opts)
                assert len(stats3["trimmer"].devices_used) >= 1 or stats3["reads"] == 0
                assert out3.getvalue() == out.getvalue() and info3.getvalue() == info.getvalue(), case["name"]
                assert (stats3["reads"], stats3["with_adapters"], stats3["bp_out"]) == \
                       (stats["reads"], stats["with_adapters"], stats["bp_out"]), case["name"]
                if stats["cutter"] is not None:
                    a, b = stats3["cutter"].histogram.counts, stats["cutter"].histogram.counts
                    assert a.sum() == b.sum()
        if case["name"] == "illumina_iupac":
            assert stats["reads"] == 100 and stats["with_adapters"] == 56
            # errors[removed_length][errors] sums to the number of matches (adapters.py:185-199)
            assert stats["cutter"].histogram.total() == 56
        if case["name"] == "max_expected_errors":
            assert stats["trimmer"].too_many_expected_errors == 2          # reference test_commandline.py:839


@pytest.mark.gpu
def test_pipeline_rounds_against_reference_adapter_cutter_semantics(hip, orc):
    """times > 1 and every action against a straightforward per-read restatement of
    AdapterCutter.match_and_trim (reference modifiers.py:209-251) driven by the oracle's
    match_to; random reads with several adapter copies."""
    import random
    from cutadapt_amd import adapters as A
    from cutadapt_amd.pipeline import BatchAdapterCutter, FastqChunk, read_fastq_chunks
    rng = random.Random(11)
    ad_seqs = ["ACGTTGCA", "GGATCCAA"]
    reads = []
    for i in range(400):
        s = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 60)))
        for _ in range(rng.randint(0, 3)):
            p = rng.randint(0, len(s))
            s = s[:p] + rng.choice(ad_seqs) + s[p:]
        reads.append(s)
    fq = "".join(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n" for i, s in enumerate(reads)).encode()
    chunk = list(read_fastq_chunks(io.BytesIO(fq)))[0]
    seqs, offsets = chunk.pack_sequences()

    from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
    # match_to() = k-mer prefilter, then locate (reference adapters.py:707-724, :815-832); the
    # prefilter is part of the semantics (it is not lossless for every adapter kind)
    finders = {
        "back": orc.KmerFinder(create_positions_and_kmers(ad_seqs[0], 3, 0.1, True, False), False, False),
        "front": orc.KmerFinder(create_positions_and_kmers(ad_seqs[1], 3, 0.1, False, True), False, False),
    }

    def expected_intervals(times, action):
        out = []
        for s in reads:
            wb, we, ms = 0, len(s), []
            for _ in range(times):
                best = None
                for seq, kind in ((ad_seqs[0], "back"), (ad_seqs[1], "front")):
                    flags = 14 if kind == "back" else 11
                    if not finders[kind].kmers_present(s[wb:we]):
                        continue
                    t = orc.Aligner(seq, 0.1, flags=flags, wildcard_ref=False, min_overlap=3).locate(s[wb:we])
                    if t is None:
                        continue
                    if best is None or t[4] > best[0][4] or (t[4] == best[0][4] and t[5] < best[0][5]):
                        best = (t, kind)
                if best is None:
                    break
                t, kind = best
                ms.append((t, kind, wb, we))
                if kind == "back":
                    we = wb + t[2]
                else:
                    wb = wb + t[3]
            if not ms:
                out.append((0, len(s), False))
            elif action == "retain":
                t, kind, b0, e0 = ms[-1]
                out.append(((0, t[3], True) if kind == "back" else (t[2], len(s), True)))
            elif action == "crop":
                t = ms[-1][0]
                out.append((t[2], t[3], True))
            elif action is None:
                out.append((0, len(s), True))
            else:
                out.append((wb, we, True))
        return out

    for times, action in ((1, "trim"), (3, "trim"), (2, "mask"), (3, "lowercase"), (1, "retain"), (1, "crop"), (2, None)):
        ads = [A.BackAdapter(ad_seqs[0]), A.FrontAdapter(ad_seqs[1])]
        res = BatchAdapterCutter(ads, times=times, action=action).process_arrays(seqs, offsets)
        exp = expected_intervals(times, action)
        got = list(zip(res["beg"].tolist(), res["end"].tolist(), res["matched"].tolist()))
        assert got == exp, (times, action, [(i, g, e) for i, (g, e) in enumerate(zip(got, exp)) if g != e][:3])


def test_paired_chunks_stay_in_step():
    """host only: two files cut into chunks with equal record counts, whatever the chunk size"""
    from cutadapt_amd.pipeline import read_paired_chunks
    r1 = "".join(f"@p{i}/1\n{'ACGT' * (i % 7)}\n+\n{'IIII' * (i % 7)}\n" for i in range(200)).encode()
    r2 = "".join(f"@p{i}/2\n{'TTGCA' * (i % 11)}\n+\n{'#####' * (i % 11)}\n" for i in range(200)).encode()
    for chunk_bytes in (50, 777, 1 << 20):
        total = 0
        for c1, c2 in read_paired_chunks(io.BytesIO(r1), io.BytesIO(r2), chunk_bytes):
            assert len(c1) == len(c2) and len(c1) > 0
            for j in range(len(c1)):
                a = bytes(c1.buf[c1.rec[j, 0]:c1.rec[j, 1]]); b = bytes(c2.buf[c2.rec[j, 0]:c2.rec[j, 1]])
                assert a[:-2] == b[:-2] and a.endswith(b"/1") and b.endswith(b"/2")
            total += len(c1)
        assert total == 200
    with pytest.raises(ValueError, match="improperly paired"):
        list(read_paired_chunks(io.BytesIO(r1), io.BytesIO(r2[: len(r2) // 2].rsplit(b"@", 1)[0]), 1 << 20))


@pytest.mark.gpu
def test_reference_paired_goldens(hip):
    """20 command lines of reference tests/test_paired.py: per-mate adapters, -q/-Q, -u/-U, -l/-L,
    --nextseq-trim, -m/-M, --pair-filter, --discard-(un)trimmed; both outputs byte for byte"""
    from cutadapt_amd.pipeline import adapter_from_spec, trim_fastq_paired
    PD = os.path.join(os.path.dirname(FQ), "paired")
    manifest = json.load(open(os.path.join(PD, "manifest.json")))
    assert len(manifest) >= 20
    kinds = {"-a": "back", "-g": "front", "-b": "anywhere"}

    def mate(opts):
        opts = dict(opts)
        params = opts.pop("params", {})
        ads = [adapter_from_spec(spec, kinds[o], **params) for o, spec in opts.pop("adapters", [])]
        if "quality_cutoff" in opts:
            opts["quality_cutoff"] = tuple(opts["quality_cutoff"])
        return dict(adapters=ads, **opts)

    for case in manifest:
        for chunk_bytes in (4 << 20, 300):
            o1, o2 = io.BytesIO(), io.BytesIO()
            stats = trim_fastq_paired(os.path.join(PD, case["in1"]), os.path.join(PD, case["in2"]), o1, o2,
                                      mate(case["r1"]), mate(case["r2"]), chunk_bytes=chunk_bytes, **case["top"])
            assert o1.getvalue() == open(os.path.join(PD, case["exp1"]), "rb").read(), (case["name"], chunk_bytes, 1)
            assert o2.getvalue() == open(os.path.join(PD, case["exp2"]), "rb").read(), (case["name"], chunk_bytes, 2)
        exp = open(os.path.join(PD, case["exp1"]), "rb").read()
        n_out = exp.count(b">") if exp.startswith(b">") else exp.count(b"\n") // 4
        assert stats["pairs_written"] == n_out, case["name"]
        if case["name"] == "paired_end":
            assert stats["pairs"] - stats["pairs_written"] == stats["filtered"].get("too_short", 0) == 1


def test_write_trimmed_runs_equal_piecewise_formatting():
    """cah_fastq_write_trimmed copies records that are kept whole and already canonical in runs; the bytes must be
    what formatting every record piece by piece gives -- LF / CRLF input, bare and repeated '+' lines, a file
    without final line feed, dropped and trimmed records in between.  (Host code: no GPU needed.)"""
    import ctypes as C
    import random
    import numpy as np
    from cutadapt_amd import _lib
    L = _lib.lib()
    rng = random.Random(5)
    for it in range(60):
        crlf = it % 5 == 4
        recs, parsed = [], []
        for i in range(rng.randint(1, 300)):
            n = rng.randint(0, 120)
            s = "".join(rng.choice("ACGTN") for _ in range(n))
            q = "".join(chr(rng.randint(33, 73)) for _ in range(n))
            name = f"r{i}" + (" c" * rng.randint(0, 3))
            plus = "+" + (name if rng.random() < 0.15 else "")
            recs.append(f"@{name}\n{s}\n{plus}\n{q}\n")
            parsed.append((name, s, q))
        text = "".join(recs)
        if crlf:
            text = text.replace("\n", "\r\n")
        if it % 7 == 3:
            text = text.rstrip("\r\n")
        buf = np.frombuffer(text.encode(), dtype=np.uint8)
        n = len(parsed)
        rec = np.zeros((n, 6), dtype=np.int64)
        nrec, consumed = C.c_int64(0), C.c_int64(0)
        _lib.check(L.cah_fastq_scan(buf.ctypes.data, len(buf), 1, n, rec.ctypes.data, C.byref(nrec), C.byref(consumed)))
        assert nrec.value == n
        beg = np.zeros(n, dtype=np.int32)
        end = np.zeros(n, dtype=np.int32)
        keep = np.ones(n, dtype=np.uint8)
        want = []
        for i, (name, s, q) in enumerate(parsed):
            a, b = 0, len(s)
            u = rng.random()
            if u < 0.25 and len(s):
                b = rng.randint(0, len(s))
            elif u < 0.3 and len(s):
                a = rng.randint(0, len(s))
            elif u < 0.4:
                keep[i] = 0
            beg[i], end[i] = a, b
            if keep[i]:
                want.append(f"@{name}\n{s[a:b]}\n+\n{q[a:b]}\n")
        out = np.zeros(len(buf) + 4 * n + 64, dtype=np.uint8)
        out_len = C.c_int64(0)
        _lib.check(L.cah_fastq_write_trimmed(buf.ctypes.data, rec.ctypes.data, n, beg.ctypes.data, end.ctypes.data,
                                             keep.ctypes.data, out.ctypes.data, len(out), C.byref(out_len)))
        assert bytes(out[:out_len.value]) == "".join(want).encode(), (it, crlf)


def test_paired_pieces_by_line_count():
    """host only: gpu_pipeline._paired_pieces cuts two FASTQ streams into pieces with equal record counts by counting
    line feeds (cah_fastq_span); the pieces put together are the inputs, whatever the block size"""
    import random
    from cutadapt_amd import _lib
    from cutadapt_amd.gpu_pipeline import _paired_pieces
    rng = random.Random(8)

    def fq(n, tag, final_newline=True):
        recs = []
        for i in range(n):
            L = rng.randint(0, 120)
            s = "".join(rng.choice("ACGTN") for _ in range(L))
            q = "".join(chr(rng.randint(33, 73)) for _ in range(L))
            q = ("@" + q[1:]) if (L and rng.random() < 0.2) else q        # quality lines may start with '@'
            recs.append(f"@{tag}{i} {'x' * rng.randint(0, 40)}\n{s}\n+\n{q}\n")
        text = "".join(recs)
        return (text if final_newline else text[:-1]).encode()

    for n, final_nl in ((0, True), (1, True), (257, True), (1000, False)):
        a, b = fq(n, "a"), fq(n, "b", final_nl)
        for block in (150, 997, 4096, 1 << 20):
            got_a, got_b, total = [], [], 0
            for d1, d2 in _paired_pieces(io.BytesIO(a), io.BytesIO(b), block):
                n1 = bytes(d1).count(b"\n")
                n2 = bytes(d2).count(b"\n") + (0 if bytes(d2).endswith(b"\n") else 1)
                assert n1 % 4 == 0 and n1 == n2 and n1 > 0, (n, block)
                total += n1 // 4
                got_a.append(bytes(d1))
                got_b.append(bytes(d2))
            assert b"".join(got_a) == a and b"".join(got_b) == b and total == n, (n, block)
    a, b = fq(50, "a"), fq(49, "b")
    with pytest.raises(ValueError, match="improperly paired"):
        list(_paired_pieces(io.BytesIO(a), io.BytesIO(b), 500))
    with pytest.raises(ValueError):
        list(_paired_pieces(io.BytesIO(a), io.BytesIO(a[:-40]), 500))
    # the counting primitive itself
    L = _lib.lib()
    buf = np.frombuffer(b"@r\nAC\n+\nII\n@s\nG\n+\nI", dtype=np.uint8)
    for final, limit, want in ((0, 10, (1, 11)), (1, 10, (2, len(buf))), (1, 1, (1, 11)), (0, 0, (0, 0))):
        nrec, used = C.c_int64(0), C.c_int64(0)
        _lib.check(L.cah_fastq_span(buf.ctypes.data, len(buf), final, limit, C.byref(nrec), C.byref(used)))
        assert (nrec.value, used.value) == want, (final, limit)
    # ... and the threaded form of it (plain files: sub-ranges read with preadv and counted side by side)
    from concurrent.futures import ThreadPoolExecutor
    from cutadapt_amd import gpu_pipeline as gp
    pool = ThreadPoolExecutor(max_workers=4)
    old = gp._LineFeedIndex.MIN_PART
    try:
        for part in (1, 7, 64, 1000):
            gp._LineFeedIndex.MIN_PART = part
            for n, tail in ((0, b""), (1, b""), (40, b"@x\nAC"), (300, b"@x\nACGT\n+\nII"), (300, b"\n\n")):
                data = np.frombuffer(fq(n, "t") + tail, dtype=np.uint8)
                index = gp._LineFeedIndex(data, pool, 5)
                nrec, used = C.c_int64(0), C.c_int64(0)
                _lib.check(L.cah_fastq_span(data.ctypes.data if len(data) else None, len(data), 0, 1 << 62, C.byref(nrec), C.byref(used)))
                assert index.records == nrec.value, (part, n)
                assert index.end_of(index.records) == used.value, (part, n)
                for some in {0, 1, nrec.value // 2, nrec.value}:
                    if some <= nrec.value:
                        _lib.check(L.cah_fastq_span(data.ctypes.data if len(data) else None, len(data), 0, some, C.byref(nrec), C.byref(used)))
                        assert index.end_of(some) == (used.value if some else 0), (part, n, some)
        import tempfile
        gp._LineFeedIndex.MIN_PART = 512
        with tempfile.TemporaryDirectory() as tmp:
            for n, final_nl in ((257, True), (2000, False)):
                a, b = fq(n, "a"), fq(n, "b", final_nl)
                pa, pb = os.path.join(tmp, "a.fastq"), os.path.join(tmp, "b.fastq")
                open(pa, "wb").write(a)
                open(pb, "wb").write(b)
                for block in (997, 4096, 40000, 1 << 20):
                    got_a, got_b, total = [], [], 0
                    for d1, d2 in _paired_pieces(pa, pb, block, threads=3):
                        n1 = bytes(d1).count(b"\n")
                        n2 = bytes(d2).count(b"\n") + (0 if bytes(d2).endswith(b"\n") else 1)
                        assert n1 % 4 == 0 and n1 == n2 and n1 > 0, (n, block)
                        total += n1 // 4
                        got_a.append(bytes(d1))
                        got_b.append(bytes(d2))
                    assert b"".join(got_a) == a and b"".join(got_b) == b and total == n, (n, block)
    finally:
        gp._LineFeedIndex.MIN_PART = old
        pool.shutdown()
