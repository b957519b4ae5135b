"""ReverseComplementer as a batch operation (reference modifiers.py:264-308, --revcomp).

CPU: the host formatter (cah_chunk_revcomp, cah_info_write_rc) against a few lines of Python.
GPU: the device kernel (cah_revcomp_reads_batch) against the same lines of Python, and BatchReverseComplementer
against the reference's rule restated per read over ORACLE results: search the read and its reverse complement
with match_and_trim, take the reverse complement iff the scores of its matches add up to more.

dnaio (the reference's dependency that owns SequenceRecord.reverse_complement()) is not in the image; its
complement table is restated in csrc/revcomp.h and below, and pinned by the reference's own fixtures
(tests/golden/fastq: revcomp_normalized, info_file_revcomp -- test_fastq_pipeline.py runs them).
"""
import io
import os

import numpy as np
import pytest

SEED0 = int(os.environ.get("CAH_TEST_SEED_OFFSET", "0"))
COMP = {}
for a, b in ("AT", "TA", "CG", "GC", "UA", "MK", "KM", "RY", "YR", "WW", "SS", "NN", "VB", "BV", "HD", "DH"):
    COMP[a] = b
    COMP[a.lower()] = b.lower()


def revcomp(s: str) -> str:
    return "".join(COMP.get(c, c) for c in reversed(s))


def random_read(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(list(alphabet), size=n)) if n else ""


def check_chunk(chunk, reads, quals, fasta, rng, first):
    flags = rng.random(len(reads)) < 0.5
    for suffix in (" rc", None):
        out = chunk.reverse_complemented(flags, suffix)
        n = len(reads)
        z = np.zeros(n, dtype=np.int32)
        lens = np.array([len(s) for s in reads], dtype=np.int32)
        got = bytes(out.write_records(z, lens))
        want = []
        for i, (s, q) in enumerate(zip(reads, quals)):
            name = f"r{first + i} x" + ((suffix or "") if flags[i] else "")
            s2, q2 = (revcomp(s), q[::-1]) if flags[i] else (s, q)
            want.append(f">{name}\n{s2}\n" if fasta else f"@{name}\n{s2}\n+\n{q2}\n")
        assert got == "".join(want).encode(), (fasta, suffix)
        # the derived chunk packs to the merged sequences, and its qualities line up with them
        seqs, offsets = out.pack_sequences()
        assert bytes(seqs) == "".join(revcomp(s) if f else s for s, f in zip(reads, flags)).encode()
        if not fasta:
            assert bytes(out.pack_qualities()) == "".join(q[::-1] if f else q for q, f in zip(quals, flags)).encode()
    # info rows: one match per read with an even index, the last column says which orientation
    rows = np.array([[i, 0, 0, min(2, len(reads[i])), 0, len(reads[i]), 0] for i in range(0, len(reads), 2)],
                    dtype=np.int64)
    out = chunk.reverse_complemented(flags, None)
    lines = out.write_info(rows, ["ad"], flags).decode().split("\n")[:-1]
    assert len(lines) == len(reads)
    for i, line in enumerate(lines):
        cols = line.split("\t")
        if i % 2 == 0:
            assert cols[-1] == ("1" if flags[i] else "0") and len(cols) == 12, line
            s2 = revcomp(reads[i]) if flags[i] else reads[i]
            assert cols[4] + cols[5] + cols[6] == s2
        else:
            assert cols[1] == "-1"
    plain = out.write_info(rows, ["ad"]).decode().split("\n")[:-1]
    assert all(l.split("\t")[-1] == "" for l in plain[::2])
    chunk.release()


def test_chunk_revcomp_and_info_rc_column():
    from cutadapt_amd.pipeline import read_fastq_chunks
    rng = np.random.default_rng(5 + SEED0)
    alphabet = "ACGTacgtNnRYMKWSVBHDU.-*[{@"
    for fasta in (False, True):
        reads = [random_read(rng, int(n), alphabet) for n in rng.integers(0, 70, size=60)]
        quals = ["".join(chr(int(q)) for q in rng.integers(33, 74, size=len(s))) for s in reads]
        if fasta:
            text = "".join(f">r{i} x\n{s}\n" for i, s in enumerate(reads))
        else:
            text = "".join(f"@r{i} x\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(zip(reads, quals)))
        all_reads, all_quals, first = reads, quals, 0
        # (a FASTA reader hands the last record over in a chunk of its own: it cannot know that the record is
        # complete before the file ends)
        for chunk in read_fastq_chunks(io.BytesIO(text.encode())):
            reads, quals = all_reads[first:first + len(chunk)], all_quals[first:first + len(chunk)]
            check_chunk(chunk, reads, quals, fasta, rng, first)
            first += len(chunk)
        assert first == len(all_reads)


@pytest.mark.gpu
def test_revcomp_kernel_every_length(hip):
    import torch
    from cutadapt_amd.adapters import _reverse_batch
    from cutadapt_amd.batch import ReadBatch
    rng = np.random.default_rng(6 + SEED0)
    alphabet = "ACGTacgtNnRYMKWSVBHDUrymkwsvbhdu.-*[{@X"
    reads = [random_read(rng, n, alphabet) for n in list(range(0, 70)) + [150, 151, 300, 1000] + [150] * 300]
    batch = ReadBatch.from_strings(reads)
    n = len(reads)

    def unpack(rb):
        data, off = rb.seqs.cpu().numpy().tobytes().decode(), rb.offsets.cpu().numpy()
        return [data[off[i]:off[i + 1]] for i in range(n)]

    assert unpack(_reverse_batch(batch, complement=True)) == [revcomp(s) for s in reads]
    assert unpack(_reverse_batch(batch)) == [s[::-1] for s in reads]
    sel = rng.random(n) < 0.5
    got = unpack(_reverse_batch(batch, complement=True, select=torch.from_numpy(sel).cuda()))
    assert got == [revcomp(s) if f else s for s, f in zip(reads, sel)]
    # qualities: another byte tensor with the batch's layout, reversed where selected
    q = torch.from_numpy(rng.integers(33, 74, size=int(batch.seqs.numel()), dtype=np.uint8)).cuda()
    out = _reverse_batch(batch, select=torch.from_numpy(sel).cuda(), data=q).cpu().numpy()
    qh, off = q.cpu().numpy(), batch.offsets.cpu().numpy()
    for i in range(n):
        w = qh[off[i]:off[i + 1]]
        assert np.array_equal(out[off[i]:off[i + 1]], w[::-1] if sel[i] else w), i
    # uniform batches stay uniform (the fast entry point)
    uni = ReadBatch.from_strings(["ACGTN" * 30] * 7)
    assert _reverse_batch(uni, complement=True).uniform_len == 150


@pytest.mark.gpu
def test_reverse_complementer_against_the_rule_over_oracle_results(hip, orc):
    import random
    from cutadapt_amd import adapters as A
    from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
    from cutadapt_amd.pipeline import BatchAdapterCutter, BatchReverseComplementer, read_fastq_chunks
    rng = random.Random(21)
    ad_seqs = ["ACGTTGCAAG", "GGATCCAATC"]
    reads = []
    for i in range(600):
        s = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 70)))
        for _ in range(rng.randint(0, 3)):
            p = rng.randint(0, len(s))
            piece = rng.choice(ad_seqs)
            if rng.random() < 0.5:
                piece = revcomp(piece)
            if rng.random() < 0.3:
                q = rng.randrange(len(piece))
                piece = piece[:q] + rng.choice("ACGT") + piece[q + 1:]
            s = s[:p] + piece + s[p:]
        reads.append(s)
    fq = "".join(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n" for i, s in enumerate(reads)).encode()
    chunk = list(read_fastq_chunks(io.BytesIO(fq)))[0]
    seqs, offsets = chunk.pack_sequences()
    finders = {
        "back": orc.KmerFinder(create_positions_and_kmers(ad_seqs[0], 3, 0.1, True, False), False, False),
        "front": orc.KmerFinder(create_positions_and_kmers(ad_seqs[1], 3, 0.1, False, True), False, False),
    }

    def match_and_trim(s, times):
        """modifiers.py:209-251 for action 'trim' -> (beg, end, matches, score)"""
        wb, we, ms, score = 0, len(s), [], 0
        for _ in range(times):
            best = None
            for seq, kind in ((ad_seqs[0], "back"), (ad_seqs[1], "front")):
                if not finders[kind].kmers_present(s[wb:we]):
                    continue
                t = orc.Aligner(seq, 0.1, flags=14 if kind == "back" else 11, wildcard_ref=False,
                                min_overlap=3).locate(s[wb:we])
                if t is None:
                    continue
                if best is None or t[4] > best[0][4] or (t[4] == best[0][4] and t[5] < best[0][5]):
                    best = (t, kind)
            if best is None:
                break
            t, kind = best
            ms.append((t, kind))
            score += t[4]
            if kind == "back":
                we = wb + t[2]
            else:
                wb = wb + t[3]
        return wb, we, ms, score

    for times in (1, 3):
        want, n_rc, per_adapter_rc = [], 0, [0, 0]
        for s in reads:
            f = match_and_trim(s, times)
            r = match_and_trim(revcomp(s), times)
            use = r[3] > f[3]                                   # modifiers.py:289: strictly greater
            n_rc += use
            b, e, ms, _ = r if use else f
            want.append((b, e, bool(ms), bool(use)))
            if use:
                for t, kind in ms:
                    per_adapter_rc[0 if kind == "back" else 1] += 1
        ads = [A.BackAdapter(ad_seqs[0]), A.FrontAdapter(ad_seqs[1])]
        cutter = BatchAdapterCutter(ads, times=times)
        rc = BatchReverseComplementer(cutter)
        res = rc.process_arrays(seqs, offsets)
        got = list(zip(res["beg"].tolist(), res["end"].tolist(), res["matched"].tolist(), res["rc"].tolist()))
        assert got == want, (times, [(i, g, w) for i, (g, w) in enumerate(zip(got, want)) if g != w][:3])
        assert rc.reverse_complemented == n_rc and n_rc > 50
        assert cutter.reverse_complemented.tolist() == per_adapter_rc
        assert cutter.with_adapters == sum(w[2] for w in want)
        assert cutter.histogram.total() == len(res["rows"])


@pytest.mark.gpu
def test_revcomp_through_the_whole_trimmer(hip):
    """--revcomp together with the modifiers around it (-q in front, --poly-a and --max-ee behind): the chunked,
    threaded run equals a per-read Python composition of the same batch operations on single-read chunks."""
    from cutadapt_amd.pipeline import adapter_from_spec, trim_fastq
    rng = np.random.default_rng(23 + SEED0)
    ad = "ACGTTGCAAGTC"
    recs, expect = [], {}
    for i in range(300):
        # (the poly-A trimmer tolerates mismatches: "...AAAC" in front of the tail would go with it)
        body = "C" + random_read(rng, int(rng.integers(20, 80))) + "GCGCGC"
        u = rng.random()
        lo = 53
        if u < 0.35:
            s = body + ad + random_read(rng, int(rng.integers(0, 10)))
            expect[f"r{i}"] = body
        elif u < 0.7:
            s = revcomp(body + "AAAAAAAAAAAA" + ad + random_read(rng, int(rng.integers(0, 10))))
            expect[f"r{i} rc"] = body                        # turned around, adapter cut, poly-A tail gone
        else:
            s, lo = body, 35
        q = "".join(chr(int(x)) for x in rng.integers(lo, 74, size=len(s)))
        recs.append(f"@r{i}\n{s}\n+\n{q}\n")
    data = "".join(recs).encode()
    opts = dict(quality_cutoff=(0, 5), poly_a=True, max_expected_errors=3.0, revcomp=True)
    whole = io.BytesIO()
    stats = trim_fastq(io.BytesIO(data), whole, [adapter_from_spec(ad, "back")], **opts)
    assert stats["reverse_complemented"] > 50
    pieces, n_rc = [], 0
    for r in recs:
        out = io.BytesIO()
        st = trim_fastq(io.BytesIO(r.encode()), out, [adapter_from_spec(ad, "back")], **opts)
        n_rc += st["reverse_complemented"]
        pieces.append(out.getvalue())
    assert whole.getvalue() == b"".join(pieces)
    assert n_rc == stats["reverse_complemented"]
    lines = whole.getvalue().decode().split("\n")
    got = {lines[k][1:]: lines[k + 1] for k in range(0, len(lines) - 1, 4)}
    assert all(got.get(name) == seq for name, seq in expect.items()), \
        [(name, seq, got.get(name)) for name, seq in expect.items() if got.get(name) != seq][:3]
    assert stats["trimmer"].too_many_expected_errors > 0
    threaded = io.BytesIO()
    st3 = trim_fastq(io.BytesIO(data), threaded, [adapter_from_spec(ad, "back")], chunk_bytes=2048, threads=3, **opts)
    assert threaded.getvalue() == whole.getvalue()
    assert st3["reverse_complemented"] == stats["reverse_complemented"]
    assert st3["cutter"].reverse_complemented.tolist() == stats["cutter"].reverse_complemented.tolist()


@pytest.mark.gpu
def test_paired_reverse_complementer_against_the_rule_over_oracle_results(hip, orc):
    """PairedReverseComplementer (reference modifiers.py:311-405): both cutters on the pair and on the pair with its mates
    exchanged, the exchanged pair wins when its scores add up to more; host-parsed and device-parsed chunks, with a
    modifier in front (-q) and behind (-l) the adapter step."""
    import random
    from cutadapt_amd import adapters as A
    from cutadapt_amd.gpu_pipeline import trim_fastq_gpu_paired
    from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
    from cutadapt_amd.pipeline import trim_fastq_paired
    rng = random.Random(31)
    ad1, ad2 = "ACGTTGCAAGTCAG", "GGATCCAATCGTTA"
    finders = {ad: orc.KmerFinder(create_positions_and_kmers(ad, 3, 0.1, True, False), False, False) for ad in (ad1, ad2)}

    def cut(ad, s):
        """one round of a 3' adapter -> (kept length, score)"""
        if not finders[ad].kmers_present(s):
            return len(s), 0
        t = orc.Aligner(ad, 0.1, flags=14, wildcard_ref=False, min_overlap=3).locate(s)
        return (len(s), 0) if t is None else (t[2], t[4])

    pairs = []
    for i in range(1500):
        r = []
        for ad_own, ad_other in ((ad1, ad2), (ad2, ad1)):
            s = "".join(rng.choice("ACGT") for _ in range(rng.randint(5, 70)))
            u = rng.random()
            if u < 0.4:
                s += ad_own[:rng.randint(3, len(ad_own))]
            elif u < 0.8:
                s += ad_other[:rng.randint(3, len(ad_other))]          # as if the mates had been swapped
            s += "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 4)))
            q = "".join(chr(rng.randint(50, 73)) for _ in s)
            if rng.random() < 0.3:
                q = q[:-2] + "##"
            r.append((s, q))
        pairs.append(r)
    fq = ["".join(f"@p{i}/{k + 1}\n{p[k][0]}\n+\n{p[k][1]}\n" for i, p in enumerate(pairs)).encode() for k in (0, 1)]
    QCUT, LENGTH = (0, 10), 40
    from test_gpu_fastq_general import quality_trim_index
    want1, want2, n_rc = [], [], 0
    for i, p in enumerate(pairs):
        trimmed = []
        for s, q in p:
            a, b = quality_trim_index(q, *QCUT)
            trimmed.append((s[a:b], q[a:b]))
        (s1, q1), (s2, q2) = trimmed
        plain = (cut(ad1, s1), cut(ad2, s2))
        swapped = (cut(ad1, s2), cut(ad2, s1))
        use = swapped[0][1] + swapped[1][1] > plain[0][1] + plain[1][1]
        n_rc += use
        if use:
            o1 = (f"p{i}/2 rc", s2[:swapped[0][0]], q2[:swapped[0][0]])
            o2 = (f"p{i}/1 rc", s1[:swapped[1][0]], q1[:swapped[1][0]])
        else:
            o1 = (f"p{i}/1", s1[:plain[0][0]], q1[:plain[0][0]])
            o2 = (f"p{i}/2", s2[:plain[1][0]], q2[:plain[1][0]])
        want1.append(f"@{o1[0]}\n{o1[1][:LENGTH]}\n+\n{o1[2][:LENGTH]}\n")
        want2.append(f"@{o2[0]}\n{o2[1][:LENGTH]}\n+\n{o2[2][:LENGTH]}\n")
    assert 300 < n_rc < 1200
    for fn, kw in ((trim_fastq_paired, {}), (trim_fastq_gpu_paired, {"threads": 2})):
        for chunk_bytes in (1 << 20, 3000):
            o1, o2 = io.BytesIO(), io.BytesIO()
            mate = lambda ad: dict(adapters=[A.BackAdapter(ad)], quality_cutoff=QCUT, length=LENGTH)
            stats = fn(io.BytesIO(fq[0]), io.BytesIO(fq[1]), o1, o2, mate(ad1), mate(ad2), revcomp=True,
                       chunk_bytes=chunk_bytes, **kw)
            assert o1.getvalue() == "".join(want1).encode(), (fn.__name__, chunk_bytes)
            assert o2.getvalue() == "".join(want2).encode(), (fn.__name__, chunk_bytes)
            assert stats["reverse_complemented"] == n_rc


def test_chunk_select_and_info_rows_on_the_original_read():
    """host only: cah_chunk_select (the two output chunks of PairedReverseComplementer) and the window walk InfoFileWriter
    does on the read as it came in (pipeline._info_rows_on_original, reference steps.py:232-247)"""
    from cutadapt_amd.pipeline import _info_rows_on_original, read_fastq_chunks
    rng = np.random.default_rng(9 + SEED0)
    n = 40
    texts = []
    for tag in ("a", "b"):
        recs = []
        for i in range(n):
            L = int(rng.integers(0, 50))
            s = random_read(rng, L, "ACGTN")
            q = "".join(chr(int(x)) for x in rng.integers(33, 74, size=L))
            recs.append((f"{tag}{i} extra", s, q))
        texts.append(recs)
    chunks = [list(read_fastq_chunks(io.BytesIO("".join(f"@{nm}\n{s}\n+\n{q}\n" for nm, s, q in recs).encode())))[0] for recs in texts]
    swap = rng.random(n) < 0.5
    for k in (0, 1):
        out = chunks[k].selected(chunks[1 - k], swap, " rc")
        lens = np.array([len((texts[1 - k] if swap[i] else texts[k])[i][1]) for i in range(n)], dtype=np.int32)
        got = bytes(out.write_records(np.zeros(n, np.int32), lens))
        want = "".join("@{}{}\n{}\n+\n{}\n".format(r[0], " rc" if swap[i] else "", r[1], r[2])
                       for i in range(n) for r in [(texts[1 - k] if swap[i] else texts[k])[i]])
        assert got == want.encode(), k
    # the window walk: a 5' match (removes what is in front) then a 3' match, coordinates found on the trimmed read
    lens = np.array([100, 80], dtype=np.int64)
    rows = np.array([[0, 1, 4, 14, 7, 90, 0],          # read 0, first match: rstart 4, rstop 14 on the read as trimmed (window 7..90)
                     [0, 0, 30, 45, 21, 90, 1],        # read 0, second match after the first removed [.., 14)
                     [1, 2, 50, 60, 0, 70, 1]], dtype=np.int64)
    before = np.array([True, False, False])
    got = _info_rows_on_original(rows, before, lens)
    # read 0: the first row is cut out of the WHOLE read (0..100); the 5' match leaves [14, 100); the second row is cut there
    assert got[:, 4:6].tolist() == [[0, 100], [14, 100], [0, 80]]
    assert got[:, 1:4].tolist() == rows[:, 1:4].tolist() and got[:, 6].tolist() == [0, 1, 1]
