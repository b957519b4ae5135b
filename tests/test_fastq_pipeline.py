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


@pytest.mark.gpu
def test_reference_commandline_goldens(hip):
    """cutadapt -a/-b ADAPTER in.fastq -o out.fastq for the simple single-end cases of reference
    tests/test_commandline.py (manifest.json lists test line, options, input, expected)."""
    from cutadapt_amd import adapters as A
    from cutadapt_amd.pipeline import trim_fastq
    manifest = json.load(open(os.path.join(FQ, "manifest.json")))
    assert len(manifest) >= 7
    for case in manifest:
        cls = {"back": A.BackAdapter, "front": A.FrontAdapter, "anywhere": A.AnywhereAdapter}[case["kind"]]
        ads = [cls(s, **case["extra"]) for s in case["adapters"]]
        out = io.BytesIO()
        for chunk_bytes in (4 << 20, 512):          # 512 forces many chunks (reference --buffer-size=512 tests)
            out = io.BytesIO()
            stats = trim_fastq(os.path.join(FQ, case["input"]), out, ads, chunk_bytes=chunk_bytes)
            expected = open(os.path.join(FQ, case["expected"]), "rb").read()
            assert out.getvalue() == expected, (case["name"], chunk_bytes)
        if case["name"] == "illumina_iupac":
            assert stats["reads"] == 100 and stats["with_adapters"] == 56
