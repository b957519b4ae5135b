"""gpu_pipeline._paired_pieces on the CPU: two FASTQ files cut into pieces of equal record counts (the job of
dnaio.read_paired_chunks, reference runners.py:104-113).  The reader prefetches the next block of each file while the
current one is counted; the pieces put back together must be the files, piece by piece with equal record counts."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.timeout(120)

from cutadapt_amd import gpu_pipeline as gp
from cutadapt_amd.pipeline import POOL


def write_fastq(path, n, rng, name_len, read_len):
    with open(path, "wb") as f:
        for i in range(n):
            L = int(rng.integers(read_len[0], read_len[1] + 1))
            name = ("r%d" % i).ljust(int(rng.integers(name_len[0], name_len[1] + 1)), "x")
            seq = "".join(rng.choice(list("ACGT"), size=L))
            f.write(("@%s\n%s\n+\n%s\n" % (name, seq, "I" * L)).encode())


def records(buf: bytes) -> int:
    assert buf.count(b"\n") % 4 == 0
    return buf.count(b"\n") // 4


@pytest.mark.parametrize("n,chunk,shape1,shape2", [
    (20000, 1 << 16, ((8, 12), (100, 100)), ((8, 12), (100, 100))),      # like sequencer output
    (30000, 1 << 15, ((8, 30), (30, 150)), ((20, 60), (100, 250))),      # one file's records are much larger: one side piles up
    (7, 1 << 20, ((8, 12), (50, 60)), ((8, 12), (50, 60))),              # one block
    (300, 300, ((8, 12), (100, 200)), ((8, 30), (150, 250))),            # blocks smaller than a record
    (2000, 700, ((8, 12), (20, 30)), ((8, 30), (200, 250))),             # ... than the larger file's records only
    (0, 1 << 16, ((8, 12), (50, 60)), ((8, 12), (50, 60))),              # empty files
])
def test_pieces_are_the_files_with_equal_record_counts(tmp_path, n, chunk, shape1, shape2):
    rng = np.random.default_rng(n + chunk)
    p1, p2 = str(tmp_path / "a.fastq"), str(tmp_path / "b.fastq")
    write_fastq(p1, n, rng, *shape1)
    write_fastq(p2, n, rng, *shape2)
    old_min, old_head = gp._LineFeedIndex.MIN_PART, None
    gp._LineFeedIndex.MIN_PART = 1 << 12                      # (so that the threaded count is used at these sizes)
    try:
        got1, got2, pieces = [], [], 0
        for d1, d2 in gp._paired_pieces(p1, p2, chunk, threads=3):
            b1, b2 = bytes(d1), bytes(d2)
            assert records(b1) == records(b2) > 0
            got1.append(b1); got2.append(b2)
            pieces += 1
            POOL.put(d1); POOL.put(d2)
    finally:
        gp._LineFeedIndex.MIN_PART = old_min
    assert b"".join(got1) == open(p1, "rb").read()
    assert b"".join(got2) == open(p2, "rb").read()
    if n > 1000:
        assert pieces > 3


def test_unequal_files_fail_like_the_reference(tmp_path):
    rng = np.random.default_rng(5)
    p1, p2 = str(tmp_path / "a.fastq"), str(tmp_path / "b.fastq")
    write_fastq(p1, 3000, rng, (8, 12), (100, 100))
    write_fastq(p2, 2000, rng, (8, 12), (100, 100))
    with pytest.raises(ValueError, match="improperly paired"):
        for d1, d2 in gp._paired_pieces(p1, p2, 1 << 16, threads=2):
            POOL.put(d1); POOL.put(d2)
