"""The N>1 path on CPU: world_size-2 gloo run of shard planning, per-rank matching and the
statistics / ordered-result merge.  The per-rank "matcher" here is the CPU oracle (tests may
use it); on the GPU box the same sharding code feeds cah_match_batch (see bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRUSEQ = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"


def test_shard_planning():
    from cutadapt_amd import sharding as S
    for n, w in [(0, 1), (10, 3), (100, 8), (7, 8), (1_000_000_007, 8)]:
        ranges = [S.shard_range(n, w, r) for r in range(w)]
        assert ranges[0][0] == 0 and sum(c for _, c in ranges) == n
        for (f0, c0), (f1, _) in zip(ranges, ranges[1:]):
            assert f0 + c0 == f1
        assert max(c for _, c in ranges) - min(c for _, c in ranges) <= 1
    chunks = S.chunk_plan(10, 4)
    assert chunks == [(0, 4), (4, 4), (8, 2)]
    assert S.deal_chunks(5, 2) == [[0, 2, 4], [1, 3]]
    parts = {2: np.array([5, 6]), 0: np.array([1, 2]), 1: np.array([3, 4])}
    assert S.merge_ordered(parts).tolist() == [1, 2, 3, 4, 5, 6]
    with pytest.raises(ValueError):
        S.shard_range(10, 2, 2)


def test_histogram_merge():
    from cutadapt_amd.sharding import MatchHistogram
    h1, h2 = MatchHistogram(2), MatchHistogram(2)
    coords = np.array([[0, 10, 5, 15, 8, 1], [0, 33, 0, 33, 33, 0], [0, 0, 0, 0, 0, 0]])
    found = np.array([True, True, False])
    h1.add_batch(coords, found, np.array([0, 1, 0]))
    h2.add_batch(coords, found, np.array([0, 0, 0]))
    h1 += h2
    assert h1.total() == 4 and h1.counts[0, 10, 1] == 2 and h1.counts[1, 33, 0] == 1 and h1.counts[0, 33, 0] == 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from cutadapt_amd import sharding as S
    from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
    from oracle import oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = S.shard_range(n_total, world, rank)
    seqs, offsets = orc.synth_reads(2, first, count, 150, [TRUSEQ])
    a = orc.Aligner(TRUSEQ, 0.1, 14, False, False, 1, 3)
    f = orc.KmerFinder(create_positions_and_kmers(TRUSEQ, 3, 0.1, True, False))
    out6, status = orc.match_batch(a, f, seqs, offsets)
    hist = S.MatchHistogram(1)
    hist.add_batch(out6.astype(np.int64), status == 1, np.zeros(count, dtype=np.int32))
    hist.all_reduce_()
    # add_rows histograms grow with the data: the ranks hold DIFFERENT shapes before the sum
    rows = S.MatchHistogram(2)
    rows.add_rows(np.array([0, 1]), np.array([5 + 100 * rank, 3]), np.array([rank, 0]))
    rows.all_reduce_()
    if rank == 0:
        np.save(os.path.join(out_dir, "rows.npy"), rows.counts)
    full = S.gather_ordered(np.concatenate([out6, status[:, None].astype(np.int32)], axis=1), first)
    if rank == 0:
        np.save(os.path.join(out_dir, "full.npy"), full)
        np.save(os.path.join(out_dir, "hist.npy"), hist.counts)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(tmp_path, orc):
    import torch.multiprocessing as mp
    from cutadapt_amd import sharding as S
    from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
    n_total = 20001
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    full = np.load(tmp_path / "full.npy")
    hist = np.load(tmp_path / "hist.npy")
    seqs, offsets = orc.synth_reads(2, 0, n_total, 150, [TRUSEQ])
    a = orc.Aligner(TRUSEQ, 0.1, 14, False, False, 1, 3)
    f = orc.KmerFinder(create_positions_and_kmers(TRUSEQ, 3, 0.1, True, False))
    out6, status = orc.match_batch(a, f, seqs, offsets)
    assert full.shape == (n_total, 7)
    assert np.array_equal(full[:, :6], out6) and np.array_equal(full[:, 6], status)
    ref_hist = S.MatchHistogram(1)
    ref_hist.add_batch(out6.astype(np.int64), status == 1, np.zeros(n_total, dtype=np.int32))
    assert np.array_equal(hist, ref_hist.counts) and ref_hist.total() == int((status == 1).sum()) > 4000
    rows = np.load(tmp_path / "rows.npy")
    assert rows.shape[1] == 106 and rows.sum() == 4          # rank 1 grew its length axis, rank 0 did not
    assert rows[0, 5, 0] == 1 and rows[0, 105, 1] == 1 and rows[1, 3, 0] == 2
