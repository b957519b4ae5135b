"""Sharding of read batches over the GPUs of one node, and merging of results.

Every read is an independent unit, so the path shards embarrassingly: contiguous read ranges
(or chunks dealt round-robin, the unit the reference uses -- 4 MiB FASTQ chunks handed to
worker processes, reference src/cutadapt/runners.py:116-134, :306) go to one process per GPU;
there is NO collective on the data path.  What has to be combined afterwards is
  * the per-read results, back in input order (reference OrderedChunkWriter,
    runners.py:224-245), and
  * the per-adapter match statistics, which are plain sums (reference
    EndStatistics.__iadd__, adapters.py:96-111; Statistics.__iadd__, report.py:81-126).
torch.distributed is used for those control-plane sums only (backend "nccl" = RCCL on the GPU
box, "gloo" in CPU tests).
"""
from typing import Dict, List, Tuple

import numpy as np


def shard_range(n_total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous range [first, first+count) of rank `rank`; sizes differ by at most one."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    base, extra = divmod(n_total, world_size)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def chunk_plan(n_reads: int, chunk_reads: int) -> List[Tuple[int, int]]:
    """[(first_read, count), ...] covering n_reads in order."""
    if chunk_reads <= 0:
        raise ValueError("chunk_reads must be positive")
    return [(s, min(chunk_reads, n_reads - s)) for s in range(0, n_reads, chunk_reads)]


def deal_chunks(n_chunks: int, world_size: int) -> List[List[int]]:
    """Round-robin dealing of chunk indices to ranks (the reader process hands each chunk to
    the next free worker; with equal-cost chunks that is round-robin)."""
    return [list(range(r, n_chunks, world_size)) for r in range(world_size)]


def merge_ordered(parts: Dict[int, np.ndarray]) -> np.ndarray:
    """Concatenate per-chunk result arrays in chunk order, whatever order they arrived in."""
    if not parts:
        return np.zeros((0,), dtype=np.int32)
    return np.concatenate([parts[i] for i in sorted(parts)], axis=0)


class MatchHistogram:
    """errors[adapter][aligned_length][n_errors] counts -- the array form of the reference's
    ``EndStatistics.errors[length][errors]`` (adapters.py:82-83, filled by add_match :185-199).
    Built with one vectorised bincount per batch; merged across ranks/chunks by addition."""

    def __init__(self, n_adapters: int, max_len: int = 64, max_errors: int = 64):
        self.counts = np.zeros((n_adapters, max_len + 1, max_errors + 1), dtype=np.int64)

    def add_batch(self, coords: np.ndarray, found: np.ndarray, adapter_index: np.ndarray) -> None:
        """coords: int[n,6] (astart, astop, rstart, rstop, score, errors)"""
        if not found.any():
            return
        c = coords[found]
        length = (c[:, 1] - c[:, 0]).astype(np.int64)
        errors = c[:, 5].astype(np.int64)
        ad = adapter_index[found].astype(np.int64)
        shape = self.counts.shape
        flat = (ad * shape[1] + length) * shape[2] + errors
        self.counts += np.bincount(flat, minlength=self.counts.size).reshape(shape)

    def add_rows(self, slot: np.ndarray, removed_length: np.ndarray, errors: np.ndarray) -> None:
        """The reference's own key: errors[removed_sequence_length][errors] per adapter end
        (adapters.py:185-199, linked :233-247).  The length axis grows on demand."""
        if len(slot) == 0:
            return
        slot = np.asarray(slot, dtype=np.int64)
        length = np.asarray(removed_length, dtype=np.int64)
        errors = np.asarray(errors, dtype=np.int64)
        need_len, need_err = int(length.max()) + 1, int(errors.max()) + 1
        if need_len > self.counts.shape[1] or need_err > self.counts.shape[2]:
            grown = np.zeros((self.counts.shape[0], max(need_len, self.counts.shape[1]),
                              max(need_err, self.counts.shape[2])), dtype=np.int64)
            grown[:, :self.counts.shape[1], :self.counts.shape[2]] = self.counts
            self.counts = grown
        shape = self.counts.shape
        flat = (slot * shape[1] + length) * shape[2] + errors
        self.counts += np.bincount(flat, minlength=self.counts.size).reshape(shape)

    def __iadd__(self, other: "MatchHistogram") -> "MatchHistogram":
        if self.counts.shape[0] != other.counts.shape[0]:
            raise ValueError("incompatible histograms")
        if self.counts.shape != other.counts.shape:
            shape = tuple(max(a, b) for a, b in zip(self.counts.shape, other.counts.shape))
            grown = np.zeros(shape, dtype=np.int64)
            grown[:, :self.counts.shape[1], :self.counts.shape[2]] = self.counts
            self.counts = grown
        self.counts[:, :other.counts.shape[1], :other.counts.shape[2]] += other.counts
        return self

    def total(self) -> int:
        return int(self.counts.sum())

    def all_reduce_(self, group=None) -> "MatchHistogram":
        """Sum over all ranks (control plane; no read data is exchanged)."""
        import torch
        import torch.distributed as dist
        on_gpu = dist.get_backend(group) == "nccl"
        # add_rows grows the length / error axes with the data, so ranks may hold different shapes: agree on
        # the largest one first (a sum over unequal tensors would hang or corrupt the result)
        # one MAX reduction carries both the largest shape and (negated) the smallest number of adapter slots, so
        # that EVERY rank sees a disagreement and raises -- a rank that went on into the sum would wait for ever
        shape = torch.tensor(list(self.counts.shape) + [-self.counts.shape[0]], dtype=torch.int64)
        if on_gpu:
            shape = shape.cuda()
        dist.all_reduce(shape, op=dist.ReduceOp.MAX, group=group)
        vals = [int(v) for v in shape.cpu().tolist()]
        shape, min_slots = tuple(vals[:3]), -vals[3]
        if shape[0] != min_slots:
            raise ValueError("incompatible histograms: ranks disagree on the number of adapter slots")
        if shape != self.counts.shape:
            grown = np.zeros(shape, dtype=np.int64)
            grown[:, :self.counts.shape[1], :self.counts.shape[2]] = self.counts
            self.counts = grown
        t = torch.from_numpy(np.ascontiguousarray(self.counts))
        if on_gpu:
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        self.counts = t.cpu().numpy()
        return self


def gather_ordered(local: np.ndarray, first: int, group=None):
    """Collect per-rank result rows on rank 0 in global read order.  Returns the full array on
    rank 0, None elsewhere.  (Results normally stay on their GPU; this is for reports/tests.)"""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    objs = [None] * world if rank == 0 else None
    dist.gather_object((first, local), objs, dst=0, group=group)
    if rank != 0:
        return None
    return np.concatenate([part for _, part in sorted(objs, key=lambda x: x[0])], axis=0)
