"""The synthetic workloads of BASELINE.json / SURVEY.md section 8(d) (configs C2..C5), shared by bench.py,
the parity tests and the CPU baseline harness: adapter sets, seeds, algorithmic bytes per unit, and the
read generator (reads are a pure function of (seed, global read index), generated directly in HBM by
cah_synth_reads; the CPU twin is oracle/host_workloads.py on top of oracle/synth_reads.c).

C3's reads additionally carry the linked adapter's anchored 5' part: 80 % of the reads (chosen by an
integer hash of the read index, identical in torch and numpy) start with 8 hash-derived bases followed by
ACGTACGT, matching the adapter ``^NNNNNNNNACGTACGT...TRUSEQ``."""
import os
import random

TRUSEQ_R1 = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"      # reference doc/guide.rst:2053
TRUSEQ_R2 = "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"      # reference doc/guide.rst:2054
# 150 bp is what every BASELINE config is quoted on; bench.py --read-len sets CAH_BENCH_READ_LEN (inherited by the CPU
# baseline's worker processes) for the other read lengths a sequencer emits (250 / 300 bp)
READ_LEN = int(os.environ.get("CAH_BENCH_READ_LEN", "150"))
GEN = {"p_adapter": 0.25, "p_edit": 0.02, "p_n": 0.005}
C3_FRONT = "NNNNNNNNACGTACGT"


def random_adapters(n: int, length: int, seed: int):
    rng = random.Random(seed)
    return ["".join(rng.choice("ACGT") for _ in range(length)) for _ in range(n)]


_C5_EXTRA = random_adapters(2, 33, 505)

SPECS = {
    "C2": {"kind": "single", "adapters": [TRUSEQ_R1], "seed": 2, "bytes_per_unit": 178, "unit": "Mreads/s",
           "metric": "Mreads/s (150 bp, 1 adapter, e=0.1)",
           "what": "single 3' adapter (TruSeq 33 bp), e=0.1, min_overlap=3"},
    "C3": {"kind": "linked", "front": C3_FRONT, "adapters": [TRUSEQ_R1], "seed": 3, "bytes_per_unit": 178,
           "unit": "Mreads/s", "metric": "Mreads/s (150 bp, linked adapter, e=0.1, IUPAC)",
           "what": "linked adapter ^NNNNNNNNACGTACGT...TruSeq (anchored 5' with IUPAC wildcards + 3'), e=0.1"},
    "C4": {"kind": "multi", "adapters": random_adapters(96, 33, 404), "seed": 4, "bytes_per_unit": 182,
           "unit": "Mreads/s", "metric": "Mreads/s (150 bp, 96 adapters, e=0.1)",
           "what": "96 distinct random 33-mers as 3' adapters (-a file:), k-mer heuristic on, e=0.1, min_overlap=3"},
    "C5": {"kind": "paired", "adapters": [TRUSEQ_R1, _C5_EXTRA[0]], "adapters2": [TRUSEQ_R2, _C5_EXTRA[1]], "seed": 5,
           "bytes_per_unit": 356, "unit": "Mpairs/s", "metric": "Mpairs/s (2 x 150 bp, 2 adapters per mate, e=0.1)",
           "what": "paired-end 2 x 150 bp, two 3' adapters per mate, e=0.1, min_overlap=3"},
}


def front_rule(idx, xp):
    """(has_front[n] bool, prefix[n, 16] uint8) of the C3 reads with global indices ``idx`` (int64 array of
    module ``xp`` = numpy or torch; only +, *, >>, &, % on int64, so both give the same bytes)."""
    has = ((idx * 2654435761) >> 16) % 5 != 0
    cols = []
    bases = (65, 67, 71, 84)                          # A C G T
    for t in range(8):
        code = ((idx * 40503 + (t + 1) * 7919) >> 5) & 3
        col = (code == 0) * bases[0] + (code == 1) * bases[1] + (code == 2) * bases[2] + (code == 3) * bases[3]
        cols.append(col)
    for ch in b"ACGTACGT":
        cols.append(idx * 0 + ch)
    return has, xp.stack(cols, 1)


def device_batch(config: str, n_reads: int, first_index: int = 0, mate: int = 0, device=None, gen=None):
    """The config's reads [first_index, first_index + n_reads) as a ReadBatch in HBM."""
    import torch
    from .batch import ReadBatch
    spec = SPECS[config]
    g = dict(GEN if gen is None else gen)
    adapters = spec["adapters2"] if (spec["kind"] == "paired" and mate == 1) else spec["adapters"]
    seed = spec["seed"] * 10 + mate if spec["kind"] == "paired" else spec["seed"]
    batch = ReadBatch.synthetic(n_reads, READ_LEN, adapters, seed=seed, first_index=first_index, device=device, **g)
    if spec["kind"] == "linked" and n_reads:
        view = batch.seqs.view(n_reads, READ_LEN)
        step = 8_000_000                                  # bounds the int64 temporaries
        for lo in range(0, n_reads, step):
            hi = min(n_reads, lo + step)
            idx = torch.arange(first_index + lo, first_index + hi, dtype=torch.int64, device=batch.device)
            has, prefix = front_rule(idx, torch)
            view[lo:hi, :16] = torch.where(has[:, None], prefix.to(torch.uint8), view[lo:hi, :16])
    return batch


# ---- ragged batches (bench.py --ragged): the same reads cut to 30 .. READ_LEN characters at their 3' end -- what a pipeline
# holds behind -q / -u / a first adapter round (reference cli.py:938-954: the adapter step comes after the quality
# trimmers).  The length of read i is a hash of its global index, the same in torch and numpy.
RAGGED_MIN = 30


def ragged_lengths(idx, read_len=None):
    """lengths of the reads with global indices ``idx`` (int64 array, numpy or torch)"""
    L = READ_LEN if read_len is None else read_len
    lo = min(RAGGED_MIN, L)
    return lo + ((idx * 2654435761 + 12345) >> 7) % (L - lo + 1)


def ragged_device_batch(batch, first_index: int = 0):
    """a uniform ReadBatch cut to ragged_lengths (new packed buffer + offsets, both in HBM)"""
    import torch
    from .batch import ReadBatch
    n, L = batch.n_reads, READ_LEN
    dev = batch.seqs.device
    lens = ragged_lengths(torch.arange(first_index, first_index + n, dtype=torch.int64, device=dev))
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=offsets[1:])
    out = torch.empty(int(offsets[-1].item()), dtype=torch.uint8, device=dev)
    view = batch.seqs.view(n, L)
    cols = torch.arange(L, device=dev)
    step = 4_000_000                                      # bounds the boolean mask
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        keep = cols[None, :] < lens[lo:hi, None]
        out[int(offsets[lo].item()):int(offsets[hi].item())] = view[lo:hi][keep]
    packed = ReadBatch(out, offsets, validated=True)
    packed.max_len = L                                     # (no read is longer than the uniform batch's)
    return packed


def ragged_view_batch(batch, first_index: int = 0):
    """a uniform ReadBatch cut to ragged_lengths WITHOUT a copy: views (starts + lengths) into the batch at its uniform
    stride -- what a pipeline holds once a modifier in front of the adapter search (quality trimming, -u -N, --length) has
    cut a sequencer's reads at their 3' end"""
    import torch
    n, dev = batch.n_reads, batch.seqs.device
    lens = ragged_lengths(torch.arange(first_index, first_index + n, dtype=torch.int64, device=dev))
    return batch.view(torch.zeros(n, dtype=torch.int64, device=dev), lens)
