"""CPU twin of cutadapt_amd/workloads.py:device_batch (TEST / MEASUREMENT INFRASTRUCTURE, like everything
under oracle/): the same reads, generated with oracle/synth_reads.c, for the parity samples of bench.py and
the CPU baseline workers."""
from cutadapt_amd import workloads as _wl
from cutadapt_amd.workloads import GEN, SPECS, front_rule


def host_reads(config: str, first_index: int, n_reads: int, mate: int = 0, gen=None):
    """(uint8[n*150], int64[n+1]) of the config's reads [first_index, first_index + n_reads)"""
    import numpy as np
    from oracle import oracle as orc
    spec = SPECS[config]
    g = dict(GEN if gen is None else gen)
    adapters = spec["adapters2"] if (spec["kind"] == "paired" and mate == 1) else spec["adapters"]
    seed = spec["seed"] * 10 + mate if spec["kind"] == "paired" else spec["seed"]
    seqs, offsets = orc.synth_reads(seed, first_index, n_reads, _wl.READ_LEN, adapters, **g)
    if spec["kind"] == "linked" and n_reads:
        idx = np.arange(first_index, first_index + n_reads, dtype=np.int64)
        has, prefix = front_rule(idx, np)
        view = seqs.reshape(n_reads, _wl.READ_LEN)
        view[:, :16] = np.where(has[:, None], prefix.astype(np.uint8), view[:, :16])
    return seqs, offsets


def host_ragged(seqs, offsets, first_index: int = 0):
    """the CPU twin of workloads.ragged_device_batch: (uint8[sum], int64[n+1])"""
    import numpy as np
    n = len(offsets) - 1
    L = _wl.READ_LEN
    lens = _wl.ragged_lengths(np.arange(first_index, first_index + n, dtype=np.int64))
    keep = np.arange(L)[None, :] < lens[:, None]
    out = seqs.reshape(n, L)[keep]
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    return np.ascontiguousarray(out), off
