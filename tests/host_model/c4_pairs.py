#!/usr/bin/env python3
"""C4 (96 adapters): which of an adapter's k-mer search sets let a (read, adapter) pair through the prefilter, and how
many columns of cost scan each kind of pair really needs.  CPU-only analysis (oracle generator + the k-mer heuristic's
sets, brute force in numpy); test infrastructure, nothing here is product code.
Usage: python tests/host_model/c4_pairs.py [n_reads]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc                                    # noqa: E402
from cutadapt_amd import workloads                                  # noqa: E402
from cutadapt_amd.kmer_heuristic import create_positions_and_kmers  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
L = 150
ads = workloads.CONFIGS["C4"]["adapters"] if hasattr(workloads, "CONFIGS") else None
if ads is None:
    cfg = [v for k, v in vars(workloads).items() if isinstance(v, dict) and "C4" in v][0]
    ads = cfg["C4"]["adapters"]
seqs, offsets = orc.synth_reads(4, 0, n, L, ads)
R = seqs.reshape(n, L)
m, k = len(ads[0]), int(0.1 * len(ads[0]))


def occurrences(kmer):
    """bool[n, L - len + 1]: kmer starts at that position"""
    ln = len(kmer)
    hit = np.ones((n, L - ln + 1), dtype=bool)
    for t, ch in enumerate(kmer):
        hit &= R[:, t:L - ln + 1 + t] == ord(ch)
    return hit


per_set = {}          # window start (negative: tail set of that reach; 0: whole read) -> pairs let through by that set
only_set = {}         # ... and by no other set
pairs = 0
exact_short = 0       # pairs let through ONLY by tail sets of reach <= 9 (rows with no error allowed: a suffix compare decides)
need_cols = []        # columns a set-aware window would scan for the pair (the widest set that hit; whole read: today's rule)
for ad in ads:
    sets = create_positions_and_kmers(ad, 3, 0.1, True, False)
    hits = {}
    for start, stop, kmers in sets:
        h = np.zeros(n, dtype=bool)
        for kmer in kmers:
            occ = occurrences(kmer)
            if start < 0:                                           # the k-mer lies inside the last -start characters
                lo = max(0, L + start)
                h |= occ[:, lo:].any(axis=1) if lo <= L - len(kmer) else False
            else:
                h |= occ.any(axis=1)
        hits[start] = h
    any_hit = np.zeros(n, dtype=bool)
    for h in hits.values():
        any_hit |= h
    pairs += int(any_hit.sum())
    for s_, h in hits.items():
        per_set[s_] = per_set.get(s_, 0) + int(h.sum())
        others = np.zeros(n, dtype=bool)
        for t_, g in hits.items():
            if t_ != s_:
                others |= g
        only_set[s_] = only_set.get(s_, 0) + int((h & ~others).sum())
    short = np.zeros(n, dtype=bool)
    longer = np.zeros(n, dtype=bool)
    for s_, h in hits.items():
        if -9 <= s_ < 0:
            short |= h
        else:
            longer |= h
    exact_short += int((short & ~longer).sum())
    # set-aware window: rows up to the reach of the widest tail set that hit (+ k + 1 columns in front); a whole-read
    # hit keeps today's m + k + 1 columns in front of the first hit to the read end (counted as 64 here)
    widest = np.zeros(n, dtype=np.int64)
    for s_, h in hits.items():
        widest = np.maximum(widest, np.where(h, 64 if s_ == 0 else -s_ + k + 1, 0))
    need_cols.append(widest[any_hit])
    if "--check-windows" in sys.argv:
        # the host model of the scan (back_scan.h under g++) with the set-aware window starts against the oracle's locate
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import ctypes as C
        import test_back_scan_model as tm
        if "model" not in globals():
            model = C.CDLL(tm.SO)
            vp, i64 = C.c_void_p, C.c_int64
            model.bm_locate_batch.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp, vp, C.c_int32, vp, C.c_int32]
            model.bm_skip_columns.argtypes = [C.c_char_p, C.c_int, C.c_int, vp, vp, i64, vp]
            checked = differ = 0
        idx = np.flatnonzero(any_hit)
        sub = R[idx].reshape(-1).copy()
        off = np.arange(len(idx) + 1, dtype=np.int64) * L
        j0 = np.zeros(len(idx), dtype=np.int32)
        model.bm_skip_columns(ad.encode(), m, k, sub.ctypes.data, off.ctypes.data, len(idx), j0.ctypes.data)
        w = widest[idx]
        j0 = np.where(hits[0][idx], j0, np.maximum(0, L - w)).astype(np.int32)     # tail-only pairs: reach + k + 1 columns
        blob, _ = tm.matcher_blob(ad, 0.1, 3)
        out6 = np.zeros((len(idx), 6), dtype=np.int32)
        status = np.zeros(len(idx), dtype=np.uint8)
        for stop_every in (0, 16):
            rc = model.bm_locate_batch(blob, sub.ctypes.data, off.ctypes.data, len(idx), j0.ctypes.data, out6.ctypes.data,
                                       status.ctypes.data, None, None, stop_every, None, -1)
            assert rc == 0
            want6, want_st = orc.Aligner(ad, 0.1, 14, False, False, 1, 3).locate_batch(sub, off)
            bad = np.flatnonzero((status != want_st) | (out6 != want6).any(axis=1))
            checked += len(idx)
            differ += len(bad)
            if len(bad):
                r = int(bad[0])
                print("DIFF", ad, bytes(R[idx[r]]).decode(), "j0", j0[r], "model", status[r], out6[r].tolist(), "oracle", want_st[r], want6[r].tolist())
need_cols = np.concatenate(need_cols)
if "--check-windows" in sys.argv:
    print(f"set-aware windows, host model against the oracle: {differ} of {checked} (pair, form) results differ")
print(f"{n} reads x {len(ads)} adapters: {pairs / n:.2f} pairs per read pass the k-mer sets")
for s_ in sorted(per_set):
    name = "whole read" if s_ == 0 else f"last {-s_:2d}"
    print(f"  set {name}: {per_set[s_] / n:.3f} pairs per read, {only_set[s_] / n:.3f} through this set alone")
print(f"pairs through tail sets of reach <= 9 only (no error allowed there: the result is a suffix compare): "
      f"{exact_short / n:.2f} per read = {exact_short / pairs:.2f} of all pairs")
rest = need_cols[need_cols > 9 + k + 1]
print(f"set-aware windows for the rest: mean {rest.mean():.1f} columns over {len(rest) / n:.2f} pairs per read "
      f"(today: 48 columns for a tail hit, more for a whole-read hit) -> {rest.sum() / n:.0f} columns per read against "
      f"{48 * pairs / n:.0f}+ today")
