#!/usr/bin/env python3
"""How full are the cost scan's waves?  (CPU-only analysis with the host model of back_scan.h; test infrastructure.)

The scan (k_back_scan, csrc/kernels.hip) runs 64 survivors of the prefilter per wave in lock step, one 16-column chunk
per iteration, until every lane is done (exact match found, early stop, read end) -- a lane that finishes early idles.
This script replays that on the C2 workload: the oracle's generator and KmerFinder give the survivors, the host model
(tests/host_model/back_model.cpp = back_scan.h under g++) gives every survivor's window start (bs_align_window of the
prefilter's column-skipping position) and the column its scan ends at; survivors are queued the kernel's way (tiles of
8192 reads, counting sort by key inside a tile) and cut into waves of 64.
Prints: chunks a wave executes vs. chunks its lanes need (lane utilisation), with and without the straggler rule
(<= 12 lanes left with >= 32 columns to go leave the wave), and what a perfect re-packing of unfinished lanes after
every chunk would execute.
Usage: python tests/host_model/scan_occupancy.py [n_reads]"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc                                    # noqa: E402
from cutadapt_amd.kmer_heuristic import create_positions_and_kmers  # noqa: E402
import test_back_scan_model as tm                                   # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
TILE, WAVE, RETRY_AT = 8192, 64, 12
ad, rate, O, L = tm.TRUSEQ, 0.1, 3, 150
m, k = len(ad), int(rate * len(ad))
model = tm.model.__wrapped__() if hasattr(tm.model, "__wrapped__") else None
if model is None:
    import subprocess
    if not os.path.exists(tm.SO):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", tm.SRC, "-o", tm.SO], check=True)
    model = C.CDLL(tm.SO)
    vp, i64 = C.c_void_p, C.c_int64
    model.bm_locate_batch.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp, vp, C.c_int32, vp, C.c_int32]
    model.bm_skip_columns.argtypes = [C.c_char_p, C.c_int, C.c_int, vp, vp, i64, vp]

seqs, offsets = orc.synth_reads(2, 0, n_reads, L, [ad])
finder = orc.KmerFinder(create_positions_and_kmers(ad, O, rate, True, False), False, False)
present = np.asarray(finder.kmers_present_batch(seqs, offsets)).astype(bool)
surv = np.flatnonzero(present)
print(f"{n_reads} reads, {len(surv)} survivors ({len(surv) / n_reads:.3f})")
# the survivors as a batch of their own
s_seqs = seqs.reshape(n_reads, L)[surv].reshape(-1).copy()
s_off = np.arange(len(surv) + 1, dtype=np.int64) * L
j0 = np.zeros(len(surv), dtype=np.int32)
model.bm_skip_columns(ad.encode(), m, k, s_seqs.ctypes.data, s_off.ctypes.data, len(surv), j0.ctypes.data)
key = (j0 + m + k + 1) >> 2                                         # (j0 = max(0, 4 key - m - k - 1); key 0..9 collapse: same window)
if "--refined" in sys.argv:
    # DESIGN 11's refined window start: min over the adapter's k+1 chunks c of (first end of c - end offset of c in
    # the adapter) - k - 1, capped at n - m - k - 1 (the tail rows' paths).  f_c by brute force here.
    R = s_seqs.reshape(len(surv), L)
    chunks, base, extra = k + 1, m // (k + 1), m % (k + 1)
    best = np.full(len(surv), L - m, dtype=np.int64)                # the cap, before the - k - 1
    pos = 0
    for c in range(chunks):
        ln = base + (1 if c < extra else 0)
        hit = np.ones((len(surv), L - ln + 1), dtype=bool)
        for t in range(ln):
            hit &= R[:, t:L - ln + 1 + t] == ord(ad[pos + t])
        first = np.where(hit.any(axis=1), hit.argmax(axis=1) + ln, 10 ** 6)    # first END (1-based column) of chunk c
        best = np.minimum(best, first - (pos + ln))
        pos += ln
    j0_new = np.maximum(0, best - k - 1).astype(np.int32)
    assert (j0_new >= j0).all()
    print(f"refined window start: mean {j0_new.mean():.1f} against {j0.mean():.1f}; later by >= 16 columns for "
          f"{(j0_new - j0 >= 16).mean():.3f} of the survivors")
    j0_plain = j0
    j0 = j0_new
blob, _ = tm.matcher_blob(ad, rate, O)
out6 = np.zeros((len(surv), 6), dtype=np.int32)
status = np.zeros(len(surv), dtype=np.uint8)
cls = np.zeros(len(surv), dtype=np.uint8)
jend = np.zeros(len(surv), dtype=np.int32)
rc = model.bm_locate_batch(blob, s_seqs.ctypes.data, s_off.ctypes.data, len(surv), j0.ctypes.data, out6.ctypes.data,
                           status.ctypes.data, cls.ctypes.data, None, 16, jend.ctypes.data, -1)
assert rc == 0
if "--refined" in sys.argv:
    # exactness: the windowed scan's results (shortcut classes + windowed DP of the model) against the oracle's locate
    oa = orc.Aligner(ad, rate, 14, False, False, 1, O)
    want6, want_st = oa.locate_batch(s_seqs, s_off)
    bad = np.flatnonzero((status != want_st) | (out6 != want6).any(axis=1))
    print(f"refined windows against the oracle's locate: {len(bad)} of {len(surv)} survivors differ")
    assert len(bad) == 0, (bad[:5], out6[bad[:5]], want6[bad[:5]])
j0a = np.maximum(0, L - ((L - j0 + 15) & ~15))                      # bs_align_window
need = (jend - j0a + 15) // 16                                      # chunks this lane is at work
if "--end-bound" in sys.argv:
    # What would an END of the window be worth?  A candidate of the last row (cost <= k over all m rows) holds one of the
    # k + 1 chunks of the whole-read search set unedited, so no such candidate ends behind (last end of chunk c) +
    # (m - end offset of c) + k; rows of the last column need a hit of one of the tail sets.  A survivor without a tail hit
    # could stop at that column.  (The scan's own early stop, bs_may_stop, already ends the lanes that found a candidate.)
    R = s_seqs.reshape(len(surv), L)
    sets = create_positions_and_kmers(ad, O, rate, True, False)
    bound = np.zeros(len(surv), dtype=np.int64)
    tail_hit = np.zeros(len(surv), dtype=bool)
    for start, stop, kmers in sets:
        for km in kmers:
            ln = len(km)
            hit = np.ones((len(surv), L - ln + 1), dtype=bool)
            for t in range(ln):
                hit &= R[:, t:L - ln + 1 + t] == ord(km[t])
            if start < 0:
                tail_hit |= hit[:, max(0, L + start):].any(axis=1)
            else:
                e_off = ad.index(km) + ln
                last_end = np.where(hit.any(axis=1), L - ln - hit[:, ::-1].argmax(axis=1) + ln, 0)
                if "--coarse" in sys.argv:
                    # what the prefilter can say for one instruction per chunk: the 16-column chunk of the read's LAST
                    # whole-read hit, whichever k-mer it was
                    bound = np.maximum(bound, np.where(last_end > 0, ((last_end + 15) & ~15) + (m - 8) + k, 0))
                else:
                    bound = np.maximum(bound, np.where(last_end > 0, last_end + (m - e_off) + k, 0))
    bound = np.where(tail_hit, L, np.minimum(bound, L))
    need_b = np.minimum(need, np.maximum(0, (bound - j0a + 15) // 16))
    print(f"end bound: survivors without a tail-set hit {1 - tail_hit.mean():.3f}; lane-chunks {int(need.sum())} -> {int(need_b.sum())} "
          f"({need_b.sum() / need.sum():.3f})")
    need = need_b
    jend = np.minimum(jend, np.maximum(j0a, bound)).astype(jend.dtype)
names = ["NONE", "EXACT_FULL", "EXACT_TAIL", "DP", "SUBS_FULL", "INDEL1_FULL"]
c7 = cls & 7
print("classes:", {names[i] if i < len(names) else i: round(float((c7 == i).mean()), 3) for i in np.unique(c7)},
      "stopped early:", round(float(((cls & 8) != 0).mean()), 3))
print(f"window start (aligned): mean {j0a.mean():.1f}; chunks needed per survivor: mean {need.mean():.2f}, "
      f"histogram {np.bincount(need, minlength=11).tolist()}")
# the queue: tiles of 8192 reads, survivors of a tile sorted by key (counting sort: stable)
if "--refined" in sys.argv and "--sort-refined" in sys.argv:
    key = (j0 + m + k + 1) >> 2                                     # the queue sorted by the refined start
order = np.lexsort((np.arange(len(surv)), key, surv // TILE))
need_q, jend_q = need[order], jend[order]
n_w = (len(order) + WAVE - 1) // WAVE
pad = n_w * WAVE - len(order)
nw = np.concatenate([need_q, np.zeros(pad, dtype=need_q.dtype)]).reshape(n_w, WAVE)
lane_chunks = int(nw.sum())
wave_chunks = int(nw.max(axis=1).sum())
print(f"lock step, no straggler rule: {wave_chunks} wave-chunks x 64 lanes for {lane_chunks} lane-chunks: "
      f"utilisation {lane_chunks / (64 * wave_chunks):.3f}")
# the straggler rule: at the head of chunk c, if <= 12 lanes are left and some have >= 32 columns to go, those leave
# (they are rescanned from their window start in waves packed 64 deep); the wave runs on for the others
jq = np.concatenate([jend_q, np.zeros(pad, dtype=jend_q.dtype)]).reshape(n_w, WAVE)
j0q = np.concatenate([j0a[order], np.zeros(pad, dtype=j0a.dtype)]).reshape(n_w, WAVE)
first_wave_chunks = 0
strag_need = []
for w in range(n_w):
    nd, je, js = nw[w].copy(), jq[w], j0q[w]
    c = 0
    while True:
        left = nd > c
        if not left.any():
            break
        if left.sum() <= RETRY_AT:
            # columns to go to the READ END decide (n - j >= 32), as in the kernel
            far = left & (L - (js + 16 * c) >= 32)
            if far.any():
                strag_need += nd[far].tolist()
                nd[far] = 0
                continue
        c += 1
    first_wave_chunks += c
strag_need = np.sort(np.array(strag_need, dtype=np.int64))[::-1]
sw = (len(strag_need) + WAVE - 1) // WAVE
strag_chunks = int(sum(strag_need[i * WAVE:(i + 1) * WAVE].max() for i in range(sw))) if sw else 0
tot = first_wave_chunks + strag_chunks
print(f"with the straggler rule: {first_wave_chunks} + {strag_chunks} (second launch, {len(strag_need)} reads) = {tot} wave-chunks: "
      f"utilisation {(lane_chunks + int(strag_need.sum())) / (64 * tot):.3f} of executed lanes, "
      f"{lane_chunks / (64 * tot):.3f} counting the rescans as overhead")
# bounds for re-packing
print(f"perfect packing (every executed lane useful): {(lane_chunks + 63) // 64} wave-chunks "
      f"= {((lane_chunks + 63) // 64) / tot:.3f} of today's")
# sorting each tile's survivors by (key, then needed chunks) is not available before the scan; but grouping by key
# alone is what the queue does.  What if waves were formed from equal-need lanes (oracle knowledge, upper bound of
# any static grouping)?
need_sorted = np.sort(need)[::-1]
nsw = np.concatenate([need_sorted, np.zeros(pad, dtype=need_sorted.dtype)]).reshape(n_w, WAVE)
print(f"waves of equal need (clairvoyant static grouping): {int(nsw.max(axis=1).sum())} wave-chunks")
# the classification of the last column (bs32_finish: a 33-row loop, ~770 VALU instructions per wave in the ISA) is needed
# by the lanes that reached the read end without an exact match or an early stop; how many waves hold none?
needs_finish = ~(((cls & 7) == 1) | ((cls & 8) != 0))
nf = np.concatenate([needs_finish[order], np.zeros(pad, dtype=bool)]).reshape(n_w, WAVE)
print(f"lanes that need the last-column classification: {needs_finish.mean():.3f}; waves with none: "
      f"{(~nf.any(axis=1)).mean():.3f}; lanes per wave that need it: mean {nf.sum(axis=1).mean():.1f}")
