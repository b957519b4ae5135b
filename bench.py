#!/usr/bin/env python3
"""bench.py -- throughput of the adapter-matching hot path on MI355X (BASELINE.json metric).

    python bench.py [--config C2|C3|C4|C5] [--gpus N] [--steps K] [--warmup W] [--reads R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload = BASELINE.json configs[1] ("C2"): 100 M synthetic 150 bp reads per
GPU, single 3' adapter (TruSeq, 33 bp), e = 0.1, min_overlap 3; 25 % of the reads carry an (edited)
adapter copy, 0.5 % of the bases are N.  --config selects the other BASELINE configs (C3 linked adapter
with IUPAC wildcards, C4 96 adapters, C5 paired-end 2 x 2 adapters); each prints the same JSON shape.
A *step* is one pass of the hot path over the whole batch (k-mer prefilter -> survivor queue -> cost scan
-> banded DP -> 6-tuple per read), reads and results resident in HBM.

Multi-GPU: reads shard embarrassingly (rank r owns read indices [r*R, (r+1)*R)), there is NO data-path
collective and no RCCL: the ranks only meet at the start/stop barriers and for the max-over-ranks time,
over a gloo process group on 127.0.0.1.  `python bench.py --gpus N` without a launcher starts the N ranks
itself (one process per device) and refuses when fewer than N devices are visible.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8.0 TB/s spec
N_SIMD = 256 * 4                   # MI355X: 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4
DEFAULT_READS = {"C2": 100_000_000, "C3": 100_000_000, "C4": 100_000_000, "C5": 125_000_000}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=None,
                    help="units (reads, or read pairs for C5) per GPU; default: the BASELINE size of the config")
    ap.add_argument("--read-len", type=int, default=None,
                    help="read length of the synthetic reads (default 150, what BASELINE quotes; 250 / 300: MiSeq-style runs)")
    ap.add_argument("--p-adapter", type=float, default=None,
                    help="override the adapter fraction of the read model (SURVEY 8(d): 0 and 1 are the extremes; "
                         "the headline number uses the default 0.25)")
    ap.add_argument("--ragged", action="store_true",
                    help="cut the reads to 30 .. read_len characters at their 3' end (a hash of the read index): the batch a "
                         "pipeline holds behind -q / -u -- views (starts + lengths) into the sequencer's batch at its uniform "
                         "stride, streamed end-aligned (C2, C4, C5)")
    ap.add_argument("--ragged-packed", action="store_true",
                    help="the same reads copied into a packed buffer with an offsets array: cah_match_batch_frames")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--check-reads", type=int, default=250_000, help="reads compared with the oracle (untimed)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default run (C2, one GPU) only: do not append the C3 / C4 / C5 lines (other_configs)")
    ap.add_argument("--other-steps", type=int, default=3, help="timed steps of each other_configs entry")
    ap.add_argument("--other-cpu-seconds", type=float, default=4.0)
    ap.add_argument("--oversubscribe", action="store_true",
                    help="TEST ONLY: let ranks share devices (rank r uses device r mod visible) so that the multi-rank "
                         "path can be exercised on a box with fewer GPUs; such a line is not a measurement")
    return ap.parse_args(argv)


# -------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torchrun starts the N ranks itself
# -------------------------------------------------------------------------------------------------
def self_launch(args) -> int:
    import torch
    visible = torch.cuda.device_count()
    if visible < args.gpus and not (args.oversubscribe and visible >= 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} requested but only {visible} HIP device(s) are visible; "
                         f"refusing to run a {args.gpus}-GPU benchmark on fewer devices")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, p.wait())
    return rc


# -------------------------------------------------------------------------------------------------
# per-config steps (device-resident)
# -------------------------------------------------------------------------------------------------
class Workload:
    def __init__(self, config, n, rank, device, gen, ragged=False):
        import torch
        from cutadapt_amd import _lib, workloads
        from cutadapt_amd import adapters as A
        from cutadapt_amd.batch import BatchResult
        self.config, self.n, self.spec = config, n, workloads.SPECS[config]
        self.gen = gen
        self.ragged = ragged
        kind = self.spec["kind"]
        first = rank * n

        def result():
            return BatchResult(torch.empty((n, 6), dtype=torch.int32, device=device),
                               torch.empty(n, dtype=torch.uint8, device=device),
                               torch.empty(n, dtype=torch.int32, device=device))
        back = [A.BackAdapter(s, max_errors=0.1, min_overlap=3) for s in self.spec["adapters"]]
        self.adapters = back
        self.batches = [workloads.device_batch(config, n, first, 0, device, gen)]
        self.outs = [result()]
        if kind == "single":
            self.plans = [back[0]._fused_plan]
        elif kind == "linked":
            self.front = A.PrefixAdapter(self.spec["front"], max_errors=0.1)
            self.plans = [self.front._fused_plan, back[0]._fused_plan]
            self.outs.append(result())
        elif kind == "multi":
            self.plans = [_lib.Plan([a.matcher_spec() for a in back])]
        elif kind == "paired":
            self.adapters2 = [A.BackAdapter(s, max_errors=0.1, min_overlap=3) for s in self.spec["adapters2"]]
            self.plans = [_lib.Plan([a.matcher_spec() for a in back]), _lib.Plan([a.matcher_spec() for a in self.adapters2])]
            self.batches.append(workloads.device_batch(config, n, first, 1, device, gen))
            self.outs.append(result())
        if ragged:
            if kind == "linked":
                raise SystemExit("bench.py: --ragged serves C2, C4 and C5")
            cut = workloads.ragged_device_batch if ragged == "packed" else workloads.ragged_view_batch
            self.batches = [cut(b, first) for b in self.batches]
            torch.cuda.empty_cache()
        for b in self.batches:
            b.workspace()

    def step(self):
        from cutadapt_amd.batch import linked_match_batch, match_batch
        kind = self.spec["kind"]
        if kind == "linked":
            linked_match_batch(self.plans[0], self.plans[1], self.batches[0], self.outs[0], self.outs[1])
        elif kind == "paired":
            match_batch(self.plans[0], self.batches[0], self.outs[0])
            match_batch(self.plans[1], self.batches[1], self.outs[1])
        else:
            match_batch(self.plans[0], self.batches[0], self.outs[0])

    # ---- parity sample: scattered blocks of rank 0's batch, bit-compared with the oracle -------------
    def parity_blocks(self, m):
        """Row ranges [start, start + rows) of the batch that the oracle re-computes: m rows in all, as (up to) five
        blocks -- the first rows, a block that straddles byte 2^31 of the read buffer, a block that starts behind byte
        2^32, the middle of the batch, and the LAST rows (last tile, straggler flush).  Blocks the batch is too small
        for fall away; overlapping ones are merged."""
        n = self.n
        from cutadapt_amd import workloads
        L = workloads.READ_LEN
        if m >= n:
            return [(0, n)]
        rows = max(1, m // 5)
        wanted = [0, (1 << 31) // L - rows // 2, (1 << 32) // L + 1, n // 2 - rows // 2, n - rows]
        blocks = []
        for s in sorted(set(max(0, min(w, n - rows)) for w in wanted if 0 <= w <= n - rows)):
            if blocks and s < blocks[-1][0] + blocks[-1][1]:
                end = max(blocks[-1][0] + blocks[-1][1], s + rows)
                blocks[-1] = (blocks[-1][0], end - blocks[-1][0])
            else:
                blocks.append((s, rows))
        return blocks

    def parity(self, m, first_index=0):
        import numpy as np
        from cutadapt_amd import workloads
        from oracle import host_workloads
        from oracle import oracle as orc
        kind = self.spec["kind"]
        gen = self.gen
        L = workloads.READ_LEN
        verdicts = {}

        def oracle_multi(adapters, seqs, offsets):
            m = len(offsets) - 1
            want6 = np.zeros((m, 6), dtype=np.int32)
            want_st = np.zeros(m, dtype=np.uint8)
            want_best = np.full(m, -1, dtype=np.int32)
            for idx, ad in enumerate(adapters):
                oa = orc.Aligner(ad.sequence, ad.max_error_rate, 14, False, False, 1, ad.min_overlap)
                of = orc.KmerFinder(ad.kmer_finder.positions_and_kmers)
                c6, st = orc.match_batch(oa, of, seqs, offsets)
                f = st == 1
                better = f & ((want_st == 0) | (c6[:, 4] > want6[:, 4]) | ((c6[:, 4] == want6[:, 4]) & (c6[:, 5] < want6[:, 5])))
                want6[better] = c6[better]
                want_best[better] = idx
                want_st[better] = 1
            return want6, want_st, want_best

        def same(out, s, want6, want_st, what, want_best=None):
            m = len(want_st)
            ok = np.array_equal(out.out6[s:s + m].cpu().numpy(), want6) and np.array_equal(out.status[s:s + m].cpu().numpy(), want_st)
            if ok and want_best is not None:
                ok = np.array_equal(out.best_adapter[s:s + m].cpu().numpy()[want_st == 1], want_best[want_st == 1])
            verdicts[what] = verdicts.get(what, True) and ok
            return ok

        ok = True
        blocks = self.parity_blocks(m)
        packed_off = None
        if self.ragged == "packed":
            packed_off = self.batches[0].offsets
        for s, rows in blocks:
            for mate, batch in enumerate(self.batches):
                seqs, offsets = host_workloads.host_reads(self.config, first_index + s, rows, mate, gen)
                if self.ragged == "views":
                    ok &= np.array_equal(batch.seqs[s * L: s * L + len(seqs)].cpu().numpy(), seqs)   # generator twin (the parent batch)
                if self.ragged:
                    seqs, offsets = host_workloads.host_ragged(seqs, offsets, first_index + s)
                if self.ragged == "packed":
                    b0 = int(batch.offsets[s].item())
                    ok &= np.array_equal(batch.seqs[b0: b0 + len(seqs)].cpu().numpy(), seqs)
                elif not self.ragged:
                    ok &= np.array_equal(batch.seqs[s * L: s * L + len(seqs)].cpu().numpy(), seqs)   # generator twin
                if kind in ("single", "multi", "paired"):
                    ads = self.adapters if mate == 0 else self.adapters2
                    want6, want_st, want_best = oracle_multi(ads, seqs, offsets)
                    ok &= same(self.outs[mate], s, want6, want_st, f"mate {mate + 1}" if kind == "paired" else "tuples",
                               want_best if len(ads) > 1 else None)
                else:
                    fs = self.front.matcher_spec()
                    ofa = orc.Aligner(fs.sequence, fs.max_error_rate, fs.flags, fs.wildcard_ref, fs.wildcard_query,
                                      fs.indel_cost, fs.min_overlap)
                    off = orc.KmerFinder(fs.kmer_sets, fs.kmer_ref_wildcards, fs.kmer_query_wildcards) if fs.kmer_sets is not None else None
                    f6, fst = orc.match_batch(ofa, off, seqs, offsets)
                    ok &= same(self.outs[0], s, f6, fst, "front stage")
                    # back stage on read[rstop:] (reference adapters.py:1222-1224)
                    starts = np.where(fst == 1, f6[:, 3], 0).astype(np.int64)
                    subs = [bytes(seqs[offsets[i] + starts[i]:offsets[i + 1]]) for i in range(rows)]
                    s2, o2 = orc.pack_reads(subs)
                    b6, bst, _ = oracle_multi(self.adapters, s2, o2)
                    ok &= same(self.outs[1], s, b6, bst, "back stage")
        total = sum(r for _, r in blocks)
        where = ", ".join(f"[{s}, {s + r})" for s, r in blocks)
        checked = ", ".join(f"{k}: {'ok' if v else 'MISMATCH'}" for k, v in verdicts.items())
        return ok, (f"{'ok' if ok else 'MISMATCH'} ({total} reads{' per mate' if kind == 'paired' else ''} bit-compared with "
                    f"the oracle in {len(blocks)} blocks of rows {where} of the timed steps' outputs -- byte offsets up to "
                    f"{(blocks[-1][0] + blocks[-1][1]) * L:,}: {checked})")


def csrc_hash() -> str:
    """sha256 over the kernel / host sources on disk (cutadapt_amd.build.source_hash): what a library built now would carry"""
    from cutadapt_amd import build
    return build.source_hash()


def library_hash() -> str:
    """the hash the LOADED library carries (cah_build_id): the sources it was actually built from"""
    from cutadapt_amd import _lib
    return _lib.build_id()


def profile_fields(config, n, dom, dom_launch_ms):
    profile_fields.whole_step = None
    """roofline.traffic / roofline.valu from the committed PMC profile (profiles/pmc_latest.json, written by
    profiles/summarize_r03.py) -- only if it was taken on THIS source tree and THIS workload; otherwise null + why."""
    tpath = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(tpath) as f:
            tj = json.load(f).get(config, {})
    except Exception:
        return None, None, "no profile (profiles/pmc_latest.json unreadable)"
    if not tj:
        return None, None, "no PMC profile of this config"
    if tj.get("reads_per_gpu") != n:
        return None, None, "the PMC profile was taken at another batch size"
    if tj.get("csrc_sha256") != library_hash():
        return None, None, "stale profile: profiles/pmc_latest.json was taken on another build of the library (cah_build_id differs)"
    k = tj.get("kernels", {}).get(dom)
    if not k:
        return None, None, "the PMC profile has no entry for the dominant kernel"
    traffic = k.get("hbm_bytes_per_launch")
    # ... and of the WHOLE step: every kernel family's counter bytes per step (profiles/summarize_r06.py sums every launch
    # of a family over the profiled steps: hbm_bytes_per_step)
    per_family = {name: e.get("hbm_bytes_per_step") for name, e in tj.get("kernels", {}).items()}
    if per_family and all(v is not None for v in per_family.values()):
        profile_fields.whole_step = {"bytes": sum(per_family.values()), "per_kernel_family": per_family}
    valu = {x: k.get(x) for x in ("kernel_full_name", "valu_busy", "valu_insts_per_launch", "waves_per_simd",
                                  "wait_frac_of_wave_cycles", "issue_stall_frac_of_wave_cycles", "clock_ghz_profiled",
                                  "profiled_ms_per_launch", "salu_insts_per_launch", "lds_insts_per_launch")}
    for x in ("valu_issue_lower", "valu_issue_upper"):
        valu[x] = k.get(x)
    valu["definition"] = ("valu_issue_lower / _upper = SQ_INSTS_VALU x 2 (x 4) cycles / (SIMD-cycles the kernel was busy: "
                          "SQ_BUSY_CYCLES per SIMD), the upper one capped at 1: a wave64 VALU instruction occupies its SIMD "
                          "for 2 cycles (and / or / xor / add / v_bitop3) or 4 (shifts, SDWA, compares, selects, v_perm: "
                          "profiles/r03/valu_ubench.txt), so the true busy fraction lies between the two; valu_busy (kept "
                          "for comparison with round 3) = SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles, which counts every "
                          "instruction as 4 cycles and can exceed 1; wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES")
    valu["source"] = tj.get("source")
    return traffic, valu, "profile taken on this build of the library (cah_build_id match)"


def run_config(args, config, n, steps, warmup, rank, world, device, gen, check_reads, cpu_seconds, want_cpu, ragged=False):
    """one BASELINE config: build the workload in HBM, time `steps` passes of the hot path, parity sample, roofline
    fields, CPU baseline.  Returns the JSON-line dict (rank 0) or None (other ranks)."""
    import torch
    import torch.distributed as dist
    from cutadapt_amd import _lib, workloads
    L = _lib.lib()
    spec = workloads.SPECS[config]

    # ---- inputs resident in HBM before the timed region ------------------------------------------
    wl = Workload(config, n, rank, device, gen, ragged)
    wl.gen = gen
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        wl.step()
    barrier()
    L.cah_profile_reset()
    L.cah_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    barrier()
    L.cah_profile_enable(0)
    elapsed = elapsed_local
    per_rank = [n * steps / elapsed_local / 1e6]
    if world > 1:
        t = torch.tensor([elapsed_local], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered

    # ---- per-kernel durations from the HIP events recorded on the launch stream ------------------
    import ctypes as C
    ms = (C.c_double * _lib.PROF_N)()
    launches = (C.c_int64 * _lib.PROF_N)()
    units = (C.c_int64 * _lib.PROF_N)()
    _lib.check(L.cah_profile_read(ms, launches, units))
    L.cah_profile_reset()
    fam = {"k_filter": _lib.PROF_FILTER, "k_back_scan": _lib.PROF_SCAN, "k_dp": _lib.PROF_DP, "k_comparer": _lib.PROF_COMPARER}
    step_ms = {k: ms[i] / steps for k, i in fam.items()}
    per_launch_ms = {k: ms[i] / max(launches[i], 1) for k, i in fam.items()}
    launches_per_step = {k: launches[i] / steps for k, i in fam.items()}
    units_per_launch = {k: units[i] / max(launches[i], 1) for k, i in fam.items()}     # reads handed to one launch
    status = wl.outs[-1].status if spec["kind"] == "linked" else wl.outs[0].status
    n_match = int((status == 1).sum().item())
    n_invalid = int((status == 2).sum().item())
    survivors, dp_reads = None, None
    if spec["kind"] == "single":
        ws = wl.batches[0].workspace()
        survivors = int(ws[256:264].view(torch.int64).item())         # queue count (workspace layout, api.cpp)
        dp_reads = int(ws[768:776].view(torch.int64).item()) + int(ws[896:904].view(torch.int64).item())

    if rank != 0:
        return None

    # ---- untimed parity check against the oracle ---------------------------------------------------
    parity = None
    if check_reads > 0:
        ok, parity = wl.parity(check_reads, rank * n)
        if not ok:
            raise SystemExit(f"{config}: parity check against the oracle FAILED: " + parity)
    del wl
    torch.cuda.empty_cache()

    # algorithmic bytes per unit: read characters + 4 (offset) + 24 (result row) [+ 4: adapter index]
    bytes_per_unit = spec["bytes_per_unit"] + (workloads.READ_LEN - 150) * (2 if spec["kind"] == "paired" else 1)
    if ragged:      # mean length (30 + read_len) / 2 instead of read_len, and the 8-byte offset a ragged batch has per read
        bytes_per_unit += (8 - (workloads.READ_LEN - (workloads.RAGGED_MIN + workloads.READ_LEN) // 2)) * (2 if spec["kind"] == "paired" else 1)
    total_units = n * world * steps
    value = total_units / elapsed / 1e6
    # dominant kernel family: the one with the largest share of a step
    dom = max(step_ms, key=lambda k: step_ms[k])
    dom_launch_ms = per_launch_ms[dom]
    reads_per_launch = units_per_launch[dom]        # (the fused multi-adapter path works in chunks of reads)
    if spec["kind"] == "single":
        reads_per_launch = {"k_filter": n, "k_back_scan": survivors, "k_dp": dp_reads if dp_reads else survivors,
                            "k_comparer": n}[dom]
    per_read_bytes = (178 if spec["kind"] != "multi" else 182) + (workloads.READ_LEN - 150)
    if ragged:
        per_read_bytes += 8 - (workloads.READ_LEN - (workloads.RAGGED_MIN + workloads.READ_LEN) // 2)
    achieved = reads_per_launch * per_read_bytes / (dom_launch_ms * 1e-3) / 1e9 if dom_launch_ms > 0 else 0.0
    kernel_sum = sum(step_ms.values())
    step_gbs = n * bytes_per_unit / (kernel_sum * 1e-3) / 1e9 if kernel_sum > 0 else 0.0
    traffic, valu, profile_note = profile_fields(config, n, dom, dom_launch_ms)
    result = {
        "metric": spec["metric"],
        "value": value,
        "unit": spec["unit"],
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {
            "workload": f"{config}: {n} x {'2 x ' if spec['kind'] == 'paired' else ''}{workloads.READ_LEN} bp synthetic "
                        f"{'read pairs' if spec['kind'] == 'paired' else 'reads'} per GPU, {spec['what']}, "
                        f"p_adapter={gen['p_adapter']}, p_edit={gen['p_edit']}, p_N={gen['p_n']}"
                        + (f", RAGGED: every read cut to {workloads.RAGGED_MIN} .. {workloads.READ_LEN} characters at its 3' end "
                           + ("(copied into a packed buffer + offsets array)" if ragged == "packed" else
                              "(views -- starts + lengths -- into the batch at its uniform stride)") if ragged else ""),
            "units_per_gpu": n,
            "read_len": workloads.READ_LEN,
            "n_adapters": len(spec["adapters"]) + len(spec.get("adapters2", [])),
            "sharding": f"{world} x contiguous read ranges, no collective on the data path (gloo barriers only)"
                        + (" -- OVERSUBSCRIBED TEST RUN, ranks share devices: not a measurement" if args.oversubscribe else ""),
            "per_rank_rate": per_rank,
            "matched_fraction": n_match / n,
            "matched_fraction_of": ("reads whose 3' adapter stage (the linked adapter's second stage) found a match"
                                    if spec["kind"] == "linked" else
                                    ("mate-1 reads with a match" if spec["kind"] == "paired" else "reads with a match")),
            "prefilter_pass_fraction": None if survivors is None else survivors / n,
            "cell_dp_fraction_of_survivors": None if not survivors else dp_reads / survivors,
            "invalid_reads": n_invalid,
            "parity_check": parity,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": dom,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "whole_step_traffic": None if not profile_fields.whole_step else dict(
                profile_fields.whole_step, over_algorithmic=profile_fields.whole_step["bytes"] / (n * bytes_per_unit),
                note="HBM bytes of ALL kernels of one step by the same PMC passes (FETCH_SIZE x 2 + WRITE_SIZE; the x 2 is "
                     "the guide's gfx950 correction for wide streaming reads and an upper bound for scattered ones)"),
            "valu": valu,
            "profile": profile_note,
            "kernel_ms_per_step": step_ms,
            "kernel_ms_per_launch": per_launch_ms,
            "launches_per_step": launches_per_step,
            "reads_per_launch": reads_per_launch,
            "algorithmic_bytes_per_unit": bytes_per_unit,
            "whole_step_GBps": step_gbs,
            "whole_step_frac": step_gbs / HBM_PEAK_GBS,
            "note": "achieved = reads of one launch of the dominant kernel x 178 (182) algorithmic bytes / its HIP-event "
                    "duration; traffic (HBM bytes per launch: FETCH_SIZE x 2 + WRITE_SIZE) and valu (counter-measured "
                    "VALU busy fraction, waves per SIMD, wait fraction) come from the committed rocprofv3 PMC passes",
        },
    }
    if want_cpu:                                       # (rank 0 only: the others returned above)
        from oracle import cpu_baseline
        try:
            result["cpu_baseline"] = cpu_baseline.run(config, gen, target_seconds=cpu_seconds)
            cb = result["cpu_baseline"]
            result["gpu_over_cpu"] = {
                "measured": value / cb["value"],
                "vs_all_host_threads_linear_extrapolation": value / (cb["value"] / cb["cores"] * cb["host_logical_cpus"]),
                "note": f"measured on {cb['cores']} worker processes (cgroup quota of this container); the second figure "
                        f"scales that linearly to all {cb['host_logical_cpus']} hardware threads of the host (an upper "
                        f"bound for the CPU: SMT siblings and memory bandwidth do not scale linearly)",
            }
        except Exception as exc:        # the GPU numbers must survive a host-side hiccup
            result["cpu_baseline"] = {"value": None, "unit": spec["unit"], "cores": cpu_baseline.available_cores(),
                                      "kind": "reference", "sample": f"failed: {exc!r}"[:300]}
    return result


def main():
    args = parse_args()
    if args.read_len is not None:
        os.environ["CAH_BENCH_READ_LEN"] = str(int(args.read_len))     # (before cutadapt_amd.workloads is imported)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        # rank 0's stdout carries ONE JSON line: whatever c10d / gloo print while connecting goes to stderr
        real_stdout = os.dup(1)
        os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    if args.oversubscribe and torch.cuda.device_count() >= 1:
        local_rank = local_rank % torch.cuda.device_count()
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants device {local_rank} but only {torch.cuda.device_count()} "
                         f"device(s) are visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)     # barriers only: no RCCL on this path

    from cutadapt_amd import workloads
    gen = dict(workloads.GEN)
    if args.p_adapter is not None:
        gen["p_adapter"] = float(args.p_adapter)
    n = args.reads if args.reads is not None else DEFAULT_READS[args.config]
    ragged_mode = "packed" if args.ragged_packed else ("views" if args.ragged else False)
    result = run_config(args, args.config, n, args.steps, args.warmup, rank, world, device, gen, args.check_reads,
                        args.cpu_seconds, not args.no_cpu_baseline and not ragged_mode, ragged_mode)
    if rank == 0:
        # The default invocation (the driver's: C2, one GPU) also carries the other BASELINE configs at their BASELINE
        # sizes -- a few steps each, with their own parity sample, roofline fraction and a short CPU baseline
        if (args.config == "C2" and world == 1 and args.reads is None and args.p_adapter is None
                and args.read_len is None and not args.no_other_configs and not ragged_mode):
            others = {}
            for cfg in ("C3", "C4", "C5"):
                try:
                    r = run_config(args, cfg, DEFAULT_READS[cfg], args.other_steps, 1, 0, 1, device, gen,
                                   args.check_reads,
                                   args.other_cpu_seconds, not args.no_cpu_baseline)
                    others[cfg] = {"metric": r["metric"], "value": r["value"], "unit": r["unit"], "steps": r["steps"],
                                   "ms_per_step": r["ms_per_step"], "workload": r["config"]["workload"],
                                   "parity_check": r["config"]["parity_check"],
                                   "matched_fraction": r["config"]["matched_fraction"],
                                   "matched_fraction_of": r["config"]["matched_fraction_of"],
                                   "roofline": {k: r["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic", "whole_step_traffic",
                                                                             "kernel_ms_per_step", "launches_per_step",
                                                                             "whole_step_frac", "profile")},
                                   "cpu_baseline": r.get("cpu_baseline"),
                                   "gpu_over_cpu": (r.get("gpu_over_cpu") or {}).get("measured")}
                except SystemExit:
                    raise
                except Exception as exc:
                    others[cfg] = {"error": repr(exc)[:300]}
            result["other_configs"] = others
            # the p_adapter = 0 ("all random") and = 1 ("every read carries an adapter") extremes of
            # the headline workload, two steps each, with their own parity samples
            extremes = {}
            for pa in (0.0, 1.0):
                try:
                    g2 = dict(gen, p_adapter=pa)
                    r = run_config(args, "C2", DEFAULT_READS["C2"], 2, 1, 0, 1, device, g2, args.check_reads,
                                   0.0, False)
                    extremes[f"p_adapter_{pa:g}"] = {
                        "value": r["value"], "unit": r["unit"], "steps": r["steps"], "ms_per_step": r["ms_per_step"],
                        "workload": r["config"]["workload"], "parity_check": r["config"]["parity_check"],
                        "matched_fraction": r["config"]["matched_fraction"],
                        "prefilter_pass_fraction": r["config"]["prefilter_pass_fraction"],
                        "kernel_ms_per_step": r["roofline"]["kernel_ms_per_step"],
                        "whole_step_frac": r["roofline"]["whole_step_frac"]}
                except SystemExit:
                    raise
                except Exception as exc:
                    extremes[f"p_adapter_{pa:g}"] = {"error": repr(exc)[:300]}
            result["p_adapter_extremes"] = extremes
            # ragged batches (what a pipeline holds behind -q / -u): the same reads cut to 30 .. 150 characters, as views
            # into the uniform batch (streamed end-aligned: k_filter_stream2's RV form for one adapter, k_multi_stream's for
            # several -- round 6)
            ragged = {}
            for cfg in ("C2", "C4"):
                try:
                    r = run_config(args, cfg, DEFAULT_READS[cfg], 2, 1, 0, 1, device, gen, args.check_reads,
                                   0.0, False, "views")
                    ragged[cfg] = {"value": r["value"], "unit": r["unit"], "steps": r["steps"], "ms_per_step": r["ms_per_step"],
                                   "workload": r["config"]["workload"], "parity_check": r["config"]["parity_check"],
                                   "matched_fraction": r["config"]["matched_fraction"],
                                   "kernel_ms_per_step": r["roofline"]["kernel_ms_per_step"],
                                   "vs_uniform": r["value"] / (result["value"] if cfg == "C2" else (others.get(cfg, {}).get("value") or float("nan")))}
                except SystemExit:
                    raise
                except Exception as exc:
                    ragged[cfg] = {"error": repr(exc)[:300]}
            result["ragged"] = ragged
            # ... and the same reads copied back to back with an offsets array (no uniform stride: what ReadBatch.from_strings
            # of unequal reads, or the reads of a FASTQ chunk, look like): cah_match_batch_frames -- end-aligned frames of the
            # longest read's length, the copy gathered from the reads' ends
            packed = {}
            for cfg in ("C2", "C4"):
                try:
                    r = run_config(args, cfg, DEFAULT_READS[cfg], 2, 1, 0, 1, device, gen, args.check_reads,
                                   0.0, False, "packed")
                    packed[cfg] = {"value": r["value"], "unit": r["unit"], "steps": r["steps"], "ms_per_step": r["ms_per_step"],
                                   "workload": r["config"]["workload"], "parity_check": r["config"]["parity_check"],
                                   "matched_fraction": r["config"]["matched_fraction"],
                                   "kernel_ms_per_step": r["roofline"]["kernel_ms_per_step"],
                                   "vs_uniform": r["value"] / (result["value"] if cfg == "C2" else (others.get(cfg, {}).get("value") or float("nan")))}
                except SystemExit:
                    raise
                except Exception as exc:
                    packed[cfg] = {"error": repr(exc)[:300]}
            result["ragged_packed"] = packed
        line = json.dumps(result)
        if world > 1:
            os.write(real_stdout, (line + "\n").encode())
        else:
            print(line)
            sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
