"""Parity at the sizes the driver measures, and the multi-rank path under the driver's eye (round-5 review items 1 and 5).

* The BASELINE workloads at 30 M reads (4.5 GB of reads per mate): blocks of rows on both sides of byte 2^31 and byte 2^32 of
  the read buffer, the middle and the LAST rows of the batch against the oracle -- the kernels rebase their 31-bit buffer
  resources per piece, draw tiles from device counters, flush a last tile and a straggler list: every one of those has a
  large-index path the small parity cases never reach (reference semantics: _align.pyx:298-587).
* `bench.py --gpus 2` (the driver's launch form, both ranks on the one device of this box): one JSON line, two ranks.
* Two processes and two threads on one device, different plans, concurrently, each against the oracle.
GPU only."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRUSEQ = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"


@pytest.mark.parametrize("config", ["C2", "C3", "C4", "C5"])
def test_blocks_on_both_sides_of_4GB_against_the_oracle(hip, orc, config):
    import torch
    sys.path.insert(0, ROOT)
    import bench
    n = 30_000_000                                       # 4.5 GB of reads: byte 2^32 is read 28 633 115's
    wl = bench.Workload(config, n, 0, torch.device("cuda", 0), None)
    blocks = wl.parity_blocks(100_000)
    L = 150
    assert blocks[0][0] == 0 and blocks[-1][0] + blocks[-1][1] == n
    assert any(s * L < (1 << 31) < (s + r) * L for s, r in blocks), blocks           # a block straddles byte 2^31
    assert any(s * L > (1 << 32) for s, r in blocks[:-1]), blocks                    # one starts behind byte 2^32
    wl.step()
    torch.cuda.synchronize()
    ok, what = wl.parity(100_000)
    assert ok, what
    assert f"{len(blocks)} blocks" in what
    # ... and a second step over the same batch gives the same rows everywhere (stale counters, queues, pools)
    first = [(o.out6.clone(), o.status.clone()) for o in wl.outs]
    wl.step()
    torch.cuda.synchronize()
    for (a6, ast), o in zip(first, wl.outs):
        assert torch.equal(ast, o.status) and torch.equal(a6, o.out6)


def test_ragged_views_blocks_at_scale(hip, orc):
    """the RV forms (views inside a uniform batch) and the frames (the same reads packed back to back with an offsets array)
    at 30 M reads, one adapter and 96: the same scattered blocks"""
    import torch
    from cutadapt_amd import _lib
    sys.path.insert(0, ROOT)
    import bench
    n = 30_000_000
    for config in ("C2", "C4"):
        for mode in ("views", "packed"):
            wl = bench.Workload(config, n, 0, torch.device("cuda", 0), None, mode)
            wl.step()
            torch.cuda.synchronize()
            if config == "C4":
                assert _lib.last_multi_path() == "stream", (mode, _lib.last_multi_path())
            ok, what = wl.parity(50_000)
            assert ok, (mode, what)
            del wl
            torch.cuda.empty_cache()


def test_bench_two_ranks_one_json_line(hip):
    """the driver's N = 2 launch form (`python bench.py --gpus 2`: bench.py starts the ranks itself), both ranks on this
    box's one device (--oversubscribe: plumbing, not a measurement): rendezvous on 127.0.0.1, contiguous read ranges, gloo
    barriers, max-over-ranks time, ONE JSON line on rank 0 with both ranks' rates, parity, roofline and cpu_baseline"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe", "--reads", "2000000",
           "--steps", "2", "--warmup", "1", "--cpu-seconds", "2", "--check-reads", "60000"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert len(line["config"]["per_rank_rate"]) == 2 and all(v > 0 for v in line["config"]["per_rank_rate"])
    assert line["config"]["parity_check"].startswith("ok"), line["config"]["parity_check"]
    assert line["config"]["units_per_gpu"] == 2_000_000
    # whole-job value = both ranks' reads over the slower rank's time
    assert line["value"] == pytest.approx(2 * 2_000_000 * 2 / (line["ms_per_step"] * 2 * 1e-3) / 1e6, rel=1e-6)
    assert line["roofline"]["frac"] > 0 and line["roofline"]["kernel"]
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "reference"
    # rank 1 owns reads [2 M, 4 M): its parity is not on the line, so check its shard here through the same generator
    import torch
    sys.path.insert(0, ROOT)
    import bench
    wl = bench.Workload("C2", 2_000_000, 1, torch.device("cuda", 0), None)
    wl.step()
    torch.cuda.synchronize()
    ok, what = wl.parity(60_000, 2_000_000)
    assert ok, what


_WORKER = r"""
import os, random, sys
sys.path.insert(0, {root!r})
import numpy as np, torch
from cutadapt_amd import _lib
from cutadapt_amd import adapters as A
from cutadapt_amd.batch import ReadBatch, match_batch
from oracle import oracle as orc
which = int(sys.argv[1])
rng = random.Random(900 + which)
many = ["".join(rng.choice("ACGT") for _ in range(33)) for _ in range(24 if which else 1)]
if not which:
    many = [{truseq!r}]
ads = [A.BackAdapter(s, max_errors=0.1, min_overlap=3) for s in many]
plan = _lib.Plan([a.matcher_spec() for a in ads])
n = 400_000
batch = ReadBatch.synthetic(n, 150, many, seed=50 + which)
last = None
for it in range(12):
    res = match_batch(plan, batch)
    torch.cuda.synchronize()
    if last is not None:
        assert torch.equal(res.out6, last[0]) and torch.equal(res.status, last[1]), it
    last = (res.out6.clone(), res.status.clone())
m = 40_000
seqs, offsets = orc.synth_reads(50 + which, 0, m, 150, many)
w6, wst = np.zeros((m, 6), dtype=np.int32), np.zeros(m, dtype=np.uint8)
for a in ads:
    c6, st = orc.match_batch(orc.Aligner(a.sequence, 0.1, 14, False, False, 1, 3), orc.KmerFinder(a.kmer_finder.positions_and_kmers), seqs, offsets)
    better = (st == 1) & ((wst == 0) | (c6[:, 4] > w6[:, 4]) | ((c6[:, 4] == w6[:, 4]) & (c6[:, 5] < w6[:, 5])))
    w6[better] = c6[better]; wst[better] = 1
assert np.array_equal(last[1][:m].cpu().numpy(), wst) and np.array_equal(last[0][:m].cpu().numpy(), w6)
print("OK", which, int(wst.sum()))
"""


def test_two_processes_on_one_device_with_different_plans(hip):
    """two PROCESSES share the device (what `--gpus N --oversubscribe` and several feeder processes per GPU do): a
    one-adapter plan and a 24-adapter streaming plan, twelve calls each at the same time, results stable from call to call
    and equal to the oracle"""
    script = _WORKER.format(root=ROOT, truseq=TRUSEQ)
    procs = [subprocess.Popen([sys.executable, "-c", script, str(w)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for w in (0, 1)]
    for w, p in enumerate(procs):
        out, err = p.communicate(timeout=900)
        assert p.returncode == 0 and f"OK {w}" in out, (out[-1000:], err[-3000:])


def test_two_threads_on_one_device_with_different_plans(hip, orc):
    """two host THREADS of one process, a stream and a plan each (per-device attribute flags under a lock, plan tables
    replicated once per device, thread-local scratch of the per-read calls): concurrent batch calls + per-read calls,
    each against the oracle"""
    import random
    import torch
    from cutadapt_amd import _lib
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch, match_batch
    errors, results = [], {}
    start = threading.Barrier(2)

    def work(which):
        try:
            torch.cuda.set_device(0)
            rng = random.Random(700 + which)
            many = [TRUSEQ] if which == 0 else ["".join(rng.choice("ACGT") for _ in range(33)) for _ in range(24)]
            ads = [A.BackAdapter(s, max_errors=0.1, min_overlap=3) for s in many]
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                plan = _lib.Plan([a.matcher_spec() for a in ads])
                n = 300_000
                batch = ReadBatch.synthetic(n, 150, many, seed=60 + which)
                start.wait(timeout=120)
                last = None
                for it in range(10):
                    res = match_batch(plan, batch)
                    stream.synchronize()
                    if last is not None:
                        assert torch.equal(res.out6, last[0]) and torch.equal(res.status, last[1]), (which, it)
                    last = (res.out6.clone(), res.status.clone())
                    # a per-read call in between (k_tiny: its scratch is thread-local)
                    mt = ads[0].match_to("ACGTTTGACCA" + many[0][:20])
                    assert mt is not None and (mt.rstart, mt.rstop, mt.errors) == (11, 31, 0), which
                stream.synchronize()
            results[which] = (many, ads, last[0].cpu().numpy(), last[1].cpu().numpy())
        except BaseException as exc:            # noqa: BLE001 (reported by the main thread)
            errors.append((which, repr(exc)))
            try:
                start.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=work, args=(w,)) for w in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    assert not errors, errors
    m = 40_000
    for which, (many, ads, g6, gst) in results.items():
        seqs, offsets = orc.synth_reads(60 + which, 0, m, 150, many)
        w6, wst = np.zeros((m, 6), dtype=np.int32), np.zeros(m, dtype=np.uint8)
        for a in ads:
            c6, st = orc.match_batch(orc.Aligner(a.sequence, 0.1, 14, False, False, 1, 3),
                                     orc.KmerFinder(a.kmer_finder.positions_and_kmers), seqs, offsets)
            better = (st == 1) & ((wst == 0) | (c6[:, 4] > w6[:, 4]) | ((c6[:, 4] == w6[:, 4]) & (c6[:, 5] < w6[:, 5])))
            w6[better] = c6[better]
            wst[better] = 1
        assert np.array_equal(gst[:m], wst) and np.array_equal(g6[:m], w6), which
        assert wst.sum() > 5000
