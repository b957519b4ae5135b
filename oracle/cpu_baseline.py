"""CPU baseline for bench.py (TEST/MEASUREMENT INFRASTRUCTURE -- never the product path).

Times the reference's own hot path -- ``BackAdapter.match_to()`` of the compiled reference in
oracle/_ref (kind "reference"), or the C restatement in oracle/cutadapt_oracle.c when the
compiled reference is not available (kind "port") -- on the host cores with one worker
process per core, the same data-parallel shape as the reference's ParallelPipelineRunner
(reference src/cutadapt/runners.py:275-412): every worker holds its own adapter object and
a contiguous shard of pre-generated reads; only the matching loop is timed (no FASTQ I/O),
throughput = reads / slowest worker.

Workers are plain subprocesses (``python -m oracle.cpu_baseline --worker ...``) with a hard
timeout, so a broken worker can never hang the benchmark.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cgroup_cpu_limit():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota) or None."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return float(quota) / float(period)
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0:
            return quota / period
    except Exception:
        pass
    return None


def available_cores() -> int:
    """hardware threads this process may actually use: affinity mask, capped by the cgroup quota"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    lim = cgroup_cpu_limit()
    if lim is not None:
        n = max(1, min(n, int(lim + 0.5)))
    return n


CHUNK = 200_000   # reads generated / converted / matched at a time inside a worker (bounds memory)


def worker(kind, seed, first, n_reads, read_len, adapter_seq, max_errors, min_overlap, gen,
           budget_seconds=None):
    """match reads [first, first+n_reads) chunk by chunk; only the matching loops are timed.
    With budget_seconds the worker stops after that much matching time (or 3x that much wall
    time), whatever it has processed by then.  Returns (seconds, hits, reads_done)."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    if kind == "reference":
        from oracle import ref_loader
        ref = ref_loader.load()
        if ref is None:
            raise RuntimeError("oracle/_ref not available")
        adapter = ref.adapters.BackAdapter(adapter_seq, max_errors=max_errors, min_overlap=min_overlap)
        match_to = adapter.match_to
    else:
        from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
        a = orc.Aligner(adapter_seq, max_errors, 14, False, False, 1, min_overlap)
        f = orc.KmerFinder(create_positions_and_kmers(adapter_seq, min_overlap, max_errors, True, False))
    total_dt, hits, done = 0.0, 0, 0
    wall0 = time.perf_counter()
    for start in range(0, n_reads, CHUNK):
        if budget_seconds is not None and (total_dt >= budget_seconds or
                                           time.perf_counter() - wall0 >= 3 * budget_seconds):
            break
        cnt = min(CHUNK, n_reads - start)
        seqs, offsets = orc.synth_reads(seed, first + start, cnt, read_len, [adapter_seq],
                                        gen["p_adapter"], gen["p_edit"], gen["p_n"])
        if kind == "reference":
            raw = seqs.tobytes()
            reads = [raw[i * read_len:(i + 1) * read_len].decode("ascii") for i in range(cnt)]
            t0 = time.perf_counter()
            for r in reads:
                if match_to(r) is not None:
                    hits += 1
            total_dt += time.perf_counter() - t0
        else:
            t0 = time.perf_counter()
            _, status = orc.match_batch(a, f, seqs, offsets)
            total_dt += time.perf_counter() - t0
            hits += int((status == 1).sum())
        done += cnt
    return total_dt, hits, done


def _spawn(job: dict):
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    # workers never touch the GPU
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    return subprocess.Popen([sys.executable, "-m", "oracle.cpu_baseline", "--worker", json.dumps(job)],
                            cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def run(seed: int, read_len: int, adapter_seq: str, max_errors: float, min_overlap: int, gen: dict,
        target_seconds: float = 12.0, max_reads: int = 4_000_000_000, timeout: float = 150.0):
    """Returns dict(value=Mreads/s, cores=..., kind=..., sample=...)."""
    from oracle import oracle as orc
    from oracle import ref_loader
    orc.lib()
    kind = "reference" if ref_loader.load() is not None else "port"
    cores = available_cores()
    # calibrate on one core in-process, then size the sample for ~target_seconds on all cores
    probe = 20000
    dt, _, _ = worker(kind, seed, 0, probe, read_len, adapter_seq, max_errors, min_overlap, gen)
    rate1 = probe / max(dt, 1e-6)
    # every worker gets a disjoint range big enough for the time budget and stops on the clock
    per_worker = int(min(max(rate1 * target_seconds * 1.5, 20000), max_reads / cores))
    base = {"kind": kind, "seed": seed, "n_reads": per_worker, "read_len": read_len,
            "adapter_seq": adapter_seq, "max_errors": max_errors, "min_overlap": min_overlap, "gen": gen,
            "budget_seconds": target_seconds}
    procs = [_spawn(dict(base, first=w * per_worker)) for w in range(cores)]
    deadline = time.time() + timeout
    results = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=max(1.0, deadline - time.time()))
            if p.returncode != 0:
                raise RuntimeError("cpu_baseline worker failed: " + err.decode(errors="replace")[-500:])
            results.append(json.loads(out.decode().strip().splitlines()[-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    # workers run concurrently for the same time budget: aggregate rate = sum of per-worker rates
    total = sum(r["reads"] for r in results)
    value = sum(r["reads"] / max(r["seconds"], 1e-9) for r in results) / 1e6
    return {
        "value": value,
        "unit": "Mreads/s",
        "cores": cores,
        "kind": kind,
        "sample": f"{total} reads of the same synthetic workload, {cores} concurrent worker processes each "
                  f"matching its own read range for ~{target_seconds:.0f} s (match_to() loop only, no I/O); "
                  f"sum of per-worker rates; 1-core probe {rate1 / 1e6:.3f} Mreads/s",
        "hit_fraction": sum(r["hits"] for r in results) / max(total, 1),
    }


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--worker":
        j = json.loads(sys.argv[2])
        seconds, hits, done = worker(j["kind"], j["seed"], j["first"], j["n_reads"], j["read_len"],
                                     j["adapter_seq"], j["max_errors"], j["min_overlap"], j["gen"],
                                     j.get("budget_seconds"))
        print(json.dumps({"seconds": seconds, "hits": hits, "reads": done}))
    else:
        res = run(2, 150, "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", 0.1, 3,
                  {"p_adapter": 0.25, "p_edit": 0.02, "p_n": 0.005}, target_seconds=3.0)
        print(json.dumps(res))
