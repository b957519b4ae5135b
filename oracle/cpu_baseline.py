"""CPU baseline for bench.py (TEST/MEASUREMENT INFRASTRUCTURE -- never the product path).

Times the reference's own hot path -- ``match_to()`` of the compiled reference's adapter objects in
oracle/_ref (kind "reference": BackAdapter for C2, LinkedAdapter for C3, MultipleAdapters for C4 and
for each mate of C5), or for C2 the C restatement in oracle/cutadapt_oracle.c when the compiled
reference is not available (kind "port") -- on the host cores with one worker
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


CHUNK = 100_000   # units generated / converted / matched at a time inside a worker (bounds memory)


def _reference_matchers(ref, config):
    """the reference's adapter objects for a config: a list with one match_to callable per mate"""
    from cutadapt_amd.workloads import SPECS
    spec = SPECS[config]
    A = ref.adapters
    backs = [A.BackAdapter(s, max_errors=0.1, min_overlap=3) for s in spec["adapters"]]
    if spec["kind"] == "single":
        return [backs[0].match_to]
    if spec["kind"] == "linked":
        front = A.PrefixAdapter(spec["front"], max_errors=0.1)
        return [A.LinkedAdapter(front, backs[0], front_required=True, back_required=False, name="c3").match_to]
    if spec["kind"] == "multi":
        return [A.MultipleAdapters(backs).match_to]
    backs2 = [A.BackAdapter(s, max_errors=0.1, min_overlap=3) for s in spec["adapters2"]]
    return [A.MultipleAdapters(backs).match_to, A.MultipleAdapters(backs2).match_to]


def worker(kind, config, first, n_units, gen, budget_seconds=None):
    """match units [first, first+n_units) chunk by chunk; only the matching loops are timed.
    With budget_seconds the worker stops after that much matching time (or 3x that much wall
    time), whatever it has processed by then.  Returns (seconds, hits, units_done)."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import host_workloads
    from oracle import oracle as orc
    from cutadapt_amd.workloads import READ_LEN, SPECS
    spec = SPECS[config]
    n_mates = 2 if spec["kind"] == "paired" else 1
    if kind == "reference":
        from oracle import ref_loader
        ref = ref_loader.load()
        if ref is None:
            raise RuntimeError("oracle/_ref not available")
        matchers = _reference_matchers(ref, config)
    else:
        if spec["kind"] != "single":
            raise RuntimeError("the C port times only the single-adapter config; build oracle/_ref for the others")
        from cutadapt_amd.kmer_heuristic import create_positions_and_kmers
        ad = spec["adapters"][0]
        a = orc.Aligner(ad, 0.1, 14, False, False, 1, 3)
        f = orc.KmerFinder(create_positions_and_kmers(ad, 3, 0.1, True, False))
    total_dt, hits, done = 0.0, 0, 0
    wall0 = time.perf_counter()
    chunk = max(1000, CHUNK // (50 if spec["kind"] == "multi" else 1))
    for start in range(0, n_units, chunk):
        if budget_seconds is not None and (total_dt >= budget_seconds or
                                           time.perf_counter() - wall0 >= 3 * budget_seconds):
            break
        cnt = min(chunk, n_units - start)
        for mate in range(n_mates):
            seqs, offsets = host_workloads.host_reads(config, first + start, cnt, mate, gen)
            if kind == "reference":
                raw = seqs.tobytes()
                reads = [raw[i * READ_LEN:(i + 1) * READ_LEN].decode("ascii") for i in range(cnt)]
                match_to = matchers[mate]
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


def run(config: str, gen: dict, target_seconds: float = 12.0, max_units: int = 4_000_000_000, timeout: float = 150.0):
    """Returns dict(value=M units/s on all usable cores, cores, host_logical_cpus, cgroup_cpu_quota,
    one_core, kind, sample)."""
    from oracle import oracle as orc
    from oracle import ref_loader
    from cutadapt_amd.workloads import SPECS
    orc.lib()
    kind = "reference" if ref_loader.load() is not None else "port"
    cores = available_cores()
    # P = 1: calibrate on one core in-process, then size the sample for ~target_seconds on all cores
    probe = 300 if SPECS[config]["kind"] == "multi" else 20000
    dt, _, done = worker(kind, config, 0, probe, gen)
    rate1 = done / max(dt, 1e-6)
    # every worker gets a disjoint range big enough for the time budget and stops on the clock
    per_worker = int(min(max(rate1 * target_seconds * 1.5, 2000), max_units / cores))
    base = {"kind": kind, "config": config, "n_units": per_worker, "gen": gen, "budget_seconds": target_seconds}
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
    # Mreads/s = reads / max worker wall time (the workers run concurrently; each stops on the
    # clock, so the slowest one's matching time is the job's); the sum of the per-worker rates is kept beside it
    total = sum(r["units"] for r in results)
    value = total / max(max(r["seconds"] for r in results), 1e-9) / 1e6
    sum_of_rates = sum(r["units"] / max(r["seconds"], 1e-9) for r in results) / 1e6
    unit = SPECS[config]["unit"]
    return {
        "value": value,
        "sum_of_worker_rates": sum_of_rates,
        "unit": unit,
        "cores": cores,
        "host_logical_cpus": os.cpu_count() or cores,
        "cgroup_cpu_quota": cgroup_cpu_limit(),
        "one_core": {"value": rate1 / 1e6, "unit": unit, "cores": 1},
        "kind": kind,
        "sample": f"{total} units of the same synthetic workload ({config}), P = {cores} concurrent worker processes "
                  f"(the container's CPU quota; the host has {os.cpu_count()} hardware threads) each matching its own "
                  f"range for ~{target_seconds:.0f} s (match_to() loops of the reference's adapter objects only, no I/O); "
                  f"value = units of all workers / the slowest worker's matching time; P = 1 probe {rate1 / 1e6:.4f} {unit}",
        # what match_to() returned a match for: the LinkedAdapter as a whole for C3, mate 1 OR mate 2 counted per read
        # for C5 -- not the same quantity as the GPU line's matched_fraction (see bench.py: matched_fraction_of)
        "hit_fraction": sum(r["hits"] for r in results) / max(total, 1),
        "hit_fraction_of": "calls of match_to() that returned a match (C3: the linked adapter as a whole; C5: per mate)",
    }


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--worker":
        j = json.loads(sys.argv[2])
        seconds, hits, done = worker(j["kind"], j["config"], j["first"], j["n_units"], j["gen"], j.get("budget_seconds"))
        print(json.dumps({"seconds": seconds, "hits": hits, "units": done}))
    else:
        cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
        print(json.dumps(run(cfg, {"p_adapter": 0.25, "p_edit": 0.02, "p_n": 0.005}, target_seconds=3.0)))
