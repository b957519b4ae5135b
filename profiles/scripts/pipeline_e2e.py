# end-to-end FASTQ -> trimmed FASTQ throughput with worker threads (in memory)
import sys, time, io, json, os
sys.path.insert(0, '.')
import numpy as np, torch
from cutadapt_amd.adapters import BackAdapter
from cutadapt_amd.batch import ReadBatch
from cutadapt_amd.pipeline import trim_fastq
T = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
n = 12_000_000
b = ReadBatch.synthetic(n, 150, [T], seed=2)
seqs = b.seqs.cpu().numpy().reshape(n, 150)
name = np.frombuffer(b"@read_000000000\n", dtype=np.uint8)
rec_len = len(name) + 150 + 3 + 150 + 1
buf = np.empty((n, rec_len), dtype=np.uint8)
buf[:, :len(name)] = name
idx = np.arange(n)
for d in range(9):
    buf[:, 6 + 8 - d] = 48 + (idx // 10**d) % 10
o = len(name)
buf[:, o:o+150] = seqs; buf[:, o+150] = 10; buf[:, o+151] = ord('+'); buf[:, o+152] = 10
buf[:, o+153:o+303] = ord('I'); buf[:, o+303] = 10
data = buf.tobytes()
del buf
class Null:
    def write(self, b): return len(b)
res = {"reads": n, "fastq_bytes": len(data), "cpus": len(os.sched_getaffinity(0))}
ref = None
for threads, chunk in ((1, 32), (4, 32), (8, 16), (16, 16), (16, 8), (16, 32)):
    ad = BackAdapter(T, max_errors=0.1, min_overlap=3)
    trim_fastq(io.BytesIO(data[: 320 << 20]), Null(), ad, chunk_bytes=chunk << 20, threads=threads)   # warm
    t0 = time.perf_counter()
    if ref is None:
        out = io.BytesIO()
        stats = trim_fastq(io.BytesIO(data), out, ad, chunk_bytes=chunk << 20, threads=threads)
        ref = (stats["with_adapters"], stats["bp_out"])
    else:
        stats = trim_fastq(io.BytesIO(data), Null(), ad, chunk_bytes=chunk << 20, threads=threads)
        assert (stats["with_adapters"], stats["bp_out"]) == ref
    dt = time.perf_counter() - t0
    res[f"threads{threads}_chunk{chunk}MiB"] = {"Mreads_per_s": round(n / dt / 1e6, 2), "GB_per_s_in": round(len(data) / dt / 1e9, 2)}
print(json.dumps(res))
