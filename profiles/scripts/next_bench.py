"""Throughput of the 'next'-row kernels on resident data; prints one JSON object."""
import json, random, sys
sys.path.insert(0, '.')
import numpy as np, torch
from cutadapt_amd import adapters as A, _lib, qualtrim as qt
from cutadapt_amd.batch import ReadBatch
n, L = 50_000_000, 150
dev = torch.device("cuda")
batch = ReadBatch.synthetic(n, L, ["AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"], seed=7, p_adapter=0.25)
g = torch.Generator(device="cuda"); g.manual_seed(1)
quals = torch.randint(33 + 2, 33 + 41, (n * L,), dtype=torch.uint8, device=dev, generator=g)
# a quality drop at the 3' end of 30% of the reads
q2 = quals.view(n, L)
drop = torch.rand(n, device=dev, generator=g) < 0.3
q2[:, L - 12:] = torch.where(drop[:, None], torch.full((1, 12), 33 + 3, dtype=torch.uint8, device=dev), q2[:, L - 12:])
# poly-A tails on 20% of the reads
tail = torch.rand(n, device=dev, generator=g) < 0.2
s2 = batch.seqs.view(n, L)
s2[:, L - 25:] = torch.where(tail[:, None], torch.full((1, 25), ord("A"), dtype=torch.uint8, device=dev), s2[:, L - 25:])
off = batch.offsets
L_ = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
ss = torch.zeros((n, 2), dtype=torch.int32, device=dev); i32 = torch.zeros(n, dtype=torch.int32, device=dev)
ee = torch.zeros(n, dtype=torch.float64, device=dev); stat = torch.zeros(n, dtype=torch.uint8, device=dev)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
res = {"reads": n, "read_len": L}
def rec(name, ms, bytes_per_read):
    res[name] = {"ms": round(ms, 3), "Greads_per_s": round(n / ms / 1e6, 2), "algorithmic_bytes_per_read": bytes_per_read,
                 "algorithmic_GBps": round(n * bytes_per_read / ms / 1e6, 1), "hbm_frac": round(n * bytes_per_read / ms / 1e6 / 8000, 4)}
rec("quality_trim(-q 20)", timed(lambda: _lib.check(L_.cah_quality_trim_batch(quals.data_ptr(), off.data_ptr(), None, n, 0, 20, 33, ss.data_ptr(), st))), 8 + 8 + 2)
res["quality_trim(-q 20)"]["trimmed_fraction"] = float(((ss[:, 1] - ss[:, 0]) < L).float().mean())
rec("nextseq_trim(20)", timed(lambda: _lib.check(L_.cah_nextseq_trim_batch(batch.seqs.data_ptr(), quals.data_ptr(), off.data_ptr(), None, n, 20, 33, i32.data_ptr(), st))), 8 + 4 + 4)
rec("poly_a_trim", timed(lambda: _lib.check(L_.cah_poly_a_trim_batch(batch.seqs.data_ptr(), off.data_ptr(), None, n, 0, i32.data_ptr(), st))), L + 8 + 4)
res["poly_a_trim"]["trimmed_fraction"] = float((i32 < L).float().mean())
rec("expected_errors", timed(lambda: _lib.check(L_.cah_expected_errors_batch(quals.data_ptr(), off.data_ptr(), None, n, 33, ee.data_ptr(), stat.data_ptr(), st))), L + 8 + 9)
# adapter index: 96 barcodes of 10 bp, e = 0.1 with indels, planted at the start of 90% of the reads
rng = random.Random(3)
barcodes = ["".join(rng.choice("ACGT") for _ in range(10)) for _ in range(96)]
ix = A.AdapterIndex([A.PrefixAdapter(b, max_errors=0.1, indels=True) for b in barcodes], prefix=True)
codes = torch.tensor([[ord(c) for c in b] for b in barcodes], dtype=torch.uint8, device=dev)
which = torch.randint(0, 96, (n,), device=dev, generator=g)
plant = torch.rand(n, device=dev, generator=g) < 0.9
s2[:, :10] = torch.where(plant[:, None], codes[which], s2[:, :10])
out6 = torch.zeros((n, 6), dtype=torch.int32, device=dev)
rec("index_lookup(96 barcodes)", timed(lambda: _lib.check(L_.cah_index_lookup_batch(ix._h.handle, batch.seqs.data_ptr(), off.data_ptr(), None, n, out6.data_ptr(), i32.data_ptr(), stat.data_ptr(), st))), 11 + 8 + 24 + 4 + 1)
res["index_lookup(96 barcodes)"]["matched_fraction"] = float((stat == 1).float().mean())
res["index_lookup(96 barcodes)"]["index_strings"] = len(ix)
print(json.dumps(res))
