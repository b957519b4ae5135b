"""BASELINE.json configs C3, C4, C5 as parity cases at reduced size:
the HIP path through the adapter classes vs the oracle applying the reference's rules
(LinkedAdapter.match_to, reference adapters.py:1215-1227; MultipleAdapters.match_to,
:1265-1286).  GPU only."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TRUSEQ_R1 = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
TRUSEQ_R2 = "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"


def rs(rng, n, al="ACGT"):
    return "".join(rng.choice(al) for _ in range(n))


def mutate(rng, s, n_edits, al="ACGT"):
    s = list(s)
    for _ in range(n_edits):
        if not s:
            break
        x = rng.randrange(len(s))
        op = rng.randint(0, 2)
        if op == 0:
            s[x] = rng.choice(al)
        elif op == 1:
            s.insert(x, rng.choice(al))
        else:
            del s[x]
    return "".join(s)


def oracle_single(orc, adapter, reads):
    """kmers_present -> locate of one cutadapt_amd adapter object, computed by the oracle.
    Returns (int[n,6] in match coordinates, found[n])."""
    spec = adapter.matcher_spec()
    seqs, offsets = orc.pack_reads(reads)
    if spec.kind == 0:
        oa = orc.Aligner(spec.sequence, spec.max_error_rate, spec.flags, spec.wildcard_ref,
                         spec.wildcard_query, spec.indel_cost, spec.min_overlap)
        of = orc.KmerFinder(spec.kmer_sets, spec.kmer_ref_wildcards, spec.kmer_query_wildcards) \
            if spec.kmer_sets is not None else None
        o6, st = orc.match_batch(oa, of, seqs, offsets)
    else:
        cls = orc.PrefixComparer if spec.kind == 1 else orc.SuffixComparer
        oc = cls(spec.sequence, spec.max_error_rate, spec.wildcard_ref, spec.wildcard_query, spec.min_overlap)
        o6, st = oc.locate_batch(seqs, offsets)
    assert not (st == 2).any()
    return o6.astype(np.int64), st == 1


def oracle_multiple(orc, adapters, reads):
    n = len(reads)
    coords = np.zeros((n, 6), dtype=np.int64)
    found = np.zeros(n, dtype=bool)
    best = np.zeros(n, dtype=np.int64)
    for idx, ad in enumerate(adapters):
        c, f = oracle_single(orc, ad, reads)
        better = f & (~found | (c[:, 4] > coords[:, 4]) | ((c[:, 4] == coords[:, 4]) & (c[:, 5] < coords[:, 5])))
        coords[better] = c[better]
        best[better] = idx
        found |= better
    return coords, found, best


@pytest.mark.parametrize("read_wildcards", [False, True])
def test_c3_linked_anchored_iupac(hip, orc, read_wildcards):
    """C3: linked adapter, anchored 5' with IUPAC N wildcards + 3' TruSeq, e=0.1"""
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch
    rng = random.Random(303 + read_wildcards)
    front_seq = "NNNNNNNNACGTACGT"
    front = A.PrefixAdapter(front_seq, max_errors=0.1, read_wildcards=read_wildcards)
    back = A.BackAdapter(TRUSEQ_R1, max_errors=0.1, min_overlap=3, read_wildcards=read_wildcards)
    assert front.adapter_wildcards and type(front.aligner).__name__ == "Aligner"
    linked = A.LinkedAdapter(front, back, front_required=True, back_required=False, name="c3")
    reads = []
    for _ in range(30000):
        r = ""
        if rng.random() < 0.8:
            r += mutate(rng, rs(rng, 8) + "ACGTACGT", rng.choice([0, 0, 0, 1, 2]))
        r += rs(rng, rng.randint(0, 110))
        if rng.random() < 0.5:
            r += mutate(rng, TRUSEQ_R1, rng.choice([0, 0, 1, 2, 3]))[:rng.randint(3, 40)]
        r += rs(rng, rng.randint(0, 10))
        if rng.random() < 0.3:
            r = "".join(c if rng.random() > 0.02 else "N" for c in r)
        reads.append(r[:150])
    lb = linked.match_to_batch(ReadBatch.from_strings(reads))
    # oracle: the reference's two-stage rule
    fc, ff = oracle_single(orc, front, reads)
    sliced = [r[int(fc[i, 3]):] if ff[i] else r for i, r in enumerate(reads)]
    bc, bf = oracle_single(orc, back, sliced)
    ok = ff.copy()                       # front required; back optional once the front matched
    assert np.array_equal(lb.found, ok)
    assert np.array_equal(lb.front.found, ff & ok) and np.array_equal(lb.back.found, bf & ok)
    assert np.array_equal(lb.front.coords[ok], fc[ok])
    both = ok & bf
    assert np.array_equal(lb.back.coords[both], bc[both])
    assert ok.sum() > 12000 and both.sum() > 3000
    m = lb.match(int(np.nonzero(both)[0][0]))
    assert m.front_match.rstart == 0 and m.back_match is not None


def test_c4_96_adapters(hip, orc):
    """C4: 96 distinct 33-mers as 3' adapters (-a file:), k-mer heuristic on"""
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch
    rng = random.Random(404)
    seqs = [rs(rng, 33) for _ in range(96)]
    ads = [A.BackAdapter(s, max_errors=0.1, min_overlap=3) for s in seqs]
    multi = A.MultipleAdapters(ads)
    n = 40000
    batch = ReadBatch.synthetic(n, 150, seqs, seed=4)
    bm = multi.match_to_batch(batch)
    torch.cuda.synchronize()
    sq, offs = orc.synth_reads(4, 0, n, 150, seqs)
    assert np.array_equal(batch.seqs.cpu().numpy(), sq)
    reads = [bytes(sq[i * 150:(i + 1) * 150]).decode() for i in range(n)]
    coords, found, best = oracle_multiple(orc, ads, reads)
    assert np.array_equal(bm.found, found)
    assert np.array_equal(bm.coords[found], coords[found])
    assert np.array_equal(bm.adapter_index[found], best[found])
    assert found.sum() > 0.3 * n and len(set(best[found].tolist())) == 96


def test_c5_paired_end(hip, orc):
    """C5: paired-end 2 x 150 bp, two adapters per mate: the mates are matched independently
    (reference PairedEndModifierWrapper, modifiers.py:74-79)"""
    import torch
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch
    rng = random.Random(505)
    second1, second2 = rs(rng, 33), rs(rng, 33)
    n = 40000
    for seed, seqs in ((51, [TRUSEQ_R1, second1]), (52, [TRUSEQ_R2, second2])):
        ads = [A.BackAdapter(s, max_errors=0.1, min_overlap=3) for s in seqs]
        multi = A.MultipleAdapters(ads)
        batch = ReadBatch.synthetic(n, 150, seqs, seed=seed)
        bm = multi.match_to_batch(batch)
        torch.cuda.synchronize()
        sq, _ = orc.synth_reads(seed, 0, n, 150, seqs)
        reads = [bytes(sq[i * 150:(i + 1) * 150]).decode() for i in range(n)]
        coords, found, best = oracle_multiple(orc, ads, reads)
        assert np.array_equal(bm.found, found)
        assert np.array_equal(bm.coords[found], coords[found])
        assert np.array_equal(bm.adapter_index[found], best[found])
        assert 0.2 * n < found.sum() < 0.4 * n


def test_paired_adapter_cutter_best_pair(hip, orc):
    """--pair-adapters as a batch op (reference PairedAdapterCutter._find_best_match_pair,
    modifiers.py:480-503): adapter i must be in R1 AND adapter i in R2; highest total score, then fewest total
    errors, then the first pair.  Oracle: the rule restated over per-adapter oracle results."""
    from cutadapt_amd import adapters as A
    from cutadapt_amd.batch import ReadBatch
    from cutadapt_amd.pipeline import BatchPairedAdapterCutter
    rng = random.Random(909)
    n_pairs = 5
    s1 = [rs(rng, 20) for _ in range(n_pairs)]
    s2 = [rs(rng, 20) for _ in range(n_pairs)]
    s1[3] = s1[1]                      # equal adapters: ties between pairs -> the first pair wins
    s2[3] = s2[1]
    ads1 = [A.BackAdapter(s, max_errors=0.1, min_overlap=3) for s in s1]
    ads2 = [A.FrontAdapter(s, max_errors=0.1, min_overlap=3) if i == 2 else A.BackAdapter(s, max_errors=0.1, min_overlap=3)
            for i, s in enumerate(s2)]
    reads1, reads2 = [], []
    for _ in range(6000):
        i, j = rng.randrange(n_pairs), rng.randrange(n_pairs)
        if rng.random() < 0.6:
            j = i
        r1 = rs(rng, rng.randint(20, 100)) + (mutate(rng, s1[i], rng.choice([0, 0, 1, 2]))[:rng.randint(3, 20)] if rng.random() < 0.8 else "")
        if isinstance(ads2[j], A.FrontAdapter):
            r2 = (mutate(rng, s2[j], rng.choice([0, 1]))[-rng.randint(3, 20):] if rng.random() < 0.8 else "") + rs(rng, rng.randint(20, 100))
        else:
            r2 = rs(rng, rng.randint(20, 100)) + (mutate(rng, s2[j], rng.choice([0, 0, 1, 2]))[:rng.randint(3, 20)] if rng.random() < 0.8 else "")
        reads1.append(r1 + rs(rng, rng.randint(0, 5)))
        reads2.append(r2)
    pc = BatchPairedAdapterCutter(ads1, ads2)
    found, idx, c1, c2 = pc.best_pairs(ReadBatch.from_strings(reads1), ReadBatch.from_strings(reads2))
    n = len(reads1)
    want_found = np.zeros(n, dtype=bool)
    want_idx = np.zeros(n, dtype=np.int64)
    w1, w2 = np.zeros((n, 6), dtype=np.int64), np.zeros((n, 6), dtype=np.int64)
    best_score = np.zeros(n, dtype=np.int64)
    best_err = np.zeros(n, dtype=np.int64)
    for i in range(n_pairs):
        a6, af = oracle_single(orc, ads1[i], reads1)
        b6, bf = oracle_single(orc, ads2[i], reads2)
        both = af & bf
        score, err = a6[:, 4] + b6[:, 4], a6[:, 5] + b6[:, 5]
        better = both & (~want_found | (score > best_score) | ((score == best_score) & (err < best_err)))
        want_found |= better
        best_score[better], best_err[better], want_idx[better] = score[better], err[better], i
        w1[better], w2[better] = a6[better], b6[better]
    assert np.array_equal(found, want_found)
    assert np.array_equal(idx[found], want_idx[found])
    assert np.array_equal(c1[found], w1[found]) and np.array_equal(c2[found], w2[found])
    assert found.sum() > 800 and len(set(want_idx[found].tolist())) >= 4 and 3 not in set(want_idx[found].tolist())


@pytest.mark.parametrize("config", ["C3", "C4", "C5"])
def test_baseline_workloads_500k_reads_against_the_oracle(hip, orc, config):
    """The BASELINE.json workloads themselves (cutadapt_amd/workloads.py: the generator, adapter sets and seeds
    bench.py times), 500 000 reads (per mate) through the same step bench.py runs, every tuple bit-compared with the
    oracle applying the reference's rules (bench.py Workload.parity)."""
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n = 500_000
    wl = bench.Workload(config, n, 0, torch.device("cuda", 0), None)
    wl.step()
    torch.cuda.synchronize()
    ok, what = wl.parity(n)
    assert ok, what
    assert f"{n} reads" in what
