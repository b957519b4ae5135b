"""Aligner.enable_debug(): the DP matrices the reference collects in DPMatrix objects (_align.pyx:58-92, :279-296,
:385-390, :485-489) come out of the HIP path cell for cell -- compared with the compiled reference (oracle/_ref)
where it is built, else with properties every matrix must have."""
import random

import pytest

pytestmark = pytest.mark.gpu


def rs(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


@pytest.fixture(scope="module")
def hip():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from cutadapt_amd import _lib
    _lib.lib()
    return _lib


def test_debug_matrices_equal_the_reference(hip):
    from cutadapt_amd.align import Aligner
    from oracle import build_ref, ref_loader
    ref = ref_loader.load() if build_ref.is_built() else None
    rng = random.Random(11)
    cases = 0
    for it in range(60):
        m = rng.choice([1, 3, 8, 13, 20, 33, 64, 70])
        n = rng.choice([0, 1, 5, 20, 40, 90])
        adapter = rs(rng, m, "ACGT" if it % 3 else "ACGTN")
        read = list(rs(rng, n))
        if n > 4 and rng.random() < 0.7:
            p = rng.randrange(n)
            piece = list(adapter[:rng.randint(1, m)])
            if piece and rng.random() < 0.5:
                piece[rng.randrange(len(piece))] = rng.choice("ACGT")
            read[p:p + len(piece)] = piece
            read = read[:n]
        read = "".join(read)
        kw = dict(max_error_rate=rng.choice([0.0, 0.1, 0.2, 0.3]), flags=rng.randint(0, 15),
                  wildcard_ref=(it % 3 == 0), wildcard_query=rng.random() < 0.3,
                  indel_cost=rng.choice([1, 1, 2, 100000]), min_overlap=rng.randint(1, 4))
        try:
            a = Aligner(adapter, **kw)
        except ValueError:
            continue
        plain = a.locate(read)
        assert a.dpmatrix is None
        a.enable_debug()
        got = a.locate(read)
        assert got == plain, (adapter, read, kw)
        cost, score = a.dpmatrix, a.scorematrix
        assert len(cost._rows) == m + 1 and all(len(r) == n + 1 for r in cost._rows)
        text = str(cost)
        assert text.count("\n") == m + 1
        if ref is not None:
            ra = ref.align.Aligner(adapter, **kw)
            ra.enable_debug()
            assert ra.locate(read) == got
            assert str(ra.dpmatrix) == text, (adapter, read, kw)
            assert str(ra.scorematrix) == str(score), (adapter, read, kw)
        cases += 1
    assert cases >= 40


def test_adapter_debug_prints_matrices(hip, capsys):
    from cutadapt_amd.adapters import BackAdapter
    ad = BackAdapter("AGATCGGAAGAGC", max_errors=0.1, min_overlap=3)
    read = "ACGTTTGACCAAGATCGGAAGAGCTT"
    want = ad.match_to(read)
    ad.enable_debug()
    got = ad.match_to(read)
    out = capsys.readouterr().out
    assert "Edit distances:" in out and "Scores:" in out
    assert (got.rstart, got.rstop, got.errors) == (want.rstart, want.rstop, want.errors)
