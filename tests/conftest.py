import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # CAH_TEST_SEED_OFFSET=<int>: soak runs -- every seeded generator of the tests (random.Random(seed),
    # numpy.random.default_rng(seed)) gets its seed shifted, so the same tests draw other adapters and reads.
    # (Tests that also assert how MANY cases of a kind they drew may then fail on that count; a parity failure
    # says "differ".)
    off = int(os.environ.get("CAH_TEST_SEED_OFFSET", "0") or 0)
    if off:
        import random
        import numpy as np
        base_random, base_rng = random.Random, np.random.default_rng

        class ShiftedRandom(base_random):
            def __init__(self, seed=None):
                super().__init__(seed + off if isinstance(seed, int) else seed)

        random.Random = ShiftedRandom
        np.random.default_rng = lambda seed=None, *a, **k: base_rng(seed + off if isinstance(seed, int) else seed, *a, **k)


@pytest.fixture(scope="session")
def golden():
    import json

    def load(name):
        with open(os.path.join(ROOT, "tests", "golden", name)) as f:
            return json.load(f)
    return load


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (C restatement), built on demand"""
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def ref():
    """the compiled reference (oracle/_ref) or None when it is not available"""
    from oracle import ref_loader
    return ref_loader.load()


@pytest.fixture(scope="session")
def hip():
    """the HIP library; GPU tests must FAIL (not skip) when it cannot be used"""
    from cutadapt_amd import _lib
    L = _lib.lib()
    assert _lib.device_count() >= 1, "no HIP device visible: GPU tests need a real MI355X"
    info = _lib.device_info(0)
    assert info["arch"].startswith("gfx950"), info
    return L
