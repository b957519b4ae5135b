import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
