"""The C ABI from plain C (examples/c_abi_demo.c): compiles and links against the in-tree library with
gcc alone (CPU check), and reproduces the expected matches on the GPU box -- no Python, no torch in the
process."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    from cutadapt_amd import _lib
    _lib.lib()                                   # makes sure the library is built
    exe = os.path.join(str(tmp_path), "c_abi_demo")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "examples", "c_abi_demo.c"),
           "-I", os.path.join(ROOT, "include"), "-L", os.path.join(ROOT, "cutadapt_amd"), "-lcutadapt_hip",
           "-Wl,-rpath," + os.path.join(ROOT, "cutadapt_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def test_c_demo_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
    assert os.path.exists(exe)
    if not os.path.exists("/dev/kfd"):
        # without a GPU the demo must refuse loudly, not fall back to anything
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_c_demo_runs(tmp_path, hip):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "read 0: adapter[0:33] matches read[15:48], score 33, 0 error(s)" in r.stdout
    assert "read 1: no adapter" in r.stdout
    assert "read 2: adapter[0:29] matches read[26:55], score 27, 1 error(s)" in r.stdout
    assert "read 3: adapter[0:5] matches read[50:55], score 5, 0 error(s)" in r.stdout
