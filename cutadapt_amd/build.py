"""Build libcutadapt_hip.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m cutadapt_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so sits next to this file so that it
travels with the gpurun snapshot and is what the Python layer loads (never a JIT cache).
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libcutadapt_hip.so")
# the same sources with -DCAH_DEV_KNOBS: test-only switches (CAH_TEST_M2_UNGATED) the product library does not hold;
# tests/test_gpu_multi2.py runs it in a process of its own (CAH_LIB_PATH)
DEV_LIB_PATH = os.path.join(_HERE, "libcutadapt_hip_dev.so")
DEV_SOURCES = ["api.cpp"]          # the sources that read CAH_DEV_KNOBS: the others' product objects are linked as they are
SOURCES = ["api.cpp", "kernels.hip", "stream2.hip", "multi.hip", "multi2.hip", "long.hip", "fastq_gpu.hip", "synth_kernel.hip", "fastq.cpp", "index.hip", "qualtrim.hip"]
HEADERS = ["cah_device.h", "kernels.h", "back_scan.h", "dev_common.h", "filter_common.h", "stream2.h", "multi2.h", "revcomp.h", os.path.join("..", "..", "include", "cutadapt_hip.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def source_hash() -> str:
    """sha256 over everything the library is built from: csrc/*.{hip,h,cpp} + include/cutadapt_hip.h.  Embedded in the
    library at build time (cah_build_id); bench.py ties PMC profiles to it."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp")))
    for f in names + [os.path.join("..", "..", "include", "cutadapt_hip.h")]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read() + b"\0")
    return h.hexdigest()


_ID_MARKER = b"CAH_BUILD_ID="


def library_build_id(path: str = None) -> str:
    """the build id embedded in a built library ("" if it cannot be read: missing file, older build).  Read from the FILE's
    bytes (api.cpp puts the marker ``CAH_BUILD_ID=<sha256>`` in front of what cah_build_id() returns), never by loading the
    library: ctypes does not unload, and the loader hands out the already mapped image for a path it has seen, so a process
    that had looked at a stale library this way kept running it after the rebuild (advisor, round 5)."""
    import re
    try:
        with open(path or LIB_PATH, "rb") as fh:
            m = re.search(_ID_MARKER + rb"([0-9a-f]{64})", fh.read())
        return m.group(1).decode() if m else ""
    except OSError:
        return ""


def needs_build() -> bool:
    """the library is missing or was built from other sources than those on disk (its embedded hash says so: file times
    alone would pass a stale binary that was touched, or one that travelled without its sources)"""
    if not os.path.exists(LIB_PATH):
        return True
    return library_build_id() != source_hash()


def build_library(force: bool = False, verbose: bool = False, extra_flags=None, out_path=None) -> str:
    """extra_flags/out_path: developer knobs for A/B builds of kernel variants (e.g.
    ["-DCAH_SCHED_ROWS=4"]); the product build uses neither."""
    if extra_flags or out_path:
        return _build(extra_flags or [], out_path or LIB_PATH, verbose, tag="_" + str(abs(hash(tuple(extra_flags or []))) % 100000))
    if not force and not needs_build():
        return LIB_PATH
    return _build([], LIB_PATH, verbose, tag="", force=force)


def build_dev_library(force: bool = False, verbose: bool = False) -> str:
    """libcutadapt_hip_dev.so: the product's objects, with DEV_SOURCES compiled again under -DCAH_DEV_KNOBS"""
    build_library(force=force, verbose=verbose)
    if not force and os.path.exists(DEV_LIB_PATH) and library_build_id(DEV_LIB_PATH) == source_hash() \
            and os.path.getmtime(DEV_LIB_PATH) >= os.path.getmtime(LIB_PATH):
        return DEV_LIB_PATH
    return _build(["-DCAH_DEV_KNOBS"], DEV_LIB_PATH, verbose, tag="_dev", only=DEV_SOURCES)


def _build(extra_flags, lib_path, verbose, tag, force=False, only=None) -> str:
    """only: the sources compiled with extra_flags into _obj<tag>; the others' objects are taken from the product build"""
    objs = []
    obj_dir = os.path.join(_HERE, "csrc", "_obj" + tag)
    os.makedirs(obj_dir, exist_ok=True)
    procs = []
    build_id = source_hash()
    newest_header = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))
    for src in SOURCES:
        if only is not None and src not in only:
            objs.append(os.path.join(_HERE, "csrc", "_obj", src + ".o"))
            continue
        obj = os.path.join(obj_dir, src + ".o")
        objs.append(obj)
        src_path = os.path.join(CSRC, src)
        # an object is kept while it is newer than its source and every header (headers are not tracked per source);
        # api.cpp carries the build id -- the hash of ALL sources -- and is compiled whenever that changed
        id_file = obj + ".build_id"
        stale_id = False
        if src == "api.cpp":
            try:
                stale_id = open(id_file).read() != build_id
            except OSError:
                stale_id = True
        if not force and not stale_id and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src_path), newest_header):
            continue
        flags = list(extra_flags)
        if src == "api.cpp":
            flags.append(f'-DCAH_BUILD_ID="{build_id}"')
            with open(id_file, "w") as fh:
                fh.write(build_id)
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip"] + flags + \
              ["-c", src_path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib_path] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return lib_path


if __name__ == "__main__":
    path = build_library(force="--force" in sys.argv, verbose=True)
    print("built", path)
    if "--dev" in sys.argv:
        print("built", build_dev_library(verbose=True))
