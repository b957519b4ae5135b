#!/usr/bin/env python3
"""Register / spill / scratch figures of the kernels in a built library, read from the BINARY: the gfx950 code objects are
unbundled from the library's .hip_fatbin section (clang offload bundles) and their kernel metadata read with
llvm-readelf --notes.  Usage: kernel_resources.py [library.so] [-o out.json]"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    with tempfile.TemporaryDirectory() as td:
        raw = os.path.join(td, "fatbin.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, raw], check=True)
        data = open(raw, "rb").read()
    pos = 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", data, pos + 24)[0]
        p = pos + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                yield data[pos + off:pos + off + size]
        pos += 24


def kernels(lib):
    out = {}
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
            g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
            out[dem.split("(")[0].replace("void ", "")] = {
                "vgpr_count": g("vgpr_count"), "sgpr_count": g("sgpr_count"), "vgpr_spill_count": g("vgpr_spill_count"),
                "sgpr_spill_count": g("sgpr_spill_count"), "scratch_bytes_per_lane": g("private_segment_fixed_size"),
                "lds_static_bytes": g("group_segment_fixed_size"), "max_flat_workgroup_size": g("max_flat_workgroup_size")}
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    args = [a for a in sys.argv[1:] if a != "-o"]
    lib = args[0] if args and args[0].endswith(".so") else os.path.join(here, "cutadapt_amd", "libcutadapt_hip.so")
    res = kernels(lib)
    if "-o" in sys.argv:
        dst = sys.argv[sys.argv.index("-o") + 1]
        with open(dst, "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
    for k in sorted(res):
        if re.match(r"k_(filter_stream2<2, 4, true, false, false, false>|back_scan<false, 2>|dp_packed<36, false>|multi_stream<true>|multi_scan<2>)", k):
            print(k, res[k])
    print(len(res), "kernels")
