#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in raytracer_amd/lib/librt_hip.so (from the code object's metadata):
  tools/kernel_resources.py [substring ...]   -> one line per kernel whose demangled name holds every substring"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(so):
    data = open(so, "rb").read()
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    n = struct.unpack_from("<Q", data, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + tl].decode()
        off += tl
        if "gfx950" in triple:
            return data[i + o:i + o + sz]
    raise SystemExit("no gfx950 code object in " + so)


def main():
    so = os.path.join(ROOT, "raytracer_amd", "lib", "librt_hip.so")
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code_object(so))
        f.flush()
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    kernels = re.split(r"\n\s+- \.agpr_count:", notes)[1:]
    rows = []
    for k in kernels:
        get = lambda key: re.search(r"\." + key + r":\s+(\S+)", k)
        name = subprocess.run(["c++filt", get("name").group(1)], capture_output=True, text=True).stdout.strip()
        rows.append((name, int(get("vgpr_count").group(1)), int(get("sgpr_count").group(1)), int(get("private_segment_fixed_size").group(1)),
                     int(get("group_segment_fixed_size").group(1)), int(get("vgpr_spill_count").group(1))))
    for name, v, s, scratch, lds, spill in sorted(rows):
        short = name.split("(")[0]
        if all(a in short for a in sys.argv[1:]):
            print(f"{short:70s} vgpr {v:3d} sgpr {s:3d} scratch {scratch:4d} B  lds {lds:6d} B  spilled vgprs {spill}")


if __name__ == "__main__":
    main()
