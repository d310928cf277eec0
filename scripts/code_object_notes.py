#!/usr/bin/env python3
"""Per-kernel resources (VGPRs, spills, LDS, scratch) of the SHIPPED library, read from the code-object
metadata of every gfx950 image bundled in libtriro_hip.so (no GPU needed).
usage: python scripts/code_object_notes.py [path/to/libtriro_hip.so] [name filter]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so_path, arch="gfx950"):
    """the device images for `arch` inside the clang offload bundles of a host shared object"""
    b = open(so_path, "rb").read()
    i = b.find(MAGIC)
    while i >= 0:
        n, = struct.unpack_from("<Q", b, i + 24)
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", b, p)
            p += 24
            triple = b[p:p + tl].decode()
            p += tl
            if arch in triple and size > 0:
                yield b[i + off:i + off + size]
        i = b.find(MAGIC, i + 1)


def kernels(so_path, arch="gfx950"):
    """[{name (demangled), vgpr, vgpr_spill, sgpr_spill, lds, scratch}] of every kernel"""
    out = []
    for img in code_objects(so_path, arch):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        blocks = re.split(r"\n\s+- \.agpr_count:", txt)[1:]
        names = []
        for blk in blocks:
            nm = re.search(r"\.name:\s+(\S+)", blk).group(1)
            g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))  # noqa: E731
            names.append(nm)
            out.append({"mangled": nm, "vgpr": g("vgpr_count"), "vgpr_spill": g("vgpr_spill_count"),
                        "sgpr_spill": g("sgpr_spill_count"), "lds": g("group_segment_fixed_size"),
                        "scratch": g("private_segment_fixed_size")})
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        for k, d in zip(out[-len(names):], dem):
            k["name"] = d.replace("(anonymous namespace)::", "").split("(")[0]
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    so = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(here, "..", "trimesh-ray-optix_amd", "lib", "libtriro_hip.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in kernels(so):
        if flt in k["name"]:
            alloc = (k["vgpr"] + 7) // 8 * 8
            print(f"{k['name']:78s} vgpr {k['vgpr']:3d} ({min(8, 512 // max(alloc, 8))} waves/SIMD) spill {k['vgpr_spill']} "
                  f"sgpr_spill {k['sgpr_spill']:3d} lds {k['lds']:6d} scratch {k['scratch']}")
