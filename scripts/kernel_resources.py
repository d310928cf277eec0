#!/usr/bin/env python3
"""VGPR / LDS / scratch of every kernel in a device assembly file (hipcc -S --cuda-device-only).
usage: python scripts/kernel_resources.py /tmp/traverse.s [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: int(re.search(r"\.amdhsa_" + k + r"\s+(\d+)", body).group(1))  # noqa: E731
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("(anonymous namespace)::", "").split("(")[0]
    if flt in dem:
        v = g("next_free_vgpr")
        alloc = (v + 7) // 8 * 8
        print(f"{dem:70s} vgpr {v:3d} (alloc {alloc:3d} -> {min(8, 512 // alloc)} waves/SIMD)  lds {g('group_segment_fixed_size'):6d}  scratch {g('private_segment_fixed_size')}")
