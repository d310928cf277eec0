#!/usr/bin/env python3
"""Print the loop nest (LLVM loop-depth comments) of every kernel in a gfx950 .s file.
usage: scripts/loopnest.py file.s [filter]"""
import re
import sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\.Lfunc_end", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt and flt not in name:
        continue
    loops = re.findall(r"^(\.LBB\d+_\d+):\s*; (?:=>\s*)?(?:This |  Parent Loop).*?$|; =>\s*This (?:Inner )?Loop Header: Depth=(\d+)", body, re.M)
    depths = re.findall(r"This (Inner )?Loop Header: Depth=(\d+)", body)
    short = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)[:60]
    print(short, "loops:", " ".join(("I" if i else "L") + d for i, d in depths))
