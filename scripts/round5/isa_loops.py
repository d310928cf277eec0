#!/usr/bin/env python3
"""Static look at a kernel of the shipped library: instruction mix of its loops (no GPU needed).
usage: python scripts/round5/isa_loops.py <substring of the demangled kernel name> [path/to/lib.so]
Prints, for every backward branch (a loop), the span it closes and the VALU / SALU / VMEM / LDS counts inside."""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import code_object_notes as con

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(so, want):
    for img in con.code_objects(so):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img); f.flush()
            ks = [k for k in con.kernels_of_image(f.name)] if hasattr(con, "kernels_of_image") else None
            txt = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True).stdout
        cur, out = None, {}
        for line in txt.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1); out[cur] = []; continue
            if cur and line.startswith("\t"):
                out[cur].append(line)
        names = list(out)
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        for n, d in zip(names, dem):
            d = d.replace("(anonymous namespace)::", "").split("(")[0]
            if want in d and not n.endswith(".kd"):
                yield d, out[n]


def classify(op):
    if op.startswith("v_"): return "valu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"


def instructions(lines):
    """[(address, text)] of a kernel's disassembly lines"""
    ins = []
    for l in lines:
        body = l.split("//")[0].strip()
        m = re.search(r"//\s*([0-9A-F]+):", l)
        if body:
            ins.append((int(m.group(1), 16) if m else None, body))
    return ins


def loops(ins):
    """[(first, last)] instruction index spans closed by a backward branch"""
    addr_idx = {a: i for i, (a, _) in enumerate(ins)}
    out = []
    for i, (a, b) in enumerate(ins):
        m = re.match(r"(s_cbranch_\w+|s_branch)\s+(\d+)", b)
        if not m or int(m.group(2)) < 32768:
            continue
        j = addr_idx.get(a + 4 + (int(m.group(2)) - 65536) * 4)
        if j is not None:
            out.append((j, i))
    return out


def scratch_in_trip_loops(so, want, marker="v_perm_b32", smallest_only=False, near_call=0):
    """scratch instructions inside the innermost loops that hold a `marker` instruction (the grid nodes' box test = the
    traversal trips), per kernel whose demangled name contains `want`: {name: [instruction text]}.
    near_call = N: scratch instructions within N instructions of a call (s_swappc_b64) do not count -- round 6: the
    registers the float64 part of the predicate clobbers are saved around that (rare, cold) call."""
    res = {}
    for name, lines in disassemble(so, want):
        ins = instructions(lines)
        lp = loops(ins)
        spans = set()
        for p, (_, b) in enumerate(ins):
            if not b.startswith(marker):
                continue
            inner = [sp for sp in lp if sp[0] <= p <= sp[1]]
            if inner:
                spans.add(min(inner, key=lambda sp: sp[1] - sp[0]))
        if smallest_only and spans:       # (a marker may also sit outside the trips, e.g. the root visit of a refill)
            spans = {min(spans, key=lambda sp: sp[1] - sp[0])}
        calls = [i for i, (_, t) in enumerate(ins) if t.startswith("s_swappc_b64")]
        res[name] = [ins[i][1] for (a, z) in spans for i in range(a, z + 1) if ins[i][1].startswith("scratch_")
                     and not (near_call and any(abs(i - c) <= near_call for c in calls))]
    return res


def main():
    want = sys.argv[1]
    here = os.path.dirname(os.path.abspath(__file__))
    so = sys.argv[2] if len(sys.argv) > 2 else os.path.join(here, "..", "..", "trimesh-ray-optix_amd", "lib", "libtriro_hip.so")
    for name, lines in disassemble(so, want):
        ins = []
        for l in lines:
            body = l.split("//")[0].strip()
            m = re.search(r"//\s*([0-9A-F]+):", l)
            addr = int(m.group(1), 16) if m else None
            if body: ins.append((addr, body))
        addr_idx = {a: i for i, (a, _) in enumerate(ins)}
        print(f"== {name}: {len(ins)} instructions, {sum(1 for _, b in ins if b.startswith('v_'))} VALU")
        for i, (a, b) in enumerate(ins):
            m = re.match(r"(s_cbranch_\w+|s_branch)\s+(\d+)", b)
            if not m: continue
            off = int(m.group(2))
            if off < 32768: continue
            tgt = a + 4 + (off - 65536) * 4
            j = addr_idx.get(tgt)
            if j is None: continue
            mix = {}
            for _, bb in ins[j:i + 1]:
                c = classify(bb.split()[0]); mix[c] = mix.get(c, 0) + 1
            pk = sum(1 for _, bb in ins[j:i + 1] if bb.startswith("v_pk_"))
            print(f"  loop [{j}..{i}] {i - j + 1:5d} instr  " + " ".join(f"{k}={v}" for k, v in sorted(mix.items())) + f"  (v_pk={pk})")


if __name__ == "__main__":
    main()
