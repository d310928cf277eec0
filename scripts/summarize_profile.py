#!/usr/bin/env python3
"""Summarise a gpurun_out/prof_<tag>/ directory (scripts/profile_bench.sh) into
profiles/<tag>_summary.md + .json: per-kernel stats from --kernel-trace --stats and per-launch
PMC averages for the query kernel.  FETCH_SIZE/WRITE_SIZE are in KiB (rocprofv3); on gfx950
FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM) -- both the raw
and the x2 figure are recorded."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1]
kernel_key = sys.argv[2] if len(sys.argv) > 2 else "k_query"
root = os.path.join("gpurun_out", f"prof_{tag}")
out = {"tag": tag, "kernel_filter": kernel_key}


def short_name(name):
    n = name.replace("(anonymous namespace)::", "")
    if n.startswith("void "):
        n = n[5:]
    depth, out = 0, []
    for ch in n:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out)[:90]


stats = glob.glob(os.path.join(root, "trace", "*", "*_kernel_stats.csv"))
if stats:
    out["kernel_stats"] = []
    for r in csv.DictReader(open(stats[0])):
        out["kernel_stats"].append({"name": short_name(r["Name"]), "calls": int(r["Calls"]),
                                    "avg_ns": float(r["AverageNs"]), "min_ns": int(r["MinNs"]),
                                    "max_ns": int(r["MaxNs"]), "pct": float(r["Percentage"])})
# the timed region of bench.py = the LAST `steps` dispatches of the dominant kernel (the first
# call runs without a learned launch order, then `warmup` untimed calls)
trace = glob.glob(os.path.join(root, "trace", "*", "*_kernel_trace.csv"))
if trace:
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(trace[0]))
            if kernel_key in r["Kernel_Name"]]
    steps = int(os.environ.get("PROFILE_STEPS", "20"))
    # bench.py: dispatch 0 = first call, then `warmup_steps_done` warm-up calls, then `steps` timed
    # calls, then (default flags) the cold / moving-camera / instrumented companions.  The count of
    # warm-up calls is in the JSON line bench.py printed (trace.log); without it fall back to "the
    # last `steps` dispatches" (run_query.py, bench.py --no-companions).
    skip = None
    tlog = os.path.join(root, "trace.log")
    if os.path.exists(tlog):
        for ln in open(tlog, errors="replace"):
            if ln.startswith('{"metric"'):
                try:
                    j = json.loads(ln)
                    skip = 1 + int(j["config"]["warmup_steps_done"])
                    steps = int(j["steps"])
                except Exception:
                    pass
    if skip is not None and len(durs) >= skip + steps:
        win = durs[skip:skip + steps]
    elif len(durs) >= steps:
        win = durs[-steps:]
    else:
        win = []
    if win:
        out["timed_region"] = {"kernel": kernel_key, "steps": steps, "avg_us": sum(win) / len(win) / 1e3,
                               "min_us": min(win) / 1e3, "max_us": max(win) / 1e3,
                               "first_call_us": durs[0] / 1e3}
pmc = defaultdict(list)
meta = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for fcsv in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(fcsv)):
            if kernel_key in r["Kernel_Name"]:
                pmc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {"grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"]), "vgpr": int(r["VGPR_Count"]),
                        "sgpr": int(r["SGPR_Count"]), "lds": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"])}
out["launch"] = meta
out["pmc_avg_per_launch"] = {k: sum(v) / len(v) for k, v in pmc.items()}
p = out["pmc_avg_per_launch"]
d = {}
if "FETCH_SIZE" in p:
    d["fetch_bytes_raw"] = p["FETCH_SIZE"] * 1024
    d["fetch_bytes_x2_gfx950"] = p["FETCH_SIZE"] * 2048
if "WRITE_SIZE" in p:
    d["write_bytes"] = p["WRITE_SIZE"] * 1024
if "TCC_HIT_sum" in p and "TCC_MISS_sum" in p:
    d["l2_hit_rate"] = p["TCC_HIT_sum"] / max(1.0, p["TCC_HIT_sum"] + p["TCC_MISS_sum"])
if "SQ_INSTS_VALU" in p and "SQ_WAVES" in p:
    d["valu_insts_per_wave"] = p["SQ_INSTS_VALU"] / max(1.0, p["SQ_WAVES"])
if "SQ_THREAD_CYCLES_VALU" in p and "SQ_ACTIVE_INST_VALU" in p:
    d["valu_thread_cycles_per_active_inst_cycle"] = p["SQ_THREAD_CYCLES_VALU"] / max(1.0, p["SQ_ACTIVE_INST_VALU"])
if "SQ_WAIT_ANY" in p and "SQ_WAVE_CYCLES" in p:
    d["wait_any_frac"] = p["SQ_WAIT_ANY"] / max(1.0, p["SQ_WAVE_CYCLES"])
    d["active_inst_frac"] = p.get("SQ_ACTIVE_INST_ANY", 0) / max(1.0, p["SQ_WAVE_CYCLES"])
    d["wait_inst_frac"] = p.get("SQ_WAIT_INST_ANY", 0) / max(1.0, p["SQ_WAVE_CYCLES"])
out["derived"] = d
os.makedirs("profiles", exist_ok=True)
json.dump(out, open(os.path.join("profiles", f"{tag}_summary.json"), "w"), indent=1)
with open(os.path.join("profiles", f"{tag}_summary.md"), "w") as f:
    cmd = os.environ.get("PROFILE_CMD", "bench.py --steps 20 --warmup 3 --no-cpu-baseline")
    f.write(f"# rocprofv3 summary `{tag}` ({cmd}, MI355X)\n\n")
    f.write("## kernel-trace --stats\n\n| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
    for k in out.get("kernel_stats", []):
        f.write(f"| `{k['name']}` | {k['calls']} | {k['avg_ns']/1e3:.1f} | {k['min_ns']/1e3:.1f} | {k['max_ns']/1e3:.1f} | {k['pct']:.2f} |\n")
    if "timed_region" in out:
        t = out["timed_region"]
        f.write(f"\nTimed region ({t['steps']} timed dispatches of `{t['kernel']}`, i.e. without the first call "
                f"({t['first_call_us']:.1f} us, no learned launch order yet), the warm-up and bench.py's companions): "
                f"avg {t['avg_us']:.1f} us, min {t['min_us']:.1f}, max {t['max_us']:.1f}.\n")
    f.write(f"\n## PMC (separate passes), kernel filter `{kernel_key}`, averages per launch\n\nlaunch: {meta}\n\n| counter | value |\n|---|---|\n")
    for k, v in sorted(p.items()):
        f.write(f"| {k} | {v:.6g} |\n")
    f.write("\n## derived\n\n")
    for k, v in d.items():
        f.write(f"- {k}: {v:.6g}\n")
print(json.dumps(out["launch"]))
print(json.dumps(out["pmc_avg_per_launch"], indent=0))
print(json.dumps(d, indent=0))
