#!/usr/bin/env python3
"""Markdown table of profiles/r06_watertight_bound.jsonl (README.md)."""
import json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rows = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", "r06_watertight_bound.jsonl")) if l.startswith("{")]
print("| config | rays | hit mask differs (contract only / watertight only) | triangle differs (ties at equal distance / other) | hit count differs | largest relative difference on the same triangle: t | uv | loc |")
print("|---|---|---|---|---|---|---|---|")
shards = [r for r in rows if r["config"].startswith("C5(ii)")]
def line(name, rs):
    n = sum(r["rays"] for r in rs)
    f = lambda k: sum(r[k] for r in rs)
    m = lambda k: max(r[k] for r in rs)
    ns = f"{n:,}".replace(",", " ")
    print(f"| {name} | {ns} | {f('only_contract') + f('only_watertight')} ({f('only_contract')} / {f('only_watertight')}) | {f('tri_diff_same_t') + f('tri_diff_other')} ({f('tri_diff_same_t')} / {f('tri_diff_other')}) | {f('count_diff')} | {m('max_rel_t_diff_same_tri'):.1e} | {m('max_rel_uv_diff'):.1e} | {m('max_rel_loc_diff'):.1e} |")
for r in rows:
    if r["config"].startswith("C5(ii)") or r["config"].startswith("C3 (first"):
        continue
    name = r["config"].split(",")[0].replace("(stand-in (81920 tris; no Stanford bunny file in this image))", "(stand-in mesh)")
    if r["config"].startswith("C3"): name = "C3, all 10 M hash rays"
    line(name + (", " + r["config"].split(", ", 1)[1] if ", " in r["config"] and not r["config"].startswith("C3") else ""), [r])
    if r["config"].startswith("C5(i)"):
        line("C5(ii), ALL 100 M hash rays (eight shards)", shards)
