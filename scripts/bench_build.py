#!/usr/bin/env python3
"""Builder timings on the GPU box: first build, rebuild (tr_bvh_update) and refit of the headline
mesh and the bunny stand-in.  usage: python scripts/bench_build.py [--reps 10]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "trimesh-ray-optix_amd"))
import torch
import workloads as W
from triro.ray.ray_optix import RayMeshIntersector
from triro.backend import ops as hops

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--opt", action="append", default=[], help="library option name=value (e.g. node_layout=0)"); a = ap.parse_args()
for kv in a.opt:
    k, val = kv.split("=", 1); hops.set_option(k, int(val))
dev = torch.device("cuda:0")
def wall(fn, reps):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort(); return ts[len(ts) // 2], ts[0]
for name, (v, f) in (("bunny stand-in", W.bunny_standin()), ("headline", W.headline_mesh(8))):
    v = torch.from_numpy(v).to(dev); f = torch.from_numpy(f).to(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = RayMeshIntersector(vertices=v, faces=f); torch.cuda.synchronize()
    first = (time.perf_counter() - t0) * 1e3
    reb = wall(lambda: r.update_raw(v, f), a.reps)
    hops.set_option("build_cache", 0)
    reb0 = wall(lambda: r.update_raw(v, f), a.reps)
    hops.set_option("build_cache", 1)
    ref = wall(lambda: r.refit(v), a.reps)
    print(json.dumps({"mesh": name, "opts": a.opt, "tris": int(f.shape[0]), "first_build_ms": round(first, 3),
                      "rebuild_ms_median_min": [round(x, 3) for x in reb],
                      "rebuild_nocache_ms_median_min": [round(x, 3) for x in reb0],
                      "refit_ms_median_min": [round(x, 3) for x in ref],
                      "mtris_per_s_rebuild": round(f.shape[0] / reb[0] / 1e3, 1)}))
