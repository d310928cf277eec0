#!/usr/bin/env python3
"""Incoherent (hash) rays of a given count against the headline mesh or the bunny stand-in.
usage: python scripts/run_hash.py --n 1048576 --mesh headline --query closest [--opt k=v ...]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np, torch
import workloads as W
from triro.backend import ops as hops
from triro.ray.ray_optix import RayMeshIntersector
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20); ap.add_argument("--mesh", default="headline"); ap.add_argument("--query", default="closest")
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--shape", default="", help="e.g. 8,-1 : reshape the batch to [8, n/8, 3]"); ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
dev = torch.device("cuda:0")
for kv in a.opt:
    k, v_ = kv.split("="); hops.set_option(k, int(v_))
v, f = W.headline_mesh(8) if a.mesh == "headline" else W.bunny_standin()
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
r = RayMeshIntersector(vertices=T(v), faces=T(f))
o, d = W.hash_rays_torch(a.n, 99, v.min(0) * 1.5, v.max(0) * 1.5, device=dev)
if a.shape:
    dims = [int(x) for x in a.shape.split(",")]
    o, d = o.reshape(*dims, 3), d.reshape(*dims, 3)
fn = {"closest": lambda: r.intersects_closest(o, d), "any": lambda: r.intersects_any(o, d), "count": lambda: r.intersects_count(o, d)}[a.query]
for _ in range(4): fn()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
for e0, e1 in ev:
    e0.record(); fn(); e1.record()
torch.cuda.synchronize()
ms = [e0.elapsed_time(e1) for e0, e1 in ev]
print(json.dumps({"mesh": a.mesh, "query": a.query, "rays": a.n, "opts": a.opt, "ms_mean": round(float(np.mean(ms)), 4), "ms_min": round(min(ms), 4),
                  "mrays_per_s": round(a.n / np.mean(ms) / 1e3, 1)}), flush=True)
