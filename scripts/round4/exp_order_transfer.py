#!/usr/bin/env python3
"""VERDICT r03 "next" #7b: the first launch at 1024^2 after 12 launches at 768^2 on the same handle and stream (a
resolution change), with and without the resampled launch order (option order_transfer); next to it the steady state
and the first launch on a fresh handle.  Kernel-to-kernel HIP events around the call (the two scheduling kernels of
the transfer are inside).  One JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np, torch
import workloads as W
import triro.backend.ops as hops
from triro.ray.ray_optix import RayMeshIntersector
dev = torch.device("cuda:0")
v, f = W.headline_mesh(8)
vt, ft = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
rad = float(np.linalg.norm(v, axis=1).max())


def rays(res):
    o_np, d_np = W.pinhole_grid(res, res, distance=2.5 * rad)
    return torch.from_numpy(np.ascontiguousarray(o_np)).to(dev), torch.from_numpy(d_np).to(dev)


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


o7, d7 = rays(768); o10, d10 = rays(1024); o5, d5 = rays(512)
# the caching allocator must already own blocks for every output size: a hipMalloc inside the timed call would leave the
# GPU idle between the two events
warm = RayMeshIntersector(vertices=vt, faces=ft)
for oo, dd in ((o7, d7), (o10, d10), (o5, d5)):
    for _ in range(3):
        warm.intersects_closest(oo, dd)
del warm
torch.cuda.synchronize()
res = {}
for transfer in (0, 1):
    hops.set_option("order_transfer", transfer)
    firsts, seconds, downs = [], [], []
    for rep in range(10):
        r = RayMeshIntersector(vertices=vt, faces=ft)
        for _ in range(12):
            r.intersects_closest(o7, d7)
        firsts.append(timed(lambda: r.intersects_closest(o10, d10)))
        seconds.append(timed(lambda: r.intersects_closest(o10, d10)))
        for _ in range(10):
            r.intersects_closest(o10, d10)
        downs.append(timed(lambda: r.intersects_closest(o5, d5)))
        del r
    res[f"order_transfer={transfer}"] = {"first_1024_after_12x768_ms": round(float(np.median(firsts)), 4), "second_1024_ms": round(float(np.median(seconds)), 4),
                                         "first_512_after_1024_ms": round(float(np.median(downs)), 4)}
hops.set_option("order_transfer", 1)
r = RayMeshIntersector(vertices=vt, faces=ft)
for _ in range(60):
    r.intersects_closest(o10, d10)
res["steady_1024_ms"] = round(float(np.median([timed(lambda: r.intersects_closest(o10, d10)) for _ in range(20)])), 4)
for _ in range(60):
    r.intersects_closest(o5, d5)
res["steady_512_ms"] = round(float(np.median([timed(lambda: r.intersects_closest(o5, d5)) for _ in range(20)])), 4)
print(json.dumps(res), flush=True)
