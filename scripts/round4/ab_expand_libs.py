"""A/B of the record expansion kernels between two builds of the library on the same box (separate processes):
python scripts/round4/ab_expand_libs.py [--old-abi]   with TRIRO_HIP_LIBRARY naming the build.
--old-abi: the build predates ABI 8 (no 4-byte records): the binding's table is trimmed for this run only."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import workloads as W  # noqa: E402
import triro.backend.ops as hops  # noqa: E402

old = "--old-abi" in sys.argv
if old:
    hops.ABI_VERSION = 7
    for k in ("tr_intersects_closest_slots", "tr_closest_from_slots"):
        hops.ABI.pop(k)
from triro.ray.ray_optix import RayMeshIntersector  # noqa: E402

dev = torch.device("cuda:0")
v, f = W.headline_mesh(8)
r = RayMeshIntersector(vertices=torch.from_numpy(v).to(dev), faces=torch.from_numpy(f).to(dev))
rad = float(np.linalg.norm(v, axis=1).max())
o_np, d_np = W.pinhole_grid(1024, 1024, distance=2.5 * rad)
o = torch.from_numpy(np.ascontiguousarray(o_np)).to(dev)
rows = 7 * 1024
d = torch.from_numpy(np.concatenate([np.roll(d_np, 7 * k, axis=0) for k in range(1, 8)], 0)).to(dev)
oo = o.repeat(7, 1, 1)
n = rows * 1024
rec = r.intersects_closest_packed(oo, d, slots=True)
outs = (torch.zeros(n, dtype=torch.bool, device=dev), torch.zeros(n, dtype=torch.bool, device=dev),
        torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros((n, 3), device=dev), torch.zeros((n, 2), device=dev))


def timed(fn, reps=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


res = {"lib": os.environ.get("TRIRO_HIP_LIBRARY", "default")}
for k in range(3):
    res[f"packed12_rows_ms_{k}"] = round(timed(lambda: r.closest_expand(rec, outs=outs, slots=True, row_length=1024)), 4)
res["packed12_flat_ms"] = round(timed(lambda: r.closest_expand(rec, outs=outs, slots=True)), 4)
if not old:
    sl = r.intersects_closest_slots(oo, d)
    for k in range(3):
        res[f"slot4_rows_ms_{k}"] = round(timed(lambda: r.closest_from_slots(oo, d, sl, outs=outs, row_length=1024)), 4)
    res["slot4_flat_ms"] = round(timed(lambda: r.closest_from_slots(oo.reshape(-1, 3), d.reshape(-1, 3), sl, outs=outs)), 4)
    ob = o[:1, :1].expand(rows, 1024, 3)
    res["slot4_rows_broadcast_origin_ms"] = round(timed(lambda: r.closest_from_slots(ob, d, sl, outs=outs, row_length=1024)), 4)
print(res)
