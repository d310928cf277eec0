#!/usr/bin/env python3
"""Run ONE query of one BASELINE.json config repeatedly (timing and rocprofv3 target).

  python scripts/run_query.py --config c4 --query count [--steps 20] [--warmup 4] [--opt name=value ...]
  configs: c2 (bunny stand-in 81 920 tris, 1024^2 pinhole), c3 (10 M hash shadow rays vs the stand-in),
           c4 (4 nested shells 1 310 720 tris, 1024^2 pinhole), c5i (headline mesh, 1024^2 pinhole; --res),
           c5s (headline mesh, one 12.5 M-ray shard of the 100 M hash rays)
           terrain (workloads.terrain(): 1 048 352-tri height field, grazing camera, 16:9 rays, --res = width)
           soup (1 000 000 small triangles scattered in a cube, camera outside, 16:9 rays, --res = width)
           room (workloads.interior_room(): 909 088 tris, camera INSIDE, 640x360 rays in the reference's
           published shape -- stride-0 origin; --res scales the image: res x res*9/16)
  queries: closest any first count location closest_compact
Prints one JSON line: ms per call (HIP events on the launch stream, mean and min) and Mrays/s."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import workloads as W  # noqa: E402
from triro.backend import ops as hops  # noqa: E402
from triro.ray.ray_optix import RayMeshIntersector  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c4")
ap.add_argument("--query", default="count")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=4)
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--subdiv", type=int, default=8, help="c5i / c5s: icosphere subdivisions of the headline mesh")
ap.add_argument("--flat", action="store_true", help="pass image-shaped rays as a flat [N, 3] batch")
ap.add_argument("--stats", action="store_true", help="also run the instrumented kernel (traversal counters)")
ap.add_argument("--each", action="store_true", help="also print the time of every step")
ap.add_argument("--rays", type=int, default=0, help="c3 / c5s: number of hash rays instead of the config's 10 M / 12.5 M")
ap.add_argument("--far", type=float, default=1.0, help="pinhole configs: the camera this many times farther away with the field of view narrowed "
                "by the same factor (the same image of the mesh from a far origin: ADVICE r05, the margins of the fused box test and of the inside test)")
ap.add_argument("--presort", type=int, default=0, help="c3 / c5s: sort the rays by (origin cell with this many bits per axis, direction octant) BEFORE the timed region")
a = ap.parse_args()
dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
for kv in a.opt:
    k, v_ = kv.split("=", 1)
    hops.set_option(k, int(v_))

if a.config in ("c2", "c3"):
    v, f, _bunny_label = W.bunny_mesh()
elif a.config == "c4":
    v, f = W.nested_shells(7)
elif a.config == "room":
    v, f = W.interior_room()
elif a.config == "terrain":
    v, f = W.terrain()
elif a.config == "soup":
    v, f = W.random_soup(1_000_000, seed=5, extent=1.0, size=0.02)      # tests/test_gpu_soup.py's cloud
else:
    v, f = W.headline_mesh(a.subdiv)
r = RayMeshIntersector(vertices=T(v), faces=T(f))
rad = float(np.linalg.norm(v, axis=1).max())
if a.config == "c3":
    n = a.rays or 10_000_000
    o, d = W.hash_rays_torch(n, 1234, v.min(0) * 1.5, v.max(0) * 1.5, device=dev)
elif a.config == "c5s":
    n = a.rays or 100_000_000 // 8
    o, d = W.hash_rays_torch(n, 99, v.min(0) * 1.5, v.max(0) * 1.5, device=dev)
elif a.config in ("room", "terrain", "soup"):
    # the reference's ray shape (16:9, stride-0 origin); --res = image width (default 640; terrain also 1024 ...)
    w = 640 if (a.res == 1024 and a.config == "room") else a.res
    h = w * 9 // 16
    eye, target = {"room": (W.INTERIOR_EYE, W.INTERIOR_TARGET), "terrain": (W.TERRAIN_EYE, W.TERRAIN_TARGET),
                   "soup": ((-2.2, 0.6, -1.8), (0.2, -0.1, 0.1))}[a.config]
    _, dn = W.ref_shape_rays(eye, target, w, h, 444.0 * w / 640)
    o = torch.from_numpy(np.array(eye, np.float32)).to(dev).expand(h, w, 3)
    d = T(dn)
    if a.flat:
        o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
    n = w * h
else:
    dist = 2.5 if a.config == "c4" else 2.5 * rad
    import math
    vf = 40.0 if a.far == 1.0 else 2.0 * math.degrees(math.atan(math.tan(math.radians(20.0)) / a.far))
    on, dn = W.pinhole_grid(a.res, a.res, vfov_deg=vf, distance=dist * a.far)
    o, d = T(on), T(dn)
    if a.flat:
        o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
    n = a.res * a.res
if a.presort > 0:
    lo_t = torch.as_tensor(v.min(0) * 1.5, device=dev)
    ext_t = torch.as_tensor(v.max(0) * 1.5 - v.min(0) * 1.5, device=dev)
    cell = ((o.reshape(-1, 3) - lo_t) / ext_t * (1 << a.presort)).clamp_(0, (1 << a.presort) - 1).to(torch.int32)
    key = (((cell[:, 0] << a.presort) | cell[:, 1]) << a.presort) | cell[:, 2]
    dd = d.reshape(-1, 3)
    key = (key << 3) | ((dd[:, 0] < 0).to(torch.int32) << 2) | ((dd[:, 1] < 0).to(torch.int32) << 1) | (dd[:, 2] < 0).to(torch.int32)
    perm = torch.sort(key)[1]
    o, d = o.reshape(-1, 3)[perm].contiguous(), dd[perm].contiguous()
    del cell, key, perm, dd
fn = {"closest": lambda: r.intersects_closest(o, d), "any": lambda: r.intersects_any(o, d),
      "first": lambda: r.intersects_first(o, d), "count": lambda: r.intersects_count(o, d),
      "location": lambda: r.intersects_location(o, d),
      "closest_compact": lambda: r.intersects_closest(o, d, stream_compaction=True)}[a.query]
for _ in range(a.warmup):
    out = fn()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
t0 = time.perf_counter()
for e0, e1 in ev:
    e0.record()
    out = fn()
    e1.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / a.steps
ms = [e0.elapsed_time(e1) for e0, e1 in ev]
res = {"config": a.config, "query": a.query, "rays": n, "tris": int(len(f)), "opts": a.opt,
       "ms_mean": round(float(np.mean(ms)), 4), "ms_min": round(min(ms), 4), "wall_ms": round(wall * 1e3, 4),
       "mrays_per_s": round(n / np.mean(ms) / 1e3, 1)}
if a.each:
    res["ms_each"] = [round(x, 3) for x in ms]
if a.query == "location":
    res["hits"] = int(out[0].shape[0])
if a.stats:
    q = a.query if a.query in hops.QUERY_IDS else "closest"
    st = hops.trace_stats(r.as_wrapper, o, d, q)
    res["stats_per_ray"] = {k: round(v_ / max(st["rays"], 1), 3) for k, v_ in st.items() if k != "rays"}
print(json.dumps(res), flush=True)
