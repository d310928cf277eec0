#!/usr/bin/env python3
"""Throughput and roofline figures of the other BASELINE.json configs on one GPU (parity for them is in
tests/test_gpu_configs.py, tests/test_gpu_round3.py, tests/test_gpu_interior.py).  One JSON line per
config; run on the GPU box:
    python scripts/bench_configs.py > gpurun_out/configs.jsonl

Per config: ms per call (wall clock over back-to-back calls, synchronised at the end), Mrays/s,
  algorithmic_bytes  compulsory I/O of the query (closest 50 B/ray, any 25, first / count 28,
                     location 60 B/ray + 20 B/hit, compaction + 30 B/hit) + ONE read of the node and triangle
                     arrays the launch walks (32-byte grid nodes or exact 64-byte nodes: tr_bvh_last_launch)
  achieved_gbps, frac  algorithmic_bytes / time against the 8 TB/s HBM peak
  traffic_bytes      FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE per launch from the rocprofv3 PMC
                     summary of the same config under profiles/ (scripts/profile_query.sh), when there is one
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import workloads as W  # noqa: E402
from triro.ray.ray_optix import RayMeshIntersector  # noqa: E402

dev = torch.device("cuda:0")
HBM_PEAK = 8000.0e9
PER_RAY = {"closest": 50, "any": 25, "first": 28, "count": 28, "location": 60, "closest_compact": 50}
SUMMARY_TAG = os.environ.get("PROFILE_ROUND", "r03")


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def timeit(fn, reps=20, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def timeit_median(fn, reps=9, warm=2):
    """host-synchronous operations (build, refit): median of individually timed calls"""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def traffic_of(tag):
    p = os.path.join(ROOT, "profiles", f"{SUMMARY_TAG}_{tag}_summary.json")
    if not os.path.exists(p):
        return None, None
    try:
        d = json.load(open(p)).get("derived", {})
        if "fetch_bytes_x2_gfx950" in d and "write_bytes" in d:
            return int(d["fetch_bytes_x2_gfx950"] + d["write_bytes"]), os.path.relpath(p, ROOT)
    except Exception:
        pass
    return None, None


def report(name, n, sec, **kw):
    print(json.dumps(dict(config=name, rays=n, ms=round(sec * 1e3, 4), mrays_per_s=round(n / sec / 1e6, 1), **kw)), flush=True)


def query(name, r, n, query, fn, tag=None, reps=20, warm=8, streaming=False, hits=None, **kw):
    """time one query of one config and attach the roofline figures"""
    sec = timeit(fn, reps=reps, warm=warm)
    info = r.bvh_info()
    grid = streaming or query in ("count", "location")
    if not grid:
        try:
            grid = bool(r.as_wrapper.last_launch()["grid_nodes"])
        except Exception:
            grid = False
    node_bytes = info["num_nodes"] * (32 if grid else 64)
    algo = n * PER_RAY[query] + node_bytes + info["tri_bytes"]
    if hits is not None:
        algo += hits * (20 if query == "location" else 30)
    traffic, src = traffic_of(tag) if tag else (None, None)
    ach = algo / sec
    report(name, n, sec, tris=int(info["num_tris"]), query=query, node_flavour="32-byte grid nodes" if grid else "exact 64-byte nodes",
           algorithmic_bytes=int(algo), achieved_gbps=round(ach / 1e9, 1), frac=round(ach / HBM_PEAK, 5),
           traffic_bytes=traffic, traffic_over_algorithmic=(round(traffic / algo, 2) if traffic else None), traffic_source=src,
           **({"hits": int(hits)} if hits is not None else {}), **kw)


v, f, bunny_label = W.bunny_mesh()
r = RayMeshIntersector(vertices=T(v), faces=T(f))
o, d = W.pinhole_grid(1024, 1024, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
ot, dt = T(o), T(d)
query(f"C2 closest, bunny {bunny_label}, 1024^2 pinhole", r, 1 << 20, "closest", lambda: r.intersects_closest(ot, dt), tag="c2closest")
n = 10_000_000
o3, d3 = W.hash_rays_torch(n, 1234, v.min(0) * 1.5, v.max(0) * 1.5, device=dev)
query("C3 any, 10M hash shadow rays vs the C2 mesh", r, n, "any", lambda: r.intersects_any(o3, d3), tag="c3any", reps=10, warm=4, streaming=True)
query("C3' closest on the same 10M rays", r, n, "closest", lambda: r.intersects_closest(o3, d3), tag="c3closest", reps=10, warm=4, streaming=True)
del o3, d3
v, f = W.nested_shells(7)
r = RayMeshIntersector(vertices=T(v), faces=T(f))
o, d = W.pinhole_grid(1024, 1024)
ot, dt = T(o), T(d)
nh = r.intersects_location(ot, dt)[0].shape[0]
query("C4 location (count+scan+fill), 4 nested shells, 1024^2 pinhole", r, 1 << 20, "location",
      lambda: r.intersects_location(ot, dt), tag="c4location", reps=10, hits=nh)
nhit = int(r.intersects_closest(ot, dt)[0].sum())
query("C4 closest + stream compaction", r, 1 << 20, "closest_compact", lambda: r.intersects_closest(ot, dt, stream_compaction=True),
      tag="c4closest", hits=nhit)
query("C4 closest", r, 1 << 20, "closest", lambda: r.intersects_closest(ot, dt), tag="c4closest")
query("C4 count", r, 1 << 20, "count", lambda: r.intersects_count(ot, dt), tag="c4count")
v, f = W.interior_room()
r = RayMeshIntersector(vertices=T(v), faces=T(f))
_, d = W.ref_shape_rays(W.INTERIOR_EYE, W.INTERIOR_TARGET)
ot = torch.from_numpy(np.array(W.INTERIOR_EYE, np.float32)).to(dev).expand(360, 640, 3)
dt = T(d)
query("ROOM closest, interior scene, camera inside, 640x360 reference-shaped rays", r, 640 * 360, "closest",
      lambda: r.intersects_closest(ot, dt), tag="roomclosest", reps=50, warm=20)
query("ROOM count, same rays", r, 640 * 360, "count", lambda: r.intersects_count(ot, dt), tag="roomcount", reps=30, warm=12)
nh = r.intersects_location(ot, dt)[0].shape[0]
query("ROOM location, same rays", r, 640 * 360, "location", lambda: r.intersects_location(ot, dt), reps=20, warm=8, hits=nh)
v, f = W.terrain()
r = RayMeshIntersector(vertices=T(v), faces=T(f))
_, d = W.ref_shape_rays(W.TERRAIN_EYE, W.TERRAIN_TARGET, 1024, 576, 444.0 * 1024 / 640)
ot = torch.from_numpy(np.array(W.TERRAIN_EYE, np.float32)).to(dev).expand(576, 1024, 3)
dt = T(d)
query("TERRAIN closest, open height field, grazing camera, 1024x576 reference-shaped rays", r, 1024 * 576, "closest",
      lambda: r.intersects_closest(ot, dt), reps=50, warm=20)
query("TERRAIN count, same rays", r, 1024 * 576, "count", lambda: r.intersects_count(ot, dt), reps=30, warm=12)
nh = r.intersects_location(ot, dt)[0].shape[0]
query("TERRAIN location, same rays", r, 1024 * 576, "location", lambda: r.intersects_location(ot, dt), reps=20, warm=8, hits=nh)
v, f = W.headline_mesh(8)
t0 = time.perf_counter()
r = RayMeshIntersector(vertices=T(v), faces=T(f))
torch.cuda.synchronize()
build_s = time.perf_counter() - t0
report("C5 BVH first build (incl. host->device upload), headline mesh", len(f), build_s, note="units are triangles, not rays")
vd, fd = r.mesh_vertices, r.mesh_faces
report("C5 BVH rebuild (update_raw, mesh resident), headline mesh", len(f), timeit_median(lambda: r.update_raw(vd, fd)),
       note="units are triangles")
o, d = W.pinhole_grid(1024, 1024, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
ot, dt = T(o), T(d)
query("C5(i) closest, headline mesh, 1024^2 pinhole (the metric's config; bench.py is the authority)", r, 1 << 20, "closest",
      lambda: r.intersects_closest(ot, dt), tag="c5i", reps=100, warm=40)
query("C5(i) count on the same rays", r, 1 << 20, "count", lambda: r.intersects_count(ot, dt), reps=20, warm=10)
n = 100_000_000 // 8
o5, d5 = W.hash_rays_torch(n, 99, v.min(0) * 1.5, v.max(0) * 1.5, device=dev)
query("C5(ii) closest, one 12.5M-ray shard of the 100M hash rays", r, n, "closest", lambda: r.intersects_closest(o5, d5),
      tag="c5s", reps=8, streaming=True)
v2 = W.displaced(v, seed=1, amplitude=0.05)
vt = T(v2)
report("refit of the headline mesh (same faces, new vertices)", len(f), timeit_median(lambda: r.refit(vt)),
       note="units are triangles")
