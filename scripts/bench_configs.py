#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs on one GPU (parity for them is in
tests/test_gpu_configs.py).  Prints one JSON line per config; run on the GPU box:
    python scripts/bench_configs.py > gpurun_out/configs.jsonl"""
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


def report(name, n, sec, **kw):
    print(json.dumps(dict(config=name, rays=n, ms=round(sec * 1e3, 4), mrays_per_s=round(n / sec / 1e6, 1), **kw)), flush=True)


v, f = W.bunny_standin()
r = RayMeshIntersector(vertices=T(v), faces=T(f))
o, d = W.pinhole_grid(1024, 1024, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
ot, dt = T(o), T(d)
report("C2 closest, bunny stand-in 81920 tris, 1024^2 pinhole", 1 << 20, timeit(lambda: r.intersects_closest(ot, dt)), tris=len(f))
n = 10_000_000
o3, d3 = W.hash_rays_torch(n, 1234, v.min(0) * 1.5, v.max(0) * 1.5, device=dev)
report("C3 any, 10M hash shadow rays vs bunny stand-in", n, timeit(lambda: r.intersects_any(o3, d3), reps=10), tris=len(f))
report("C3' closest on the same 10M rays", n, timeit(lambda: r.intersects_closest(o3, d3), reps=10), tris=len(f))
del o3, d3
v, f = W.nested_shells(7)
r = RayMeshIntersector(vertices=T(v), faces=T(f))
o, d = W.pinhole_grid(1024, 1024)
ot, dt = T(o), T(d)
nh = r.intersects_location(ot, dt)[0].shape[0]
report("C4 location (count+scan+fill), 4 nested shells 1310720 tris, 1024^2 pinhole", 1 << 20,
       timeit(lambda: r.intersects_location(ot, dt), reps=10), tris=len(f), hits=nh)
report("C4 closest + stream compaction", 1 << 20, timeit(lambda: r.intersects_closest(ot, dt, stream_compaction=True)), tris=len(f))
report("C4 count", 1 << 20, timeit(lambda: r.intersects_count(ot, dt)), tris=len(f))
v, f = W.interior_room()
r = RayMeshIntersector(vertices=T(v), faces=T(f))
_, d = W.ref_shape_rays(W.INTERIOR_EYE, W.INTERIOR_TARGET)
ot = torch.from_numpy(np.array(W.INTERIOR_EYE, np.float32)).to(dev).expand(360, 640, 3)
dt = T(d)
report("ROOM closest, interior scene 909088 tris, camera inside, 640x360 reference-shaped rays", 640 * 360,
       timeit(lambda: r.intersects_closest(ot, dt), reps=50, warm=20), tris=len(f))
report("ROOM count, same rays", 640 * 360, timeit(lambda: r.intersects_count(ot, dt), reps=30, warm=12), tris=len(f))
report("ROOM location, same rays", 640 * 360, timeit(lambda: r.intersects_location(ot, dt), reps=20, warm=8), tris=len(f))
v, f = W.headline_mesh(8)
t0 = time.perf_counter()
r = RayMeshIntersector(vertices=T(v), faces=T(f))
torch.cuda.synchronize()
build_s = time.perf_counter() - t0
report("C5 BVH first build (incl. host->device upload), headline mesh", len(f), build_s, note="units are triangles, not rays")
vd, fd = r.mesh_vertices, r.mesh_faces
report("C5 BVH rebuild (update_raw, mesh resident), headline mesh", len(f), timeit_median(lambda: r.update_raw(vd, fd)),
       note="units are triangles")
n = 100_000_000 // 8
o5, d5 = W.hash_rays_torch(n, 99, v.min(0) * 1.5, v.max(0) * 1.5, device=dev)
report("C5(ii) closest, one 12.5M-ray shard of the 100M hash rays, 1310720 tris", n,
       timeit(lambda: r.intersects_closest(o5, d5), reps=8), tris=len(f))
v2 = W.displaced(v, seed=1, amplitude=0.05)
vt = T(v2)
report("refit of the headline mesh (same faces, new vertices)", len(f), timeit_median(lambda: r.refit(vt)),
       note="units are triangles")
