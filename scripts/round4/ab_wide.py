#!/usr/bin/env python3
"""A/B of the streaming launch on the binary grid nodes (wide=0) and on the 8-wide compressed nodes (wide=1):
the incoherent BASELINE configs (C3 any / first / closest / count on the C2 mesh, a C5(ii) shard, a 5.2 M-triangle
mesh).  One JSON line per (config, flavour); results of both flavours must be identical."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import workloads as W  # noqa: E402
import triro.backend.ops as hops  # noqa: E402
from triro.ray.ray_optix import RayMeshIntersector  # noqa: E402

dev = torch.device("cuda:0")
extra = dict(kv.split("=") for kv in sys.argv[1:])


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def timeit(fn, reps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def run(name, r, o, d, queries, reps=8, warm=3):
    n = o.shape[0]
    ref = {}
    for wide in (0, 1):
        hops.set_option("wide", wide)
        for q in queries:
            fn = {"any": r.intersects_any, "first": r.intersects_first, "closest": r.intersects_closest, "count": r.intersects_count}[q]
            out = fn(o, d)
            out = out if isinstance(out, tuple) else (out,)
            same = None
            if wide == 0:
                ref[q] = [x.clone() for x in out]
            else:
                same = all(torch.equal(a, b) for a, b in zip(out, ref[q]))
            sec = timeit(lambda: fn(o, d), reps, warm)
            st = None
            try:
                hops.set_option("stream", 2)
                m = min(n, 1 << 21)
                t = hops.trace_stats(r.as_wrapper, o[:m], d[:m], q)
                st = {"node_visits_per_ray": round(t["node_visits"] / t["rays"], 2), "tri_tests_per_ray": round(t["tri_tests"] / t["rays"], 2)}
            except Exception as exc:
                st = str(exc)
            finally:
                hops.set_option("stream", 1)
            print(json.dumps(dict(config=name, query=q, wide=wide, rays=n, ms=round(sec * 1e3, 4), grays_per_s=round(n / sec / 1e9, 3),
                                  identical_to_binary=same, stats=st, tris=int(r.bvh_info()["num_tris"]), depth=int(r.bvh_info()["depth"]))), flush=True)
    hops.set_option("wide", 2)


for k, v_ in extra.items():
    hops.set_option(k, int(v_))
v, f, label = W.bunny_mesh()
r = RayMeshIntersector(vertices=T(v), faces=T(f))
o, d = W.hash_rays_torch(10_000_000, 1234, v.min(0) * 1.5, v.max(0) * 1.5, device=dev)
run(f"C3: 10M hash rays vs the C2 mesh ({label})", r, o, d, ("any", "first", "closest", "count"))
del o, d, r
v, f = W.headline_mesh(8)
r = RayMeshIntersector(vertices=T(v), faces=T(f))
o, d = W.hash_rays_torch(12_500_000, 99, v.min(0) * 1.5, v.max(0) * 1.5, device=dev)
run("C5(ii) shard: 12.5M hash rays vs the headline mesh", r, o, d, ("closest", "any", "count"))
del o, d, r
if os.environ.get("AB_WIDE_BIG", "1") == "1":
    v, f = W.headline_mesh(9)
    r = RayMeshIntersector(vertices=T(v), faces=T(f))
    o, d = W.hash_rays_torch(12_500_000, 99, v.min(0) * 1.5, v.max(0) * 1.5, device=dev)
    run("5.2M-triangle sphere, 12.5M hash rays", r, o, d, ("closest",), reps=5, warm=2)
