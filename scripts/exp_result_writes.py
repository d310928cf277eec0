#!/usr/bin/env python3
"""Upper bound on what staging the streaming launch's results could bring (VERDICT r02 item 3 ii): the same
trace with the five dense outputs (26 B/ray in five arrays, written lane by lane as rays finish), with the
12-byte packed record (ONE store per ray) and -- the floor -- with `first` (4 B/ray).
usage (GPU box): python scripts/exp_result_writes.py > gpurun_out/result_writes.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import workloads as W  # noqa: E402
from triro.ray.ray_optix import RayMeshIntersector  # noqa: E402

dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731


def timed(fn, reps=10, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, (v, f), n, seed in (("C5(ii) shard", W.headline_mesh(8), 12_500_000, 99), ("C3' (bunny stand-in)", W.bunny_standin(), 10_000_000, 1234)):
    r = RayMeshIntersector(vertices=T(v), faces=T(f))
    o, d = W.hash_rays_torch(n, seed, v.min(0) * 1.5, v.max(0) * 1.5, device=dev)
    for rep in range(2):
        row = {"config": name, "rays": n,
               "closest_dense_ms": round(timed(lambda: r.intersects_closest(o, d)), 4),
               "closest_packed_ms": round(timed(lambda: r.intersects_closest_packed(o, d)), 4),
               "first_ms": round(timed(lambda: r.intersects_first(o, d)), 4)}
        print(json.dumps(row), flush=True)
    del o, d, r
