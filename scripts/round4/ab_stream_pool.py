"""A/B of the streaming launch with pooled leaves (option stream_pool): results must be identical, time per query.
python scripts/round4/ab_stream_pool.py [--mins 32,64,96] [--waits 8,16,64] [--configs c5s,c3,big]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import workloads as W  # noqa: E402
import triro.backend.ops as hops  # noqa: E402
from triro.ray.ray_optix import RayMeshIntersector  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mins", default="32,64,96")
ap.add_argument("--waits", default="16")
ap.add_argument("--refills", default="32")
ap.add_argument("--configs", default="c5s,c3")
a = ap.parse_args()
dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for cfg in a.configs.split(","):
    if cfg == "c5s":
        v, f = W.headline_mesh(8)
        n, seed = 12_500_000, 99
    elif cfg == "c3":
        v, f, _ = W.bunny_mesh()
        n, seed = 10_000_000, 1234
    else:
        v, f = W.headline_mesh(9)          # 5.2 M triangles: DEEP addressing
        n, seed = 12_500_000, 99
    r = RayMeshIntersector(vertices=T(v), faces=T(f))
    o, d = W.hash_rays_torch(n, seed, v.min(0) * 1.5, v.max(0) * 1.5, device=dev)
    hops.set_option("wide", 0)
    qs = {"closest": lambda: r.intersects_closest(o, d), "first": lambda: r.intersects_first(o, d),
          "any": lambda: r.intersects_any(o, d), "count": lambda: r.intersects_count(o, d)}
    base, base_ms = {}, {}
    hops.set_option("stream_pool", 0)
    for q, fn in qs.items():
        out = fn()
        base[q] = [x.clone() for x in (out if isinstance(out, tuple) else (out,))]
        base_ms[q] = timed(fn, 8)
    for rf in [int(x) for x in a.refills.split(",")]:
        hops.set_option("stream_refill", rf)
        for pm in [int(x) for x in a.mins.split(",")]:
            for pw in [int(x) for x in a.waits.split(",")]:
                hops.set_option("stream_pool", 1)
                hops.set_option("stream_pool_min", pm)
                hops.set_option("stream_pool_wait", pw)
                row = {"config": cfg, "tris": int(len(f)), "rays": n, "stream_refill": rf, "pool_min": pm, "pool_wait": pw}
                for q, fn in qs.items():
                    out = fn()
                    out = out if isinstance(out, tuple) else (out,)
                    same = all(torch.equal(x, y) for x, y in zip(out, base[q]))
                    ms = timed(fn, 8)
                    row[q] = {"ms": round(ms, 4), "base_ms": round(base_ms[q], 4), "ratio": round(ms / base_ms[q], 3), "same": bool(same)}
                print(json.dumps(row), flush=True)
    hops.set_option("stream_pool", 0)
    hops.set_option("stream_refill", 32)
    del r
