#!/usr/bin/env python3
"""VERDICT r03 "next" #6: can ONE call get the ramp-down overlap that two calls in flight get (0.205 -> 0.183 ms per
launch)?  The idea was to cut the launch into two halves on two streams inside the call.  This measures the concept
from outside the library: the headline batch as (a) one launch, (b) two half-image launches on two streams joined by
events (what a split inside the call would do), in top / bottom halves and in interleaved 64-row bands, and (c) two
FULL launches in flight (the `pipelined` companion) -- per batch of 1 M rays.  One JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import workloads as W  # noqa: E402
from triro.ray.ray_optix import RayMeshIntersector  # noqa: E402

dev = torch.device("cuda:0")
v, f = W.headline_mesh(8)
r = RayMeshIntersector(vertices=torch.from_numpy(v).to(dev), faces=torch.from_numpy(f).to(dev))
o_np, d_np = W.pinhole_grid(1024, 1024, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
o = torch.from_numpy(np.ascontiguousarray(o_np)).to(dev)
d = torch.from_numpy(d_np).to(dev)
s = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
main = torch.cuda.current_stream(dev)


def timed(fn, reps=600, warm=80):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def one():
    r.intersects_closest(o, d)


def halves(parts):
    def fn():
        evs = []
        for k, (oo, dd) in enumerate(parts):
            s[k].wait_stream(main)
            with torch.cuda.stream(s[k]):
                r.intersects_closest(oo, dd)
            main.wait_stream(s[k])
    return fn


top_bottom = [(o[:512], d[:512]), (o[512:], d[512:])]
# interleaved bands of 64 rows: rows [0:64], [128:192], ... and the others (each a [8, 64, 1024, 3] view: general strides)
ob, db = o.view(8, 2, 64, 1024, 3), d.view(8, 2, 64, 1024, 3)
bands = [(ob[:, 0].reshape(512, 1024, 3).contiguous(), db[:, 0].reshape(512, 1024, 3).contiguous()),
         (ob[:, 1].reshape(512, 1024, 3).contiguous(), db[:, 1].reshape(512, 1024, 3).contiguous())]
k = [0]


def two_in_flight():
    with torch.cuda.stream(s[k[0] & 1]):
        r.intersects_closest(o, d)
    k[0] += 1


for st in s:
    st.wait_stream(main)
res = {"one_launch_ms": round(timed(one), 4),
       "two_half_launches_top_bottom_ms": round(timed(halves(top_bottom)), 4),
       "two_half_launches_interleaved_bands_ms": round(timed(halves(bands)), 4),
       "half_launch_alone_ms": round(timed(lambda: r.intersects_closest(*top_bottom[0])), 4),
       "two_full_launches_in_flight_ms_per_launch": round(timed(two_in_flight), 4),
       "note": "per 1 M-ray batch; the half launches of one batch start together and end together: their tails coincide, "
               "where two full launches in flight are offset by half a launch"}
print(json.dumps(res), flush=True)
