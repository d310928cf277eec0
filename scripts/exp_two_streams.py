#!/usr/bin/env python3
"""Does a second stream fill the ramp-down of a launch?  The headline batch traced K times back to back on ONE
stream (what bench.py times) against the same K launches dealt round-robin to TWO / THREE streams (each with
its own output buffers and its own learned launch order: the order is kept per (handle, stream)).
usage (GPU box): python scripts/exp_two_streams.py > gpurun_out/two_streams.jsonl"""
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
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731


def run(r, ot, dt, nstreams, steps, warm):
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream(dev))
    outs = [None] * nstreams
    for k in range(warm * nstreams):
        with torch.cuda.stream(streams[k % nstreams]):
            outs[k % nstreams] = r.intersects_closest(ot, dt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        with torch.cuda.stream(streams[k % nstreams]):
            outs[k % nstreams] = r.intersects_closest(ot, dt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, outs


for name, (v, f), res in (("headline", W.headline_mesh(8), 1024), ("headline", W.headline_mesh(8), 512), ("c4 shells", W.nested_shells(7), 1024),
                          ("interior", W.interior_room(), 0)):
    r = RayMeshIntersector(vertices=T(v), faces=T(f))
    if res:
        rad = float(np.linalg.norm(v, axis=1).max())
        o, d = W.pinhole_grid(res, res, distance=2.5 * rad) if name == "headline" else W.pinhole_grid(res, res)
        ot, dt = T(o), T(d)
    else:
        _, d = W.ref_shape_rays(W.INTERIOR_EYE, W.INTERIOR_TARGET)
        ot = torch.from_numpy(np.array(W.INTERIOR_EYE, np.float32)).to(dev).expand(360, 640, 3)
        dt = T(d)
    ref = None
    for ns in (1, 2, 3, 1, 2):
        ms, outs = run(r, ot, dt, ns, 600, 40)
        if ref is None:
            ref = [x.clone() for x in outs[0]]
        same = all(all(torch.equal(a, b) for a, b in zip(o_, ref)) for o_ in outs if o_ is not None)
        print(json.dumps({"scene": name, "rays": int(ot.shape[0] * ot.shape[1]), "streams": ns, "ms_per_launch": round(ms, 4),
                          "mrays_per_s": round(ot.shape[0] * ot.shape[1] / ms / 1e3, 1), "results_identical": bool(same)}), flush=True)
