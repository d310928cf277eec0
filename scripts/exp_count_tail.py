#!/usr/bin/env python3
"""C4 count (uniform waves of ~150 trips): how much of the launch is its ramp-down?  One stream vs two launches in
flight, and the streaming launch (ray refill) forced on the same image.
usage (GPU box): python scripts/exp_count_tail.py > gpurun_out/count_tail.jsonl"""
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

dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731


def run(fn, nstreams, steps, warm):
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream(dev))
    out = None
    for k in range(warm * nstreams):
        with torch.cuda.stream(streams[k % nstreams]):
            out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        with torch.cuda.stream(streams[k % nstreams]):
            out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


v, f = W.nested_shells(7)
r = RayMeshIntersector(vertices=T(v), faces=T(f))
for res in (1024, 768, 1448):
    o, d = W.pinhole_grid(res, res)
    ot, dt = T(o), T(d)
    ref = r.intersects_count(ot, dt).clone()
    for label, opts, ns in (("direct, 1 stream", {}, 1), ("direct, 2 streams", {}, 2), ("streaming launch, 1 stream", {"stream": 2}, 1),
                            ("streaming launch, refill 8", {"stream": 2, "stream_refill": 8}, 1), ("direct, 1 stream", {}, 1)):
        for k_, v_ in opts.items():
            hops.set_option(k_, v_)
        ms, out = run(lambda: r.intersects_count(ot, dt), ns, 60, 12)
        for k_ in opts:
            hops.set_option(k_, {"stream": 1, "stream_refill": 32}[k_])
        print(json.dumps({"scene": "c4 shells", "query": "count", "rays": res * res, "mode": label, "ms_per_launch": round(ms, 4),
                          "results_identical": bool(torch.equal(out, ref))}), flush=True)
