#!/usr/bin/env python3
"""Evidence for SURVEY.md 8(f) rank 3 (ray-reordering front end for incoherent batches): what a sort
of the rays buys the trace and what it costs, on BASELINE configs C3 / C5(ii).

Rays are binned by (origin cell on a 2^b grid over the mesh bounds, direction octant) -- the key a
front end would use --, sorted with torch.sort (stand-in for a dedicated radix pass; the key and
payload traffic of such a pass is reported next to it), traced in sorted order, and the results
scattered back.  Prints one JSON line per configuration:
  trace_ms          the product as it is (unsorted rays)
  trace_sorted_ms   the same kernel on the sorted copy (rays gathered beforehand)
  key_ms / sort_ms / gather_ms / scatter_ms   the front end's own passes (torch ops)
  net_ms            key + sort + gather + trace_sorted + scatter
usage (GPU box): python scripts/exp_ray_sort.py > gpurun_out/ray_sort.jsonl"""
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


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


def run(name, v, f, n, seed, query, bits):
    r = RayMeshIntersector(vertices=T(v), faces=T(f))
    lo, hi = v.min(0) * 1.5, v.max(0) * 1.5
    o, d = W.hash_rays_torch(n, seed, lo, hi, device=dev)
    q = (lambda oo, dd: r.intersects_closest(oo, dd)) if query == "closest" else (lambda oo, dd: (r.intersects_any(oo, dd),))
    trace_ms, ref = timed(lambda: q(o, d))
    lo_t, ext_t = torch.as_tensor(lo, device=dev), torch.as_tensor(hi - lo, device=dev)

    def make_key():
        cell = ((o - lo_t) / ext_t * (1 << bits)).clamp_(0, (1 << bits) - 1).to(torch.int32)
        key = cell[:, 0]
        key = (key << bits) | cell[:, 1]
        key = (key << bits) | cell[:, 2]
        octant = ((d[:, 0] < 0).to(torch.int32) << 2) | ((d[:, 1] < 0).to(torch.int32) << 1) | (d[:, 2] < 0).to(torch.int32)
        return (key << 3) | octant
    key_ms, key = timed(make_key)
    sort_ms, (_, perm) = timed(lambda: torch.sort(key))
    gather_ms, (os_, ds_) = timed(lambda: (o[perm], d[perm]))
    trace_sorted_ms, out_s = timed(lambda: q(os_, ds_))

    def scatter():
        outs = []
        for x in out_s:
            y = torch.empty_like(x)
            y[perm] = x
            outs.append(y)
        return outs
    scatter_ms, back = timed(scatter)
    same = all(torch.equal(a, b) for a, b in zip(back, ref))
    net = key_ms + sort_ms + gather_ms + trace_sorted_ms + scatter_ms
    print(json.dumps({"config": name, "query": query, "rays": n, "tris": int(len(f)), "cell_bits_per_axis": bits,
                      "trace_ms": round(trace_ms, 3), "trace_sorted_ms": round(trace_sorted_ms, 3), "key_ms": round(key_ms, 3),
                      "sort_ms": round(sort_ms, 3), "gather_ms": round(gather_ms, 3), "scatter_ms": round(scatter_ms, 3),
                      "net_ms": round(net, 3), "net_vs_unsorted": round(net / trace_ms, 3),
                      "min_pass_bytes": int(n * (24 + 8 + 8 + 24 + 26 + 26)),   # read rays, write+read key/idx, write sorted rays, read+write results
                      "results_identical_after_scatter": bool(same)}), flush=True)


vb, fb = W.bunny_standin()
vh, fh = W.headline_mesh(8)
if len(sys.argv) > 1 and sys.argv[1] == "stream":
    # round 3: the data point VERDICT r02 asked for -- the SORTED rays through the STREAMING launch (what an
    # index-only binning front end could hand it at best: the sort itself is not charged to trace_sorted_ms)
    from triro.backend import ops as hops
    for st in (1, 0):          # 1: the product's choice (streaming launch for these batches), 0: direct launch
        hops.set_option("stream", st)
        for bits in (3, 4):
            run(f"C3 any [stream={st}]", vb, fb, 10_000_000, 1234, "any", bits)
            run(f"C3' closest [stream={st}]", vb, fb, 10_000_000, 1234, "closest", bits)
            run(f"C5(ii) 12.5M-ray shard [stream={st}]", vh, fh, 12_500_000, 99, "closest", bits)
    sys.exit(0)
for bits in (2, 3, 4, 5):
    run("C3 (10M hash rays vs bunny stand-in)", vb, fb, 10_000_000, 1234, "any", bits)
    run("C3' closest", vb, fb, 10_000_000, 1234, "closest", bits)
    run("C5(ii) one 12.5M-ray shard vs headline mesh", vh, fh, 12_500_000, 99, "closest", bits)
