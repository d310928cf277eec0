#!/usr/bin/env python3
"""update_raw (rebuild) + one closest query per frame on the headline config: what does the query cost
right after a rebuild?  (The learned launch order survives the rebuild as a one-frame-stale hint.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import workloads as W  # noqa: E402
from triro.backend import ops as hops  # noqa: E402
from triro.ray.ray_optix import RayMeshIntersector  # noqa: E402

dev = torch.device("cuda:0")
for kv in sys.argv[1:]:
    k, v_ = kv.split("=", 1)
    hops.set_option(k, int(v_))
v, f = W.headline_mesh(8)
vt, ft = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
frames = [vt * (1.0 + 0.002 * k) for k in range(8)]        # a breathing mesh
r = RayMeshIntersector(vertices=vt, faces=ft)
rad = float(np.linalg.norm(v, axis=1).max())
on, dn = W.pinhole_grid(1024, 1024, distance=2.5 * rad)
o, d = torch.from_numpy(on).to(dev), torch.from_numpy(dn).to(dev)
seq = list(range(8)) + list(range(6, 0, -1))
for k in range(10):
    r.update_raw(frames[seq[k % len(seq)]], ft)
    r.intersects_closest(o, d)
torch.cuda.synchronize()
n = 60
evq = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
evb = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
for k in range(n):
    evb[k].record()
    r.update_raw(frames[seq[k % len(seq)]], ft)
    evq[k][0].record()
    r.intersects_closest(o, d)
    evq[k][1].record()
torch.cuda.synchronize()
q = [a.elapsed_time(b) for a, b in evq]
b = [evb[k].elapsed_time(evq[k][0]) for k in range(n)]
print(json.dumps({"opts": sys.argv[1:], "query_ms_mean": round(float(np.mean(q)), 4), "query_ms_min": round(min(q), 4),
                  "rebuild_ms_mean": round(float(np.mean(b)), 4)}))
