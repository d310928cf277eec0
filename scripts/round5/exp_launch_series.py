#!/usr/bin/env python3
"""Per-launch kernel time of launches 1..N of the headline batch on FRESH handles (median over handles): how fast the
learned launch order converges, with the options given as name=value.  One JSON line per option set."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np, torch
import workloads as W
import triro.backend.ops as hops
from triro.ray.ray_optix import RayMeshIntersector
dev = torch.device("cuda:0")
N = int(os.environ.get("SERIES_N", "40")); H = int(os.environ.get("SERIES_HANDLES", "6"))
v, f = W.headline_mesh(8)
vt, ft = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
rad = float(np.linalg.norm(v, axis=1).max())
o_np, d_np = W.pinhole_grid(1024, 1024, distance=2.5 * rad)
O, D = torch.from_numpy(np.ascontiguousarray(o_np)).to(dev), torch.from_numpy(d_np).to(dev)
warm = RayMeshIntersector(vertices=vt, faces=ft)
for _ in range(3):
    warm.intersects_closest(O, D)
del warm
for spec in (sys.argv[1:] or [""]):
    opts = dict(kv.split("=") for kv in spec.split(",") if kv)
    for k, val in opts.items():
        hops.set_option(k, int(val))
    series = np.zeros((H, N))
    for h in range(H):
        r = RayMeshIntersector(vertices=vt, faces=ft)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
        ev[0].record()
        for k in range(N):
            r.intersects_closest(O, D)
            ev[k + 1].record()
        torch.cuda.synchronize()
        series[h] = [ev[k].elapsed_time(ev[k + 1]) for k in range(N)]
    med = np.median(series, axis=0)
    print(json.dumps({"opts": opts, "launch_ms_median_over_handles": [round(float(x), 4) for x in med],
                      "sum_first_5_ms": round(float(med[:5].sum()), 4), "mean_6_25_ms": round(float(med[5:25].mean()), 4),
                      "mean_last_10_ms": round(float(med[-10:].mean()), 4)}), flush=True)
    for k in opts:
        hops.set_option(k, 1)
