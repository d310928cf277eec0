#!/usr/bin/env python3
"""order_transfer = 0 / 1 / 2: launches 1 ... 6 after a change of resolution and on a fresh handle,
and the steady state they lead to (median of launches 40 ... 60), on the headline mesh, the terrain, the shells and
the interior room -- the scenes on which an earlier version of the direct transfer left a worse split set for good.
Kernel-to-kernel HIP events around each call.  One JSON line per (scene, mode)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np, torch
import workloads as W
import triro.backend.ops as hops
from triro.ray.ray_optix import RayMeshIntersector
dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def scene(name):
    if name == "headline":
        v, f = W.headline_mesh(8)
        rad = float(np.linalg.norm(v, axis=1).max())
        mk = lambda w, h: W.pinhole_grid(w, h, distance=2.5 * rad)  # noqa: E731
        sizes = ((768, 768), (1024, 1024))
    elif name == "shells":
        v, f = W.nested_shells(7)
        mk = lambda w, h: W.pinhole_grid(w, h, distance=2.5)  # noqa: E731
        sizes = ((768, 768), (1024, 1024))
    elif name == "terrain":
        v, f = W.terrain()
        mk = lambda w, h: (np.broadcast_to(np.array(W.TERRAIN_EYE, np.float32), (h, w, 3)),  # noqa: E731
                           W.ref_shape_rays(W.TERRAIN_EYE, W.TERRAIN_TARGET, w, h, 444.0 * w / 640)[1])
        sizes = ((768, 432), (1024, 576))
    else:
        v, f = W.interior_room()
        mk = lambda w, h: (np.broadcast_to(np.array(W.INTERIOR_EYE, np.float32), (h, w, 3)),  # noqa: E731
                           W.ref_shape_rays(W.INTERIOR_EYE, W.INTERIOR_TARGET, w, h, 444.0 * w / 640)[1])
        sizes = ((960, 544), (1280, 720))
    return v, f, mk, sizes


for name in sys.argv[1:] or ["headline", "shells", "terrain", "room"]:
    v, f, mk, (small, big) = scene(name)
    vt, ft = T(v), T(f)
    rs = {}
    for key, (w, h) in (("small", small), ("big", big)):
        o, d = mk(w, h)
        rs[key] = (T(o) if o.strides[0] else torch.from_numpy(np.ascontiguousarray(o[:1, :1])).to(dev).expand(h, w, 3), T(d))
    warm = RayMeshIntersector(vertices=vt, faces=ft)
    for k in rs:
        for _ in range(3):
            warm.intersects_closest(*rs[k])
    del warm
    for mode in (0, 1, 2):
        hops.set_option("order_transfer", mode)
        series, fresh, steady = [], [], []
        for rep in range(6):
            r = RayMeshIntersector(vertices=vt, faces=ft)
            fr = [timed(lambda: r.intersects_closest(*rs["small"])) for _ in range(6)]
            for _ in range(8):
                r.intersects_closest(*rs["small"])
            se = [timed(lambda: r.intersects_closest(*rs["big"])) for _ in range(6)]
            for _ in range(34):
                r.intersects_closest(*rs["big"])
            st = [timed(lambda: r.intersects_closest(*rs["big"])) for _ in range(20)]
            li = r.as_wrapper.last_launch()
            series.append(se); fresh.append(fr); steady.append(float(np.median(st)))
            del r
        med = lambda xs: [round(float(np.median([x[k] for x in xs])), 4) for k in range(len(xs[0]))]  # noqa: E731
        print(json.dumps({"scene": name, "tris": int(len(f)), "order_transfer": mode, "small": small, "big": big,
                          "fresh_handle_small_launches_1_6_ms": med(fresh), "big_after_14_small_launches_1_6_ms": med(series),
                          "steady_big_ms": round(float(np.median(steady)), 4), "steady_big_all": [round(x, 4) for x in steady],
                          "split_blocks": int(li["split_blocks"]), "tile_rows_lg": int(li["tile_rows_lg"])}), flush=True)
    hops.set_option("order_transfer", 1)
