import json, os, sys
ROOT = os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np, torch
import workloads as W
import triro.backend.ops as hops
from triro.ray.ray_optix import RayMeshIntersector
dev = torch.device("cuda:0")
v, f = W.headline_mesh(8)
vt, ft = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
rad = float(np.linalg.norm(v, axis=1).max())
def rays(res):
    o_np, d_np = W.pinhole_grid(res, res, distance=2.5 * rad)
    return torch.from_numpy(np.ascontiguousarray(o_np)).to(dev), torch.from_numpy(d_np).to(dev)
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1), 4)
o7, d7 = rays(768); o10, d10 = rays(1024)
warm = RayMeshIntersector(vertices=vt, faces=ft)
for oo, dd in ((o7, d7), (o10, d10)):
    for _ in range(3): warm.intersects_closest(oo, dd)
del warm
DEF = {"order_transfer": 1, "tile": 1, "split": 1, "tile_small": 4, "grid_nodes": 1}
for name, opts in (("A transfer=0", {"order_transfer": 0}), ("B transfer=1", {"order_transfer": 1}),
                   ("C transfer=0 tile=0 (rows; 8x8 tiles never)", {"order_transfer": 0, "tile": 0}),
                   ("D transfer=1 tile=0 split=0 (order only)", {"order_transfer": 1, "tile": 0, "split": 0}),
                   ("E transfer=1 grid_nodes=2", {"order_transfer": 1, "grid_nodes": 2}),
                   ("F transfer=0 grid_nodes=2", {"order_transfer": 0, "grid_nodes": 2}),
                   ("A2 transfer=0", {"order_transfer": 0}), ("B2 transfer=1", {"order_transfer": 1})):
    for k, val in DEF.items(): hops.set_option(k, val)
    for k, val in opts.items(): hops.set_option(k, val)
    firsts, seqs = [], None
    for rep in range(4):
        r = RayMeshIntersector(vertices=vt, faces=ft)
        for _ in range(14): r.intersects_closest(o7, d7)
        t10 = [timed(lambda: r.intersects_closest(o10, d10)) for _ in range(30)]
        firsts.append(t10[0]); seqs = t10
        del r
    print(name, "| first", firsts, "| last rep:", seqs[:6], "... mean of 20-30:", round(float(np.mean(seqs[20:])), 4), flush=True)
for k, val in DEF.items(): hops.set_option(k, val)
