#!/usr/bin/env python3
"""Why does a moving camera cost 9 % (bench.py: moving_camera_kernel_ms 0.197 against 0.181)?  Three loops on the headline
mesh, event-timed: (a) the same ray tensors every launch, (b) EIGHT COPIES of the same rays round-robin (other addresses,
same image: the learned order fits exactly), (c) eight frames of a camera orbiting 0.25 degrees per step, ping-pong
(bench.py's loop).  One JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np, torch
import workloads as W
from triro.ray.ray_optix import RayMeshIntersector
dev = torch.device("cuda:0")
from triro.backend import ops as hops
OPTS = [a for a in sys.argv[1:] if "=" in a]
for a in OPTS:
    k, val = a.split("="); hops.set_option(k, int(val))
v, f = W.headline_mesh(8)
r = RayMeshIntersector(vertices=torch.from_numpy(v).to(dev), faces=torch.from_numpy(f).to(dev))
rad = float(np.linalg.norm(v, axis=1).max())
o_np, d_np = W.pinhole_grid(1024, 1024, distance=2.5 * rad)
o_np = np.ascontiguousarray(o_np)


def rot(a, deg):
    c, s = np.cos(np.radians(deg)), np.sin(np.radians(deg))
    m = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
    return (a.reshape(-1, 3) @ m.T).reshape(a.shape).astype(np.float32)


def loop(frames, seq, n=280, warm=56):
    for k in range(warm):
        r.intersects_closest(*frames[seq[k % len(seq)]])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(n):
        r.intersects_closest(*frames[seq[k % len(seq)]])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


T = lambda a: torch.from_numpy(a).to(dev)
same = [(T(o_np), T(d_np))]
copies = [(T(o_np.copy()), T(d_np.copy())) for _ in range(8)]
copies_b = [(T(o_np[:1, :1]).expand(1024, 1024, 3), T(d_np.copy())) for _ in range(8)]      # stride-0 origin, as workloads hands it out
moving = [(T(rot(o_np, 0.25 * k)), T(rot(d_np, 0.25 * k))) for k in range(8)]
pp = list(range(8)) + list(range(6, 0, -1))
res = {}
for rep in range(2):
    res.setdefault("same_tensors_ms", []).append(round(loop(same, [0]), 4))
    res.setdefault("eight_copies_round_robin_ms", []).append(round(loop(copies, list(range(8))), 4))
    res.setdefault("eight_copies_stride0_origin_ms", []).append(round(loop(copies_b, list(range(8))), 4))
    res.setdefault("moving_camera_ping_pong_ms", []).append(round(loop(moving, pp), 4))
    res.setdefault("moving_camera_frame3_only_ms", []).append(round(loop(moving, [3]), 4))
res["opts"] = OPTS
print(json.dumps(res))
