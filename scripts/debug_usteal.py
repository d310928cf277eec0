import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np, torch
import workloads as W
from triro.backend import ops as hops
from triro.ray.ray_optix import RayMeshIntersector
dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
hops.set_option("adaptive", 0)
v, f = W.bunny_standin()
r = RayMeshIntersector(vertices=T(v), faces=T(f))
o, d = W.pinhole_grid(256, 256, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
ot, dt = T(o), T(d)
hops.set_option("usteal", 0)
ref = r.intersects_count(ot, dt).clone()
prev = [0, 0, 0, 0]
for us in (4095, 1000, 200, 64, 16):
    hops.set_option("usteal", us)
    got = r.intersects_count(ot, dt)
    torch.cuda.synchronize()
    dbg = (ctypes.c_uint * 4)()
    hops.get_module().tr_debug_usteal(dbg)
    cur = list(dbg)
    print("usteal", us, "debug counters (waves capped, lanes unfinished, handovers, -):", [a - b for a, b in zip(cur, prev)],
          "equal:", bool(torch.equal(ref, got)), "diff rays:", int((ref != got).sum()), "sum ref/got", int(ref.sum()), int(got.sum()))
    prev = cur
