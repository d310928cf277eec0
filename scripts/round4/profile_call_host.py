import cProfile, pstats, io, os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np, torch
import workloads as W
from triro.ray.ray_optix import RayMeshIntersector
dev = torch.device("cuda:0")
v, f = W.icosphere(3)
r = RayMeshIntersector(vertices=torch.from_numpy(v.astype(np.float32)).to(dev), faces=torch.from_numpy(f.astype(np.int32)).to(dev))
o_np, d_np = W.pinhole_grid(16, 16)
o, d = torch.from_numpy(np.ascontiguousarray(o_np)).to(dev), torch.from_numpy(d_np).to(dev)
for _ in range(200): r.intersects_closest(o, d)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5000): r.intersects_closest(o, d)
torch.cuda.synchronize()
print("us per call (async loop):", (time.perf_counter() - t0) / 5000 * 1e6)
for name, fn in (("any", r.intersects_any), ("first", r.intersects_first), ("count", r.intersects_count)):
    t0 = time.perf_counter()
    for _ in range(5000): fn(o, d)
    torch.cuda.synchronize()
    print(name, "us per call:", (time.perf_counter() - t0) / 5000 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(5000): r.intersects_closest(o, d)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3800])
