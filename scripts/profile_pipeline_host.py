import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np, torch, workloads as W
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29737')
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from triro.ray.ray_optix import RayMeshIntersector
from triro.ray.sharded import ShardedRayMeshIntersector
v, f = W.headline_mesh(6); r = RayMeshIntersector(vertices=torch.from_numpy(v).to(dev), faces=torch.from_numpy(f).to(dev))
o, d = W.pinhole_grid(1024, 1024, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
o = torch.from_numpy(np.ascontiguousarray(o)).to(dev); d = torch.from_numpy(d).to(dev)
S = ShardedRayMeshIntersector(r, force_collectives=True)
for _ in range(30): S.closest_of_shard_async(o, d, 1 << 20, batch_shape=(1024, 1024), dst=0).wait()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
hs = [S.closest_of_shard_async(o, d, 1 << 20, batch_shape=(1024, 1024), dst=0) for _ in range(200)]
pr.disable()
for h in hs: h.wait()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
dist.destroy_process_group()
