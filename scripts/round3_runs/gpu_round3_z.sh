#!/bin/bash
B="python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29735 bench.py --gpus 1 --no-cpu-baseline --no-companions --steps 400"
for A in "" "--force-gather"; do
  timeout 300 $B $A 2>/dev/null | python3 -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{\"metric\"'):
        r=json.loads(ln); print('[$A]', r['value'], 'ms/step', r['ms_per_step'], 'caller-stream ms', r['roofline']['kernel_avg_ms'], 'min', r['roofline']['kernel_min_ms'])"
done
timeout 200 python - <<'PY'
import os, sys, time
sys.path[:0]=[os.environ['GRAFT_REPO_ROOT'], os.path.join(os.environ['GRAFT_REPO_ROOT'],'trimesh-ray-optix_amd')]
import numpy as np, torch, workloads as W
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR','127.0.0.1'); os.environ.setdefault('MASTER_PORT','29736')
dev=torch.device('cuda:0'); torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from triro.ray.ray_optix import RayMeshIntersector
from triro.ray.sharded import ShardedRayMeshIntersector
v,f=W.headline_mesh(8); r=RayMeshIntersector(vertices=torch.from_numpy(v).to(dev), faces=torch.from_numpy(f).to(dev))
o,d=W.pinhole_grid(1024,1024,distance=2.5*float(np.linalg.norm(v,axis=1).max()))
o=torch.from_numpy(np.ascontiguousarray(o)).to(dev); d=torch.from_numpy(d).to(dev)
S=ShardedRayMeshIntersector(r, force_collectives=True)
for _ in range(30): S.closest_of_shard_async(o,d,1<<20,batch_shape=(1024,1024),dst=0).wait()
torch.cuda.synchronize()
# host time of issuing one step (no waiting for the GPU)
t0=time.perf_counter(); hs=[S.closest_of_shard_async(o,d,1<<20,batch_shape=(1024,1024),dst=0) for _ in range(50)]; t1=time.perf_counter()
for h in hs: h.wait()
torch.cuda.synchronize(); t2=time.perf_counter()
print('pipeline: host issue per step %.1f us, total per step %.1f us' % ((t1-t0)/50*1e6, (t2-t0)/50*1e6))
t0=time.perf_counter(); outs=[r.intersects_closest(o,d) for _ in range(50)]; t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print('plain: host issue per step %.1f us, total per step %.1f us' % ((t1-t0)/50*1e6, (t2-t0)/50*1e6))
dist.destroy_process_group()
PY
