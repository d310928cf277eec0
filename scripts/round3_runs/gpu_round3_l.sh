#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3l_bench.json 2> gpurun_out/r3l_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r3l_bench.json').read().strip().splitlines()[-1])
rl=r['roofline']
print(r['value'], rl['kernel_avg_ms'], r['verified'], 'cold', rl['cold_kernel_ms'], 'first', rl.get('first_launch_kernel_ms'), 'moving', rl['moving_camera_kernel_ms'])
PY
Q="timeout 90 python scripts/run_query.py --steps 1 --warmup 0 --each"
for CFG in "c5i" "c4" "c2" "room" "c5i --res 512"; do
 for O in "--opt prepass=1" "--opt prepass=0"; do
  timeout 120 python - $CFG $O <<'PY'
import sys, os, json
ROOT=os.environ.get('GRAFT_REPO_ROOT','.')
sys.path[:0]=[ROOT, os.path.join(ROOT,'trimesh-ray-optix_amd')]
import numpy as np, torch, workloads as W
from triro.backend import ops as hops
from triro.ray.ray_optix import RayMeshIntersector
a=sys.argv[1:]
cfg=a[0]; res=1024
if '--res' in a: res=int(a[a.index('--res')+1])
for i,x in enumerate(a):
    if x=='--opt': k,v=a[i+1].split('='); hops.set_option(k,int(v))
dev=torch.device('cuda:0'); T=lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
if cfg=='c2': v,f=W.bunny_standin()
elif cfg=='c4': v,f=W.nested_shells(7)
elif cfg=='room': v,f=W.interior_room()
else: v,f=W.headline_mesh(8)
rad=float(np.linalg.norm(v,axis=1).max())
if cfg=='room':
    _,dn=W.ref_shape_rays(W.INTERIOR_EYE,W.INTERIOR_TARGET); o=torch.from_numpy(np.array(W.INTERIOR_EYE,np.float32)).to(dev).expand(360,640,3); d=T(dn)
else:
    on,dn=W.pinhole_grid(res,res,distance=2.5 if cfg=='c4' else 2.5*rad); o,d=T(on),T(dn)
vt,ft=T(v),T(f)
out={}
for q in ('closest','any','count'):
    ms=[]
    for _ in range(6):
        rr=RayMeshIntersector(vertices=vt,faces=ft)
        fn={'closest':rr.intersects_closest,'any':rr.intersects_any,'count':rr.intersects_count}[q]
        fn(o[:1,:64].contiguous(), d[:1,:64].contiguous()); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(o,d); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1)); del rr
    out[q]=round(float(np.median(ms)),4)
print(json.dumps({'config':cfg,'res':res,'opts':[x for x in a if '=' in x],'first_launch_ms':out}))
PY
 done
done 2>&1 | grep -v amdgpu | tee gpurun_out/r3l_first_launch.jsonl
