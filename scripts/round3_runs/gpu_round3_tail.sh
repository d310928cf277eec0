#!/bin/bash
# the launch's last blocks (cheapest of the learned order) halved: option split_tail = blocks per XCD
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "trimesh-ray-optix_amd"), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch
import workloads as W
from oracle.oracle import OracleIntersector
from triro.ray.ray_optix import RayMeshIntersector
from triro.backend import ops as hops
dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
for tail in (32, 200):
    hops.set_option("split_tail", tail)
    for (v, f), res in ((W.headline_mesh(6), 512), (W.nested_shells(5), 384)):
        r = RayMeshIntersector(vertices=T(v), faces=T(f))
        rad = float(np.linalg.norm(v, axis=1).max())
        o, d = W.pinhole_grid(res, res, distance=2.5 * rad)
        exp = OracleIntersector(v, f, 1).closest_raw(o.reshape(-1, 3), d.reshape(-1, 3))
        ot, dt = T(o), T(d)
        for k in range(12):
            got = [g.cpu().numpy().reshape(e.shape) for g, e in zip(r.intersects_closest(ot, dt), exp)]
            assert all(np.array_equal(g, e) for g, e in zip(got, exp)), (tail, k)
        print("parity ok", tail, len(f), r.as_wrapper.last_launch())
PY
for TL in 0 64 128 256 0 64 128 256; do
  for A in "--config c5i --query closest --steps 100 --warmup 40" "--config c5i --query closest --res 512 --steps 100 --warmup 40" "--config c5i --query closest --res 2048 --steps 40 --warmup 30" "--config c4 --query closest --steps 100 --warmup 40" "--config c2 --query closest --steps 100 --warmup 40" "--config room --query closest --steps 100 --warmup 40" "--config c5i --query any --steps 60 --warmup 30"; do
    timeout 90 python scripts/run_query.py $A --opt split_tail=$TL 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('tail=$TL', r['config'], r['query'], r['rays'], r['ms_mean'], r['ms_min'])"
  done
done
