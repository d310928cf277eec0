#!/bin/bash
OUT=gpurun_out/r04_run12
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_round4.py -x -q -p no:cacheprovider -k "slot_form or expansion" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
for cus in 0 8 64; do
  echo "== expand_cus=$cus" >> $OUT/micro.txt
  timeout 300 python - $cus >> $OUT/micro.txt 2>&1 <<'PY'
import os, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "trimesh-ray-optix_amd")]
import numpy as np, torch
import workloads as W
import triro.backend.ops as hops
from triro.ray.ray_optix import RayMeshIntersector
dev = torch.device("cuda:0")
hops.set_option("expand_cus", int(sys.argv[1]))
v, f = W.headline_mesh(8)
r = RayMeshIntersector(vertices=torch.from_numpy(v).to(dev), faces=torch.from_numpy(f).to(dev))
o_np, d_np = W.pinhole_grid(1024, 1024, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
o = torch.from_numpy(np.ascontiguousarray(o_np)).to(dev)
recs = torch.cat([r.intersects_closest_packed(o, torch.from_numpy(np.roll(d_np, k * 7 + 7, axis=0)).to(dev), slots=True) for k in range(7)])
miss = torch.full_like(recs, -1)
n = recs.shape[0]
outs = (torch.empty(n, dtype=torch.bool, device=dev), torch.empty(n, dtype=torch.bool, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.empty((n, 3), device=dev), torch.empty((n, 2), device=dev))
for name, rr in (("image", recs), ("all-miss", miss)):
    for rl in (1024, 0):
        for _ in range(5): r.closest_expand(rr, outs=outs, slots=True, row_length=rl)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): r.closest_expand(rr, outs=outs, slots=True, row_length=rl)
        torch.cuda.synchronize()
        print(name, "row_length", rl, round((time.perf_counter() - t0) / 30 * 1e3, 4), "ms")
PY
done
cat $OUT/micro.txt | grep -v amdgpu.ids
E="python bench.py --steps 300 --warmup 30"
for cus in 0; do
  timeout 600 $E --emulate-world 8 --arrival none --opt expand_cus=$cus >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
done
cat $OUT/emulate.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); e = r['emulation']; c = r['config']
    print(c['workload'][:22], 'N', r['emulated_world'], c.get('opts'), 'arr', c['arrival'], '| plain', e['plain_1gpu_ms_per_step'], 'rank0', e['rank0_ms_per_step'], 'expand', e['expansion_alone_ms'], e['expansion_GBps'], 'GB/s | implied', e['implied_scaling_vs_1gpu'], 'ok', r['verified'])
"
