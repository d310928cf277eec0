#!/bin/bash
OUT=gpurun_out/r04_run10
mkdir -p $OUT
REPO=$PWD
for seed in 41 42 43; do
  timeout 1500 python scripts/fuzz_parity.py --iters 200 --seed $seed > $OUT/fuzz_seed$seed.txt 2>&1
  tail -2 $OUT/fuzz_seed$seed.txt
done
# PMC of the expansion in 8x8 tiles (slot form) and linear
cat > /tmp/exp_prof.py <<'PY'
import os, sys, time
ROOT = os.environ["REPO"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np, torch
import workloads as W
from triro.ray.ray_optix import RayMeshIntersector
dev = torch.device("cuda:0")
v, f = W.headline_mesh(8)
r = RayMeshIntersector(vertices=torch.from_numpy(v).to(dev), faces=torch.from_numpy(f).to(dev))
o_np, d_np = W.pinhole_grid(1024, 1024, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
o = torch.from_numpy(np.ascontiguousarray(o_np)).to(dev)
recs = torch.cat([r.intersects_closest_packed(o, torch.from_numpy(np.roll(d_np, k * 7 + 7, axis=0)).to(dev), slots=True) for k in range(7)])
n = recs.shape[0]
outs = (torch.empty(n, dtype=torch.bool, device=dev), torch.empty(n, dtype=torch.bool, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.empty((n, 3), device=dev), torch.empty((n, 2), device=dev))
rl = int(sys.argv[1])
for _ in range(8):
    r.closest_expand(recs, outs=outs, slots=True, row_length=rl)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp REPO
for rl in 1024 0; do
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum"; do
    tag=$(echo $pmc | tr ' ' '_' | cut -c1-30)
    timeout 300 rocprofv3 --pmc $pmc --output-format csv -d $REPO/$OUT/pmc_rl${rl}_$tag -- python3 /tmp/exp_prof.py $rl > $REPO/$OUT/pmc_rl${rl}_$tag.log 2>&1
  done
done
cd $REPO
python3 - <<'PY'
import glob, csv, collections
for rl in (1024, 0):
    d = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r04_run10/pmc_rl{rl}_*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "expand" in r["Kernel_Name"]:
                d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("row_length", rl, {k: round(sum(v) / len(v), 1) for k, v in sorted(d.items())})
PY
