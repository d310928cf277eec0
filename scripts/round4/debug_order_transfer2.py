import ctypes as C, json, os, sys
ROOT = os.getcwd()
os.environ["TRIRO_HIP_LIBRARY"] = os.path.join(ROOT, "trimesh-ray-optix_amd/lib_var/dbgsched/libtriro_hip.so")
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np, torch
import workloads as W
import triro.backend.ops as hops
from triro.ray.ray_optix import RayMeshIntersector
dev = torch.device("cuda:0")
MAX = 131072
v, f = W.headline_mesh(8)
vt, ft = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
rad = float(np.linalg.norm(v, axis=1).max())
def rays(res):
    o_np, d_np = W.pinhole_grid(res, res, distance=2.5 * rad)
    return torch.from_numpy(np.ascontiguousarray(o_np)).to(dev), torch.from_numpy(d_np).to(dev)
lib = hops.get_module()
lib.tr_debug_sched.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
def dump(r):
    buf = np.zeros(3 * MAX + 12, np.uint32)
    rc = lib.tr_debug_sched(r.as_wrapper._inner, torch.cuda.current_stream(dev).cuda_stream, 1, buf.ctypes.data, len(buf))
    assert rc == 0, rc
    return buf
o7, d7 = rays(768); o10, d10 = rays(1024)
def order_blocks(buf, nslots):
    e = buf[MAX:MAX + nslots]
    return e & 0x07ffffff, e >> 30, (e >> 28) & 3
res = {}
for transfer in (0, 1):
    hops.set_option("order_transfer", transfer)
    r = RayMeshIntersector(vertices=vt, faces=ft)
    for _ in range(14): r.intersects_closest(o7, d7)
    b7 = dump(r)
    r.intersects_closest(o10, d10)
    b1 = dump(r)          # after the first 1024 launch: order = sorted from ITS measured costs; prev = those costs
    for _ in range(30): r.intersects_closest(o10, d10)
    b2 = dump(r)
    res[transfer] = (b7, b1, b2)
    prev1, prev2 = b1[2 * MAX + 12:2 * MAX + 12 + 8192].astype(np.float64), b2[2 * MAX + 12:2 * MAX + 12 + 8192].astype(np.float64)
    print("transfer", transfer, "stamp", b1[2 * MAX:2 * MAX + 2], b2[2 * MAX:2 * MAX + 2], "prev1 mean/max", prev1.mean(), prev1.max(), "prev2 mean/max", prev2.mean(), prev2.max(),
          "corr(prev1, prev2)", round(float(np.corrcoef(prev1, prev2)[0, 1]), 3), flush=True)
    blk, lg, part = order_blocks(b2, 8960)
    print("   steady split entries:", int((lg > 0).sum()), "sentinels:", int((blk == 0x07ffffff).sum()), flush=True)
    blk1, lg1, _ = order_blocks(b1, 8960)
    print("   after launch 1 split entries:", int((lg1 > 0).sum()), "sentinels:", int((blk1 == 0x07ffffff).sum()), flush=True)
p0 = res[0][2][2 * MAX + 12:2 * MAX + 12 + 8192].astype(np.float64)
p1 = res[1][2][2 * MAX + 12:2 * MAX + 12 + 8192].astype(np.float64)
print("corr of steady costs transfer 0 vs 1:", round(float(np.corrcoef(p0, p1)[0, 1]), 3))
s0 = set(order_blocks(res[0][2], 8960)[0][order_blocks(res[0][2], 8960)[1] > 0].tolist())
s1 = set(order_blocks(res[1][2], 8960)[0][order_blocks(res[1][2], 8960)[1] > 0].tolist())
print("split sets: |0|", len(s0), "|1|", len(s1), "common", len(s0 & s1))
