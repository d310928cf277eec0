#!/usr/bin/env python3
"""EXPERIMENT: refit in one launch (k_refit_up, TRIRO_REFIT_SINGLE=1) against the level-by-level pass.  Run twice (with and
without the variable); each run prints a digest of the node arrays after every refit of a sequence of deformations and
the refit time: the digests of the two runs must be equal.  usage: exp_refit_single.py <mesh: c5i|c4|c2|deep> [reps]"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np, torch
import workloads as W
from triro.ray.ray_optix import RayMeshIntersector
dev = torch.device("cuda:0")
name = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
v, f = {"c5i": lambda: W.headline_mesh(8), "c4": W.nested_shells, "c2": W.bunny_mesh, "deep": lambda: W.deep_tree_mesh(3000)}[name]()
v = v.astype(np.float32)
r = RayMeshIntersector(vertices=torch.from_numpy(v).to(dev), faces=torch.from_numpy(f).to(dev))
digests, ms = [], []
for k in range(reps):
    vk = (v * np.float32(1.0 + 0.01 * np.sin(0.7 * k)) + np.float32(0.003 * k) * np.sin(v[:, ::-1] * np.float32(3 + k))).astype(np.float32)
    vt = torch.from_numpy(vk).to(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r.refit(vt)
    torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    nodes, links, tris = r.as_wrapper.download()
    qn, frame = r.as_wrapper.download_qnodes()
    h = hashlib.sha256(); h.update(np.ascontiguousarray(nodes).tobytes()); h.update(np.ascontiguousarray(qn).tobytes()); h.update(np.ascontiguousarray(frame).tobytes())
    digests.append(h.hexdigest()[:16])
print(json.dumps({"mesh": name, "single": os.environ.get("TRIRO_REFIT_SINGLE") is not None, "tris": int(len(f)), "depth": r.bvh_info()["depth"],
                  "refit_ms_median": round(float(np.median(ms[2:])), 4), "refit_ms_min": round(float(np.min(ms)), 4), "digests": digests}))
