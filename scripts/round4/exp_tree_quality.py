#!/usr/bin/env python3
"""Host experiment (no GPU): how much would a better TREE buy?  Node visits of the closest-hit and of the count
traversal on the headline image (and C4) for hierarchies over the same Morton-sorted leaves: the Karras radix tree the
builder emits, and a tree whose every split minimises the surface-area heuristic over all positions of the
Morton-ordered range (a full SAH sweep: the best a builder that keeps the leaf order can do).
usage: python scripts/round4/exp_tree_quality.py [--res 512]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "host_sim")]
import numpy as np
import workloads as W
from sim import SimBVH
ap = argparse.ArgumentParser(); ap.add_argument("--res", type=int, default=512)
a = ap.parse_args()
for cfg, (v, f), dist in (("C5(i) headline", W.headline_mesh(8), None), ("C4 shells", W.nested_shells(7), 2.5), ("terrain", W.terrain(), "terrain")):
    if dist == "terrain":
        w, h = 512, 288
        _, d = W.ref_shape_rays(W.TERRAIN_EYE, W.TERRAIN_TARGET, w, h, 444.0 * w / 640)
        o = np.broadcast_to(np.asarray(W.TERRAIN_EYE, np.float32), d.shape)
    else:
        dd = dist if dist is not None else 2.5 * float(np.linalg.norm(v, axis=1).max())
        o, d = W.pinhole_grid(a.res, a.res, distance=dd)
    # 8x8 tile order: what a wave of the direct launch holds
    H, Wd = d.shape[0] // 8 * 8, d.shape[1] // 8 * 8
    o = np.ascontiguousarray(np.broadcast_to(o, d.shape)[:H, :Wd]).reshape(H // 8, 8, Wd // 8, 8, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 3)
    d = np.ascontiguousarray(d[:H, :Wd]).reshape(H // 8, 8, Wd // 8, 8, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 3)
    for name, mode in (("karras (shipped)", -1), ("SAH sweep over the Morton order", 4)):
        B = SimBVH(v, f, force_mode=mode)
        nv, tt = B.steps(o, d)
        waves = nv.reshape(-1, 64)
        st = B.packet_stats(o, d, 64)
        print(json.dumps({"config": cfg, "tree": name, "depth": int(B.depth), "rays": int(len(o)),
                          "closest_node_visits_mean": round(float(nv.mean()), 2), "closest_tri_tests_mean": round(float(tt.mean()), 2),
                          "closest_slowest_ray_per_wave_mean": round(float(waves.max(1).mean()), 2),
                          "count_node_visits_mean": round(float(st[:, 0].sum() / len(o)), 2),
                          "count_slowest_ray_per_wave_mean": round(float(st[:, 1].mean()), 2)}), flush=True)
