#!/usr/bin/env python3
"""Host experiment (no GPU): node visits of the closest-hit traversal on the headline batch for different
hierarchies over the same Morton-sorted leaves -- the Karras radix tree the builder emits (split at the
highest differing Morton bit), a count-balanced tree (split at the middle index) and a hybrid.  Uses the
product's own per-lane traversal code through tests/host_sim.  Prints mean visits per ray and the mean
over 64-ray waves (consecutive pixels of a row) of the slowest ray -- the wave-trip count the GPU pays.
usage: python scripts/exp_tree_shape.py [--res 512] [--subdiv 8]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "host_sim")]
import numpy as np
import workloads as W
from sim import SimBVH
ap = argparse.ArgumentParser(); ap.add_argument("--res", type=int, default=512); ap.add_argument("--subdiv", type=int, default=8)
a = ap.parse_args()
v, f = W.headline_mesh(a.subdiv)
rad = float(np.linalg.norm(v, axis=1).max())
o, d = W.pinhole_grid(a.res, a.res, distance=2.5 * rad)
o = np.ascontiguousarray(o).reshape(-1, 3); d = d.reshape(-1, 3)
for name, mode in (("karras (shipped)", -1), ("balanced (middle index)", 2), ("hybrid (karras unless lopsided > 7:1)", 3)):
    B = SimBVH(v, f, force_mode=mode)
    nv, tt = B.steps(o, d)
    waves = nv.reshape(-1, 64)
    print(json.dumps({"tree": name, "depth": int(B.depth), "rays": int(len(o)), "node_visits_mean": round(float(nv.mean()), 2),
                      "node_visits_max": int(nv.max()), "tri_tests_mean": round(float(tt.mean()), 2),
                      "wave_max_mean": round(float(waves.max(1).mean()), 2), "lane_utilisation": round(float(nv.mean() / waves.max(1).mean()), 3)}), flush=True)
