#!/usr/bin/env python3
"""Host experiment (no GPU): what would 2-triangle leaves buy?  Every internal node whose two children
are leaves would become ONE leaf holding both triangles: its node visit (a fetch + two box tests)
disappears, and both triangles are tested whenever the node's box -- kept in its parent -- is hit.
Counts, with the product's per-lane traversal (tests/host_sim, built with -DTR_COUNT_BOTTOM), the
visits of such nodes and the box tests their children passed, for the closest-hit query.
usage: python scripts/exp_pair_leaves.py [--res 512]"""
import argparse, ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "host_sim")]
import numpy as np
import workloads as W
from sim import SimBVH
ap = argparse.ArgumentParser(); ap.add_argument("--res", type=int, default=512)
a = ap.parse_args()
hs = os.path.join(ROOT, "tests", "host_sim")
so = "/tmp/libhost_sim_bottom.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-Wno-unknown-pragmas",
                       "-DTR_COUNT_BOTTOM", "-o", so, os.path.join(hs, "host_sim.cpp")])
L = C.CDLL(so)
L.sim_steps_bottom.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 4
for name, (v, f), dist in (("headline mesh (1.31 M tris)", W.headline_mesh(8), None), ("4 nested shells (1.31 M tris)", W.nested_shells(7), 2.5),
                           ("bunny stand-in (82 k tris)", W.bunny_standin(), None)):
    rad = float(np.linalg.norm(v, axis=1).max())
    o, d = W.pinhole_grid(a.res, a.res, distance=dist or 2.5 * rad)
    o = np.ascontiguousarray(o, np.float32).reshape(-1, 3); d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
    B = SimBVH(v, f)
    n = len(o)
    nv, tt, bt, bh = (np.zeros(n, np.int32) for _ in range(4))
    L.sim_steps_bottom(B.nodes.ctypes.data, B.links.ctypes.data, B.tris.ctypes.data, B.nf, o.ctypes.data, d.ctypes.data, n,
                       nv.ctypes.data, tt.ctypes.data, bt.ctypes.data, bh.ctypes.data)
    c = B.nodes[:, 12:14].view(np.int32)
    both = int(((c[:, 0] < 0) & (c[:, 1] < 0)).sum())
    pair_visits = nv.mean() - bt.mean()
    # tests of the two children of a collapsed node: both, whenever the node is reached (an upper bound on
    # what the queued-leaf culling of the real trip would still skip); the others as today
    pair_tests = tt.mean() + (2 * bt.mean() - bh.mean())
    print(json.dumps({"scene": name, "rays": n, "nodes": int(len(c)), "nodes_with_two_leaf_children": both,
                      "node_visits_per_ray": round(float(nv.mean()), 2), "tri_tests_per_ray": round(float(tt.mean()), 2),
                      "visits_of_two_leaf_nodes_per_ray": round(float(bt.mean()), 2),
                      "their_children_passing_the_box_test": round(float(bh.mean()), 2),
                      "with_pair_leaves": {"node_visits_per_ray": round(float(pair_visits), 2), "tri_tests_per_ray": round(float(pair_tests), 2),
                                           "node_visits_change": round(float(pair_visits / nv.mean() - 1), 4),
                                           "tri_tests_change": round(float(pair_tests / tt.mean() - 1), 4)}}), flush=True)
