#!/usr/bin/env python3
"""Would a WAVE-PACKET traversal (one node per wave-trip, fetched by the scalar unit, both child boxes tested by all
64 lanes, a shared stack) beat the per-lane traversal for count / location on image tiles?  (VERDICT r03 "next" #3)

Host simulation on the product's own hierarchy (tests/host_sim: the same builder, the same slab arithmetic): for every
8x8 pixel tile of the C4 image (and the headline image) the unordered traversal's
  lane-visits (sum over the 64 rays), the slowest ray's visits (>= wave-trips of the per-lane kernel's node phase) and
  the number of DISTINCT nodes the tile's rays visit (= wave-trips of a packet traversal), the same for leaves.
A packet trip is cheaper than a per-lane trip (no gathers, no trail words: ~45 instead of ~100 VALU instructions), but
there are as many of them as the tile has distinct nodes.    CPU only:  python scripts/round4/exp_packet_count.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "host_sim")]
import numpy as np  # noqa: E402

import workloads as W  # noqa: E402
import sim  # noqa: E402


def tiles(o, d, step=4):
    """rays of every `step`-th 8x8 tile in both directions, in wave order (64 consecutive rays = one tile)"""
    h, w = o.shape[:2]
    oo, dd = [], []
    for ty in range(0, h // 8, step):
        for tx in range(0, w // 8, step):
            oo.append(o[8 * ty:8 * ty + 8, 8 * tx:8 * tx + 8].reshape(-1, 3))
            dd.append(d[8 * ty:8 * ty + 8, 8 * tx:8 * tx + 8].reshape(-1, 3))
    return np.concatenate(oo), np.concatenate(dd)


for name, (v, f), dist in (("C4 nested shells (count 0.72 ms)", W.nested_shells(7), 2.5), ("C5(i) headline mesh (count 0.37 ms)", W.headline_mesh(8), None)):
    if dist is None:
        dist = 2.5 * float(np.linalg.norm(v, axis=1).max())
    o, d = W.pinhole_grid(1024, 1024, distance=dist)
    o = np.broadcast_to(o, d.shape) if o.shape != d.shape else o
    S = sim.SimBVH(v, f)
    to, td = tiles(np.ascontiguousarray(o), d)
    st = S.packet_stats(to, td, 64)
    st = st[st[:, 0] > 0]
    lane, slow, uni, lsum, lslow, luni = [st[:, k].astype(np.float64) for k in range(6)]
    res = dict(config=name, tiles=int(len(st)), triangles=int(len(f)),
               node_visits_per_ray=round(float(lane.sum() / (64 * len(st))), 1),
               slowest_ray_visits_per_tile=round(float(slow.mean()), 1),
               distinct_nodes_per_tile=round(float(uni.mean()), 1),
               packet_trips_over_per_lane_trips=round(float(uni.sum() / slow.sum()), 2),
               leaf_tests_per_ray=round(float(lsum.sum() / (64 * len(st))), 2),
               busiest_ray_leaf_tests_per_tile=round(float(lslow.mean()), 1),
               distinct_leaves_per_tile=round(float(luni.mean()), 1),
               lane_utilisation_of_a_packet_node_trip=round(float(lane.sum() / (64 * uni.sum())), 3),
               note="per-lane kernel: ~100 VALU + 2 gathers per trip, trips >= slowest ray's visits; packet: ~45 VALU + 1 scalar load "
                    "per trip, trips = distinct nodes (+ a leaf trip per distinct leaf instead of per busiest-ray leaf)")
    # VALU estimate per wave: per-lane = slowest x 100 + busiest-leaf x 110; packet = distinct nodes x 45 + distinct leaves x 110
    res["valu_per_wave_per_lane_model"] = int(slow.mean() * 100 + lslow.mean() * 110)
    res["valu_per_wave_packet_model"] = int(uni.mean() * 45 + luni.mean() * 110)
    print(json.dumps(res), flush=True)
