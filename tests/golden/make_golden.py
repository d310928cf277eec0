"""Generates the committed golden fixtures tests/golden/*.npz with the CPU oracle in
BRUTE-FORCE mode (ground truth of the arithmetic contract).

The reference itself cannot produce vectors (OptiX + NVIDIA hardware; its tests hold none),
so the fixtures record: inputs, and the oracle's outputs on them.  Run from the repo root:

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from oracle.oracle import OracleIntersector  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def record(name, v, f, o, d):
    o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
    R = OracleIntersector(v, f, mode=0)
    hit, front, tri, loc, uv, t = R.closest_raw(o, d)
    cnt = R.intersects_count(o, d)
    lloc, lray, ltri, lt = R.intersects_location(o, d, with_t=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), vertices=v, faces=f, origins=o, directions=d,
                        hit=hit, front=front, tri=tri, loc=loc, uv=uv, t=t, count=cnt,
                        loc_loc=lloc, loc_ray=lray, loc_tri=ltri, loc_t=lt)
    print(name, "rays", len(o), "tris", len(f), "hits", int(hit.sum()), "multi", len(lray))


def main():
    # C1 (BASELINE.json config 1, reduced grid): icosphere 80 tris, ortho rays
    v, f = W.icosphere(1)
    o, d = W.ortho_grid(64)
    record("c1_icosphere80_ortho64", v, f, o, d)
    # README variant: perspective, broadcast origin, un-normalised directions
    v, f = W.icosphere(3)
    o, d = W.readme_perspective(96)
    record("readme_icosphere1280_persp96", v, f, np.ascontiguousarray(o), d)
    # multi-hit: nested shells, rays through the centre region
    v, f = W.nested_shells(2, radii=(1.0, 0.8, 0.6, 0.4, 0.3))   # 10 hits on central rays (> cap 8)
    o, d = W.pinhole_grid(48, 48, distance=2.5)
    record("shells5_pinhole48", v, f, np.ascontiguousarray(o), d)
    # incoherent rays vs triangle soup (both windings, overlaps)
    v, f = W.random_soup(400, seed=3)
    o, d = W.hash_rays(4096, 1234, v.min(0) * 1.5, v.max(0) * 1.5)
    record("soup400_hash4096", v, f, o, d)
    # axis-aligned box: rays exactly on edges/faces, zero direction components
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32)
    f = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1],
                  [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], np.int32)
    g = np.linspace(-1.5, 1.5, 25, dtype=np.float32)
    yy, xx = np.meshgrid(g, g, indexing="ij")
    o = np.stack([xx, yy, np.full_like(xx, 3)], -1).reshape(-1, 3)
    d = np.broadcast_to(np.array([0, 0, -1], np.float32), o.shape)
    o2 = np.stack([np.full_like(xx, -3), xx, yy], -1).reshape(-1, 3)
    d2 = np.broadcast_to(np.array([1, 0, 0], np.float32), o2.shape)
    record("cube_axis_rays", v, f, np.concatenate([o, o2]), np.concatenate([d, d2]))


if __name__ == "__main__":
    main()
