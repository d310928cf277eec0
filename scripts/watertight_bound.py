#!/usr/bin/env python3
"""How far can "bit-exact vs the reference" be off?  

The reference's triangle test is OptiX' built-in one (RT cores, optixTrace: shaders.cu:86,163), documented as
watertight; this repository's contract is Moller-Trumbore in float32 (csrc/tr_math.h), which the GPU path and the
oracle evaluate bit for bit alike -- but which can answer differently from a watertight test for a ray that
grazes an edge or a vertex.  This script counts those rays per BASELINE config: the contract's closest hit and
hit count (oracle, = the HIP path bit for bit, tests/test_gpu_*.py) against the WATERTIGHT float64 test of Woop /
Benthin / Wald 2013 (oracle/triro_oracle.c, "WATERTIGHT float64 reference"; shares nothing with the contract).

    python scripts/watertight_bound.py [--quick] [--full] > profiles/r05_watertight_bound.jsonl        (CPU only)

Per config: rays, and the number of rays whose
  only_contract / only_watertight   hit mask differs (the second is the "crack": a ray lost between two triangles)
  tri_diff_same_t                   both hit, different triangle, distances equal within 1e-5 relative: the ray crosses
                                    a shared edge or vertex, either triangle is a correct answer (tie-break differs)
  tri_diff_other                    both hit, different triangle, different distance: a real disagreement
  count_diff                        hit counts differ (any cause)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import workloads as W  # noqa: E402
from oracle.oracle import OracleIntersector  # noqa: E402


def compare(name, v, f, o, d):
    R = OracleIntersector(v, f, mode=1)
    hit, _, tri, _, _, t = R.closest_raw(o, d)
    cnt = R.intersects_count(o, d)
    wtri, wt, wcnt = R.watertight(o, d)
    hit, tri, t, cnt = hit.reshape(-1), tri.reshape(-1), t.reshape(-1).astype(np.float64), cnt.reshape(-1)
    wtri, wt, wcnt = wtri.reshape(-1), wt.reshape(-1), wcnt.reshape(-1)
    whit = wtri >= 0
    both = hit & whit
    tdiff = both & (tri != wtri)
    with np.errstate(invalid="ignore"):
        dt = np.where(both, np.abs(np.where(both, t, 0.0) - np.where(both, wt, 0.0)), np.inf)
    close = dt <= 1e-5 * np.maximum(1.0, np.abs(np.where(both, wt, 1.0)))
    n = hit.size
    res = dict(config=name, rays=int(n), triangles=int(len(f)), hits_contract=int(hit.sum()), hits_watertight=int(whit.sum()),
               only_contract=int((hit & ~whit).sum()), only_watertight=int((~hit & whit).sum()),
               tri_diff_same_t=int((tdiff & close).sum()), tri_diff_other=int((tdiff & ~close).sum()),
               count_diff=int((cnt != wcnt).sum()),
               max_rel_t_diff_same_tri=float(np.max(dt[both & ~tdiff] / np.maximum(1.0, np.abs(wt[both & ~tdiff])), initial=0.0)))
    res["hit_mask_diff_rate"] = (res["only_contract"] + res["only_watertight"]) / n
    res["real_disagreement_rate"] = (res["only_contract"] + res["only_watertight"] + res["tri_diff_other"]) / n
    return res


def configs(quick=False, full=False):
    """full (round 5, VERDICT r04 "next" #7): additionally ALL 10 M rays of C3 and one 12.5 M-ray shard of C5(ii)"""
    res = 256 if quick else 1024
    v, f, label = W.bunny_mesh()
    o, d = W.pinhole_grid(res, res, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
    yield f"C2 ({label}), {res}^2 pinhole", v, f, o, d
    if not quick:
        lo, hi = v.min(0) * 1.5, v.max(0) * 1.5
        o3, d3 = W.hash_rays(2_000_000, 1234, lo, hi)
        yield "C3 (first 2M of the 10M hash rays)", v, f, o3, d3
        if full:
            o3, d3 = W.hash_rays(10_000_000, 1234, lo, hi)
            yield "C3 (all 10M hash rays)", v, f, o3, d3
            del o3, d3
    v, f = W.nested_shells(5 if quick else 7)
    o, d = W.pinhole_grid(res, res)
    yield f"C4 nested shells, {res}^2 pinhole", v, f, o, d
    v, f = W.headline_mesh(6 if quick else 8)
    o, d = W.pinhole_grid(res, res, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
    yield f"C5(i) headline mesh, {res}^2 pinhole", v, f, o, d
    if full and not quick:
        lo, hi = v.min(0) * 1.5, v.max(0) * 1.5
        o5, d5 = W.hash_rays(12_500_000, 99, lo, hi)
        yield "C5(ii) shard 0 (12.5M of the 100M hash rays)", v, f, o5, d5
        del o5, d5
    v, f = W.terrain() if not quick else W.terrain(n=180)
    w, h = (1024, 576) if not quick else (256, 144)
    _, d = W.ref_shape_rays(W.TERRAIN_EYE, W.TERRAIN_TARGET, w, h, 444.0 * w / 640)
    o = np.broadcast_to(np.asarray(W.TERRAIN_EYE, np.float32), d.shape)
    yield f"TERRAIN grazing camera, {w}x{h}", v, f, o, d


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    for name, v, f, o, d in configs(quick, "--full" in sys.argv):
        print(json.dumps(compare(name, v, f, np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32))), flush=True)
