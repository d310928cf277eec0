#!/usr/bin/env python3
"""The contract against an independent WATERTIGHT float64 reference: hit mask, triangle, count -- and what the API returns.

The reference's triangle test is OptiX' built-in one (RT cores, optixTrace: shaders.cu:86,163), documented as
watertight.  Since round 6 the contract (csrc/tr_math.h, version 3; restated in oracle/triro_oracle.c) is watertight
too: Moller-Trumbore in float32 decides where a proven error bound lets it, float64 edge functions decide the rest, and
the barycentrics of the winning triangle come from float64.  This script compares, per BASELINE config, the contract's
closest hit / hit count / uv / loc (oracle = the HIP path bit for bit, tests/test_gpu_*.py) with the published form of
Woop / Benthin / Wald 2013 evaluated in float64 (oracle_watertight: shear with divisions, its own permutation and its
own BVH walk on float64 slabs -- an implementation of its own).

    python scripts/watertight_bound.py [--quick] [--full] > profiles/r06_watertight_bound.jsonl        (CPU only)

Per config: rays, and the number of rays whose
  only_contract / only_watertight   hit mask differs (the second would be a "crack": a ray lost between two triangles)
  tri_diff_same_t                   both hit, different triangle, distances equal within 1e-5 relative: the ray crosses
                                    a shared edge or vertex, either triangle is a correct answer (tie-break differs)
  tri_diff_other                    both hit, different triangle, different distance: a real disagreement
  count_diff                        hit counts differ (any cause)
and, over the rays that hit the SAME triangle in both:
  max_rel_t_diff_same_tri           distance (not returned by the API; float32 Moller-Trumbore unless the float64 part ran)
  max_abs_uv_diff / max_rel_uv_diff barycentrics (w0, w1) as returned; relative = |diff| / max(|ref|, 1e-3)
  max_abs_loc_diff / max_rel_loc_diff   location; relative = max-norm of the difference / max-norm of the float64 location
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
    hit, _, tri, loc, uv, t = R.closest_raw(o, d)
    cnt = R.intersects_count(o, d)
    # the reference sees the rays the contract traces: anchored where they enter the mesh's box (a float32 ray transform:
    # OracleIntersector.anchor); distances of both are measured from there
    wtri, wt, wcnt, wuvw, wloc = R.watertight(R.anchor(o, d), d, with_bary=True)
    hit, tri, t, cnt = hit.reshape(-1), tri.reshape(-1), t.reshape(-1).astype(np.float64), cnt.reshape(-1)
    loc, uv = loc.reshape(-1, 3).astype(np.float64), uv.reshape(-1, 2).astype(np.float64)
    wtri, wt, wcnt, wuvw, wloc = wtri.reshape(-1), wt.reshape(-1), wcnt.reshape(-1), wuvw.reshape(-1, 3), wloc.reshape(-1, 3)
    whit = wtri >= 0
    both = hit & whit
    tdiff = both & (tri != wtri)
    with np.errstate(invalid="ignore"):
        dt = np.where(both, np.abs(np.where(both, t, 0.0) - np.where(both, wt, 0.0)), np.inf)
    close = dt <= 1e-5 * np.maximum(1.0, np.abs(np.where(both, wt, 1.0)))
    n = hit.size
    same = both & ~tdiff
    duv = np.abs(uv[same] - wuvw[same][:, :2])
    dloc = np.abs(loc[same] - wloc[same]).max(axis=1, initial=0.0) if same.any() else np.zeros(0)
    res = dict(config=name, rays=int(n), triangles=int(len(f)), hits_contract=int(hit.sum()), hits_watertight=int(whit.sum()),
               only_contract=int((hit & ~whit).sum()), only_watertight=int((~hit & whit).sum()),
               tri_diff_same_t=int((tdiff & close).sum()), tri_diff_other=int((tdiff & ~close).sum()),
               count_diff=int((cnt != wcnt).sum()),
               max_rel_t_diff_same_tri=float(np.max(dt[same] / np.maximum(1.0, np.abs(wt[same])), initial=0.0)),
               max_abs_uv_diff=float(duv.max(initial=0.0)),
               max_rel_uv_diff=float((duv / np.maximum(np.abs(wuvw[same][:, :2]), 1e-3)).max(initial=0.0)),
               max_abs_loc_diff=float(dloc.max(initial=0.0)),
               max_rel_loc_diff=float((dloc / np.maximum(np.abs(wloc[same]).max(axis=1, initial=0.0), 1e-30)).max(initial=0.0)) if same.any() else 0.0)
    res["hit_mask_diff_rate"] = (res["only_contract"] + res["only_watertight"]) / n
    res["real_disagreement_rate"] = (res["only_contract"] + res["only_watertight"] + res["tri_diff_other"]) / n
    return res


def configs(quick=False, full=False):
    """full: additionally ALL 10 M rays of C3 and (round 6) ALL EIGHT 12.5 M-ray shards of C5(ii)"""
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
        for sh in range(8):
            o5, d5 = W.hash_rays(12_500_000, 99, lo, hi, start=sh * 12_500_000)
            yield f"C5(ii) shard {sh} (12.5M of the 100M hash rays)", v, f, o5, d5
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
