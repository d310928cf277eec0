#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep (run on the GPU box): random meshes, ray sets, ray tensor
shapes and launch-shape options (work stealing thresholds, tiles, block sizes, adaptive order).
Every query must match the oracle bit for bit.  usage: python scripts/fuzz_parity.py [--iters 60] [--seed 1]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np, torch
import workloads as W
from oracle.oracle import OracleIntersector
from triro.ray.ray_optix import RayMeshIntersector
from triro.backend import ops as hops

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=60); ap.add_argument("--seed", type=int, default=1); ap.add_argument("--kind", type=int, default=-1, help="force one mesh family (5 = grid meshes with lattice rays: exact ties)")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
DEFAULTS = {"steal": 1, "tile": 1, "block_size": 128, "adaptive": 1, "xcd_chunk": 128, "compact": 1, "scramble": 1,
            "persistent": 0, "blocks_per_cu": 8,
            "tile_small": 4, "unordered": 1, "leaf_vote": 32, "stream": 1, "stream_rays": 256, "stream_refill": 0, "stream_dynamic": 1,
            "split": 1, "split_steal": 8, "grid_nodes": 1, "split_outlier": 1, "split_floor": 40, "usteal": 1, "lds_top": 0, "occ8": 0,
            "wide": 2, "wide_stack": 12, "wide_direct": 1, "expand4": 1, "expand_cus": 0, "expand_tiles": 1, "sort_inline": 1}
bad = 0
for it in range(a.iters):
    kind = rng.integers(0, 6)
    if a.kind >= 0: kind = a.kind
    grid = None
    if kind == 5:
        # round 6: EXACT TIES as the ordinary case -- a height field (or a stack of them) on an integer grid, rays on a lattice
        # that run through its vertices, edges and cell diagonals (one owner per edge: tr_math.h "exact ties")
        gn = int(rng.integers(3, 40)); layers = int(rng.integers(1, 4))
        g = np.arange(gn, dtype=np.float32)
        X, Y = np.meshgrid(g, g, indexing="ij")
        vs, fs = [], []
        for L in range(layers):
            Z = ((X * int(rng.integers(1, 9)) + Y * int(rng.integers(1, 9))) % int(rng.integers(1, 6))).astype(np.float32) * np.float32(rng.choice([0.0, 0.25, 1.0])) + 3 * L
            idx = np.arange(gn * gn).reshape(gn, gn) + L * gn * gn
            qa, qb, qc, qd = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
            vs.append(np.stack([X, Y, Z], -1).reshape(-1, 3)); fs.append(np.concatenate([np.stack([qa, qb, qc], 1), np.stack([qa, qc, qd], 1)]))
        v, f = np.concatenate(vs).astype(np.float32), np.concatenate(fs).astype(np.int32)
        grid = gn
    elif kind == 0: v, f = W.icosphere(int(rng.integers(1, 6)))
    elif kind == 1: v, f = W.random_soup(int(rng.integers(2, 6000)), seed=int(rng.integers(1 << 30)), size=float(rng.uniform(0.02, 0.6)))
    elif kind == 2: v, f = W.nested_shells(int(rng.integers(2, 5)))
    elif kind == 3:
        v, f = W.icosphere(int(rng.integers(3, 6))); v = W.displaced(v, seed=int(rng.integers(1000)), amplitude=float(rng.uniform(0.01, 0.3)))
    else: v, f = W.deep_tree_mesh(int(rng.integers(100, 3000)))
    if grid is None:
        v = (v * np.float32(rng.uniform(0.1, 20.0)) + rng.uniform(-3, 3, 3).astype(np.float32)).astype(np.float32)
    else:
        v = (v * np.float32(rng.choice([0.5, 1.0, 2.0])) + rng.integers(-4, 5, 3).astype(np.float32)).astype(np.float32)      # (stays on a binary lattice)
    lo, hi = v.min(0), v.max(0); ext = np.maximum(hi - lo, 1e-3)
    rk = rng.integers(0, 4)
    if grid is not None and rng.random() < 0.8:
        # lattice rays: origins on a half-step lattice over (and a little beyond) the patch, directions along an axis, a face
        # diagonal or a small-integer vector -- through vertices, along edges, across cell diagonals
        step = float(v[1, 1] - v[0, 1]) if len(v) > 1 and v[1, 1] != v[0, 1] else 1.0
        h = np.arange(lo[0] - step, hi[0] + step + 1e-3, step / 2, dtype=np.float32)
        k = np.arange(lo[1] - step, hi[1] + step + 1e-3, step / 2, dtype=np.float32)
        gx, gy = np.meshgrid(h, k, indexing="ij")
        dirs = np.array([[0, 0, -1], [0, 0, 1], [1, 0, -1], [0, 1, -1], [1, 1, -2], [1, -1, -1], [2, 1, -4], [1, 0, 0], [0, 1, 0], [1, 1, 0]], np.float32)
        dsel = dirs[rng.integers(0, len(dirs), gx.size)] * np.float32(rng.choice([1.0, 0.5, 3.0]))
        up = np.where(dsel[:, 2] > 0, lo[2] - 2 * step, np.where(dsel[:, 2] < 0, hi[2] + 2 * step, lo[2] + step * rng.integers(0, 4, gx.size) / 4)).astype(np.float32)
        o = np.stack([gx.ravel(), gy.ravel(), up], -1).astype(np.float32)
        flat = dsel[:, 2] == 0
        o[flat, 0] = lo[0] - 2 * step
        d = dsel.astype(np.float32)
    elif rk == 3:
        # rays that START ON the mesh (vertices, edge midpoints, points inside a face): t_key = +-0.0
        # ties, the family that exposed the -0.0 key of the stealing merge
        n = int(rng.integers(64, 20000)); fi = rng.integers(0, len(f), n); tv = v[f[fi]]
        w = rng.random((n, 3)).astype(np.float32); w[rng.random(n) < 0.4] = [1, 0, 0]; w[rng.random(n) < 0.2] = [0.5, 0.5, 0]
        w /= w.sum(1, keepdims=True)
        o = np.where((w == [1, 0, 0]).all(1)[:, None], tv[:, 0], (w[:, :1] * tv[:, 0] + w[:, 1:2] * tv[:, 1] + w[:, 2:] * tv[:, 2])).astype(np.float32)
        nrm = np.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0]).astype(np.float32)
        pick = rng.integers(0, 4, n)[:, None]
        d = np.where(pick == 0, nrm, np.where(pick == 1, -nrm, np.where(pick == 2, (lo + hi) / 2 - o, rng.normal(size=(n, 3))))).astype(np.float32)
    elif rk == 0:
        h, w = int(rng.integers(1, 40)) * 8, int(rng.integers(1, 40)) * 8
        o, d = W.pinhole_grid(w, h, distance=float(2.5 * np.linalg.norm(ext)))
        o = o + ((lo + hi) / 2).astype(np.float32)
    elif rk == 1:
        n = int(rng.integers(1, 60000)); o, d = W.hash_rays(n, int(rng.integers(1 << 20)), lo - 0.5 * ext, hi + 0.5 * ext)
    else:
        n = int(rng.integers(64, 30000)); o, d = W.hash_rays(n, int(rng.integers(1 << 20)), lo - 0.1 * ext, hi + 0.1 * ext)
        d = ((lo + hi) / 2 - o + rng.normal(0, 0.05, o.shape) * ext).astype(np.float32)   # aimed at the mesh
    opts = {"steal": int(rng.choice([0, 1, 2, 5, 17, 64])), "tile": int(rng.choice([0, 1, 2])),
            "block_size": int(rng.choice([64, 128, 128, 128, 128, 256])), "adaptive": int(rng.choice([0, 1, 1])),
            "xcd_chunk": int(rng.choice([0, 16, 128, 300])), "compact": int(rng.choice([0, 1, 1])), "scramble": int(rng.choice([0, 1])),
            # optional launch shape (persistent batches engage on batches larger than the resident grid)
            "persistent": int(rng.choice([0, 0, 1])), "blocks_per_cu": int(rng.choice([1, 8])),
            # unordered two-phase schedule of count / location (2: any as well) and its leaf-phase vote
            "unordered": int(rng.choice([0, 1, 1, 2])), "leaf_vote": int(rng.choice([1, 4, 16, 48, 64])),
            # streaming launch with wave-level ray refill (2 = forced at any size and shape)
            "tile_small": int(rng.choice([0, 1, 2, 3, 4])),
            "stream": int(rng.choice([0, 1, 2, 2])), "stream_rays": int(rng.choice([64, 100, 512, 4096])),
            "stream_refill": int(rng.choice([0, 1, 8, 16, 40, 64])), "stream_dynamic": int(rng.choice([0, 1, 1])),
            # block splitting of the stealing launch shapes (from the second launch of a batch on)
            "split": int(rng.choice([0, 1, 1, 2, 3, 4])), "split_steal": int(rng.choice([0, 2, 8, 64])),
            "grid_nodes": int(rng.choice([0, 1, 2, 2])),
            # round 3: device-side split criterion, stealing in the unordered count launch, LDS-staged node packets
            "split_outlier": int(rng.choice([0, 1, 1, 4, 8, 30])), "split_floor": int(rng.choice([0, 0, 0, 40])),
            "usteal": int(rng.choice([0, 1, 1, 2, 8, 64])), "lds_top": int(rng.choice([0, 0, 1, 2])), "occ8": int(rng.choice([0, 1, 2, 2])),
            # round 4: 8-wide compressed nodes (streaming launch, direct launch), their stack split, the expansion kernels
            "wide": int(rng.choice([0, 1, 1, 2])), "wide_stack": int(rng.choice([1, 2, 5, 12])), "wide_direct": int(rng.choice([0, 1, 2, 3, 3])),
            "expand4": int(rng.choice([0, 1, 1, 2, 3])), "expand_cus": int(rng.choice([0, 0, 1, 3])), "expand_tiles": int(rng.choice([0, 1])),
            # round 5: the deferred sort of the learned order as a workgroup of the next launch
            "sort_inline": int(rng.choice([0, 1, 1, 1]))}
    for k, val in opts.items(): hops.set_option(k, val)
    try:
        r = RayMeshIntersector(vertices=T(v), faces=T(f)); R = OracleIntersector(v, f, 1)
        ot, dt = T(o), T(d); of, df = o.reshape(-1, 3), d.reshape(-1, 3)
        for rep in range(5 if (opts["adaptive"] and o.size >= 3 * 8192) else 3):     # (enough launches of a shape for its steady state: deferred sorts)
            hit, front, tri, loc, uv = [x.cpu().numpy() for x in r.intersects_closest(ot, dt)]
            eh, ef, et, el, eu = R.closest_raw(of, df)[:5]
            ok = (np.array_equal(hit.reshape(-1), eh) and np.array_equal(front.reshape(-1), ef) and np.array_equal(tri.reshape(-1), et)
                  and np.array_equal(loc.reshape(-1, 3), el) and np.array_equal(uv.reshape(-1, 2), eu))
            cnt = R.intersects_count(of, df)
            ok &= np.array_equal(r.intersects_count(ot, dt).cpu().numpy().reshape(-1), cnt)
            ok &= np.array_equal(r.intersects_any(ot, dt).cpu().numpy().reshape(-1), cnt > 0)
            ok &= np.array_equal(r.intersects_first(ot, dt).cpu().numpy().reshape(-1), et)
            lo3, ra, tr_ = [x.cpu().numpy() for x in r.intersects_location(ot, dt)]
            el2, er2, et2 = R.intersects_location(of, df)
            ok &= np.array_equal(ra, er2) and np.array_equal(tr_, et2) and np.array_equal(lo3, el2)
            # round 3: 12-byte packed closest records expand to the dense outputs bit for bit
            exp5 = r.closest_expand(r.intersects_closest_packed(ot, dt), batch_shape=ot.shape[:-1])
            ok &= all(np.array_equal(x.cpu().numpy().reshape(e.shape), e) for x, e in zip(exp5, (hit, front, tri, loc, uv)))
            # round 4: slot-form records, expanded linearly and (image-shaped batches) in 8x8 tiles
            recs = r.intersects_closest_packed(ot, dt, slots=True)
            for rl in ((0, ot.shape[1]) if ot.dim() == 3 else (0,)):
                exp6 = r.closest_expand(recs, batch_shape=ot.shape[:-1], slots=True, row_length=int(rl))
                ok &= all(np.array_equal(x.cpu().numpy().reshape(e.shape), e) for x, e in zip(exp6, (hit, front, tri, loc, uv)))
            # ... and 4-byte records (the slot alone) finished from the rays
            sl4 = r.intersects_closest_slots(ot, dt)
            for rl in ((0, ot.shape[1]) if ot.dim() == 3 else (0,)):
                exp7 = r.closest_from_slots(ot, dt, sl4, row_length=int(rl))
                ok &= all(np.array_equal(x.cpu().numpy().reshape(e.shape), e) for x, e in zip(exp7, (hit, front, tri, loc, uv)))
            if not ok: break
    finally:
        for k, val in DEFAULTS.items(): hops.set_option(k, val)
    if not ok:
        bad += 1
        print("MISMATCH iter", it, "mesh kind", kind, "tris", len(f), "rays", o.shape, "opts", opts, flush=True)
    elif it % 10 == 0:
        print("iter", it, "ok  tris", len(f), "rays", tuple(o.shape), opts, flush=True)
print("done:", a.iters, "iterations,", bad, "mismatches")
sys.exit(1 if bad else 0)
