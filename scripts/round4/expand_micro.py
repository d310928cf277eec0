#!/usr/bin/env python3
"""What bounds tr_closest_expand?  Calibration of the box (torch fill / copy of the same byte counts) next to the
expansion kernels (expand4 = 0..3) on (a) records that are all misses -- a pure stream: 12 B in, 26 B out per ray,
no gathers -- and (b) the records of the headline image (55 % hits: face + vertex row gathers), at 7.3 M rays (what the
destination rank of an 8-GPU c5i run expands per step) and at 93 M rays (c5ii).  One JSON line per measurement.
    python scripts/round4/expand_micro.py [--only MODE] [--reps N]        (rocprofv3 --pmc target with --only)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import workloads as W  # noqa: E402
import triro.backend.ops as hops  # noqa: E402
from triro.ray.ray_optix import RayMeshIntersector  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", type=int, default=-1, help="run only this expand4 mode on the image records (profiling target)")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--big", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def out(**kw):
    print(json.dumps(kw), flush=True)


v, f = W.headline_mesh(8)
r = RayMeshIntersector(vertices=torch.from_numpy(v).to(dev), faces=torch.from_numpy(f).to(dev))
rad = float(np.linalg.norm(v, axis=1).max())
o_np, d_np = W.pinhole_grid(1024, 1024, distance=2.5 * rad)
o = torch.from_numpy(np.ascontiguousarray(o_np)).to(dev)
recs = []
for k in range(7):
    d = torch.from_numpy(np.roll(d_np, k * 7 + 7, axis=0)).to(dev)
    recs.append(r.intersects_closest_packed(o, d))
img = torch.cat(recs)                       # 7.3 M records of seven different images
n = img.shape[0]
miss = torch.full_like(img, -1)


def outs_for(m):
    return (torch.empty(m, dtype=torch.bool, device=dev), torch.empty(m, dtype=torch.bool, device=dev),
            torch.empty(m, dtype=torch.int32, device=dev), torch.empty((m, 3), device=dev), torch.empty((m, 2), device=dev))


if a.only >= 0:
    hops.set_option("expand4", a.only)
    outs = outs_for(n)
    sec = timeit(lambda: r.closest_expand(img, outs=outs), reps=a.reps)
    out(what="image records", mode=a.only, rays=n, ms=round(sec * 1e3, 4), gbps=round(n * 38 / sec / 1e9, 1))
    sys.exit(0)
# calibration: the same bytes with torch's own kernels
for rays in (n, 93_000_000 if a.big else n):
    wbuf = torch.empty(rays * 26, dtype=torch.uint8, device=dev)
    rbuf = torch.empty(rays * 12, dtype=torch.uint8, device=dev)
    sec = timeit(lambda: wbuf.fill_(1))
    out(what="torch fill (26 B/ray)", rays=rays, ms=round(sec * 1e3, 4), gbps=round(rays * 26 / sec / 1e9, 1))
    wb2 = wbuf[:rays * 12]
    sec = timeit(lambda: wb2.copy_(rbuf))
    out(what="torch copy (12 B/ray read + 12 B/ray write)", rays=rays, ms=round(sec * 1e3, 4), gbps=round(rays * 24 / sec / 1e9, 1))
    f4 = torch.empty(rays * 26 // 16, 4, dtype=torch.float32, device=dev)
    sec = timeit(lambda: f4.fill_(1.0))
    out(what="torch fill float4-ish", rays=rays, ms=round(sec * 1e3, 4), gbps=round(f4.numel() * 4 / sec / 1e9, 1))
    del wbuf, rbuf, f4
for name, rec in (("all-miss records", miss), ("image records (55 % hits)", img)):
    reps_list = [(n, rec)]
    if a.big:
        big = rec.repeat(13, 1)[:93_000_000]
        reps_list.append((big.shape[0], big))
    for rays, rr in reps_list:
        outs = outs_for(rays)
        for mode in (0, 1, 2, 3):
            hops.set_option("expand4", mode)
            sec = timeit(lambda: r.closest_expand(rr, outs=outs), reps=10)
            out(what=name, mode=mode, rays=rays, ms=round(sec * 1e3, 4), gbps=round(rays * 38 / sec / 1e9, 1))
        del outs
hops.set_option("expand4", 1)
# slot form: one 48-byte triangle record per hit
recs = []
for k in range(7):
    d = torch.from_numpy(np.roll(d_np, k * 7 + 7, axis=0)).to(dev)
    recs.append(r.intersects_closest_packed(o, d, slots=True))
img_s = torch.cat(recs)
for rays, rr in ((n, img_s),) + (((93_000_000, img_s.repeat(13, 1)[:93_000_000]),) if a.big else ()):
    outs = outs_for(rays)
    sec = timeit(lambda: r.closest_expand(rr, outs=outs, slots=True), reps=10)
    out(what="image records, SLOT form", rays=rays, ms=round(sec * 1e3, 4), gbps=round(rays * 38 / sec / 1e9, 1))
    sec = timeit(lambda: r.closest_expand(rr, outs=outs, slots=True, row_length=1024), reps=10)
    out(what="image records, SLOT form, 8x8 tiles", rays=rays, ms=round(sec * 1e3, 4), gbps=round(rays * 38 / sec / 1e9, 1))
    del outs
# incoherent hits: the records of 12.5 M hash rays (a C5(ii) shard), both forms
ho, hd = W.hash_rays_torch(12_500_000, 99, v.min(0) * 1.5, v.max(0) * 1.5, device=dev)
for slots in (False, True):
    rec = r.intersects_closest_packed(ho, hd, slots=slots)
    outs = outs_for(rec.shape[0])
    sec = timeit(lambda: r.closest_expand(rec, outs=outs, slots=slots), reps=10)
    out(what="C5(ii) shard records, " + ("SLOT" if slots else "face") + " form", rays=rec.shape[0], hit_fraction=round(float((rec[:, 0] >= 0).float().mean()), 4),
        ms=round(sec * 1e3, 4), gbps=round(rec.shape[0] * 38 / sec / 1e9, 1))
