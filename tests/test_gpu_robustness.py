"""GPU vs brute-force oracle on adversarial geometry: duplicates, zero-area and needle
triangles, huge / tiny / far-from-origin coordinates, rays starting on and inside the mesh,
axis-aligned rays on axis-aligned geometry.  Everything bit-exact."""
import numpy as np
import pytest
import torch

import workloads as W
from oracle.oracle import OracleIntersector

pytestmark = pytest.mark.gpu


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def check(v, f, o, d, dev, mode=0):
    from triro.ray.ray_optix import RayMeshIntersector
    r = RayMeshIntersector(vertices=T(v, dev), faces=T(f, dev))
    R = OracleIntersector(v, f, mode)
    ot, dt = T(o, dev), T(d, dev)
    hit, front, tri, loc, uv = [x.cpu().numpy() for x in r.intersects_closest(ot, dt)]
    eh, ef, et, el, eu, _ = R.closest_raw(o, d)
    assert np.array_equal(hit, eh) and np.array_equal(front, ef) and np.array_equal(tri, et)
    assert np.array_equal(loc, el) and np.array_equal(uv, eu)
    cnt = R.intersects_count(o, d)
    assert np.array_equal(r.intersects_count(ot, dt).cpu().numpy(), cnt)
    assert np.array_equal(r.intersects_any(ot, dt).cpu().numpy(), cnt > 0)
    lo, ra, tr_ = [x.cpu().numpy() for x in r.intersects_location(ot, dt)]
    el2, er2, et2 = R.intersects_location(o, d)
    assert np.array_equal(ra, er2) and np.array_equal(tr_, et2) and np.array_equal(lo, el2)
    return hit


def rays_for(v, n, seed):
    lo, hi = v.min(0), v.max(0)
    ext = np.maximum(hi - lo, 1e-3 * np.maximum(np.abs(hi), 1e-30))
    return W.hash_rays(n, seed, lo - 0.5 * ext, hi + 0.5 * ext)


def test_duplicates_and_degenerates(device):
    rng = np.random.default_rng(0)
    v, f = W.random_soup(300, seed=1)
    f = np.concatenate([f, f[:100], f[:100]])                    # every triangle of the first 100 three times
    v = np.concatenate([v, rng.random((30, 3)).astype(np.float32)])
    zero_area = np.array([[900, 900, 901], [902, 902, 902], [903, 904, 903]], np.int32)   # repeated vertices
    f = np.concatenate([f, zero_area])
    o, d = rays_for(v, 20000, 3)
    hit = check(v, f, o, d, device)
    assert hit.any()
    # collinear (zero-area) triangles with distinct vertices + needles
    v2 = np.array([[0, 0, 0], [1, 1, 1], [2, 2, 2], [0, 0, 0], [1e-7, 0, 0], [0, 5, 0], [0, 0, 1], [3, 0, 1], [0, 1e-6, 1]], np.float32)
    f2 = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8]], np.int32)
    o2, d2 = rays_for(v2, 20000, 4)
    check(v2, f2, o2, d2, device)


@pytest.mark.parametrize("scale,offset", [(1e6, 0.0), (1e-6, 0.0), (1.0, 1e5), (1e3, -7e4)])
def test_coordinate_ranges(device, scale, offset):
    v, f = W.icosphere(3)
    v = (v * np.float32(scale) + np.float32(offset)).astype(np.float32)
    o, d = rays_for(v, 20000, 5)
    check(v, f, o, d, device)
    # camera far away from a mesh that sits far from the origin
    c = v.mean(0)
    oo = np.tile((c + np.float32(50 * scale) * np.array([0.3, 0.2, 1.0], np.float32)).astype(np.float32), (4096, 1))
    tgt, _ = rays_for(v, 4096, 6)
    check(v, f, oo, (tgt - oo).astype(np.float32), device)


def test_rays_starting_on_and_inside(device):
    v, f = W.icosphere(3)
    rng = np.random.default_rng(2)
    cen = v[f].mean(1)                                          # points ON the surface (triangle centroids)
    d_out = (cen / np.linalg.norm(cen, axis=1, keepdims=True)).astype(np.float32)
    o = np.concatenate([cen, cen, np.zeros((500, 3), np.float32), v[:300]]).astype(np.float32)
    d = np.concatenate([d_out, -d_out, rng.normal(size=(500, 3)).astype(np.float32), -v[:300]]).astype(np.float32)
    check(v, f, o, d, device)


def test_axis_aligned_everything(device):
    # a 6x6x6 block of axis-aligned quads; rays along +-x/y/z exactly on the grid lines,
    # through vertices and edges, with zero direction components
    g = np.arange(0, 7, dtype=np.float32)
    vs, fs = [], []
    for z in g[:4]:
        base = len(vs) * 1
        gx, gy = np.meshgrid(g, g, indexing="ij")
        pts = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, z)], 1)
        off = sum(len(p) for p in vs)
        vs.append(pts)
        idx = np.arange(49).reshape(7, 7)
        a, b, c, dd = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
        fs.append(np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, dd], 1)]) + off)
    v = np.concatenate(vs).astype(np.float32)
    f = np.concatenate(fs).astype(np.int32)
    h = np.arange(-1, 8, 0.5, dtype=np.float32)
    hx, hy = np.meshgrid(h, h, indexing="ij")
    o = np.stack([hx.ravel(), hy.ravel(), np.full(hx.size, 9.0, np.float32)], 1)
    d = np.tile(np.array([0, 0, -1], np.float32), (len(o), 1))
    o2 = np.stack([np.full(hx.size, -3.0, np.float32), hx.ravel(), hy.ravel() * 0.5], 1)   # in-plane rays (z = k/2)
    d2 = np.tile(np.array([1, 0, 0], np.float32), (len(o2), 1))
    hit = check(v, f, np.concatenate([o, o2]), np.concatenate([d, d2]), device)
    n1 = len(o)
    # (round 6: one owner per edge on exact ties -- the grid patch is HALF-OPEN like a rasterised quad: the interior, the
    # interior grid lines and exactly one of each pair of opposite boundary lines; check() has compared everything with
    # the oracle, counts included: every vertical ray that hits the four stacked patches counts exactly 4)
    x, y = o[:, 0], o[:, 1]
    assert hit[:n1][(x > 0) & (x < 6) & (y > 0) & (y < 6)].all() and not hit[:n1][(x < 0) | (x > 6) | (y < 0) | (y > 6)].any()
    for u, w in ((x, y), (y, x)):
        lo, hi = hit[:n1][(u == 0) & (w > 0) & (w < 6)], hit[:n1][(u == 6) & (w > 0) & (w < 6)]
        assert len(lo) == len(hi) and lo.all() != hi.all() and (lo.all() or not lo.any()) and (hi.all() or not hi.any())
    assert int(hit[:n1][((x == 0) | (x == 6)) & ((y == 0) | (y == 6))].sum()) == 1


def test_concurrent_streams_and_threads_share_one_handle(device):
    """Many host threads, each on its own stream, query ONE handle with different batch sizes while
    the launch-order hints (per handle, per stream; 8 slots) are learned and re-sorted.  More
    streams than slots on purpose: the extra ones run without hints.  Every result must equal the
    single-stream answer; hints are scheduling only."""
    import threading
    from triro.ray.ray_optix import RayMeshIntersector
    v, f = W.bunny_standin()
    r = RayMeshIntersector(vertices=T(v, device), faces=T(f, device))
    o, d = W.hash_rays(400_000, 31, v.min(0) * 1.4, v.max(0) * 1.4)
    ot, dt = T(o, device), T(d, device)
    ref = [x.clone() for x in r.intersects_closest(ot, dt)]
    cnt_ref = r.intersects_count(ot, dt).clone()
    loc_ref = {}
    torch.cuda.synchronize()
    sizes = [400_000, 131_072, 65_537, 300_001, 8_193, 200_000, 99_999, 262_144, 50_000, 399_999, 77_777, 16_384]
    errors = []
    for n in set(sizes):    # multi-hit answers (scan + fill) per size, computed alone
        loc_ref[n] = [x.clone() for x in r.intersects_location(ot[:n], dt[:n])]
    torch.cuda.synchronize()

    def worker(k):
        try:
            s = torch.cuda.Stream(device=device)
            n = sizes[k]
            with torch.cuda.stream(s):
                for it in range(12):
                    out = r.intersects_closest(ot[:n], dt[:n])
                    if it % 4 == 3:
                        c = r.intersects_count(ot[:n], dt[:n])
                        s.synchronize()
                        if not torch.equal(c, cnt_ref[:n]):
                            errors.append((k, it, "count"))
                        got = r.intersects_location(ot[:n], dt[:n])     # scans on this stream
                        s.synchronize()
                        if not all(torch.equal(x, y) for x, y in zip(got, loc_ref[n])):
                            errors.append((k, it, "location"))
                    s.synchronize()
                    for x, y in zip(out, ref):
                        if not torch.equal(x, y[:n]):
                            errors.append((k, it, "closest"))
                            return
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(sizes))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


def test_randomised_parity_sweep(device):
    """scripts/fuzz_parity.py with a fixed seed: random meshes, rays, tensor shapes and launch-shape
    options (stealing thresholds, tiles, block sizes, ...), every query bit-exact against the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_parity.py"), "--iters", "16", "--seed", "3"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
