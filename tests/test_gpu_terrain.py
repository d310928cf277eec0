"""An OPEN scene at full size: a 1 048 352-triangle height field seen from just above the ground
(`workloads.terrain`).  Rays graze the surface for a long way before they hit (up to 11 surfaces per ray),
the upper part of the image misses everything -- skewed block costs and rays that leave the mesh, the
opposite of the closed blobs and of the interior scene the launch policy was tuned on.  All five queries,
the multi-hit list and stream compaction against the oracle, bit for bit, over the launches in which the
launch order, the split set and the node flavour are learned."""
import numpy as np
import pytest
import torch

import workloads as W
from oracle.oracle import OracleIntersector

pytestmark = pytest.mark.gpu


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


@pytest.fixture(scope="module")
def land(device):
    from triro.ray.ray_optix import RayMeshIntersector
    v, f = W.terrain()
    assert len(f) == 1048352
    return v, f, RayMeshIntersector(vertices=T(v, device), faces=T(f, device)), OracleIntersector(v, f, 1)


@pytest.mark.parametrize("w,h,focal", [(640, 360, 444.0), (1024, 1024, 900.0)])
def test_grazing_camera_over_the_terrain(land, device, w, h, focal):
    v, f, r, R = land
    o, d = W.ref_shape_rays(W.TERRAIN_EYE, W.TERRAIN_TARGET, w=w, h=h, f=focal)
    ot = torch.from_numpy(np.array(W.TERRAIN_EYE, np.float32)).to(device).expand(h, w, 3)      # stride-0 origin
    dt = T(d, device)
    of, df = np.ascontiguousarray(o.reshape(-1, 3)), d.reshape(-1, 3)
    eh, ef, et, el, eu, _ = R.closest_raw(of, df)
    cnt = R.intersects_count(of, df)
    assert 0.3 < eh.mean() < 0.8 and cnt.max() >= 8 and ef[eh].mean() > 0.99   # sky above, ridges behind ridges
    for k in range(12):
        hit, front, tri, loc, uv = [x.cpu().numpy() for x in r.intersects_closest(ot, dt)]
        assert hit.shape == (h, w)
        assert np.array_equal(hit.reshape(-1), eh) and np.array_equal(tri.reshape(-1), et), f"launch {k}"
        assert np.array_equal(front.reshape(-1), ef), f"launch {k}"
        assert np.array_equal(loc.reshape(-1, 3), el) and np.array_equal(uv.reshape(-1, 2), eu), f"launch {k}"
        if k % 3 == 0:
            assert np.array_equal(r.intersects_first(ot, dt).cpu().numpy().reshape(-1), et)
            assert np.array_equal(r.intersects_any(ot, dt).cpu().numpy().reshape(-1), eh)
            assert np.array_equal(r.intersects_count(ot, dt).cpu().numpy().reshape(-1), cnt), f"launch {k}"
    li = r.as_wrapper.last_launch()
    assert li["shape"] in (1, 3) and li["learned_order"] == 1
    e_loc, e_ray, e_tri = R.intersects_location(of, df)
    loc, ray, tri = r.intersects_location(ot, dt)
    assert np.array_equal(ray.cpu().numpy(), e_ray) and np.array_equal(tri.cpu().numpy(), e_tri)
    assert np.array_equal(loc.cpu().numpy(), e_loc)
    hit, front, ridx, tric, locc, uvc = r.intersects_closest(ot, dt, stream_compaction=True)
    assert np.array_equal(ridx.cpu().numpy(), np.flatnonzero(eh).astype(np.int32)) and np.array_equal(tric.cpu().numpy(), et[eh])
    assert np.array_equal(locc.cpu().numpy(), el[eh])


def test_incoherent_rays_over_the_terrain(land, device):
    """4.5 M hash rays in a slab around the surface (the streaming launch: above 4 M rays): most of them leave the mesh"""
    v, f, r, R = land
    lo, hi = np.array([-21, -3, -21], np.float32), np.array([21, 5, 21], np.float32)
    n = 4_500_000
    o, d = W.hash_rays_torch(n, 11, lo, hi, device=device)
    hit, front, tri, loc, uv = r.intersects_closest(o, d)
    sub = slice(0, n, 6)
    on, dn = o[sub].cpu().numpy(), d[sub].cpu().numpy()
    eh, ef, et, el, eu, _ = R.closest_raw(on, dn)
    assert 0.05 < eh.mean() < 0.95
    assert np.array_equal(hit[sub].cpu().numpy(), eh) and np.array_equal(tri[sub].cpu().numpy(), et)
    assert np.array_equal(front[sub].cpu().numpy(), ef) and np.array_equal(loc[sub].cpu().numpy(), el)
    assert np.array_equal(uv[sub].cpu().numpy(), eu)
    assert torch.equal(r.intersects_first(o, d), tri) and torch.equal(r.intersects_any(o, d), hit)
    assert np.array_equal(r.intersects_count(o, d)[sub].cpu().numpy(), R.intersects_count(on, dn))
