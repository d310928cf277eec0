"""An interior scene at full size (VERDICT r02 missing #3 / next #5): 909 088 triangles, the camera
INSIDE a closed room, 5.7 surfaces per ray on average (up to 13: beyond the multi-hit cap), rays in
the reference's published shape (640 x 360, f = 444 px, stride-0 origin: test/performance_test.py:
10-20, 39-44).  All five queries, stream compaction and contains_points against the oracle, bit for
bit, over several launches (cold -> learned order -> split slots)."""
import numpy as np
import pytest
import torch

import workloads as W
from oracle.oracle import OracleIntersector

pytestmark = pytest.mark.gpu


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


@pytest.fixture(scope="module")
def room(device):
    from triro.ray.ray_optix import RayMeshIntersector
    v, f = W.interior_room()
    assert len(f) == 909088
    return v, f, RayMeshIntersector(vertices=T(v, device), faces=T(f, device)), OracleIntersector(v, f, 1)


def test_reference_shaped_camera_inside_the_room(room, device):
    v, f, r, R = room
    o, d = W.ref_shape_rays(W.INTERIOR_EYE, W.INTERIOR_TARGET)
    assert o.strides[:2] == (0, 0) and o.shape == (360, 640, 3)           # the stride-0 origin of the reference
    ot = torch.from_numpy(np.array(W.INTERIOR_EYE, np.float32)).to(device).expand(360, 640, 3)
    dt = T(d, device)
    assert ot.stride()[:2] == (0, 0)
    of, df = np.ascontiguousarray(o.reshape(-1, 3)), d.reshape(-1, 3)
    eh, ef, et, el, eu, _ = R.closest_raw(of, df)
    cnt = R.intersects_count(of, df)
    assert eh.all() and cnt.mean() > 5 and cnt.max() > 8                   # closed room, deep
    e_loc, e_ray, e_tri = R.intersects_location(of, df)
    for k in range(8):
        hit, front, tri, loc, uv = [x.cpu().numpy() for x in r.intersects_closest(ot, dt)]
        assert hit.shape == (360, 640) and hit.all()
        assert np.array_equal(tri.reshape(-1), et) and np.array_equal(front.reshape(-1), ef), f"launch {k}"
        assert np.array_equal(loc.reshape(-1, 3), el) and np.array_equal(uv.reshape(-1, 2), eu), f"launch {k}"
        assert np.array_equal(r.intersects_first(ot, dt).cpu().numpy().reshape(-1), et)
        assert r.intersects_any(ot, dt).all()
        assert np.array_equal(r.intersects_count(ot, dt).cpu().numpy().reshape(-1), cnt), f"launch {k}"
    loc, ray, tri = r.intersects_location(ot, dt)
    assert np.array_equal(ray.cpu().numpy(), e_ray) and np.array_equal(tri.cpu().numpy(), e_tri)
    assert np.array_equal(loc.cpu().numpy(), e_loc)
    assert loc.shape[0] == int(np.minimum(cnt, 8).sum())
    hit, front, ridx, tric, locc, uvc = r.intersects_closest(ot, dt, stream_compaction=True)
    assert np.array_equal(ridx.cpu().numpy(), np.arange(360 * 640, dtype=np.int32)) and np.array_equal(tric.cpu().numpy(), et)


def test_rays_that_start_inside_at_scale(room, device):
    """4.5 M hash rays with origins INSIDE the room (the streaming launch: above 4 M rays) and 300 000 secondary rays that
    start exactly on surfaces (first hits of a camera trace, reflected): every ray hits the closed room."""
    v, f, r, R = room
    lo, hi = np.array([-3.9, 0.1, -2.9], np.float32), np.array([3.9, 2.9, 2.9], np.float32)
    n = 4_500_000
    o, d = W.hash_rays_torch(n, 7, lo, hi, device=device)
    hit, front, tri, loc, uv = r.intersects_closest(o, d)
    sub = slice(0, n, 8)
    on, dn = o[sub].cpu().numpy(), d[sub].cpu().numpy()
    eh, ef, et, el, eu, _ = R.closest_raw(on, dn)
    assert np.array_equal(hit[sub].cpu().numpy(), eh) and np.array_equal(tri[sub].cpu().numpy(), et)
    assert np.array_equal(front[sub].cpu().numpy(), ef) and np.array_equal(loc[sub].cpu().numpy(), el)
    assert float(hit.float().mean()) > 0.999                                  # (a ray along a wall seam may slip out)
    assert torch.equal(r.intersects_first(o, d), tri) and torch.equal(r.intersects_any(o, d), hit)
    cnt = r.intersects_count(o, d)
    assert np.array_equal(cnt[sub].cpu().numpy(), R.intersects_count(on, dn))
    # secondary rays from the first hits, mirrored about +y: origins exactly on triangles (t_key = +-0)
    o2 = loc[:300_000].contiguous()
    d2 = (d[:300_000] * torch.tensor([1.0, -1.0, 1.0], device=device)).contiguous()
    got = [x.cpu().numpy() for x in r.intersects_closest(o2, d2)]
    exp = R.closest_raw(o2.cpu().numpy(), d2.cpu().numpy())
    for g, e in zip(got, exp[:5]):
        assert np.array_equal(g, e)
    assert (exp[5][exp[0]] == 0).sum() > 1000, "secondary rays must include zero-distance hits"


def test_contains_points_in_the_room(room, device):
    """the room's surfaces face inwards, so "inside the mesh" is inside furniture or outside the room;
    5 000 points against the oracle's restatement of ray_optix.py:231-279"""
    v, f, r, R = room
    rng = np.random.default_rng(3)
    pts = (rng.random((5000, 3)) * np.array([8.4, 3.4, 6.4]) - np.array([4.2, 0.2, 3.2])).astype(np.float32)
    got = r.contains_points(T(pts, device), _retry_direction=torch.tensor([0.3, -0.5, 0.2]))
    exp = R.contains_points(pts, _retry_dirs=iter([np.array([0.3, -0.5, 0.2], np.float32)]))
    assert np.array_equal(got.cpu().numpy(), exp)
