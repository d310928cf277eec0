"""CPU tests of the oracle (test infrastructure): known answers of the reference's own test
inputs, golden fixtures, brute force == oracle BVH, strided fetch."""
import glob
import json
import os

import numpy as np
import pytest

import workloads as W
from oracle.oracle import OracleIntersector, fetch_rays

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KA = json.load(open(os.path.join(GOLD, "known_answers.json")))


@pytest.mark.parametrize("mode", [0, 1])
def test_T1_single_triangle(mode):
    k = KA["T1"]
    R = OracleIntersector(np.array(k["vertices"], np.float32), np.array(k["faces"], np.int32), mode)
    o, d = np.array(k["origins"], np.float32), np.array(k["directions"], np.float32)
    assert R.intersects_any(o, d).tolist() == k["any"]
    assert R.intersects_first(o, d).tolist() == k["first"]
    hit, front, tri, loc, uv = R.intersects_closest(o, d)
    c = k["closest"]
    assert hit.tolist() == c["hit"] and front.tolist() == c["front"] and tri.tolist() == c["tri"]
    np.testing.assert_allclose(loc, c["loc"], atol=1e-7)
    np.testing.assert_allclose(uv, c["uv"], atol=1e-7)


@pytest.mark.parametrize("mode", [0, 1])
def test_T2_two_triangles(mode):
    k = KA["T2"]
    R = OracleIntersector(np.array(k["vertices"], np.float32), np.array(k["faces"], np.int32), mode)
    o, d = np.array(k["origins"], np.float32), np.array(k["directions"], np.float32)
    assert R.intersects_count(o, d).tolist() == k["count"]
    loc, ray, tri = R.intersects_location(o, d)
    assert ray.tolist() == k["location"]["ray_idx"]
    got = {}
    for l, r, t in zip(loc, ray, tri):
        got.setdefault(int(r), []).append((int(t), l))
    for r, exp in enumerate(k["location"]["per_ray"]):
        assert sorted(t for t, _ in got[r]) == sorted(t for t, _ in exp)
        for t, l in exp:
            np.testing.assert_allclose([g for tt, g in got[r] if tt == t][0], l, atol=1e-6)
    hit, front, tri, loc, uv = R.intersects_closest(o, d)
    c = k["closest"]
    assert hit.tolist() == c["hit"] and front.tolist() == c["front"] and tri.tolist() == c["tri"]
    np.testing.assert_allclose(uv, c["uv"], atol=1e-6)
    # intersects_id conventions (ray_optix.py:207-223)
    t2, r2 = R.intersects_id(o, d)
    assert np.array_equal(t2, R.intersects_location(o, d)[2])
    t1, r1, l1 = R.intersects_id(o, d, return_locations=True, multiple_hits=False)
    assert t1.tolist() == c["tri"] and r1.tolist() == [0, 1]


def test_T3_contains_points():
    v, f = W.icosphere(3)
    R = OracleIntersector(v, f, 0)
    assert R.contains_points(np.array(KA["T3"]["points"], np.float32)).tolist() == KA["T3"]["contains"]
    pts = np.array([[0, 0, 0], [0.5, 0.2, 0.1], [0, 0, 1.01], [2, 2, 2], [0.9, 0.9, 0.9]], np.float32)
    assert R.contains_points(pts).tolist() == [True, True, False, False, False]


def test_T4_readme_scene():
    v, f = W.icosphere(3)
    o, d = W.readme_perspective(800)
    R = OracleIntersector(v, f, 1)
    hit, front, ray_idx, tri, loc, uv = R.intersects_closest(o, d, stream_compaction=True)
    ys, xs = np.nonzero(hit)
    rad = np.sqrt((ys - 399.5) ** 2 + (xs - 399.5) ** 2).max()
    assert 139.0 < rad <= KA["T4"]["max_hit_radius_px"]
    assert front.all() and (loc[:, 2] > 0).all()
    assert np.array_equal(ray_idx, np.flatnonzero(hit.reshape(-1)).astype(np.int32))
    # uv are weights of vertices 0 and 1 (shaders.cu:149; consumer test/test.py:41)
    vv = v[f[tri]]
    rec = uv[:, :1] * vv[:, 0] + uv[:, 1:] * vv[:, 1] + (1 - uv[:, :1] - uv[:, 1:]) * vv[:, 2]
    np.testing.assert_allclose(rec, loc, atol=2e-6)


def test_icosphere_is_outward_wound():
    v, f = W.icosphere(2)
    assert v.shape == (162, 3) and f.shape == (320, 3)
    n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    assert (np.einsum("ij,ij->i", n, v[f].mean(1)) > 0).all()
    v1, f1 = W.icosphere(1)
    assert len(f1) == 80 and len(v1) == 42     # BASELINE.json config 1


@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(GOLD, "*.npz")) if not os.path.basename(p).startswith("reference_")))
@pytest.mark.parametrize("mode", [0, 1])
def test_golden_fixtures(path, mode):
    g = np.load(path)
    R = OracleIntersector(g["vertices"], g["faces"], mode)
    hit, front, tri, loc, uv, t = R.closest_raw(g["origins"], g["directions"])
    assert np.array_equal(hit, g["hit"]) and np.array_equal(front, g["front"])
    assert np.array_equal(tri, g["tri"]) and np.array_equal(t, g["t"])
    assert np.array_equal(loc, g["loc"]) and np.array_equal(uv, g["uv"])
    assert np.array_equal(R.intersects_count(g["origins"], g["directions"]), g["count"])
    lloc, lray, ltri, lt = R.intersects_location(g["origins"], g["directions"], with_t=True)
    assert np.array_equal(lray, g["loc_ray"]) and np.array_equal(ltri, g["loc_tri"])
    assert np.array_equal(lloc, g["loc_loc"]) and np.array_equal(lt, g["loc_t"])


def test_cube_boundary_rays_see_a_half_open_square():
    """Axis-parallel rays on a grid that contains the cube's silhouette lines x, y = +-1 exactly (exact ties: the ray runs
    through an edge or a vertex).  With the ownership rule of contract 3 (tr_math.h "exact ties") the cube is, seen along
    such a ray, a HALF-OPEN square -- the rasteriser's fill rule: the interior, exactly one of each pair of opposite boundary
    lines, exactly one of the four corners -- so that cubes side by side would cover every ray exactly once; and every ray
    that hits crosses the closed surface exactly twice (the first form, "zero counts as inside", counted 4 on 34 of them)."""
    g = np.load(os.path.join(GOLD, "cube_axis_rays.npz"))
    for sl in (slice(0, 625), slice(625, 1250)):
        o, d, hit, cnt = g["origins"][sl], g["directions"][sl], g["hit"][sl], g["count"][sl]
        ax = [i for i in range(3) if d[0, i] == 0]
        x, y = o[:, ax[0]], o[:, ax[1]]
        assert hit[(np.abs(x) < 1) & (np.abs(y) < 1)].all() and not hit[(np.abs(x) > 1) | (np.abs(y) > 1)].any()
        for u, w in ((x, y), (y, x)):
            lo, hi = hit[(u == -1) & (np.abs(w) < 1)], hit[(u == 1) & (np.abs(w) < 1)]
            assert len(lo) == len(hi) == 15 and (lo.all() != hi.all()) and (lo.all() or not lo.any()) and (hi.all() or not hi.any())
        assert int(hit[(np.abs(x) == 1) & (np.abs(y) == 1)].sum()) == 1
        assert set(np.unique(cnt)) == {0, 2} and np.array_equal(cnt > 0, hit)


def test_rays_through_the_vertices_of_a_grid_mesh_count_once():
    """Exact ties as the ordinary case: a height field on an integer grid under vertical rays through EVERY vertex, edge
    midpoint and cell centre.  One owner per edge (tr_math.h "exact ties"): every ray over the patch's interior crosses the
    surface exactly once -- the first form of contract 3 counted 2 on the edges and up to 6 at the vertices --, the open
    patch is half-open at its boundary (one boundary row and one boundary column belong to it, like a rasterised quad), and
    rays aimed in float32 exactly at vertices and edge points of a closed mesh from inside count exactly once."""
    g = np.arange(-8, 9, dtype=np.float32)
    X, Y = np.meshgrid(g, g, indexing="ij")
    Z = ((X * 7 + Y * 3) % 5).astype(np.float32) * 0.25
    v = np.stack([X, Y, Z], -1).reshape(-1, 3)
    idx = np.arange(17 * 17).reshape(17, 17)
    a, b, c, dd = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    f = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, dd], 1)]).astype(np.int32)
    h = np.arange(-8, 8.01, 0.5, dtype=np.float32)
    gx, gy = np.meshgrid(h, h, indexing="ij")
    o = np.stack([gx, gy, np.full_like(gx, 10)], -1).reshape(-1, 3)
    d = np.tile(np.array([[0, 0, -1]], np.float32), (len(o), 1))
    for mode in (0, 1):
        cnt = OracleIntersector(v, f, mode).intersects_count(o, d).reshape(gx.shape)
        assert set(np.unique(cnt)) == {0, 1} and (cnt[1:-1, 1:-1] == 1).all()
        for edge_lo, edge_hi in ((cnt[0, 1:-1], cnt[-1, 1:-1]), (cnt[1:-1, 0], cnt[1:-1, -1])):
            assert {int(edge_lo.min()), int(edge_lo.max()), int(edge_hi.min()), int(edge_hi.max())} == {0, 1} and edge_lo.min() == edge_lo.max() != edge_hi.max()
        assert int(cnt[0, 0] + cnt[0, -1] + cnt[-1, 0] + cnt[-1, -1]) == 1
    v, f = W.icosphere(3)
    v = W.displaced(v, seed=3, amplitude=0.08)
    R = OracleIntersector(v, f, 1)
    rng = np.random.default_rng(5)
    eye = (rng.normal(size=(60000, 3)) * 0.05).astype(np.float32)
    e = rng.integers(0, len(f), 30000)
    targets = np.concatenate([v[rng.integers(0, len(v), 30000)], (v[f[e, 0]] + np.float32(0.5) * (v[f[e, 1]] - v[f[e, 0]])).astype(np.float32)])
    assert (R.intersects_count(eye, (targets - eye).astype(np.float32)) == 1).all()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_brute_equals_bvh_random(seed):
    v, f = W.random_soup(600, seed=seed)
    o, d = W.hash_rays(6000, 100 + seed, v.min(0) * 1.5, v.max(0) * 1.5)
    a = OracleIntersector(v, f, 0)
    b = OracleIntersector(v, f, 1)
    for x, y in zip(a.closest_raw(o, d), b.closest_raw(o, d)):
        assert np.array_equal(x, y)
    assert np.array_equal(a.intersects_count(o, d), b.intersects_count(o, d))
    for x, y in zip(a.intersects_location(o, d, with_t=True), b.intersects_location(o, d, with_t=True)):
        assert np.array_equal(x, y)


def test_location_cap_and_order():
    v, f = W.nested_shells(2, radii=(1.0, 0.8, 0.6, 0.4, 0.3))
    R = OracleIntersector(v, f, 1)
    o = np.array([[0.01, 0.02, 3]], np.float32)
    d = np.array([[0, 0, -1]], np.float32)
    assert R.intersects_count(o, d).tolist() == [10]
    loc, ray, tri, t = R.intersects_location(o, d, with_t=True)
    assert len(ray) == 8 and (np.diff(t) >= 0).all()        # clamp to MAX_ANYHIT_SIZE, nearest first
    assert abs(loc[0, 2] - 1.0) < 0.02 and loc[-1, 2] < 0   # starts at the outer shell


def test_invalid_and_degenerate():
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [2, 2, 2], [2, 2, 2], [2, 2, 2]], np.float32)
    f = np.array([[0, 1, 2], [3, 4, 5]], np.int32)          # second triangle is degenerate
    R = OracleIntersector(v, f, 0)
    o = np.array([[0.2, 0.2, 1], [np.nan, 0, 1], [0.2, 0.2, 1], [2, 2, 3], [0.2, 0.2, 1]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, -1], [0, 0, np.inf], [0, 0, -1], [0, 0, 0]], np.float32)
    hit = R.intersects_any(o, d)
    assert hit.tolist() == [True, False, False, False, False]
    # tmax = 1e7 (shaders.cu:86): a triangle farther than that is not hit
    o = np.array([[0.2, 0.2, 2e7], [0.2, 0.2, 9e6]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, -1]], np.float32)
    assert R.intersects_any(o, d).tolist() == [False, True]


def test_tmax_is_measured_from_the_origin_also_for_rays_that_would_be_anchored():
    """Ray anchoring (tr_ray_anchor) moves a far origin to the mesh's box and measures distances from there -- which must not
    lengthen the reference's interval [0, 1e7] (shaders.cu:86): a ray is only anchored when the WHOLE box lies within tmax of
    its origin.  A mesh 5e6 ... 1.0001e7 units down the ray: the triangle at 9.9e6 is hit, the one at 1.00005e7 is not (the
    first form of the anchor, `tn < 1e7`, started counting at 5e6 and hit it at t' = 5e6); brute force == BVH walk."""
    sq = lambda x: [[x, -4, -4], [x, 4, -4], [x, 0, 4]]  # noqa: E731
    v = np.array(sq(5.0e6) + sq(9.9e6) + sq(1.00005e7) + sq(1.0001e7), np.float32)
    f = np.arange(12, dtype=np.int32).reshape(4, 3)
    o = np.array([[0, 0, 0], [5.5e6, 0, 0], [9.95e6, 0, 0], [0, 0, 0]], np.float32)
    d = np.array([[1, 0, 0], [1, 0, 0], [1, 0, 0], [2, 0, 0]], np.float32)
    for mode in (0, 1):
        R = OracleIntersector(v, f, mode)
        assert R.intersects_count(o, d).tolist() == [2, 3, 2, 4]         # (direction of length 2: t = distance / 2, all four within 1e7)
        hit, front, tri, loc, uv = R.intersects_closest(o, d)[:5]
        assert tri.tolist() == [0, 1, 2, 0]
    # a mesh that does lie within tmax IS anchored, with the same answers as from the far origin in float64
    v2 = np.array(sq(5.0e6) + sq(5.0e6 + 8), np.float32)
    R = OracleIntersector(v2, np.arange(6, dtype=np.int32).reshape(2, 3), 1)
    oa = R.anchor(o[:1], d[:1])
    assert oa[0, 0] > 4.9e6 and R.intersects_count(o[:1], d[:1]).tolist() == [2]


def test_fetch_rays_strides():
    rng = np.random.default_rng(0)
    base = rng.random((5, 7, 6)).astype(np.float32)
    d = base[:, :, ::2]                                    # last-dim stride 2
    o = np.broadcast_to(np.array([1, 2, 3], np.float32), d.shape)   # stride-0 origin
    oo, dd = fetch_rays(o, d)
    assert np.array_equal(dd, d.reshape(-1, 3)) and np.array_equal(oo, np.tile([1, 2, 3], (35, 1)))
    t = base.transpose(1, 0, 2)[:, :, 3:]                  # permuted leading dims
    oo, dd = fetch_rays(t, t)
    assert np.array_equal(oo, np.ascontiguousarray(t).reshape(-1, 3))
    four = rng.random((2, 3, 4, 3)).astype(np.float32)
    oo, _ = fetch_rays(four, four)
    assert np.array_equal(oo, four.reshape(-1, 3))


def test_oracle_reproduces_the_reference_s_published_readme_image():
    """The one OUTPUT artefact the reference publishes for this path: assets/location.png, its own OptiX
    path's location map of the README quick-start (README.md:31-53), kept as a fixture
    (tests/golden/reference_readme_location_axes.npz).  The oracle's location map of the same inputs,
    rendered the way `plt.imshow(locs)` renders it, must reproduce that image: silhouette to a fraction
    of a pixel, location values to one 8-bit level on average.  A pin at image precision -- the only
    pin on reference-produced data that exists; bit-level parity stays unpinned."""
    pytest.importorskip("PIL")
    from readme_image import assert_matches_reference_image, compare_with_reference_image
    v, f = W.icosphere(3)                         # trimesh.creation.icosphere() default: 1 280 faces
    o, d = W.readme_perspective(800)
    R = OracleIntersector(v, f, 1)
    hit, front, ridx, tri, loc, uv = R.intersects_closest(np.ascontiguousarray(o.reshape(-1, 3)), d.reshape(-1, 3),
                                                          stream_compaction=True)
    locs = np.zeros((800 * 800, 3), np.float32)
    locs[hit.reshape(-1)] = loc                   # README.md:49-50
    m = compare_with_reference_image(locs.reshape(800, 800, 3))
    assert_matches_reference_image(m)
    # the comparison has bite: a 4 % error in the locations, or a silhouette one pixel too large, fails it
    with pytest.raises(AssertionError):
        assert_matches_reference_image(compare_with_reference_image((locs * 0.96).reshape(800, 800, 3)))
    grown = locs.reshape(800, 800, 3).copy()
    mask = hit.reshape(800, 800)
    edge = np.zeros_like(mask)
    edge[1:-1, 1:-1] = (mask[:-2, 1:-1] | mask[2:, 1:-1] | mask[1:-1, :-2] | mask[1:-1, 2:] | mask[:-2, :-2] | mask[2:, 2:]) & ~mask[1:-1, 1:-1]
    for _ in range(3):
        grown[edge] = (0.0, 0.0, 1.0)
        mask = mask | edge
        edge = np.zeros_like(mask)
        edge[1:-1, 1:-1] = (mask[:-2, 1:-1] | mask[2:, 1:-1] | mask[1:-1, :-2] | mask[1:-1, 2:]) & ~mask[1:-1, 1:-1]
    with pytest.raises(AssertionError):
        assert_matches_reference_image(compare_with_reference_image(grown))
