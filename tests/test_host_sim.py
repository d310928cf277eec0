"""CPU logic checks of the PRODUCT headers (tr_math.h, tr_lbvh.h, tr_bvh.h) compiled with g++
by tests/host_sim: Karras hierarchy invariants and the stackless trail traversal must return
exactly what the oracle returns.  (The GPU runs the same functions per lane.)"""
import numpy as np
import pytest

import workloads as W
from oracle.oracle import OracleIntersector
from sim import SimBVH

Q_ANY, Q_FIRST, Q_CLOSEST, Q_COUNT = 0, 1, 2, 3


def check_tree(B):
    """every leaf reachable exactly once; parent/sibling links consistent; boxes nest"""
    nn, nf = len(B.nodes), B.nf
    assert nn == nf - 1
    c = B.nodes[:, 12:14].view(np.int32)
    par = B.nodes[:, 14].view(np.int32)
    sib = B.nodes[:, 15].view(np.int32)
    assert np.array_equal(B.links[:, 0], par) and np.array_equal(B.links[:, 1], sib)
    leaves = ~c[c < 0]
    assert np.array_equal(np.sort(leaves), np.arange(nf))
    internal = c[c >= 0]
    assert np.array_equal(np.sort(internal), np.arange(1, nn))
    assert par[0] == -1
    boxes = B.nodes[:, :12].view(np.float32)
    for k in (0, 1):
        ch = c[:, k]
        m = ch >= 0
        assert np.array_equal(par[ch[m]], np.flatnonzero(m))
        assert np.array_equal(sib[ch[m]], c[m, 1 - k])
        cb = boxes[ch[m]]
        lo = np.minimum(cb[:, 0:3], cb[:, 6:9])
        hi = np.maximum(cb[:, 3:6], cb[:, 9:12])
        assert np.array_equal(lo, boxes[m, 6 * k:6 * k + 3]) and np.array_equal(hi, boxes[m, 6 * k + 3:6 * k + 6])
    faces = B.tris[:, 9].view(np.int32)
    assert np.array_equal(np.sort(faces), np.arange(nf))
    # the 32-byte grid nodes of the unordered schedule: same topology, boxes that CONTAIN the exact
    # child boxes (bit-exact decode), nest, and are tight to one grid cell where float spacing allows
    assert np.array_equal(B.qnodes[:, 6:8].view(np.int32), c)
    qb = B.qchild_boxes()
    scale = B.frame[3:]
    for k in (0, 1):
        ex_lo = boxes[:, 6 * k:6 * k + 3]                           # stored lo.x lo.y lo.z | hi.z hi.x hi.y
        ex_hi = boxes[:, [6 * k + 4, 6 * k + 5, 6 * k + 3]]
        assert np.all(qb[:, k, :3] <= ex_lo) and np.all(qb[:, k, 3:] >= ex_hi)
        slack = np.maximum(scale, np.spacing(np.maximum(np.abs(ex_lo), np.abs(ex_hi)).astype(np.float32)) * 2)
        assert np.all(ex_lo - qb[:, k, :3] <= slack * 1.0001) and np.all(qb[:, k, 3:] - ex_hi <= slack * 1.0001)
        ch = c[:, k]
        m = ch >= 0
        cb = qb[ch[m]]
        assert np.all(qb[m, k, :3] <= np.minimum(cb[:, 0, :3], cb[:, 1, :3]))
        assert np.all(qb[m, k, 3:] >= np.maximum(cb[:, 0, 3:], cb[:, 1, 3:]))


def compare_all(v, f, o, d, **kw):
    B = SimBVH(v, f, **kw)
    if len(f) >= 2:
        check_tree(B)
        assert B.depth <= 64
    R = OracleIntersector(v, f, 1)
    h, fr, tri, loc, uv, t = R.closest_raw(o, d)
    r = B.query(Q_CLOSEST, o, d)
    assert np.array_equal(r["hit"], h.ravel()) and np.array_equal(r["front"], fr.ravel())
    assert np.array_equal(r["tri"], tri.ravel())
    assert np.array_equal(r["loc"], loc.reshape(-1, 3)) and np.array_equal(r["uv"], uv.reshape(-1, 2))
    cnt = R.intersects_count(o, d).ravel()
    assert np.array_equal(B.query(Q_COUNT, o, d)["count"], cnt)
    assert np.array_equal(B.query(Q_ANY, o, d)["hit"], cnt > 0)
    assert np.array_equal(B.query(Q_FIRST, o, d)["tri"], tri.ravel())
    if len(f) >= 2:
        c2, ltri, lt = B.location(o, d)
        _, _, tri_o, t_o = R.intersects_location(o, d, with_t=True)
        sel = np.arange(8)[None, :] < np.minimum(c2, 8)[:, None]
        assert np.array_equal(c2, cnt) and np.array_equal(ltri[sel], tri_o) and np.array_equal(lt[sel], t_o)
    return B


@pytest.mark.parametrize("ring", [True, False])
def test_icosphere_perspective_and_ortho(ring):
    import sim
    sim.use_ring(ring)
    try:
        v, f = W.icosphere(4)
        B = compare_all(v, f, *W.readme_perspective(160))
        compare_all(v, f, *W.ortho_grid(160))
        st = B.query(Q_CLOSEST, *W.ortho_grid(160))["stats"]
        assert (st[3] == 0) == ring or not ring     # with the ring, climbs only happen past 16 levels
    finally:
        sim.use_ring(True)


@pytest.mark.parametrize("kw", [dict(), dict(force_mode=1), dict(morton_shift=40), dict(morton_shift=63)])
def test_soup_key_modes(kw):
    v, f = W.random_soup(2500, seed=5)
    o, d = W.hash_rays(12000, 9, v.min(0) * 1.5, v.max(0) * 1.5)
    B = compare_all(v, f, o, d, **kw)
    if kw.get("morton_shift") == 63:
        assert B.depth <= 64   # all keys equal: hierarchy comes from the position tie-break


def test_height_fallback_to_bounded_keys():
    v, f = W.deep_tree_mesh(4000)
    B0 = SimBVH(v, f, force_mode=0)
    assert B0.depth > 64                       # plain Morton keys overflow the 64-bit trail
    B = SimBVH(v, f)
    assert B.key_mode == 1 and B.depth <= 64   # builder falls back to depth-bounded keys
    check_tree(B)
    o, d = W.hash_rays(2000, 3, [-0.1] * 3, [1.1] * 3)
    o[:500] = [1e-10, 1e-10, 1.0]
    d[:500] = [0, 0, -1]
    R = OracleIntersector(v, f, 1)
    assert np.array_equal(B.query(Q_COUNT, o, d)["count"], R.intersects_count(o, d))
    assert np.array_equal(B.query(Q_FIRST, o, d)["tri"], R.intersects_first(o, d))
    assert R.intersects_count(o[:1], d[:1])[0] >= 4000


@pytest.mark.parametrize("mode", [1, 2, 5, 6])
def test_fused_trip_variants(mode):
    """the kernels' fused trip (64-bit and compact 32-bit state; +4: over the 32-byte grid nodes with
    the full predicate at the leaves, as the streaming launch runs it) == oracle"""
    import sim
    sim.use_fused(mode)
    try:
        v, f = W.icosphere(5)
        compare_all(v, f, *W.pinhole_grid(128, 128))
        v, f = W.random_soup(2500, seed=8)
        o, d = W.hash_rays(12000, 4, v.min(0) * 1.5, v.max(0) * 1.5)
        compare_all(v, f, o, d)
        if mode in (1, 5):                       # deep trees need the 64-bit trail
            compare_all(v, f, o, d, morton_shift=63)
    finally:
        sim.use_fused(0)


@pytest.mark.parametrize("ring", [True, False])
def test_unordered_two_phase_schedule(ring):
    """any / count / location through tr_unord_step (leaves queued, tested late): same answers as
    the oracle on coherent, incoherent, multi-layer (> 8 hits: list replacement) and deep trees"""
    import sim
    sim.use_unordered(True)
    sim.use_ring(ring)
    try:
        v, f = W.icosphere(4)
        compare_all(v, f, *W.pinhole_grid(96, 96))
        v, f = W.random_soup(2500, seed=8)
        o, d = W.hash_rays(12000, 4, v.min(0) * 1.5, v.max(0) * 1.5)
        compare_all(v, f, o, d)
        compare_all(v, f, o, d, morton_shift=63)
        v, f = W.nested_shells(3, radii=(1.0, 0.8, 0.6, 0.4, 0.3))
        o, d = W.pinhole_grid(96, 96)
        compare_all(v, f, np.ascontiguousarray(o), d)
        v, f = W.deep_tree_mesh(3000)
        o, d = W.hash_rays(2000, 3, [-0.1] * 3, [1.1] * 3)
        o[:500] = [1e-10, 1e-10, 1.0]
        d[:500] = [0, 0, -1]
        compare_all(v, f, o, d)
    finally:
        sim.use_unordered(False)
        sim.use_ring(True)


def test_nested_shells_multihit():
    v, f = W.nested_shells(3, radii=(1.0, 0.8, 0.6, 0.4, 0.3))
    o, d = W.pinhole_grid(96, 96)
    compare_all(v, f, np.ascontiguousarray(o), d)


def test_tiny_meshes():
    v, f = W.two_triangles()
    o = np.array([[0, 0, -4], [0, 0.1, 4], [5, 5, 5]], np.float32)
    d = np.array([[0, 0, 1], [0, 0, -1], [0, 0, 1]], np.float32)
    compare_all(v, f, o, d)
    compare_all(v[:3], f[:1], o, d)


def test_golden_cube_boundary():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cube_axis_rays.npz"))
    B = SimBVH(g["vertices"], g["faces"])
    r = B.query(Q_CLOSEST, g["origins"], g["directions"])
    assert np.array_equal(r["hit"], g["hit"]) and np.array_equal(r["tri"], g["tri"])
    assert np.array_equal(B.query(Q_COUNT, g["origins"], g["directions"])["count"], g["count"])


# ---- round 5: the fused conservative box test (tr_ray_fuse / tr_qnode_slabs) ----------------------------------
def _nasty_rays(lo, hi, rng, n):
    """rays that stress the margins of the fused form: far and near origins, origins ON grid planes and mesh
    vertices, axis-aligned and nearly axis-aligned directions, huge / tiny direction components"""
    ext = np.maximum(hi - lo, 1e-6)
    c = (lo + hi) / 2
    o = rng.uniform(-1, 1, (n, 3)) * ext * rng.choice([0.4, 1.0, 3.0, 50.0, 1e4], (n, 1)) + c
    d = rng.normal(size=(n, 3))
    k = rng.integers(0, 8, n)
    ax = rng.integers(0, 3, n)
    idx = np.arange(n)
    z = (k == 0) | (k == 1)
    d[idx[z], ax[z]] = 0.0                                   # parallel to a coordinate plane (reciprocal clamped)
    z2 = k == 1
    d[idx[z2], (ax[z2] + 1) % 3] = 0.0                       # axis-aligned
    t = k == 2
    d[idx[t], ax[t]] *= 1e-30                                # beyond kmax, not zero
    t = k == 3
    d[idx[t], ax[t]] *= 1e-12
    t = k == 4
    d[t] *= 1e20
    t = k == 5
    d[t] *= 1e-20
    t = k == 6                                               # origin exactly on the box / in a bounding plane
    o[idx[t], ax[t]] = np.where(rng.random(t.sum()) < 0.5, lo[ax[t]], hi[ax[t]])
    return o.astype(np.float32), d.astype(np.float32)


@pytest.mark.parametrize("case", ["sphere", "soup", "far_from_origin", "flat", "huge", "tiny"])
def test_fused_box_test_contains_the_contracts(case):
    rng = np.random.default_rng(7)
    if case == "sphere":
        v, f = W.icosphere(4); v = W.displaced(v, seed=3, amplitude=0.1)
    elif case == "soup":
        v, f = W.random_soup(3000, seed=5, size=0.2)
    elif case == "far_from_origin":
        v, f = W.icosphere(3); v = (v * np.float32(0.01) + np.float32([1000.0, -2000.0, 512.25])).astype(np.float32)
    elif case == "flat":
        v, f = W.icosphere(3); v = v.copy(); v[:, 2] = np.float32(0.25)          # a degenerate axis
    elif case == "huge":
        v, f = W.icosphere(3); v = (v * np.float32(3e12)).astype(np.float32)
    else:
        v, f = W.icosphere(3); v = (v * np.float32(1e-12) + np.float32(1e-9)).astype(np.float32)
    B = SimBVH(v, f)
    lo, hi = v.min(0).astype(np.float64), v.max(0).astype(np.float64)
    o, d = _nasty_rays(lo, hi, rng, 4000)
    # plus rays that start on mesh vertices and on decoded grid planes
    qb = B.qchild_boxes().reshape(-1, 6)
    pick = rng.integers(0, len(qb), 500)
    o2 = np.where(rng.random((500, 3)) < 0.5, qb[pick, :3], qb[pick, 3:]).astype(np.float32)
    d2 = rng.normal(size=(500, 3)).astype(np.float32)
    d2[rng.random(500) < 0.3, 0] = 0.0
    o = np.concatenate([o, o2, v[rng.integers(0, len(v), 500)]]); d = np.concatenate([d, d2, rng.normal(size=(500, 3)).astype(np.float32)])
    bad, pairs, acc_fused, acc_contract = B.check_fused(o, d, node_stride=3)
    assert pairs > 1_000_000
    assert bad == 0
    assert acc_fused >= acc_contract
    badw, pairsw, fw, cw = B.check_fused_wide(o, d, node_stride=2)
    assert pairsw > 300_000 and badw == 0 and fw >= cw
    if case in ("sphere", "soup"):
        # the price of one fma per plane on ordinary rays (a camera within three extents, aimed at the mesh): the fused
        # test accepts well under one per cent more children than the contract's
        ext = hi - lo
        oo = ((lo + hi) / 2 + rng.normal(size=(3000, 3)) / np.linalg.norm(rng.normal(size=(3000, 3)), axis=1, keepdims=True) * 0 +
              rng.uniform(-3, 3, (3000, 3)) * ext).astype(np.float32)
        dd = ((lo + hi) / 2 + rng.uniform(-0.5, 0.5, (3000, 3)) * ext - oo).astype(np.float32)
        bad2, pairs2, f2, c2 = B.check_fused(oo, dd, node_stride=1)
        assert bad2 == 0 and f2 >= c2
        assert f2 <= c2 * 1.005 + 10, (f2, c2)
        bad3, _, f3, c3 = B.check_fused_wide(oo, dd, node_stride=1)
        assert bad3 == 0 and c3 <= f3 <= c3 * 1.01 + 10, (f3, c3)


def test_fused_box_test_keeps_results_on_axis_aligned_rays():
    """orthographic rays along an axis onto an axis-aligned box whose faces lie in grid planes, with ray origins IN
    the planes of the side faces: the clamped reciprocals and the 2^-100 pad of the contract, through the grid-node
    flavours of the fused trip (64- and 32-bit state) and the unordered schedule"""
    import sim
    v, f = W._box((0.0, 0.0, 0.0), (1.0, 1.0, 0.5), 6)
    v = v.astype(np.float32); f = f.astype(np.int32)
    o, d = W.ortho_grid(49)                     # linspace(-1, 1, 49) contains -1, 0 and 1 exactly
    o = np.ascontiguousarray(o).reshape(-1, 3); d = np.ascontiguousarray(d).reshape(-1, 3)
    for mode in (5, 6):
        sim.use_fused(mode)
        try:
            compare_all(v, f, o, d)
        finally:
            sim.use_fused(0)
    sim.use_unordered(True)
    try:
        compare_all(v, f, o, d)
    finally:
        sim.use_unordered(False)


def _adversarial_pairs(rng, n):
    """(origin, direction, triangle) triples built to sit ON the decision boundaries of the inside test: the ray is aimed
    at a point of the triangle's plane at a signed distance from an edge / a vertex that sweeps 1e-10 ... 1 of the
    triangle's size (both sides), from cameras 1 ... 3e4 sizes away, at incidence angles down to 3e-4 rad, on triangles
    with aspect ratios up to 1e4, at coordinate scales 1e-5 ... 1e7 (lengths beyond 2^40 and products that overflow
    float32 included) and offsets of up to 1e3 extents from the origin"""
    a = rng.standard_normal((n, 3))
    e1 = rng.standard_normal((n, 3))
    e2 = rng.standard_normal((n, 3)) * (10.0 ** rng.uniform(-4, 0, (n, 1)))       # needles
    b, c = a + e1, a + e2
    # a target: on edge a-b (u in [0, 1]) or at a vertex, displaced ACROSS the edge by eps * size
    u = rng.random((n, 1))
    u[rng.random(n) < 0.25] = 0.0
    nrm = np.cross(e1, e2)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True) + 1e-300
    across = np.cross(nrm, e1)
    across /= np.linalg.norm(across, axis=1, keepdims=True) + 1e-300
    size = np.linalg.norm(e1, axis=1, keepdims=True)
    eps = (10.0 ** rng.uniform(-10, 0, (n, 1))) * rng.choice([-1.0, 1.0], (n, 1))
    eps[rng.random(n) < 0.1] = 0.0
    target = a + u * e1 + eps * size * across
    # the eye: distance 1 ... 1e6 sizes, elevation down to 1e-4 rad above the plane
    dist = size * 10.0 ** rng.uniform(0, 4.5, (n, 1))
    elev = 10.0 ** rng.uniform(-3.5, 0.19, (n, 1))
    inplane = rng.standard_normal((n, 3))
    inplane -= nrm * np.sum(inplane * nrm, axis=1, keepdims=True)
    inplane /= np.linalg.norm(inplane, axis=1, keepdims=True) + 1e-300
    eye = target + dist * (np.cos(elev) * inplane + np.sin(elev) * nrm * rng.choice([-1.0, 1.0], (n, 1)))
    scale = 10.0 ** rng.uniform(-5, 7, (n, 1))
    shift = rng.standard_normal((n, 3)) * (10.0 ** rng.uniform(-3, 3, (n, 1))) * (rng.random((n, 1)) < 0.5)
    tri = np.concatenate([(x + shift) * scale for x in (a, b, c)], axis=1).astype(np.float32)
    eye32 = ((eye + shift) * scale).astype(np.float32)
    tgt32 = ((target + shift) * scale).astype(np.float32)
    dirs = (tgt32 - eye32).astype(np.float32) * rng.choice([1.0, 0.37, 1e-3, 250.0], (n, 1)).astype(np.float32)   # un-normalised
    return eye32, dirs.astype(np.float32), tri


def test_the_float32_inside_test_is_only_ever_right_when_it_answers():
    """VERDICT r05 "next" #1 ("prove the margin ... on adversarial rays"): wherever the float32 part of the predicate
    (tr_tri_fast: Moller-Trumbore + the running error bound) ANSWERS -- proven hit or proven miss -- the float64 edge
    functions give the same answer, an independent long-double evaluation of the exact triple products agrees, and its
    distance is within 2^-11 of the float64 one (the slack TR_CULL_SLACK grants); everything else it leaves UNDECIDED.
    4 M pairs that sit on the decision boundaries (see _adversarial_pairs) + 1 M unrelated random pairs."""
    import sim
    rng = np.random.default_rng(2026)
    o, d, tri = _adversarial_pairs(rng, 4_000_000)
    o2 = rng.standard_normal((1_000_000, 3)).astype(np.float32) * 3
    d2 = rng.standard_normal((1_000_000, 3)).astype(np.float32)
    t2 = (rng.standard_normal((1_000_000, 9)) * 0.7).astype(np.float32)
    o, d, tri = np.concatenate([o, o2]), np.concatenate([d, d2]), np.concatenate([tri, t2])
    code, tf, he, te, truth = sim.tri_fast_vs_exact(o, d, tri)
    decided = code != 2
    # a proven hit is a hit of the exact part and of the long-double arbiter
    hit = code == 1
    assert np.all(he[hit] == 1), int((he[hit] != 1).sum())
    assert np.all(truth[hit] != 0), int((truth[hit] == 0).sum())
    # its distance: within 2^-11 relative of the float64 distance
    rel = np.abs(tf[hit].astype(np.float64) - te[hit].astype(np.float64)) / np.maximum(np.abs(te[hit].astype(np.float64)), 1e-300)
    assert rel.max(initial=0.0) < 2.0 ** -11, float(rel.max())
    # a proven miss is a miss of the exact part -- unless the miss is about the RANGE of t (behind the origin / beyond 1e7),
    # where both parts see a hit of the line and disagree at most within the slack at the two ends of the interval
    miss = code == 0
    line_hit = miss & (he == 1)
    tl = te[line_hit].astype(np.float64)
    assert np.all((tl < 1e-30) | (tl > 1e7 * (1 - 2.0 ** -10))), "a proven miss that the exact part counts as a hit"
    inside_says_out = miss & (truth == 1)
    # (truth == 1 with a miss: the line hits, the range check said no: the exact part must have said the same about t)
    assert np.all(he[inside_says_out & ~line_hit] == 0)
    # the corpus has bite: a good part of it is undecided or close to it, and all three answers occur
    frac_und = 1.0 - decided[:4_000_000].mean()
    assert 0.05 < frac_und < 0.95, frac_und
    assert hit.sum() > 50_000 and miss.sum() > 200_000
    # ordinary pairs are decided in float32 almost always
    assert (code[4_000_000:] == 2).mean() < 2e-3
