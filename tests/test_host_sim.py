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
