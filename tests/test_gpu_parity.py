"""GPU parity tests: the HIP path (through the C ABI, via the reference-shaped Python class)
against the CPU oracle on identical inputs.  Bar: hit / front / tri_idx / ray_idx / count
bit-exact; loc / uv within 1e-5 relative (BASELINE.json north_star) -- in practice they are
bit-exact too because the arithmetic contract fixes every operation."""
import glob
import os

import numpy as np
import pytest
import torch

import workloads as W
from oracle.oracle import OracleIntersector

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL, ATOL = 1e-5, 1e-6


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def make(v, f, dev):
    from triro.ray.ray_optix import RayMeshIntersector
    return RayMeshIntersector(vertices=T(v, dev), faces=T(f, dev))


def assert_closest_equal(got, exp):
    hit, front, tri, loc, uv = [g.cpu().numpy() for g in got]
    eh, ef, et, el, eu = exp[:5]
    assert np.array_equal(hit, eh), f"hit mask: {np.sum(hit != eh)} rays differ"
    assert np.array_equal(front, ef)
    assert np.array_equal(tri, et), f"tri_idx: {np.sum(tri != et)} rays differ"
    np.testing.assert_allclose(loc, el, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(uv, eu, rtol=RTOL, atol=ATOL)


def full_compare(v, f, o, d, dev, mode=1):
    """all five queries + location on one (mesh, rays) pair"""
    r = make(v, f, dev)
    R = OracleIntersector(v, f, mode)
    ot, dt = T(o, dev), T(d, dev)
    assert_closest_equal(r.intersects_closest(ot, dt), R.closest_raw(o, d))
    cnt = R.intersects_count(o, d)
    assert np.array_equal(r.intersects_count(ot, dt).cpu().numpy(), cnt)
    assert np.array_equal(r.intersects_any(ot, dt).cpu().numpy(), cnt > 0)
    assert np.array_equal(r.intersects_first(ot, dt).cpu().numpy(), R.intersects_first(o, d))
    loc, ray, tri = [x.cpu().numpy() for x in r.intersects_location(ot, dt)]
    el, er, et = R.intersects_location(o, d)
    assert np.array_equal(ray, er) and np.array_equal(tri, et)
    np.testing.assert_allclose(loc, el, rtol=RTOL, atol=ATOL)
    return r, R


@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(GOLD, "*.npz")) if not os.path.basename(p).startswith("reference_")))
def test_golden_fixtures(path, device):
    g = np.load(path)
    r = make(g["vertices"], g["faces"], device)
    o, d = T(g["origins"], device), T(g["directions"], device)
    hit, front, tri, loc, uv = [x.cpu().numpy() for x in r.intersects_closest(o, d)]
    assert np.array_equal(hit, g["hit"]) and np.array_equal(front, g["front"]) and np.array_equal(tri, g["tri"])
    np.testing.assert_allclose(loc, g["loc"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(uv, g["uv"], rtol=RTOL, atol=ATOL)
    assert np.array_equal(loc, g["loc"]) and np.array_equal(uv, g["uv"]), "contract promises bit-exact floats"
    assert np.array_equal(r.intersects_count(o, d).cpu().numpy(), g["count"])
    lloc, lray, ltri = [x.cpu().numpy() for x in r.intersects_location(o, d)]
    assert np.array_equal(lray, g["loc_ray"]) and np.array_equal(ltri, g["loc_tri"])
    np.testing.assert_allclose(lloc, g["loc_loc"], rtol=RTOL, atol=ATOL)


def test_known_answers_T1_T2(device):
    import json
    ka = json.load(open(os.path.join(GOLD, "known_answers.json")))
    k = ka["T1"]
    r = make(np.array(k["vertices"], np.float32), np.array(k["faces"], np.int32), device)   # single triangle
    o, d = T(np.array(k["origins"], np.float32), device), T(np.array(k["directions"], np.float32), device)
    assert r.intersects_any(o, d).tolist() == k["any"]
    assert r.intersects_first(o, d).tolist() == k["first"]
    hit, front, tri, loc, uv = r.intersects_closest(o, d)
    assert hit.tolist() == k["closest"]["hit"] and front.tolist() == k["closest"]["front"]
    assert tri.tolist() == k["closest"]["tri"]
    np.testing.assert_allclose(uv.cpu().numpy(), k["closest"]["uv"], atol=1e-7)
    assert r.intersects_count(o, d).tolist() == [1, 0]
    loc, ray, tri = r.intersects_location(o, d)
    assert ray.tolist() == [0] and tri.tolist() == [0]
    k = ka["T2"]
    r = make(np.array(k["vertices"], np.float32), np.array(k["faces"], np.int32), device)
    o, d = T(np.array(k["origins"], np.float32), device), T(np.array(k["directions"], np.float32), device)
    assert r.intersects_count(o, d).tolist() == k["count"]
    loc, ray, tri = r.intersects_location(o, d)
    assert ray.tolist() == k["location"]["ray_idx"] and tri.tolist() == [1, 0, 0, 1]
    hit, front, tri, loc, uv = r.intersects_closest(o, d)
    assert front.tolist() == k["closest"]["front"] and tri.tolist() == k["closest"]["tri"]


def test_readme_quickstart(device):
    """README.md:24-51 / test/test.py:15-28 with stride-0 origins and stream compaction."""
    from triro.ray.ray_optix import RayMeshIntersector

    class Mesh:   # stands in for trimesh.Trimesh (mesh= path, ray_optix.py:25-31)
        pass
    m = Mesh()
    m.vertices, m.faces = W.icosphere(3)
    m.vertices = m.vertices.astype(np.float64)   # trimesh hands over float64 / int64
    m.faces = m.faces.astype(np.int64)
    r = RayMeshIntersector(mesh=m)
    y, x = torch.meshgrid([torch.linspace(1, -1, 800), torch.linspace(-1, 1, 800)], indexing="ij")
    z = -torch.ones_like(x)
    dirs = torch.stack([x, y, z], dim=-1).to(device)
    origins = torch.tensor([0, 0, 3], dtype=torch.float32, device=device).broadcast_to(dirs.shape)
    hit, front, ray_idx, tri_idx, location, uv = r.intersects_closest(origins, dirs, stream_compaction=True)
    R = OracleIntersector(m.vertices, m.faces, 1)
    eh, ef, er, et, el, eu = R.intersects_closest(origins.cpu().numpy(), dirs.cpu().numpy(), stream_compaction=True)
    assert hit.shape == (800, 800) and hit.dtype == torch.bool and ray_idx.dtype == torch.int32
    assert np.array_equal(hit.cpu().numpy(), eh) and np.array_equal(front.cpu().numpy(), ef)
    assert np.array_equal(ray_idx.cpu().numpy(), er) and np.array_equal(tri_idx.cpu().numpy(), et)
    np.testing.assert_allclose(location.cpu().numpy(), el, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(uv.cpu().numpy(), eu, rtol=RTOL, atol=ATOL)
    locs = torch.zeros((800, 800, 3), device=device)
    locs[hit] = location                                  # README.md:47-48 usage
    assert front.all() and (location[:, 2] > 0).all()


def test_c1_icosphere80_ortho800(device):
    """BASELINE.json config 1 on the GPU path: 80 tris, 800x800 ortho rays, brute-force oracle."""
    v, f = W.icosphere(1)
    o, d = W.ortho_grid(800)
    r = make(v, f, device)
    R = OracleIntersector(v, f, 0)
    assert_closest_equal(r.intersects_closest(T(o, device), T(d, device).contiguous()), R.closest_raw(o, d))


def test_all_queries_soup_incoherent(device):
    v, f = W.random_soup(20000, seed=11)
    o, d = W.hash_rays(200000, 1234, v.min(0) * 1.5, v.max(0) * 1.5)
    full_compare(v, f, o, d, device)


def test_all_queries_bunny_standin_pinhole(device):
    """BASELINE.json config 2 shape (reduced grid for the oracle): stand-in mesh, pinhole rays."""
    v, f = W.bunny_standin()
    o, d = W.pinhole_grid(512, 512, distance=2.5 * 1.12)
    full_compare(v, f, o, d, device)


def test_nested_shells_multihit_cap(device):
    """BASELINE.json config 4 shape: > 8 hits on central rays -> clamp to MAX_ANYHIT_SIZE."""
    v, f = W.nested_shells(5, radii=(1.0, 0.8, 0.6, 0.4, 0.3))
    o, d = W.pinhole_grid(256, 256)
    r, R = full_compare(v, f, o, d, device)
    cnt = r.intersects_count(T(o, device), T(d, device))
    assert int(cnt.max()) == 10
    loc, ray, tri = r.intersects_location(T(o, device), T(d, device))
    per_ray = torch.bincount(ray.long(), minlength=cnt.numel())
    assert int(per_ray.max()) == 8
    assert torch.equal(per_ray, torch.clamp(cnt.reshape(-1), max=8).long())
    tri2, ray2, loc2 = r.intersects_id(T(o, device), T(d, device), return_locations=True)
    assert torch.equal(tri2, tri) and torch.equal(ray2, ray) and torch.equal(loc2, loc)
    tri1, ray1 = r.intersects_id(T(o, device), T(d, device), multiple_hits=False)
    hit, _, tric, _, _ = r.intersects_closest(T(o, device), T(d, device))
    assert torch.equal(tri1, tric[hit]) and torch.equal(ray1.long(), torch.nonzero(hit.reshape(-1))[:, 0])


def test_strided_and_batched_inputs(device):
    v, f = W.icosphere(4)
    r = make(v, f, device)
    R = OracleIntersector(v, f, 1)
    rng = np.random.default_rng(1)
    base = (rng.random((6, 9, 11, 6)).astype(np.float32) * 2 - 1)
    bt = T(base, device)
    # (a) 3 batch dims, last-dim stride 2, offset view; origins and directions from different views
    d_t, o_t = bt[..., ::2], bt[..., 3:] * 3.0
    d_n, o_n = base[..., ::2], base[..., 3:] * 3.0
    assert not d_t.is_contiguous()
    got = r.intersects_closest(o_t.contiguous(), d_t)
    assert got[0].shape == (6, 9, 11) and got[3].shape == (6, 9, 11, 3) and got[4].shape == (6, 9, 11, 2)
    assert_closest_equal(got, R.closest_raw(o_n, d_n))
    # (b) permuted leading dims (non-contiguous), broadcast origin
    d_t2 = bt.permute(2, 0, 1, 3)[..., :3]
    o_t2 = torch.tensor([0.1, 0.2, 2.5], device=device).expand(11, 6, 9, 3)
    d_n2 = base.transpose(2, 0, 1, 3)[..., :3]
    o_n2 = np.broadcast_to(np.array([0.1, 0.2, 2.5], np.float32), d_n2.shape)
    assert_closest_equal(r.intersects_closest(o_t2, d_t2), R.closest_raw(o_n2, d_n2))
    assert np.array_equal(r.intersects_count(o_t2, d_t2).cpu().numpy(), R.intersects_count(o_n2, d_n2))
    # (c) single ray, 1-D tensors [3]
    o1 = torch.tensor([0.0, 0.0, 3.0], device=device)
    d1 = torch.tensor([0.0, 0.0, -1.0], device=device)
    hit, front, tri, loc, uv = r.intersects_closest(o1, d1)
    assert hit.shape == () and bool(hit) and loc.shape == (3,)
    # (d) flat [N,3] and [H,W,3] give the same answers
    o, d = W.pinhole_grid(64, 48)
    a = r.intersects_first(T(o, device), T(d, device))
    b = r.intersects_first(T(o, device).reshape(-1, 3), T(d, device).reshape(-1, 3))
    assert a.shape == (48, 64) and torch.equal(a.reshape(-1), b)


def test_edge_cases(device):
    from triro.ray.ray_optix import RayMeshIntersector
    v, f = W.icosphere(2)
    r = make(v, f, device)
    # empty ray batch
    e = torch.zeros((0, 3), device=device)
    hit, front, tri, loc, uv = r.intersects_closest(e, e)
    assert hit.shape == (0,) and loc.shape == (0, 3)
    assert r.intersects_count(e, e).shape == (0,)
    loc, ray, tri = r.intersects_location(e, e)
    assert loc.shape == (0, 3) and ray.shape == (0,)
    hit, front, ray_idx, tri, loc, uv = r.intersects_closest(e, e, stream_compaction=True)
    assert ray_idx.shape == (0,)
    # NaN / Inf / zero-direction rays miss; the others are unaffected
    o = torch.tensor([[0, 0, 3], [float("nan"), 0, 3], [0, 0, 3], [0, 0, 3], [0, 0, 3]], device=device)
    d = torch.tensor([[0, 0, -1], [0, 0, -1], [0, 0, float("inf")], [0, 0, 0], [0, 0, 1]], device=device)
    assert r.intersects_any(o, d).tolist() == [True, False, False, False, False]
    assert r.intersects_first(o, d).tolist()[1:] == [-1, -1, -1, -1]
    hit, front, tri, loc, uv = r.intersects_closest(o, d)
    assert torch.all(loc[1:] == 0) and torch.all(uv[1:] == 0) and not front[1:].any()   # shaders.cu:128-135
    # empty mesh
    r0 = RayMeshIntersector(vertices=torch.zeros((0, 3), device=device), faces=torch.zeros((0, 3), dtype=torch.int32, device=device))
    assert not r0.intersects_any(o, d).any() and r0.intersects_count(o, d).sum() == 0
    # tmax = 1e7
    o = torch.tensor([[0, 0, 2e7], [0, 0, 9e6]], device=device)
    d = torch.tensor([[0, 0, -1.0], [0, 0, -1.0]], device=device)
    assert r.intersects_any(o, d).tolist() == [False, True]
    # validation
    with pytest.raises(ValueError):
        r.intersects_any(o.double(), d.double())
    with pytest.raises(ValueError):
        r.intersects_any(o.cpu(), d.cpu())
    with pytest.raises(ValueError):
        r.intersects_any(o, d[:1])
    with pytest.raises(ValueError):
        r.intersects_any(torch.zeros(2, 2, 2, 2, 3, device=device), torch.zeros(2, 2, 2, 2, 3, device=device))


def test_builder_invariants_and_host_traversal_of_gpu_tree(device):
    """the GPU-built arrays satisfy the LBVH invariants and, traversed by the host simulation
    of the same per-lane code, give the GPU's own answers"""
    from sim import SimBVH
    from test_host_sim import check_tree
    v, f = W.random_soup(30000, seed=4)
    r = make(v, f, device)
    info = r.bvh_info()
    assert info["num_tris"] == len(f) and info["num_nodes"] == len(f) - 1 and 0 < info["depth"] <= 64
    np.testing.assert_array_equal(np.float32(info["aabb_min"]) <= v.min(0), True)
    nodes, links, tris = r.as_wrapper.download()
    qnodes, frame = r.as_wrapper.download_qnodes()
    B = SimBVH(arrays=(nodes, links, tris), qarrays=(qnodes, frame))
    check_tree(B)
    # same tree as the host construction (deterministic builder: replicas on other GPUs agree)
    H = SimBVH(v, f)
    assert np.array_equal(H.tris, tris) and np.array_equal(H.nodes, nodes) and H.depth == info["depth"]
    assert np.array_equal(H.qnodes, qnodes) and np.array_equal(H.frame, frame)      # the 32-byte grid copy too
    o, d = W.hash_rays(20000, 5, v.min(0) * 1.5, v.max(0) * 1.5)
    g = r.intersects_first(T(o, device), T(d, device)).cpu().numpy()
    assert np.array_equal(B.query(1, o, d)["tri"], g)
    import sim
    sim.use_unordered(True)                      # host traversal of the GPU's grid nodes == GPU count
    try:
        assert np.array_equal(B.query(3, o, d)["count"], r.intersects_count(T(o, device), T(d, device)).cpu().numpy())
    finally:
        sim.use_unordered(False)


def test_deep_tree_falls_back_to_bounded_keys(device):
    v, f = W.deep_tree_mesh(4000)
    r = make(v, f, device)
    info = r.bvh_info()
    assert info["key_mode"] == 1 and info["depth"] <= 64
    o, d = W.hash_rays(2000, 3, [-0.1] * 3, [1.1] * 3)
    o[:500] = [1e-10, 1e-10, 1.0]
    d[:500] = [0, 0, -1]
    R = OracleIntersector(v, f, 1)
    assert np.array_equal(r.intersects_count(T(o, device), T(d, device)).cpu().numpy(), R.intersects_count(o, d))
    assert np.array_equal(r.intersects_first(T(o, device), T(d, device)).cpu().numpy(), R.intersects_first(o, d))


def test_update_raw_rebuilds(device):
    v, f = W.icosphere(3)
    r = make(v, f, device)
    o, d = W.readme_perspective(64)
    a = r.intersects_first(T(o, device), T(d, device))
    v2, f2 = W.icosphere(4)
    r.update_raw(T(v2 * 0.5, device), T(f2, device))          # larger mesh: arena regrows
    R = OracleIntersector(v2 * np.float32(0.5), f2, 1)
    assert np.array_equal(r.intersects_first(T(o, device), T(d, device)).cpu().numpy(), R.intersects_first(o, d))
    assert torch.allclose(r.mesh_aabb[1], torch.full((3,), 0.5, device=device), atol=1e-6)
    r.update_raw(T(v, device), T(f, device))                  # smaller again: arena reused
    assert torch.equal(r.intersects_first(T(o, device), T(d, device)), a)


def test_rebuild_shares_cached_temporaries(device):
    """The builder's temporaries are cached per device and the refit-round count of a rebuild is
    guessed from the previous height: interleave handles of different sizes, rebuild a shallow
    tree into a deep one (guess too small -> extra rounds), and switch the cache off."""
    from triro.backend import ops as hops
    o, d = W.hash_rays(20000, 5, [-1.5] * 3, [1.5] * 3)
    ot, dt = T(o, device), T(d, device)
    meshes = [W.icosphere(2), W.bunny_standin(), W.random_soup(5000, seed=2), W.deep_tree_mesh(3000)]
    exp = [OracleIntersector(v, f, 1).closest_raw(o, d) for v, f in meshes]
    rs = [make(v, f, device) for v, f in meshes]
    for cache in (1, 0, 1):
        hops.set_option("build_cache", cache)
        try:
            for k in range(len(meshes)):
                # every handle takes over the NEXT mesh: sizes and heights change both ways
                v, f = meshes[(k + 1) % len(meshes)]
                rs[k].update_raw(T(v, device), T(f, device))
                assert_closest_equal(rs[k].intersects_closest(ot, dt), exp[(k + 1) % len(meshes)])
            for k in range(len(meshes)):
                v, f = meshes[k]
                rs[k].update_raw(T(v, device), T(f, device))
                assert_closest_equal(rs[k].intersects_closest(ot, dt), exp[k])
        finally:
            hops.set_option("build_cache", 1)


def test_contains_points(device):
    v, f = W.icosphere(3)
    r = make(v, f, device)
    assert r.contains_points(torch.tensor([[0, 0, 0.999]], device=device)).tolist() == [True]   # test/test.py:64
    rng = np.random.default_rng(0)
    pts = (rng.random((5000, 3)).astype(np.float32) * 2.4 - 1.2)
    R = OracleIntersector(v, f, 1)
    got = r.contains_points(T(pts, device)).cpu().numpy()
    exp = R.contains_points(pts)
    assert np.array_equal(got, exp)
    rad = np.linalg.norm(pts, axis=1)
    assert got[rad < 0.97].all() and not got[rad > 1.0].any()
    assert not r.contains_points(torch.full((3, 3), 5.0, device=device)).any()


def test_stream_semantics(device):
    """work is enqueued on torch's current stream (the reference uses a private stream)"""
    v, f = W.icosphere(4)
    r = make(v, f, device)
    o, d = W.pinhole_grid(256, 256)
    ot, dt = T(o, device), T(d, device)
    ref = r.intersects_first(ot, dt)
    s = torch.cuda.Stream(device=device)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        o2 = ot * 1.0                       # produced on s; the query must see it
        out = r.intersects_first(o2, dt)
    s.synchronize()
    assert torch.equal(out, ref)


def test_launch_shapes_agree(device):
    """persistent (work-counter) and direct launches return identical results"""
    import triro.backend.ops as hops
    v, f = W.bunny_standin()
    r = make(v, f, device)
    o, d = W.hash_rays(3_000_000, 77, v.min(0) * 1.5, v.max(0) * 1.5)
    ot, dt = T(o, device), T(d, device)
    try:
        hops.set_option("persistent", 1)
        a = r.intersects_closest(ot, dt)
        ca = r.intersects_count(ot, dt)
        hops.set_option("persistent", 0)
        b = r.intersects_closest(ot, dt)
        cb = r.intersects_count(ot, dt)
    finally:
        hops.set_option("persistent", 0)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.equal(ca, cb)
    # every launch-shape knob of the direct kernel is a pure scheduling choice
    defaults = {"adaptive": 1, "scramble": 1, "block_size": 128, "xcd_chunk": 128, "compact": 1, "steal": 1}
    try:
        for name, values in (("adaptive", (0,)), ("scramble", (0,)), ("block_size", (64, 256)),
                             ("xcd_chunk", (0, 48, 1024)), ("compact", (0,)), ("steal", (0, 2, 8, 64))):
            for val in values:
                hops.set_option(name, val)
                if name == "scramble":
                    hops.set_option("adaptive", 0)        # the scrambled order is the one without hints
                for _ in range(2):                        # second call runs on the learned order
                    c = r.intersects_closest(ot, dt)
                    for x, y in zip(c, b):
                        assert torch.equal(x, y), (name, val)
                if name == "steal":                       # the other queries that can steal
                    assert torch.equal(r.intersects_count(ot, dt), cb), (name, val)
                    assert torch.equal(r.intersects_first(ot, dt), b[2]), (name, val)
                    assert torch.equal(r.intersects_any(ot, dt), b[0]), (name, val)
                hops.set_option(name, defaults[name])
                hops.set_option("adaptive", 1)
    finally:
        for name, val in defaults.items():
            hops.set_option(name, val)
    # size-independent properties at full size (10M-ray class inputs are covered in bench.py)
    hit, front, tri, loc, uv = a
    assert torch.equal(hit, tri >= 0) and torch.equal(hit, ca > 0)
    assert torch.equal(r.intersects_first(ot, dt), tri)
    assert torch.equal(r.intersects_any(ot, dt), hit)
    # loc == sum_i w_i V_i with uv = (w0, w1)
    vt, ft = T(v, device), T(f, device).long()
    tv = vt[ft[tri[hit].long()]]
    w = uv[hit]
    rec = w[:, :1] * tv[:, 0] + w[:, 1:] * tv[:, 1] + (1 - w[:, :1] - w[:, 1:]) * tv[:, 2]
    assert torch.allclose(rec, loc[hit], rtol=1e-5, atol=2e-6)
    stats = hops.trace_stats_closest(r.as_wrapper, ot[:100000], dt[:100000])
    assert stats["rays"] == 100000 and stats["node_visits"] > 0


@pytest.mark.parametrize("threshold", [2, 8])
def test_work_stealing_is_exact(device, threshold):
    """Intra-wave work stealing forced on with a tiny trip threshold (idle lanes take subtrees of
    busy lanes from trip 2 / 8 on): closest, first and count against the oracle on coherent and
    incoherent rays, a multi-layer scene, a soup, and the deep tree that needs the 64-bit trail."""
    import triro.backend.ops as hops
    cases = [
        (W.nested_shells(4), tuple(x.reshape(-1, 3) for x in W.pinhole_grid(160, 120))),
        (W.random_soup(4000, seed=9), W.hash_rays(30000, 21, [-1.3] * 3, [1.3] * 3)),
        (W.deep_tree_mesh(3000), W.hash_rays(20000, 22, [-0.2] * 3, [1.2] * 3)),
        (W.bunny_standin(), tuple(x.reshape(-1, 3) for x in W.pinhole_grid(256, 192))),
    ]
    try:
        hops.set_option("steal", threshold)
        for (v, f), (o, d) in cases:
            r = make(v, f, device)
            R = OracleIntersector(v, f, 1)
            ot, dt = T(o, device), T(d, device)
            for _ in range(2):
                assert_closest_equal(r.intersects_closest(ot, dt), R.closest_raw(o, d))
            assert np.array_equal(r.intersects_count(ot, dt).cpu().numpy(), R.intersects_count(o, d))
            assert np.array_equal(r.intersects_first(ot, dt).cpu().numpy(), R.intersects_first(o, d))
            assert np.array_equal(r.intersects_any(ot, dt).cpu().numpy(), R.intersects_count(o, d) > 0)
    finally:
        hops.set_option("steal", 1)


def test_tile_mapping_is_a_pure_permutation(device):
    """Image-shaped batches may be traced in 8x8 pixel tiles per wave (option tile; automatic from
    4 M rays on).  Outputs are written at the true flat index: forced on (2) == off (0) for
    [H,W,3], [B,H,W,3], a broadcast origin, strided directions, and sizes where it must not engage."""
    import triro.backend.ops as hops
    v, f = W.bunny_standin()
    r = make(v, f, device)
    o, d = W.pinhole_grid(200, 136)                      # [H=136, W=200, 3]: W % 8 == 0, H % 8 == 0
    ot, dt = T(o, device), T(d, device)
    H, Wd = ot.shape[:2]
    assert Wd % 8 == 0 and H % 8 == 0
    cases = {
        "HW": (ot, dt),
        "BHW": (torch.stack([ot, ot * 1.01]), torch.stack([dt, dt])),
        "broadcast origin": (ot[:1, :1].expand(H, Wd, 3), dt),
        "strided dirs": (ot, torch.stack([dt, dt], dim=2)[:, :, 1]),
        "W not multiple of 8": (ot[:, :Wd - 3], dt[:, :Wd - 3]),
        "odd row count": (ot[:H - 3], dt[:H - 3]),
    }
    try:
        for name, (oo, dd) in cases.items():
            hops.set_option("tile", 0)
            ref = r.intersects_closest(oo, dd)
            cref = r.intersects_count(oo, dd)
            hops.set_option("tile", 2)
            for _ in range(2):
                got = r.intersects_closest(oo, dd)
                for x, y in zip(got, ref):
                    assert torch.equal(x, y), name
            assert torch.equal(r.intersects_count(oo, dd), cref), name
            hops.set_option("tile", 1)
            for shape in (1, 2, 3, 4, 0):          # 2x32, 4x16, 8x8 tiles, by density, rows
                hops.set_option("tile_small", shape)
                for _ in range(2):
                    got = r.intersects_closest(oo, dd)
                    for x, y in zip(got, ref):
                        assert torch.equal(x, y), (name, shape)
                assert torch.equal(r.intersects_first(oo, dd), ref[2]), (name, shape)
                assert torch.equal(r.intersects_any(oo, dd), ref[0]), (name, shape)
    finally:
        hops.set_option("tile", 1)
        hops.set_option("tile_small", 4)


def test_adaptive_launch_order_never_changes_results(device):
    """the launch order learned from the previous launch (per handle and stream) is a pure
    scheduling hint: repeated calls, other batch sizes, other streams and a rebuild in between
    return what the non-adaptive launch returns"""
    import triro.backend.ops as hops
    v, f = W.bunny_standin()
    r = make(v, f, device)
    o, d = W.pinhole_grid(512, 512, distance=2.5 * 1.12)
    ot, dt = T(o, device), T(d, device)
    try:
        hops.set_option("adaptive", 0)
        ref = r.intersects_closest(ot, dt)
        ref_cnt = r.intersects_count(ot, dt)
        hops.set_option("adaptive", 1)
        for _ in range(4):                                   # call 1 measures, calls 2.. use the order
            got = r.intersects_closest(ot, dt)
            for a, b in zip(got, ref):
                assert torch.equal(a, b)
        assert torch.equal(r.intersects_count(ot, dt), ref_cnt)
        sub = r.intersects_first(ot[:200], dt[:200])         # different block count in between
        assert torch.equal(sub, ref[2][:200])
        s1, s2 = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
        torch.cuda.synchronize()
        outs = []
        for _ in range(3):                                   # two streams interleaved on one handle
            for st in (s1, s2):
                with torch.cuda.stream(st):
                    outs.append(r.intersects_first(ot, dt))
        torch.cuda.synchronize()
        for x in outs:
            assert torch.equal(x, ref[2])
        v2, f2 = W.icosphere(5)
        r.update_raw(T(v2, device), T(f2, device))           # rebuild invalidates the hints
        R = OracleIntersector(v2, f2, 1)
        for _ in range(2):
            assert np.array_equal(r.intersects_first(ot, dt).cpu().numpy(), R.intersects_first(o, d))
    finally:
        hops.set_option("adaptive", 1)


def test_refit_and_serialization(device, tmp_path):
    """refit for unchanged topology, BVH (de)serialisation"""
    from triro.ray.ray_optix import RayMeshIntersector
    v, f = W.icosphere(5)
    r = make(v, f, device)
    o, d = W.pinhole_grid(200, 200)
    ot, dt = T(o, device), T(d, device)
    # deform: anisotropic scale + smooth displacement, same faces
    v2 = (W.displaced(v, seed=3, amplitude=0.2) * np.array([1.0, 0.7, 1.3], np.float32)).astype(np.float32)
    r.refit(T(v2, device))
    R2 = OracleIntersector(v2, f, 1)
    assert_closest_equal(r.intersects_closest(ot, dt), R2.closest_raw(o, d))
    assert np.array_equal(r.intersects_count(ot, dt).cpu().numpy(), R2.intersects_count(o, d))
    info = r.bvh_info()
    assert np.all(np.float32(info["aabb_min"]) <= v2.min(0)) and np.all(np.float32(info["aabb_max"]) >= v2.max(0))
    rb = make(v2, f, device)                      # a fresh build of the deformed mesh agrees
    for a, b in zip(r.intersects_closest(ot, dt), rb.intersects_closest(ot, dt)):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        r.refit(T(v2[:-1], device))
    # save / load round trip: identical arena, identical answers, no rebuild
    path = str(tmp_path / "mesh_bvh.npz")
    r.save(path)
    r3 = RayMeshIntersector.load(path, device=device)
    n1, l1, t1 = r.as_wrapper.download()
    n3, l3, t3 = r3.as_wrapper.download()
    assert np.array_equal(n1, n3) and np.array_equal(l1, l3) and np.array_equal(t1, t3)
    (q1, f1), (q3, f3), (qb, fb) = r.as_wrapper.download_qnodes(), r3.as_wrapper.download_qnodes(), rb.as_wrapper.download_qnodes()
    assert np.array_equal(q1, q3) and np.array_equal(f1, f3)          # loaded handle: same grid nodes and frame
    assert np.array_equal(f1, fb)                                     # refit re-derives the grid from the new bounds
    assert r3.bvh_info()["depth"] == info["depth"]
    for a, b in zip(r3.intersects_closest(ot, dt), r.intersects_closest(ot, dt)):
        assert torch.equal(a, b)
    lo, ra, tr_ = r3.intersects_location(ot, dt)
    lo2, ra2, tr2 = r.intersects_location(ot, dt)
    assert torch.equal(ra, ra2) and torch.equal(tr_, tr2)
    blob = r.as_wrapper.serialize().copy()
    blob[0] ^= 0xFF
    from triro.ray.ray_optix import OptixAccelStructureWrapper
    with pytest.raises(ValueError):
        OptixAccelStructureWrapper().deserialize(blob, device)


def test_location_fused_equals_two_pass(device):
    import triro.backend.ops as hops
    v, f = W.nested_shells(4, radii=(1.0, 0.8, 0.6, 0.4, 0.3))
    r = make(v, f, device)
    o, d = W.pinhole_grid(300, 200)
    ot, dt = T(o, device), T(d, device)
    a = hops.intersects_location(r.as_wrapper, ot, dt, fused=True)
    b = hops.intersects_location(r.as_wrapper, ot, dt, fused=False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    a = hops.intersects_location(r.as_wrapper, ot, dt, ray_base=1000, fused=True)
    assert int(a[1].min()) >= 1000
