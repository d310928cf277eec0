"""8-wide compressed nodes of the streaming launch (csrc/tr_wide.h, option wide): structure of the collapsed
hierarchy and bit-exact results against the oracle and against the binary streaming launch."""
import numpy as np
import pytest
import torch

import workloads as W
from oracle.oracle import OracleIntersector

pytestmark = pytest.mark.gpu


@pytest.fixture
def opts():
    import triro.backend.ops as hops
    saved = {}

    def set_(**kw):
        for k, v in kw.items():
            saved.setdefault(k, None)
            hops.set_option(k, v)
    yield set_
    defaults = {"stream": 1, "wide": 2, "wide_stack": 12, "stream_rays": 256, "stream_refill": 0, "wide_direct": 1, "adaptive": 1}
    for k in saved:
        hops.set_option(k, defaults[k])


def _mk(v, f, device):
    from triro.ray.ray_optix import RayMeshIntersector
    return RayMeshIntersector(vertices=torch.from_numpy(v).to(device), faces=torch.from_numpy(f).to(device))


def _check_all(r, R, o, d, tag):
    on, dn = o.cpu().numpy(), d.cpu().numpy()
    hit, front, tri, loc, uv = r.intersects_closest(o, d)
    eh, ef, et, el, eu = R.intersects_closest(on, dn)
    assert np.array_equal(hit.cpu().numpy(), eh), tag
    assert np.array_equal(tri.cpu().numpy(), et), tag
    assert np.array_equal(front.cpu().numpy(), ef), tag
    assert np.array_equal(loc.cpu().numpy(), el) and np.array_equal(uv.cpu().numpy(), eu), tag
    assert np.array_equal(r.intersects_first(o, d).cpu().numpy(), et), tag
    cnt = R.intersects_count(on, dn)
    assert np.array_equal(r.intersects_count(o, d).cpu().numpy(), cnt), tag
    assert np.array_equal(r.intersects_any(o, d).cpu().numpy(), cnt > 0), tag
    return int(eh.sum()), int(cnt.max())


@pytest.mark.parametrize("wide_stack", [12, 2])
def test_wide_streaming_matches_the_oracle(device, opts, wide_stack):
    """every query family through k_query_wide (stream forced) on a displaced sphere, nested shells (8 hits per
    ray) and an unstructured soup (dozens of hits per ray: deep stacks), with the stack in LDS and with all but two
    entries of it in the global spill rows"""
    opts(stream=2, wide=1, wide_stack=wide_stack)
    for name, (v, f), nrays in (("sphere", W.headline_mesh(5), 60_001), ("shells", W.nested_shells(4), 40_000),
                                ("soup", W.random_soup(20_000, seed=3), 30_000)):
        r = _mk(v, f, device)
        R = OracleIntersector(v, f, mode=1)
        lo, hi = v.min(0) * 1.5, v.max(0) * 1.5
        o, d = W.hash_rays_torch(nrays, 77, lo, hi, device=device)
        nh, mx = _check_all(r, R, o, d, (name, wide_stack))
        assert nh > nrays // 50, name
        if name == "soup":
            assert mx >= 10
        # rays from inside the scene, axis-aligned rays, rays that start on the surface
        c = torch.from_numpy(((v.min(0) + v.max(0)) / 2).astype(np.float32)).to(device)
        o2 = c.expand(4096, 3).contiguous()
        d2 = torch.nn.functional.normalize(torch.randn(4096, 3, device=device, generator=torch.Generator(device=device).manual_seed(5)), dim=1)
        d2[:64] = torch.tensor([1.0, 0.0, 0.0], device=device)
        d2[64:128] = torch.tensor([0.0, 0.0, -1.0], device=device)
        _check_all(r, R, o2, d2, (name, "inside"))


def test_wide_equals_binary_streaming_and_survives_rebuilds(device, opts):
    """wide = 1 against wide = 0 on a 1 M-ray batch (torch.equal), then refit, update_raw and a save / load round trip:
    the wide nodes are derived data and must follow"""
    from triro.ray.ray_optix import RayMeshIntersector
    v, f = W.headline_mesh(6)
    r = _mk(v, f, device)
    lo, hi = v.min(0) * 1.5, v.max(0) * 1.5
    o, d = W.hash_rays_torch(1 << 20, 99, lo, hi, device=device)
    opts(stream=2, wide=0)
    base = r.intersects_closest(o, d)
    base_cnt = r.intersects_count(o, d)
    opts(wide=1)
    for a, e in zip(r.intersects_closest(o, d), base):
        assert torch.equal(a, e)
    assert torch.equal(r.intersects_count(o, d), base_cnt)
    s2 = torch.cuda.Stream(device)                       # a second stream meets the wide nodes built on the first
    s2.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(s2):
        got2 = r.intersects_closest(o, d)
    torch.cuda.current_stream(device).wait_stream(s2)
    for a, e in zip(got2, base):
        assert torch.equal(a, e)
    v2 = W.displaced(v, seed=1, amplitude=0.05)
    r.refit(torch.from_numpy(v2).to(device))
    R2 = OracleIntersector(v2, f, mode=1)
    _check_all(r, R2, o[:50_000], d[:50_000], "refit")
    v3, f3 = W.nested_shells(4)
    r.update_raw(torch.from_numpy(v3).to(device), torch.from_numpy(f3).to(device))
    R3 = OracleIntersector(v3, f3, mode=1)
    o3, d3 = W.hash_rays_torch(50_000, 5, v3.min(0) * 1.5, v3.max(0) * 1.5, device=device)
    _check_all(r, R3, o3, d3, "update_raw")
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        r.save(os.path.join(td, "m"))
        r4 = RayMeshIntersector.load(os.path.join(td, "m"), device=device)
    _check_all(r4, R3, o3, d3, "load")


def test_wide_on_tiny_and_degenerate_meshes(device, opts):
    """1, 2, 3 triangles; duplicated triangles (equal Morton keys: a chain-like hierarchy); a zero-area triangle"""
    opts(stream=2, wide=1)
    base_v = np.array([[0.5, -0.5, 0], [0, 0.5, 0], [-0.5, -0.5, 0], [0.5, -0.5, -1], [0, 0.5, -1], [-0.5, -0.5, -1],
                       [0.5, -0.5, -2], [0, 0.5, -2], [-0.5, -0.5, -2]], np.float32)
    g = torch.Generator(device=device).manual_seed(1)
    o = torch.rand(5000, 3, device=device, generator=g) * 2 - 1
    o[:, 2] = 3
    d = torch.tensor([0.0, 0.0, -1.0], device=device).expand(5000, 3).contiguous() + 0.05 * (torch.rand(5000, 3, device=device, generator=g) - 0.5)
    for nt in (1, 2, 3):
        f = np.arange(3 * nt, dtype=np.int32).reshape(nt, 3)
        r = _mk(base_v, f, device)
        _check_all(r, OracleIntersector(base_v, f, mode=0), o, d, nt)
    f = np.array([[0, 1, 2]] * 40 + [[3, 4, 5]] * 3 + [[6, 6, 6]], np.int32)       # 40 copies, 3 copies, a degenerate one
    r = _mk(base_v, f, device)
    nh, mx = _check_all(r, OracleIntersector(base_v, f, mode=0), o, d, "duplicates")
    assert mx == 43


@pytest.mark.parametrize("wide_stack", [12, 3])
def test_direct_launch_on_wide_nodes_matches_the_oracle(device, opts, wide_stack):
    """option wide_direct = 3: every query family of the DIRECT launch (image tiles, learned launch order over several
    launches) on the 8-wide nodes, incl. the multi-hit list (location) with more hits than its cap"""
    opts(wide_direct=3, wide_stack=wide_stack)
    for name, (v, f), cam in (("shells", W.nested_shells(5), 2.5), ("sphere", W.headline_mesh(6), None), ("soup", W.random_soup(30_000, seed=5), 4.0)):
        r = _mk(v, f, device)
        R = OracleIntersector(v, f, mode=1)
        dist = cam if cam is not None else 2.5 * float(np.linalg.norm(v, axis=1).max())
        o_np, d_np = W.pinhole_grid(256, 192, distance=dist)
        o = torch.from_numpy(np.ascontiguousarray(o_np)).to(device)
        d = torch.from_numpy(d_np).to(device)
        for launch in range(5):            # cold order, measured order, re-measured order
            nh, mx = _check_all(r, R, o, d, (name, wide_stack, launch))
        assert r.as_wrapper.last_launch()["shape"] == 4
        loc, ray, tri = r.intersects_location(o, d)
        el, er, et = R.intersects_location(o_np, d_np)
        assert np.array_equal(ray.cpu().numpy(), er) and np.array_equal(tri.cpu().numpy(), et), name
        assert np.array_equal(loc.cpu().numpy(), el), name
        if name == "soup":
            assert mx > 8                   # the list replaces entries beyond its cap
        # flat, ragged batch (no tiles) and a tiny one
        for m in (1, 63, 5001):
            _check_all(r, R, o.reshape(-1, 3)[:m].contiguous(), d.reshape(-1, 3)[:m].contiguous(), (name, "flat", m))


def test_graph_replay_of_a_wide_launch_follows_a_refit(device, opts):
    """a HIP graph that captured a launch on the 8-wide nodes must see the geometry of a later refit: the wide nodes of a
    handle that has them are rebuilt by refit / update_raw themselves (same buffer), not lazily by the next query"""
    opts(stream=2, wide=1)
    v, f = W.headline_mesh(5)
    r = _mk(v, f, device)
    lo, hi = v.min(0) * 1.5, v.max(0) * 1.5
    o, d = W.hash_rays_torch(200_000, 7, lo, hi, device=device)
    gs = torch.cuda.Stream(device)
    gs.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(gs):
        for _ in range(3):
            r.intersects_closest(o, d)          # builds the wide nodes, sizes every buffer
    torch.cuda.current_stream(device).wait_stream(gs)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=gs):
        out = r.intersects_closest(o, d)
    graph.replay()
    torch.cuda.synchronize()
    R = OracleIntersector(v, f, mode=1)
    e = R.intersects_closest(o.cpu().numpy(), d.cpu().numpy())
    assert np.array_equal(out[2].cpu().numpy(), e[2]) and np.array_equal(out[3].cpu().numpy(), e[3])
    v2 = W.displaced(v, seed=3, amplitude=0.1)
    r.refit(torch.from_numpy(v2).to(device))
    graph.replay()
    torch.cuda.synchronize()
    e2 = OracleIntersector(v2, f, mode=1).intersects_closest(o.cpu().numpy(), d.cpu().numpy())
    assert not np.array_equal(e2[2], e[2])
    assert np.array_equal(out[0].cpu().numpy(), e2[0]) and np.array_equal(out[2].cpu().numpy(), e2[2])
    assert np.array_equal(out[3].cpu().numpy(), e2[3]) and np.array_equal(out[4].cpu().numpy(), e2[4])
