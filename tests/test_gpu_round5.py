"""Round 5 GPU tests: the exact replica hash, the fused box test on rays that stress its margins."""
import numpy as np
import pytest
import torch

import workloads as W
from oracle.oracle import OracleIntersector
from test_gpu_round2 import T, make, assert_closest_bitexact, on_surface_rays

pytestmark = pytest.mark.gpu


def test_replica_hash_is_exact_and_stable(device):
    """tr_bvh_replica_hash (ABI 9): equal for two builds of the same mesh (the builder is deterministic), for a saved
    and re-loaded hierarchy; different as soon as ONE vertex coordinate or one face differs -- also where no probe ray
    of round 4's fingerprint would have landed"""
    from triro.ray.ray_optix import RayMeshIntersector
    v, f = W.icosphere(5)
    a, b = make(v, f, device), make(v, f, device)
    ha = a.replica_hash()
    assert ha == b.replica_hash() and ha != 0 and a.replica_fingerprint() == ha
    v2 = v.copy()
    v2[len(v) // 2, 1] = np.nextafter(v2[len(v) // 2, 1], np.float32(2))          # one ulp in one coordinate
    assert make(v2, f, device).replica_hash() != ha
    f2 = f.copy()
    f2[[10, 11]] = f2[[11, 10]]                                                    # two faces swap their ids
    assert make(v, f2, device).replica_hash() != ha
    assert make(v[:, [1, 0, 2]], f, device).replica_hash() != ha
    g0 = a.generation
    a.refit(torch.from_numpy(v2).to(device))
    assert a.generation == g0 + 1 and a.replica_hash() != ha
    a.update_raw(torch.from_numpy(v).to(device), torch.from_numpy(f).to(device))
    assert a.generation == g0 + 2 and a.replica_hash() == ha
    one = RayMeshIntersector(vertices=torch.from_numpy(v[:3]).to(device), faces=torch.tensor([[0, 1, 2]], dtype=torch.int32, device=device))
    assert one.replica_hash() not in (0, ha)


@pytest.mark.parametrize("case", ["far_from_origin", "far_camera", "axis_aligned", "flat_mesh"])
def test_fused_box_test_on_rays_that_stress_its_margins(device, case):
    """the fused conservative box test (tr_ray_fuse) through every launch family that walks grid / 8-wide nodes --
    stealing closest on forced grid nodes, unordered count, the list query, the streaming launch, the wide launches --
    on meshes far from the origin, cameras far from the mesh, rays parallel to the axes (clamped reciprocals) and a
    mesh that is flat in one axis: bit for bit against the oracle"""
    import triro.backend.ops as hops
    rng = np.random.default_rng(11)
    if case == "far_from_origin":
        v, f = W.icosphere(5)
        v = (W.displaced(v, seed=2, amplitude=0.05) * np.float32(0.05) + np.float32([900.0, -1700.0, 333.25])).astype(np.float32)
        o, d = W.pinhole_grid(256, 192, distance=0.2)
        o = (o + v.mean(0)).astype(np.float32)
    elif case == "far_camera":
        v, f = W.icosphere(5)
        o, d = W.pinhole_grid(256, 192, distance=5.0e4, vfov_deg=0.003)
    elif case == "axis_aligned":
        v, f = W._box((0.0, 0.0, 0.0), (1.0, 1.0, 0.5), 24)
        v, f = v.astype(np.float32), f.astype(np.int32)
        o, d = W.ortho_grid(193)                                   # linspace(-1, 1, 193): origins in the side faces' planes
        o, d = np.ascontiguousarray(o), np.ascontiguousarray(d)
    else:
        v, f = W.icosphere(5)
        v = v.copy()
        v[:, 2] = np.float32(0.25)
        o, d = W.pinhole_grid(256, 192, distance=2.5)
    R = OracleIntersector(v, f, 1)
    r = make(v, f, device)
    fo, fd = o.reshape(-1, 3), d.reshape(-1, 3)
    exp = R.closest_raw(fo, fd)
    cnt = R.intersects_count(fo, fd)
    eloc = R.intersects_location(fo, fd)
    ot, dt = T(o, device), T(d, device)
    try:
        for opts in ({"grid_nodes": 2}, {"grid_nodes": 2, "stream": 2}, {"stream": 2, "wide": 1}, {"wide_direct": 3}, {"grid_nodes": 2, "adaptive": 0}):
            for k, val in opts.items():
                hops.set_option(k, val)
            for rep in range(3):
                got = [g.reshape((-1,) + tuple(g.shape[o.ndim - 1:])) for g in r.intersects_closest(ot, dt)]
                assert_closest_bitexact(got, exp, f"{case} {opts} launch {rep}")
            assert np.array_equal(r.intersects_any(ot, dt).cpu().numpy().reshape(-1), cnt > 0)
            assert np.array_equal(r.intersects_count(ot, dt).cpu().numpy().reshape(-1), cnt)
            loc, ridx, tri = r.intersects_location(ot, dt)
            assert np.array_equal(ridx.cpu().numpy(), eloc[1]) and np.array_equal(tri.cpu().numpy(), eloc[2])
            assert np.array_equal(loc.cpu().numpy(), eloc[0])
            for k in opts:
                hops.set_option(k, {"grid_nodes": 1, "stream": 1, "wide": 2, "wide_direct": 1, "adaptive": 1}[k])
    finally:
        for k, val in {"grid_nodes": 1, "stream": 1, "wide": 2, "wide_direct": 1, "adaptive": 1}.items():
            hops.set_option(k, val)


def test_deferred_sort_rides_in_the_next_launch(device):
    """option sort_inline (default): from a batch shape's fourth launch on, the sort of the measured block costs is not a
    kernel behind the measuring launch but a workgroup of the NEXT launch of that shape (query_direct_body, tr_sort_job),
    which still reads the old order.  Every launch equals the oracle; the carrying launches are the ones right after a
    measuring one; a rebuild, a change of shape or of query between the two runs the pending sort as a kernel; the orders
    stay valid (learned order + split blocks) throughout; sort_inline = 0 gives the same results."""
    from triro.backend import ops as hops
    v, f = W.icosphere(6)
    v = W.displaced(v, seed=3)
    res = 256
    o, d = W.pinhole_grid(res, res, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
    exp = OracleIntersector(v, f).intersects_closest(np.ascontiguousarray(o).reshape(-1, 3), d.reshape(-1, 3))
    exp = [e.reshape(res, res, *e.shape[1:]) for e in exp[:5]]
    O, D = T(np.ascontiguousarray(o), device), T(d, device)
    r = make(v, f, device)
    carried = []
    for k in range(14):
        got = r.intersects_closest(O, D)
        assert_closest_bitexact(got, exp, f"launch {k}")
        li = r.as_wrapper.last_launch()
        carried.append(li["sort_carried"])
        if k >= 2:
            assert li["learned_order"] == 1 and li["shape"] == 1 and li["grid_nodes"] == 1
    # launches count from 1 in the slot: the 1st (plain shape) .. then measuring when launches < 3 or launches % 4 == 3
    assert sum(carried) >= 2 and carried[:4] == [0, 0, 0, 0], carried
    first = carried.index(1)
    assert carried[first::4][:2] == [1, 1] and sum(carried[first:first + 4]) == 1, carried
    # a pending sort meets another query, another shape, a rebuild
    for _ in range(8):              # on to the launch right after a carrying one ...
        r.intersects_closest(O, D)
        if r.as_wrapper.last_launch()["sort_carried"]:
            break
    assert r.as_wrapper.last_launch()["sort_carried"] == 1
    for _ in range(2):              # ... two more without measurement ...
        r.intersects_closest(O, D)
    r.intersects_closest(O, D)      # ... and the measuring one: its sort is pending now
    assert np.array_equal(r.intersects_any(O, D).cpu().numpy(), exp[0])
    assert_closest_bitexact(r.intersects_closest(O, D), exp, "after an any-hit launch in between")
    for _ in range(8):
        assert_closest_bitexact(r.intersects_closest(O[: res // 2], D[: res // 2]), [e[: res // 2] for e in exp], "half image")
        assert_closest_bitexact(r.intersects_closest(O, D), exp, "full image again")
    flags = []
    for k in range(12):             # any-hit launches carry it too (their own instantiation of the carrying kernel)
        assert np.array_equal(r.intersects_any(O, D).cpu().numpy(), exp[0]), f"any-hit launch {k}"
        flags.append(r.as_wrapper.last_launch()["sort_carried"])
    assert sum(flags) >= 2, flags
    cnt_exp = OracleIntersector(v, f).intersects_count(np.ascontiguousarray(o).reshape(-1, 3), d.reshape(-1, 3)).reshape(res, res)
    flags = []
    for k in range(14):             # ... and the count launches that steal
        assert np.array_equal(r.intersects_count(O, D).cpu().numpy(), cnt_exp), f"count launch {k}"
        li = r.as_wrapper.last_launch()
        flags.append(li["sort_carried"])
    assert li["shape"] == 3 and sum(flags) >= 2, (flags, li)
    r.update_raw(torch.from_numpy(v).to(device), torch.from_numpy(f).to(device))
    for k in range(10):
        assert_closest_bitexact(r.intersects_closest(O, D), exp, f"after the rebuild, launch {k}")
    assert r.as_wrapper.last_launch()["learned_order"] == 1
    hops.set_option("sort_inline", 0)
    try:
        r2 = make(v, f, device)
        for k in range(10):
            assert_closest_bitexact(r2.intersects_closest(O, D), exp, f"sort_inline=0, launch {k}")
            assert r2.as_wrapper.last_launch()["sort_carried"] == 0
    finally:
        hops.set_option("sort_inline", 1)


def test_a_graph_recorded_while_a_sort_is_pending(device):
    """A launch recorded into a HIP graph right after a measuring launch (its sort deferred, pending on that stream) neither
    carries that sort nor gets it recorded in front of it: the sort waits for the next launch that is really enqueued.
    Replays and eager launches on the stream keep giving the oracle's bits, and the learned order survives."""
    v, f = W.icosphere(6)
    v = W.displaced(v, seed=4)
    res = 256
    o, d = W.pinhole_grid(res, res, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
    exp = [e.reshape(res, res, *e.shape[1:]) for e in
           OracleIntersector(v, f).intersects_closest(np.ascontiguousarray(o).reshape(-1, 3), d.reshape(-1, 3))[:5]]
    O, D = T(np.ascontiguousarray(o), device), T(d, device)
    r = make(v, f, device)
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        for _ in range(12):                      # on to the launch right after a carrying one, then to the next measuring one
            r.intersects_closest(O, D)
            if r.as_wrapper.last_launch()["sort_carried"]:
                break
        assert r.as_wrapper.last_launch()["sort_carried"] == 1
        for _ in range(3):
            r.intersects_closest(O, D)           # the third of these measures: its sort is pending now
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = r.intersects_closest(O, D)
    assert r.as_wrapper.last_launch()["sort_carried"] == 0        # the recorded launch did not take it
    for k in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert_closest_bitexact(out, exp, f"replay {k}")
    with torch.cuda.stream(side):
        got = r.intersects_closest(O, D)         # the next real launch on the stream carries the pending sort
        assert r.as_wrapper.last_launch()["sort_carried"] == 1
        for k in range(6):
            got = r.intersects_closest(O, D)
            assert r.as_wrapper.last_launch()["learned_order"] == 1
    side.synchronize()
    assert_closest_bitexact(got, exp, "eager launches after the recording")
    g.replay()
    torch.cuda.synchronize()
    assert_closest_bitexact(out, exp, "replay after more eager launches")
