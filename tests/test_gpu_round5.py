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
