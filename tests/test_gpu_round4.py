"""Round 4 GPU tests: the multi-GPU path with more than one rank on ONE device (gloo, host-staged), the
destination rank's emulation, the vectorised record expansion, the public async call without collectives."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import workloads as W
from oracle.oracle import OracleIntersector

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port(base):
    from conftest import free_port
    return str(free_port())


def _bench(extra, timeout=900):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_two_ranks_share_one_gpu_real_tracer():
    """tests/gloo_world2_gpu.py: two processes, both on cuda:0, real tracer, every query family, every
    destination, ragged / weighted / image shards -- all torch.equal to the unsharded call"""
    script = os.path.join(ROOT, "tests", "gloo_world2_gpu.py")
    port = _port(29700)
    procs = [subprocess.Popen([sys.executable, script, str(r), "2", port], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=840) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and so.strip().endswith("OK"), (p.returncode, so[-1000:], se[-4000:])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("extra", [
    [],                                                            # c5i weak: the stack of batches cut at weighted rows
    ["--dst-share", "1"],                                          # ... one 256^2 batch per rank (even shards)
    ["--records", "packed"],                                       # ... 12-byte records (rank 0 without the rays)
    ["--scaling", "strong"],                                       # ONE batch in two row bands
    ["--scaling", "strong", "--dst-share", "0.5"],                 # ... rank 0 takes a smaller band
    ["--workload", "c5ii", "--total-rays", "300001"],              # ragged shards of one flat batch
    ["--workload", "c5ii", "--total-rays", "300001", "--dst-share", "auto", "--chunks", "3"],
], ids=["weak", "weak-even", "weak-packed12", "strong", "strong-weighted", "c5ii-ragged", "c5ii-weighted-chunks"])
def test_bench_two_ranks_on_one_gpu_over_gloo(extra):
    """bench.py --gpus 2 --backend gloo with the REAL tracer (both ranks on cuda:0): the launcher, the pipeline and
    the self-verification (last timed step == cold first call) with two ranks and device tensors"""
    r = _bench(["--gpus", "2", "--backend", "gloo", "--subdiv", "5", "--res", "256", "--steps", "4", "--warmup", "2",
                "--min-warmup-ms", "0", "--no-cpu-baseline", "--no-companions"] + extra)
    assert r["n_gpus"] == 2 and r["verified"] is True
    assert "NOT a measurement" in r["data"]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("extra", [[], ["--workload", "c5ii", "--total-rays", "4000001"]], ids=["weak", "c5ii"])
def test_bench_eight_ranks_on_one_gpu_over_gloo(extra):
    """the driver's `bench.py --gpus 8` with the transport replaced (gloo, host-staged; all ranks on cuda:0): eight
    shards with rank 0's narrower, seven peers' 4-byte records in one gather per step, finished on rank 0 -- and the
    last timed step equal to the cold first call bit for bit"""
    r = _bench(["--gpus", "8", "--backend", "gloo", "--subdiv", "6", "--res", "256", "--steps", "3", "--warmup", "1",
                "--min-warmup-ms", "0", "--no-cpu-baseline", "--no-companions"] + extra)
    assert r["n_gpus"] == 8 and r["verified"] is True and "NOT a measurement" in r["data"]
    sh = r["config"]["shard_rays"]
    assert len(sh) == 8 and len(set(sh[1:])) == 1 and 0 < sh[0] < sh[1] and sum(sh) == r["config"]["rays_total"]
    assert "4 B/ray slot records over gloo" in r["config"]["parallelism"]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("extra", [
    ["--emulate-world", "8"],
    ["--emulate-world", "8", "--dst-share", "1"],
    ["--emulate-world", "8", "--records", "packed"],
    ["--emulate-world", "4", "--scaling", "strong", "--dst-share", "auto"],
    ["--emulate-world", "8", "--workload", "c5ii", "--total-rays", "2000003", "--dst-share", "auto", "--chunks", "2"],
    ["--emulate-world", "3", "--workload", "c5ii", "--total-rays", "500000", "--arrival-priority", "--records", "packed"],
    ["--emulate-world", "8", "--arrival", "none", "--opt", "expand4=3"],
], ids=["weak8", "weak8-even", "weak8-packed12", "strong4-weighted", "c5ii8-weighted", "c5ii3-priority", "weak8-no-arrival-tiles"])
def test_emulated_destination_rank_is_bit_exact(extra):
    """bench.py --emulate-world: rank 0's step with N-1 chunks of records arriving as device copies and expanded on
    the side stream; every row of the gathered outputs == a dense trace of that rank's rays"""
    r = _bench(["--subdiv", "5", "--res", "256", "--steps", "6", "--warmup", "3"] + extra)
    assert r["verified"] is True and r["emulated_world"] >= 3
    e = r["emulation"]
    assert e["rank0_ms_per_step"] > 0 and e["plain_1gpu_ms_per_step"] > 0 and e["implied_scaling_vs_1gpu"] > 0


def test_vectorised_expansion_matches_the_scalar_kernel_and_the_dense_trace(device):
    """tr_closest_expand: four rays per thread with 16-byte accesses (aligned rows) + the one-ray kernel for the
    rest; both == intersects_closest bit for bit, at every alignment of the row range"""
    import triro.backend.ops as hops
    from triro.ray.ray_optix import RayMeshIntersector
    v, f = W.headline_mesh(5)
    r = RayMeshIntersector(vertices=torch.from_numpy(v).to(device), faces=torch.from_numpy(f).to(device))
    rad = float(np.linalg.norm(v, axis=1).max())
    o_np, d_np = W.pinhole_grid(203, 101, distance=2.5 * rad)          # 20 503 rays: not a multiple of 4
    o = torch.from_numpy(np.ascontiguousarray(o_np)).to(device).reshape(-1, 3)
    d = torch.from_numpy(d_np).to(device).reshape(-1, 3)
    n = o.shape[0]
    exp = r.intersects_closest(o, d)
    packed = r.intersects_closest_packed(o, d)
    assert 0.2 < float(exp[0].float().mean()) < 0.95
    try:
        for opt in (1, 0, 2, 3):
            hops.set_option("expand4", opt)
            got = r.closest_expand(packed)
            for a, e in zip(got, exp):
                assert torch.equal(a, e), opt
            # row ranges at every alignment, into slices of full-size outputs (what the destination rank does)
            for lo in (0, 1, 2, 3, 4, 16, 1001):
                for hi in (n, n - 1, n - 2, n - 3, lo + 5, lo + 4, lo + 4096 + 1024, lo + 8192 + 5):
                    outs = (torch.zeros(n, dtype=torch.bool, device=device), torch.zeros(n, dtype=torch.bool, device=device),
                            torch.full((n,), -7, dtype=torch.int32, device=device), torch.full((n, 3), 9.0, device=device),
                            torch.full((n, 2), 9.0, device=device))
                    r.closest_expand(packed[lo:hi], outs=tuple(x[lo:hi] for x in outs))
                    for a, e in zip(outs, exp):
                        assert torch.equal(a[lo:hi], e[lo:hi]), (opt, lo, hi)
                    # nothing outside the range was written
                    assert int((outs[2][:lo] != -7).sum()) == 0 and int((outs[2][hi:] != -7).sum()) == 0
                    assert float((outs[3][:lo] != 9.0).sum()) == 0 and float((outs[3][hi:] != 9.0).sum()) == 0
    finally:
        hops.set_option("expand4", 1)
    # records that point outside the mesh (corrupt input) are misses, never out-of-bounds reads
    bad = packed.clone()
    bad[::7, 0] = 0x3fffffff
    hit = r.closest_expand(bad)[0]
    assert not bool(hit[::7].any())


def test_slot_form_records_expand_to_the_dense_outputs(device):
    """ABI 7: records that name the arena slot of the triangle (intersects_closest_packed(slots=True)) expand -- one
    48-byte triangle record per hit -- to the bits of intersects_closest; a second intersector built from the same
    mesh is a bit-identical replica and expands them just the same (what the destination rank of a sharded run does)"""
    from triro.ray.ray_optix import RayMeshIntersector
    v, f = W.headline_mesh(6)
    mk = lambda: RayMeshIntersector(vertices=torch.from_numpy(v).to(device), faces=torch.from_numpy(f).to(device))  # noqa: E731
    r, replica = mk(), mk()
    rad = float(np.linalg.norm(v, axis=1).max())
    o_np, d_np = W.pinhole_grid(301, 203, distance=2.5 * rad)
    o = torch.from_numpy(np.ascontiguousarray(o_np)).to(device)
    d = torch.from_numpy(d_np).to(device)
    lo, hi = v.min(0) * 1.5, v.max(0) * 1.5
    ho, hd = W.hash_rays_torch(3_000_001, 99, lo, hi, device=device)          # the streaming launch writes records too
    for oo, dd in ((o, d), (ho, hd)):
        exp = r.intersects_closest(oo, dd)
        rec_s = r.intersects_closest_packed(oo, dd, slots=True)
        rec_f = r.intersects_closest_packed(oo, dd)
        assert torch.equal(rec_s[:, 1:], rec_f[:, 1:]) and torch.equal(rec_s[:, 0] < 0, rec_f[:, 0] < 0)
        assert not torch.equal(rec_s[:, 0], rec_f[:, 0])                        # slots are not face indices
        b = oo.shape[:-1]
        for who in (r, replica):
            got = who.closest_expand(rec_s, batch_shape=b, slots=True)
            for a, e in zip(got, exp):
                assert torch.equal(a, e)
        # image rows in 8x8 pixel tiles (tr_closest_expand_slots_rows), into row slices of full-size outputs
        if oo.dim() == 3:
            H, Wd = 96, 352                                        # a multiple of 8 rows, of 32 pixels
            o2 = torch.from_numpy(np.ascontiguousarray(W.pinhole_grid(Wd, H, distance=2.5 * rad)[0])).to(device)
            d2 = torch.from_numpy(W.pinhole_grid(Wd, H, distance=2.5 * rad)[1]).to(device)
            exp2 = [x.reshape(H * Wd, *x.shape[2:]) for x in r.intersects_closest(o2, d2)]
            rec2 = r.intersects_closest_packed(o2, d2, slots=True)
            for rows in ((0, H), (8, 72), (16, 24)):
                a_, z_ = rows[0] * Wd, rows[1] * Wd
                outs = (torch.zeros(H * Wd, dtype=torch.bool, device=device), torch.zeros(H * Wd, dtype=torch.bool, device=device),
                        torch.full((H * Wd,), -7, dtype=torch.int32, device=device), torch.full((H * Wd, 3), 9.0, device=device),
                        torch.full((H * Wd, 2), 9.0, device=device))
                replica.closest_expand(rec2[a_:z_], outs=tuple(x[a_:z_] for x in outs), slots=True, row_length=Wd)
                for a, e in zip(outs, exp2):
                    assert torch.equal(a[a_:z_], e[a_:z_]), rows
                assert int((outs[2][:a_] != -7).sum()) == 0 and int((outs[2][z_:] != -7).sum()) == 0
            # shapes the tiled kernel does not take (rows not a multiple of 8, width not of 32) fall back, same bits
            got = r.closest_expand(rec2[:5 * Wd].contiguous(), slots=True, row_length=Wd)
            for a, e in zip(got, exp2):
                assert torch.equal(a, e[:5 * Wd])
        # small and unaligned row ranges (the one-ray kernel), corrupt slots
        n = rec_s.shape[0]
        flat = [x.reshape(n, *x.shape[len(b):]) for x in exp]
        for lo_, hi_ in ((0, 5), (3, 4000), (7, 5000 + 7)):
            got = r.closest_expand(rec_s[lo_:hi_].contiguous(), slots=True)
            for a, e in zip(got, flat):
                assert torch.equal(a, e[lo_:hi_])
        bad = rec_s.clone()
        bad[::5, 0] = 0x3fffffff
        assert not bool(r.closest_expand(bad, slots=True)[0].reshape(-1)[::5].any())


def test_destination_traces_dense_in_place(device):
    """intersects_closest_into: rows of preallocated full-size outputs == the ordinary call"""
    from triro.ray.ray_optix import RayMeshIntersector
    v, f = W.headline_mesh(4)
    r = RayMeshIntersector(vertices=torch.from_numpy(v).to(device), faces=torch.from_numpy(f).to(device))
    o_np, d_np = W.pinhole_grid(64, 48, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
    o, d = torch.from_numpy(np.ascontiguousarray(o_np)).to(device), torch.from_numpy(d_np).to(device)
    exp = [x.reshape(64 * 48, *x.shape[2:]) for x in r.intersects_closest(o, d)]
    n = 64 * 48
    outs = (torch.zeros(n, dtype=torch.bool, device=device), torch.zeros(n, dtype=torch.bool, device=device),
            torch.zeros(n, dtype=torch.int32, device=device), torch.zeros((n, 3), device=device), torch.zeros((n, 2), device=device))
    r.intersects_closest_into(o[:16], d[:16], tuple(x[:16 * 64] for x in outs))
    r.intersects_closest_into(o[16:], d[16:], tuple(x[16 * 64:] for x in outs))
    for a, e in zip(outs, exp):
        assert torch.equal(a, e)
    with pytest.raises(ValueError):
        r.intersects_closest_into(o[:16], d[:16], tuple(x[:5] for x in outs))


def test_async_closest_without_collectives_orders_the_side_stream(device):
    """closest_of_shard_async called directly with world == 1 and no forced collectives -- the
    expansion on the side stream must run behind the trace on the caller's stream (round 3 waited on itself)"""
    from triro.ray.ray_optix import RayMeshIntersector
    from triro.ray.sharded import ShardedRayMeshIntersector
    v, f = W.headline_mesh(7)
    r = RayMeshIntersector(vertices=torch.from_numpy(v).to(device), faces=torch.from_numpy(f).to(device))
    o_np, d_np = W.pinhole_grid(1024, 512, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
    o, d = torch.from_numpy(np.ascontiguousarray(o_np)).to(device), torch.from_numpy(d_np).to(device)
    exp = r.intersects_closest(o, d)
    S = ShardedRayMeshIntersector(r)
    assert S.world == 1 and not S.force_collectives

    class PackedOnly:      # a tracer without the dense in-place call: the destination packs and expands its own rows
        def __init__(self, inner):
            self.inner = inner
            self.intersects_closest_packed = inner.intersects_closest_packed
            self.closest_expand = inner.closest_expand
            self.intersects_closest = inner.intersects_closest
    for local in (r, PackedOnly(r)):
        S.local = local
        for k in range(6):
            # a long kernel in front of the trace on the caller's stream: an expansion that does not wait starts early
            junk = torch.randn(4096, 4096, device=device) @ torch.randn(4096, 4096, device=device)
            got = S.closest_of_shard_async(o, d, o.shape[0] * o.shape[1], batch_shape=o.shape[:2], dst=0, chunks=1 + k % 3).wait()
            torch.cuda.synchronize()
            for a, e in zip(got, exp):
                assert torch.equal(a, e), (type(local).__name__, k)
            del junk


def test_learned_order_survives_a_change_of_resolution(device):
    """VERDICT r03 #7b: the first launch of a new image resolution takes an order resampled from the previous
    resolution's block costs (k_sched_rescale) -- and returns the same bits as ever"""
    import triro.backend.ops as hops
    from triro.ray.ray_optix import RayMeshIntersector
    v, f = W.headline_mesh(6)
    R = OracleIntersector(v, f, mode=1)
    rad = float(np.linalg.norm(v, axis=1).max())

    def rays(res):
        o_np, d_np = W.pinhole_grid(res, res, distance=2.5 * rad)
        return o_np, d_np, torch.from_numpy(np.ascontiguousarray(o_np)).to(device), torch.from_numpy(d_np).to(device)
    try:
        for transfer in (1, 0):
            hops.set_option("order_transfer", transfer)
            r = RayMeshIntersector(vertices=torch.from_numpy(v).to(device), faces=torch.from_numpy(f).to(device))
            _, _, o1, d1 = rays(384)
            for _ in range(14):
                r.intersects_closest(o1, d1)
            assert r.as_wrapper.last_launch()["learned_order"] == 1
            for res in (512, 256, 640):               # up, down, up: every change starts from the previous shape
                o_np, d_np, o2, d2 = rays(res)
                hit, front, tri, loc, uv = r.intersects_closest(o2, d2)
                assert r.as_wrapper.last_launch()["learned_order"] == transfer, (transfer, res)
                eh, ef, et, el, eu = R.intersects_closest(o_np, d_np)
                assert np.array_equal(hit.cpu().numpy(), eh) and np.array_equal(tri.cpu().numpy(), et)
                assert np.array_equal(loc.cpu().numpy(), el) and np.array_equal(uv.cpu().numpy(), eu)
                assert np.array_equal(r.intersects_count(o2, d2).cpu().numpy(), R.intersects_count(o_np, d_np))
                for _ in range(3):
                    r.intersects_closest(o2, d2)
            # a flat batch in between cannot borrow an image's costs, and does not break anything
            hit = r.intersects_closest(o2.reshape(-1, 3), d2.reshape(-1, 3))[0]
            assert np.array_equal(hit.cpu().numpy(), eh.reshape(-1))
    finally:
        hops.set_option("order_transfer", 1)


def test_image_batches_with_a_ragged_last_tile_row(device):
    """Image-shaped batches whose row count is NOT a multiple of 8 (from 64 rows on) keep the tile launch shapes: the
    block -> ray map is laid over the batch padded to whole 8-row tiles and the padding is out of range.  Every query
    over 14 launches (plain first launch, learned order, split slots) == the oracle; the slot-form expansion in
    8 x 32 blocks with a partial last row of blocks == the dense trace, and writes nothing outside its rows."""
    from triro.ray.ray_optix import RayMeshIntersector
    v, f = W.headline_mesh(6)
    R = OracleIntersector(v, f, mode=1)
    r = RayMeshIntersector(vertices=torch.from_numpy(v).to(device), faces=torch.from_numpy(f).to(device))
    rad = float(np.linalg.norm(v, axis=1).max())
    for H, Wd in ((203, 352), (69, 1024), (1050, 96)):
        o_np, d_np = W.pinhole_grid(Wd, H, distance=2.5 * rad)
        o = torch.from_numpy(np.ascontiguousarray(o_np)).to(device)
        d = torch.from_numpy(d_np).to(device)
        eh, ef, et, el, eu = R.intersects_closest(o_np, d_np)
        ec = R.intersects_count(o_np, d_np)
        assert 0.05 < eh.mean() and (eh.mean() < 0.98 or Wd < 128)       # (the narrow band sees only the mesh)
        for k in range(14):
            hit, front, tri, loc, uv = r.intersects_closest(o, d)
            assert np.array_equal(hit.cpu().numpy(), eh) and np.array_equal(front.cpu().numpy(), ef), (H, Wd, k)
            assert np.array_equal(tri.cpu().numpy(), et) and np.array_equal(loc.cpu().numpy(), el), (H, Wd, k)
            assert np.array_equal(uv.cpu().numpy(), eu), (H, Wd, k)
            if k % 4 == 0:
                assert np.array_equal(r.intersects_count(o, d).cpu().numpy(), ec), (H, Wd, k)
                assert np.array_equal(r.intersects_first(o, d).cpu().numpy(), et), (H, Wd, k)
                assert np.array_equal(r.intersects_any(o, d).cpu().numpy(), eh), (H, Wd, k)
        assert r.as_wrapper.last_launch()["tile_rows_lg"] > 0, (H, Wd)          # a tile shape, not rows of 64 pixels
        loc_l, ray_l, tri_l = r.intersects_location(o, d)
        e_loc, e_ray, e_tri = R.intersects_location(o_np, d_np)
        assert np.array_equal(ray_l.cpu().numpy(), e_ray) and np.array_equal(tri_l.cpu().numpy(), e_tri)
        assert np.array_equal(loc_l.cpu().numpy(), e_loc)
        # expansion of a row range with a ragged number of rows into slices of full-size outputs
        n = H * Wd
        rec = r.intersects_closest_packed(o, d, slots=True)
        exp = [torch.from_numpy(x.reshape(n, *x.shape[2:])).to(device) for x in (eh, ef, et, el, eu)]
        for rows in ((0, H), (3, H - 2), (8, 8 + 65)):
            a_, z_ = rows[0] * Wd, rows[1] * Wd
            outs = (torch.zeros(n, dtype=torch.bool, device=device), torch.zeros(n, dtype=torch.bool, device=device),
                    torch.full((n,), -7, dtype=torch.int32, device=device), torch.full((n, 3), 9.0, device=device),
                    torch.full((n, 2), 9.0, device=device))
            r.closest_expand(rec[a_:z_], outs=tuple(x[a_:z_] for x in outs), slots=True, row_length=Wd)
            for a, e in zip(outs, exp):
                assert torch.equal(a[a_:z_], e[a_:z_]), (H, Wd, rows)
            assert int((outs[2][:a_] != -7).sum()) == 0 and int((outs[2][z_:] != -7).sum()) == 0
            assert int((outs[3][:a_] != 9.0).sum()) == 0 and int((outs[3][z_:] != 9.0).sum()) == 0


def test_four_byte_slot_records_finish_to_the_dense_outputs(device):
    """tr_intersects_closest_slots (4 B/ray: the arena slot of the nearest triangle) + tr_closest_from_slots (the end of
    a dense trace, from the ray and that one triangle) == tr_intersects_closest bit for bit: image and incoherent
    batches, strided and stride-0 rays, a replica's records, row ranges into slices of full-size outputs in both
    kernel shapes (blocks of 8 x 32 pixels / linear), small ranges, misses, corrupt slots."""
    from triro.ray.ray_optix import RayMeshIntersector
    v, f = W.headline_mesh(6)
    mk = lambda: RayMeshIntersector(vertices=torch.from_numpy(v).to(device), faces=torch.from_numpy(f).to(device))  # noqa: E731
    r, replica = mk(), mk()
    rad = float(np.linalg.norm(v, axis=1).max())
    H, Wd = 203, 352
    o_np, d_np = W.pinhole_grid(Wd, H, distance=2.5 * rad)
    o = torch.from_numpy(o_np[:1, :1].copy()).to(device).expand(H, Wd, 3)            # the README's stride-0 origin
    d = torch.from_numpy(d_np).to(device)
    lo, hi = v.min(0) * 1.5, v.max(0) * 1.5
    ho, hd = W.hash_rays_torch(3_000_001, 99, lo, hi, device=device)                 # the streaming launch
    # a non-contiguous view: every second ray of a twice as long batch
    so, sd = W.hash_rays_torch(200_000, 7, lo, hi, device=device)
    so, sd = so[::2], sd[::2]
    for name, (oo, dd) in {"image": (o, d), "hash": (ho, hd), "strided": (so, sd)}.items():
        exp = r.intersects_closest(oo, dd)
        assert 0.02 < float(exp[0].float().mean()) < 0.98, name
        for k in range(3):          # cold, learned order, split slots
            sl = replica.intersects_closest_slots(oo, dd)
        assert sl.dtype == torch.int32 and tuple(sl.shape) == (oo.numel() // 3,)
        assert torch.equal(sl.reshape(exp[0].shape) >= 0, exp[0]), name
        # against the 12-byte slot form
        rec = r.intersects_closest_packed(oo, dd, slots=True)
        assert torch.equal(torch.where(rec[:, 0] >= 0, rec[:, 0] & 0x3fffffff, torch.full_like(rec[:, 0], -1)), sl), name
        got = r.closest_from_slots(oo, dd, sl, row_length=Wd if name == "image" else 0)
        for a, e in zip(got, exp):
            assert torch.equal(a, e), name
        n = sl.shape[0]
        flat = [x.reshape(n, *x.shape[oo.dim() - 1:]) for x in exp]
        fo, fd = oo.reshape(-1, 3) if name != "image" else None, dd.reshape(-1, 3) if name != "image" else None
        ranges = ((0, 5), (3, 4000), (7, 5000 + 7), (1001, n - 3)) if name != "image" else ((0, H), (3, H - 2), (8, 8 + 65), (16, 24))
        for a_, z_ in ranges:
            if name == "image":
                ro, rd, rl = oo[a_:z_], dd[a_:z_], Wd
                a_, z_ = a_ * Wd, z_ * Wd
            else:
                ro, rd, rl = fo[a_:z_], fd[a_:z_], 0
            outs = (torch.zeros(n, dtype=torch.bool, device=device), torch.zeros(n, dtype=torch.bool, device=device),
                    torch.full((n,), -7, dtype=torch.int32, device=device), torch.full((n, 3), 9.0, device=device),
                    torch.full((n, 2), 9.0, device=device))
            r.closest_from_slots(ro, rd, sl[a_:z_], outs=tuple(x[a_:z_] for x in outs), row_length=rl)
            for a, e in zip(outs, flat):
                assert torch.equal(a[a_:z_], e[a_:z_]), (name, a_, z_)
            assert int((outs[2][:a_] != -7).sum()) == 0 and int((outs[2][z_:] != -7).sum()) == 0
        bad = sl.clone()
        bad[::5] = 0x3fffffff           # beyond the arena: a miss, no memory access
        assert not bool(r.closest_from_slots(oo, dd, bad)[0].reshape(-1)[::5].any())
    with pytest.raises(ValueError):
        r.closest_from_slots(ho, hd, sl[:5])


def test_four_byte_records_edge_cases(device):
    """the 4-byte records on the inputs the dense path is tested with: a single-triangle mesh (no hierarchy), zero
    rays, NaN / Inf rays (miss), a stride-0 origin with three batch dims and non-contiguous directions, wrong arguments"""
    from triro.ray.ray_optix import RayMeshIntersector
    Td = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)  # noqa: E731
    v1 = np.array([[0.5, -0.5, 0], [0, 0.5, 0], [-0.5, -0.5, 0]], np.float32)
    f1 = np.array([[0, 1, 2]], np.int32)
    r1 = RayMeshIntersector(vertices=Td(v1), faces=Td(f1))
    o = torch.tensor([[0, 0, 4.0], [10, 10, 10], [float("nan"), 0, 1], [0, 0, -4.0]], device=device)
    d = torch.tensor([[0, 0, -1.0], [0, 1, 0], [0, 0, -1], [0, 0, float("inf")]], device=device)
    dense = r1.intersects_closest(o, d)
    sl = r1.intersects_closest_slots(o, d)
    assert sl.tolist() == [0, -1, -1, -1]
    for a, e in zip(r1.closest_from_slots(o, d, sl), dense):
        assert torch.equal(a, e)
    z3 = torch.zeros(0, 3, device=device)
    empty = r1.intersects_closest_slots(z3, z3)
    assert empty.shape == (0,)
    assert [tuple(x.shape) for x in r1.closest_from_slots(z3, z3, empty)] == [(0,), (0,), (0,), (0, 3), (0, 2)]
    v, f = W.icosphere(4)
    r = RayMeshIntersector(vertices=Td(v), faces=Td(f))
    dn = W.pinhole_grid(40, 24)[1].reshape(2, 12, 40, 3)
    big = torch.zeros(2, 12, 40, 6, device=device)
    big[..., ::2] = Td(dn)
    dirs = big[..., ::2]
    assert not dirs.is_contiguous()
    org = torch.tensor([0.0, 0.0, 2.5], device=device).expand(2, 12, 40, 3)
    dense = r.intersects_closest(org, dirs)
    sl = r.intersects_closest_slots(org, dirs)
    assert tuple(sl.shape) == (960,) and 0.1 < float((sl >= 0).float().mean()) < 0.9
    for a, e in zip(r.closest_from_slots(org, dirs, sl), dense):
        assert a.shape == e.shape and torch.equal(a, e)
    with pytest.raises(ValueError):
        r.closest_from_slots(org, dirs, sl.float())                              # not int32
    with pytest.raises(ValueError):
        r.closest_from_slots(org, dirs, sl[:7])                                  # not one slot per ray
    with pytest.raises(ValueError):
        r.intersects_closest_slots(org, dirs, out=torch.zeros(7, dtype=torch.int32, device=device))


def test_replica_fingerprint_tells_slot_layouts_apart(device):
    """RayMeshIntersector.replica_fingerprint (what the sharded front end compares across ranks before slot-form
    records travel): equal for two builds of one mesh, for a rebuild and after a refit to the same vertices; different
    for another mesh and for the same mesh built with another node / triangle order."""
    import triro.backend.ops as hops
    from triro.ray.ray_optix import RayMeshIntersector
    Td = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)  # noqa: E731
    v, f = W.headline_mesh(5)
    a, b = RayMeshIntersector(vertices=Td(v), faces=Td(f)), RayMeshIntersector(vertices=Td(v), faces=Td(f))
    fa = a.replica_fingerprint()
    assert fa == b.replica_fingerprint() == a.replica_fingerprint()
    b.update_raw(Td(v), Td(f))
    assert b.replica_fingerprint() == fa
    v2 = W.displaced(W.icosphere(5)[0], seed=3, amplitude=0.05)
    c = RayMeshIntersector(vertices=Td(v2), faces=Td(f))
    assert c.replica_fingerprint() != fa
    # the same triangles in another face order: other slots for the same geometry?  No -- the builder orders the arena
    # by Morton code, not by input position -- but the face ids the slots lead to differ, and so may the layout of ties:
    # what has to hold is only that records of `a` expanded on a replica with an EQUAL fingerprint give a's dense outputs
    rad = float(np.linalg.norm(v, axis=1).max())
    o_np, d_np = W.pinhole_grid(128, 96, distance=2.5 * rad)
    o, d = Td(o_np), Td(d_np)
    dense = a.intersects_closest(o, d)
    for x, e in zip(b.closest_from_slots(o, d, a.intersects_closest_slots(o, d)), dense):
        assert torch.equal(x, e)
