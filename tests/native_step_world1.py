"""Child process of tests/test_gpu_round6.py::test_native_step_*: the native N-rank step (include/triro_rccl.h,
csrc/gather_rccl.cpp) on ONE GPU -- what a single-GPU box can check of it:
  * NO_EXCHANGE: the destination's side of a pretended world (the peers' 4-byte records are already there), image and
    flat batches, weighted shards, several chunks, two steps in flight -- torch.equal to the plain call AND to the Python
    pipeline (closest_of_shard_async through EmulatedWorld);
  * LOOPBACK on REAL RCCL in a communicator of one rank: this rank plays every rank, the peers' chunks travel through
    ncclSend / ncclRecv (to itself) -- argument checks, dtype, stream ordering, grouped calls;
  * a peer's side (rank != dst) without a communicator: records only.
Prints OK on success."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import workloads as W  # noqa: E402
import triro.backend.ops as hops  # noqa: E402
from triro.ray.ray_optix import RayMeshIntersector  # noqa: E402
from triro.ray.sharded import EmulatedWorld, ShardedRayMeshIntersector, dst_bounds, shard_bounds  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
assert hops.rccl_available(), hops.get_rccl_module().tr_rccl_last_error()
v, f = W.headline_mesh(5)
r = RayMeshIntersector(vertices=torch.from_numpy(v).to(dev), faces=torch.from_numpy(f).to(dev))
rad = float(np.linalg.norm(v, axis=1).max())
N = 4
for shape in ("image", "flat"):
    if shape == "image":
        H, Wd = 96, 128
        o_np, d_np = W.pinhole_grid(Wd, H, distance=2.5 * rad)
        o = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(o_np, d_np.shape))).to(dev)
        d = torch.from_numpy(d_np).to(dev)
        n_total, bshape, q = H * Wd, (H, Wd), Wd
    else:
        n_total = 50_001
        o, d = W.hash_rays_torch(n_total, 7, v.min(0) * 1.5, v.max(0) * 1.5, device=dev)
        bshape, q = (n_total,), 1
    exp = r.intersects_closest(o, d)
    assert 0.1 < float(exp[0].float().mean()) < 0.95
    for bounds in ([shard_bounds(n_total // q, N, k) for k in range(N)], [(a // q, z // q) for a, z in dst_bounds(n_total, N, 0, 0.5, q)]):
        bounds = [(a * q, z * q) for a, z in bounds]
        # the peers' records, traced beforehand as a peer would
        rec = torch.full((n_total,), -7, dtype=torch.int32, device=dev)
        fo, fd = o.reshape(-1, 3), d.reshape(-1, 3)
        for k in range(1, N):
            a, z = bounds[k]
            r.intersects_closest_slots(fo[a:z], fd[a:z], out=rec[a:z])
        a0, z0 = bounds[0]
        my = (o[a0 // q:z0 // q], d[a0 // q:z0 // q]) if q > 1 else (fo[a0:z0], fd[a0:z0])
        E = EmulatedWorld(r, N, rec, arrival="none")
        for chunks in (1, 3):
            kw = dict(batch_shape=bshape, dst=0, chunks=chunks, bounds=bounds, row_quantum=q if q > 1 else None, all_rays=(o, d))
            py = E.closest_of_shard_async(my[0], my[1], n_total, records="slot", **kw).wait()
            h1 = E.closest_of_shard_native(my[0], my[1], n_total, flags=hops.STEP_NO_EXCHANGE, records=rec, world=N, rank=0, **kw)
            h2 = E.closest_of_shard_native(my[0], my[1], n_total, flags=hops.STEP_NO_EXCHANGE, records=rec, world=N, rank=0, **kw)   # two in flight
            for got in (h1.wait(), h2.wait()):
                torch.cuda.synchronize()
                for g, e, p in zip(got, exp, py):
                    assert torch.equal(g, e) and torch.equal(g, p), (shape, chunks, "NO_EXCHANGE")
            # the same through RCCL: one rank plays all four (send / recv to itself)
            S = ShardedRayMeshIntersector(r)
            got = S.closest_of_shard_native(my[0], my[1], n_total, flags=hops.STEP_LOOPBACK, world=N, rank=0, **kw).wait()
            torch.cuda.synchronize()
            for g, e in zip(got, exp):
                assert torch.equal(g, e), (shape, chunks, "LOOPBACK")
        # a peer's side: records of its shard, nothing else (no exchange without a communicator: NO_EXCHANGE)
        a1, z1 = bounds[1]
        mine = (o[a1 // q:z1 // q], d[a1 // q:z1 // q]) if q > 1 else (fo[a1:z1], fd[a1:z1])
        out_rec = torch.full((z1 - a1,), -9, dtype=torch.int32, device=dev)
        S = ShardedRayMeshIntersector(r)
        S.closest_of_shard_native(mine[0], mine[1], n_total, batch_shape=bshape, dst=0, chunks=2, bounds=bounds, row_quantum=q if q > 1 else None,
                                  flags=hops.STEP_NO_EXCHANGE, records=out_rec, world=N, rank=1).wait()
        torch.cuda.synchronize()
        assert torch.equal(out_rec, rec[a1:z1])
print("OK")
