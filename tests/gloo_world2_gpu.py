"""Child process of tests/test_gpu_round4.py::test_two_ranks_share_one_gpu_real_tracer: ONE of two ranks of a
gloo communicator that BOTH use cuda:0 with the real RayMeshIntersector (device tensors travel through the
host: triro.ray.sharded stages them).  Functional only -- never a measurement -- but the first time the
non-destination branch of the packed pipeline, the ragged point-to-point exchange, the weighted shards and
the side-stream / event ordering meet device tensors with more than one rank.  Every result must be
torch.equal to the unsharded call.  usage: gloo_world2_gpu.py RANK WORLD PORT; prints OK on success."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import workloads as W  # noqa: E402
from triro.ray.ray_optix import RayMeshIntersector  # noqa: E402
from triro.ray.sharded import ShardedRayMeshIntersector  # noqa: E402

rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("gloo", rank=rank, world_size=world)
try:
    v, f = W.headline_mesh(5)
    r = RayMeshIntersector(vertices=torch.from_numpy(v).to(dev), faces=torch.from_numpy(f).to(dev))
    rad = float(np.linalg.norm(v, axis=1).max())
    o_np, d_np = W.pinhole_grid(256, 192, distance=2.5 * rad)
    o, d = torch.from_numpy(np.ascontiguousarray(o_np)).to(dev), torch.from_numpy(d_np).to(dev)
    lo, hi = v.min(0) * 1.5, v.max(0) * 1.5
    ho, hd = W.hash_rays_torch(100_001, 99, lo, hi, start=0, device=dev)      # ragged: odd count, flat, incoherent
    # the replica handshake of the slot-form records: both ranks built the same mesh, the fingerprints agree
    S0 = ShardedRayMeshIntersector(r)
    assert S0.slots and S0.slot_records and S0._fp_ok is True
    batches = {"image": (o, d), "hash": (ho, hd)}
    for name, (bo, bd) in batches.items():
        exp = r.intersects_closest(bo, bd)
        exp_c = r.intersects_closest(bo, bd, stream_compaction=True)
        exp_l = r.intersects_location(bo, bd)
        exp_n = r.intersects_count(bo, bd)
        assert 0.02 < float(exp[0].float().mean()) < 0.95, name
        # "packed": 4-byte slot records (the batch is visible on every rank, the destination holds the rays);
        # "packed12": the 12-byte {slot, u, v} records of a destination without the rays; "dense": per-output gathers
        for mode in ("packed", "packed12", "dense"):
            for share in (None, 0.5):
                S = ShardedRayMeshIntersector(r, gather_mode=mode.rstrip("12"), dst_share=share)
                if mode == "packed12":
                    S.slot_records = False
                assert S._stage and S.world == world and (S.slot_records or mode != "packed")
                for dst in (0, 1 % world, None):
                    for chunks in (1, 3):
                        got = S.intersects_closest(bo, bd, dst=dst, chunks=chunks)
                        if dst is None or dst == rank:
                            for a, e in zip(got, exp):
                                assert torch.equal(a, e), (name, mode, share, dst, chunks)
                        else:
                            assert got is None
                    got = S.intersects_closest(bo, bd, stream_compaction=True, dst=dst)
                    if dst is None or dst == rank:
                        for a, e in zip(got, exp_c):
                            assert torch.equal(a, e), (name, mode, "compaction", dst)
                    got = S.intersects_location(bo, bd, dst=dst)
                    if dst is None or dst == rank:
                        for a, e in zip(got, exp_l):
                            assert torch.equal(a, e), (name, mode, "location", dst)
                    got = S.intersects_count(bo, bd, dst=dst)
                    if dst is None or dst == rank:
                        assert torch.equal(got, exp_n), (name, mode, "count", dst)
                # two queries in flight, then a rank that holds ONLY its shard (bench.py's c5ii path)
                h1 = S.intersects_closest_async(bo, bd, dst=0, chunks=2)
                h2 = S.intersects_closest_async(bo, bd, dst=0, chunks=4)
                for h in (h1, h2):
                    g = h.wait()
                    if rank == 0:
                        for a, e in zip(g, exp):
                            assert torch.equal(a, e), (name, mode, "in flight")
                if name == "hash":
                    n = bo.shape[0]
                    bb = S.bounds(n, 0, 1, weighted=True)
                    a_, z_ = bb[rank]
                    g = S.closest_of_shard_async(bo[a_:z_].contiguous(), bd[a_:z_].contiguous(), n, dst=0, chunks=3, bounds=bb).wait()
                    if rank == 0:
                        for a, e in zip(g, exp):
                            assert torch.equal(a, e), (name, mode, "own shard only")
                    if S.slot_records:     # ... whose destination was handed the whole batch: 4-byte records
                        g = S.closest_of_shard_async(bo[a_:z_].contiguous(), bd[a_:z_].contiguous(), n, dst=0, chunks=3, bounds=bb,
                                                     records="slot", all_rays=(bo, bd) if rank == 0 else None).wait()
                        if rank == 0:
                            for a, e in zip(g, exp):
                                assert torch.equal(a, e), (name, mode, "own shard only, slot records")
    torch.cuda.synchronize()
    dist.barrier()
    print("OK")
finally:
    dist.destroy_process_group()
