"""Child process of tests/test_gpu_round3.py::test_sharded_exchange_on_rccl_world1: the collectives
of triro.ray.sharded on the REAL backend ("nccl" = RCCL) in a communicator of one rank -- what a
single-GPU box can check of the multi-GPU path (the receive-into-place exchange had only
ever run under gloo).  A self-gather moves no data between GPUs but goes through RCCL's argument
checks, dtype support, receive-into-views and async work handles.  Prints OK on success."""
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

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29611")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
try:
    v, f = W.headline_mesh(5)
    r = RayMeshIntersector(vertices=torch.from_numpy(v).to(dev), faces=torch.from_numpy(f).to(dev))
    o_np, d_np = W.pinhole_grid(256, 192, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
    o, d = torch.from_numpy(np.ascontiguousarray(o_np)).to(dev), torch.from_numpy(d_np).to(dev)
    exp = r.intersects_closest(o, d)
    assert 0.2 < float(exp[0].float().mean()) < 0.9
    for mode in ("packed", "dense", "padded"):
        S = ShardedRayMeshIntersector(r, gather_mode=mode, force_collectives=True)
        for dst in (0, None):
            for chunks in (1, 3):
                got = S.intersects_closest(o, d, dst=dst, chunks=chunks)
                for a, e in zip(got, exp):
                    assert torch.equal(a, e), (mode, dst, chunks)
        h1 = S.intersects_closest_async(o, d, dst=0, chunks=2)          # two queries in flight
        h2 = S.intersects_closest_async(o, d, dst=0, chunks=4)
        for h in (h1, h2):
            for a, e in zip(h.wait(), exp):
                assert torch.equal(a, e), mode
        assert torch.equal(S.intersects_count(o, d, dst=0), r.intersects_count(o, d))
        assert torch.equal(S.intersects_any(o, d, dst=None), r.intersects_any(o, d))
        assert torch.equal(S.intersects_first(o, d, dst=0), r.intersects_first(o, d))
        for a, e in zip(S.intersects_location(o, d, dst=0), r.intersects_location(o, d)):
            assert torch.equal(a, e), mode
        for a, e in zip(S.intersects_closest(o, d, stream_compaction=True, dst=None), r.intersects_closest(o, d, stream_compaction=True)):
            assert torch.equal(a, e), mode
    # grouped point-to-point ops (the ragged-chunk path) on RCCL: a self send / receive pair per dtype
    for dt in (torch.uint8, torch.int32, torch.float32, torch.int64):
        src = (torch.arange(3000, device=dev) % 251).to(dt).reshape(1000, 3)
        dstt = torch.zeros(1500, 3, dtype=dt, device=dev)
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.irecv, dstt[250:1250], 0), dist.P2POp(dist.isend, src, 0)])
        for q in reqs:
            q.wait()
        torch.cuda.synchronize()
        assert torch.equal(dstt[250:1250], src) and int(dstt[:250].sum()) == 0
    torch.cuda.synchronize()
    print("OK")
finally:
    dist.destroy_process_group()
