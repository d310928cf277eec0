#!/usr/bin/env python3
"""Host time of ONE pipelined step of a ray-sharded closest-hit query: the Python driver (closest_of_shard_async:
Python + torch.distributed) against the native step (closest_of_shard_native: one C call, include/triro_rccl.h).

    python scripts/profile_pipeline_host.py [--world1] [--emulate N]        (one GPU)

--world1   : a communicator of ONE rank on the real backend (RCCL): the destination traces its image, no peers
--emulate N: the destination's side of a pretended N-rank world (the peers' 4-byte records are already there):
             its own trace + the expansion of N-1 shards, two streams

Prints microseconds of host time per step (the wall clock around ENQUEUING 200 steps, GPU work still in flight) for both
drivers, and checks that they return the same bits.  VERDICT r05 "next" #3: <= 60 us for the native step."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import workloads as W  # noqa: E402


def host_us(fn, n=200, warm=30):
    for _ in range(warm):
        fn().wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hs = [fn() for _ in range(n)]
    el = time.perf_counter() - t0
    out = None
    for h in hs:
        out = h.wait()
    torch.cuda.synchronize()
    return el / n * 1e6, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world1", action="store_true")
    ap.add_argument("--emulate", type=int, default=0)
    ap.add_argument("--res", type=int, default=1024)
    args = ap.parse_args()
    if not args.world1 and not args.emulate:
        args.world1, args.emulate = True, 8
    import triro.backend.ops as hops
    from triro.ray.ray_optix import RayMeshIntersector
    from triro.ray.sharded import EmulatedWorld, ShardedRayMeshIntersector, dst_bounds
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    v, f = W.headline_mesh(6)
    r = RayMeshIntersector(vertices=torch.from_numpy(v).to(dev), faces=torch.from_numpy(f).to(dev))
    res = args.res
    o, d = W.pinhole_grid(res, res, distance=2.5 * float(np.linalg.norm(v, axis=1).max()))
    o = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(o, d.shape))).to(dev)
    d = torch.from_numpy(d).to(dev)
    n = res * res
    if args.world1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29737")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        S = ShardedRayMeshIntersector(r, force_collectives=True)
        kw = dict(batch_shape=(res, res), dst=0)
        py, a = host_us(lambda: S.closest_of_shard_async(o, d, n, records="slot", all_rays=(o, d), **kw))
        nat, b = host_us(lambda: S.closest_of_shard_native(o, d, n, all_rays=(o, d), **kw))
        same = all(torch.equal(x, y) for x, y in zip(a, b))
        print(f"world 1 on RCCL, {n} rays: host time per step  python {py:.1f} us   native {nat:.1f} us   same bits: {same}")
        dist.destroy_process_group()
    if args.emulate:
        N = args.emulate
        n_total = N * n
        O = o.repeat(N, 1, 1)
        D = d.repeat(N, 1, 1)
        bounds = dst_bounds(n_total, N, 0, 0.39, res)
        rec = torch.empty((n_total,), dtype=torch.int32, device=dev)
        fo, fd = O.reshape(-1, 3), D.reshape(-1, 3)
        for k in range(1, N):
            a_, z_ = bounds[k]
            r.intersects_closest_slots(fo[a_:z_], fd[a_:z_], out=rec[a_:z_])
        E = EmulatedWorld(r, N, rec, arrival="none")
        a0, z0 = bounds[0]
        mo, md = O[a0 // res:z0 // res], D[a0 // res:z0 // res]
        kw = dict(batch_shape=(N * res, res), dst=0, bounds=bounds, row_quantum=res, all_rays=(O, D))
        py, a = host_us(lambda: E.closest_of_shard_async(mo, md, n_total, records="slot", **kw), n=100, warm=10)
        nat, b = host_us(lambda: E.closest_of_shard_native(mo, md, n_total, flags=hops.STEP_NO_EXCHANGE, records=rec, world=N, rank=0, **kw), n=100, warm=10)
        same = all(torch.equal(x, y) for x, y in zip(a, b))
        print(f"emulated world {N} (rank 0: {z0 - a0} rays + {n_total - (z0 - a0)} records), host time per step  python {py:.1f} us   native {nat:.1f} us   same bits: {same}")


if __name__ == "__main__":
    main()
