"""world_size-2 gloo tests (CPU) of the ray-sharding / gather logic of
triro.ray.sharded.ShardedRayMeshIntersector.  The local tracer is a CPU stand-in built on
the oracle (tests may use it); on the GPU box the same class wraps RayMeshIntersector and the
collectives run over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import workloads as W
from oracle.oracle import OracleIntersector

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class CpuLocal:
    """RayMeshIntersector-shaped CPU tracer (torch tensors in/out) backed by the oracle."""

    def __init__(self, v, f):
        self.R = OracleIntersector(v, f, 1, threads=2)

    @staticmethod
    def _t(*xs):
        return tuple(torch.from_numpy(np.ascontiguousarray(x)) for x in xs)

    def intersects_any(self, o, d):
        return self._t(self.R.intersects_any(o.numpy(), d.numpy()))[0]

    def intersects_first(self, o, d):
        return self._t(self.R.intersects_first(o.numpy(), d.numpy()))[0]

    def intersects_count(self, o, d):
        return self._t(self.R.intersects_count(o.numpy(), d.numpy()))[0]

    def intersects_closest(self, o, d, stream_compaction=False):
        return self._t(*self.R.intersects_closest(o.numpy(), d.numpy(), stream_compaction=stream_compaction))

    def intersects_location(self, o, d):
        return self._t(*self.R.intersects_location(o.numpy(), d.numpy()))


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from triro.ray.sharded import ShardedRayMeshIntersector, shard_bounds
        v, f = W.nested_shells(2, radii=(1.0, 0.7, 0.5, 0.35, 0.2))
        o_np, d_np = W.pinhole_grid(37, 23)              # 851 rays: not divisible by world
        o, d = torch.from_numpy(np.ascontiguousarray(o_np)), torch.from_numpy(d_np)
        S = ShardedRayMeshIntersector(CpuLocal(v, f))
        ref = CpuLocal(v, f)
        # every allocation of the gather path is recorded: receives must land in the final
        # outputs (no padded per-rank buffers, no concatenation copies)
        allocs = []
        real_alloc = ShardedRayMeshIntersector._alloc

        def counting_alloc(shape, dtype, device):
            allocs.append((tuple(shape), dtype))
            return real_alloc(shape, dtype, device)
        S._alloc = counting_alloc
        out = {}
        out["closest"] = S.intersects_closest(o, d, dst=0)
        # dst=0 gather of the five closest-hit outputs: rank 0 allocates exactly the five full
        # outputs, the other rank nothing at all
        want = [((851,), torch.uint8), ((851,), torch.uint8), ((851,), torch.int32),
                ((851, 3), torch.float32), ((851, 2), torch.float32)]
        alloc_ok = (allocs == want) if rank == 0 else (allocs == [])
        allocs.clear()
        out["closest_all"] = S.intersects_closest(o, d, dst=None)
        out["compact"] = S.intersects_closest(o, d, stream_compaction=True, dst=0)
        out["count"] = S.intersects_count(o, d, dst=0)
        out["any"] = S.intersects_any(o, d, dst=None)
        out["first"] = S.intersects_first(o, d, dst=0)
        out["location"] = S.intersects_location(o, d, dst=0)
        ok = True
        fo, fd = o.reshape(-1, 3), d.reshape(-1, 3)
        exp = ref.intersects_closest(fo, fd)
        if rank == 0:
            for a, e in zip(out["closest"], exp):
                ok &= torch.equal(a.reshape(e.shape), e)
            hit, front, ridx, tri, loc, uv = out["compact"]
            eh, ef, er, et, el, eu = ref.intersects_closest(fo, fd, stream_compaction=True)
            ok &= torch.equal(hit.reshape(-1), eh) and torch.equal(front, ef) and torch.equal(ridx, er)
            ok &= torch.equal(tri, et) and torch.equal(loc, el) and torch.equal(uv, eu)
            ok &= torch.equal(out["count"].reshape(-1), ref.intersects_count(fo, fd))
            ok &= torch.equal(out["first"].reshape(-1), ref.intersects_first(fo, fd))
            l, r, t = out["location"]
            el, er, et = ref.intersects_location(fo, fd)
            ok &= torch.equal(l, el) and torch.equal(r, er) and torch.equal(t, et)
            ok &= out["closest"][0].shape == (23, 37)
        else:
            ok &= out["closest"] is None and out["compact"] is None and out["location"] is None
        for a, e in zip(out["closest_all"], exp):          # all_gather variant: every rank
            ok &= torch.equal(a.reshape(e.shape), e)
        ok &= torch.equal(out["any"].reshape(-1), ref.intersects_any(fo, fd))
        ok &= alloc_ok
        # equal chunks (850 rays over 2 ranks) take the single-collective path
        o2, d2 = fo[:850], fd[:850]
        allocs.clear()
        e2 = ref.intersects_closest(o2, d2)
        for dst_ in (0, 1, None):
            g2 = S.intersects_closest(o2, d2, dst=dst_)
            if dst_ is None or dst_ == rank:
                for a, e in zip(g2, e2):
                    ok &= torch.equal(a, e)
            else:
                ok &= g2 is None
        c2 = S.intersects_count(o2, d2, dst=1)
        ok &= (torch.equal(c2, ref.intersects_count(o2, d2)) if rank == 1 else c2 is None)
        l2 = S.intersects_location(o2, d2, dst=None)          # variable-size rows to every rank
        for a, e in zip(l2, ref.intersects_location(o2, d2)):
            ok &= torch.equal(a, e)
        covered = sum(hi - lo for lo, hi in (shard_bounds(851, world, r) for r in range(world)))
        ok &= covered == 851 and shard_bounds(851, world, 0)[0] == 0
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == {0: True, 1: True}


def test_shard_bounds():
    from triro.ray.sharded import shard_bounds
    for n in (0, 1, 7, 8, 9, 1000003):
        for w in (1, 2, 3, 8):
            chunks = [shard_bounds(n, w, r) for r in range(w)]
            assert chunks[0][0] == 0 and chunks[-1][1] == n
            assert all(chunks[i][1] == chunks[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in chunks]
            assert max(sizes) - min(sizes) <= 1
