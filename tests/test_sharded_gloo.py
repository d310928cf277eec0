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
        self.v, self.f = np.asarray(v, np.float32), np.asarray(f, np.int64)
        self.packed_calls = []       # (rays, shape) of every packed trace: the chunking under test

    # the packed pipeline of triro.ray.sharded: 12 B/ray records and their expansion.  The stand-in
    # keeps (tri | front << 30, w bits, u bits) and re-derives loc from them in numpy -- consistent with
    # its own intersects_closest below (which is expand(packed)), not with the product's arithmetic:
    # what is tested here is the exchange, the chunking and the row bookkeeping.
    def intersects_closest_packed(self, o, d, out=None):
        self.packed_calls.append((o.numel() // 3, tuple(o.shape)))
        hit, front, tri, loc, uv = self.R.intersects_closest(o.numpy().reshape(-1, 3), d.numpy().reshape(-1, 3))
        rec = np.empty((len(hit), 3), np.int32)
        rec[:, 0] = np.where(hit, tri | (front.astype(np.int32) << 30), -1)
        rec[:, 1:] = np.ascontiguousarray(uv, np.float32).view(np.int32)
        t = torch.from_numpy(rec)
        if out is None:
            return t
        out.copy_(t)
        return out

    def closest_expand(self, packed, batch_shape=None, outs=None):
        rec = packed.numpy()
        hit = rec[:, 0] >= 0
        tri = np.where(hit, rec[:, 0] & 0x3fffffff, -1).astype(np.int32)
        front = hit & ((rec[:, 0] >> 30) & 1).astype(bool)
        uv = np.ascontiguousarray(rec[:, 1:]).view(np.float32).copy()
        w, u = uv[:, 0:1], uv[:, 1:2]
        tv = self.v[self.f[np.where(hit, tri, 0)]]
        loc = (w * tv[:, 0] + u * tv[:, 1] + (np.float32(1) - w - u) * tv[:, 2]).astype(np.float32)
        loc[~hit] = 0
        uv[~hit] = 0
        res = self._t(hit, front, tri, loc, uv)
        if outs is not None:
            for dst_, src_ in zip(outs, res):
                dst_.copy_(src_)
            return outs
        b = tuple(batch_shape) if batch_shape is not None else (len(hit),)
        return (res[0].reshape(b), res[1].reshape(b), res[2].reshape(b), res[3].reshape(*b, 3), res[4].reshape(*b, 2))

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
        if not stream_compaction:
            return self.closest_expand(self.intersects_closest_packed(o, d), batch_shape=o.shape[:-1])
        hit, front, tri, loc, uv = self.intersects_closest(o, d)
        hit = hit.reshape(-1)
        ridx = torch.arange(hit.numel(), dtype=torch.int32)[hit]
        return hit.reshape(o.shape[:-1]), front.reshape(-1)[hit], ridx, tri.reshape(-1)[hit], loc.reshape(-1, 3)[hit], uv.reshape(-1, 2)[hit]

    def intersects_location(self, o, d):
        return self._t(*self.R.intersects_location(o.numpy(), d.numpy()))


class CpuLocalInto(CpuLocal):
    """... with the destination rank's dense in-place trace (RayMeshIntersector.intersects_closest_into)"""

    def __init__(self, v, f):
        super().__init__(v, f)
        self.into_calls = []

    def intersects_closest_into(self, o, d, outs):
        self.into_calls.append((o.numel() // 3, tuple(o.shape)))
        res = self.closest_expand(CpuLocal.intersects_closest_packed(self, o, d))
        self.packed_calls.pop()
        for dst_, src_ in zip(outs, res):
            dst_.copy_(src_.reshape(dst_.shape))
        return outs


class CpuLocalSlots(CpuLocalInto):
    """... and with the slot forms of the records (RayMeshIntersector.packed_slots / slot_records).  The stand-in's "slot"
    is the face index; closest_from_slots re-traces the rays it is handed and CHECKS that they are the rays the slots
    belong to -- the pairing of ray rows and record rows on the destination is what this exercises."""
    packed_slots = True
    slot_records = True

    def __init__(self, v, f):
        super().__init__(v, f)
        self.slot_calls, self.from_calls = [], []

    def intersects_closest_packed(self, o, d, out=None, slots=False):
        return CpuLocal.intersects_closest_packed(self, o, d, out)

    def closest_expand(self, packed, batch_shape=None, outs=None, slots=False, row_length=0):
        return CpuLocal.closest_expand(self, packed, batch_shape, outs)

    def intersects_closest_slots(self, o, d, out=None):
        self.slot_calls.append((o.numel() // 3, tuple(o.shape)))
        rec = CpuLocal.intersects_closest_packed(self, o, d)
        self.packed_calls.pop()
        t = torch.where(rec[:, 0] >= 0, rec[:, 0] & 0x3fffffff, torch.full_like(rec[:, 0], -1)).to(torch.int32)
        if out is None:
            return t
        out.copy_(t)
        return out

    def closest_from_slots(self, o, d, slots, outs=None, row_length=0):
        self.from_calls.append((o.numel() // 3, tuple(o.shape), int(row_length)))
        rec = CpuLocal.intersects_closest_packed(self, o.contiguous(), d.contiguous())
        self.packed_calls.pop()
        mine = torch.where(rec[:, 0] >= 0, rec[:, 0] & 0x3fffffff, torch.full_like(rec[:, 0], -1)).to(torch.int32)
        assert torch.equal(mine, slots.reshape(-1)), "the destination paired record rows with the wrong rays"
        return CpuLocal.closest_expand(self, rec, batch_shape=o.shape[:-1], outs=outs)


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
        S = ShardedRayMeshIntersector(CpuLocal(v, f), gather_mode="dense")
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
        # ---- the packed, chunked, asynchronous closest-hit pipeline (round 3) ---------------------
        P = ShardedRayMeshIntersector(CpuLocal(v, f))                 # gather_mode "packed" is the default
        ok &= P._can_pack()
        allocs.clear()
        P._alloc = counting_alloc
        got = P.intersects_closest(o, d, dst=0, chunks=3)             # ragged shards (426 / 425), 3 chunks each
        lo_, hi_ = shard_bounds(851, world, rank)
        ok &= [c[0] for c in P.local.packed_calls] == [shard_bounds(hi_ - lo_, 3, k)[1] - shard_bounds(hi_ - lo_, 3, k)[0] for k in range(3)]
        if rank == 0:       # the full packed buffer + the five outputs; the other rank only its own records
            ok &= allocs == [((851, 3), torch.int32), ((26 * 864,), torch.uint8)]      # the records + ONE pool for the five outputs
            ok &= got[0].shape == (23, 37) and got[0].dtype == torch.bool and got[3].shape == (23, 37, 3) and got[2].dtype == torch.int32
            for a, e in zip(got, exp):
                ok &= torch.equal(a.reshape(e.shape), e)
        else:
            ok &= allocs == [((425, 3), torch.int32)] and got is None
        for dst_, ch in ((1, 1), (None, 2), (None, 5)):
            h = P.intersects_closest_async(o2, d2, dst=dst_, chunks=ch)     # equal shards: one collective per chunk
            g2 = h.wait()
            if dst_ is None or dst_ == rank:
                for a, e in zip(g2, e2):
                    ok &= torch.equal(a, e)
            else:
                ok &= g2 is None
        # an image-shaped batch cut at row boundaries keeps its shape on every rank (tiles on the GPU)
        o3_np, d3_np = W.pinhole_grid(37, 24)
        o3, d3 = torch.from_numpy(np.ascontiguousarray(o3_np)), torch.from_numpy(d3_np)
        P.local.packed_calls.clear()
        g3 = P.intersects_closest(o3, d3, dst=0, chunks=4)
        ok &= [c[1] for c in P.local.packed_calls] == [(3, 37, 3)] * 4
        e3 = ref.intersects_closest(o3, d3)
        if rank == 0:
            ok &= g3[0].shape == (24, 37) and g3[3].shape == (24, 37, 3)
            for a, e in zip(g3, e3):
                ok &= torch.equal(a, e)
        ok &= torch.equal(P.intersects_count(o3, d3, dst=None), ref.intersects_count(o3, d3))
        # fewer rays than chunks: every rank must still cut its shard into the same number of chunks
        for n_small in (5, 2, 1):
            g_s = P.intersects_closest(fo[:n_small], fd[:n_small], dst=0, chunks=4)
            e_s = ref.intersects_closest(fo[:n_small], fd[:n_small])
            if rank == 0:
                for a, e in zip(g_s, e_s):
                    ok &= torch.equal(a, e)
        # a rank that holds ONLY its shard (bench.py c5ii): closest_of_shard_async
        h = P.closest_of_shard_async(fo[lo_:hi_], fd[lo_:hi_], 851, dst=0, chunks=2)
        g4 = h.wait()
        if rank == 0:
            for a, e in zip(g4, exp):
                ok &= torch.equal(a.reshape(e.shape), e)
        # ---- round 4: the destination traces dense in place, weighted shards, host staging ---------------
        from triro.ray.sharded import weighted_bounds, auto_dst_share, dst_bounds
        for stage in (False, True):          # True: every exchange through the host-staging transport (gloo + device tensors)
            for share in (None, 0.5, 0.0):
                D = ShardedRayMeshIntersector(CpuLocalInto(v, f), dst_share=share, stage_through_host=stage or None)
                for dst_, ch in ((0, 3), (1, 1), (None, 2)):
                    D.local.into_calls.clear(); D.local.packed_calls.clear()
                    g6 = D.intersects_closest(o, d, dst=dst_, chunks=ch)
                    bb = D.bounds(851, dst_, 37, weighted=True)
                    if share == 0.5 and dst_ is not None:
                        ok &= bb == dst_bounds(851, world, dst_, 0.5, 37)
                        ok &= (bb[dst_][1] - bb[dst_][0]) < (bb[1 - dst_][1] - bb[1 - dst_][0])
                    if dst_ is None or dst_ == rank:
                        for a, e in zip(g6, exp):
                            ok &= torch.equal(a.reshape(e.shape), e)
                    else:
                        ok &= g6 is None
                    mine_rays = bb[rank][1] - bb[rank][0]
                    if dst_ == rank:       # dense in place, nothing packed on the destination
                        ok &= sum(c[0] for c in D.local.into_calls) == mine_rays and not D.local.packed_calls
                    else:
                        ok &= sum(c[0] for c in D.local.packed_calls) == mine_rays and not D.local.into_calls
                # image batch, weighted shards cut at row boundaries: both ranks keep the image shape
                g7 = D.intersects_closest(o3, d3, dst=0, chunks=2)
                if share == 0.5:
                    ok &= all(lo_ % 37 == 0 and hi_ % 37 == 0 for lo_, hi_ in D.bounds(37 * 24, 0, 37, weighted=True))
                if rank == 0:
                    for a, e in zip(g7, e3):
                        ok &= torch.equal(a, e)
                # the other queries through the same transport
                ok &= torch.equal(D.intersects_count(o, d, dst=None).reshape(-1), ref.intersects_count(fo, fd))
                l8 = D.intersects_location(o, d, dst=1)
                if rank == 1:
                    for a, e in zip(l8, ref.intersects_location(fo, fd)):
                        ok &= torch.equal(a, e)
                c8 = D.intersects_closest(o, d, stream_compaction=True, dst=None)
                for a, e in zip(c8, ref.intersects_closest(fo, fd, stream_compaction=True)):
                    ok &= torch.equal(a.reshape(e.shape), e)
        # ---- 4-byte records: the batch is visible on every rank, so the destination holds the rays -------------
        for stage in (False, True):
            for share in (None, 0.5):
                Z = ShardedRayMeshIntersector(CpuLocalSlots(v, f), dst_share=share, stage_through_host=stage or None)
                ok &= Z.slot_records
                for dst_, ch in ((0, 3), (1, 1), (None, 2)):
                    for lst in (Z.local.into_calls, Z.local.packed_calls, Z.local.slot_calls, Z.local.from_calls):
                        lst.clear()
                    g9 = Z.intersects_closest(o, d, dst=dst_, chunks=ch)
                    bb = Z.bounds(851, dst_, 37, weighted=True)
                    mine_rays = bb[rank][1] - bb[rank][0]
                    if dst_ is None or dst_ == rank:
                        for a, e in zip(g9, exp):
                            ok &= torch.equal(a.reshape(e.shape), e)
                        # the destination finished (only) the rays somebody else traced -- all of them with dst=None
                        ok &= sum(c[0] for c in Z.local.from_calls) == (851 if dst_ is None else 851 - mine_rays)
                    else:
                        ok &= g9 is None and not Z.local.from_calls
                    ok &= not Z.local.packed_calls               # no 12-byte record anywhere
                    if dst_ == rank:
                        ok &= sum(c[0] for c in Z.local.into_calls) == mine_rays and not Z.local.slot_calls
                    else:
                        ok &= sum(c[0] for c in Z.local.slot_calls) == mine_rays and not Z.local.into_calls
                # image batch with a stride-0 origin (the README's camera): rows stay views, row_length is handed on
                Z.local.from_calls.clear()
                ob = o3[:1, :1].expand(24, 37, 3)
                g10 = Z.intersects_closest(ob, d3, dst=0, chunks=2)
                e10 = ref.intersects_closest(ob.reshape(-1, 3), d3.reshape(-1, 3))
                if rank == 0:
                    for a, e in zip(g10, e10):
                        ok &= torch.equal(a.reshape(e.shape), e)
                    ok &= all(c[2] == 37 and len(c[1]) == 3 for c in Z.local.from_calls) and len(Z.local.from_calls) > 0
                # a rank that holds only its shard hands all_rays in on the destination
                bb = Z.bounds(851, 0, 1, weighted=True)
                h = Z.closest_of_shard_async(fo[bb[rank][0]:bb[rank][1]], fd[bb[rank][0]:bb[rank][1]], 851, dst=0, chunks=2,
                                             bounds=bb, records="slot", all_rays=(fo, fd) if rank == 0 else None)
                g11 = h.wait()
                if rank == 0:
                    for a, e in zip(g11, exp):
                        ok &= torch.equal(a.reshape(e.shape), e)
        # ---- replica handshake: slot-form records only travel when every rank's replica has the same slot layout --------
        class Fingerprinted(CpuLocalSlots):
            def __init__(self, v_, f_, fp):
                super().__init__(v_, f_)
                self.fp, self.fp_calls = fp, 0

            def replica_fingerprint(self):
                self.fp_calls += 1
                return self.fp
        for differ in (False, True):
            F = ShardedRayMeshIntersector(Fingerprinted(v, f, 1234 + (rank if differ else 0)))
            for _ in range(2):
                g12 = F.intersects_closest(o, d, dst=0, chunks=2)
                if rank == 0:
                    for a, e in zip(g12, exp):
                        ok &= torch.equal(a.reshape(e.shape), e)
            ok &= F.local.fp_calls == 1                                     # asked once per hierarchy
            ok &= F.slot_records == (not differ) and F.slots == (not differ)
            if differ:      # every rank fell back to the 12-byte records alike: nothing slot-shaped was traced or finished
                ok &= not F.local.slot_calls and not F.local.from_calls
                ok &= (len(F.local.packed_calls) > 0) == (rank != 0)
            else:
                ok &= not F.local.packed_calls and (len(F.local.slot_calls) > 0) == (rank != 0)
        ok &= 0.0 < auto_dst_share(8) < 1.0 and auto_dst_share(1) == 1.0
        # round 1's padded exchange stays selectable (fallback until the in-place path has run on RCCL)
        Q = ShardedRayMeshIntersector(CpuLocal(v, f), gather_mode="padded")
        g5 = Q.intersects_closest(o, d, dst=0)
        l5 = Q.intersects_location(o, d, dst=None)
        if rank == 0:
            for a, e in zip(g5, exp):
                ok &= torch.equal(a.reshape(e.shape), e)
        for a, e in zip(l5, ref.intersects_location(fo, fd)):
            ok &= torch.equal(a, e)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from conftest import free_port
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == {0: True, 1: True}


def test_weighted_bounds():
    from triro.ray.sharded import weighted_bounds
    for n in (0, 1, 7, 1000, 1024 * 1024, 100_000_000):
        for w in ([1, 1], [0.5, 1, 1, 1], [0.0, 1, 1], [1, 1, 1, 1, 1, 1, 1, 0.52]):
            for q in (1, 8, 1024):
                ch = weighted_bounds(n, w, q)
                assert ch[0][0] == 0 and ch[-1][1] == n and all(ch[i][1] == ch[i + 1][0] for i in range(len(w) - 1))
                assert all(hi >= lo for lo, hi in ch)
                if n % q == 0:
                    assert all(lo % q == 0 for lo, _ in ch)
    ch = weighted_bounds(800, [0.5, 1, 1, 1], 1)
    assert abs((ch[0][1] - ch[0][0]) - 800 * 0.5 / 3.5) <= 1


def test_shard_bounds():
    from triro.ray.sharded import shard_bounds
    for n in (0, 1, 7, 8, 9, 1000003):
        for w in (1, 2, 3, 8):
            chunks = [shard_bounds(n, w, r) for r in range(w)]
            assert chunks[0][0] == 0 and chunks[-1][1] == n
            assert all(chunks[i][1] == chunks[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in chunks]
            assert max(sizes) - min(sizes) <= 1


def test_dst_bounds_gives_equal_peer_shards():
    from triro.ray.sharded import dst_bounds
    for n in (0, 1, 5, 851, 1024 * 1024, 8192 * 1024, 100_000_000):
        for world in (1, 2, 3, 8):
            for dst in sorted({0, world - 1, world // 2}):
                for share in (0.0, 0.25, 0.47, 1.0):
                    for q in (1, 37, 1024):
                        b = dst_bounds(n, world, dst, share, q)
                        assert len(b) == world and b[0][0] == 0 and b[-1][1] == n
                        assert all(b[k][1] == b[k + 1][0] for k in range(world - 1))
                        sizes = [z - a for a, z in b]
                        assert len({sizes[r] for r in range(world) if r != dst}) <= 1          # the peers: equal
                        if n % q == 0 and q > 1:
                            assert all(a % q == 0 for a, _ in b)
                        if world > 1 and share == 1.0 and n % (world * q) == 0:
                            assert len(set(sizes)) == 1                                       # even
                        if world > 1 and sizes[(dst + 1) % world] > 0:
                            assert sizes[dst] <= sizes[(dst + 1) % world] + (world - 1) * q    # (+ the remainder)
