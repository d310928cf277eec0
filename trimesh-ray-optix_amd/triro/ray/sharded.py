"""Ray-sharded multi-GPU front end (not in the reference, which is single-GPU:
base.cpp:15-17 holds one process-global context).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm).  The
path shards by rays: every ray is independent, so the flat ray range [0, n) is cut into
`world_size` contiguous chunks, every rank traces its chunk against ITS OWN replica of the BVH
(the builder is deterministic, so replicas built from the same mesh are identical) and there is
no exchange during traversal.  The only collective is the optional result gather:

  fixed-size outputs (any/first/closest/count): one `gather` (or `all_gather_into_tensor`)
      per output tensor whose receive buffers are slices of the preallocated destination
      (grouped point-to-point receives when the chunks are ragged);
  variable-size outputs (location, stream compaction): `all_gather` of the per-rank row
      counts, then the same receive-into-place exchange with per-rank lengths; `ray_idx` is
      made global before the exchange.

`local` can be any object with the RayMeshIntersector query methods (tests inject a CPU
stand-in so the sharding logic runs under gloo without a GPU).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous chunk [lo, hi) of rank `rank`; chunks differ by at most one ray."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedRayMeshIntersector:
    def __init__(self, local, group: Optional[dist.ProcessGroup] = None):
        self.local = local
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    # ---- helpers -------------------------------------------------------------------------
    def _my_rays(self, origins: torch.Tensor, directions: torch.Tensor):
        """flat [n,3] views of this rank's chunk (the full batch is visible on every rank)"""
        b = origins.shape[:-1]
        n = origins.numel() // 3
        lo, hi = shard_bounds(n, self.world, self.rank)
        # expand() keeps stride-0 origins cheap; reshape copies only the chunk that is traced
        o = origins.expand(*b, 3).reshape(-1, 3)[lo:hi]
        d = directions.expand(*b, 3).reshape(-1, 3)[lo:hi]
        return b, n, lo, hi, o, d

    # Gathers write straight into slices of the preallocated destination: no padded per-rank
    # buffers, no torch.cat -- at the 100 M-ray config those were two extra full copies of up to
    # 2.6 GB.  Every allocation of this module goes through _alloc (tests count them).
    @staticmethod
    def _alloc(shape, dtype, device):
        return torch.empty(shape, dtype=dtype, device=device)

    def _exchange(self, src: torch.Tensor, out: Optional[torch.Tensor], bounds, dst: Optional[int]):
        """rows [bounds[r][0], bounds[r][1]) of `out` <- rank r's `src`, for every r; `out` exists
        on dst (all ranks when dst is None).  Receives land in views of `out`; the local chunk
        is one device copy.  Equal chunks use one collective (RCCL gather / all-gather =
        grouped send/recv over distinct xGMI links), ragged ones grouped point-to-point ops."""
        world, rank = self.world, self.rank
        sizes = [hi - lo for lo, hi in bounds]
        equal = len(set(sizes)) == 1
        want = dst is None or rank == dst
        if equal and sizes[0] > 0:
            if dst is None:
                dist.all_gather_into_tensor(out, src.contiguous(), group=self.group)
            else:
                views = [out[lo:hi] for lo, hi in bounds] if want else None
                dist.gather(src.contiguous(), views, dst=dst, group=self.group)
            return
        ops = []
        src = src.contiguous()
        if want:
            lo, hi = bounds[rank]
            out[lo:hi].copy_(src)
            for r in range(world):
                if r != rank and sizes[r] > 0:
                    ops.append(dist.P2POp(dist.irecv, out[bounds[r][0]:bounds[r][1]], r, group=self.group))
        if sizes[rank] > 0:
            targets = [r for r in range(world) if r != rank] if dst is None else ([dst] if rank != dst else [])
            for r in targets:
                ops.append(dist.P2POp(dist.isend, src, r, group=self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()

    def _gather_fixed(self, x: torch.Tensor, n: int, dst: Optional[int]):
        """x: this rank's [m, ...] rows -> [n, ...] on dst (None = all ranks)"""
        if self.world == 1:
            return x
        isbool = x.dtype == torch.bool
        src = x.view(torch.uint8) if isbool else x
        want = dst is None or self.rank == dst
        out = self._alloc((n, *src.shape[1:]), src.dtype, src.device) if want else None
        bounds = [shard_bounds(n, self.world, r) for r in range(self.world)]
        self._exchange(src, out, bounds, dst)
        if not want:
            return None
        return out.view(torch.bool) if isbool else out

    def _gather_rows(self, xs: Sequence[torch.Tensor], dst: Optional[int]):
        """variable-length row sets (same length within xs) -> concatenated in rank order"""
        if self.world == 1:
            return list(xs)
        dev = xs[0].device
        # the only extra exchange of the variable-size outputs: `world` row counts
        cnt = torch.tensor([xs[0].shape[0]], dtype=torch.int64, device=dev)
        cnts = self._alloc((self.world,), torch.int64, dev)
        dist.all_gather_into_tensor(cnts, cnt, group=self.group)
        counts = [int(c) for c in cnts.tolist()]
        bounds, acc = [], 0
        for c in counts:
            bounds.append((acc, acc + c))
            acc += c
        want = dst is None or self.rank == dst
        outs = []
        for x in xs:
            isbool = x.dtype == torch.bool
            src = x.view(torch.uint8) if isbool else x
            out = self._alloc((acc, *src.shape[1:]), src.dtype, dev) if want else None
            self._exchange(src, out, bounds, dst)
            outs.append((out.view(torch.bool) if isbool else out) if want else None)
        return outs

    # ---- queries (same names / return orders as RayMeshIntersector) -------------------------
    def intersects_any(self, origins, directions, dst: Optional[int] = 0):
        b, n, lo, hi, o, d = self._my_rays(origins, directions)
        out = self._gather_fixed(self.local.intersects_any(o, d), n, dst)
        return None if out is None else out.reshape(b)

    def intersects_first(self, origins, directions, dst: Optional[int] = 0):
        b, n, lo, hi, o, d = self._my_rays(origins, directions)
        out = self._gather_fixed(self.local.intersects_first(o, d), n, dst)
        return None if out is None else out.reshape(b)

    def intersects_count(self, origins, directions, dst: Optional[int] = 0):
        b, n, lo, hi, o, d = self._my_rays(origins, directions)
        out = self._gather_fixed(self.local.intersects_count(o, d), n, dst)
        return None if out is None else out.reshape(b)

    def intersects_closest(self, origins, directions, stream_compaction: bool = False,
                           dst: Optional[int] = 0):
        b, n, lo, hi, o, d = self._my_rays(origins, directions)
        if not stream_compaction:
            res = self.local.intersects_closest(o, d)
            outs = [self._gather_fixed(x, n, dst) for x in res]
            if outs[0] is None:
                return None
            hit, front, tri, loc, uv = outs
            return hit.reshape(b), front.reshape(b), tri.reshape(b), loc.reshape(*b, 3), uv.reshape(*b, 2)
        hit, front, ray_idx, tri, loc, uv = self.local.intersects_closest(o, d, stream_compaction=True)
        ray_idx = ray_idx + lo          # local -> global flat ray index
        hit_all = self._gather_fixed(hit, n, dst)
        front, ray_idx, tri, loc, uv = self._gather_rows([front, ray_idx, tri, loc, uv], dst)
        if hit_all is None:
            return None
        return hit_all.reshape(b), front, ray_idx, tri, loc, uv

    def intersects_location(self, origins, directions, dst: Optional[int] = 0):
        b, n, lo, hi, o, d = self._my_rays(origins, directions)
        loc, ray_idx, tri = self.local.intersects_location(o, d)
        ray_idx = ray_idx + lo
        loc, ray_idx, tri = self._gather_rows([loc, ray_idx, tri], dst)
        if loc is None:
            return None
        return loc, ray_idx, tri
