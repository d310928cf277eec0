"""Ray-sharded multi-GPU front end (not in the reference, which is single-GPU:
base.cpp:15-17 holds one process-global context).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm).  The
path shards by rays: every ray is independent, so the flat ray range [0, n) is cut into
`world_size` contiguous chunks, every rank traces its chunk against ITS OWN replica of the BVH
(the builder is deterministic, so replicas built from the same mesh are identical) and there is
no exchange during traversal.  The only collective is the optional result gather:

  fixed-size outputs (any/first/closest/count): one padded `gather` (or `all_gather`) per
      output tensor, straight into slices of the destination;
  variable-size outputs (location, stream compaction): `all_gather` of the per-rank row
      counts, then a padded gather of the rows; `ray_idx` is already global because the
      kernels add `ray_base` (include/triro_hip.h: tr_intersects_location_fill,
      tr_compact_closest).

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

    def _gather_fixed(self, x: torch.Tensor, n: int, dst: Optional[int]):
        """x: this rank's [m, ...] rows -> [n, ...] on dst (None = all ranks)"""
        if self.world == 1:
            return x
        per = (n + self.world - 1) // self.world
        isbool = x.dtype == torch.bool
        src = x.view(torch.uint8) if isbool else x
        pad = torch.zeros((per, *src.shape[1:]), dtype=src.dtype, device=src.device)
        pad[: src.shape[0]] = src
        want = dst is None or self.rank == dst
        bufs = [torch.empty_like(pad) for _ in range(self.world)] if want else None
        if dst is None:
            dist.all_gather(bufs, pad, group=self.group)
        else:
            dist.gather(pad, bufs, dst=dst, group=self.group)
        if not want:
            return None
        parts = []
        for r in range(self.world):
            lo, hi = shard_bounds(n, self.world, r)
            parts.append(bufs[r][: hi - lo])
        out = torch.cat(parts, 0)
        return out.view(torch.bool) if isbool else out

    def _gather_rows(self, xs: Sequence[torch.Tensor], dst: Optional[int]):
        """variable-length row sets (same length within xs) -> concatenated in rank order"""
        if self.world == 1:
            return list(xs)
        dev = xs[0].device
        cnt = torch.tensor([xs[0].shape[0]], dtype=torch.int64, device=dev)
        cnts = [torch.zeros_like(cnt) for _ in range(self.world)]
        dist.all_gather(cnts, cnt, group=self.group)
        counts = [int(c.item()) for c in cnts]
        mx = max(counts) if counts else 0
        outs = []
        want = dst is None or self.rank == dst
        for x in xs:
            isbool = x.dtype == torch.bool
            src = x.view(torch.uint8) if isbool else x
            pad = torch.zeros((mx, *src.shape[1:]), dtype=src.dtype, device=dev)
            pad[: src.shape[0]] = src
            bufs = [torch.empty_like(pad) for _ in range(self.world)] if want else None
            if dst is None:
                dist.all_gather(bufs, pad, group=self.group)
            else:
                dist.gather(pad, bufs, dst=dst, group=self.group)
            if want:
                o = torch.cat([bufs[r][: counts[r]] for r in range(self.world)], 0)
                outs.append(o.view(torch.bool) if isbool else o)
            else:
                outs.append(None)
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
