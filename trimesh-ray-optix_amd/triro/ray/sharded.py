"""Ray-sharded multi-GPU front end (not in the reference, which is single-GPU:
base.cpp:15-17 holds one process-global context).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm).  The
path shards by rays: every ray is independent, so the flat ray range [0, n) is cut into
`world_size` contiguous chunks, every rank traces its chunk against ITS OWN replica of the BVH
(the builder is deterministic, so replicas built from the same mesh are identical) and there is
no exchange during traversal.  The only collective is the optional result gather:

  closest hit (the headline query): every rank traces its shard in K chunks into 12-byte packed
      records {face | front << 30, u, v} (tr_intersects_closest_packed) and hands each chunk to ONE
      asynchronous exchange as soon as it is traced -- the exchange of chunk k runs on RCCL's stream
      while chunk k+1 is traced; the destination rank expands the records back into the five dense
      outputs (tr_closest_expand: bit-identical to tracing there) on a side stream as the chunks
      arrive.  12 B/ray over xGMI instead of 26, one collective per chunk instead of five per call.
  other fixed-size outputs (any/first/count): one `gather` (or `all_gather_into_tensor`) per output
      tensor whose receive buffers are slices of the preallocated destination (grouped
      point-to-point receives when the chunks are ragged);
  variable-size outputs (location, stream compaction): `all_gather` of the per-rank row
      counts, then the same receive-into-place exchange with per-rank lengths; `ray_idx` is
      made global before the exchange.

`local` can be any object with the RayMeshIntersector query methods (tests inject a CPU
stand-in so the sharding logic runs under gloo without a GPU).  The packed pipeline is used when
`local` has `intersects_closest_packed` / `closest_expand`; otherwise, and with
TRIRO_SHARDED_GATHER=dense, closest hits take the per-output exchange of the other queries;
TRIRO_SHARDED_GATHER=padded selects round 1's padded-buffer gather for every exchange (kept as a
fallback until the receive-into-place path has run on a multi-GPU RCCL node).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous chunk [lo, hi) of rank `rank`; chunks differ by at most one ray."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def default_chunks(rays_per_rank: int) -> int:
    """Chunks per shard of the packed closest-hit pipeline: at least ~3 M rays each (smaller launches
    lose the streaming launch's efficiency), at most 8."""
    return max(1, min(8, rays_per_rank // 3_000_000))


class PendingClosest:
    """An in-flight gathered closest-hit query (ShardedRayMeshIntersector.intersects_closest_async).
    wait(): makes the caller's current stream wait for the gather + expansion and returns
    (hit, front, tri, loc, uv) on the destination rank(s), None elsewhere."""

    def __init__(self, outputs, event=None, works=(), keep=()):
        self._outputs, self._event, self._works, self._keep = outputs, event, list(works), keep

    def wait(self):
        for w in self._works:      # CPU (gloo) path and non-destination ranks: plain completion
            w.wait()
        self._works = []
        if self._event is not None:
            torch.cuda.current_stream().wait_event(self._event)
            self._event = None
        self._keep = ()
        return self._outputs

    def __del__(self):
        # The buffers were allocated on the caller's stream and are written on the side stream: they
        # must not return to the allocator before that work is done.  wait() orders the caller's stream
        # behind it; a handle that is dropped without wait() blocks here instead (rare, and correct).
        ev = getattr(self, "_event", None)
        if ev is not None:
            try:
                ev.synchronize()
            except Exception:
                pass


class ShardedRayMeshIntersector:
    def __init__(self, local, group: Optional[dist.ProcessGroup] = None, gather_mode: Optional[str] = None,
                 force_collectives: bool = False):
        # force_collectives: run the collectives even in a communicator of ONE rank (tests: the RCCL
        # calls of this module on a single-GPU box; a self-gather moves nothing but takes every code path
        # of the equal-chunk exchange)
        self.force_collectives = force_collectives and dist.is_initialized()
        self.local = local
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.gather_mode = gather_mode or os.environ.get("TRIRO_SHARDED_GATHER", "packed")
        if self.gather_mode not in ("packed", "dense", "padded"):
            raise ValueError("gather_mode must be 'packed', 'dense' or 'padded'")
        self._side = None      # side stream of the destination rank (wait for chunk, expand)

    # ---- helpers -------------------------------------------------------------------------
    def _my_rays(self, origins: torch.Tensor, directions: torch.Tensor):
        """this rank's chunk of the batch (the full batch is visible on every rank).  An image-shaped
        batch [H, W, 3] that is cut at row boundaries keeps its shape (and its stride-0 broadcast), so
        the shard is traced with the image launch shapes (tiles); everything else becomes flat [m, 3]."""
        b = origins.shape[:-1]
        n = origins.numel() // 3
        lo, hi = shard_bounds(n, self.world, self.rank)
        if origins.dim() == 3 and b[1] > 0 and lo % b[1] == 0 and hi % b[1] == 0 and hi > lo:
            return b, n, lo, hi, origins[lo // b[1]:hi // b[1]], directions.expand(*b, 3)[lo // b[1]:hi // b[1]]
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

    def _exchange(self, src: torch.Tensor, out: Optional[torch.Tensor], bounds, dst: Optional[int],
                  async_op: bool = False) -> List:
        """rows [bounds[r][0], bounds[r][1]) of `out` <- rank r's `src`, for every r; `out` exists
        on dst (all ranks when dst is None).  Receives land in views of `out`; the local chunk
        is one device copy (none when `src` already IS that view).  Equal chunks use one collective
        (RCCL gather / all-gather = grouped send/recv over distinct xGMI links), ragged ones grouped
        point-to-point ops.  async_op: returns the outstanding work handles instead of waiting."""
        world, rank = self.world, self.rank
        sizes = [hi - lo for lo, hi in bounds]
        equal = len(set(sizes)) == 1
        want = dst is None or rank == dst
        if self.gather_mode == "padded":
            return self._exchange_padded(src, out, bounds, dst)
        if equal and sizes[0] > 0:
            src = src.contiguous()
            views = [out[lo:hi] for lo, hi in bounds] if want else None
            if want and views[rank].data_ptr() == src.data_ptr():
                src = views[rank]           # traced in place: the collective's self-copy is a no-op
            if dst is None:
                whole = bounds[0][0] == 0 and all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1)) and \
                    bounds[-1][1] == out.shape[0]
                if whole:
                    w = dist.all_gather_into_tensor(out, src, group=self.group, async_op=async_op)
                else:       # chunk k of every rank: the slices are not adjacent in `out`
                    w = dist.all_gather(views, src, group=self.group, async_op=async_op)
            else:
                w = dist.gather(src, views, dst=dst, group=self.group, async_op=async_op)
            return [w] if async_op and w is not None else []
        ops = []
        src = src.contiguous()
        if want:
            lo, hi = bounds[rank]
            mine = out[lo:hi]
            if mine.data_ptr() != src.data_ptr() and hi > lo:
                mine.copy_(src)
            for r in range(world):
                if r != rank and sizes[r] > 0:
                    ops.append(dist.P2POp(dist.irecv, out[bounds[r][0]:bounds[r][1]], r, group=self.group))
        if sizes[rank] > 0:
            targets = [r for r in range(world) if r != rank] if dst is None else ([dst] if rank != dst else [])
            for r in targets:
                ops.append(dist.P2POp(dist.isend, src, r, group=self.group))
        if not ops:
            return []
        reqs = dist.batch_isend_irecv(ops)
        if async_op:
            return list(reqs)
        for req in reqs:
            req.wait()
        return []

    def _exchange_padded(self, src, out, bounds, dst):
        """round 1's exchange: every rank pads its rows to the longest chunk, one (all_)gather of the
        padded buffers, copies into place.  Two extra copies; kept as the conservative fallback."""
        sizes = [hi - lo for lo, hi in bounds]
        m = max(sizes) if sizes else 0
        if m == 0:
            return []
        want = dst is None or self.rank == dst
        pad = torch.zeros((m, *src.shape[1:]), dtype=src.dtype, device=src.device)
        pad[:sizes[self.rank]].copy_(src)
        bufs = [torch.empty_like(pad) for _ in range(self.world)] if want else None
        if dst is None:
            dist.all_gather(bufs, pad, group=self.group)
        else:
            dist.gather(pad, bufs, dst=dst, group=self.group)
        if want:
            for r, (lo, hi) in enumerate(bounds):
                out[lo:hi].copy_(bufs[r][:hi - lo])
        return []

    def _gather_fixed(self, x: torch.Tensor, n: int, dst: Optional[int]):
        """x: this rank's [m, ...] rows -> [n, ...] on dst (None = all ranks)"""
        if self.world == 1 and not self.force_collectives:
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
        if self.world == 1 and not self.force_collectives:
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

    # ---- packed closest-hit pipeline ----------------------------------------------------------
    def _can_pack(self) -> bool:
        return (self.gather_mode == "packed" and hasattr(self.local, "intersects_closest_packed")
                and hasattr(self.local, "closest_expand"))

    def closest_of_shard_async(self, o: torch.Tensor, d: torch.Tensor, n_total: int, batch_shape=None,
                               dst: Optional[int] = 0, chunks: Optional[int] = None) -> PendingClosest:
        """Closest hit of ONE batch of `n_total` rays of which this rank holds (only) its shard `o`, `d`
        = rows shard_bounds(n_total, world, rank) of the batch; results for all n_total rays on rank
        `dst` (None: on every rank), shaped `batch_shape` (default [n_total]).  Pipeline per rank:
        trace chunk k (packed, 12 B/ray) -> asynchronous exchange of chunk k while chunk k+1 is traced
        -> on the destination, a side stream waits for chunk k and expands it into the dense outputs.
        Returns at once; PendingClosest.wait() orders the caller's stream behind the result."""
        world, rank = self.world, self.rank
        lo, hi = shard_bounds(n_total, world, rank)
        m = hi - lo
        if o.numel() // 3 != m:
            raise ValueError(f"rank {rank} holds {o.numel() // 3} rays but its shard of {n_total} is {m}")
        dev = o.device
        want = dst is None or rank == dst
        b = tuple(batch_shape) if batch_shape is not None else (n_total,)
        K = chunks if chunks else default_chunks(max(1, n_total // world))
        image = o.dim() == 3           # image-shaped shard: chunk by whole rows
        per_row = o.shape[1] if image else 1
        sizes = [shard_bounds(n_total, world, r) for r in range(world)]
        if image and any((z - a) % per_row for a, z in sizes):
            # (cannot happen through intersects_closest; a caller that hands in an image-shaped shard of a
            # batch whose other shards are not whole rows gets the flat path)
            o, d, image, per_row = o.reshape(-1, 3), d.expand(*o.shape).reshape(-1, 3), False, 1
        # every rank must cut its shard into the SAME number of chunks (one exchange per chunk): bound K by
        # the smallest shard, which every rank can compute
        K = max(1, min(K, min((z - a) // per_row for a, z in sizes)))
        # the destination traces straight into its slice of the full packed buffer
        packed_all = self._alloc((n_total, 3), torch.int32, dev) if want else None
        mine = packed_all[lo:hi] if want else (self._alloc((m, 3), torch.int32, dev) if m > 0 else None)
        outs = flat_outs = None
        if want:
            # the five dense outputs out of ONE allocation (26 B per ray: loc | uv | tri | hit | front, each
            # part aligned to 16 B): an allocator call costs as much as a small kernel launch
            n16 = (n_total + 15) // 16 * 16
            pool = self._alloc((26 * n16,), torch.uint8, dev)
            loc = pool[:12 * n16].view(torch.float32)[:3 * n_total].view(n_total, 3)
            uv = pool[12 * n16:20 * n16].view(torch.float32)[:2 * n_total].view(n_total, 2)
            tri = pool[20 * n16:24 * n16].view(torch.int32)[:n_total]
            hit = pool[24 * n16:25 * n16].view(torch.bool)[:n_total]
            front = pool[25 * n16:26 * n16].view(torch.bool)[:n_total]
            flat_outs = (hit, front, tri, loc, uv)
            outs = (hit.view(b), front.view(b), tri.view(b), loc.view(*b, 3), uv.view(*b, 2))
        cuda = dev.type == "cuda"
        side = None
        if cuda and want:
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            side = self._side
            side.wait_stream(torch.cuda.current_stream(dev))     # the allocations above are ready
        works_all = []
        for k in range(K):
            # chunk k of every rank (every rank can compute everybody's bounds)
            cb = []
            for r in range(world):
                rlo, rhi = shard_bounds(n_total, world, r)
                if image:      # rows of rank r's shard: all shards of an image batch are whole rows
                    rrows = (rhi - rlo) // per_row
                    a, z = shard_bounds(rrows, K, k)
                    cb.append((rlo + a * per_row, rlo + z * per_row))
                else:
                    a, z = shard_bounds(rhi - rlo, K, k)
                    cb.append((rlo + a, rlo + z))
            a, z = cb[rank][0] - lo, cb[rank][1] - lo
            if z > a:
                if image:
                    self.local.intersects_closest_packed(o[a // per_row:z // per_row], d[a // per_row:z // per_row], out=mine[a:z])
                else:
                    self.local.intersects_closest_packed(o[a:z], d[a:z], out=mine[a:z])
            if world > 1 or self.force_collectives:
                src = mine[a:z] if m > 0 else torch.empty((0, 3), dtype=torch.int32, device=dev)
                works = self._exchange(src, packed_all, cb, dst, async_op=True)
            else:
                works = []
            if want:
                def expand_chunk():
                    spans = [(ra, rz) for ra, rz in cb if rz > ra]
                    if spans and all(spans[j][1] == spans[j + 1][0] for j in range(len(spans) - 1)):
                        spans = [(spans[0][0], spans[-1][1])]          # one chunk per rank: one contiguous range
                    for ra, rz in spans:
                        self.local.closest_expand(packed_all[ra:rz], outs=tuple(x[ra:rz] for x in flat_outs))
                if side is not None:
                    with torch.cuda.stream(side):
                        for w in works:
                            w.wait()                                      # side stream behind the exchange of chunk k
                        side.wait_stream(torch.cuda.current_stream(dev))  # ... and behind the local trace of chunk k
                        expand_chunk()
                else:
                    for w in works:
                        w.wait()
                    expand_chunk()
            else:
                works_all.extend(works)
        event = None
        if side is not None:
            event = torch.cuda.Event()
            event.record(side)
        return PendingClosest(outs, event, works_all, keep=(packed_all, mine, o, d))

    def intersects_closest_async(self, origins, directions, dst: Optional[int] = 0,
                                 chunks: Optional[int] = None) -> PendingClosest:
        """intersects_closest (stream_compaction=False) of a batch that is visible on every rank, as an
        in-flight handle; see closest_of_shard_async."""
        b, n, lo, hi, o, d = self._my_rays(origins, directions)
        if not self._can_pack() or (self.world == 1 and not self.force_collectives):
            return PendingClosest(self.intersects_closest(origins, directions, dst=dst))
        return self.closest_of_shard_async(o, d, n, batch_shape=b, dst=dst, chunks=chunks)

    # ---- queries (same names / return orders as RayMeshIntersector) -------------------------
    def intersects_any(self, origins, directions, dst: Optional[int] = 0):
        b, n, lo, hi, o, d = self._my_rays(origins, directions)
        out = self._gather_fixed(self.local.intersects_any(o, d).reshape(-1), n, dst)
        return None if out is None else out.reshape(b)

    def intersects_first(self, origins, directions, dst: Optional[int] = 0):
        b, n, lo, hi, o, d = self._my_rays(origins, directions)
        out = self._gather_fixed(self.local.intersects_first(o, d).reshape(-1), n, dst)
        return None if out is None else out.reshape(b)

    def intersects_count(self, origins, directions, dst: Optional[int] = 0):
        b, n, lo, hi, o, d = self._my_rays(origins, directions)
        out = self._gather_fixed(self.local.intersects_count(o, d).reshape(-1), n, dst)
        return None if out is None else out.reshape(b)

    def intersects_closest(self, origins, directions, stream_compaction: bool = False,
                           dst: Optional[int] = 0, chunks: Optional[int] = None):
        b, n, lo, hi, o, d = self._my_rays(origins, directions)
        m = hi - lo
        if not stream_compaction:
            if (self.world > 1 or self.force_collectives) and self._can_pack():
                return self.closest_of_shard_async(o, d, n, batch_shape=b, dst=dst, chunks=chunks).wait()
            res = self.local.intersects_closest(o, d)
            outs = [self._gather_fixed(x.reshape(m, *x.shape[o.dim() - 1:]), n, dst) for x in res]
            if outs[0] is None:
                return None
            hit, front, tri, loc, uv = outs
            return hit.reshape(b), front.reshape(b), tri.reshape(b), loc.reshape(*b, 3), uv.reshape(*b, 2)
        hit, front, ray_idx, tri, loc, uv = self.local.intersects_closest(o, d, stream_compaction=True)
        ray_idx = ray_idx + lo          # local -> global flat ray index
        hit_all = self._gather_fixed(hit.reshape(-1), n, dst)
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
