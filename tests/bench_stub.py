"""Stand-in tracer for the CPU self-test of bench.py's launcher / sharding path
(`bench.py --backend gloo --stub bench_stub:make`).  NOT a ray tracer and never measured: it
returns cheap deterministic functions of the rays with the output shapes and dtypes of
RayMeshIntersector.intersects_closest, so that N gloo ranks can exercise shard_bounds, the
receive-into-place gather, the max-over-ranks timing and the JSON relay without a GPU."""
import torch


class StubTracer:
    def __init__(self, v, f, device):
        self.nf = int(len(f))
        self.device = device

    def bvh_info(self):
        return {"depth": 0, "node_bytes": 64 * max(self.nf - 1, 0), "tri_bytes": 48 * self.nf}

    # 12-byte records {tri | front << 30 (-1: miss), s bits, 0} and their expansion: the shape of the
    # packed pipeline of triro.ray.sharded; intersects_closest is expand(packed), so both paths agree
    packed_slots = True      # (the stand-in's "slot" is its triangle number)
    slot_records = True

    def intersects_closest_packed(self, origins, directions, out=None, slots=False):
        b = origins.shape[:-1]
        s = (origins.expand(*b, 3) * 3.0 + directions).sum(-1).reshape(-1).to(torch.float32)
        hit = s > 0
        tri = (s.abs() * 1000).to(torch.int32) % max(self.nf, 1)
        rec = torch.zeros((s.numel(), 3), dtype=torch.int32)
        rec[:, 0] = torch.where(hit, tri | ((tri & 1) << 30), torch.full_like(tri, -1))
        rec[:, 1] = s.view(torch.int32)
        if out is None:
            return rec
        out.copy_(rec)
        return out

    # 4-byte records: the triangle number alone; whoever holds the rays finishes the query (and here CHECKS that the
    # rays it was handed are the ones the slots belong to)
    def intersects_closest_slots(self, origins, directions, out=None):
        rec = self.intersects_closest_packed(origins, directions)
        sl = torch.where(rec[:, 0] >= 0, rec[:, 0] & 0x3fffffff, torch.full_like(rec[:, 0], -1))
        if out is None:
            return sl
        out.copy_(sl)
        return out

    def closest_from_slots(self, origins, directions, slots, outs=None, row_length=0):
        rec = self.intersects_closest_packed(origins, directions)
        mine = torch.where(rec[:, 0] >= 0, rec[:, 0] & 0x3fffffff, torch.full_like(rec[:, 0], -1))
        assert torch.equal(mine, slots.reshape(-1)), "record rows paired with the wrong rays"
        return self.closest_expand(rec, origins.shape[:-1], outs)

    def intersects_closest_into(self, origins, directions, outs):
        res = self.closest_expand(self.intersects_closest_packed(origins, directions))
        for dst_, src_ in zip(outs, res):
            dst_.copy_(src_.reshape(dst_.shape))
        return outs

    def closest_expand(self, packed, batch_shape=None, outs=None, slots=False, row_length=0):
        hit = packed[:, 0] >= 0
        tri = torch.where(hit, packed[:, 0] & 0x3fffffff, torch.full_like(packed[:, 0], -1))
        front = hit & (((packed[:, 0] >> 30) & 1) == 1)
        s = torch.where(hit, packed[:, 1].contiguous().view(torch.float32), torch.zeros(len(hit)))
        loc = torch.stack([s, 2 * s, 3 * s], -1)
        uv = torch.stack([s, -s], -1)
        res = (hit, front, tri, loc, uv)
        if outs is not None:
            for dst_, src_ in zip(outs, res):
                dst_.copy_(src_)
            return outs
        b = tuple(batch_shape) if batch_shape is not None else (len(hit),)
        return hit.reshape(b), front.reshape(b), tri.reshape(b), loc.reshape(*b, 3), uv.reshape(*b, 2)

    def intersects_closest(self, origins, directions, stream_compaction=False):
        return self.closest_expand(self.intersects_closest_packed(origins, directions), origins.shape[:-1])


def make(v, f, device):
    return StubTracer(v, f, device)
