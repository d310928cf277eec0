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

    def intersects_closest(self, origins, directions, stream_compaction=False):
        b = origins.shape[:-1]
        s = (origins.expand(*b, 3) * 3.0 + directions).sum(-1)
        hit = s > 0
        tri = (s.abs() * 1000).to(torch.int32) % max(self.nf, 1)
        loc = origins.expand(*b, 3) + directions
        uv = directions[..., :2].clone()
        return hit, ~hit, tri, loc.contiguous(), uv


def make(v, f, device):
    return StubTracer(v, f, device)
