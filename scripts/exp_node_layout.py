#!/usr/bin/env python3
"""Experiment for VERDICT r02 item 3(i): does the ORDER of the nodes in memory matter?

The builder emits nodes in Karras numbering (internal node i covers a key range that starts or ends at
leaf i: sibling pairs are adjacent, small subtrees contiguous).  Here the arena of the headline mesh is
downloaded as a blob (tr_bvh_serialize), the node arrays (64-byte nodes, links, 32-byte grid nodes) are
permuted on the host -- topology unchanged, ids rewritten --, uploaded again (tr_bvh_deserialize) and the
BASELINE queries are timed on each layout:

  karras     as built
  dfs        depth-first preorder (c0 right behind its parent)
  bfs        level order
  bfs12+dfs  the top 12 levels in level order, depth-first below
  treelet3   treelets of 3 levels (7 nodes = 224 B of grid nodes), depth-first over treelets, level order inside
  random     a random permutation (how much the layout can matter at all)

Results must not change (checked).  usage (GPU box): python scripts/exp_node_layout.py > gpurun_out/node_layout.jsonl"""
import json
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "trimesh-ray-optix_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import workloads as W  # noqa: E402
from triro.ray.ray_optix import OptixAccelStructureWrapper, RayMeshIntersector  # noqa: E402

dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
HDR = 80


def align(x, a=256):
    return (x + a - 1) // a * a


def split_blob(blob):
    num_tris, num_nodes, arena_bytes = struct.unpack_from("<qqq", blob, 8)
    nn, nf = num_nodes, num_tris
    o_nodes = 0
    o_links = align(o_nodes + 64 * nn)
    o_tris = align(o_links + 8 * nn)
    o_q = align(o_tris + 48 * nf)
    assert align(o_q + 32 * nn) == arena_bytes, (align(o_q + 32 * nn), arena_bytes)
    a = blob[HDR:HDR + arena_bytes]
    nodes = a[o_nodes:o_nodes + 64 * nn].view(np.int32).reshape(nn, 16).copy()
    links = a[o_links:o_links + 8 * nn].view(np.int32).reshape(nn, 2).copy()
    q = a[o_q:o_q + 32 * nn].view(np.int32).reshape(nn, 8).copy()
    return (nn, nf, (o_nodes, o_links, o_tris, o_q)), nodes, links, q


def orders(c0, c1, nn):
    """new index of every old node, per layout"""
    out = {"karras": np.arange(nn)}
    depth = np.zeros(nn, np.int32)
    dfs = np.zeros(nn, np.int64)
    stack, k = [0], 0
    while stack:                              # preorder; c0 first
        n = stack.pop()
        dfs[n] = k
        k += 1
        a, b = c0[n], c1[n]
        if b >= 0:
            depth[b] = depth[n] + 1
            stack.append(b)
        if a >= 0:
            depth[a] = depth[n] + 1
            stack.append(a)
    out["dfs"] = dfs
    out["bfs"] = np.empty(nn, np.int64)
    out["bfs"][np.lexsort((dfs, depth))] = np.arange(nn)
    top = depth < 12
    key = np.where(top, depth.astype(np.int64), 1 << 40)
    o = np.empty(nn, np.int64)
    o[np.lexsort((dfs, key))] = np.arange(nn)
    out["bfs12+dfs"] = o
    # treelets of 3 levels: treelet roots at depth % 3 == 0; depth-first over treelets, level order inside
    tl = np.zeros(nn, np.int64)
    k = 0
    stack = [0]
    while stack:
        r = stack.pop()
        level = [r]
        nxt_roots = []
        for lv in range(3):
            nl = []
            for n in level:
                tl[n] = k
                k += 1
                for c in (c0[n], c1[n]):
                    if c >= 0:
                        (nl if lv < 2 else nxt_roots).append(c)
            level = nl
        stack.extend(reversed(nxt_roots))
    out["treelet3"] = tl
    rnd = np.random.default_rng(0).permutation(nn)
    j = int(np.nonzero(rnd == 0)[0][0])
    rnd[j], rnd[0] = rnd[0], 0                 # the root stays node 0
    out["random"] = rnd
    return out


def permute(blob, meta, nodes, links, q, new_of_old):
    nn, nf, (o_nodes, o_links, o_tris, o_q) = meta
    assert new_of_old[0] == 0 and len(np.unique(new_of_old)) == nn

    def remap(ids):                           # internal ids -> new ids; leaves (< 0) and -1 stay
        return np.where(ids >= 0, new_of_old[np.maximum(ids, 0)], ids).astype(np.int32)
    n2, l2, q2 = np.empty_like(nodes), np.empty_like(links), np.empty_like(q)
    src = nodes.copy()
    src[:, 12] = remap(nodes[:, 12]); src[:, 13] = remap(nodes[:, 13])
    src[:, 14] = remap(nodes[:, 14]); src[:, 15] = remap(nodes[:, 15])
    src[0, 15] = 0
    n2[new_of_old] = src
    ls = links.copy()
    ls[:, 0] = remap(links[:, 0]); ls[:, 1] = remap(links[:, 1]); ls[0, 1] = 0
    l2[new_of_old] = ls
    qs = q.copy()
    qs[:, 6] = remap(q[:, 6]); qs[:, 7] = remap(q[:, 7])
    q2[new_of_old] = qs
    out = blob.copy()
    a = out[HDR:]
    a[o_nodes:o_nodes + 64 * nn] = n2.view(np.uint8).reshape(-1)
    a[o_links:o_links + 8 * nn] = l2.view(np.uint8).reshape(-1)
    a[o_q:o_q + 32 * nn] = q2.view(np.uint8).reshape(-1)
    return out


def timed(fn, reps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


v, f = W.headline_mesh(8)
base = RayMeshIntersector(vertices=T(v), faces=T(f))
blob = base.as_wrapper.serialize()
meta, nodes, links, q = split_blob(blob)
t0 = time.time()
layouts = orders(nodes[:, 12], nodes[:, 13], meta[0])
print(json.dumps({"note": f"layouts computed in {time.time() - t0:.1f} s on the host", "nodes": int(meta[0])}), flush=True)
rad = float(np.linalg.norm(v, axis=1).max())
o_img, d_img = [T(x) for x in W.pinhole_grid(1024, 1024, distance=2.5 * rad)]
n5 = 12_500_000
o5, d5 = W.hash_rays_torch(n5, 99, v.min(0) * 1.5, v.max(0) * 1.5, device=dev)
ref_img = ref_sh = None
for name, new_of_old in layouts.items():
    r = RayMeshIntersector.__new__(RayMeshIntersector)
    r.mesh_vertices, r.mesh_faces, r._mesh_aabb = base.mesh_vertices, base.mesh_faces, None
    r.as_wrapper = OptixAccelStructureWrapper()
    r.as_wrapper.deserialize(permute(blob, meta, nodes, links, q, new_of_old), dev)
    ms_img, out_img = timed(lambda: r.intersects_closest(o_img, d_img), 200, 60)
    ms_cnt, _ = timed(lambda: r.intersects_count(o_img, d_img), 40, 16)
    ms_sh, out_sh = timed(lambda: r.intersects_closest(o5, d5), 10, 4)
    ms_any, _ = timed(lambda: r.intersects_any(o5, d5), 10, 4)
    if ref_img is None:
        ref_img, ref_sh = [x.clone() for x in out_img], [x.clone() for x in out_sh]
    same = all(torch.equal(a, b) for a, b in zip(out_img, ref_img)) and all(torch.equal(a, b) for a, b in zip(out_sh, ref_sh))
    print(json.dumps({"layout": name, "c5i_closest_ms": round(ms_img, 4), "c5i_count_ms": round(ms_cnt, 4),
                      "c5s_closest_ms": round(ms_sh, 4), "c5s_any_ms": round(ms_any, 4), "results_identical": bool(same)}), flush=True)
    del r
