"""ctypes front-end of tests/host_sim/libhost_sim.so (TEST-ONLY logic simulation of the
product's host+device headers; see host_sim.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libhost_sim.so")
        src = os.path.join(_HERE, "host_sim.cpp")
        hdr = os.path.join(_HERE, "..", "..", "trimesh-ray-optix_amd", "csrc")
        newest = max([os.path.getmtime(src)] + [os.path.getmtime(os.path.join(hdr, h))
                                                for h in ("tr_math.h", "tr_bvh.h", "tr_lbvh.h", "tr_wide.h")])
        if not os.path.exists(so) or os.path.getmtime(so) < newest:
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                                   "-mfma", "-Wno-unknown-pragmas", "-o", so, src])
        L = C.CDLL(so)
        L.sim_build.restype = C.c_void_p
        L.sim_build.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.sim_destroy.argtypes = [C.c_void_p]
        L.sim_depth.argtypes = [C.c_void_p]
        L.sim_key_mode.argtypes = [C.c_void_p]
        L.sim_num_nodes.argtypes = [C.c_void_p]
        L.sim_num_nodes.restype = C.c_int64
        L.sim_get.argtypes = [C.c_void_p] * 4
        L.sim_get_qnodes.argtypes = [C.c_void_p] * 3
        L.sim_set_qnodes.argtypes = [C.c_void_p] * 2
        L.sim_query.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_int64, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 7
        L.sim_location.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 3
        L.sim_use_ring.argtypes = [C.c_int]
        L.sim_use_fused.argtypes = [C.c_int]
        L.sim_use_unordered.argtypes = [C.c_int]
        L.sim_steps.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.sim_packet_stats.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.sim_check_fused_wide.restype = C.c_int64
        L.sim_check_fused_wide.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.sim_check_fused.restype = C.c_int64
        L.sim_check_fused.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.sim_tri_fast_vs_exact.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 5
        _LIB = L
    return _LIB


def tri_fast_vs_exact(o, d, tri):
    """per (ray, triangle) pair: (code of the float32 part, its t, hit of the float64 part, its t, long-double truth)"""
    o, d = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
    tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 9)
    n = len(o)
    code, he, truth = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32)
    tf, te = np.empty(n, np.float32), np.empty(n, np.float32)
    lib().sim_tri_fast_vs_exact(o.ctypes.data, d.ctypes.data, tri.ctypes.data, n, code.ctypes.data, tf.ctypes.data,
                                he.ctypes.data, te.ctypes.data, truth.ctypes.data)
    return code, tf, he, te, truth


NODE_WORDS, LINK_WORDS, TRI_WORDS = 16, 2, 12


class SimBVH:
    def __init__(self, verts=None, faces=None, force_mode=-1, morton_shift=0, arrays=None, qarrays=None):
        """arrays=(nodes, links, tris) [+ qarrays=(qnodes, frame)]: traverse arrays downloaded from the
        GPU builder"""
        if arrays is not None:
            self.nodes, self.links, self.tris = arrays
            self.nf = len(self.tris)
            self.depth = None
            self.qnodes, self.frame = qarrays if qarrays is not None else (np.zeros((len(self.nodes), 8), np.uint32), np.zeros(6, np.float32))
            return
        v = np.ascontiguousarray(verts, np.float32)
        f = np.ascontiguousarray(faces, np.int32)
        L = lib()
        h = L.sim_build(v.ctypes.data, len(v), f.ctypes.data, len(f), force_mode, morton_shift)
        self.nf = len(f)
        nn = L.sim_num_nodes(h)
        self.depth = L.sim_depth(h)
        self.key_mode = L.sim_key_mode(h)
        self.nodes = np.zeros((nn, NODE_WORDS), np.uint32)
        self.links = np.zeros((nn, LINK_WORDS), np.int32)
        self.tris = np.zeros((self.nf, TRI_WORDS), np.uint32)
        L.sim_get(h, self.nodes.ctypes.data, self.links.ctypes.data, self.tris.ctypes.data)
        self.qnodes = np.zeros((nn, 8), np.uint32)
        self.frame = np.zeros(6, np.float32)        # base[3], scale[3]
        L.sim_get_qnodes(h, self.qnodes.ctypes.data, self.frame.ctypes.data)
        L.sim_destroy(h)

    def qchild_boxes(self):
        """decoded grid boxes of both children of every 32-byte node: [N, 2, 6] = lo.xyz, hi.xyz (float32,
        bit-identical to the traversal's fma: q * scale is exact, one rounding of the sum)"""
        q = self.qnodes[:, :6].astype(np.uint32)
        base, scale = self.frame[:3], self.frame[3:]

        def dec(v, ax):
            return (v.astype(np.float64) * np.float64(scale[ax]) + np.float64(base[ax])).astype(np.float32)
        out = np.zeros((len(q), 2, 6), np.float32)
        for k in (0, 1):
            w = q[:, 3 * k:3 * k + 3]
            out[:, k, 0] = dec(w[:, 0] & 0xffff, 0); out[:, k, 1] = dec(w[:, 0] >> 16, 1)
            out[:, k, 2] = dec(w[:, 1] & 0xffff, 2); out[:, k, 5] = dec(w[:, 1] >> 16, 2)
            out[:, k, 3] = dec(w[:, 2] & 0xffff, 0); out[:, k, 4] = dec(w[:, 2] >> 16, 1)
        return out

    def query(self, q, o, d):
        lib().sim_set_qnodes(self.qnodes.ctypes.data, self.frame.ctypes.data)
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        hit = np.zeros(n, np.uint8); front = np.zeros(n, np.uint8); tri = np.zeros(n, np.int32)
        loc = np.zeros((n, 3), np.float32); uv = np.zeros((n, 2), np.float32)
        cnt = np.zeros(n, np.int32); stats = np.zeros(4, np.uint64)
        lib().sim_query(q, self.nodes.ctypes.data, self.links.ctypes.data, self.tris.ctypes.data, self.nf,
                        o.ctypes.data, d.ctypes.data, n, hit.ctypes.data, front.ctypes.data, tri.ctypes.data,
                        loc.ctypes.data, uv.ctypes.data, cnt.ctypes.data, stats.ctypes.data)
        return dict(hit=hit.astype(bool), front=front.astype(bool), tri=tri, loc=loc, uv=uv, count=cnt,
                    stats=stats)

    def steps(self, o, d):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        nv = np.zeros(n, np.int32); tt = np.zeros(n, np.int32)
        lib().sim_steps(self.nodes.ctypes.data, self.links.ctypes.data, self.tris.ctypes.data, self.nf,
                        o.ctypes.data, d.ctypes.data, n, nv.ctypes.data, tt.ctypes.data)
        return nv, tt

    def check_fused(self, o, d, node_stride=1):
        """the fused box test of the grid nodes against the contract's on the same decoded boxes (sim_check_fused):
        (violations, pairs checked, children accepted by the fused test, children accepted by the contract's)"""
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        out = np.zeros(3, np.int64)
        bad = lib().sim_check_fused(self.qnodes.ctypes.data, len(self.qnodes), self.frame.ctypes.data, o.ctypes.data,
                                    d.ctypes.data, len(o), node_stride, out.ctypes.data)
        return int(bad), int(out[0]), int(out[1]), int(out[2])

    def check_fused_wide(self, o, d, node_stride=1):
        """the same for the 8-wide nodes derived from this hierarchy (sim_check_fused_wide)"""
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        out = np.zeros(3, np.int64)
        bad = lib().sim_check_fused_wide(self.nodes.ctypes.data, len(self.nodes), self.frame.ctypes.data, o.ctypes.data,
                                         d.ctypes.data, len(o), node_stride, out.ctypes.data)
        return int(bad), int(out[0]), int(out[1]), int(out[2])

    def packet_stats(self, o, d, group=64):
        """per group of `group` consecutive rays: [lane-visits, slowest ray's visits, distinct nodes, leaf tests, busiest
        ray's leaf tests, distinct leaves] of the unordered (count) traversal on the exact nodes (sim_packet_stats)"""
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        out = np.zeros(((n + group - 1) // group, 6), np.int64)
        lib().sim_packet_stats(self.nodes.ctypes.data, self.nf, o.ctypes.data, d.ctypes.data, n, group, out.ctypes.data)
        return out

    def location(self, o, d, cap=8):
        lib().sim_set_qnodes(self.qnodes.ctypes.data, self.frame.ctypes.data)
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        cnt = np.zeros(n, np.int32); tri = np.zeros((n, cap), np.int32); t = np.zeros((n, cap), np.float32)
        lib().sim_location(self.nodes.ctypes.data, self.links.ctypes.data, self.tris.ctypes.data, self.nf,
                           o.ctypes.data, d.ctypes.data, n, cap, cnt.ctypes.data, tri.ctypes.data, t.ctypes.data)
        return cnt, tri, t


def use_ring(on: bool):
    lib().sim_use_ring(1 if on else 0)


def use_fused(mode: int):
    """0 = generic node/leaf schedule, 1 = fused trip, 2 = fused trip with 32-bit state/offsets; +4 = over the 32-byte grid nodes (no intervals in the leaf FIFO)"""
    lib().sim_use_fused(mode)


def use_unordered(on: bool):
    """any / count / location through the unordered two-phase schedule (queued leaves)"""
    lib().sim_use_unordered(1 if on else 0)
