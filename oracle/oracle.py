"""ctypes/numpy front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  It loads ``oracle/libtriro_oracle.so`` (built from
``oracle/triro_oracle.c`` -- see that file's header for the parity status:
*parity unpinned* against the real OptiX path at the bit level; pinned at image precision on
the reference's published README rendering) and restates, in numpy, the
Python-level conventions of the reference's public class:

* return orders / dtypes / shapes      triro/ray/ray_optix.py:117-146,157-164,191-223
* stream compaction                    triro/ray/ray_optix.py:142-144
* clamp-to-8 + exclusive scan          triro/backend/ray.cpp:333-342
* contains_points parity logic         triro/ray/ray_optix.py:231-279
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
MAX_ANYHIT_SIZE = 8  # triro/backend/LaunchParams.h:8


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc)."""
    so = os.path.join(_HERE, "libtriro_oracle.so")
    src = os.path.join(_HERE, "triro_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtriro_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "libtriro_oracle.so")
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    fp, ip, u8p, i64p = (C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8),
                         C.POINTER(C.c_int64))
    L.oracle_mesh_create.restype = C.c_void_p
    L.oracle_mesh_create.argtypes = [fp, C.c_int64, ip, C.c_int64]
    L.oracle_mesh_destroy.argtypes = [C.c_void_p]
    L.oracle_closest.argtypes = [C.c_void_p, fp, fp, C.c_int64, C.c_int, C.c_int,
                                 u8p, u8p, ip, fp, fp, fp]
    L.oracle_count.argtypes = [C.c_void_p, fp, fp, C.c_int64, C.c_int, C.c_int, ip]
    L.oracle_location_fill.argtypes = [C.c_void_p, fp, fp, C.c_int64, C.c_int, C.c_int,
                                       C.c_int32, i64p, fp, ip, ip, fp]
    L.oracle_fetch_rays.argtypes = [fp, fp, i64p, i64p, i64p, C.c_int64, fp, fp]
    L.oracle_watertight.argtypes = [C.c_void_p, fp, fp, C.c_int64, C.c_int, ip, C.POINTER(C.c_double), ip,
                                    C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.oracle_anchor_rays.argtypes = [C.c_void_p, fp, fp, C.c_int64, fp]
    L.oracle_num_threads.restype = C.c_int
    _LIB = L
    return L


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


def fetch_rays(origins: np.ndarray, directions: np.ndarray):
    """Strided ray fetch exactly as the device code does it (shaders.cu:27-63).

    ``origins``/``directions`` are float32 numpy arrays/views of shape [*b,3] with
    arbitrary (also zero) strides; returns dense [n,3] copies obtained through the
    restated index arithmetic, not through numpy's own striding."""
    assert origins.dtype == np.float32 and directions.dtype == np.float32
    shape = list(origins.shape)
    assert len(shape) <= 4 and shape[-1] == 3
    n = int(np.prod(shape[:-1], dtype=np.int64))

    def pad(vals, fill):
        return np.array([fill] * (4 - len(vals)) + list(vals), dtype=np.int64)

    shp = pad(shape, np.iinfo(np.int64).max)
    ost = pad([s // 4 for s in origins.strides], 0)
    dst = pad([s // 4 for s in directions.strides], 0)
    o = np.empty((n, 3), np.float32)
    d = np.empty((n, 3), np.float32)
    # base pointers of the views (element 0)
    ob = C.cast(origins.ctypes.data, C.POINTER(C.c_float))
    db = C.cast(directions.ctypes.data, C.POINTER(C.c_float))
    lib().oracle_fetch_rays(ob, db, _p(shp, C.c_int64), _p(ost, C.c_int64),
                            _p(dst, C.c_int64), n, _p(o, C.c_float), _p(d, C.c_float))
    return o, d


class OracleIntersector:
    """numpy mirror of ``triro.ray.ray_optix.RayMeshIntersector`` (ray_optix.py:18-279).

    mode: 0 = brute force over all triangles (ground truth), 1 = oracle BVH (fast)."""

    def __init__(self, vertices, faces, mode: int = 1, threads: int = 0):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float32)
        self.faces = np.ascontiguousarray(faces, dtype=np.int32)
        self.mode = mode
        self.threads = threads
        self.mesh_aabb = (self.vertices.min(0), self.vertices.max(0))
        self._h = lib().oracle_mesh_create(_p(self.vertices, C.c_float), len(self.vertices),
                                           _p(self.faces, C.c_int32), len(self.faces))

    def __del__(self):
        try:
            if self._h:
                lib().oracle_mesh_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- helpers -----------------------------------------------------------
    def _rays(self, origins, directions):
        origins = np.asarray(origins)
        directions = np.asarray(directions)
        if origins.dtype != np.float32:
            origins = origins.astype(np.float32)
        if directions.dtype != np.float32:
            directions = directions.astype(np.float32)
        b = origins.shape[:-1]
        o, d = fetch_rays(origins, directions)
        return b, o, d

    # -- queries -----------------------------------------------------------
    def closest_raw(self, origins, directions):
        b, o, d = self._rays(origins, directions)
        n = len(o)
        hit = np.empty(n, np.uint8)
        front = np.empty(n, np.uint8)
        tri = np.empty(n, np.int32)
        loc = np.empty((n, 3), np.float32)
        uv = np.empty((n, 2), np.float32)
        t = np.empty(n, np.float32)
        lib().oracle_closest(self._h, _p(o, C.c_float), _p(d, C.c_float), n, self.mode,
                             self.threads, _p(hit, C.c_uint8), _p(front, C.c_uint8),
                             _p(tri, C.c_int32), _p(loc, C.c_float), _p(uv, C.c_float),
                             _p(t, C.c_float))
        return (hit.astype(bool).reshape(b), front.astype(bool).reshape(b), tri.reshape(b),
                loc.reshape(*b, 3), uv.reshape(*b, 2), t.reshape(b))

    def closest_timed(self, o, d, threads: int, passes: int = 1):
        """seconds of `passes` calls of the C entry point ALONE (oracle_closest on dense [n, 3] rays with
        `threads` OpenMP threads, dynamic schedule over rays) -- bench.py's cpu_baseline; nothing of numpy
        is inside the clock.  The output arrays are allocated and touched once, before the clock: a fresh
        np.empty per pass puts ~8 000 page faults per million rays into the timed call, and in a VM those
        (16 us each, serialised on the address-space lock) cost more than the tracing on 8 threads."""
        import time
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        # (np.full writes every page; np.zeros would hand out untouched copy-on-write zero pages)
        hit, front = np.full(n, 1, np.uint8), np.full(n, 1, np.uint8)
        tri, t = np.full(n, 1, np.int32), np.full(n, 1, np.float32)
        loc, uv = np.full((n, 3), 1, np.float32), np.full((n, 2), 1, np.float32)
        args = (self._h, _p(o, C.c_float), _p(d, C.c_float), n, self.mode, int(threads), _p(hit, C.c_uint8),
                _p(front, C.c_uint8), _p(tri, C.c_int32), _p(loc, C.c_float), _p(uv, C.c_float), _p(t, C.c_float))
        el = 0.0
        for _ in range(passes):
            t0 = time.perf_counter()
            lib().oracle_closest(*args)
            el += time.perf_counter() - t0
        return el

    def anchor(self, origins, directions):
        """the origins every query of the contract really traces: a ray that starts outside the mesh's box is moved along
        itself to just before its entry point (contract 3, "ray anchoring": triro_oracle.c anchor_ray = csrc/tr_math.h
        tr_ray_anchor); float32 [*b, 3].  For comparisons with references that take plain rays."""
        b, o, d = self._rays(origins, directions)
        out = np.empty_like(o)
        lib().oracle_anchor_rays(self._h, _p(o, C.c_float), _p(d, C.c_float), len(o), _p(out, C.c_float))
        return out.reshape(*b, 3)

    def watertight(self, origins, directions, with_bary=False):
        """(tri[*b] int32 (-1 miss), t[*b] float64, count[*b] int32) under the WATERTIGHT float64 test of
        Woop / Benthin / Wald 2013 in its published form (shear with a division, per-ray permutation) -- an
        implementation of its own beside the contract's float64 part: the yardstick for the contract's hit mask and,
        with_bary, for what the API returns: + (uvw[*b,3], loc[*b,3]) float64 barycentric weights / location of
        the hit (triro_oracle.c, "WATERTIGHT float64 reference")"""
        b, o, d = self._rays(origins, directions)
        n = len(o)
        tri, cnt = np.empty(n, np.int32), np.empty(n, np.int32)
        t = np.empty(n, np.float64)
        uvw = np.empty((n, 3), np.float64) if with_bary else None
        loc = np.empty((n, 3), np.float64) if with_bary else None
        lib().oracle_watertight(self._h, _p(o, C.c_float), _p(d, C.c_float), n, self.threads, _p(tri, C.c_int32),
                                _p(t, C.c_double), _p(cnt, C.c_int32), _p(uvw, C.c_double), _p(loc, C.c_double))
        if with_bary:
            return tri.reshape(b), t.reshape(b), cnt.reshape(b), uvw.reshape(*b, 3), loc.reshape(*b, 3)
        return tri.reshape(b), t.reshape(b), cnt.reshape(b)

    def intersects_closest(self, origins, directions, stream_compaction=False):
        hit, front, tri, loc, uv, _ = self.closest_raw(origins, directions)
        if stream_compaction:  # ray_optix.py:142-144
            ray_idx = np.arange(hit.size, dtype=np.int32)[hit.reshape(-1)]
            return hit, front[hit], ray_idx, tri[hit], loc[hit], uv[hit]
        return hit, front, tri, loc, uv

    def intersects_first(self, origins, directions):
        return self.closest_raw(origins, directions)[2]

    def intersects_count(self, origins, directions):
        b, o, d = self._rays(origins, directions)
        cnt = np.empty(len(o), np.int32)
        lib().oracle_count(self._h, _p(o, C.c_float), _p(d, C.c_float), len(o), self.mode,
                           self.threads, _p(cnt, C.c_int32))
        return cnt.reshape(b)

    def intersects_any(self, origins, directions):
        return self.intersects_count(origins, directions) > 0

    def intersects_location(self, origins, directions, with_t=False):
        b, o, d = self._rays(origins, directions)
        n = len(o)
        cnt = self.intersects_count(o, d).reshape(-1)
        clamped = np.minimum(cnt, MAX_ANYHIT_SIZE).astype(np.int64)  # ray.cpp:334-335
        incl = np.cumsum(clamped)                                    # ray.cpp:336
        nhits = int(incl[-1]) if n else 0                            # ray.cpp:339
        offsets = np.concatenate([[0], incl[:-1]]).astype(np.int64)  # ray.cpp:340-341
        loc = np.empty((nhits, 3), np.float32)
        ray = np.empty(nhits, np.int32)
        tri = np.empty(nhits, np.int32)
        t = np.empty(nhits, np.float32)
        lib().oracle_location_fill(self._h, _p(o, C.c_float), _p(d, C.c_float), n, self.mode,
                                   self.threads, MAX_ANYHIT_SIZE, _p(offsets, C.c_int64),
                                   _p(loc, C.c_float), _p(ray, C.c_int32),
                                   _p(tri, C.c_int32), _p(t, C.c_float))
        if with_t:
            return loc, ray, tri, t
        return loc, ray, tri  # ray.cpp:377

    def intersects_id(self, origins, directions, return_locations=False, multiple_hits=True):
        if multiple_hits:  # ray_optix.py:207-215
            loc, ray, tri = self.intersects_location(origins, directions)
            return (tri, ray, loc) if return_locations else (tri, ray)
        hit, _, tri, loc, _ = self.intersects_closest(origins, directions)
        ray = np.arange(hit.size, dtype=np.int32)[hit.reshape(-1)]
        return (tri[hit], ray, loc[hit]) if return_locations else (tri[hit], ray)

    def contains_points(self, points, check_direction=None, _retry_dirs=None):
        """ray_optix.py:231-279.  ``_retry_dirs``: iterator of replacement directions
        for the reference's ``torch.rand(3) - 0.5`` (:273) so tests are deterministic."""
        points = np.asarray(points, np.float32)
        contains = np.zeros(points.shape[:-1], bool)
        inside = ~((~(points > self.mesh_aabb[0])).any(1) | (~(points < self.mesh_aabb[1])).any(1))
        if not inside.any():
            return contains
        default = np.array([0.4395064455, 0.617598629942, 0.652231566745], np.float32)
        dirv = default if check_direction is None else np.asarray(check_direction, np.float32)
        dirs = np.tile(dirv, [*contains.shape, 1])
        hc = np.stack([self.intersects_count(points, dirs),
                       self.intersects_count(points, -dirs)], 0)
        mod2 = np.remainder(hc, 2)
        agree = np.all(mod2, 0)
        # `inside_aabb & agree & hit_count_mod_2[0] == 1` parses as ((a & b & c) == 1)
        contain = (inside & agree & mod2[0].astype(bool)) == 1
        broken = ~agree & (hc == 0).any(0)
        if not broken.any():
            return contain
        if check_direction is None:
            nd = next(_retry_dirs) if _retry_dirs is not None else (
                np.random.default_rng(0).random(3).astype(np.float32) - 0.5)
            contains = contain.copy()
            contains[broken] = self.contains_points(points[broken], nd, _retry_dirs)
        return contains


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def usable_cpus() -> int:
    """CPUs this process can really run on: the affinity mask cut by the cgroup's CPU quota (a
    container that shows 128 cores may be allowed 8: OpenMP's default would oversubscribe them)."""
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, math.ceil(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, math.ceil(q / per)))
            break
        except Exception:
            continue
    return max(1, n)
