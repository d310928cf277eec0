"""Python op shims over the C ABI of libtriro_hip.so (include/triro_hip.h).

Mirrors triro/backend/ops.py of the reference function for function:

  get_module()                                   ops.py:12-46   (JIT build + import of the
                                                 pybind11 module) -> ctypes load of the
                                                 PREBUILT library; no JIT, no pybind11
  init_optix ... build_sbts                      ops.py:49-81   -> tr_init (idempotent)
  intersects_any/first/closest/count/location    ops.py:84-192  -> tr_intersects_*

Differences that are deliberate: inputs are validated and errors raise
(ValueError / RuntimeError) instead of printing and returning undefined tensors
(ray.cpp:104-123,164-165); float32 is enforced (the reference silently reinterprets other
dtypes); work is enqueued on torch's CURRENT stream of the tensors' device instead of a
private stream (base.cpp:34); outputs are allocated here with torch.empty and handed to the
library as raw pointers.

There is no CPU fallback: without the library or without a GPU every query raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Tuple

import torch

_INT64_MAX = (1 << 63) - 1
MAX_ANYHIT_SIZE = 8   # LaunchParams.h:8
MAX_SIZE_LENGTH = 4   # LaunchParams.h:9

_LIB_NAME = "libtriro_hip.so"
ABI_VERSION = 10    # TR_ABI_VERSION of include/triro_hip.h this binding was written against
_lib = None
_lib_error = None


class TrRays(C.Structure):
    """tr_rays (include/triro_hip.h) == RayInput of LaunchParams.h:11-28."""
    _fields_ = [("d_origins", C.c_void_p), ("d_directions", C.c_void_p), ("nray", C.c_int64),
                ("shape", C.c_int64 * 4), ("ostride", C.c_int64 * 4), ("dstride", C.c_int64 * 4)]


class TrTopology(C.Structure):
    """tr_topology (include/triro_hip.h)"""
    _fields_ = [("num_cus", C.c_int32), ("num_xcd", C.c_int32), ("waves_per_cu", C.c_int32), ("reserved", C.c_int32),
                ("l2_bytes", C.c_int64), ("resident_lanes", C.c_int64), ("steal_max_rays", C.c_int64),
                ("wide_min_rays", C.c_int64), ("count_stream_min_rays", C.c_int64)]


class TrBvhInfo(C.Structure):
    _fields_ = [("device", C.c_int32), ("num_tris", C.c_int64), ("num_nodes", C.c_int64),
                ("depth", C.c_int32), ("key_mode", C.c_int32), ("arena_bytes", C.c_int64),
                ("node_bytes", C.c_int64), ("tri_bytes", C.c_int64),
                ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3)]


class TrLaunchInfo(C.Structure):
    _fields_ = [("rays", C.c_int64), ("blocks", C.c_int64), ("slots", C.c_int64), ("query", C.c_int32),
                ("shape", C.c_int32), ("tile_rows_lg", C.c_int32), ("split_blocks", C.c_int32),
                ("learned_order", C.c_int32), ("grid_nodes", C.c_int32), ("addressing", C.c_int32),
                ("sort_carried", C.c_int32)]


class TrTraceStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("climb_steps", C.c_uint64)]


def library_path() -> str:
    env = os.environ.get("TRIRO_HIP_LIBRARY")
    if env:
        return env
    return os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                        "lib", _LIB_NAME)


# every symbol include/triro_hip.h declares: (restype, argtypes)
_vp, _i64, _i32, _int = C.c_void_p, C.c_int64, C.c_int32, C.c_int
ABI = {
    "tr_init": (_int, [_int]),
    "tr_abi_version": (_int, []),
    "tr_last_error": (C.c_char_p, []),
    "tr_bvh_build": (_int, [_vp, _i64, _vp, _i64, _vp, C.POINTER(_vp)]),
    "tr_bvh_update": (_int, [_vp, _vp, _i64, _vp, _i64, _vp]),
    "tr_bvh_refit": (_int, [_vp, _vp, _i64, _vp, _i64, _vp]),
    "tr_bvh_destroy": (_int, [_vp]),
    "tr_bvh_serialized_size": (_i64, [_vp]),
    "tr_bvh_serialize": (_int, [_vp, _vp, _i64, _vp]),
    "tr_bvh_deserialize": (_int, [_vp, _i64, _vp, C.POINTER(_vp)]),
    "tr_bvh_get_info": (_int, [_vp, C.POINTER(TrBvhInfo)]),
    "tr_bvh_last_launch": (_int, [_vp, C.POINTER(TrLaunchInfo)]),
    "tr_device_topology": (_int, [_int, C.POINTER(TrTopology)]),
    "tr_bvh_replica_hash": (_int, [_vp, C.POINTER(C.c_uint64), _vp]),
    "tr_bvh_download": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "tr_bvh_download_qnodes": (_int, [_vp, _vp, _vp, _vp]),
    "tr_intersects_any": (_int, [_vp, C.POINTER(TrRays), _vp, _vp]),
    "tr_intersects_first": (_int, [_vp, C.POINTER(TrRays), _vp, _vp]),
    "tr_intersects_closest": (_int, [_vp, C.POINTER(TrRays), _vp, _vp, _vp, _vp, _vp, _vp]),
    "tr_intersects_count": (_int, [_vp, C.POINTER(TrRays), _vp, _vp]),
    "tr_intersects_closest_packed": (_int, [_vp, C.POINTER(TrRays), _vp, _vp]),
    "tr_closest_expand": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tr_intersects_closest_packed_slots": (_int, [_vp, C.POINTER(TrRays), _vp, _vp]),
    "tr_closest_expand_slots": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tr_closest_expand_slots_rows": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tr_intersects_closest_slots": (_int, [_vp, C.POINTER(TrRays), _vp, _vp]),
    "tr_closest_from_slots": (_int, [_vp, C.POINTER(TrRays), _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tr_hits_scan": (_int, [_vp, _i64, _i32, _vp, _vp, C.POINTER(_i64), _vp]),
    "tr_intersects_location_fill": (_int, [_vp, C.POINTER(TrRays), _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "tr_intersects_count_topk": (_int, [_vp, C.POINTER(TrRays), _i32, _vp, _vp, _vp]),
    "tr_location_fill_slots": (_int, [_vp, C.POINTER(TrRays), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "tr_mask_scan": (_int, [_vp, _i64, _vp, _vp, C.POINTER(_i64), _vp]),
    "tr_compact_closest": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tr_trace_stats_closest": (_int, [_vp, C.POINTER(TrRays), C.POINTER(TrTraceStats), _vp]),
    "tr_trace_stats_query": (_int, [_vp, C.POINTER(TrRays), _int, C.POINTER(TrTraceStats), _vp]),
    "tr_set_option": (_int, [C.c_char_p, _i64]),
}


def get_module():
    """Load libtriro_hip.so (the counterpart of ops.py:12-46's JIT build + import)."""
    global _lib, _lib_error
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        _lib_error = (f"{path} not found: build it with `make -C trimesh-ray-optix_amd/csrc` "
                      f"(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                      f"There is no CPU fallback.")
        raise RuntimeError(_lib_error)
    lib = C.CDLL(path)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)   # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.tr_abi_version() != ABI_VERSION and os.environ.get("TRIRO_ABI_ANY") != "1":      # (TRIRO_ABI_ANY=1: A/B runs against an older build)
        raise RuntimeError(f"libtriro_hip.so ABI version {lib.tr_abi_version()} != {ABI_VERSION} expected by this binding")
    _lib = lib
    return _lib


def _check(status: int):
    if status != 0:
        msg = get_module().tr_last_error().decode("utf-8", "replace")
        if status == 1:
            raise ValueError(f"triro_hip: {msg}")
        raise RuntimeError(f"triro_hip (status {status}): {msg}")


# --- the five bring-up shims of ops.py:49-81 --------------------------------------------------
def init_optix():
    """ops.py:49-53.  Loads the library; initialises the current GPU if there is one."""
    try:
        lib = get_module()
    except RuntimeError:
        return   # import must not fail without the library; queries will
    if torch.cuda.is_available():
        _check(lib.tr_init(torch.cuda.current_device()))


def create_optix_context():
    """ops.py:56-60 -- nothing left to do (no OptiX context on gfx950)."""


def create_optix_module():
    """ops.py:63-67 -- kernels are linked into libtriro_hip.so."""


def create_optix_pipelines():
    """ops.py:70-74 -- the five pipelines are five kernel instantiations."""


def build_sbts():
    """ops.py:77-81 -- no shader binding tables."""


# --- marshaling -------------------------------------------------------------------------------
def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _fill4(vals, default):
    """fillArray of ray.cpp:151-159: right-align into MAX_SIZE_LENGTH slots."""
    vals = list(vals)
    return (C.c_int64 * 4)(*([default] * (MAX_SIZE_LENGTH - len(vals)) + vals))


def check_rays(origins: torch.Tensor, dirs: torch.Tensor):
    """tensorInputCheck (ray.cpp:104-123) plus the checks the reference forgets."""
    for name, t in (("origins", origins), ("directions", dirs)):
        if not isinstance(t, torch.Tensor):
            raise ValueError(f"{name} must be a torch.Tensor")
        if not t.is_cuda:
            raise ValueError(f"{name} must reside on a GPU (cuda/HIP) device")
        if t.layout != torch.strided:
            raise ValueError(f"{name} layout must be torch.strided")
        if t.dtype != torch.float32:
            raise ValueError(f"{name} must be float32, got {t.dtype}")
        if t.dim() < 1 or t.dim() > MAX_SIZE_LENGTH or t.shape[-1] != 3:
            raise ValueError(f"{name} must have shape [*b, 3] with at most 3 batch dims, got {tuple(t.shape)}")
    if origins.shape != dirs.shape:
        raise ValueError(f"origins {tuple(origins.shape)} and directions {tuple(dirs.shape)} differ in shape")
    if origins.device != dirs.device:
        raise ValueError("origins and directions are on different devices")


def make_rays(origins: torch.Tensor, dirs: torch.Tensor) -> TrRays:
    """LaunchParams marshaling of ray.cpp:173-179."""
    r = TrRays()
    r.d_origins = origins.data_ptr()
    r.d_directions = dirs.data_ptr()
    r.nray = origins.numel() // 3
    r.shape = _fill4(origins.shape, _INT64_MAX)
    r.ostride = _fill4(origins.stride(), 0)
    r.dstride = _fill4(dirs.stride(), 0)
    return r


def _handle(accel_structure, rays_tensor=None):
    h = accel_structure._inner
    if not h:
        raise RuntimeError("acceleration structure has not been built")
    if rays_tensor is not None:
        # the kernels dereference both the rays and the BVH arena: they must share a GPU
        bdev = getattr(accel_structure, "device_index", None)
        if bdev is not None and rays_tensor.device.index != bdev:
            raise ValueError(f"rays are on {rays_tensor.device} but the acceleration structure lives on "
                             f"cuda:{bdev}; move the rays or build the intersector on that device")
    return h


# --- queries (ops.py:84-192) ----------------------------------------------------------------------
def intersects_any(accel_structure, origins, dirs) -> torch.Tensor:
    """ops.py:84-100 / ray.cpp:161-189.  Bool[*b]."""
    check_rays(origins, dirs)
    out = torch.empty(origins.shape[:-1], dtype=torch.bool, device=origins.device)
    with torch.cuda.device(origins.device):
        _check(get_module().tr_intersects_any(_handle(accel_structure, origins), C.byref(make_rays(origins, dirs)),
                                              out.data_ptr(), _stream_ptr(origins.device)))
    return out


def intersects_first(accel_structure, origins, dirs) -> torch.Tensor:
    """ops.py:103-119 / ray.cpp:191-219.  Int32[*b], -1 on miss."""
    check_rays(origins, dirs)
    out = torch.empty(origins.shape[:-1], dtype=torch.int32, device=origins.device)
    with torch.cuda.device(origins.device):
        _check(get_module().tr_intersects_first(_handle(accel_structure, origins), C.byref(make_rays(origins, dirs)),
                                                out.data_ptr(), _stream_ptr(origins.device)))
    return out


def intersects_closest(accel_structure, origins, dirs, outs=None) -> Tuple[torch.Tensor, ...]:
    """ops.py:122-149 / ray.cpp:231-289.  (hit, front, tri_idx, loc, uv).  `outs`: optional preallocated
    contiguous destinations (bool [n], bool [n], int32 [n], float32 [n, 3], float32 [n, 2] with n = the
    number of rays -- e.g. this rank's rows of gathered full-size outputs, triro.ray.sharded)."""
    check_rays(origins, dirs)
    b, dev = origins.shape[:-1], origins.device
    if outs is not None:
        n = origins.numel() // 3
        want = ((torch.bool, (n,)), (torch.bool, (n,)), (torch.int32, (n,)), (torch.float32, (n, 3)), (torch.float32, (n, 2)))
        if len(outs) != 5 or any(t.dtype != dt or tuple(t.shape) != sh or not t.is_contiguous() or t.device != dev
                                 for t, (dt, sh) in zip(outs, want)):
            raise ValueError("outs must be contiguous (bool[n], bool[n], int32[n], float32[n,3], float32[n,2]) on the rays' device")
        hit, front, tri, loc, uv = outs
    else:
        hit = torch.empty(b, dtype=torch.bool, device=dev)
        front = torch.empty(b, dtype=torch.bool, device=dev)
        tri = torch.empty(b, dtype=torch.int32, device=dev)
        loc = torch.empty((*b, 3), dtype=torch.float32, device=dev)
        uv = torch.empty((*b, 2), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _check(get_module().tr_intersects_closest(
            _handle(accel_structure, origins), C.byref(make_rays(origins, dirs)), hit.data_ptr(), front.data_ptr(),
            tri.data_ptr(), loc.data_ptr(), uv.data_ptr(), _stream_ptr(dev)))
    return hit, front, tri, loc, uv


def intersects_closest_packed(accel_structure, origins, dirs, out: torch.Tensor = None, slots: bool = False) -> torch.Tensor:
    """Closest hit as 12 bytes per ray: int32 [n, 3] rows {face | front << 30 (-1 on a miss), u bits,
    v bits} (tr_packed_hit).  What a ray-sharded run sends over xGMI instead of the 26 B/ray of the five
    dense outputs; `closest_expand` rebuilds those bit for bit.  `out`: optional preallocated [n, 3]
    int32 destination (a slice of a gather buffer)."""
    check_rays(origins, dirs)
    n, dev = origins.numel() // 3, origins.device
    if out is None:
        out = torch.empty((n, 3), dtype=torch.int32, device=dev)
    elif out.dtype != torch.int32 or tuple(out.shape) != (n, 3) or not out.is_contiguous() or out.device != dev:
        raise ValueError("out must be a contiguous int32 [n, 3] tensor on the rays' device")
    # slots: the record names the ARENA SLOT of the triangle instead of its face index (tr_intersects_closest_packed_slots):
    # cheaper to expand (closest_expand_slots), valid only for this hierarchy or a bit-identical replica of it
    fn = get_module().tr_intersects_closest_packed_slots if slots else get_module().tr_intersects_closest_packed
    with torch.cuda.device(dev):
        _check(fn(_handle(accel_structure, origins), C.byref(make_rays(origins, dirs)), out.data_ptr(), _stream_ptr(dev)))
    return out


def intersects_closest_slots(accel_structure, origins, dirs, out: torch.Tensor = None) -> torch.Tensor:
    """tr_intersects_closest_slots: closest hit as 4 bytes per ray -- int32 [n]: the arena slot of the nearest
    triangle, -1 for a miss.  For a destination that holds the rays (closest_from_slots finishes the query there);
    valid for this hierarchy or a bit-identical replica.  `out`: optional preallocated int32 [n] destination."""
    check_rays(origins, dirs)
    n, dev = origins.numel() // 3, origins.device
    if out is None:
        out = torch.empty((n,), dtype=torch.int32, device=dev)
    elif out.dtype != torch.int32 or tuple(out.shape) != (n,) or not out.is_contiguous() or out.device != dev:
        raise ValueError("out must be a contiguous int32 [n] tensor on the rays' device")
    with torch.cuda.device(dev):
        _check(get_module().tr_intersects_closest_slots(_handle(accel_structure, origins), C.byref(make_rays(origins, dirs)),
                                                        out.data_ptr(), _stream_ptr(dev)))
    return out


def closest_from_slots(accel_structure, origins, dirs, slots: torch.Tensor, outs=None, row_length: int = 0):
    """tr_closest_from_slots: (rays, their slots from intersects_closest_slots on any bit-identical replica) ->
    (hit, front, tri_idx, loc, uv), the bits of intersects_closest on the same rays.  `outs`: optional preallocated
    contiguous destinations (bool [n], bool [n], int32 [n], float32 [n, 3], float32 [n, 2]); otherwise the outputs
    take the rays' batch shape.  row_length: the rays are whole rows of an image of that width."""
    check_rays(origins, dirs)
    n, dev = origins.numel() // 3, origins.device
    if slots.dtype != torch.int32 or tuple(slots.shape) != (n,) or not slots.is_contiguous() or slots.device != dev:
        raise ValueError("slots must be a contiguous int32 [n] tensor on the rays' device, one per ray")
    if outs is not None:
        want = ((torch.bool, (n,)), (torch.bool, (n,)), (torch.int32, (n,)), (torch.float32, (n, 3)), (torch.float32, (n, 2)))
        if len(outs) != 5 or any(t.dtype != dt or tuple(t.shape) != sh or not t.is_contiguous() or t.device != dev
                                 for t, (dt, sh) in zip(outs, want)):
            raise ValueError("outs must be contiguous (bool[n], bool[n], int32[n], float32[n,3], float32[n,2]) on the rays' device")
        hit, front, tri, loc, uv = outs
    else:
        b = origins.shape[:-1]
        hit = torch.empty(b, dtype=torch.bool, device=dev)
        front = torch.empty(b, dtype=torch.bool, device=dev)
        tri = torch.empty(b, dtype=torch.int32, device=dev)
        loc = torch.empty((*b, 3), dtype=torch.float32, device=dev)
        uv = torch.empty((*b, 2), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _check(get_module().tr_closest_from_slots(_handle(accel_structure, origins), C.byref(make_rays(origins, dirs)),
                                                  slots.data_ptr(), int(row_length), hit.data_ptr(), front.data_ptr(),
                                                  tri.data_ptr(), loc.data_ptr(), uv.data_ptr(), _stream_ptr(dev)))
    return hit, front, tri, loc, uv


def closest_expand_slots(accel_structure, packed: torch.Tensor, batch_shape=None, outs=None, row_length: int = 0):
    """tr_closest_expand_slots: slot-form records (intersects_closest_packed(..., slots=True), of this hierarchy or of a
    bit-identical replica) -> (hit, front, tri_idx, loc, uv), bit-identical to intersects_closest on the same rays."""
    if packed.dtype != torch.int32 or packed.dim() != 2 or packed.shape[1] != 3 or not packed.is_contiguous() or not packed.is_cuda:
        raise ValueError("packed must be a contiguous int32 [n, 3] tensor on the GPU")
    dev, n = packed.device, packed.shape[0]
    if outs is not None:
        want = ((torch.bool, (n,)), (torch.bool, (n,)), (torch.int32, (n,)), (torch.float32, (n, 3)), (torch.float32, (n, 2)))
        if len(outs) != 5 or any(t.dtype != dt or tuple(t.shape) != sh or not t.is_contiguous() or t.device != dev
                                 for t, (dt, sh) in zip(outs, want)):
            raise ValueError("outs must be contiguous (bool[n], bool[n], int32[n], float32[n,3], float32[n,2]) on the device of packed")
        hit, front, tri, loc, uv = outs
    else:
        b = tuple(batch_shape) if batch_shape is not None else (n,)
        hit = torch.empty(b, dtype=torch.bool, device=dev)
        front = torch.empty(b, dtype=torch.bool, device=dev)
        tri = torch.empty(b, dtype=torch.int32, device=dev)
        loc = torch.empty((*b, 3), dtype=torch.float32, device=dev)
        uv = torch.empty((*b, 2), dtype=torch.float32, device=dev)
        if hit.numel() != n:
            raise ValueError(f"batch_shape {b} does not hold {n} rays")
    # row_length: the records are whole rows of an image of that width (tr_closest_expand_slots_rows: 8x8 tiles per wave)
    with torch.cuda.device(dev):
        _check(get_module().tr_closest_expand_slots_rows(_handle(accel_structure, packed), packed.data_ptr(), n, int(row_length),
                                                         hit.data_ptr(), front.data_ptr(), tri.data_ptr(), loc.data_ptr(),
                                                         uv.data_ptr(), _stream_ptr(dev)))
    return hit, front, tri, loc, uv


def closest_expand(packed: torch.Tensor, vertices: torch.Tensor, faces: torch.Tensor, batch_shape=None, outs=None):
    """tr_closest_expand: packed [n, 3] int32 -> (hit, front, tri_idx, loc, uv) shaped `batch_shape`
    (default [n]), bit-identical to intersects_closest on the same rays.  vertices / faces: the float32
    [nv, 3] / int32 [nf, 3] arrays the BVH was built from, on the device of `packed`.  `outs`: optional
    preallocated contiguous destinations (bool [n], bool [n], int32 [n], float32 [n, 3], float32 [n, 2]
    -- e.g. row slices of full-size outputs)."""
    if packed.dtype != torch.int32 or packed.dim() != 2 or packed.shape[1] != 3 or not packed.is_contiguous():
        raise ValueError("packed must be a contiguous int32 [n, 3] tensor")
    if vertices.dtype != torch.float32 or faces.dtype != torch.int32 or not vertices.is_contiguous() or not faces.is_contiguous():
        raise ValueError("vertices must be contiguous float32 [nv, 3] and faces contiguous int32 [nf, 3]")
    dev = packed.device
    if not packed.is_cuda or vertices.device != dev or faces.device != dev:
        raise ValueError("packed, vertices and faces must live on the same GPU")
    n = packed.shape[0]
    if outs is not None:
        want = ((torch.bool, (n,)), (torch.bool, (n,)), (torch.int32, (n,)), (torch.float32, (n, 3)), (torch.float32, (n, 2)))
        if len(outs) != 5 or any(t.dtype != dt or tuple(t.shape) != sh or not t.is_contiguous() or t.device != dev
                                 for t, (dt, sh) in zip(outs, want)):
            raise ValueError("outs must be contiguous (bool[n], bool[n], int32[n], float32[n,3], float32[n,2]) on the device of packed")
        with torch.cuda.device(dev):
            _check(get_module().tr_closest_expand(packed.data_ptr(), n, vertices.data_ptr(), vertices.shape[0],
                                                  faces.data_ptr(), faces.shape[0], *(t.data_ptr() for t in outs), _stream_ptr(dev)))
        return tuple(outs)
    b = tuple(batch_shape) if batch_shape is not None else (n,)
    hit = torch.empty(b, dtype=torch.bool, device=dev)
    front = torch.empty(b, dtype=torch.bool, device=dev)
    tri = torch.empty(b, dtype=torch.int32, device=dev)
    loc = torch.empty((*b, 3), dtype=torch.float32, device=dev)
    uv = torch.empty((*b, 2), dtype=torch.float32, device=dev)
    if hit.numel() != n:
        raise ValueError(f"batch_shape {b} does not hold {n} rays")
    with torch.cuda.device(dev):
        _check(get_module().tr_closest_expand(packed.data_ptr(), n, vertices.data_ptr(), vertices.shape[0],
                                              faces.data_ptr(), faces.shape[0], hit.data_ptr(), front.data_ptr(),
                                              tri.data_ptr(), loc.data_ptr(), uv.data_ptr(), _stream_ptr(dev)))
    return hit, front, tri, loc, uv


def intersects_count(accel_structure, origins, dirs) -> torch.Tensor:
    """ops.py:152-168 / ray.cpp:291-322.  Int32[*b]."""
    check_rays(origins, dirs)
    out = torch.empty(origins.shape[:-1], dtype=torch.int32, device=origins.device)
    with torch.cuda.device(origins.device):
        _check(get_module().tr_intersects_count(_handle(accel_structure, origins), C.byref(make_rays(origins, dirs)),
                                                out.data_ptr(), _stream_ptr(origins.device)))
    return out


def intersects_location(accel_structure, origins, dirs, ray_base: int = 0, fused: bool = True) -> Tuple[torch.Tensor, ...]:
    """ops.py:171-192 / ray.cpp:324-378.  (loc[h,3], ray_idx[h], tri_idx[h]); at most
    MAX_ANYHIT_SIZE hits per ray, grouped by ray; within a ray ordered by distance (the
    reference leaves that order unspecified).

    fused=True (default): one traversal (count + slots of the 8 nearest hits), scan, fill from
    the slots.  fused=False: the reference's protocol, count pass -> scan -> second traversal
    (ray.cpp:330, 333-342, 372-374); both give identical results."""
    check_rays(origins, dirs)
    dev = origins.device
    lib = get_module()
    n = origins.numel() // 3
    _check_ray_idx_range(n, ray_base, "intersects_location / intersects_id")
    if fused:
        with torch.cuda.device(dev):
            stream = _stream_ptr(dev)
            rays = make_rays(origins, dirs)
            count = torch.empty(n, dtype=torch.int32, device=dev)
            slots = torch.empty((n, MAX_ANYHIT_SIZE, 2), dtype=torch.int32, device=dev)   # tr_hit_entry {t_key, slot}
            _check(lib.tr_intersects_count_topk(_handle(accel_structure, origins), C.byref(rays), MAX_ANYHIT_SIZE,
                                                count.data_ptr(), slots.data_ptr(), stream))
            offsets = torch.empty(n, dtype=torch.int64, device=dev)
            total_d = torch.empty(1, dtype=torch.int64, device=dev)
            total = C.c_int64(0)
            _check(lib.tr_hits_scan(count.data_ptr(), n, MAX_ANYHIT_SIZE, offsets.data_ptr(), total_d.data_ptr(),
                                    C.byref(total), stream))
            nhits = int(total.value)
            loc = torch.empty((nhits, 3), dtype=torch.float32, device=dev)
            tri = torch.empty(nhits, dtype=torch.int32, device=dev)
            ray = torch.empty(nhits, dtype=torch.int32, device=dev)
            _check(lib.tr_location_fill_slots(_handle(accel_structure, origins), C.byref(rays), MAX_ANYHIT_SIZE,
                                              count.data_ptr(), offsets.data_ptr(), slots.data_ptr(),
                                              loc.data_ptr(), ray.data_ptr(), tri.data_ptr(), ray_base, stream))
        return loc, ray, tri
    with torch.cuda.device(dev):
        stream = _stream_ptr(dev)
        rays = make_rays(origins, dirs)
        count = torch.empty(n, dtype=torch.int32, device=dev)
        _check(lib.tr_intersects_count(_handle(accel_structure, origins), C.byref(rays), count.data_ptr(), stream))
        offsets = torch.empty(n, dtype=torch.int64, device=dev)
        total_d = torch.empty(1, dtype=torch.int64, device=dev)
        total = C.c_int64(0)
        _check(lib.tr_hits_scan(count.data_ptr(), n, MAX_ANYHIT_SIZE, offsets.data_ptr(), total_d.data_ptr(),
                                C.byref(total), stream))   # host sync == ray.cpp:339 .item<int>()
        nhits = int(total.value)
        loc = torch.empty((nhits, 3), dtype=torch.float32, device=dev)
        tri = torch.empty(nhits, dtype=torch.int32, device=dev)
        ray = torch.empty(nhits, dtype=torch.int32, device=dev)
        _check(lib.tr_intersects_location_fill(_handle(accel_structure, origins), C.byref(rays), MAX_ANYHIT_SIZE,
                                               offsets.data_ptr(), loc.data_ptr(), ray.data_ptr(),
                                               tri.data_ptr(), ray_base, stream))
    return loc, ray, tri


def _check_ray_idx_range(n: int, ray_base: int, what: str):
    """ray_idx outputs are int32 by API (ray_optix.py:143-144 `arange(..., dtype=int32)`, ray.cpp:352): a flat ray index of
    2^31 or more does not fit.  The reference's device code overflows silently far earlier (`int` arithmetic on idx * 3
    in getRay, shaders.cu:37: from 715 827 883 rays on); here the traversal is 64-bit throughout (intersects_any / first /
    closest / count take batches of 2^31 rays and more, tests/test_gpu_round6.py) and the queries that RETURN ray indices
    refuse what their output type cannot hold."""
    if n + int(ray_base) >= (1 << 31):
        raise ValueError(f"{what}: {n} rays (+ ray_base {ray_base}) reach 2^31: beyond the int32 ray_idx of the API; split the batch")


def compact_closest(hit, front, tri, loc, uv, ray_base: int = 0):
    """Fused stream compaction (ray_optix.py:142-144: five boolean-mask gathers, one host
    sync each) -> one scan + one gather kernel, one host sync for the size.
    Returns (front[h], ray_idx[h], tri[h], loc[h,3], uv[h,2])."""
    dev = hit.device
    lib = get_module()
    n = hit.numel()
    _check_ray_idx_range(n, ray_base, "stream compaction")
    with torch.cuda.device(dev):
        stream = _stream_ptr(dev)
        offsets = torch.empty(n, dtype=torch.int64, device=dev)
        total_d = torch.empty(1, dtype=torch.int64, device=dev)
        total = C.c_int64(0)
        _check(lib.tr_mask_scan(hit.data_ptr(), n, offsets.data_ptr(), total_d.data_ptr(), C.byref(total), stream))
        h = int(total.value)
        front_o = torch.empty(h, dtype=torch.bool, device=dev) if front is not None else None
        ray_o = torch.empty(h, dtype=torch.int32, device=dev)
        tri_o = torch.empty(h, dtype=torch.int32, device=dev) if tri is not None else None
        loc_o = torch.empty((h, 3), dtype=torch.float32, device=dev) if loc is not None else None
        uv_o = torch.empty((h, 2), dtype=torch.float32, device=dev) if uv is not None else None
        p = lambda t: t.data_ptr() if t is not None else None
        _check(lib.tr_compact_closest(hit.data_ptr(), offsets.data_ptr(), n, p(front), p(tri), p(loc), p(uv),
                                      ray_base, p(front_o), ray_o.data_ptr(), p(tri_o), p(loc_o), p(uv_o), stream))
    return front_o, ray_o, tri_o, loc_o, uv_o


def trace_stats_closest(accel_structure, origins, dirs) -> dict:
    """Diagnostic: per-launch traversal counters of the instrumented closest-hit kernel."""
    check_rays(origins, dirs)
    st = TrTraceStats()
    with torch.cuda.device(origins.device):
        _check(get_module().tr_trace_stats_closest(_handle(accel_structure, origins), C.byref(make_rays(origins, dirs)),
                                                   C.byref(st), _stream_ptr(origins.device)))
    return dict(rays=st.rays, node_visits=st.node_visits, tri_tests=st.tri_tests, climb_steps=st.climb_steps)


QUERY_IDS = {"any": 0, "first": 1, "closest": 2, "count": 3, "location": 4}


def trace_stats(accel_structure, origins, dirs, query: str = "closest") -> dict:
    """Diagnostic: traversal counters of the instrumented kernel of any query."""
    check_rays(origins, dirs)
    st = TrTraceStats()
    with torch.cuda.device(origins.device):
        _check(get_module().tr_trace_stats_query(_handle(accel_structure, origins), C.byref(make_rays(origins, dirs)),
                                           QUERY_IDS[query], C.byref(st), _stream_ptr(origins.device)))
    return dict(rays=st.rays, node_visits=st.node_visits, tri_tests=st.tri_tests, climb_steps=st.climb_steps)


def device_topology(device: int = 0) -> dict:
    """tr_device_topology: what the launch policy reads from the device and the ray-count boundaries it derives"""
    t = TrTopology()
    _check(get_module().tr_device_topology(int(device), C.byref(t)))
    return {k: int(getattr(t, k)) for k, _ in TrTopology._fields_ if k != "reserved"}


def set_option(name: str, value: int):
    _check(get_module().tr_set_option(name.encode(), int(value)))


# --- the native N-rank step (include/triro_rccl.h, csrc/gather_rccl.cpp) --------------------------------------------
_RCCL_LIB_NAME = "libtriro_rccl.so"
_rccl_module = None


class TrShardStep(C.Structure):
    """tr_shard_step of include/triro_rccl.h"""
    _fields_ = [("n_total", C.c_int64), ("world", C.c_int32), ("rank", C.c_int32), ("dst", C.c_int32), ("chunks", C.c_int32),
                ("per_row", C.c_int64), ("bounds", C.POINTER(C.c_int64)), ("my_rays", C.POINTER(TrRays)),
                ("all_rays", C.POINTER(TrRays)), ("d_records", C.c_void_p), ("d_staging", C.c_void_p), ("d_hit", C.c_void_p),
                ("d_front", C.c_void_p), ("d_tri", C.c_void_p), ("d_loc3", C.c_void_p), ("d_uv2", C.c_void_p),
                ("stream", C.c_void_p), ("side_stream", C.c_void_p), ("done_event", C.c_void_p), ("flags", C.c_int32)]


STEP_NO_EXCHANGE, STEP_LOOPBACK, STEP_TEST_DROP_SEND = 1, 2, 4
COMM_ID_BYTES = 128


def rccl_library_path() -> str:
    return os.environ.get("TRIRO_RCCL_LIBRARY") or os.path.join(os.path.dirname(library_path()), _RCCL_LIB_NAME)


def get_rccl_module():
    """libtriro_rccl.so (one C call per pipelined step of a ray-sharded closest-hit query).  Raises when it is not built;
    whether RCCL itself can be found is a question for rccl_available()."""
    global _rccl_module
    if _rccl_module is None:
        get_module()                       # libtriro_hip.so first: the step library links against it
        path = rccl_library_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: build it (make -C trimesh-ray-optix_amd/csrc all, __graft_entry__.build())")
        lib = C.CDLL(path)
        lib.tr_rccl_last_error.restype = C.c_char_p
        lib.tr_rccl_available.restype = _int
        lib.tr_comm_unique_id.argtypes = [C.c_void_p]
        lib.tr_comm_create.argtypes = [C.c_void_p, _int, _int, _int, C.POINTER(_vp)]
        lib.tr_comm_destroy.argtypes = [_vp]
        lib.tr_comm_abort.argtypes = [_vp]
        lib.tr_sharded_closest_step.argtypes = [_vp, _vp, C.POINTER(TrShardStep)]
        for f in ("tr_comm_unique_id", "tr_comm_create", "tr_comm_destroy", "tr_comm_abort", "tr_sharded_closest_step"):
            getattr(lib, f).restype = _int
        _rccl_module = lib
    return _rccl_module


def rccl_available() -> bool:
    try:
        return get_rccl_module().tr_rccl_available() == 0
    except Exception:
        return False


def _check_rccl(rc: int):
    if rc != 0:
        raise RuntimeError("libtriro_rccl: " + (get_rccl_module().tr_rccl_last_error() or b"?").decode())
