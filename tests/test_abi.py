"""CPU checks of the drop-in boundary: libtriro_hip.so loads, exports every symbol the header
declares, the ctypes table covers the header, the Python surface mirrors the reference's, and
the product path fails loudly (no CPU fallback) when there is no GPU."""
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "triro_hip.h")


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tr_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    syms = header_symbols()
    for s in ["tr_init", "tr_bvh_build", "tr_bvh_update", "tr_bvh_destroy", "tr_intersects_any",
              "tr_intersects_first", "tr_intersects_closest", "tr_intersects_count", "tr_hits_scan",
              "tr_intersects_location_fill", "tr_mask_scan", "tr_compact_closest", "tr_last_error"]:
        assert s in syms


def test_library_exports_every_declared_symbol():
    import ctypes
    import triro.backend.ops as hops
    path = hops.library_path()
    assert os.path.exists(path), "build with __graft_entry__.build()"
    lib = ctypes.CDLL(path)
    for s in header_symbols():
        assert hasattr(lib, s), f"{s} declared in include/triro_hip.h but not exported"
    assert set(hops.ABI) == set(header_symbols())
    assert hops.get_module().tr_abi_version() == hops.ABI_VERSION


def test_abi_version_constant_matches_header_and_build_entry():
    """The binding's ABI_VERSION is the header's TR_ABI_VERSION, and __graft_entry__.build() (the
    driver's build check) passes with it."""
    import re
    import triro.backend.ops as hops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "triro_hip.h")).read()
    assert int(re.search(r"#define\s+TR_ABI_VERSION\s+(\d+)", hdr).group(1)) == hops.ABI_VERSION
    import __graft_entry__ as g
    g.build()


def test_struct_layout_matches_header():
    import ctypes
    import triro.backend.ops as hops
    assert ctypes.sizeof(hops.TrRays) == 8 + 8 + 8 + 3 * 32      # LaunchParams.h:11-28 minus hit ptrs
    assert ctypes.sizeof(hops.TrTraceStats) == 32


def test_public_surface_matches_reference():
    from triro.ray.ray_optix import OptixAccelStructureWrapper, RayMeshIntersector
    import triro
    assert triro.__version__.startswith("1.3.1")
    expected = {
        "update_raw": ["vertices", "faces"],
        "intersects_any": ["origins", "directions"],
        "intersects_first": ["origins", "directions"],
        "intersects_closest": ["origins", "directions", "stream_compaction"],
        "intersects_location": ["origins", "directions"],
        "intersects_count": ["origins", "directions"],
        "intersects_id": ["origins", "directions", "return_locations", "multiple_hits"],
        "contains_points": ["points", "check_direction"],
    }
    for name, args in expected.items():
        sig = inspect.signature(getattr(RayMeshIntersector, name))
        assert list(sig.parameters)[1:1 + len(args)] == args, name
    assert inspect.signature(RayMeshIntersector.intersects_closest).parameters["stream_compaction"].default is False
    p = inspect.signature(RayMeshIntersector.intersects_id).parameters
    assert p["return_locations"].default is False and p["multiple_hits"].default is True
    assert hasattr(OptixAccelStructureWrapper, "build_accel_structure")
    with pytest.raises(ValueError):
        RayMeshIntersector()                       # ray_optix.py:40-41


def test_ops_shims_exist():
    import triro.backend.ops as hops
    for n in ["get_module", "init_optix", "create_optix_context", "create_optix_module",
              "create_optix_pipelines", "build_sbts", "intersects_any", "intersects_first",
              "intersects_closest", "intersects_count", "intersects_location"]:
        assert callable(getattr(hops, n))


def test_input_validation_raises():
    import triro.backend.ops as hops
    o = torch.zeros(4, 3)
    with pytest.raises(ValueError):
        hops.check_rays(o, o)                      # CPU tensors
    r = hops.make_rays(torch.zeros(2, 5, 3), torch.zeros(2, 5, 3).expand(2, 5, 3))
    assert list(r.shape) == [(1 << 63) - 1, 2, 5, 3] and list(r.ostride) == [0, 15, 3, 1]
    b = hops.make_rays(torch.zeros(1, 3).expand(7, 3), torch.zeros(7, 3))
    assert list(b.ostride) == [0, 0, 0, 1] and b.nray == 7


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_no_cpu_fallback():
    import numpy as np
    from triro.ray.ray_optix import RayMeshIntersector
    v = np.zeros((3, 3), np.float32)
    f = np.array([[0, 1, 2]], np.int32)
    with pytest.raises(RuntimeError):
        RayMeshIntersector(vertices=torch.from_numpy(v), faces=torch.from_numpy(f))


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "trimesh-ray-optix_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert "libtriro_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, fn


def test_every_option_is_documented():
    """every tr_set_option name of csrc/api.hip appears in the C header and in INTEGRATION.md"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    api = open(os.path.join(root, "trimesh-ray-optix_amd", "csrc", "api.hip")).read()
    names = re.findall(r'\{"(\w+)", &tr_options::', api)
    assert len(names) >= 15
    hdr = open(os.path.join(root, "include", "triro_hip.h")).read()
    integ = open(os.path.join(root, "INTEGRATION.md")).read()
    assert [n for n in names if f'"{n}"' not in hdr] == []
    assert [n for n in names if f"`{n}`" not in integ] == []


def test_native_step_library_exports_every_declared_symbol():
    """include/triro_rccl.h <-> libtriro_rccl.so (no compute, no RCCL call: the symbols and the struct layout only)"""
    import ctypes
    import re
    import triro.backend.ops as hops
    path = hops.rccl_library_path()
    assert os.path.exists(path), "build with __graft_entry__.build()"
    lib = ctypes.CDLL(hops.library_path(), mode=ctypes.RTLD_GLOBAL) and ctypes.CDLL(path)
    hdr = open(os.path.join(ROOT, "include", "triro_rccl.h")).read()
    names = set(re.findall(r"\b(tr_[a-z_0-9]+)\s*\(", hdr))
    assert {"tr_rccl_available", "tr_comm_unique_id", "tr_comm_create", "tr_comm_destroy", "tr_sharded_closest_step", "tr_rccl_last_error"} <= names
    for n in names:
        assert hasattr(lib, n), n
    # tr_shard_step: 8 + 4*4 + 8 + 3*8 (pointers) + ... = the layout the binding assumes
    assert ctypes.sizeof(hops.TrShardStep) == 8 + 16 + 8 + 8 * 13 + 8      # (the trailing int32 flags is padded to 8)
    assert hops.TrShardStep.per_row.offset == 24 and hops.TrShardStep.bounds.offset == 32 and hops.TrShardStep.flags.offset == 136
