"""CPU tests of the synthetic workloads (tests/bench inputs, not product code): the mesh-file loader
behind $TRIRO_BUNNY (the Stanford bunny if a file is supplied) and the interior scene."""
import os
import struct

import numpy as np

import workloads as W
from oracle.oracle import OracleIntersector


def _write_meshes(tmp, v, f):
    with open(os.path.join(tmp, "m.obj"), "w") as fh:
        for p in v:
            fh.write(f"v {p[0]:.9g} {p[1]:.9g} {p[2]:.9g}\n")
        for t in f:
            fh.write(f"f {t[0] + 1}/1/1 {t[1] + 1}/2/2 {t[2] + 1}/3/3\n")
    hdr = f"ply\nformat {{}} 1.0\ncomment test\nelement vertex {len(v)}\nproperty float x\nproperty float y\nproperty float z\nproperty float confidence\nelement face {len(f)}\nproperty list uchar int vertex_indices\nend_header\n"
    with open(os.path.join(tmp, "a.ply"), "w") as fh:
        fh.write(hdr.format("ascii"))
        for p in v:
            fh.write(f"{p[0]:.9g} {p[1]:.9g} {p[2]:.9g} 0.5\n")
        for t in f:
            fh.write(f"3 {t[0]} {t[1]} {t[2]}\n")
    with open(os.path.join(tmp, "b.ply"), "wb") as fh:
        fh.write(hdr.format("binary_little_endian").encode())
        for p in v:
            fh.write(struct.pack("<4f", p[0], p[1], p[2], 0.5))
        for t in f:
            fh.write(struct.pack("<B3i", 3, *[int(x) for x in t]))


def test_mesh_file_loader_and_bunny_env(tmp_path, monkeypatch):
    v, f = W.icosphere(2)
    _write_meshes(str(tmp_path), v, f)
    for name in ("m.obj", "a.ply", "b.ply"):
        v2, f2 = W.load_mesh_file(os.path.join(str(tmp_path), name))
        assert v2.dtype == np.float32 and f2.dtype == np.int32
        assert np.array_equal(f2, f) and np.allclose(v2, v, rtol=0, atol=1e-7), name
    monkeypatch.delenv("TRIRO_BUNNY", raising=False)
    vs, fs, label = W.bunny_mesh()
    assert len(fs) == 81920 and "stand-in" in label
    monkeypatch.setenv("TRIRO_BUNNY", os.path.join(str(tmp_path), "b.ply"))
    vb, fb, label = W.bunny_mesh()
    assert np.array_equal(fb, f) and np.array_equal(vb, v) and "b.ply" in label and "stand-in" not in label
    # a quad face is fanned into two triangles
    with open(os.path.join(str(tmp_path), "q.obj"), "w") as fh:
        fh.write("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")
    vq, fq = W.load_mesh_file(os.path.join(str(tmp_path), "q.obj"))
    assert fq.tolist() == [[0, 1, 2], [0, 2, 3]]


def test_interior_room_is_closed_and_deep():
    """workloads.interior_room at low detail: every ray cast from the inside camera hits, first hits are
    front faces (walls face inwards, furniture outwards), and rays cross several surfaces."""
    v, f = W.interior_room(detail=0.02)
    assert 10_000 < len(f) < 40_000 and int(f.max()) < len(v)
    o, d = W.ref_shape_rays(W.INTERIOR_EYE, W.INTERIOR_TARGET, 160, 90, 111.0)
    assert o.shape == (90, 160, 3) and o.strides[:2] == (0, 0)
    assert np.allclose(np.linalg.norm(d, axis=-1), 1.0, atol=1e-6)
    R = OracleIntersector(v, f, 1)
    hit, front, tri, loc, uv, t = R.closest_raw(np.ascontiguousarray(o.reshape(-1, 3)), d.reshape(-1, 3))
    assert hit.all() and front.all()
    cnt = R.intersects_count(np.ascontiguousarray(o.reshape(-1, 3)), d.reshape(-1, 3))
    assert cnt.mean() > 4.5 and cnt.max() > 8
    assert np.all(np.abs(loc) <= np.array([4.001, 3.001, 3.001]))


def test_terrain_is_deterministic_and_open():
    v, f = W.terrain(64)
    v2, f2 = W.terrain(64)
    assert np.array_equal(v, v2) and np.array_equal(f, f2)
    assert v.dtype == np.float32 and f.dtype == np.int32 and len(f) == 2 * 64 * 64 and len(v) == 65 * 65
    # upward-facing: every triangle normal has a positive y component
    a, b, c = v[f[:, 0]].astype(np.float64), v[f[:, 1]].astype(np.float64), v[f[:, 2]].astype(np.float64)
    assert (np.cross(b - a, c - a)[:, 1] > 0).all()
    assert len(W.terrain()[1]) == 1048352
    # the default camera sees both ground and sky
    from oracle.oracle import OracleIntersector
    o, d = W.ref_shape_rays(W.TERRAIN_EYE, W.TERRAIN_TARGET, w=160, h=90, f=111.0)
    hit = OracleIntersector(*W.terrain(128), 1).closest_raw(np.ascontiguousarray(o.reshape(-1, 3)), d.reshape(-1, 3))[0]
    assert 0.3 < hit.mean() < 0.8 and not hit.reshape(90, 160)[:20].any() and hit.reshape(90, 160)[-20:].all()
